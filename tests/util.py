"""Helpers shared by the tests (fixture loading, option parsing)."""
import os

import numpy as np

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

MFCC_INT = {"remove_dc_offset", "round_to_power_of_two", "snip_edges", "num_bins", "htk_mode",
            "num_ceps", "use_energy", "raw_energy", "htk_compat"}


def load(name):
    return np.load(os.path.join(GOLD, name + ".npz"), allow_pickle=False)


def parse_opts(arr):
    """['k=v', ...] -> dict with numbers converted."""
    out = {}
    for s in arr:
        k, v = str(s).split("=", 1)
        try:
            out[k] = int(v)
        except ValueError:
            try:
                out[k] = float(v)
            except ValueError:
                out[k] = v
    return out


def mfcc_kwargs(opts):
    """Split a fixture's option dict into (struct field kwargs, vtln_warp)."""
    o = dict(opts)
    vtln = float(o.pop("vtln_warp", 1.0))
    kw = {}
    for k, v in o.items():
        if k in MFCC_INT:
            kw[k] = int(v)
        elif k == "window_type":
            kw[k] = v
        else:
            kw[k] = float(v)
    return kw, vtln


def read_htk(path):
    """HTK parameter file: 12-byte big-endian header then big-endian float32 rows."""
    raw = open(path, "rb").read()
    n = int.from_bytes(raw[0:4], "big")
    size = int.from_bytes(raw[8:10], "big")
    d = size // 4
    return np.frombuffer(raw[12:12 + n * size], dtype=">f4").reshape(n, d).astype(np.float32)


def read_wav_pcm16(path):
    """Minimal RIFF reader for the test fixtures (mono PCM-16)."""
    raw = open(path, "rb").read()
    assert raw[:4] == b"RIFF" and raw[8:12] == b"WAVE"
    pos = 12
    sr = None
    while pos < len(raw):
        cid = raw[pos:pos + 4]
        sz = int.from_bytes(raw[pos + 4:pos + 8], "little")
        if cid == b"fmt ":
            sr = int.from_bytes(raw[pos + 12:pos + 16], "little")
        if cid == b"data":
            sz = min(sz, len(raw) - pos - 8)
            return sr, np.frombuffer(raw[pos + 8:pos + 8 + sz - sz % 2], dtype="<i2")
        pos += 8 + sz + (sz & 1)
    raise ValueError("no data chunk")


# ---------------------------------------------------------------- files for the lattice tests
from voicebridge_amd.refio import write_const_fst, write_feats_ark, write_mdl, write_vector_fst, write_wav  # noqa: E402,F401


def read_lattice_ark_bytes(raw):
    """read_lattice_ark of an archive held in memory"""
    import tempfile
    with tempfile.NamedTemporaryFile(suffix=".ark") as f:
        f.write(raw)
        f.flush()
        return read_lattice_ark(f.name)


def read_lattice_ark(path):
    """Binary lattice archive -> {key: (kind, start, states)}; states = list of (final, arcs); for
    'lattice4' final = (graph, acoustic), arcs = (ilabel, olabel, graph, acoustic, next); for
    'compactlattice44' final = (graph, acoustic, tids), arcs = (label, graph, acoustic, tids, next).
    Floats are kept as their 32-bit patterns' values (np.float32) so comparisons are exact."""
    raw = open(path, "rb").read()
    pos = 0
    out = {}

    def take(fmt, n=1):
        nonlocal pos
        v = np.frombuffer(raw, fmt, n, pos)
        pos += v.nbytes
        return v if n != 1 else v[0]

    def fstr():
        nonlocal pos
        n = int(take("<i4"))
        s = raw[pos:pos + n]
        pos += n
        return s.decode()
    while pos < len(raw):
        sp = raw.index(b" ", pos)
        key = raw[pos:sp].decode()
        pos = sp + 1
        assert int(take("<i4")) == 2125659606
        fsttype, arctype = fstr(), fstr()
        assert fsttype == "vector"
        take("<i4"), take("<i4"), take("<u8")
        start, ns, _na = int(take("<i8")), int(take("<i8")), int(take("<i8"))
        states = []
        for _ in range(ns):
            if arctype == "lattice4":
                fin = tuple(take("<f4", 2).tolist())
                n = int(take("<i8"))
                arcs = []
                for _ in range(n):
                    il, ol = take("<i4", 2).tolist()
                    g, a = take("<f4", 2).tolist()
                    arcs.append((il, ol, g, a, int(take("<i4"))))
            else:
                assert arctype == "compactlattice44", arctype

                def cw():
                    g, a = take("<f4", 2).tolist()
                    k = int(take("<i4"))
                    return g, a, tuple(take("<i4", k).tolist()) if k != 1 else (int(take("<i4")),)
                fin = cw()
                n = int(take("<i8"))
                arcs = []
                for _ in range(n):
                    il, _ol = take("<i4", 2).tolist()
                    g, a, tids = cw()
                    arcs.append((il, g, a, tids, int(take("<i4"))))
            states.append((fin, arcs))
        out[key] = (arctype, start, states)
    return out


# the search's admission rule (vbh_decode_cfg::admission) and the oracle mode that restates it (oracle/vb_oracle.h, vbo_lat_cfg::exact_cutoff):
# "ordered" = the reference's online tightening as a prefix minimum over 64 buckets of the source state's hash (the product's default),
# "exact" = tot <= the frame's final next_cutoff (rounds 1-3)
ORACLE_ADMISSION = {"ordered": dict(exact_cutoff=3, adm_bucket_bits=6), "exact": dict(exact_cutoff=1)}
ADMISSIONS = ("ordered", "exact")
