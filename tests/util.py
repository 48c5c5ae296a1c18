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
def _tok(s):
    return s.encode() + b" "


def _i32(v):
    return b"\x04" + np.int32(v).tobytes()


def _f32(v):
    return b"\x04" + np.float32(v).tobytes()


def _ivec(v):
    v = np.asarray(v, "<i4")
    return b"\x04" + np.int32(len(v)).tobytes() + v.tobytes()


def _fvec(v):
    v = np.asarray(v, "<f4")
    return b"FV " + _i32(len(v)) + v.tobytes()


def _fmat(m):
    m = np.asarray(m, "<f4")
    return b"FM " + _i32(m.shape[0]) + _i32(m.shape[1]) + m.tobytes()


def write_mdl(path, n_cd, hmm_states, offs, weights, miv, iv, self_p=0.6):
    """Binary .mdl: `n_cd` phones, each a left-to-right HMM of `hmm_states` emitting
    states (self-loop then forward transition) with pdf = phone_index * hmm_states + state, i.e.
    the transition-ids synth.lexicon_graph() uses (pdf p: self-loop 2p+1, forward 2p+2); then the
    diagonal GMMs.  Readable by the reference (TransitionModel::Read, AmDiagGmm::Read) and by
    vbio::ReadModel."""
    H = hmm_states
    out = b"\0B" + _tok("<TransitionModel>") + _tok("<Topology>")
    phones = list(range(1, n_cd + 1))
    out += _ivec(phones) + _ivec([-1] + [0] * n_cd)
    out += _i32(1)                      # one topology entry shared by all phones
    out += _i32(H + 1)
    for k in range(H):
        out += _i32(k) + _i32(2) + _i32(k) + _f32(self_p) + _i32(k + 1) + _f32(1.0 - self_p)
    out += _i32(-1) + _i32(0)           # final non-emitting state
    out += _tok("</Topology>") + _tok("<Triples>") + _i32(n_cd * H)
    for ph in range(n_cd):
        for k in range(H):
            out += _i32(ph + 1) + _i32(k) + _i32(ph * H + k)
    out += _tok("</Triples>") + _tok("<LogProbs>")
    lp = [0.0] + [np.log(self_p), np.log(1.0 - self_p)] * (n_cd * H)
    out += _fvec(lp) + _tok("</LogProbs>") + _tok("</TransitionModel>")
    P = len(offs) - 1
    D = miv.shape[1]
    out += _tok("<DIMENSION>") + _i32(D) + _tok("<NUMPDFS>") + _i32(P)
    for p in range(P):
        a, b = int(offs[p]), int(offs[p + 1])
        out += _tok("<DiagGMM>") + _tok("<WEIGHTS>") + _fvec(weights[a:b]) + _tok("<MEANS_INVVARS>") + _fmat(miv[a:b])
        out += _tok("<INV_VARS>") + _fmat(iv[a:b]) + _tok("</DiagGMM>")
    with open(path, "wb") as f:
        f.write(out)


def write_vector_fst(path, final, arc_offset, arcs, start=0):
    """OpenFst VectorFst<StdArc> binary file from the graph arrays of synth.*_graph()."""
    S = len(final)
    a = np.asarray(arcs)

    def s_(x):
        return np.int32(len(x)).tobytes() + x
    out = np.int32(2125659606).tobytes() + s_(b"vector") + s_(b"standard") + np.int32(2).tobytes() + np.int32(0).tobytes()
    out += np.uint64(3).tobytes() + np.int64(start).tobytes() + np.int64(S).tobytes() + np.int64(len(a)).tobytes()
    parts = [out]
    for s in range(S):
        lo, hi = int(arc_offset[s]), int(arc_offset[s + 1])
        parts.append(np.float32(final[s]).tobytes() + np.int64(hi - lo).tobytes() + a[lo:hi].tobytes())
    with open(path, "wb") as f:
        f.write(b"".join(parts))


def write_feats_ark(path, feats):
    with open(path, "wb") as f:
        for k, m in feats.items():
            m = np.asarray(m, "<f4")
            f.write(k.encode() + b" \0B" + _fmat(m))


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
