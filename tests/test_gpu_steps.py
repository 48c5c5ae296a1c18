"""GPU: the step API (voicebridge_amd/host/voicebridge_steps.cc; seam B1 of SURVEY.md 8b) on a data
directory: MakeMfcc -> ComputeCmvnStats -> Decode, checked against the bin functions run by hand and
against the reference's own programs where they are present (oracle/_ref/bin)."""
import ctypes as C
import os

import numpy as np
import pytest

from tests.test_gpu_binfuncs import REFBIN, call, have_ref, ref, write_wavs
from tests.test_host_io import read_fm_ark
from tests.util import load

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def host():
    from voicebridge_amd import build
    lib = C.CDLL(build.HOST_LIB)
    lib.vb_steps_last_error.restype = C.c_char_p
    lib.vb_Decode.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_int, C.c_float, C.c_int, C.c_double, C.c_double, C.c_int]
    return lib


def make_datadir(d, n_utt=6, n_spk=3):
    from voicebridge_amd import synth
    data = d / "data" / "test"
    data.mkdir(parents=True)
    waves = [synth.wave(100 + i, seconds=0.6 + 0.15 * i) for i in range(n_utt)]
    write_wavs(data, waves)
    utts = ["utt%02d" % i for i in range(n_utt)]
    spk = ["spk%d" % (i * n_spk // n_utt) for i in range(n_utt)]
    (data / "utt2spk").write_text("".join("%s %s\n" % (u, s) for u, s in zip(utts, spk)))
    s2u = {}
    for u, s in zip(utts, spk):
        s2u.setdefault(s, []).append(u)
    (data / "spk2utt").write_text("".join("%s %s\n" % (s, " ".join(us)) for s, us in s2u.items()))
    conf = d / "mfcc.conf"
    conf.write_text("--use-energy=false\n--dither=0\n")
    return data, conf, utts


def test_make_mfcc_cmvn_and_decode_steps(host, tmp_path):
    d = tmp_path
    data, conf, utts = make_datadir(d)
    # ---- MakeMfcc: nj = 2, compressed archives, utt2num_frames
    rc = host.vb_MakeMfcc(str(data).encode(), str(conf).encode(), 2, 1, 1)
    assert rc == 0, host.vb_steps_last_error()
    feats_scp = (data / "feats.scp").read_text().split("\n")[:-1]
    assert [l.split()[0] for l in feats_scp] == utts
    for j in (1, 2):
        assert (data / "mfcc" / ("raw_mfcc_test.%d.ark" % j)).exists() and (data / "log" / ("make_mfcc_test.%d.log" % j)).exists()
    assert not list((data / "log").glob("*.mfctemp"))
    n_frames = dict(l.split() for l in (data / "utt2num_frames").read_text().split("\n")[:-1])
    # the same features as the bin functions by hand (uncompressed), up to the codec's quantisation
    assert call(host.vb_ComputeMFCCFeats, ["compute-mfcc-feats", "--config=%s" % conf, "scp:%s" % (data / "wav.scp"), "ark:%s" % (d / "raw.ark")], "") == 0
    assert call(host.vb_CopyFeats, ["copy-feats", "scp:%s" % (data / "feats.scp"), "ark:%s" % (d / "back.ark")], "") == 0
    raw, back = read_fm_ark(str(d / "raw.ark")), read_fm_ark(str(d / "back.ark"))
    assert list(raw) == utts and list(back) == utts
    for u in utts:
        assert raw[u].shape == back[u].shape and int(n_frames[u]) == raw[u].shape[0]
        rng = raw[u].max(0) - raw[u].min(0)
        assert np.all(np.abs(raw[u] - back[u]) <= rng / 40.0 + 1e-4)      # (quartile-wise 8-bit codec)
    if have_ref("copy-feats"):      # the reference reads our compressed archives to the same numbers
        ref("copy-feats", "scp:%s" % (data / "feats.scp"), "ark:%s" % (d / "ref_back.ark"))
        rb = read_fm_ark(str(d / "ref_back.ark"))
        for u in utts:
            np.testing.assert_array_equal(rb[u], back[u])
    # a second run moves feats.scp to backup/
    assert host.vb_MakeMfcc(str(data).encode(), str(conf).encode(), 3, 0, 0) == 0
    assert (data / "backup" / "feats.scp").exists()
    # ---- ComputeCmvnStats
    assert host.vb_ComputeCmvnStatsDir(str(data).encode()) == 0, host.vb_steps_last_error()
    cm = (data / "cmvn.scp").read_text().split("\n")[:-1]
    assert [l.split()[0] for l in cm] == ["spk0", "spk1", "spk2"] and (data / "data" / "cmvn_test.ark").exists()
    # ---- SplitData keeps speakers together
    assert host.vb_SplitData(str(data).encode(), 2, 0) == 0
    parts = [(data / "split2" / str(j) / "utt2spk").read_text().split("\n")[:-1] for j in (1, 2)]
    assert sum(len(p) for p in parts) == len(utts) and all(parts)
    assert not ({l.split()[1] for l in parts[0]} & {l.split()[1] for l in parts[1]})
    # ---- Decode: model directory exp/tri with final.mdl, graph directory with HCLG.fst
    g = load("files")
    exp = d / "exp" / "tri"
    graph = exp / "graph"
    graph.mkdir(parents=True)
    (exp / "final.mdl").write_bytes(g["file_final_mdl"].tobytes())
    (graph / "HCLG.fst").write_bytes(g["file_HCLG_fst"].tobytes())
    dec = exp / "decode_test"
    # scoring is not part of this build: refused unless switched off
    assert host.vb_Decode(str(graph).encode(), str(data).encode(), str(dec).encode(), b"", 2, 0.083333, 7000, 13.0, 6.0, 0) == -1
    assert b"scoring" in host.vb_steps_last_error()
    rc = host.vb_Decode(str(graph).encode(), str(data).encode(), str(dec).encode(), b"", 2, 0.083333, 7000, 13.0, 6.0, 1)
    assert rc == 0, host.vb_steps_last_error()
    assert (dec / "num_jobs").read_text().strip() == "2"
    for j in (1, 2):
        assert (dec / ("lat.%d" % j)).stat().st_size > 0 and "Overall log-likelihood" in (dec / "log" / ("decode.%d.log" % j)).read_text()
        assert not (data / "split2" / str(j) / "apply_cmvn.temp").exists()
    if have_ref("lattice-copy"):    # the reference reads the lattice archives the step wrote
        for j in (1, 2):
            ref("lattice-copy", "ark:%s" % (dec / ("lat.%d" % j)), "ark,t:%s" % (d / ("lat%d.txt" % j)))
        got = (d / "lat1.txt").read_text() + (d / "lat2.txt").read_text()
        assert all(u in got for u in utts)
