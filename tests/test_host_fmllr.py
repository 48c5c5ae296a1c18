"""Host: the programs between the two searches of DecodeFmllr (voicebridge_amd/host/voicebridge_fmllrbin.cc,
fmllr.cc, lattice.cc) against a golden run of the reference's own programs
(tests/golden/fmllr.npz, made by oracle/gen_golden_lat.py) and, where oracle/_ref/bin is present,
against those programs on a fresh case.

Bars: lattice posteriors and silence weighting are bit-exact (archives byte-identical).  Everything
downstream of per-Gaussian log-likelihoods differs from the reference in float summation order
(the reference calls BLAS sgemv) and, for the transforms, in the route to the matrix inverse (LAPACK
there, Gauss-Jordan here): Gaussian posteriors 1e-5 absolute, fMLLR matrices 2e-4 absolute (entries
O(1)), rescored lattices identical in structure with acoustic costs within 1e-6 relative."""
import ctypes as C
import os

import numpy as np
import pytest

from tests import util
from tests.test_host_io import read_fm_ark
from tests.test_host_lattice import REFBIN, assert_same_archives, call, host  # noqa: F401

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NEEDED = ("gmm-latgen-faster", "lattice-determinize-pruned", "lattice-to-post", "weight-silence-post", "gmm-post-to-gpost",
          "gmm-est-fmllr-gpost", "gmm-est-fmllr", "compose-transforms", "gmm-rescore-lattice")


def read_gpost_ark(path):
    raw = open(path, "rb").read()
    pos, out = 0, {}

    def i32():
        nonlocal pos
        assert raw[pos] == 4
        v = int(np.frombuffer(raw, "<i4", 1, pos + 1)[0])
        pos += 5
        return v
    while pos < len(raw):
        sp = raw.index(b" ", pos)
        key = raw[pos:sp].decode()
        pos = sp + 1
        assert raw[pos:pos + 2] == b"\0B"
        pos += 2
        frames = []
        for _ in range(i32()):
            fr = []
            for _ in range(i32()):
                pdf = i32()
                assert raw[pos:pos + 3] == b"FV "
                pos += 3
                n = i32()
                fr.append((pdf, np.frombuffer(raw, "<f4", n, pos).copy()))
                pos += 4 * n
            frames.append(fr)
        out[key] = frames
    return out


def mine_runner(host, d):
    fns = {"lattice-to-post": host.vb_LatticeToPost, "weight-silence-post": host.vb_WeightSilencePost, "gmm-post-to-gpost": host.vb_GmmPostToGpost,
           "gmm-est-fmllr-gpost": host.vb_GmmEstFmllrGpost, "gmm-est-fmllr": host.vb_GmmEstFmllr, "compose-transforms": host.vb_ComposeTransforms,
           "gmm-rescore-lattice": host.vb_GmmRescoreLattice}

    def run(name, *args):
        log = str(d / "m_log.txt")
        assert call(fns[name], [name] + list(args), log) == 0, name + ": " + open(log).read()[-600:]
    return run


def compare_chain(d, mine="m_", want=""):
    p = lambda n: str(d / n)
    assert open(p(mine + "post.ark"), "rb").read() == open(p(want + "post.ark"), "rb").read()       # bit-exact
    assert open(p(mine + "wsp.ark"), "rb").read() == open(p(want + "wsp.ark"), "rb").read()
    a, b = read_gpost_ark(p(mine + "gpost.ark")), read_gpost_ark(p(want + "gpost.ark"))
    assert list(a) == list(b)
    for k in b:
        assert len(a[k]) == len(b[k])
        for fa, fb in zip(a[k], b[k]):
            assert sorted(x[0] for x in fa) == sorted(x[0] for x in fb)
            db = dict(fb)
            for pdf, v in fa:
                np.testing.assert_allclose(v, db[pdf], atol=1e-5, rtol=1e-4)
    for t in ("pre_trans", "trans_tmp", "trans"):
        a, b = read_fm_ark(p(mine + t + ".ark")), read_fm_ark(p(want + t + ".ark"))
        assert list(a) == list(b) == ["spk1", "spk2"]
        for k in b:
            assert a[k].shape == (39, 40)
            np.testing.assert_allclose(a[k], b[k], atol=2e-4, rtol=0)
            assert np.abs(b[k] - np.eye(39, 40)).max() > 0.05          # a real update, not the unit start
    assert_same_archives(p(mine + "grl.ark"), p(want + "grl.ark"), rtol=1e-6, atol=1e-4)


def test_fmllr_chain_matches_golden(host, tmp_path):
    import sys
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import gen_golden_lat as gg
    d = tmp_path
    g = util.load("fmllr")
    for k in g.files:
        if k.startswith("file_"):
            name = k[5:].rsplit("_", 1)
            (d / (name[0] + "." + name[1])).write_bytes(g[k].tobytes())
    (d / "spk2utt").write_text(str(g["text_spk2utt"]))
    gg.run_fmllr_chain(str(d), mine_runner(host, d), "m_", trans_tmp_from="")      # the estimation programs and gmm-rescore-lattice: tests/test_gpu_fmllr.py
    compare_chain(d)


@pytest.mark.skipif(not all(os.access(os.path.join(REFBIN, b), os.X_OK) for b in NEEDED), reason="oracle/_ref/bin not built")
def test_fresh_case_vs_reference_programs(host, tmp_path):
    import sys
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import gen_golden_lat as gg
    d = tmp_path
    gg.make_fmllr_case(str(d), seed=17, frames=(210, 95, 160, 140))
    gg.run_fmllr_chain(str(d), gg.ref, "r_")
    gg.run_fmllr_chain(str(d), mine_runner(host, d), "m_", trans_tmp_from="r_")
    compare_chain(d, "m_", "r_")
