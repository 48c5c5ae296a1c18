"""GPU parity: fMLLR sufficient statistics (vbh_fmllr_accumulate, voicebridge_amd/csrc/vbh_fmllr.hip) against
a double-precision restatement of FmllrDiagGmmAccs (K/transform/fmllr-diag-gmm.cc:30-44,100-113,562-596)
in numpy, and -- through the bin function GmmEstFmllr, whose statistics come from this kernel -- against the
transforms of the reference's gmm-est-fmllr (tests/golden/fmllr.npz).

Tolerances: the per-frame terms a, b are float sums in a different order than the reference's BLAS calls and
the products are accumulated in double in a different order: statistics 1e-5 relative to their scale,
transforms 2e-4 absolute (entries O(1))."""
import ctypes as C
import os

import numpy as np
import pytest

from tests import util

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def ctx():
    from voicebridge_amd import api
    c = api.Context(0)
    yield c
    c.close()


def numpy_stats(offs, gconsts, miv, iv, feats, foff, utt2spk, n_spk, poff, ppdf, pw, ali=None):
    D = feats.shape[1]
    a_g, a_m, a_i = ali if ali is not None else (gconsts, miv, iv)
    beta = np.zeros(n_spk); K = np.zeros((n_spk, D, D + 1)); G = np.zeros((n_spk, D, D + 1, D + 1))
    for u in range(len(foff) - 1):
        s = utt2spk[u]
        if s < 0:
            continue
        for t in range(foff[u], foff[u + 1]):
            x = feats[t].astype(np.float64)
            xp = np.append(x, 1.0)
            a = np.zeros(D); b = np.zeros(D); cnt = 0.0
            for q in range(poff[t], poff[t + 1]):
                p = ppdf[q]
                g0, g1 = offs[p], offs[p + 1]
                ll = a_g[g0:g1] + a_m[g0:g1].astype(np.float64) @ x - 0.5 * (a_i[g0:g1].astype(np.float64) @ (x * x))
                post = np.exp(ll - ll.max()); post = post / post.sum() * pw[q]
                cnt += post.sum(); a += post @ miv[g0:g1]; b += post @ iv[g0:g1]
            beta[s] += cnt
            K[s] += np.outer(a, xp)
            G[s] += b[:, None, None] * np.outer(xp, xp)[None]
    return beta, K, G


@pytest.mark.parametrize("D,n_utt,n_spk,two_models", [(39, 5, 2, False), (40, 3, 3, True), (13, 2, 1, False)])
def test_fmllr_stats_vs_numpy(ctx, D, n_utt, n_spk, two_models):
    import torch
    from voicebridge_amd import api, synth
    rng = np.random.default_rng(D + n_utt)
    P = 30
    offs, w, means, iv = synth.gmm_model(P, 150 if D != 13 else 2400, D, seed=D)      # D = 13: pdfs with more than 64 Gaussians too
    miv = (means * iv).astype(np.float32)
    gmm = api.DiagGmm(ctx, offs, w, miv, iv)
    ali, ali_arrays = None, None
    if two_models:       # an "alignment model" with the same Gaussians per pdf and other parameters
        means2 = means + 0.1 * rng.standard_normal(means.shape).astype(np.float32)
        miv2 = (means2 * iv).astype(np.float32)
        ali = api.DiagGmm(ctx, offs, w, miv2, iv)
        ali_arrays = (ali.gconsts().astype(np.float64), miv2, iv)
    lens = rng.integers(70, 333, n_utt)           # ragged: not multiples of the 64-frame stage or the 4-frame MFMA step
    foff = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    T = int(foff[-1])
    feats = (rng.standard_normal((T, D)) * 1.2).astype(np.float32)
    utt2spk = rng.integers(0, n_spk, n_utt).astype(np.int32)
    if n_utt > 2:
        utt2spk[1] = -1                             # a skipped utterance
    npairs = rng.integers(0, 4, T)                 # frames without posteriors too
    poff = np.concatenate([[0], np.cumsum(npairs)]).astype(np.int64)
    ppdf = rng.integers(0, P, int(poff[-1])).astype(np.int32)
    pw = rng.uniform(0.05, 1.0, int(poff[-1])).astype(np.float32)
    beta, K, G, ms = gmm.fmllr_stats(torch.from_numpy(feats).to(ctx.device), foff, utt2spk, n_spk, poff, ppdf, pw, ali_gmm=ali)
    rb, rK, rG = numpy_stats(offs, gmm.gconsts().astype(np.float64), miv, iv, feats, foff, utt2spk, n_spk, poff, ppdf, pw, ali=ali_arrays)
    np.testing.assert_allclose(beta, rb, rtol=1e-5, atol=1e-4)
    np.testing.assert_allclose(K, rK, rtol=1e-5, atol=1e-5 * np.abs(rK).max())
    np.testing.assert_allclose(G, rG, rtol=1e-5, atol=1e-5 * np.abs(rG).max())
    assert np.array_equal(G, np.swapaxes(G, 2, 3))          # full symmetric matrices come back
    gmm.close()
    if ali:
        ali.close()


@pytest.mark.parametrize("D,n_spk", [(39, 5), (40, 3), (13, 2)])
def test_fmllr_update_on_the_device_vs_host(ctx, D, n_spk):
    """vbh_fmllr_update_full (one wave per speaker, from the statistics where vbh_fmllr_accumulate left them) against the host's
    FmllrAccs::Update on the same statistics (host/fmllr.cc, itself pinned to the reference's gmm-est-fmllr by the golden tests below):
    the same algorithm in double, so the transforms agree far inside the 5e-6 the reference comparison allows; a speaker below min-count
    keeps its initial transform; a non-unit initial transform is taken up"""
    import ctypes as C
    import torch
    from voicebridge_amd import api, synth
    from voicebridge_amd import build
    rng = np.random.default_rng(100 + D)
    P = 40
    offs, w, means, iv = synth.gmm_model(P, 200, D, seed=D + 1)
    miv = (means * iv).astype(np.float32)
    gmm = api.DiagGmm(ctx, offs, w, miv, iv)
    lens = rng.integers(400, 900, 2 * n_spk)
    lens[-1] = 30                                    # the last speaker's second utterance: tiny
    foff = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    T = int(foff[-1])
    feats = (rng.standard_normal((T, D)) * 1.1 + 0.3).astype(np.float32)
    utt2spk = (np.arange(2 * n_spk) % n_spk).astype(np.int32)
    utt2spk[n_spk - 1] = -1                          # ... and its first one skipped: that speaker stays below min-count
    poff = np.arange(T + 1, dtype=np.int64)
    ppdf = rng.integers(0, P, T).astype(np.int32)
    pw = np.ones(T, np.float32)
    beta, K, G, _ = gmm.fmllr_stats(torch.from_numpy(feats).to(ctx.device), foff, utt2spk, n_spk, poff, ppdf, pw)
    assert beta[n_spk - 1] < 100 < beta[0]
    host = C.CDLL(build.HOST_LIB)
    host.vb_fmllr_update.restype = C.c_int
    host.vb_fmllr_update.argtypes = [C.c_int, C.c_double, C.c_void_p, C.c_void_p, C.c_char_p, C.c_float, C.c_int, C.c_void_p, C.POINTER(C.c_float)]
    for init in ("unit", "other"):
        x0 = np.tile(np.hstack([np.eye(D, dtype=np.float32), np.zeros((D, 1), np.float32)]), (n_spk, 1, 1))
        if init == "other":
            x0 = (x0 + 0.02 * rng.standard_normal(x0.shape)).astype(np.float32)
        xd, impr_d, ms = gmm.fmllr_update_full(n_spk, xform=x0, min_count=100.0, num_iters=40)
        for s in range(n_spk):
            xh = x0[s].copy()
            impr_h = C.c_float()
            assert host.vb_fmllr_update(D, float(beta[s]), np.ascontiguousarray(K[s]).ctypes.data, np.ascontiguousarray(G[s]).ctypes.data, b"full", 100.0, 40,
                                        xh.ctypes.data, C.byref(impr_h)) == 0
            np.testing.assert_allclose(xd[s], xh, rtol=0, atol=2e-6 * max(1.0, np.abs(xh).max()))
            assert abs(impr_d[s] - impr_h.value) <= 1e-4 * max(1.0, abs(impr_h.value))
            if s == n_spk - 1:
                assert np.array_equal(xd[s], x0[s]) and impr_d[s] == 0.0      # below min-count: untouched
            else:
                assert impr_d[s] > 0.0 and not np.array_equal(xd[s], x0[s])
        print("D = %d, %d speakers, initial transform %s: device update %.2f ms" % (D, n_spk, init, ms))
    gmm.close()


def test_fmllr_update_on_the_device_refuses_singular_statistics(ctx):
    """constant features make every G_d singular: the host's update throws "fMLLR: singular matrix", the device's fails the call with the
    same words (and names the speaker); statistics of another shape than asked for are refused before anything runs"""
    import torch
    from voicebridge_amd import api, synth
    D, P = 13, 8
    offs, w, means, iv = synth.gmm_model(P, 16, D, seed=3)
    gmm = api.DiagGmm(ctx, offs, w, (means * iv).astype(np.float32), iv)
    T = 900
    feats = np.zeros((T, D), np.float32)
    foff = np.array([0, T], np.int64)
    gmm.fmllr_stats(torch.from_numpy(feats).to(ctx.device), foff, np.zeros(1, np.int32), 1, np.arange(T + 1, dtype=np.int64), np.zeros(T, np.int32), np.ones(T, np.float32))
    with pytest.raises(RuntimeError, match="singular matrix"):
        gmm.fmllr_update_full(1, min_count=100.0)
    with pytest.raises(RuntimeError, match="holds no statistics of 2 speakers"):
        gmm.fmllr_update_full(2, min_count=100.0)
    gmm.close()


def test_gmm_est_fmllr_bin_function_vs_golden(tmp_path):
    """GmmEstFmllr and the gmm-post-to-gpost | gmm-est-fmllr-gpost pair, both fed by the kernel: the transforms of the
    reference's programs on the golden case"""
    from tests.test_host_io import read_fm_ark
    from tests.test_host_lattice import call
    from voicebridge_amd import build
    host = C.CDLL(build.HOST_LIB)
    d = tmp_path
    g = util.load("fmllr")
    for k in ("file_final_mdl", "file_feats_ark", "file_wsp_ark", "file_trans_tmp_ark", "file_pre_trans_ark", "file_gpost_ark"):
        name = k[5:].rsplit("_", 1)
        (d / (name[0] + "." + name[1])).write_bytes(g[k].tobytes())
    (d / "spk2utt").write_text(str(g["text_spk2utt"]))
    log = str(d / "log.txt")
    args = ["--fmllr-min-count=50", "--spk2utt=ark:%s" % (d / "spk2utt"), str(d / "final.mdl"), "ark:%s" % (d / "feats.ark"), "ark:%s" % (d / "wsp.ark")]
    assert call(host.vb_GmmEstFmllr, ["gmm-est-fmllr"] + args + ["ark:%s" % (d / "m.ark")], log) == 0, open(log).read()
    mine, want = read_fm_ark(str(d / "m.ark")), read_fm_ark(str(d / "trans_tmp.ark"))
    assert list(mine) == list(want)
    for k in want:
        np.testing.assert_allclose(mine[k], want[k], atol=2e-4, rtol=0)
    # the Gaussian-posterior form (first pass of DecodeFmllr): the reference's gpost archive in, its pre_trans out
    assert call(host.vb_GmmEstFmllrGpost, ["gmm-est-fmllr-gpost"] + args[:4] + ["ark:%s" % (d / "gpost.ark"), "ark:%s" % (d / "g.ark")], log) == 0, open(log).read()
    mine, want = read_fm_ark(str(d / "g.ark")), read_fm_ark(str(d / "pre_trans.ark"))
    assert list(mine) == list(want)
    for k in want:
        np.testing.assert_allclose(mine[k], want[k], atol=2e-4, rtol=0)
    # per utterance (no --spk2utt)
    assert call(host.vb_GmmEstFmllr, ["gmm-est-fmllr"] + args[:1] + args[2:] + ["ark:%s" % (d / "u.ark")], log) == 0, open(log).read()
    assert list(read_fm_ark(str(d / "u.ark"))) == ["uttA", "uttB", "uttC", "uttD"]


def test_gmm_est_fmllr_ali_model_vs_gpost_pair(tmp_path):
    """--ali-model (extension; what DecodeFmllr's first pass uses): Gaussian posteriors from an alignment model, statistics
    from the adapt model, without a gpost table -- the same transforms as gmm-post-to-gpost <alimdl> | gmm-est-fmllr-gpost
    <adapt model> (each of which test_gmm_est_fmllr_bin_function_vs_golden / test_host_fmllr pin to the reference)"""
    from tests.test_host_io import read_fm_ark
    from tests.test_host_lattice import call
    from voicebridge_amd import build, synth
    host = C.CDLL(build.HOST_LIB)
    d = tmp_path
    g = util.load("fmllr")
    for k in ("file_final_mdl", "file_feats_ark", "file_wsp_ark"):
        name = k[5:].rsplit("_", 1)
        (d / (name[0] + "." + name[1])).write_bytes(g[k].tobytes())
    (d / "spk2utt").write_text(str(g["text_spk2utt"]))
    # the golden case's model (oracle/gen_golden_lat.py make_case, seed 5) with other means: same topology and Gaussian counts
    n_cd, H = 12, 3
    P = n_cd * H
    offs, w, means, iv = synth.gmm_model(P, P * 3, 39, seed=6)
    rng = np.random.default_rng(0)
    means2 = means + 0.3 * rng.standard_normal(means.shape)
    util.write_mdl(str(d / "ali.mdl"), n_cd, H, offs, w, (means2 * iv).astype(np.float32), iv)
    log = str(d / "log.txt")
    common = ["--fmllr-min-count=50", "--spk2utt=ark:%s" % (d / "spk2utt")]
    tail = ["ark:%s" % (d / "feats.ark"), "ark:%s" % (d / "wsp.ark")]
    assert call(host.vb_GmmEstFmllr, ["gmm-est-fmllr"] + common + ["--ali-model=%s" % (d / "ali.mdl"), str(d / "final.mdl")] + tail + ["ark:%s" % (d / "fused.ark")], log) == 0, open(log).read()
    assert call(host.vb_GmmPostToGpost, ["gmm-post-to-gpost", str(d / "ali.mdl")] + tail + ["ark:%s" % (d / "gp.ark")], log) == 0, open(log).read()
    assert call(host.vb_GmmEstFmllrGpost, ["gmm-est-fmllr-gpost"] + common + [str(d / "final.mdl"), tail[0], "ark:%s" % (d / "gp.ark"), "ark:%s" % (d / "pair.ark")], log) == 0, open(log).read()
    fused, pair = read_fm_ark(str(d / "fused.ark")), read_fm_ark(str(d / "pair.ark"))
    assert list(fused) == list(pair) == ["spk1", "spk2"]
    for k in pair:
        np.testing.assert_allclose(fused[k], pair[k], atol=2e-4, rtol=0)
    # and the alignment model matters: with the adapt model's own posteriors the transforms are different ones
    assert call(host.vb_GmmEstFmllr, ["gmm-est-fmllr"] + common + [str(d / "final.mdl")] + tail + ["ark:%s" % (d / "own.ark")], log) == 0, open(log).read()
    own = read_fm_ark(str(d / "own.ark"))
    assert max(np.abs(own[k] - fused[k]).max() for k in own) > 1e-2
    # a model with other Gaussian counts is refused
    offs3, w3, means3, iv3 = synth.gmm_model(P, P * 2, 39, seed=6)
    util.write_mdl(str(d / "bad.mdl"), n_cd, H, offs3, w3, (means3 * iv3).astype(np.float32), iv3)
    assert call(host.vb_GmmEstFmllr, ["gmm-est-fmllr"] + common + ["--ali-model=%s" % (d / "bad.mdl"), str(d / "final.mdl")] + tail + ["ark:%s" % (d / "x.ark")], log) != 0
    assert "--ali-model" in open(log).read()


@pytest.mark.parametrize("update_type", ["diag", "offset", "none"])
def test_gmm_est_fmllr_update_types_vs_golden(tmp_path, update_type):
    """--fmllr-update-type=diag | offset | none (FmllrDiagGmmAccs::Update, K/transform/fmllr-diag-gmm.cc:131-167,275-379) against
    the transforms of the reference's gmm-est-fmllr on the golden case (tests/golden/fmllr_types.npz, oracle/gen_golden_lat.py
    main_fmllr_types)"""
    from tests.test_host_io import read_fm_ark
    from tests.test_host_lattice import call
    from voicebridge_amd import build
    host = C.CDLL(build.HOST_LIB)
    d = tmp_path
    g, gt = util.load("fmllr"), util.load("fmllr_types")
    for k in ("file_final_mdl", "file_feats_ark", "file_wsp_ark"):
        name = k[5:].rsplit("_", 1)
        (d / (name[0] + "." + name[1])).write_bytes(g[k].tobytes())
    (d / "want.ark").write_bytes(gt["file_trans_%s_ark" % update_type].tobytes())
    (d / "spk2utt").write_text(str(g["text_spk2utt"]))
    log = str(d / "log.txt")
    assert call(host.vb_GmmEstFmllr, ["gmm-est-fmllr", "--fmllr-min-count=50", "--fmllr-update-type=" + update_type, "--spk2utt=ark:%s" % (d / "spk2utt"),
                                      str(d / "final.mdl"), "ark:%s" % (d / "feats.ark"), "ark:%s" % (d / "wsp.ark"), "ark:%s" % (d / "m.ark")], log) == 0, open(log).read()
    mine, want = read_fm_ark(str(d / "m.ark")), read_fm_ark(str(d / "want.ark"))
    assert list(mine) == list(want) == ["spk1", "spk2"]
    for k in want:
        np.testing.assert_allclose(mine[k], want[k], atol=2e-4, rtol=0)
        if update_type != "none":
            assert np.abs(want[k] - np.eye(39, 40)).max() > 0.05
    assert call(host.vb_GmmEstFmllr, ["gmm-est-fmllr", "--fmllr-update-type=banana", str(d / "final.mdl"), "ark:%s" % (d / "feats.ark"), "ark:%s" % (d / "wsp.ark"),
                                      "ark:%s" % (d / "x.ark")], log) != 0
    assert "fmllr-update-type must be one of" in open(log).read()


def test_fmllr_update_properties(tmp_path):
    """properties of the estimate that do not depend on a reference run: below --fmllr-min-count the
    unit transform is written; the update raises the auxiliary function (logged improvement > 0)"""
    from tests.test_host_io import read_fm_ark
    from tests.test_host_lattice import call
    from voicebridge_amd import build
    host = C.CDLL(build.HOST_LIB)
    d = tmp_path
    g = util.load("fmllr")
    for k in ("file_final_mdl", "file_feats_ark", "file_wsp_ark"):
        name = k[5:].rsplit("_", 1)
        (d / (name[0] + "." + name[1])).write_bytes(g[k].tobytes())
    (d / "spk2utt").write_text(str(g["text_spk2utt"]))
    log = str(d / "log.txt")
    base = ["gmm-est-fmllr", "--spk2utt=ark:%s" % (d / "spk2utt"), str(d / "final.mdl"), "ark:%s" % (d / "feats.ark"), "ark:%s" % (d / "wsp.ark")]
    assert call(host.vb_GmmEstFmllr, base + ["ark:%s" % (d / "unit.ark")], log) == 0      # default min-count 500 > the ~270 frames per speaker
    for m in read_fm_ark(str(d / "unit.ark")).values():
        np.testing.assert_array_equal(m, np.eye(39, 40, dtype=np.float32))
    assert "below min-count" in open(log).read()
    os.remove(log)
    assert call(host.vb_GmmEstFmllr, base[:1] + ["--fmllr-min-count=50"] + base[1:] + ["ark:%s" % (d / "t.ark")], log) == 0
    imprs = [float(l.split("auxf-impr from fMLLR is")[1].split(",")[0]) for l in open(log) if "auxf-impr" in l]
    assert len(imprs) == 2 and all(x > 0 for x in imprs)
    assert call(host.vb_GmmEstFmllr, base[:1] + ["--fmllr-update-type=banana"] + base[1:] + ["ark:%s" % (d / "x.ark")], log) == -1    # refused, not ignored


def test_rescore_and_lattice_to_compact_conversion_vs_golden(tmp_path):
    """a state-level lattice read where a CompactLattice is expected is folded by Factor + TopSort
    (K/fstext/lattice-utils-inl.h:27-100): gmm-rescore-lattice with --old-acoustic-scale=1 on features
    of the right length adds scores but keeps the structure the reference's conversion produced"""
    from tests.test_host_lattice import assert_same_archives, call
    from voicebridge_amd import build
    host = C.CDLL(build.HOST_LIB)
    d = tmp_path
    g = util.load("fmllr")
    for k in ("file_final_mdl", "file_feats_ark", "file_raw_ark", "file_grl_ark"):
        name = k[5:].rsplit("_", 1)
        (d / (name[0] + "." + name[1])).write_bytes(g[k].tobytes())
    log = str(d / "log.txt")
    assert call(host.vb_GmmRescoreLattice, ["gmm-rescore-lattice", str(d / "final.mdl"), "ark:%s" % (d / "raw.ark"), "ark:%s" % (d / "feats.ark"),
                                            "ark:%s" % (d / "m.ark")], log) == 0, open(log).read()
    # scores from the on-demand GPU kernel: costs within float summation order of the reference's
    assert_same_archives(str(d / "m.ark"), str(d / "grl.ark"), rtol=1e-6, atol=1e-4)
    mine, want = util.read_lattice_ark(str(d / "m.ark")), util.read_lattice_ark(str(d / "grl.ark"))
    for k in want:
        assert [(len(a), [(x[0], x[3], x[4]) for x in a]) for _f, a in mine[k][2]] == [(len(a), [(x[0], x[3], x[4]) for x in a]) for _f, a in want[k][2]]
    # features shorter than the lattice: reported, utterance skipped, return code 1 when nothing is left
    (d / "short.ark").write_bytes(b"")
    util.write_feats_ark(str(d / "short.ark"), {"uttA": np.zeros((5, 39), np.float32)})
    assert call(host.vb_GmmRescoreLattice, ["gmm-rescore-lattice", str(d / "final.mdl"), "ark:%s" % (d / "raw.ark"), "ark:%s" % (d / "short.ark"),
                                            "ark:%s" % (d / "m2.ark")], log) == 1


@pytest.mark.parametrize("P,G,D,T", [(40, 260, 39, 150), (5, 700, 13, 70)])
def test_gmm_loglikes_pairs_vs_all_pdf_kernel(ctx, P, G, D, T):
    """the on-demand form (vbh_gmm_loglikes_pairs) against the batched all-pdf kernel on the same frames,
    incl. pdfs with more than 64 Gaussians, frames without pairs and repeated pdfs"""
    import torch
    from voicebridge_amd import api, synth
    offs, w, means, iv = synth.gmm_model(P, G, D, seed=G)
    miv = (means * iv).astype(np.float32)
    gmm = api.DiagGmm(ctx, offs, w, miv, iv)
    rng = np.random.default_rng(T)
    feats = torch.from_numpy((rng.standard_normal((T, D)) * 1.3).astype(np.float32)).to(ctx.device)
    full = gmm.loglikes(feats).cpu().numpy()
    n = rng.integers(0, 5, T)
    poff = np.concatenate([[0], np.cumsum(n)]).astype(np.int64)
    ppdf = rng.integers(0, P, int(poff[-1])).astype(np.int32)
    out = np.zeros(len(ppdf), np.float32)
    fn = gmm.lib.vbh_gmm_loglikes_pairs
    fn.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]
    assert fn(gmm.h, feats.data_ptr(), T, poff.ctypes.data, ppdf.ctypes.data, out.ctypes.data) == 0
    want = np.array([full[t, ppdf[q]] for t in range(T) for q in range(poff[t], poff[t + 1])], np.float32)
    np.testing.assert_allclose(out, want, rtol=1e-4, atol=1e-3)
    gmm.close()


def test_fused_front_end_equals_feature_archives(tmp_path):
    """--fe-* on gmm-est-fmllr and gmm-rescore-lattice (device_feats.h; what DecodeFmllr uses instead of the reference's
    sifeats / pass-1 / final feature archives): the same transforms and the same rescored lattices as the programs fed with
    features that apply-cmvn | add-deltas | transform-feats wrote to an archive first"""
    from tests.test_host_io import read_fm_ark
    from tests.test_host_lattice import assert_same_archives, call
    from voicebridge_amd import build, synth
    host = C.CDLL(build.HOST_LIB)
    d = tmp_path
    g = util.load("fmllr")
    for k in ("file_final_mdl", "file_feats_ark", "file_wsp_ark", "file_raw_ark"):
        name = k[5:].rsplit("_", 1)
        (d / (name[0] + "." + name[1])).write_bytes(g[k].tobytes())
    (d / "spk2utt").write_text(str(g["text_spk2utt"]))
    spk2utt = {l.split()[0]: l.split()[1:] for l in str(g["text_spk2utt"]).split("\n") if l.strip()}
    (d / "utt2spk").write_text("".join("%s %s\n" % (u, s) for s, us in spk2utt.items() for u in us))
    # raw 13-dim features with the golden utterances' keys and lengths; per-speaker CMVN statistics and transforms
    rng = np.random.default_rng(3)
    golden = read_fm_ark(str(d / "feats.ark"))
    raw = {k: (rng.standard_normal((v.shape[0], 13)) * np.linspace(5, 1, 13) + np.linspace(-2, 2, 13)).astype(np.float32) for k, v in golden.items()}
    util.write_feats_ark(str(d / "raw13.ark"), raw)
    log = str(d / "log.txt")
    assert call(host.vb_ComputeCmvnStats, ["compute-cmvn-stats", "--spk2utt=ark:%s" % (d / "spk2utt"), "ark:%s" % (d / "raw13.ark"), "ark:%s" % (d / "cmvn.ark")], log) == 0, open(log).read()
    util.write_feats_ark(str(d / "pre.ark"), {s: synth.affine(39, seed=40 + i) for i, s in enumerate(sorted(spk2utt))})
    u2s = "--utt2spk=ark:%s" % (d / "utt2spk")
    assert call(host.vb_ApplyCmvn, ["apply-cmvn", u2s, "ark:%s" % (d / "cmvn.ark"), "ark:%s" % (d / "raw13.ark"), "ark:%s" % (d / "c.ark")], log) == 0
    assert call(host.vb_AddDeltas, ["add-deltas", "ark:%s" % (d / "c.ark"), "ark:%s" % (d / "dl.ark")]) == 0
    assert call(host.vb_TransformFeats, ["transform-feats", u2s, "ark:%s" % (d / "pre.ark"), "ark:%s" % (d / "dl.ark"), "ark:%s" % (d / "proc.ark")], log) == 0
    fe = ["--fe-cmvn=ark:%s" % (d / "cmvn.ark"), "--fe-utt2spk=ark:%s" % (d / "utt2spk"), "--fe-type=delta", "--fe-trans=ark:%s" % (d / "pre.ark")]
    common = ["--fmllr-min-count=50", "--spk2utt=ark:%s" % (d / "spk2utt")]
    # ---- gmm-est-fmllr
    assert call(host.vb_GmmEstFmllr, ["gmm-est-fmllr"] + common + [str(d / "final.mdl"), "ark:%s" % (d / "proc.ark"), "ark:%s" % (d / "wsp.ark"), "ark:%s" % (d / "t_arch.ark")], log) == 0, open(log).read()
    assert call(host.vb_GmmEstFmllr, ["gmm-est-fmllr"] + common + fe + [str(d / "final.mdl"), "ark:%s" % (d / "raw13.ark"), "ark:%s" % (d / "wsp.ark"), "ark:%s" % (d / "t_fused.ark")], log) == 0, open(log).read()
    a, b = read_fm_ark(str(d / "t_arch.ark")), read_fm_ark(str(d / "t_fused.ark"))
    assert list(a) == list(b) and len(a) == len(spk2utt)
    for k in a:
        np.testing.assert_allclose(b[k], a[k], atol=2e-5, rtol=0)
        assert np.abs(a[k] - np.eye(39, 40)).max() > 1e-3          # real updates, not the unit transform of a skipped speaker
    # ---- gmm-rescore-lattice
    assert call(host.vb_GmmRescoreLattice, ["gmm-rescore-lattice", str(d / "final.mdl"), "ark:%s" % (d / "raw.ark"), "ark:%s" % (d / "proc.ark"), "ark:%s" % (d / "r_arch.ark")], log) == 0, open(log).read()
    assert call(host.vb_GmmRescoreLattice, ["gmm-rescore-lattice"] + fe + [str(d / "final.mdl"), "ark:%s" % (d / "raw.ark"), "ark:%s" % (d / "raw13.ark"), "ark:%s" % (d / "r_fused.ark")], log) == 0, open(log).read()
    assert_same_archives(str(d / "r_fused.ark"), str(d / "r_arch.ark"), rtol=1e-6, atol=1e-4)
    # the options only make sense together
    assert call(host.vb_GmmRescoreLattice, ["gmm-rescore-lattice", "--fe-type=delta", str(d / "final.mdl"), "ark:%s" % (d / "raw.ark"), "ark:%s" % (d / "raw13.ark"), "ark:%s" % (d / "x.ark")], log) == -1
    assert "--fe-cmvn" in open(log).read()
