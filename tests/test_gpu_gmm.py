"""GPU parity: all-pdf diagonal-GMM log-likelihoods vs the golden vectors (reference) and the oracle."""
import numpy as np
import pytest

from tests.util import load

pytestmark = pytest.mark.gpu

# north_star: GMM log-likelihoods "within stated float tolerance".  Stated: 1e-4 relative
#  with a small absolute floor for values near zero.
RTOL, ATOL = 1e-4, 1e-3


@pytest.fixture(scope="module")
def ctx():
    from voicebridge_amd import api
    c = api.Context(0)
    yield c
    c.close()


@pytest.mark.parametrize("name", ["gmm_p8_d39", "gmm_p64_d40"])
def test_gmm_golden(ctx, name):
    import torch
    from voicebridge_amd import api
    g = load(name)
    m = api.DiagGmm(ctx, g["offs"], g["weights"], g["miv"], g["inv_vars"])
    gc = m.gconsts()
    fin = np.isfinite(g["gconsts"])
    np.testing.assert_array_equal(np.isneginf(gc), np.isneginf(g["gconsts"]))   # zero-weight component
    np.testing.assert_allclose(gc[fin], g["gconsts"][fin], rtol=1e-6, atol=1e-4)
    ll = m.loglikes(torch.from_numpy(g["feats"]).to(ctx.device))
    ctx.sync()
    np.testing.assert_allclose(ll.cpu().numpy(), g["loglikes"], rtol=RTOL, atol=ATOL)
    m.close()


@pytest.mark.parametrize("P,G,D,T", [(50, 300, 39, 200), (7, 7, 13, 65), (300, 1800, 40, 130), (3, 240, 39, 5),
                                     (3, 700, 39, 70), (2, 2100, 20, 130), (40, 6000, 39, 200), (120, 120, 13, 70), (90, 150, 39, 64),
                                     (1, 1, 5, 3), (1, 333, 39, 64), (500, 9000, 39, 100), (4000, 16000, 39, 70), (64, 8192, 39, 66),
                                     (1000, 1000, 20, 65), (700, 2100, 60, 64), (33, 5000, 39, 64)])
def test_gmm_vs_oracle(ctx, P, G, D, T):
    """ragged frame counts (not multiples of the 64-frame tile), 1-Gaussian pdfs, pdfs with
    ~100 Gaussians, tiles split over blockIdx.y, pdfs with more Gaussians than one 128-column tile
    (scored in parts and folded); in the dense packing of the default LogSumExp: pdfs that run from one 32-row block into the next
    (their sum is carried on), more than 16 one-Gaussian pdfs in a row (the 16-ends-per-block rule), one pdf that is the whole model"""
    import torch
    from oracle import vbo
    from voicebridge_amd import api, synth
    offs, w, means, iv = synth.gmm_model(P, G, D, seed=P + G)
    miv = (means * iv).astype(np.float32)
    rng = np.random.default_rng(T)
    feats = (rng.standard_normal((T, D)) * 1.3).astype(np.float32)
    m = api.DiagGmm(ctx, offs, w, miv, iv)
    ll = m.loglikes(torch.from_numpy(feats).to(ctx.device))
    ctx.sync()
    ref = vbo.gmm_loglikes(offs, vbo.gmm_gconsts(w, miv, iv), miv, iv, feats)
    np.testing.assert_allclose(ll.cpu().numpy(), ref, rtol=RTOL, atol=ATOL)
    m.close()


@pytest.mark.parametrize("cut_step", [None, "1", "3"])
@pytest.mark.parametrize("sizes", [[130] * 29 + [1] * 64 + [130] * 30, [100] * 13 + [1] * 48 + [7, 1, 1, 90] * 20, [1] * 37 + [200] * 9 + [1] * 51])
def test_gmm_mixed_model_staging_rows(ctx, sizes, cut_step, monkeypatch):
    """many 100+-Gaussian pdfs around runs of one-Gaussian pdfs.  With superblocks cut at arbitrary pdfs (VBH_GMM_CUT_STEP=1 / 3)
    a run of finished pdfs starts at a pdf that is no multiple of 4, the aligned flush can leave 16-18 waiting, and the next block's ends
    used to run over the lane's staging row (LDP): the flush now empties the row then, and the default packing cuts at multiples of 4."""
    import torch
    from oracle import vbo
    from voicebridge_amd import api
    if cut_step is not None:
        monkeypatch.setenv("VBH_GMM_CUT_STEP", cut_step)
    D, T = 39, 130
    rng = np.random.default_rng(len(sizes))
    offs = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int32)
    G = int(offs[-1])
    w = np.concatenate([rng.dirichlet(np.ones(n)) for n in sizes]).astype(np.float32)
    means = rng.standard_normal((G, D)).astype(np.float32)
    iv = (1.0 / rng.uniform(0.5, 2.0, (G, D))).astype(np.float32)
    miv = (means * iv).astype(np.float32)
    feats = (rng.standard_normal((T, D)) * 1.3).astype(np.float32)
    ref = vbo.gmm_loglikes(offs, vbo.gmm_gconsts(w, miv, iv), miv, iv, feats)
    for padded in (False, True):
        m = api.DiagGmm(ctx, offs, w, miv, iv)
        d = torch.from_numpy(feats).to(ctx.device)
        if padded:
            out = torch.zeros((T, m.padded_pdfs), dtype=torch.float32, device=ctx.device)
            m.loglikes(d, out=out)
            ll = out[:, :len(sizes)]
        else:
            ll = m.loglikes(d)
        ctx.sync()
        np.testing.assert_allclose(ll.cpu().numpy(), ref, rtol=RTOL, atol=ATOL)
        m.close()


@pytest.mark.parametrize("P,G,D,T", [(50, 300, 39, 200), (3, 700, 39, 70), (120, 120, 13, 70)])
def test_gmm_strict_and_twopass_vs_oracle(ctx, P, G, D, T):
    """the block-local packing (pdfs never straddle a row-block, large pdfs in parts) behind the two-pass modes"""
    import torch
    from oracle import vbo
    from voicebridge_amd import api, synth
    offs, w, means, iv = synth.gmm_model(P, G, D, seed=P + G)
    miv = (means * iv).astype(np.float32)
    feats = (np.random.default_rng(T).standard_normal((T, D)) * 1.3).astype(np.float32)
    ref = vbo.gmm_loglikes(offs, vbo.gmm_gconsts(w, miv, iv), miv, iv, feats)
    for mode in ("strict", "twopass"):
        m = api.DiagGmm(ctx, offs, w, miv, iv, lse=mode)
        ll = m.loglikes(torch.from_numpy(feats).to(ctx.device))
        ctx.sync()
        np.testing.assert_allclose(ll.cpu().numpy(), ref, rtol=RTOL, atol=ATOL)
        m.close()


def test_gmm_padded_rows_and_the_search_on_them(ctx):
    """vbh_gmm_loglikes_ld: rows padded_pdfs floats apart hold the same scores bit for bit, the pad columns are left alone, and the
    lattice search gives the same result on the padded matrix (its num_pdfs argument is the row length)"""
    import torch
    from voicebridge_amd import api, synth
    P, G, D, T = 203, 1500, 39, 333
    offs, w, means, iv = synth.gmm_model(P, G, D, seed=5)
    miv = (means * iv).astype(np.float32)
    feats = torch.from_numpy((np.random.default_rng(2).standard_normal((T, D)) * 1.2).astype(np.float32)).to(ctx.device)
    for mode in ("fast", "strict"):
        m = api.DiagGmm(ctx, offs, w, miv, iv, lse=mode)
        assert m.padded_pdfs == 224
        dense = m.loglikes(feats)
        pad = torch.full((T, m.padded_pdfs), 7.0, dtype=torch.float32, device=ctx.device)
        m.loglikes(feats, out=pad)
        ctx.sync()
        assert torch.equal(pad[:, :P], dense) and bool((pad[:, P:] == 7.0).all())
        m.close()
    # the lattice search on the padded matrix: same words, alignment and counters
    ntid = 3 * P
    g = synth.decoding_graph(3000, ntid, seed=9, eps_frac=0.3)
    t2p = synth.tid2pdf(ntid, P, seed=9)
    graph = api.Graph(ctx, g["final"], g["arc_offset"], g["arcs"], 0)
    foff = np.array([0, 100, 210, 333], np.int64)
    dec = api.LatticeDecoder(ctx, graph, 3, 123, beam=11.0, max_active=3000, min_active=100, lattice_beam=5.0, acoustic_scale=0.1)
    out = []
    for mat in (dense, pad):
        dec.decode(mat, foff, t2p)
        out.append([dec.result(u) for u in range(3)])
    for ra, rb in zip(*out):
        np.testing.assert_array_equal(ra["words"], rb["words"])
        np.testing.assert_array_equal(ra["ali"], rb["ali"])
        assert ra["best_graph"] == rb["best_graph"] and ra["best_acoustic"] == rb["best_acoustic"]
    dec.close(); graph.close()


def test_gmm_reference_sizes_property(ctx):
    """BASELINE sizes (P=2500, G=15000, D=39): scoring a frame batch must equal scoring its
    rows one tile at a time in any order (each output depends on its own frame only), and a
    sample of rows must match the oracle."""
    import torch
    from oracle import vbo
    from voicebridge_amd import api, synth
    offs, w, means, iv = synth.gmm_model(2500, 15000, 39, seed=1)
    miv = (means * iv).astype(np.float32)
    rng = np.random.default_rng(3)
    feats = (rng.standard_normal((2998, 39)) * 1.2).astype(np.float32)
    m = api.DiagGmm(ctx, offs, w, miv, iv)
    d = torch.from_numpy(feats).to(ctx.device)
    full = m.loglikes(d)
    perm = torch.from_numpy(rng.permutation(2998)).to(ctx.device)
    shuf = m.loglikes(d[perm].contiguous())
    ctx.sync()
    assert torch.equal(full[perm], shuf)                       # bit-exact: row independence
    rows = rng.choice(2998, 6, replace=False)
    ref = vbo.gmm_loglikes(offs, vbo.gmm_gconsts(w, miv, iv), miv, iv, feats[rows])
    np.testing.assert_allclose(full[torch.from_numpy(rows).to(ctx.device)].cpu().numpy(), ref, rtol=RTOL, atol=ATOL)
    m.close()
