"""GPU parity: FasterDecoder vs the oracle in the same admission mode -- ordered (the default: the reference's rule in a canonical order,
oracle exact_cutoff = 3) and exact (the order-independent subset, exact_cutoff = 1) -- bit-exact best path (words, alignment, float32
weights), and vs the reference's golden best paths."""
import glob
import os

import numpy as np
import pytest

from tests.util import GOLD, load, parse_opts
from tests.test_gpu_decode import arcs_of, make_case

pytestmark = pytest.mark.gpu

CASES = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLD, "decode_*.npz")))
ORACLE_MODE = {"ordered": 3, "exact": 1}
ADMS = ("ordered", "exact")


@pytest.fixture(scope="module")
def ctx():
    from voicebridge_amd import api
    c = api.Context(0)
    yield c
    c.close()


def assert_same_path(got, ref):
    for k in ("ok", "reached_final", "num_frames"):
        assert got[k] == ref[k], k
    np.testing.assert_array_equal(got["words"], ref["words"])
    np.testing.assert_array_equal(got["ali"], ref["ali"])
    assert np.float32(got["best_graph"]) == np.float32(ref["best_graph"])
    assert np.float32(got["best_acoustic"]) == np.float32(ref["best_acoustic"])


@pytest.mark.parametrize("adm", ADMS)
@pytest.mark.parametrize("name", CASES)
def test_faster_golden_cases(ctx, name, adm):
    import torch
    from oracle import vbo
    from voicebridge_amd import api
    g = load(name)
    o = parse_opts(g["opts"])
    arcs = arcs_of(g)
    kw = dict(beam=o["beam"], max_active=int(o["max_active"]), min_active=20)
    ref = vbo.decode_faster(vbo.make_fst(g["fst_final"], g["fst_off"], arcs, 0), g["loglikes"], g["tid2pdf"],
                            acwt=o["acwt"], exact_cutoff=ORACLE_MODE[adm], **kw)
    graph = api.Graph(ctx, g["fst_final"], g["fst_off"], arcs, 0)
    T = g["loglikes"].shape[0]
    dec = api.FasterDecoder(ctx, graph, 1, T, acoustic_scale=o["acwt"], admission=adm, **kw)
    ll = torch.from_numpy(g["loglikes"]).to(ctx.device)
    ctx.torch_sync()
    dec.decode(ll, [0, T], g["tid2pdf"])
    got = dec.result(0)
    assert_same_path(got, ref)
    dec.close(); graph.close()
    # and the reference's own best path (golden vectors from the reference build): every case in both modes
    np.testing.assert_array_equal(got["words"], g["ref_fast_words"])
    np.testing.assert_array_equal(got["ali"], g["ref_fast_ali"])
    np.testing.assert_allclose([got["best_graph"], got["best_acoustic"]], g["ref_fast_best_weight"], rtol=2e-6)


@pytest.mark.parametrize("S,P,T,seed,cfg", [
    (20000, 64, 120, 1, dict(beam=16.0, max_active=2147483647, min_active=20)),
    (20000, 64, 90, 2, dict(beam=13.0, max_active=300, min_active=20)),            # max-active binds
    (50000, 200, 77, 3, dict(beam=10.0, max_active=2000, min_active=200)),
    (3000, 32, 60, 4, dict(beam=8.0, max_active=2147483647, min_active=0)),
    (5000, 32, 0, 5, dict(beam=16.0, max_active=7000, min_active=20)),              # empty utterance
    (5000, 32, 1, 6, dict(beam=16.0, max_active=7000, min_active=20)),
])
@pytest.mark.parametrize("adm", ADMS)
def test_faster_batch_vs_oracle(ctx, S, P, T, seed, cfg, adm):
    import torch
    from oracle import vbo
    from voicebridge_amd import api
    g, t2p, ll = make_case(S, P, max(T, 1), seed, eps_frac=0.3, spread=8.0 if cfg["max_active"] < 1000 else 30.0)
    lens = [T, T // 2, max(T - 3, 0)]
    foff = np.zeros(4, np.int64)
    foff[1:] = np.cumsum(lens)
    batch = np.concatenate([ll[:n] for n in lens]) if sum(lens) else np.zeros((0, P), np.float32)
    graph = api.Graph(ctx, g["final"], g["arc_offset"], g["arcs"], 0)
    dec = api.FasterDecoder(ctx, graph, 3, max(lens), admission=adm, **cfg)
    d = torch.from_numpy(np.ascontiguousarray(batch)).to(ctx.device) if len(batch) else torch.zeros((1, P), device=ctx.device)
    ctx.torch_sync()
    dec.decode(d, foff, t2p)
    fst = vbo.make_fst(g["final"], g["arc_offset"], g["arcs"], 0)
    for u, n in enumerate(lens):
        ref = vbo.decode_faster(fst, ll[:n] if n else np.zeros((0, P), np.float32), t2p, exact_cutoff=ORACLE_MODE[adm], **cfg)
        assert_same_path(dec.result(u), ref)
    dec.close(); graph.close()


@pytest.mark.parametrize("S,P,T,seed,spread,cfg", [
    # every state of a 60000-state graph alive in a frame: far more than the 8192 slots of the LDS map -- the second map level, its
    # list of slots, the numbering and the winners of second-level states
    (60000, 64, 40, 7, 30.0, dict(beam=60.0, max_active=2147483647, min_active=0)),
    # all scores equal: masses of equal costs -- the order statistics' by-value histogram has one full step and falls back to the
    # radix select, with max-active binding in every frame
    (20000, 16, 50, 8, 0.0, dict(beam=13.0, max_active=300, min_active=20)),
    # min-active binds in most frames (a narrow beam under widely spread scores)
    (30000, 64, 80, 9, 40.0, dict(beam=2.0, max_active=5000, min_active=150)),
])
@pytest.mark.parametrize("adm", ADMS)
def test_faster_map_spill_ties_and_min_active(ctx, S, P, T, seed, spread, cfg, adm):
    import torch
    from oracle import vbo
    from voicebridge_amd import api
    g, t2p, ll = make_case(S, P, T, seed, eps_frac=0.3, spread=spread)
    graph = api.Graph(ctx, g["final"], g["arc_offset"], g["arcs"], 0)
    dec = api.FasterDecoder(ctx, graph, 2, T, admission=adm, **cfg)
    foff = np.array([0, T, T + T // 2], np.int64)
    d = torch.from_numpy(np.ascontiguousarray(np.concatenate([ll, ll[:T // 2]]))).to(ctx.device)
    ctx.torch_sync()
    dec.decode(d, foff, t2p)
    fst = vbo.make_fst(g["final"], g["arc_offset"], g["arcs"], 0)
    for u, n in enumerate((T, T // 2)):
        ref = vbo.decode_faster(fst, ll[:n], t2p, exact_cutoff=ORACLE_MODE[adm], **cfg)
        assert_same_path(dec.result(u), ref)
    dec.close(); graph.close()


def test_faster_garbage_collection(ctx, monkeypatch):
    """a history far too small for the utterance: unreachable tokens are collected on the way and the
    best path must not change"""
    import torch
    from oracle import vbo
    from voicebridge_amd import api
    cfg = dict(beam=9.0, max_active=400, min_active=20)
    S, P, T, seed = 60000, 96, 160, 31
    g, t2p, ll = make_case(S, P, T, seed, eps_frac=0.3, spread=25.0)
    fst = vbo.make_fst(g["final"], g["arc_offset"], g["arcs"], 0)
    ref = vbo.decode_faster(fst, ll, t2p, exact_cutoff=3, **cfg)      # (the default admission: ordered)
    graph = api.Graph(ctx, g["final"], g["arc_offset"], g["arcs"], 0)
    d = torch.from_numpy(ll).to(ctx.device)
    ctx.torch_sync()
    seen = []
    for hist in (0, 6000, 5400):
        if hist:
            monkeypatch.setenv("VBH_FASTER_FRAME_CAP", "2000")
            monkeypatch.setenv("VBH_FASTER_HISTORY", str(hist))
        dec = api.FasterDecoder(ctx, graph, 1, T, **cfg)
        dec.decode(d, [0, T], t2p)
        got = dec.result(0)
        assert_same_path(got, ref)
        seen.append(got["n_tokens"])
        dec.close()
    assert seen[1] < seen[0] or seen[2] < seen[0]      # the small histories really were collected
    graph.close()
