"""GPU parity: the lattice search vs the oracle in the same admission mode (bit-exact: token
sets, float32 costs, every lattice arc, best path) and vs the reference's golden best paths."""
import glob
import os

import numpy as np
import pytest

from tests.util import ADMISSIONS, GOLD, ORACLE_ADMISSION, load, parse_opts

pytestmark = pytest.mark.gpu

CASES = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLD, "decode_*.npz")))
# (case, admission) pairs whose best path the reference's HashList order itself decides: the search must NOT come out as the reference's there
ORDER_DECIDES = {("decode_g1_maxact", "ordered")}
LAT_KEYS = ["tok_frame", "tok_state", "tok_cost", "tok_extra", "tok_final", "link_src", "link_dst", "link_ilabel",
            "link_olabel", "link_graph", "link_acoustic", "ali", "words"]


@pytest.fixture(scope="module")
def ctx():
    from voicebridge_amd import api
    c = api.Context(0)
    yield c
    c.close()


def arcs_of(g):
    from voicebridge_amd import synth
    ai = g["fst_arcs_i"]
    arcs = np.zeros(len(ai), synth.ARC_DTYPE)
    arcs["ilabel"], arcs["olabel"], arcs["nextstate"] = ai[:, 0], ai[:, 1], ai[:, 2]
    arcs["weight"] = g["fst_arcs_w"]
    return arcs


def assert_same_lattice(got, ref):
    for k in ("ok", "reached_final", "num_frames"):
        assert got[k] == ref[k], k
    for k in LAT_KEYS:
        np.testing.assert_array_equal(got[k], ref[k], err_msg=k)     # float32 arrays compared bit for bit
    assert np.float32(got["best_graph"]) == np.float32(ref["best_graph"])
    assert np.float32(got["best_acoustic"]) == np.float32(ref["best_acoustic"])
    a, b = got["final_relative_cost"], ref["final_relative_cost"]
    assert (np.isinf(a) and np.isinf(b)) or a == b
    assert got["n_frames_maxactive_bound"] == ref["n_frames_maxactive_bound"]
    assert got["n_frames_minactive_bound"] == ref["n_frames_minactive_bound"]


@pytest.mark.parametrize("adm", ADMISSIONS)
@pytest.mark.parametrize("name", CASES)
def test_decode_golden_cases(ctx, name, adm):
    import torch
    from oracle import vbo
    from voicebridge_amd import api
    g = load(name)
    o = parse_opts(g["opts"])
    arcs = arcs_of(g)
    kw = dict(beam=o["beam"], max_active=int(o["max_active"]), min_active=int(o["min_active"]),
              lattice_beam=o["lattice_beam"], prune_interval=int(o["prune_interval"]))
    ref = vbo.decode_lattice_faster(vbo.make_fst(g["fst_final"], g["fst_off"], arcs, 0), g["loglikes"], g["tid2pdf"],
                                    acwt=o["acwt"], **ORACLE_ADMISSION[adm], **kw)
    graph = api.Graph(ctx, g["fst_final"], g["fst_off"], arcs, 0)
    T = g["loglikes"].shape[0]
    dec = api.LatticeDecoder(ctx, graph, 1, T, acoustic_scale=o["acwt"], admission=adm, **kw)
    ll = torch.from_numpy(g["loglikes"]).to(ctx.device)
    ctx.torch_sync()
    dec.decode(ll, [0, T], g["tid2pdf"])
    got = dec.result(0)
    assert_same_lattice(got, ref)
    dec.close(); graph.close()
    # and the reference's own best path (golden vectors from the reference build): a hard assertion for every case and mode but one.  The
    # one random-graph case with max-active binding in every frame is decided by the visiting order itself: there the ordered rule's
    # canonical order (and the superset rule, oracle exact_cutoff = 2) end on another path than the reference's HashList order -- the
    # oracle pins the ordered result bit for bit above, and the expected failure is STRICT: should the case ever come out as the
    # reference's, or any other case stop doing so, the suite says so
    if (name, adm) in ORDER_DECIDES:
        same = (np.array_equal(got["words"], g["ref_lat_words"]) and np.array_equal(got["ali"], g["ref_lat_ali"]) and
                np.array_equal(np.float32([got["best_graph"], got["best_acoustic"]]), g["ref_lat_best_weight"]))
        assert not same, "%s under %s admission now reproduces the reference's path: take it out of ORDER_DECIDES" % (name, adm)
        return
    np.testing.assert_array_equal(got["words"], g["ref_lat_words"])
    np.testing.assert_array_equal(got["ali"], g["ref_lat_ali"])
    np.testing.assert_array_equal(np.float32([got["best_graph"], got["best_acoustic"]]), g["ref_lat_best_weight"])


def make_case(S, P, T, seed, spread=30.0, **gkw):
    from voicebridge_amd import synth
    ntid = 3 * P
    g = synth.decoding_graph(S, ntid, seed=seed, num_words=200, **gkw)
    t2p = synth.tid2pdf(ntid, P, seed=seed)
    rng = np.random.default_rng(seed + 100)
    ll = (rng.standard_normal((T, P)) * spread - 80.0).astype(np.float32)
    return g, t2p, ll


@pytest.mark.parametrize("adm", ADMISSIONS)
@pytest.mark.parametrize("S,P,T,seed,cfg", [
    (20000, 64, 120, 1, dict(beam=13.0, max_active=7000, min_active=200, lattice_beam=6.0)),
    (20000, 64, 90, 2, dict(beam=13.0, max_active=300, min_active=20, lattice_beam=6.0)),       # max-active binds
    (50000, 200, 77, 3, dict(beam=10.0, max_active=2000, min_active=200, lattice_beam=6.0, prune_interval=7)),
    (3000, 32, 60, 4, dict(beam=8.0, max_active=2147483647, min_active=0, lattice_beam=4.0)),
    (5000, 32, 0, 5, dict(beam=13.0, max_active=7000, min_active=200, lattice_beam=6.0)),        # empty utterance
    (5000, 32, 1, 6, dict(beam=13.0, max_active=7000, min_active=200, lattice_beam=6.0)),
])
def test_decode_batch_vs_oracle(ctx, S, P, T, seed, cfg, adm):
    """larger graphs with epsilon chains; a ragged batch of three utterances through one call"""
    import torch
    from oracle import vbo
    from voicebridge_amd import api
    g, t2p, ll = make_case(S, P, max(T, 1), seed, eps_frac=0.3, spread=8.0 if cfg["max_active"] < 1000 else 30.0)
    lens = [T, T // 2, max(T - 3, 0)]
    foff = np.zeros(4, np.int64)
    foff[1:] = np.cumsum(lens)
    batch = np.concatenate([ll[:n] for n in lens]) if sum(lens) else np.zeros((0, P), np.float32)
    graph = api.Graph(ctx, g["final"], g["arc_offset"], g["arcs"], 0)
    dec = api.LatticeDecoder(ctx, graph, 3, max(lens), admission=adm, **cfg)
    if "VBH_DECODE_LAYOUT" not in os.environ:        # (both layouts of the kernel are met by these cases: light configurations get two workgroups per CU)
        assert dec.layout == (2 if cfg["max_active"] <= 4000 else 1)
    d = torch.from_numpy(np.ascontiguousarray(batch)).to(ctx.device) if len(batch) else torch.zeros((1, P), device=ctx.device)
    ctx.torch_sync()
    dec.decode(d, foff, t2p)
    fst = vbo.make_fst(g["final"], g["arc_offset"], g["arcs"], 0)
    for u, n in enumerate(lens):
        ref = vbo.decode_lattice_faster(fst, ll[:n] if n else np.zeros((0, P), np.float32), t2p, **ORACLE_ADMISSION[adm], **cfg)
        assert_same_lattice(dec.result(u), ref)
    dec.close(); graph.close()


def irregular_graph(seed):
    """a synthetic graph bent into the shapes its generator does not make (out-degrees 2-5, start 0): states WITHOUT arcs (final sinks),
    chains of one-arc states, a hub with 45 arcs, and a start state that is not state 0"""
    from voicebridge_amd import synth
    S, ntid = 3000, 3 * 48
    g = synth.decoding_graph(S, ntid, seed=seed, num_words=120, eps_frac=0.25)
    rng = np.random.default_rng(seed)
    off, arcs, final = g["arc_offset"], g["arcs"], g["final"].copy()
    lists = [list(arcs[off[s]:off[s + 1]]) for s in range(S)]
    sinks = rng.choice(np.arange(50, S), 150, replace=False)
    for s in sinks:
        lists[s] = []
        final[s] = np.float32(rng.uniform(0.0, 2.0))
    for s in rng.choice(np.setdiff1d(np.arange(50, S), sinks), 300, replace=False):
        lists[s] = lists[s][:1]
    hub = 17
    extra = np.zeros(45, arcs.dtype)
    extra["ilabel"] = rng.integers(1, ntid + 1, 45); extra["olabel"] = rng.integers(0, 120, 45)
    extra["weight"] = rng.uniform(0.1, 3.0, 45).astype(np.float32); extra["nextstate"] = rng.integers(0, S, 45)
    lists[hub] = list(extra)
    for k in range(6):      # the first states lead to the hub, so that it is reached
        a = lists[k][0].copy(); a["nextstate"] = hub; lists[k][0] = a
    # the start state becomes state 1234: states 0 and 1234 trade numbers
    perm = np.arange(S); perm[0], perm[1234] = 1234, 0
    new_lists = [None] * S
    for s in range(S):
        out = []
        for a in lists[s]:
            b = a.copy(); b["nextstate"] = perm[a["nextstate"]]; out.append(b)
        new_lists[perm[s]] = out
    new_final = np.empty_like(final); new_final[perm] = final
    new_off = np.zeros(S + 1, np.int64); new_off[1:] = np.cumsum([len(l) for l in new_lists])
    new_arcs = np.array([a for l in new_lists for a in l], dtype=arcs.dtype)
    return dict(final=new_final, arc_offset=new_off, arcs=new_arcs, start=1234), synth.tid2pdf(ntid, 48, seed=seed)


@pytest.mark.parametrize("adm", ADMISSIONS)
@pytest.mark.parametrize("lds_map", [True, False])
def test_irregular_graph_shapes(ctx, monkeypatch, adm, lds_map):
    """the packed graph of the lattice search (a state's record in front of its arcs, ids = record indices, blocks that do not straddle
    lines, at least two records per state): states without arcs, one-arc chains, a 45-arc hub, a start state other than 0 -- and the
    states the lattice reports are the graph's own"""
    import torch
    from oracle import vbo
    from voicebridge_amd import api
    if not lds_map:
        monkeypatch.setenv("VBH_NO_LDS_MAP", "1")
    g, t2p = irregular_graph(7)
    deg = np.diff(g["arc_offset"])
    assert (deg == 0).sum() >= 100 and (deg == 1).sum() >= 100 and deg.max() == 45 and g["start"] != 0
    rng = np.random.default_rng(11)
    T, P = 80, 48
    ll = (rng.standard_normal((T, P)) * 20.0 - 60.0).astype(np.float32)
    cfg = dict(beam=12.0, max_active=1500, min_active=100, lattice_beam=6.0, prune_interval=9)
    graph = api.Graph(ctx, g["final"], g["arc_offset"], g["arcs"], g["start"])
    dec = api.LatticeDecoder(ctx, graph, 1, T, admission=adm, **cfg)
    d = torch.from_numpy(ll).to(ctx.device)
    ctx.torch_sync()
    dec.decode(d, [0, T], t2p)
    got = dec.result(0)
    fst = vbo.make_fst(g["final"], g["arc_offset"], g["arcs"], g["start"])
    ref = vbo.decode_lattice_faster(fst, ll, t2p, **ORACLE_ADMISSION[adm], **cfg)
    assert_same_lattice(got, ref)
    assert got["tok_state"].max() < len(g["final"]) and len(got["tok_state"]) > T
    dec.close(); graph.close()


@pytest.mark.parametrize("every", [3, 7, 25])
@pytest.mark.parametrize("lds_map", [True, False])
def test_decode_independent_of_pruning_schedule(ctx, monkeypatch, every, lds_map):
    """the kernel prunes on its own schedule (at least 100 frames apart by default); the lattice must
    not depend on it, nor on which level of the state map (LDS / HBM) holds the states"""
    import torch
    from oracle import vbo
    from voicebridge_amd import api
    monkeypatch.setenv("VBH_PRUNE_EVERY", str(every))
    if not lds_map:
        monkeypatch.setenv("VBH_NO_LDS_MAP", "1")
    cfg = dict(beam=11.0, max_active=3000, min_active=100, lattice_beam=5.0)
    S, P, T, seed = 30000, 96, 140, 11
    g, t2p, ll = make_case(S, P, T, seed, eps_frac=0.3, spread=30.0)
    lens = [T, 61]
    foff = np.zeros(3, np.int64)
    foff[1:] = np.cumsum(lens)
    batch = np.concatenate([ll[:n] for n in lens])
    graph = api.Graph(ctx, g["final"], g["arc_offset"], g["arcs"], 0)
    dec = api.LatticeDecoder(ctx, graph, 2, max(lens), **cfg)
    d = torch.from_numpy(np.ascontiguousarray(batch)).to(ctx.device)
    ctx.torch_sync()
    dec.decode(d, foff, t2p)
    fst = vbo.make_fst(g["final"], g["arc_offset"], g["arcs"], 0)
    for u, n in enumerate(lens):
        ref = vbo.decode_lattice_faster(fst, ll[:n], t2p, **ORACLE_ADMISSION["ordered"], **cfg)
        assert_same_lattice(dec.result(u), ref)
    dec.close(); graph.close()


def test_decode_begin_end_two_contexts(ctx):
    """two batches in flight on two contexts/streams sharing one graph (what bench.py's pipeline does):
    each must give the lattices of the one-call form"""
    import torch
    from oracle import vbo
    from voicebridge_amd import api
    cfg = dict(beam=11.0, max_active=3000, min_active=100, lattice_beam=5.0)
    S, P, T, seed = 30000, 96, 120, 21
    g, t2p, ll = make_case(S, P, T, seed, eps_frac=0.3, spread=30.0)
    lens = [[T, 61, 17], [88, T]]
    ctx2 = api.Context(0, cus=2)            # confined to two CUs (vbh_ctx_create_on_cus): three utterances on two workgroups
    graph = api.Graph(ctx, g["final"], g["arc_offset"], g["arcs"], 0)
    decs = [api.LatticeDecoder(c, graph, 3, T, **cfg) for c in (ctx, ctx2)]
    bufs, foffs = [], []
    for ln in lens:
        fo = np.zeros(len(ln) + 1, np.int64)
        fo[1:] = np.cumsum(ln)
        foffs.append(fo)
        bufs.append(torch.from_numpy(np.ascontiguousarray(np.concatenate([ll[:n] for n in ln]))).to(ctx.device))
    ctx.torch_sync()
    ctx2.wait_for(ctx)                      # (nothing pending: exercises the call)
    for d, b, fo in zip(decs, bufs, foffs):
        d.decode_begin(b, fo, t2p)
    with pytest.raises(Exception):
        decs[0].decode_begin(bufs[0], foffs[0], t2p)      # a second begin before end is refused
    for d in reversed(decs):
        d.decode_end()
    fst = vbo.make_fst(g["final"], g["arc_offset"], g["arcs"], 0)
    for d, ln in zip(decs, lens):
        for u, n in enumerate(ln):
            ref = vbo.decode_lattice_faster(fst, ll[:n], t2p, **ORACLE_ADMISSION["ordered"], **cfg)
            assert_same_lattice(d.result(u), ref)
    for d in decs:
        d.close()
    graph.close(); ctx2.close()


def test_small_shared_pool_under_concurrent_decoders(ctx):
    """decoders made for small batches share one pool of slots that is smaller than the number of workgroups the device can hold:
    three host threads decoding at once make their grids wait for slots (claim_slot) -- same lattices as alone, nobody falls to the
    host's retry path"""
    import threading
    import torch
    from voicebridge_amd import api
    cfg = dict(beam=11.0, max_active=3000, min_active=100, lattice_beam=5.0)
    S, P, T, seed, NU = 20000, 64, 150, 21, 8
    g, t2p, ll = make_case(S, P, T, seed, eps_frac=0.3, spread=30.0)
    foff = np.arange(NU + 1, dtype=np.int64) * T
    graph = api.Graph(ctx, g["final"], g["arc_offset"], g["arcs"], 0)
    rng = np.random.default_rng(5)
    mats = [np.concatenate([ll[rng.permutation(T)] for _ in range(NU)]) for _ in range(3)]
    ref = []
    dec0 = api.LatticeDecoder(ctx, graph, NU, T, **cfg)
    for m in mats:
        d = torch.from_numpy(m).to(ctx.device); ctx.torch_sync()
        dec0.decode(d, foff, t2p)
        ref.append([dec0.result(u) for u in range(NU)])
    out, stats, errs = [None] * 3, [None] * 3, []

    def work(i):
        try:
            c = api.Context(0)
            dec = api.LatticeDecoder(c, graph, NU, T, **cfg)       # (finds dec0's pool: 8 slots)
            d = torch.from_numpy(mats[i]).to(c.device); c.torch_sync()
            for _ in range(3):
                dec.decode(d, foff, t2p)
            out[i] = [dec.result(u) for u in range(NU)]
            stats[i] = dec.batch_stats()
            dec.close(); c.close()
        except Exception as e:      # noqa: BLE001
            errs.append(repr(e))
    th = [threading.Thread(target=work, args=(i,)) for i in range(3)]
    for t in th: t.start()
    for t in th: t.join()
    assert not errs, errs
    for i in range(3):
        assert stats[i]["retried_in_big_slot"] == 0 and stats[i]["failed"] == 0, stats[i]
        for a, b in zip(out[i], ref[i]):
            np.testing.assert_array_equal(a["words"], b["words"])
            np.testing.assert_array_equal(a["ali"], b["ali"])
            assert a["num_toks"] == b["num_toks"] and a["num_links"] == b["num_links"]
    dec0.close(); graph.close()
