"""Pins oracle/vb_oracle_decode.c against the reference's LatticeFasterDecoder and
FasterDecoder (golden vectors from oracle/gen_golden.py; the reference has no decoder tests
of its own).

Two modes of the restatement are checked:
  * exact_cutoff=False replays the reference's visiting order (HashList order, online
    tightening of next_cutoff) and must reproduce the reference bit for bit: surviving
    (frame, HCLG state) tokens, their tot_cost / extra_cost, every lattice arc, the best
    path and its weight -- in beam-limited, max-active-limited, min-active-limited and
    no-final-state cases alike;
  * exact_cutoff=True is the order-independent variant (the GPU search's admission = exact).  Its
    lattice must be a sub-lattice of the reference's with identical costs, and its best
    path must be the reference's best path;
  * exact_cutoff=3 is the reference's admission rule in a canonical order (the GPU search's default,
    admission = ordered): between the other two in the work it does, the reference's best path
    wherever the visiting order does not decide it, and independent of the number of buckets where
    nothing binds.
"""
import glob
import os

import numpy as np
import pytest

from oracle import vbo
from tests.util import GOLD, load, parse_opts
from voicebridge_amd import synth

CASES = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLD, "decode_*.npz")))


def fst_from(g):
    ai = g["fst_arcs_i"]
    arcs = np.zeros(len(ai), synth.ARC_DTYPE)
    arcs["ilabel"], arcs["olabel"], arcs["nextstate"] = ai[:, 0], ai[:, 1], ai[:, 2]
    arcs["weight"] = g["fst_arcs_w"]
    return vbo.make_fst(g["fst_final"], g["fst_off"], arcs, 0)


def cfg_from(o):
    return dict(acwt=o["acwt"], beam=o["beam"], max_active=int(o["max_active"]),
                min_active=int(o["min_active"]), lattice_beam=o["lattice_beam"],
                prune_interval=int(o["prune_interval"]))


def link_table(src, dst, il, ol, gr, ac, tok_frame, tok_state):
    """canonical, order-free representation of the lattice arcs"""
    rows = [(int(tok_frame[s]), int(tok_state[s]), int(tok_frame[d]), int(tok_state[d]), int(i), int(o),
             np.float32(g).tobytes(), np.float32(a).tobytes())
            for s, d, i, o, g, a in zip(src, dst, il, ol, gr, ac)]
    return sorted(rows)


def ref_links(g):
    return link_table(g["ref_lat_link_src"], g["ref_lat_link_dst"], g["ref_lat_link_ilabel"],
                      g["ref_lat_link_olabel"], g["ref_lat_link_graph"], g["ref_lat_link_acoustic"],
                      g["ref_lat_tok_frame"], g["ref_lat_tok_state"])


def our_links(r):
    return link_table(r["link_src"], r["link_dst"], r["link_ilabel"], r["link_olabel"], r["link_graph"],
                      r["link_acoustic"], r["tok_frame"], r["tok_state"])


def test_have_cases():
    assert len(CASES) == 12


@pytest.mark.parametrize("name", CASES)
def test_lattice_decoder_replays_reference_exactly(name):
    g = load(name)
    r = vbo.decode_lattice_faster(fst_from(g), g["loglikes"], g["tid2pdf"], exact_cutoff=False,
                                  **cfg_from(parse_opts(g["opts"])))
    assert [r["ok"], r["reached_final"], r["num_frames"]] == g["ref_lat_meta"].tolist()
    np.testing.assert_array_equal(r["cost_offsets"], g["ref_lat_cost_offsets"])
    np.testing.assert_array_equal(r["tok_frame"], g["ref_lat_tok_frame"])
    np.testing.assert_array_equal(r["tok_state"], g["ref_lat_tok_state"])        # "lattice state IDs"
    np.testing.assert_array_equal(r["tok_cost"], g["ref_lat_tok_cost"])          # float32 bit-exact
    np.testing.assert_array_equal(r["tok_extra"], g["ref_lat_tok_extra"])
    assert our_links(r) == ref_links(g)
    np.testing.assert_array_equal(r["words"], g["ref_lat_words"])
    np.testing.assert_array_equal(r["ali"], g["ref_lat_ali"])
    np.testing.assert_array_equal(np.float32([r["best_graph"], r["best_acoustic"]]), g["ref_lat_best_weight"])
    frc = g["ref_lat_fmeta"][0]
    assert (np.isinf(frc) and np.isinf(r["final_relative_cost"])) or r["final_relative_cost"] == frc
    # raw lattice size as GetRawLattice reports it
    assert [len(r["tok_state"]), len(r["link_src"])] == g["ref_lat_raw_size"].tolist()


@pytest.mark.parametrize("name", CASES)
def test_exact_cutoff_variant_is_sublattice_with_same_best_path(name):
    g = load(name)
    o = parse_opts(g["opts"])
    r = vbo.decode_lattice_faster(fst_from(g), g["loglikes"], g["tid2pdf"], exact_cutoff=True, **cfg_from(o))
    np.testing.assert_array_equal(r["words"], g["ref_lat_words"])
    np.testing.assert_array_equal(r["ali"], g["ref_lat_ali"])
    np.testing.assert_array_equal(np.float32([r["best_graph"], r["best_acoustic"]]), g["ref_lat_best_weight"])
    if r["n_frames_maxactive_bound"] == 0 and r["n_frames_minactive_bound"] == 0:
        # purely beam-limited search: every token/arc we keep is in the reference's lattice,
        # with the same float32 costs
        ref_tok = {(int(f), int(s)): c for f, s, c in zip(
            g["ref_lat_tok_frame"], g["ref_lat_tok_state"], g["ref_lat_tok_cost"])}
        for f, s, c in zip(r["tok_frame"], r["tok_state"], r["tok_cost"]):
            assert ref_tok[(int(f), int(s))] == c
        assert set(our_links(r)) <= set(ref_links(g))


@pytest.mark.parametrize("name", CASES)
def test_faster_decoder_matches_reference(name):
    g = load(name)
    o = parse_opts(g["opts"])
    r = vbo.decode_faster(fst_from(g), g["loglikes"], g["tid2pdf"], acwt=o["acwt"], beam=o["beam"],
                          max_active=int(o["max_active"]), min_active=20)
    assert [r["ok"], r["reached_final"], r["num_frames"]] == g["ref_fast_meta"].tolist()
    np.testing.assert_array_equal(r["words"], g["ref_fast_words"])
    np.testing.assert_array_equal(r["ali"], g["ref_fast_ali"])
    # totals are re-associated by RemoveEpsLocal in the reference (faster-decoder.cc:145)
    np.testing.assert_allclose([r["best_graph"], r["best_acoustic"]], g["ref_fast_best_weight"], rtol=2e-6)


@pytest.mark.parametrize("mode", [1, 3])
@pytest.mark.parametrize("name", CASES)
def test_faster_decoder_exact_variant_vs_reference(name, mode):
    """the order-independent FasterDecoder (what the GPU implements: mode 3 = ordered admission, the product default, mode 1 = the
    exact subset) against the reference's golden vectors: same words, alignment and weights on every golden case, for any hash size"""
    g = load(name)
    o = parse_opts(g["opts"])
    kw = dict(acwt=o["acwt"], beam=o["beam"], max_active=int(o["max_active"]), min_active=20)
    ex = vbo.decode_faster(fst_from(g), g["loglikes"], g["tid2pdf"], exact_cutoff=mode, **kw)
    assert [ex["ok"], ex["reached_final"], ex["num_frames"]] == g["ref_fast_meta"].tolist()
    # (every case in both modes, decode_g1_maxact included: what had looked like a path decided by the visiting order there was the float32
    # rounding of GetCutoff's order statistics, faster-decoder.h:174 -- round 5)
    np.testing.assert_array_equal(ex["words"], g["ref_fast_words"])
    np.testing.assert_array_equal(ex["ali"], g["ref_fast_ali"])
    np.testing.assert_allclose([ex["best_graph"], ex["best_acoustic"]], g["ref_fast_best_weight"], rtol=2e-6)
    for hr in (1.0, 3.7):
        ex2 = vbo.decode_faster(fst_from(g), g["loglikes"], g["tid2pdf"], exact_cutoff=mode, hash_ratio=hr, **kw)
        np.testing.assert_array_equal(ex2["ali"], ex["ali"])
        assert (ex2["best_graph"], ex2["best_acoustic"]) == (ex["best_graph"], ex["best_acoustic"])


def test_reference_lattice_depends_on_hash_size_but_exact_variant_does_not():
    """The reference's extra arcs depend on its hash table size (hash_ratio, and in
    gmm-latgen-faster on the utterances decoded earlier by the same decoder object); the
    order-independent lattice is the same for all of them and contained in each."""
    g = load("decode_g1_beam10")
    o = cfg_from(parse_opts(g["opts"]))
    f = fst_from(g)
    ex = set(our_links(vbo.decode_lattice_faster(f, g["loglikes"], g["tid2pdf"], exact_cutoff=True, **o)))
    for hr in (1.0, 2.0, 3.7, 50.0):
        r = vbo.decode_lattice_faster(f, g["loglikes"], g["tid2pdf"], exact_cutoff=False, hash_ratio=hr, **o)
        assert r["n_frames_maxactive_bound"] == 0
        r2 = vbo.decode_lattice_faster(f, g["loglikes"], g["tid2pdf"], exact_cutoff=True, hash_ratio=hr, **o)
        assert set(our_links(r2)) == ex


@pytest.mark.parametrize("name", CASES)
def test_ordered_admission_mode(name):
    """exact_cutoff=3 (vb_oracle.h): the reference's prefix-minimum admission over buckets of the source state's hash.  It admits at
    least what the order-independent rule admits and at most what the superset rule (mode 2: the initial next_cutoff for every arc)
    admits, frame by frame on equal tokens -- over a whole utterance that shows as tokens created; its best path is the reference's in
    every golden case but the one whose path the visiting order itself decides (a random graph with max-active binding in every frame,
    where the superset rule ends elsewhere too); runs are deterministic."""
    g = load(name)
    o = parse_opts(g["opts"])
    fst, cfg = fst_from(g), cfg_from(o)
    r1 = vbo.decode_lattice_faster(fst, g["loglikes"], g["tid2pdf"], exact_cutoff=1, **cfg)
    r2 = vbo.decode_lattice_faster(fst, g["loglikes"], g["tid2pdf"], exact_cutoff=2, **cfg)
    r3 = vbo.decode_lattice_faster(fst, g["loglikes"], g["tid2pdf"], exact_cutoff=3, adm_bucket_bits=6, **cfg)
    r3b = vbo.decode_lattice_faster(fst, g["loglikes"], g["tid2pdf"], exact_cutoff=3, adm_bucket_bits=6, **cfg)
    for k in ("tok_frame", "tok_state", "tok_cost", "tok_extra", "link_src", "link_dst", "link_graph", "link_acoustic", "words", "ali"):
        np.testing.assert_array_equal(r3[k], r3b[k], err_msg=k)
    assert r3["ok"] == 1 and r3["num_frames"] == g["loglikes"].shape[0]
    assert r1["n_tok_created"] <= r3["n_tok_created"] <= r2["n_tok_created"]
    if name != "decode_g1_maxact":
        np.testing.assert_array_equal(r3["words"], g["ref_lat_words"])
        np.testing.assert_array_equal(r3["ali"], g["ref_lat_ali"])
        np.testing.assert_array_equal(np.float32([r3["best_graph"], r3["best_acoustic"]]), g["ref_lat_best_weight"])
    else:
        assert not np.array_equal(r2["words"], g["ref_lat_words"]) or not np.array_equal(r2["ali"], g["ref_lat_ali"])
    # with one bucket (bits = 1 is the fewest: two) the rule comes close to the superset rule, with 2^20 to one arc order per state
    for bits in (1, 12, 20):
        rb = vbo.decode_lattice_faster(fst, g["loglikes"], g["tid2pdf"], exact_cutoff=3, adm_bucket_bits=bits, **cfg)
        assert rb["ok"] == 1 and r1["n_tok_created"] <= rb["n_tok_created"] <= r2["n_tok_created"]
