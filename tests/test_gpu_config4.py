"""BASELINE config 4's feature chain at the size the bench runs the search (VERDICT r4 "next" 1): LDA+MLLT+SAT -- CMVN, splice +-4 (117),
LDA 40 x 117 (final.mat), per-speaker affine 40 x 41, D = 40 -- on 8 utterances of 30 s (T = 2998), P = 2500 / G = 15000, the 180 k-word
lexicon graph (~2 M states), beam 13, lattice-beam 6, max-active 7000.  The whole GPU path from PCM to lattices, checked

  * against tests/golden/config4_refpaths.npz: words, alignment and log-likelihood per frame the REFERENCE's own programs decode for the
    same seeded inputs (oracle/gen_golden_config4.py: compute-mfcc-feats | compute-cmvn-stats | apply-cmvn | splice-feats | transform-feats
    final.mat | transform-feats --utt2spk trans | gmm-latgen-faster -- the chain VB/scr/steps/decode_gmm.cpp:243-267 builds when the model
    directory holds a final.mat), and rows of the reference's final 40-dimensional features;
  * against the oracle in the GPU's admission mode on the GPU's own scores: bit-exact token sets, float32 costs, every arc, best path.

The 8-GPU point of config 4 is the driver's to run (one GPU per gpurun call here); the chain through the step API (Decode() on a model
directory with final.mat / splice_opts) is tests/test_gpu_steps.py::test_decode_lda_mllt_sat_model_against_the_reference_chain.
"""
import numpy as np
import pytest

from tests.util import ORACLE_ADMISSION, load
from tests.test_gpu_config3 import LAT_KEYS, lexicon_graph

pytestmark = pytest.mark.gpu


def test_lda_mllt_sat_chain_at_bench_scale():
    import torch  # noqa: F401
    from oracle import gen_golden_config4 as c4
    from oracle import vbo
    from voicebridge_amd import api
    fx = load("config4_refpaths")
    assert int(fx["n_utt"]) == c4.N_UTT and int(fx["words_in_graph"]) == c4.WORDS and int(fx["row_step"]) == c4.ROW_STEP
    ctx = api.Context(0)
    waves, utt2spk, xforms, lda = c4.workload()
    mf = api.Mfcc(ctx)
    raw, foff = mf.compute(waves)
    post = api.FeaturePipeline(ctx, cmvn=True, mode="splice", splice_left=4, splice_right=4, lda=lda, spk_xform_shape=(c4.D, c4.D + 1))
    stats = post.cmvn_stats(raw, foff, utt2spk, c4.N_SPK)
    feats = post.apply(raw, foff, utt2spk=utt2spk, cmvn_stats=stats, spk_xforms=xforms)
    ctx.sync()
    assert feats.shape[1] == c4.D and all(int(d) == 2998 for d in np.diff(foff))
    # the stage config 4 differs from config 3 in: splice + LDA + affine, against rows of the reference's own features
    f0 = feats[foff[0]:foff[1]].cpu().numpy()
    err = np.abs(f0[::c4.ROW_STEP] - fx["ref_feats_rows_0"]).max()
    print("\nfinal features of utterance 0 vs the reference's (every %d-th row): max abs %.2e" % (c4.ROW_STEP, err))
    assert err < 5e-3          # (MFCC tolerance 1e-3 through a 117 -> 40 contraction and a 40 x 41 affine map)
    offs, w, miv, iv = c4.model_from(fx["mu"], fx["std"])
    gmm = api.DiagGmm(ctx, offs, w, miv, iv)
    ll = gmm.loglikes(feats)
    ctx.sync()
    g, t2p, _lex = lexicon_graph()
    assert c4.WORDS == int(load("config3_refpaths")["words_in_graph"])      # (the same graph as config 3's fixture: built once per run)
    graph = api.Graph(ctx, g["final"], g["arc_offset"], g["arcs"], 0)
    cfg = dict(beam=c4.CFG["beam"], max_active=c4.CFG["max_active"], min_active=c4.CFG["min_active"], lattice_beam=c4.CFG["lattice_beam"])
    dec = api.LatticeDecoder(ctx, graph, c4.N_UTT, 2998, acoustic_scale=c4.CFG["acwt"], **cfg)
    dec.decode(ll, foff, t2p)
    assert dec.batch_stats()["failed"] == 0
    ll_host = ll.cpu().numpy()
    fst = vbo.make_fst(g["final"], g["arc_offset"], g["arcs"], 0)
    same = 0
    for u in range(c4.N_UTT):
        got = dec.result(u)
        assert got["ok"] == 1
        ref = vbo.decode_lattice_faster(fst, ll_host[foff[u]:foff[u + 1]], t2p, acwt=c4.CFG["acwt"], **ORACLE_ADMISSION["ordered"], **cfg)
        for k in LAT_KEYS:
            np.testing.assert_array_equal(got[k], ref[k], err_msg="utt %d %s" % (u, k))
        eq = np.array_equal(got["words"], fx["ref_words_%d" % u]) and np.array_equal(got["ali"], fx["ref_ali_%d" % u])
        same += int(eq)
        like = -(got["best_graph"] + got["best_acoustic"]) / got["num_frames"]
        if eq:
            np.testing.assert_allclose(like, float(fx["ref_like_per_frame_%d" % u]), rtol=2e-4, atol=2e-4)
    print("words and alignments equal to the reference programs' on %d / %d utterances" % (same, c4.N_UTT))
    assert same == c4.N_UTT
    dec.close(); graph.close(); gmm.close(); ctx.close()
