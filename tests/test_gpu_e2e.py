"""End-to-end parity (PCM -> words) in the regime the bench's headline is measured in, against the REFERENCE's own programs
(VERDICT r2, next-round item 1).  tests/golden/speech_e2e.npz (oracle/gen_golden_e2e.py, made in the build container) holds what
compute-mfcc-feats | compute-cmvn-stats | apply-cmvn | add-deltas | transform-feats | gmm-latgen-faster decoded for 64 utterances
of the speech workload (16 speakers x 4 x 30 s, ~53 words each, max-active deciding the cutoff in every sixth frame), and for four
of them the reference's MFCCs, final features and rows of its score matrix.  Here:

  * the whole GPU chain from PCM: how many utterances get the reference's words, WER against the transcripts on both sides
    and the bootstrap interval ComputeWerBootci gives for the pair;
  * stage swaps on the four staged utterances -- the reference's MFCCs into the GPU's post-processing, the reference's
    features into the GPU's scoring, the GPU's features into the oracle's scoring -- each followed by the GPU search, with
    the per-stage error asserted, so that a word difference can be attributed to a stage;
  * the strict scoring mode (the reference's LogSumExp arithmetic, vbh_gmm_set_lse_mode) beside the default one.

Tolerances (stated contract: SURVEY.md 7): MFCC <= 1e-3 abs on cepstra of magnitude 10-100; features after CMVN / deltas /
transform <= 2e-3 abs; log-likelihoods <= 1e-4 relative (strict mode: 2e-5, the oracle's own distance from the reference).
"""
import functools
import os

import numpy as np
import pytest

from tests.util import load

pytestmark = pytest.mark.gpu

MFCC_ATOL = 1e-3
FEAT_ATOL = 2e-3
LL_RTOL = 1e-4
LL_RTOL_STRICT = 2e-5


@functools.lru_cache(maxsize=1)
def workload():
    from oracle import gen_golden_e2e as e2e
    from tests.test_gpu_config3 import lexicon_graph
    g, t2p, lex = lexicon_graph()
    ids, spk, waves, hop, truth = e2e.test_set(lex)
    return g, t2p, ids, spk, waves, truth


@pytest.fixture(scope="module")
def chain():
    import torch
    from oracle import gen_golden_e2e as e2e
    from oracle import gen_golden_speech as gs
    from voicebridge_amd import api, synth
    fx = load("speech_e2e")
    assert int(fx["n_utt"]) == e2e.N_UTT and int(fx["words_in_graph"]) == e2e.WORDS
    g, t2p, ids, spk, waves, truth = workload()
    n = len(ids)
    ctx = api.Context(0)
    mf = api.Mfcc(ctx)
    raw, foff = mf.compute([w for w in waves])
    post = api.FeaturePipeline(ctx, cmvn=True, mode="delta", spk_xform_shape=(e2e.D, e2e.D + 1))
    n_spk = int(spk.max()) + 1
    xf = np.stack([synth.affine(e2e.D, seed=s, scale=gs.XFORM_SCALE) for s in range(n_spk)])
    u2s = spk.astype(np.int32)
    stats = post.cmvn_stats(raw, foff, u2s, n_spk)
    feats = post.apply(raw, foff, utt2spk=u2s, cmvn_stats=stats, spk_xforms=xf)
    ctx.sync()
    offs, w, miv, iv = e2e.model()
    gmm = api.DiagGmm(ctx, offs, w, miv, iv)
    graph = api.Graph(ctx, g["final"], g["arc_offset"], g["arcs"], 0)
    cfg = dict(beam=e2e.CFG["beam"], max_active=e2e.CFG["max_active"], min_active=e2e.CFG["min_active"], lattice_beam=e2e.CFG["lattice_beam"])
    T = int(np.diff(foff).max())
    dec = api.LatticeDecoder(ctx, graph, n, T, acoustic_scale=e2e.CFG["acwt"], **cfg)

    def search(ll, fo):
        dec.decode(ll, fo, t2p)
        return [dec.result(u, arrays=False) for u in range(len(fo) - 1)]
    yield dict(ctx=ctx, fx=fx, torch=torch, mf=mf, post=post, gmm=gmm, raw=raw, feats=feats, foff=foff, u2s=u2s, n_spk=n_spk, xf=xf, truth=truth,
               search=search, model=(offs, w, miv, iv), n=n, n_stage=int(fx["n_stage"]), row_step=int(fx["row_step"]), dev=ctx.device)
    dec.close(); graph.close(); gmm.close(); ctx.close()


def _wer(truth, words):
    from voicebridge_amd import dist as vdist
    c = vdist.score(truth, words)
    return 100.0 * c[0] / c[1], c


def _bootci(truth, hyp_a, hyp_b, tmp_path):
    """the host library's ComputeWerBootci (byte-equal to the reference's compute-wer-bootci, tests/test_host_scoring.py) on two
    hypothesis sets: (wer_a, lo_a, hi_a, wer_b, lo_b, hi_b, probability that b improves on a)"""
    import ctypes as C
    import re
    from voicebridge_amd import build
    host = C.CDLL(build.HOST_LIB)
    names = ["utt%04d" % u for u in range(len(truth))]
    for fn, seqs in (("ref.txt", truth), ("a.txt", hyp_a), ("b.txt", hyp_b)):
        with open(os.path.join(str(tmp_path), fn), "w") as f:
            for k, s in zip(names, seqs):
                f.write("%s %s\n" % (k, " ".join(str(int(x)) for x in s)))
    q = lambda n: os.path.join(str(tmp_path), n)
    a = ["compute-wer-bootci", "--mode=present", "ark:" + q("ref.txt"), "ark:" + q("a.txt"), q("ci"), "ark:" + q("b.txt")]
    arr = (C.c_char_p * len(a))(*[x.encode() for x in a])
    assert host.vb_ComputeWerBootci(len(a), arr, q("log").encode()) == 0
    out = open(q("log")).read()
    m1 = re.search(r"Set1: %WER ([0-9.]+) 95% Conf Interval \[ ([0-9.]+), ([0-9.]+) \]", out)
    m2 = re.search(r"Set2: %WER ([0-9.]+) 95% Conf Interval \[ ([0-9.]+), ([0-9.]+) \]", out)
    m3 = re.search(r"Probability of Set2 improving Set1: ([0-9.]+)", out)
    assert m1 and m2 and m3, out
    return tuple(float(x) for x in m1.groups() + m2.groups() + m3.groups())


def test_words_from_pcm_against_the_reference_programs(chain, tmp_path):
    s = chain
    fx = s["fx"]
    ll = s["gmm"].loglikes(s["feats"])
    res = s["search"](ll, s["foff"])
    got = [r["words"] for r in res]
    ref = [fx["ref_words_%d" % u] for u in range(s["n"])]
    same = [bool(np.array_equal(a, b)) for a, b in zip(got, ref)]
    wer_gpu, c_gpu = _wer(s["truth"], got)
    wer_ref, c_ref = _wer(s["truth"], ref)
    np.testing.assert_array_equal(c_ref, fx["wer_counters_reference"])      # (the transcripts regenerated here are the generator's)
    like = np.array([-(r["best_graph"] + r["best_acoustic"]) / r["num_frames"] for r in res])
    dlike = np.abs(like - fx["ref_like_per_frame"])
    ci = _bootci(s["truth"], ref, got, tmp_path)
    print("\nGPU chain from PCM vs the reference's programs: words equal on %d / %d utterances; WER reference %.2f %%, GPU %.2f %% (%d words); "
          "|log-like per frame| differs by <= %.2e (equal-word utterances: %.2e); bootstrap: reference %.2f [%.2f, %.2f], GPU %.2f [%.2f, %.2f], "
          "P(GPU improves on reference) = %.2f" % ((sum(same), s["n"], wer_ref, wer_gpu, int(c_ref[1]), dlike.max(), dlike[np.array(same)].max()) + ci))
    assert all(r["ok"] == 1 for r in res)
    # What can be asked here.  The reference is order dependent in this regime: the same program on the same inputs with --hash-ratio=1 / 3
    # instead of 2 (another visiting order in ProcessEmitting, so other arcs above the final next_cutoff get through) changes the words
    # of 1 / 2 of these 64 utterances (fixture: ref_same_words_hash_ratio_*).  The GPU search applies the reference's rule in a canonical
    # order (admission = ordered), so it differs from the reference's default run the way the reference's other runs do, plus the float
    # tolerance of the front end: 60 / 64 here, errors on both sides.  The statistical statement -- no bias against the reference -- is
    # made on 512 utterances (tests/test_gpu_e2e512.py); this sample is too small for more than: nine in ten utterances word-identical,
    # each side's WER inside the other's 95 % bootstrap interval, and the GPU not on the losing side of every resampling (round 3's
    # order-independent rule was: P(GPU improves) = 0.00 on this sample -- and 0.98 on the 512).
    self_same = min(int(fx["ref_same_words_hash_ratio_1"].sum()), int(fx["ref_same_words_hash_ratio_3"].sum()))
    assert sum(same) >= self_same - 4 and sum(same) >= (s["n"] * 9) // 10
    assert ci[1] <= wer_gpu <= ci[2] and ci[4] <= wer_ref <= ci[5]
    assert abs(wer_gpu - wer_ref) < 0.25 * (ci[2] - ci[1])
    assert ci[6] >= 5.0          # (ComputeWerBootci prints the probability in per cent)
    # where the words agree the path is the same and its score differs by the float tolerance of the chain only
    assert dlike[np.array(same)].max() < 2e-3


def _stage_inputs(s):
    fx, torch, dev = s["fx"], s["torch"], s["dev"]
    k = s["n_stage"]
    T = int(s["foff"][1] - s["foff"][0])
    ref_mfcc = np.concatenate([fx["ref_mfcc_%d" % u] for u in range(k)])
    ref_feats = np.concatenate([fx["ref_feats_%d" % u] for u in range(k)])
    fo = s["foff"][:k + 1]
    return k, T, ref_mfcc, ref_feats, fo, torch, dev


def test_stage_errors_against_the_reference(chain):
    """per-stage distance of the GPU from the reference on the staged utterances, each stage fed the REFERENCE's input"""
    s = chain
    k, T, ref_mfcc, ref_feats, fo, torch, dev = _stage_inputs(s)
    fx = s["fx"]
    # MFCC from PCM
    e_mfcc = float(np.abs(s["raw"][:k * T].cpu().numpy() - ref_mfcc).max())
    # post-processing on the reference's MFCCs (the four staged utterances are speaker 0's, so the CMVN statistics are the speaker's)
    d_ref_mfcc = torch.from_numpy(ref_mfcc).to(dev)
    u2s = np.zeros(k, np.int32)
    st = s["post"].cmvn_stats(d_ref_mfcc, fo, u2s, 1)
    f_ = s["post"].apply(d_ref_mfcc, fo, utt2spk=u2s, cmvn_stats=st, spk_xforms=s["xf"][:1])
    s["ctx"].sync()
    e_post = float(np.abs(f_.cpu().numpy() - ref_feats).max())
    # scoring on the reference's features, both modes, against the rows of gmm-compute-likes
    d_ref_feats = torch.from_numpy(ref_feats).to(dev)
    rows = np.concatenate([fx["ref_scores_rows_%d" % u] for u in range(k)])
    P_ref = rows.shape[1]
    e_ll = {}
    for mode in ("fast", "strict"):
        s["gmm"].set_lse_mode(mode)
        ll = s["gmm"].loglikes(d_ref_feats).cpu().numpy()
        got = np.concatenate([ll[u * T:(u + 1) * T:s["row_step"], :P_ref] for u in range(k)])
        e_ll[mode] = float((np.abs(got - rows) / np.abs(rows)).max())
    s["gmm"].set_lse_mode("fast")
    print("\nstage errors vs the reference (staged utterances): MFCC max abs %.2e; post-processing of the reference's MFCCs max abs %.2e; "
          "scores of the reference's features max rel %.2e (default) / %.2e (strict LogSumExp)" % (e_mfcc, e_post, e_ll["fast"], e_ll["strict"]))
    assert e_mfcc <= MFCC_ATOL
    assert e_post <= FEAT_ATOL
    assert e_ll["fast"] <= LL_RTOL and e_ll["strict"] <= LL_RTOL_STRICT


def test_stage_swaps_attribute_word_differences(chain):
    """the GPU search behind every combination of reference / GPU / oracle stages: which stage moves words, if any"""
    from oracle import vbo
    s = chain
    k, T, ref_mfcc, ref_feats, fo, torch, dev = _stage_inputs(s)
    fx = s["fx"]
    offs, w, miv, iv = s["model"]
    ref = [fx["ref_words_%d" % u] for u in range(k)]
    gc = vbo.gmm_gconsts(w, miv, iv)
    rows = {}

    def count(name, ll):
        res = s["search"](ll, fo)
        rows[name] = sum(int(np.array_equal(r["words"], ref[u])) for u, r in enumerate(res))
    gpu_feats = s["feats"][:k * T]
    d_ref_mfcc = torch.from_numpy(ref_mfcc).to(dev)
    u2s = np.zeros(k, np.int32)
    st = s["post"].cmvn_stats(d_ref_mfcc, fo, u2s, 1)
    feats_from_ref_mfcc = s["post"].apply(d_ref_mfcc, fo, utt2spk=u2s, cmvn_stats=st, spk_xforms=s["xf"][:1])
    d_ref_feats = torch.from_numpy(ref_feats).to(dev)
    for mode in ("fast", "strict"):
        s["gmm"].set_lse_mode(mode)
        count("GPU MFCC -> GPU post -> GPU scores (%s)" % mode, s["gmm"].loglikes(gpu_feats))
        count("reference MFCC -> GPU post -> GPU scores (%s)" % mode, s["gmm"].loglikes(feats_from_ref_mfcc))
        count("reference features -> GPU scores (%s)" % mode, s["gmm"].loglikes(d_ref_feats))
    s["gmm"].set_lse_mode("fast")
    o_gpu = np.concatenate([vbo.gmm_loglikes(offs, gc, miv, iv, gpu_feats[u * T:(u + 1) * T].cpu().numpy()) for u in range(k)])
    count("GPU features -> oracle scores", torch.from_numpy(o_gpu).to(dev))
    o_ref = np.concatenate([vbo.gmm_loglikes(offs, gc, miv, iv, ref_feats[u * T:(u + 1) * T]) for u in range(k)])
    count("reference features -> oracle scores", torch.from_numpy(o_ref).to(dev))
    print("\nGPU search behind swapped stages, utterances (of %d) with the reference's words:" % k)
    for name, v in rows.items():
        print("   %-60s %d" % (name, v))
    # on the reference's own features and the checker's scores the GPU search must find the reference's words (the searches
    # agree bit for bit on identical scores; the oracle's scores are within one ulp of gmm-compute-likes)
    assert rows["reference features -> oracle scores"] == k
    assert rows["reference features -> GPU scores (strict)"] == k


def test_search_is_deterministic(chain):
    """identical input must give identical retries, counters and lattices, run after run (the order of a
    frame's tokens inside the workspace may differ between runs -- atomics decide it -- the delivered lattice may not)"""
    s = chain
    ll = s["gmm"].loglikes(s["feats"])
    first = None
    for rep in range(4):
        res = s["search"](ll, s["foff"])
        sig = [tuple(int(r[k]) for k in ("n_tokens_expanded", "n_arcs_expanded", "n_tokens_created", "n_links_created", "n_frames_maxactive_bound",
                                          "n_frames_minactive_bound", "num_toks", "num_links")) + (tuple(int(x) for x in r["words"]), float(r["best_graph"]), float(r["best_acoustic"]))
               for r in res]
        if first is None:
            first = sig
        else:
            assert sig == first, "run %d differs from run 0" % rep


def test_compressed_mfcc_bytes_against_the_reference_writer(chain, tmp_path):
    """SURVEY a9 / VERDICT r2 weak item 10: MakeMfcc stores cepstra as CompressedMatrix (one byte per value between per-column
    percentile points, K/matrix/compressed-matrix.cc:129-187, 457-480).  The host codec is byte-identical to the reference's writer
    for EQUAL input (tests/test_host_io.py); the GPU's cepstra differ from the reference's by up to 2.7e-4, so some bytes differ:
    counted here on the staged utterances -- the reference's MFCCs and the GPU's, both through the same codec."""
    import ctypes as C
    from voicebridge_amd import build, refio
    s = chain
    k, T = s["n_stage"], int(s["foff"][1] - s["foff"][0])
    host = C.CDLL(build.HOST_LIB)
    gpu = s["raw"][:k * T].cpu().numpy()
    blobs = {}
    for tag, mats in (("ref", [s["fx"]["ref_mfcc_%d" % u] for u in range(k)]), ("gpu", [gpu[u * T:(u + 1) * T] for u in range(k)])):
        refio.write_feats_ark(str(tmp_path / (tag + ".ark")), {"utt%d" % u: m for u, m in enumerate(mats)})
        assert host.vbio_copy_feats(("ark:%s" % (tmp_path / (tag + ".ark"))).encode(), ("ark:%s" % (tmp_path / (tag + ".cm"))).encode(), 1) == 0
        blobs[tag] = np.frombuffer((tmp_path / (tag + ".cm")).read_bytes(), np.uint8)
    assert blobs["ref"].size == blobs["gpu"].size
    neq = blobs["ref"] != blobs["gpu"]
    d = np.abs(blobs["ref"].astype(np.int32) - blobs["gpu"].astype(np.int32))
    print("\nCompressedMatrix bytes of the GPU's MFCCs vs the reference's (%d utterances, %d bytes): %.3f %% differ, by at most %d"
          % (k, neq.size, 100.0 * neq.mean(), int(d.max())))
    # measured r03: see DESIGN.md 5; the quantiser's step is ~0.1-0.4 cepstral units, the MFCC difference <= 2.7e-4
    assert neq.mean() < 0.02
