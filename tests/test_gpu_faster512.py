"""The 1-best search (FasterDecoder, the program north_star names: gmm-decode-faster) at scale against the REFERENCE's own program.

512 utterances (128 speakers x 4 x 30 s, 27 058 words) of the speech workload from PCM through the GPU chain and vbh_faster_decode_batch,
words against tests/golden/speech_faster512.npz (oracle/gen_golden_faster512.py: compute-mfcc-feats ... gmm-decode-faster --beam=13
--max-active=7000, min-active at FasterDecoderOptions' 20, and the same program with --hash-ratio=1 / 3 = the reference under another
visiting order of ProcessEmitting, kaldi-master/src/decoder/faster-decoder.cc:258-300).  With min-active 20 this is a much harsher
regime than the lattice search's (min-active 200): the reference itself makes 13.6 % word errors here (4.5 % with gmm-latgen-faster on
the same utterances), i.e. search errors everywhere -- which is what makes it a pin for the admission rule: which arcs above the final
cutoff get through decides words on dozens of utterances.

What can be asked here (tools/admission_study.py --decoder faster: the oracle's chain from PCM in each mode, CPU, profiles/r05_faster_study.log):
the regime is CHAOTIC in the features -- the reference's own algorithm in its own visiting order (oracle mode 0, which reproduces the
program bit for bit on identical features) agrees with the program on only 233 / 512 utterances once the features differ by the front
end's float tolerance (1e-4), at WER 13.301 % (-77 errors, P(improves) 0.71); the order-independent modes land beside it: exact 230 / 512,
13.216 % (-100, P 0.77), ordered 229 / 512, 13.168 % (-113, P 0.79).  So word identity per utterance says nothing here, and even the
reference's algorithm comes out a little better than the program's run on this sample; the statistical criteria are asked with that
yardstick: |dWER| <= 0.6 absolute (4 % relative; the GPU chain measured -0.455 ordered, -0.610 exact), the bootstrap probability of
improving on the reference inside [0.2, 0.9] for the default rule, and at least 200 word-identical utterances (the yardstick: 233).
The reference under another --hash-ratio (505 / 510 of 512, WER 13.563 / 13.619): its visiting order matters far less than the features.

History: this fixture FOUND a parity bug that twelve small golden cases had not (round 5): GetCutoff collects the costs in a
std::vector<BaseFloat> (faster-decoder.h:174), so its order statistics are float32-rounded; oracle and kernel used exact doubles, and with
min-active 20 the cutoff falls among dozens of exactly equal costs (a word end's fan-out over the phone-start states), where the rounding
decides whether `cost < cutoff` expands all of them or none.  Before the fix: WER 70.7 % on this sample (29 / 512 word-identical).
"""
import numpy as np
import pytest

from tests.util import load

pytestmark = pytest.mark.gpu


def test_512_utterances_against_gmm_decode_faster(tmp_path):
    import torch  # noqa: F401
    from oracle import gen_golden_e2e as e2e
    from oracle import gen_golden_faster512 as f512
    from oracle import gen_golden_speech as gs
    from tests.test_gpu_e2e import _bootci
    from tests.test_gpu_e2e512 import _errors, workload512
    from voicebridge_amd import api, synth
    g, t2p, ids, spk, waves, truth, _fx_lat = workload512()
    fx = load("speech_faster512")
    n = len(ids)
    assert int(fx["n_utt"]) == n and int(fx["words_in_graph"]) == e2e.WORDS
    ctx = api.Context(0)
    mf = api.Mfcc(ctx)
    raw, foff = mf.compute(waves)
    post = api.FeaturePipeline(ctx, cmvn=True, mode="delta", spk_xform_shape=(e2e.D, e2e.D + 1))
    n_spk = int(spk.max()) + 1
    xf = np.stack([synth.affine(e2e.D, seed=s, scale=gs.XFORM_SCALE) for s in range(n_spk)])
    u2s = spk.astype(np.int32)
    stats = post.cmvn_stats(raw, foff, u2s, n_spk)
    feats = post.apply(raw, foff, utt2spk=u2s, cmvn_stats=stats, spk_xforms=xf)
    offs, w, miv, iv = e2e.model()
    gmm = api.DiagGmm(ctx, offs, w, miv, iv)
    ll = gmm.loglikes(feats)
    ctx.sync()
    graph = api.Graph(ctx, g["final"], g["arc_offset"], g["arcs"], 0)
    C = f512.CFG
    ref = [fx["ref_words_%d" % u] for u in range(n)]
    e_ref = _errors(truth, ref)
    nw = sum(len(t) for t in truth)
    np.testing.assert_array_equal([e_ref.sum(), nw], fx["wer_counters_reference"][:2])      # (the transcripts regenerated here are the generator's)
    self_same = [int(fx["ref_same_words_hash_ratio_%g" % hr].sum()) for hr in fx["hash_ratios"]]
    self_wer = [100.0 * fx["ref_wer_counters_hash_ratio_%g" % hr][0] / fx["ref_wer_counters_hash_ratio_%g" % hr][1] for hr in fx["hash_ratios"]]
    wer_ref = 100.0 * e_ref.sum() / nw
    print("\n%d utterances, %d words; the reference's gmm-decode-faster: WER %.3f %%; under --hash-ratio=1 / 3 word-identical to itself on %s of %d, WER %s"
          % (n, nw, wer_ref, self_same, n, ["%.3f" % x for x in self_wer]))
    out = {}
    for adm in ("ordered", "exact"):
        dec = api.FasterDecoder(ctx, graph, n, int(np.diff(foff).max()), beam=C["beam"], max_active=C["max_active"], min_active=C["min_active"],
                                beam_delta=C["beam_delta"], acoustic_scale=C["acwt"], admission=adm)
        dec.decode(ll, foff, t2p)
        res = [dec.result(u) for u in range(n)]
        dec.close()
        assert all(r["ok"] == 1 for r in res)
        got = [r["words"] for r in res]
        same = [bool(np.array_equal(a, b)) for a, b in zip(got, ref)]
        e = _errors(truth, got)
        (tmp_path / adm).mkdir()
        ci = _bootci(truth, ref, got, tmp_path / adm)
        like = np.array([-(r["best_graph"] + r["best_acoustic"]) / max(r["num_frames"], 1) for r in res])
        dlike = np.abs(like - fx["ref_like_per_frame"])
        toks = float(np.mean([r["n_tokens"] / max(r["num_frames"], 1) for r in res]))
        out[adm] = dict(same=sum(same), wer=100.0 * e.sum() / nw, p_improves=ci[6] / 100.0, worse=int((e > e_ref).sum()), better=int((e < e_ref).sum()),
                        dlike_same=float(dlike[np.array(same)].max()), toks=toks)
        print("admission %-8s words equal to the reference's on %d / %d; WER %.3f %% (%+d errors); utterances with more / fewer errors than the reference: %d / %d; "
              "bootstrap: reference %.2f [%.2f, %.2f], GPU %.2f [%.2f, %.2f], P(GPU improves on the reference) = %.1f %%; |log-like per frame| differs by <= %.1e where "
              "the words agree; tokens kept per frame %.0f"
              % ((adm, sum(same), n, out[adm]["wer"], int(e.sum() - e_ref.sum()), out[adm]["worse"], out[adm]["better"]) + ci + (out[adm]["dlike_same"], toks)))
    graph.close(); gmm.close(); ctx.close()
    o = out["ordered"]
    assert abs(o["wer"] - wer_ref) <= 0.6
    assert 0.2 <= o["p_improves"] <= 0.9
    assert o["dlike_same"] < 2e-2        # (same words, possibly another alignment: 1.1e-2 measured)
    assert o["same"] >= 200 and max(self_same) >= o["same"]
    # the exact rule, for the record: not worse than the reference either
    assert out["exact"]["wer"] <= wer_ref + 0.6
