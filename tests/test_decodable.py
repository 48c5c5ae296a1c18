"""include/voicebridge_decodable.h -- the reference's DecodableInterface over GPU scores (SURVEY.md 8b seam B3, BASELINE config 2:
"MFCC + GMM-score only, decoder still CPU").

CPU: the header compiles on its own (stand-alone base class) and serves a score matrix with the reference's semantics.
GPU: the REFERENCE's LatticeFasterDecoder (oracle/_ref/vb_ref_decodable: its sources compiled by `make -C oracle ref`, linked
with the product's HIP library) decodes from scores computed on the GPU through this adapter; the result must be what the
reference computes from those scores (= the oracle's replay mode, pinned bit for bit to the reference) and its best path
must be the GPU search's."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "oracle", "_ref", "vb_ref_decodable")

PROG = r"""
#include <cstdio>
#include "voicebridge_decodable.h"
int main() {
  const float scores[6] = {1.f, 2.f, 3.f, 4.f, 5.f, 6.f};        // 2 frames x 3 pdfs
  const int32_t tid2pdf[5] = {-1, 2, 2, 0, 1};
  vbh::DecodableGpuScores<> d(scores, 2, 3, tid2pdf, 4, 0.5f);
  vbh::DecodableBase *b = &d;
  if (b->NumFramesReady() != 2 || b->NumIndices() != 4 || b->IsLastFrame(0) || !b->IsLastFrame(1)) return 1;
  if (b->LogLikelihood(0, 1) != 1.5f || b->LogLikelihood(1, 3) != 2.0f || b->LogLikelihood(1, 4) != 2.5f) return 2;
  int caught = 0;
  try { b->LogLikelihood(0, 5); } catch (const std::runtime_error &) { caught++; }
  try { b->LogLikelihood(2, 1); } catch (const std::runtime_error &) { caught++; }
  try { b->LogLikelihood(0, 0); } catch (const std::runtime_error &) { caught++; }
  if (caught != 3) return 3;
  vbh::GpuAcousticScores s;                                       // (instantiates the C-ABI calls; not run without a GPU)
  if (s.NumUtterances() != -1 && s.NumUtterances() != 0) return 4;
  printf("ok\n");
  return 0;
}
"""


def test_header_stands_alone(tmp_path):
    from voicebridge_amd import build
    if not os.path.exists(build.LIB):
        build.build()
    src = tmp_path / "t.cc"
    src.write_text(PROG)
    exe = tmp_path / "t"
    subprocess.check_call(["g++", "-std=c++11", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"), str(src), "-o", str(exe),
                           "-L" + build.LIBDIR, "-lvoicebridge_hip", "-Wl,-rpath," + build.LIBDIR])
    assert subprocess.check_output([str(exe)]).decode().strip() == "ok"


@pytest.mark.gpu
@pytest.mark.skipif(not os.access(BIN, os.X_OK), reason="oracle/_ref/vb_ref_decodable not built (make -C oracle ref in the build container)")
@pytest.mark.parametrize("seconds,P,G,n_words", [(2.5, 600, 2400, 3000), (30.0, 2000, 10000, 20000)])
def test_reference_decoder_runs_on_gpu_scores(tmp_path, seconds, P, G, n_words):
    """the second case is BASELINE config 2 at tri1 sizes: P = 2000 pdfs, G = 10 000 Gaussians, one 30 s utterance"""
    from oracle import vbo
    from voicebridge_amd import api, synth
    ctx = api.Context(0)
    waves = [synth.wave(3, seconds)]
    mf = api.Mfcc(ctx)
    raw, foff = mf.compute(waves)
    post = api.FeaturePipeline(ctx, cmvn=True, mode="delta", spk_xform_shape=(39, 40))
    stats = post.cmvn_stats(raw, foff, [0], 1)
    feats = post.apply(raw, foff, utt2spk=[0], cmvn_stats=stats, spk_xforms=synth.affine(39, seed=1)[None])
    ctx.sync()
    f_host = feats.cpu().numpy()
    offs, w, means, iv = synth.gmm_model(P, G, 39, seed=5, mean_scale=1.5, dim_scale=f_host.std(0) + 1e-3, var_spread=0.2)
    means = means + f_host.mean(0)
    miv = (means * iv).astype(np.float32)
    g, t2p = synth.lexicon_graph(n_words, P, seed=6)
    cfg = dict(beam=13.0, max_active=7000, min_active=200, lattice_beam=6.0)
    d = tmp_path
    ai = np.stack([g["arcs"]["ilabel"], g["arcs"]["olabel"], g["arcs"]["nextstate"]], 1).astype(np.int32)
    for k, v in dict(feats=f_host, offs=offs.astype(np.int32), weights=w, miv=miv, iv=iv, tid2pdf=t2p.astype(np.int32), fst_final=g["final"],
                     fst_off=g["arc_offset"].astype(np.int64), fst_arcs_i=ai, fst_arcs_w=g["arcs"]["weight"].astype(np.float32)).items():
        np.save(str(d / (k + ".npy")), np.ascontiguousarray(v))
    (d / "opts.txt").write_text("".join("%s=%s\n" % kv for kv in dict(cfg, acwt=0.083333).items()))
    subprocess.check_call([BIN, str(d)], env=dict(os.environ, OPENBLAS_NUM_THREADS="1"))
    words, ali, bw = np.load(str(d / "out_words.npy")), np.load(str(d / "out_ali.npy")), np.load(str(d / "out_best_weight.npy"))
    meta = np.load(str(d / "out_meta.npy"))
    assert meta[0] == 1 and meta[2] == len(f_host)
    # the same scores on this side
    gmm = api.DiagGmm(ctx, offs, w, miv, iv)
    ll = gmm.loglikes(feats)
    ctx.sync()
    fst = vbo.make_fst(g["final"], g["arc_offset"], g["arcs"], 0)
    # (1) what the reference computes from these scores: the oracle's replay of the reference's visiting order
    ref = vbo.decode_lattice_faster(fst, ll.cpu().numpy(), t2p, acwt=0.083333, exact_cutoff=False, **cfg)
    np.testing.assert_array_equal(words, ref["words"])
    np.testing.assert_array_equal(ali, ref["ali"])
    np.testing.assert_array_equal(bw, np.float32([ref["best_graph"], ref["best_acoustic"]]))
    # (2) the GPU search on the same scores has the same best path
    graph = api.Graph(ctx, g["final"], g["arc_offset"], g["arcs"], 0)
    dec = api.LatticeDecoder(ctx, graph, 1, len(f_host), acoustic_scale=0.083333, **cfg)
    dec.decode(ll, foff, t2p)
    got = dec.result(0)
    assert got["ok"] == 1 and got["num_frames"] == len(f_host)
    if seconds < 10:
        np.testing.assert_array_equal(got["words"], words)
        np.testing.assert_array_equal(got["ali"], ali)
        np.testing.assert_array_equal(np.float32([got["best_graph"], got["best_acoustic"]]), bw)
    else:
        # (3000 frames: the two searches visit the tokens in different orders, tests/test_gpu_e2e512.py: the best path's score may
        # differ by what one admitted arc more or less is worth)
        same = np.array_equal(got["words"], words)
        cost_ref, cost_gpu = float(bw[0] + bw[1]), float(got["best_graph"] + got["best_acoustic"])
        print("\ntri1 sizes: P %d G %d T %d, graph %d states: the reference decoder on GPU scores and the GPU search: same words %s, path cost %.3f vs %.3f"
              % (P, G, len(f_host), len(g["final"]), same, cost_ref, cost_gpu))
        assert abs(cost_ref - cost_gpu) <= 2e-3 * abs(cost_ref)
    dec.close(); graph.close(); gmm.close(); ctx.close()
