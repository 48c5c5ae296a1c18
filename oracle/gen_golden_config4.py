#!/usr/bin/env python3
"""tests/golden/config4_refpaths.npz: what the REFERENCE decodes on utterances of BASELINE config 4's feature chain (SURVEY.md 8d
config 4: LDA+MLLT+SAT -- splice +-4 (117), LDA 40 x 117 (final.mat), D = 40, per-speaker affine 40 x 41; 30 s utterances = 2998 frames,
P = 2500 / G = 15000, the 180 k-word lexicon graph, beam 13, lattice-beam 6, max-active 7000, acoustic scale 1/12).  The sibling of
gen_golden_config3.py for the chain Decode() builds when the model directory holds a final.mat (VB/scr/steps/decode_gmm.cpp:243-267:
apply-cmvn | splice-feats | transform-feats final.mat | transform-feats --utt2spk trans.JOB).  Run in the build container only:

    make -C oracle ref && python oracle/gen_golden_config4.py

Produced: best paths from the reference's own programs (oracle/_ref/bin: compute-mfcc-feats --dither=0 | compute-cmvn-stats | apply-cmvn |
splice-feats --left-context=4 --right-context=4 | transform-feats final.mat | transform-feats --utt2spk trans | gmm-latgen-faster): words,
alignment and log-likelihood per frame of every utterance; and, for the first utterance, every ROW_STEP-th row of the reference's final
40-dimensional features (the stage the chain differs from config 3's in).  The GPU box has no reference binary; tests/test_gpu_config4.py
runs the GPU path on the same seeded inputs and compares.  Inputs are regenerated from seeds by the test (voicebridge_amd/synth.py); the
fixture holds the model's placement vectors (mean / std of the features, which position the random GMM in feature space) and outputs only.
"""
import os
import re
import shutil
import subprocess
import sys
import tempfile
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
from oracle import vbo  # noqa: E402
from oracle.gen_golden_e2e import read_ark_f32  # noqa: E402
from voicebridge_amd import refio, synth  # noqa: E402

BIN = os.path.join(HERE, "_ref", "bin")
GOLD = os.path.join(ROOT, "tests", "golden")

N_UTT, N_SPK, SECONDS, ROW_STEP = 8, 2, 30.0, 37
P, G, D, WORDS = 2500, 15000, 40, 180000
CFG = dict(beam=13.0, max_active=7000, min_active=200, lattice_beam=6.0, acwt=0.083333)


def workload():
    """the seeded inputs both sides rebuild (the test imports this module's constants and this function)"""
    waves = [synth.wave(100 + i, SECONDS) for i in range(N_UTT)]
    utt2spk = np.arange(N_UTT, dtype=np.int32) % N_SPK
    xforms = np.stack([synth.affine(D, seed=40 + s) for s in range(N_SPK)])
    return waves, utt2spk, xforms, synth.lda_matrix()


def model_from(mu, std):
    offs, w, means, iv = synth.gmm_model(P, G, D, seed=4, mean_scale=1.5, dim_scale=std, var_spread=0.2)
    means = means + mu.astype(np.float32)
    return offs, w, (means * iv).astype(np.float32), iv


def main():
    t00 = time.time()
    waves, utt2spk, xforms, lda = workload()
    o = vbo.mfcc_opts()
    raws = [vbo.mfcc(w.astype(np.float32), o) for w in waves]
    stats = [sum(vbo.acc_cmvn_stats(raws[u]) for u in range(N_UTT) if utt2spk[u] == s) for s in range(N_SPK)]
    f0 = vbo.transform_feats(vbo.transform_feats(vbo.splice(vbo.apply_cmvn(raws[0], stats[utt2spk[0]])), lda), xforms[utt2spk[0]])
    mu, std = f0[:20000].mean(0), f0[:20000].std(0) + 1e-3
    offs, w, miv, iv = model_from(mu, std)
    g, t2p = synth.lexicon_graph(WORDS, P, seed=11)
    print("inputs ready: %d states / %d arcs, %.0f s" % (len(g["final"]), len(g["arcs"]), time.time() - t00), flush=True)
    d = tempfile.mkdtemp(prefix="vb_c4_")
    env = dict(os.environ, OPENBLAS_NUM_THREADS="1", OMP_NUM_THREADS="1")
    p = lambda n: os.path.join(d, n)
    out = dict(mu=mu.astype(np.float32), std=std.astype(np.float32), n_utt=N_UTT, n_spk=N_SPK, seconds=SECONDS, words_in_graph=WORDS, row_step=ROW_STEP,
               cfg=np.array(["%s=%r" % kv for kv in sorted(CFG.items())]))
    try:
        refio.write_mdl(p("final.mdl"), P // 3, 3, offs, w, miv, iv)
        refio.write_vector_fst(p("HCLG.fst"), g["final"], g["arc_offset"], g["arcs"], g["start"])
        with open(p("final.mat"), "wb") as f:
            f.write(b"\0BFM " + bytes([4]) + np.int32(D).tobytes() + bytes([4]) + np.int32(117).tobytes() + lda.astype("<f4").tobytes())
        with open(p("wav.scp"), "w") as f, open(p("utt2spk"), "w") as fu, open(p("spk2utt"), "w") as fs:
            for u in range(N_UTT):
                refio.write_wav(p("u%d.wav" % u), waves[u])
            order = sorted(range(N_UTT), key=lambda u: "s%d-u%d" % (utt2spk[u], u))
            for u in order:
                f.write("s%d-u%d %s\n" % (utt2spk[u], u, p("u%d.wav" % u)))
                fu.write("s%d-u%d s%d\n" % (utt2spk[u], u, utt2spk[u]))
            for s in range(N_SPK):
                fs.write("s%d %s\n" % (s, " ".join("s%d-u%d" % (s, u) for u in range(N_UTT) if utt2spk[u] == s)))
        refio.write_feats_ark(p("trans.ark"), {"s%d" % s: xforms[s] for s in range(N_SPK)})

        def run(name, *a):
            r = subprocess.run([os.path.join(BIN, name)] + list(a), env=env, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True)
            if r.returncode != 0:
                raise RuntimeError(name + ": " + r.stderr[-500:])
            return r.stderr
        run("compute-mfcc-feats", "--dither=0", "scp:" + p("wav.scp"), "ark:" + p("mfcc.ark"))
        run("compute-cmvn-stats", "--spk2utt=ark:" + p("spk2utt"), "ark:" + p("mfcc.ark"), "ark:" + p("cmvn.ark"))
        run("apply-cmvn", "--utt2spk=ark:" + p("utt2spk"), "ark:" + p("cmvn.ark"), "ark:" + p("mfcc.ark"), "ark:" + p("cm.ark"))
        run("splice-feats", "--left-context=4", "--right-context=4", "ark:" + p("cm.ark"), "ark:" + p("sp.ark"))
        run("transform-feats", p("final.mat"), "ark:" + p("sp.ark"), "ark:" + p("lda.ark"))
        run("transform-feats", "--utt2spk=ark:" + p("utt2spk"), "ark:" + p("trans.ark"), "ark:" + p("lda.ark"), "ark:" + p("feats.ark"))
        ft = read_ark_f32(p("feats.ark"))
        k0 = "s%d-u0" % utt2spk[0]
        assert ft[k0].shape == (2998, D), ft[k0].shape
        out["ref_feats_rows_0"] = ft[k0][::ROW_STEP].astype(np.float32)
        err = np.abs(ft[k0] - f0).max()
        print("oracle chain (splice + LDA + affine) vs the reference's features of utterance 0: max abs %.2e" % err, flush=True)
        assert err < 5e-3
        log = run("gmm-latgen-faster", "--beam=%g" % CFG["beam"], "--lattice-beam=%g" % CFG["lattice_beam"], "--max-active=%d" % CFG["max_active"],
                  "--min-active=%d" % CFG["min_active"], "--acoustic-scale=%g" % CFG["acwt"], "--allow-partial=true", "--determinize-lattice=false",
                  p("final.mdl"), p("HCLG.fst"), "ark:" + p("feats.ark"), "ark:" + p("lat.ark"), "ark,t:" + p("words.txt"), "ark,t:" + p("ali.txt"))
        for name, fn in (("words", "words.txt"), ("ali", "ali.txt")):
            for line in open(p(fn)):
                f_ = line.split()
                out["ref_%s_%d" % (name, int(f_[0].split("-u")[1]))] = np.array(f_[1:], np.int32)
        for m in re.finditer(r"Log-like per frame for utterance s\d+-u(\d+) is (\S+) over (\d+) frames", log):
            out["ref_like_per_frame_%d" % int(m.group(1))] = np.float32(m.group(2))
        assert all("ref_words_%d" % u in out for u in range(N_UTT))
        print("reference programs done, %.0f s; words per utterance %s" % (time.time() - t00, [len(out["ref_words_%d" % u]) for u in range(N_UTT)]), flush=True)
    finally:
        shutil.rmtree(d, ignore_errors=True)
    path = os.path.join(GOLD, "config4_refpaths.npz")
    np.savez_compressed(path, **out)
    print("%s  %.1f KB  (%.0f s)" % (path, os.path.getsize(path) / 1024.0, time.time() - t00))


if __name__ == "__main__":
    main()
