#!/usr/bin/env python3
"""tests/golden/speech_refpaths.npz: the REFERENCE's LatticeFasterDecoder on utterances of the bench's speech workload
(voicebridge_amd/synth_speech.py: 30 s utterances that are word sequences of the lexicon, ~50 words each; an acoustic model
fit to other utterances; beam 13, lattice-beam 6, max-active 7000, acoustic scale 1/12, the 180 k-word graph the config-3
fixture uses) -- the regime the bench's headline is measured in: WER of a few per cent, ~2000 tokens per frame, max-active
deciding the cutoff in every sixth frame.  Run in the build container only:

    make -C oracle ref && python oracle/gen_golden_speech.py

What is stored:
(1) the acoustic model's per-pdf mean / variance (the mixture components are regenerated from seeds,
    synth_speech.mixture_from_stats), fit to N_TRAIN training utterances through the oracle's feature chain;
(2) per test utterance, on the ORACLE's scores (its MFCC / CMVN / deltas / transform chain and all-pdf log-likelihoods): words,
    alignment and weight of the best path of the reference's own decoder (oracle/_ref/vb_ref_harness around its
    LatticeFasterDecoder), and how its raw lattice relates to the order-independent admission rule's (oracle
    exact_cutoff=1 = what the GPU search computes bit for bit): arcs either side lacks, tokens either side lacks;
(3) WER of both against the transcripts.
tests/test_gpu_config3.py regenerates the waveforms from seeds (numpy), computes the same scores with the
oracle, runs the GPU search on them and expects the reference's best paths bit for bit.
"""
import os
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
from oracle.gen_golden_config3 import link_rows  # noqa: E402
from voicebridge_amd import dist as vdist  # noqa: E402
from voicebridge_amd import synth, synth_speech as ss  # noqa: E402

HARNESS = os.path.join(HERE, "_ref", "vb_ref_harness")
GOLD = os.path.join(ROOT, "tests", "golden")

N_UTT, N_TRAIN, SECONDS, DEPTH = 8, 128, 30.0, 0.25
P, G, D, WORDS = 2500, 15000, 39, 180000
XFORM_SCALE = 0.005
CFG = dict(beam=13.0, max_active=7000, min_active=200, lattice_beam=6.0, acwt=0.083333)
N_HOPS = int(SECONDS * 16000) // ss.HOP


def utterances(lex, ids, spk, uniform):
    """(int16 waveforms [n x samples] (float64 numpy: the same on every machine), hop labels, transcripts) of the utterances
    `ids` by speakers `spk`"""
    hop, words = ss.transcripts(lex, ids, N_HOPS, uniform=uniform)
    return ss.waves_numpy(hop, ss.pdf_gains(P, DEPTH, dtype=np.float64), ids, spk_gain=ss.speaker_gains(spk, dtype=np.float64)), hop, words


def oracle_features(waves, spk):
    """the oracle's chain: MFCC, per-speaker CMVN, deltas, the speaker's (near-identity) transform"""
    o = vbo.mfcc_opts()
    raws = [vbo.mfcc(x.astype(np.float32), o) for x in waves]
    stats = {s: sum(vbo.acc_cmvn_stats(raws[u]) for u in range(len(raws)) if spk[u] == s) for s in np.unique(spk)}
    return [vbo.transform_feats(vbo.add_deltas(vbo.apply_cmvn(raws[u], stats[spk[u]])), synth.affine(D, seed=int(spk[u]), scale=XFORM_SCALE))
            for u in range(len(raws))]


def test_set(lex):
    ids = np.arange(N_UTT)
    spk = ids // 4
    waves, hop, truth = utterances(lex, ids, spk, False)
    return ids, spk, waves, hop, truth


def main():
    import torch
    t00 = time.time()
    g, t2p, lex = synth.lexicon_graph(WORDS, P, seed=11, return_lexicon=True)
    print("graph: %d states / %d arcs, %.0f s" % (len(g["final"]), len(g["arcs"]), time.time() - t00), flush=True)
    fit = ss.GmmFit(P, D, "cpu")
    for a in range(0, N_TRAIN, 16):
        ids = (1 << 24) + np.arange(a, min(a + 16, N_TRAIN))
        spk = (1 << 20) + (ids - (1 << 24)) // 4
        waves, hop, _ = utterances(lex, ids, spk, True)
        F = oracle_features(waves, spk)
        for u in range(len(ids)):
            fit.add(torch.from_numpy(F[u]), ss.frame_labels(hop[u:u + 1], F[u].shape[0]))
        print("training utterances %d.. done, %.0f s" % (a, time.time() - t00), flush=True)
    mu, var, unseen = fit.stats()
    offs, w, miv, iv = ss.mixture_from_stats(mu, var, G)
    ids, spk, waves, hop, truth = test_set(lex)
    F = oracle_features(waves, spk)
    gc = vbo.gmm_gconsts(w, miv, iv)
    fst = vbo.make_fst(g["final"], g["arc_offset"], g["arcs"], g["start"])
    hd = tempfile.mkdtemp(prefix="vb_sp_")
    env = dict(os.environ, OPENBLAS_NUM_THREADS="1", OMP_NUM_THREADS="1")
    out = dict(mu=mu, var=var, pdfs_unseen=unseen, n_utt=N_UTT, n_train=N_TRAIN, depth=DEPTH, words_in_graph=WORDS,
               cfg=np.array(["%s=%r" % kv for kv in sorted(CFG.items())]))
    try:
        ai = np.stack([g["arcs"]["ilabel"], g["arcs"]["olabel"], g["arcs"]["nextstate"]], 1).astype(np.int32)
        np.save(hd + "/fst_final.npy", g["final"]); np.save(hd + "/fst_off.npy", g["arc_offset"].astype(np.int64))
        np.save(hd + "/fst_arcs_i.npy", ai); np.save(hd + "/fst_arcs_w.npy", g["arcs"]["weight"].astype(np.float32))
        np.save(hd + "/tid2pdf.npy", t2p.astype(np.int32))
        with open(hd + "/opts.txt", "w") as f:
            for k, v in dict(CFG, start=0, prune_interval=25).items():
                f.write("%s=%s\n" % (k, v))
        rel = np.zeros((N_UTT, 10), np.int64)
        acc = np.zeros(N_UTT)
        refw, ourw, same = [], [], 0
        for u in range(N_UTT):
            ll = vbo.gmm_loglikes(offs, gc, miv, iv, F[u])
            acc[u] = float((ll.argmax(1) == ss.frame_labels(hop[u:u + 1], ll.shape[0])[0]).mean())
            np.save(hd + "/loglikes.npy", ll)
            subprocess.check_call([HARNESS, "decode", hd], env=env)
            L = lambda n: np.load(hd + "/out_lat_%s.npy" % n)
            ref = dict(tok_frame=L("tok_frame"), tok_state=L("tok_state"), link_src=L("link_src"), link_dst=L("link_dst"), link_ilabel=L("link_ilabel"),
                       link_olabel=L("link_olabel"), link_graph=L("link_graph"), link_acoustic=L("link_acoustic"))
            ours = vbo.decode_lattice_faster(fst, ll, t2p, exact_cutoff=True, **CFG)
            a, b = link_rows(ref), link_rows(ours)
            ta = set(zip(ref["tok_frame"].tolist(), ref["tok_state"].tolist()))
            tb = set(zip(ours["tok_frame"].tolist(), ours["tok_state"].tolist()))
            rel[u] = [len(a), len(b), len(a - b), len(b - a), len(ta), len(tb), len(ta - tb), len(tb - ta),
                      ours["n_frames_maxactive_bound"], ours["n_frames_minactive_bound"]]
            sb = (np.array_equal(ours["words"], L("words")) and np.array_equal(ours["ali"], L("ali")) and
                  np.array_equal(np.float32([ours["best_graph"], ours["best_acoustic"]]), L("best_weight")))
            same += int(sb)
            out["ref_words_%d" % u] = L("words").astype(np.int32)
            out["ref_ali_%d" % u] = L("ali").astype(np.int32)
            out["ref_best_weight_%d" % u] = L("best_weight").astype(np.float32)
            refw.append(L("words")); ourw.append(ours["words"])
            print("utt %d: frame accuracy %.3f; ref arcs %d, ours %d, ref-ours %d, ours-ref %d; tokens %d / %d (ref-ours %d, ours-ref %d); "
                  "maxbound %d minbound %d; best path identical %s  [%.0f s]" % ((u, acc[u]) + tuple(rel[u]) + (sb, time.time() - t00)), flush=True)
        out["lattice_relation"] = rel
        out["lattice_relation_cols"] = np.array(["ref_arcs", "our_arcs", "ref_minus_ours", "ours_minus_ref", "ref_toks", "our_toks", "ref_toks_minus_ours",
                                                 "our_toks_minus_ref", "frames_maxactive_bound", "frames_minactive_bound"])
        out["same_best_path_on_identical_scores"] = same
        out["frame_accuracy"] = acc
        cr, co = vdist.score(truth, refw), vdist.score(truth, ourw)
        out["waves_crc"] = np.array([int(np.bitwise_xor.reduce(waves[u].view(np.uint16).astype(np.uint64) * (np.arange(waves[u].size, dtype=np.uint64) % 65521 + 1))) for u in range(N_UTT)], np.uint64)
        out["wer_counters_reference"] = cr
        out["wer_counters_order_independent"] = co
        print("best paths identical on identical scores: %d / %d; WER against the transcripts: reference %.2f %%, order-independent %.2f %% (%d words)"
              % (same, N_UTT, 100.0 * cr[0] / cr[1], 100.0 * co[0] / co[1], cr[1]), flush=True)
    finally:
        shutil.rmtree(hd, ignore_errors=True)
    path = os.path.join(GOLD, "speech_refpaths.npz")
    np.savez_compressed(path, **out)
    print("%s  %.1f KB  (%.0f s)" % (path, os.path.getsize(path) / 1024.0, time.time() - t00))


if __name__ == "__main__":
    main()
