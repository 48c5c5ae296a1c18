#!/usr/bin/env python3
"""tests/golden/align.npz: the REFERENCE's gmm-align-compiled (kaldi-master/src/gmmbin/gmm-align-compiled.cc = the program
VoiceBridge's GmmAlignCompiled wraps, VoiceBridge/VoiceBridge/kaldi-win/src/gmmbin/gmm-align-compiled.cpp:24-160, over
AlignUtteranceWrapper, kaldi-master/src/decoder/decoder-wrappers.cc:424-516) on utterances of the speech workload, each against
its own compiled graph.  Run in the build container only:

    make -C oracle ref && python oracle/gen_golden_align.py

The case (align_case(), shared with tests/test_gpu_align.py, which regenerates everything from seeds):
  * N_UTT utterances of SECONDS s, words drawn from the lexicon, waveforms by synth_speech.waves_numpy, features through the
    oracle's chain (MFCC, per-speaker CMVN, deltas); the acoustic model of tests/golden/speech_refpaths.npz;
  * per utterance a training graph as compile-train-graphs would leave it before transition probabilities are added: the HMM
    states of the transcript in sequence, state k with a self-loop (transition-id 2 pdf + 1) and a forward arc (2 pdf + 2); every
    ALT_EVERY-th word has a second pronunciation branch (other pdfs, a graph cost on its first arc), so that the search has
    choices; the word id sits on the first arc of a word;
  * utterance WRONG's transcript has words too many at its end: with --beam=10 the search does not reach a final state
    and AlignUtteranceWrapper retries with --retry-beam=100; utterance TOO_LONG has more HMM states than frames: it fails both;
  * options of the reference's alignment steps: --transition-scale=1.0 --acoustic-scale=0.1 --self-loop-scale=0.1 --beam=10
    --retry-beam=100 (VB/scr/steps/align_si.cpp).
Stored: alignments, scores and per-frame acoustic log-likelihoods the reference wrote, which utterances it retried / failed.
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
from oracle import gen_golden_speech as gs  # noqa: E402
from voicebridge_amd import refio, synth_speech as ss  # noqa: E402

B = os.path.join(HERE, "_ref", "bin")
GOLD = os.path.join(ROOT, "tests", "golden")
N_UTT, SECONDS, ALT_EVERY, WRONG, TOO_LONG, N_EXTRA = 12, 6.0, 4, 7, 3, 1
P, G, D = gs.P, gs.G, gs.D
OPTS = ["--transition-scale=1.0", "--acoustic-scale=0.1", "--self-loop-scale=0.1", "--beam=10", "--retry-beam=100"]
ARC = np.dtype([("ilabel", "<i4"), ("olabel", "<i4"), ("weight", "<f4"), ("nextstate", "<i4")])


def utt_graph(lex, words, rng):
    """(final, arc_offset, arcs, start) of the training graph of one transcript"""
    off, wp = lex["word_off"], lex["word_pdf"]
    arcs = [[]]          # per state: (ilabel, olabel, weight, nextstate)

    def new_state():
        arcs.append([])
        return len(arcs) - 1

    def chain(first, pdfs, end, olabel):
        """HMM states pdfs[0..] on states first, (new ones); the last forward arc enters `end`"""
        s = first
        for k, pdf in enumerate(pdfs):
            nxt = end if k == len(pdfs) - 1 else new_state()
            arcs[s].append((2 * pdf + 1, 0, 0.0, s))                               # self-loop
            arcs[s].append((2 * pdf + 2, olabel if k == 0 else 0, 0.0, nxt))       # forward
            s = nxt
    cur = 0
    for n, w in enumerate(words):
        w0 = int(w) - 1
        pdfs = [int(x) for x in wp[off[w0]:off[w0 + 1]]]
        end = new_state()
        if n % ALT_EVERY == ALT_EVERY - 1:
            # two pronunciations behind input-epsilon arcs that carry the word label and the branch's graph cost
            alt = [int(x) for x in rng.integers(0, 3 * (P // 3), len(pdfs))]
            for pdfs_b, cost in ((pdfs, 0.0), (alt, 0.7)):
                b0 = new_state()
                arcs[cur].append((0, int(w), cost, b0))
                chain(b0, pdfs_b, end, 0)
        else:
            chain(cur, pdfs, end, int(w))
        cur = end
    S = len(arcs)
    final = np.full(S, np.inf, np.float32)
    final[cur] = 0.0
    offs = np.zeros(S + 1, np.int64)
    flat = []
    for s in range(S):
        flat += arcs[s]
        offs[s + 1] = len(flat)
    a = np.zeros(len(flat), ARC)
    for i, rec in enumerate(flat):
        a[i] = rec
    return final, offs, a, 0


def concat_graphs(gl):
    """the graphs one after the other: an epsilon arc from each one's final state to the next one's start"""
    fin, offs, arcs, base = [], [0], [], 0
    for i, (f, o, a, st) in enumerate(gl):
        a = a.copy()
        a["nextstate"] += base
        per = [list(a[o[s]:o[s + 1]]) for s in range(len(f))]
        if i + 1 < len(gl):
            last = int(np.flatnonzero(np.isfinite(f))[0])
            per[last].insert(0, np.array((0, 0, 0.0, base + len(f)), ARC))      # (the next graph starts at its state 0)
            f = np.full(len(f), np.inf, np.float32)
        for s in range(len(per)):
            arcs += per[s]
            offs.append(len(arcs))
        fin.append(f)
        base += len(f)
    return np.concatenate(fin), np.array(offs, np.int64), np.array(arcs, ARC), 0


def align_case(lex):
    """{key: (features [T x 39], graph)} in table order, and the model"""
    n_hops = int(SECONDS * 16000) // ss.HOP
    ids = (1 << 22) + np.arange(N_UTT)
    spk = (1 << 18) + np.arange(N_UTT) // 4
    hop, words = ss.transcripts(lex, ids, n_hops)
    waves = ss.waves_numpy(hop, ss.pdf_gains(P, gs.DEPTH, dtype=np.float64), ids, spk_gain=ss.speaker_gains(spk, dtype=np.float64))
    o = vbo.mfcc_opts()
    raws = [vbo.mfcc(x.astype(np.float32), o) for x in waves]
    stats = {s: sum(vbo.acc_cmvn_stats(raws[u]) for u in range(N_UTT) if spk[u] == s) for s in np.unique(spk)}
    feats = [vbo.add_deltas(vbo.apply_cmvn(raws[u], stats[spk[u]])) for u in range(N_UTT)]
    rng = np.random.default_rng(977)
    graphs = [utt_graph(lex, words[u], rng) for u in range(N_UTT)]
    # utterance WRONG's transcript has N_EXTRA words too many at its end (the final state is reached only in a hurry: outside
    # --beam=10 of the tokens that take their time, inside --retry-beam=100), utterance TOO_LONG more HMM states than it has frames (no beam helps)
    own = list(graphs)
    graphs[WRONG] = concat_graphs([own[WRONG], utt_graph(lex, words[8][:N_EXTRA], rng)])
    graphs[TOO_LONG] = concat_graphs(own)
    case = {}
    for u in range(N_UTT):
        case["utt%03d" % u] = (feats[u].astype(np.float32), graphs[u])
    fx = np.load(os.path.join(GOLD, "speech_refpaths.npz"))
    model = ss.mixture_from_stats(fx["mu"], fx["var"], G)
    return case, model


def write_case(d, case, model):
    offs, w, miv, iv = model
    refio.write_mdl(os.path.join(d, "final.mdl"), P // 3, 3, offs, w, miv, iv)
    refio.write_feats_ark(os.path.join(d, "feats.ark"), {k: v[0] for k, v in case.items()})
    refio.write_fst_table(os.path.join(d, "graphs.fsts"), {k: v[1] for k, v in case.items()})


def read_int_table(path):
    out = {}
    for line in open(path):
        f = line.split()
        if f:
            out[f[0]] = np.array([int(x) for x in f[1:]], np.int32)
    return out


def read_float_vec_table(path):
    out = {}
    for line in open(path):
        f = line.replace("[", " ").replace("]", " ").split()
        if f:
            out[f[0]] = np.array([float(x) for x in f[1:]], np.float32)
    return out


def main():
    from voicebridge_amd import synth
    t0 = time.time()
    g, t2p, lex = synth.lexicon_graph(gs.WORDS, P, seed=11, return_lexicon=True)
    case, model = align_case(lex)
    d = tempfile.mkdtemp(prefix="vb_align_")
    env = dict(os.environ, OPENBLAS_NUM_THREADS="1", OMP_NUM_THREADS="1")
    try:
        write_case(d, case, model)
        p = lambda n: os.path.join(d, n)
        r = subprocess.run([os.path.join(B, "gmm-align-compiled")] + OPTS + ["--write-per-frame-acoustic-loglikes=ark,t:" + p("pf"), p("final.mdl"),
                            "ark:" + p("graphs.fsts"), "ark:" + p("feats.ark"), "ark,t:" + p("ali"), "ark,t:" + p("scores")],
                           env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
        print(r.stderr[-1500:])
        assert r.returncode == 0
        ali, sc, pf = read_int_table(p("ali")), read_float_vec_table(p("scores")), read_float_vec_table(p("pf"))
        out = dict(keys=np.array(sorted(case)), opts=np.array(OPTS), retried=np.array(re.findall(r"Retrying utterance (\S+) with beam", r.stderr)),
                   failed=np.array(re.findall(r"Did not successfully decode file (\S+),", r.stderr)))
        for k in ali:
            out["ali_" + k] = ali[k]
            out["score_" + k] = np.float32(sc[k][0])
            out["per_frame_" + k] = pf[k]
        m = re.search(r"Overall log-likelihood per frame is (\S+) over (\d+) frames", r.stderr)
        out["overall"] = np.array([float(m.group(1)), float(m.group(2))])
        m = re.search(r"Done (\d+), errors on (\d+)", r.stderr)
        out["done_err"] = np.array([int(m.group(1)), int(m.group(2))])
        print("aligned %d, retried %s, failed %s  [%.0f s]" % (len(ali), list(out["retried"]), list(out["failed"]), time.time() - t0))
    finally:
        shutil.rmtree(d, ignore_errors=True)
    path = os.path.join(GOLD, "align.npz")
    np.savez_compressed(path, **out)
    print("%s  %.1f KB" % (path, os.path.getsize(path) / 1024.0))


if __name__ == "__main__":
    main()
