"""What does the search's admission rule do to the WORDS, measured on the 512 utterances of tests/golden/speech_e2e512.npz
(the reference's own programs from PCM; oracle/gen_golden_e2e512.py)?  The oracle's chain from PCM (pinned to the reference's
programs) with the search in each admission mode of oracle/vb_oracle.h (vbo_lat_cfg::exact_cutoff):

  0  the reference's order (HashList order, online tightening)          lattice-faster-decoder.cc:743-775 replayed
  1  order-independent: tot <= final next_cutoff                          rounds 1-3 of the GPU search
  2  bound = the reference's initial next_cutoff for every arc           superset of any order
  3  ordered admission in a canonical order: prefix minimum over buckets of the source state's hash (2^bits buckets)

For every mode: utterances whose words equal the programs', WER against the transcripts, the bootstrap probability that the mode
improves on the reference (compute-wer-bootci's two-system comparison, kaldi-master/src/bin/compute-wer-bootci.cc:137-155:
resampled utterances, strictly fewer errors), tokens created.  CPU only.

    python tools/admission_study.py [--utts 512] [--jobs 8] [--modes 0,1,2,3:6,3:10,3:16]

--decoder faster (round 5): the same study for the 1-best search, FasterDecoder (oracle modes 0, 1, 3 of vbo_decode_faster), against
tests/golden/speech_faster512.npz (the reference's gmm-decode-faster, min-active 20: oracle/gen_golden_faster512.py).
"""
import argparse
import os
import sys
import time
from concurrent.futures import ProcessPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def parse_modes(s):
    out = []
    for m in s.split(","):
        if ":" in m:
            a, b = m.split(":")
            out.append((int(a), int(b)))
        else:
            out.append((int(m), 6))
    return out


def work(args):
    spks, modes, decoder = args
    from oracle import vbo, gen_golden_e2e as e2e, gen_golden_speech as gs
    from voicebridge_amd import synth
    g, t2p, lex = synth.lexicon_graph(e2e.WORDS, e2e.P, seed=11, return_lexicon=True)
    offs, w, miv, iv = e2e.model()
    gc = vbo.gmm_gconsts(w, miv, iv)
    fst = vbo.make_fst(g["final"], g["arc_offset"], g["arcs"], g["start"])
    o = vbo.mfcc_opts()
    out = {}
    for s_ in spks:
        us = np.arange(4 * s_, 4 * s_ + 4)
        waves, _, truth = gs.utterances(lex, us, us // 4, False)
        raws = [vbo.mfcc(x.astype(np.float32), o) for x in waves]
        st = sum(vbo.acc_cmvn_stats(r) for r in raws)
        xf = synth.affine(e2e.D, seed=int(s_), scale=gs.XFORM_SCALE)
        for k, u in enumerate(us):
            f = vbo.transform_feats(vbo.add_deltas(vbo.apply_cmvn(raws[k], st)), xf)
            ll = vbo.gmm_loglikes(offs, gc, miv, iv, f)
            rec = dict(truth=np.asarray(truth[k], np.int32))
            for (m, bits) in modes:
                key = "%d_%d" % (m, bits)
                if decoder == "faster":
                    from oracle import gen_golden_faster512 as f512
                    C_ = f512.CFG
                    r = vbo.decode_faster(fst, ll, t2p, acwt=C_["acwt"], beam=C_["beam"], max_active=C_["max_active"], min_active=C_["min_active"],
                                          beam_delta=C_["beam_delta"], exact_cutoff=m, adm_bucket_bits=bits)
                    rec["words_" + key] = np.asarray(r["words"], np.int32)
                    rec["cost_" + key] = float(r["best_graph"] + r["best_acoustic"])
                    rec["toks_" + key] = 1; rec["links_" + key] = 0; rec["maxbound_" + key] = 0
                    continue
                r = vbo.decode_lattice_faster(fst, ll, t2p, exact_cutoff=m, adm_bucket_bits=bits, **e2e.CFG)
                rec["words_" + key] = np.asarray(r["words"], np.int32)
                rec["cost_" + key] = float(r["best_graph"] + r["best_acoustic"])
                rec["toks_" + key] = int(r["n_tok_created"])
                rec["links_" + key] = int(r["n_links_created"])
                rec["maxbound_" + key] = int(r["n_frames_maxactive_bound"])
            out[int(u)] = rec
        print("speaker %d done" % s_, flush=True)
    return out


def p_improves(err_ref, err_b, replications=10000, seed=0):
    """P(b improves on ref): bootstrap over utterances, strictly fewer errors (compute-wer-bootci.cc:137-155)"""
    rng = np.random.default_rng(seed)
    d = np.asarray(err_ref, np.int64) - np.asarray(err_b, np.int64)
    n = len(d)
    idx = rng.integers(0, n, (replications, n))
    return float((d[idx].sum(1) > 0).mean())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--utts", type=int, default=512)
    ap.add_argument("--jobs", type=int, default=8)
    ap.add_argument("--modes", default="0,1,2,3:6,3:10,3:16")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "admission_study.npz"))
    ap.add_argument("--analyse-only", action="store_true")
    ap.add_argument("--decoder", choices=("lattice", "faster"), default="lattice")
    a = ap.parse_args()
    modes = parse_modes(a.modes)
    t0 = time.time()
    from voicebridge_amd import dist as vdist
    fx = np.load(os.path.join(ROOT, "tests", "golden", "speech_faster512.npz" if a.decoder == "faster" else "speech_e2e512.npz"))
    if not a.analyse_only:
        n_spk = a.utts // 4
        chunks = [list(range(j, n_spk, a.jobs)) for j in range(a.jobs)]
        with ProcessPoolExecutor(a.jobs) as ex:
            parts = list(ex.map(work, [(c, modes, a.decoder) for c in chunks if c]))
        res = {}
        for p in parts:
            res.update(p)
        flat = {}
        for u, rec in res.items():
            for k, v in rec.items():
                flat["%s__%d" % (k, u)] = v
        os.makedirs(os.path.dirname(a.out), exist_ok=True)
        np.savez_compressed(a.out, utts=np.array(sorted(res)), modes=np.array(["%d_%d" % m for m in modes]), **flat)
        print("oracle runs: %.0f s" % (time.time() - t0), flush=True)
    z = np.load(a.out)
    utts = [int(u) for u in z["utts"]]
    keys = [str(k) for k in z["modes"]]
    truth = [z["truth__%d" % u] for u in utts]
    ref = [fx["ref_words_%d" % u] for u in utts]
    err = lambda hyp: np.array([sum(vdist.edit_counts(t, h)) for t, h in zip(truth, hyp)])
    e_ref = err(ref)
    nw = sum(len(t) for t in truth)
    print("%d utterances, %d words; the reference's programs: WER %.3f %%" % (len(utts), nw, 100.0 * e_ref.sum() / nw))
    for hr in (1, 3):
        same = fx["ref_same_words_hash_ratio_%d" % hr][utts]
        alt = [ref[i] if same[i] else fx["ref_words_hash_ratio_%d_%d" % (hr, u)] for i, u in enumerate(utts)]
        e = err(alt)
        print("  the reference with --hash-ratio=%d: words equal on %d / %d, WER %.3f %%, P(improves on default) %.2f, P(default improves on it) %.2f"
              % (hr, int(same.sum()), len(utts), 100.0 * e.sum() / nw, p_improves(e_ref, e), p_improves(e, e_ref)))
    base_toks = None
    for k in keys:
        hyp = [z["words_%s__%d" % (k, u)] for u in utts]
        same = sum(int(np.array_equal(h, r)) for h, r in zip(hyp, ref))
        e = err(hyp)
        toks = sum(int(z["toks_%s__%d" % (k, u)]) for u in utts)
        if base_toks is None:
            base_toks = toks
        worse = int((e > e_ref).sum()); better = int((e < e_ref).sum())
        print("  mode %-5s words equal to the programs' on %3d / %d; WER %.3f %% (%+d errors); utterances with more / fewer errors than the reference %d / %d; "
              "P(improves on reference) %.2f; tokens created %.4f x mode %s" % (k.replace("_", ":"), same, len(utts), 100.0 * e.sum() / nw, int(e.sum() - e_ref.sum()),
                                                                                worse, better, p_improves(e_ref, e), toks / base_toks, keys[0].replace("_", ":")))


if __name__ == "__main__":
    main()
