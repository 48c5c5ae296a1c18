#!/usr/bin/env python3
"""tests/golden/speech_faster512.npz: the REFERENCE's own gmm-decode-faster (FasterDecoder, the 1-best search) end to end, PCM to
words, on the 512 utterances of tests/golden/speech_e2e512.npz (an at-scale pin for the program north_star
names, in the regime where max-active binds and the admission rule of ProcessEmitting decides which arcs above the final
cutoff get through).  Run in the build container only:

    make -C oracle ref && python oracle/gen_golden_faster512.py

The chain, model, graph and waveforms are gen_golden_e2e512.py's (128 speakers x 4 x 30 s); the last program is

    gmm-decode-faster --beam=13 --max-active=7000 --acoustic-scale=0.083333 --allow-partial=true
      (kaldi-master/src/gmmbin/gmm-decode-faster.cc:131-166 over kaldi-master/src/decoder/faster-decoder.cc:224-346;
       min-active, beam-delta at FasterDecoderOptions' defaults 20 / 0.5, faster-decoder.h:38-43)

Stored per utterance: the words of the reference's best path and its "log-like per frame"; the words of the same program with
--hash-ratio=1 and 3 (another visiting order in ProcessEmitting, faster-decoder.cc:258-300: the reference's agreement with itself
is the yardstick for any other admission order); WER counters of all three against the transcripts; the bootstrap interval of the
reference's compute-wer-bootci.  Tests regenerate waveforms and transcripts from seeds.
"""
import os
import re
import shutil
import subprocess
import sys
import tempfile
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
from oracle import gen_golden_e2e as e2e  # noqa: E402
from oracle import gen_golden_speech as gs  # noqa: E402
from voicebridge_amd import dist as vdist  # noqa: E402
from voicebridge_amd import refio, synth  # noqa: E402

B = os.path.join(HERE, "_ref", "bin")
GOLD = os.path.join(ROOT, "tests", "golden")
N_UTT = int(os.environ.get("VB_FASTER512_UTTS", "512"))
N_JOBS = int(os.environ.get("VB_FASTER512_JOBS", "8"))
HASH_RATIOS = (1.0, 3.0)
CFG = dict(beam=13.0, max_active=7000, min_active=20, beam_delta=0.5, acwt=0.083333)


def main():
    t00 = time.time()
    g, t2p, lex = synth.lexicon_graph(e2e.WORDS, e2e.P, seed=11, return_lexicon=True)
    print("graph: %d states / %d arcs, %.0f s" % (len(g["final"]), len(g["arcs"]), time.time() - t00), flush=True)
    offs, w, miv, iv = e2e.model()
    ids = np.arange(N_UTT)
    spk = ids // 4
    spks = sorted(int(s) for s in np.unique(spk))
    xforms = {s: synth.affine(e2e.D, seed=s, scale=gs.XFORM_SCALE) for s in spks}
    d = tempfile.mkdtemp(prefix="vb_faster512_")
    env = dict(os.environ, OPENBLAS_NUM_THREADS="1", OMP_NUM_THREADS="1")
    p = lambda n: os.path.join(d, n)
    out = dict(n_utt=N_UTT, words_in_graph=e2e.WORDS, depth=gs.DEPTH, hash_ratios=np.array(HASH_RATIOS),
               cfg=np.array(["%s=%r" % kv for kv in sorted(CFG.items())]))
    truth = {}
    try:
        refio.write_mdl(p("final.mdl"), e2e.P // 3, 3, offs, w, miv, iv)
        refio.write_vector_fst(p("HCLG.fst"), g["final"], g["arc_offset"], g["arcs"], g["start"])
        per = -(-len(spks) // N_JOBS)
        jobs = [spks[j * per:(j + 1) * per] for j in range(N_JOBS) if spks[j * per:(j + 1) * per]]
        for j, js in enumerate(jobs):
            with open(p("wav.%d.scp" % j), "w") as f, open(p("utt2spk.%d" % j), "w") as fu, open(p("spk2utt.%d" % j), "w") as fs:
                for s_ in js:
                    us = np.array([int(u) for u in ids if spk[u] == s_])
                    waves, _, tr = gs.utterances(lex, us, spk[us], False)
                    for k, u in enumerate(us):
                        truth[int(u)] = tr[k]
                        refio.write_wav(p("u%05d.wav" % u), waves[k])
                        f.write("%s %s\n" % (e2e.utt_name(s_, u), p("u%05d.wav" % u)))
                        fu.write("%s s%04d\n" % (e2e.utt_name(s_, u), s_))
                    fs.write("s%04d %s\n" % (s_, " ".join(e2e.utt_name(s_, u) for u in us)))
            refio.write_feats_ark(p("trans.%d.ark" % j), {"s%04d" % s_: xforms[s_] for s_ in js})
            print("job %d: waveforms written  [%.0f s]" % (j, time.time() - t00), flush=True)

        def run(name, *a):
            r = subprocess.run([os.path.join(B, name)] + list(a), env=env, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True)
            if r.returncode != 0:
                raise RuntimeError(name + ": " + r.stderr[-400:])
            return r.stderr

        def read_words(path):
            w_ = {}
            for line in open(path):
                f_ = line.split()
                if f_:
                    w_[int(f_[0].split("-u")[1])] = np.array([int(x) for x in f_[1:]], np.int32)
            return w_

        def job(j):
            q = lambda n: p("%s.%d" % (n, j))
            run("compute-mfcc-feats", "--dither=0", "scp:" + p("wav.%d.scp" % j), "ark:" + q("mfcc"))
            run("compute-cmvn-stats", "--spk2utt=ark:" + p("spk2utt.%d" % j), "ark:" + q("mfcc"), "ark:" + q("cmvn"))
            run("apply-cmvn", "--utt2spk=ark:" + p("utt2spk.%d" % j), "ark:" + q("cmvn"), "ark:" + q("mfcc"), "ark:" + q("cm"))
            run("add-deltas", "ark:" + q("cm"), "ark:" + q("dl"))
            run("transform-feats", "--utt2spk=ark:" + p("utt2spk.%d" % j), "ark:" + p("trans.%d.ark" % j), "ark:" + q("dl"), "ark:" + q("feats"))
            common = ["--beam=%g" % CFG["beam"], "--max-active=%d" % CFG["max_active"], "--acoustic-scale=%g" % CFG["acwt"], "--allow-partial=true"]
            log = run("gmm-decode-faster", *common, p("final.mdl"), p("HCLG.fst"), "ark:" + q("feats"), "ark,t:" + q("words"))
            words, like = read_words(q("words")), {}
            for m_ in re.finditer(r"Log-like per frame for utterance \S+-u(\d+) is (\S+) over", log):
                like[int(m_.group(1))] = float(m_.group(2))
            print("job %d: default run done, %d utterances  [%.0f s]" % (j, len(words), time.time() - t00), flush=True)
            alt = {}
            for hr in HASH_RATIOS:
                run("gmm-decode-faster", *common, "--hash-ratio=%g" % hr, p("final.mdl"), p("HCLG.fst"), "ark:" + q("feats"), "ark,t:" + q("words_hr"))
                alt[hr] = read_words(q("words_hr"))
                print("job %d: --hash-ratio=%g done  [%.0f s]" % (j, hr, time.time() - t00), flush=True)
            return words, like, alt

        with ThreadPoolExecutor(len(jobs)) as ex:
            res = list(ex.map(job, range(len(jobs))))
        ref_words, ref_like = {}, {}
        alt_words = {hr: {} for hr in HASH_RATIOS}
        for wd, lk, alt in res:
            ref_words.update(wd); ref_like.update(lk)
            for hr in HASH_RATIOS:
                alt_words[hr].update(alt[hr])
        assert sorted(ref_words) == list(range(N_UTT)), len(ref_words)
        tr_list = [truth[u] for u in range(N_UTT)]
        for hr in HASH_RATIOS:
            same = [bool(np.array_equal(alt_words[hr][u], ref_words[u])) for u in range(N_UTT)]
            c_ = vdist.score(tr_list, [alt_words[hr][u] for u in range(N_UTT)])
            print("the reference with --hash-ratio=%g: words equal to its default run on %d / %d utterances, WER %.3f %%" % (hr, sum(same), N_UTT, 100.0 * c_[0] / c_[1]), flush=True)
            out["ref_same_words_hash_ratio_%g" % hr] = np.array(same)
            out["ref_wer_counters_hash_ratio_%g" % hr] = c_
            for u in range(N_UTT):
                if not same[u]:
                    out["ref_words_hash_ratio_%g_%d" % (hr, u)] = alt_words[hr][u]
        for u in range(N_UTT):
            out["ref_words_%d" % u] = ref_words[u]
        out["ref_like_per_frame"] = np.array([ref_like[u] for u in range(N_UTT)], np.float64)
        with open(p("ref.txt"), "w") as fr, open(p("hyp.txt"), "w") as fh:
            for u in range(N_UTT):
                fr.write("%s %s\n" % (e2e.utt_name(int(spk[u]), u), " ".join(str(int(x)) for x in truth[u])))
                fh.write("%s %s\n" % (e2e.utt_name(int(spk[u]), u), " ".join(str(int(x)) for x in ref_words[u])))
        r = subprocess.run([os.path.join(B, "compute-wer-bootci"), "--mode=present", "ark:" + p("ref.txt"), "ark:" + p("hyp.txt")], env=env,
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
        print(r.stdout.strip(), flush=True)
        m = re.search(r"%WER ([0-9.]+) 95% Conf Interval \[ ([0-9.]+), ([0-9.]+) \]", r.stdout)
        out["ref_wer_bootci"] = np.array([float(m.group(1)), float(m.group(2)), float(m.group(3))])
        cr = vdist.score(tr_list, [ref_words[u] for u in range(N_UTT)])
        out["wer_counters_reference"] = cr
        print("reference WER %.3f %% (%d errors / %d words)" % (100.0 * cr[0] / cr[1], cr[0], cr[1]), flush=True)
    finally:
        shutil.rmtree(d, ignore_errors=True)
    path = os.path.join(GOLD, "speech_faster512.npz" if N_UTT == 512 else "speech_faster%d_scratch.npz" % N_UTT)
    np.savez_compressed(path, **out)
    print("%s  %.1f KB  (%.0f s)" % (path, os.path.getsize(path) / 1024.0, time.time() - t00))


if __name__ == "__main__":
    main()
