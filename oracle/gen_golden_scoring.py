#!/usr/bin/env python3
"""tests/golden/scoring_stats.json: what the reference's own align-text and compute-wer-bootci (oracle/_ref/bin, compiled from
K/bin/align-text.cc and K/bin/compute-wer-bootci.cc) make of a seeded set of transcripts and hypotheses -- the fixtures
tests/test_host_scoring.py pins voicebridge_amd/host/voicebridge_scoring.cc to.  Build container only:

    make -C oracle ref && python oracle/gen_golden_scoring.py
"""
import json
import os
import subprocess
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
BIN = os.path.join(HERE, "_ref", "bin")


def corpus(seed, n_utt, vocab=40, present=1.0):
    """(ref lines, hyp lines): hypotheses are the transcripts with random substitutions, deletions and insertions; utterances
    of 0..14 words; a fraction of the hypotheses missing when present < 1"""
    rng = np.random.default_rng(seed)
    ref, hyp = [], []
    for u in range(n_utt):
        n = int(rng.integers(0, 15))
        r = ["w%d" % x for x in rng.integers(0, vocab, n)]
        h = []
        for w in r:
            x = rng.random()
            if x < 0.10:
                h.append("w%d" % rng.integers(0, vocab))
            elif x < 0.17:
                pass
            else:
                h.append(w)
            if rng.random() < 0.06:
                h.append("w%d" % rng.integers(0, vocab))
        key = "spk%d-utt%03d" % (u % 5, u)
        ref.append(" ".join([key] + r))
        if rng.random() < present:
            hyp.append(" ".join([key] + h))
    return ref, hyp


def main():
    out = {}
    d = tempfile.mkdtemp(prefix="vb_sc_")
    for name, (seed, n, present, mode, sym) in dict(a=(1, 60, 1.0, "strict", "<eps>"), b=(2, 25, 0.8, "present", "'***'"), c=(3, 1, 1.0, "all", "<eps>")).items():
        ref, hyp = corpus(seed, n, present=present)
        open(d + "/ref", "w").write("\n".join(ref) + "\n")
        open(d + "/hyp", "w").write("\n".join(hyp) + "\n")
        subprocess.run([BIN + "/align-text", "--special-symbol=" + sym, "ark:" + d + "/ref", "ark:" + d + "/hyp", "ark,t:" + d + "/ali"], check=True, capture_output=True)
        r = subprocess.run([BIN + "/compute-wer-bootci", "--mode=" + mode, "ark:" + d + "/ref", "ark:" + d + "/hyp"], check=True, capture_output=True, text=True)
        out[name] = dict(ref=ref, hyp=hyp, special_symbol=sym, mode=mode, align_text=open(d + "/ali").read(), bootci_stdout=r.stdout)
        print(name, len(ref), "utterances;", r.stdout.strip())
    ref, hyp = corpus(4, 30)
    _, hyp2 = corpus(5, 30)
    hyp2 = [h.split()[0] + " " + " ".join(r.split()[1:][: max(0, len(r.split()) - 2)]) for r, h in zip(ref, hyp2)]
    open(d + "/ref", "w").write("\n".join(ref) + "\n"); open(d + "/hyp", "w").write("\n".join(hyp) + "\n"); open(d + "/hyp2", "w").write("\n".join(hyp2) + "\n")
    r = subprocess.run([BIN + "/compute-wer-bootci", "--replications=2000", "ark:" + d + "/ref", "ark:" + d + "/hyp", "ark:" + d + "/hyp2"], check=True, capture_output=True, text=True)
    out["dual"] = dict(ref=ref, hyp=hyp, hyp2=hyp2, replications=2000, bootci_stdout=r.stdout)
    print("dual;", r.stdout.strip().replace("\n", " | "))
    path = os.path.join(ROOT, "tests", "golden", "scoring_stats.json")
    json.dump(out, open(path, "w"), indent=0)
    print(path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
