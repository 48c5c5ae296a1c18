#!/usr/bin/env python3
"""tests/golden/lattices_min.npz: the reference's own lattice-determinize-pruned --minimize=true and
lattice-determinize-phone-pruned --minimize=true (oracle/_ref/bin) on raw lattices of seeded synthetic decodes (the case of
tests/golden/lattices.npz and two wider ones: more words, wider lattice beam, so that states do merge).  Stored: model, raw
lattices, the two minimized archives per case, and the state counts with and without minimization.  Build container only:

    make -C oracle ref && python oracle/gen_golden_lat_min.py
"""
import os
import shutil
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
from oracle.gen_golden_lat import make_case, ref  # noqa: E402
from tests import util  # noqa: E402

CASES = dict(a=dict(seed=5, n_words=40, n_cd=12, frames=(120, 45, 260), beam="6"),
             b=dict(seed=21, n_words=150, n_cd=20, frames=(200, 90), beam="9"),
             c=dict(seed=33, n_words=60, n_cd=10, frames=(150, 30, 30), beam="12"))
ACWT = "0.083333"


def n_states(path):
    return sum(len(v[2]) for v in util.read_lattice_ark(path).values())


def main():
    out = {}
    for name, c in CASES.items():
        d = tempfile.mkdtemp(prefix="vb_latmin_")
        try:
            p = lambda n: os.path.join(d, n)
            make_case(d, seed=c["seed"], n_words=c["n_words"], n_cd=c["n_cd"], frames=c["frames"])
            ref("gmm-latgen-faster", "--beam=13", "--lattice-beam=" + c["beam"], "--acoustic-scale=" + ACWT, "--allow-partial=true",
                "--determinize-lattice=false", p("final.mdl"), p("HCLG.fst"), "ark:" + p("feats.ark"), "ark:" + p("raw.ark"))
            for prog, extra, tag in (("lattice-determinize-pruned", [], "p"), ("lattice-determinize-phone-pruned", [p("final.mdl")], "pp")):
                for mn in ("false", "true"):
                    ref(prog, "--acoustic-scale=" + ACWT, "--beam=" + c["beam"], "--minimize=" + mn, *extra, "ark:" + p("raw.ark"), "ark:" + p("det_%s_%s.ark" % (tag, mn)))
                print(name, prog, "states without / with --minimize:", n_states(p("det_%s_false.ark" % tag)), n_states(p("det_%s_true.ark" % tag)))
                out["%s_det_%s_min" % (name, tag)] = np.frombuffer(open(p("det_%s_true.ark" % tag), "rb").read(), np.uint8)
                out["%s_states_%s" % (name, tag)] = np.array([n_states(p("det_%s_false.ark" % tag)), n_states(p("det_%s_true.ark" % tag))])
            out[name + "_raw"] = np.frombuffer(open(p("raw.ark"), "rb").read(), np.uint8)
            out[name + "_mdl"] = np.frombuffer(open(p("final.mdl"), "rb").read(), np.uint8)
            out[name + "_beam"] = np.array(c["beam"])
        finally:
            shutil.rmtree(d, ignore_errors=True)
    out["acwt"] = np.array(ACWT)
    path = os.path.join(ROOT, "tests", "golden", "lattices_min.npz")
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path) // 1024, "KB")


if __name__ == "__main__":
    main()
