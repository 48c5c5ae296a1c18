#!/usr/bin/env python3
"""tests/golden/model_text.npz: the model of tests/golden/files.npz as the reference's gmm-copy --binary=false writes it (the text
form of TransitionModel + AmDiagGmm, K/hmm/hmm-topology.cc:41-127 for the topology's own grammar).  Build container only:

    make -C oracle ref && python oracle/gen_golden_text_model.py
"""
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
from tests import util  # noqa: E402


def main():
    g = util.load("files")
    d = tempfile.mkdtemp(prefix="vb_txtmdl_")
    open(d + "/final.mdl", "wb").write(g["file_final_mdl"].tobytes())
    subprocess.run([os.path.join(HERE, "_ref", "bin", "gmm-copy"), "--binary=false", d + "/final.mdl", d + "/final_text.mdl"], check=True, capture_output=True)
    txt = open(d + "/final_text.mdl", "rb").read()
    path = os.path.join(ROOT, "tests", "golden", "model_text.npz")
    np.savez_compressed(path, file_final_text_mdl=np.frombuffer(txt, np.uint8))
    print(path, os.path.getsize(path), "bytes;", len(txt), "bytes of text;", txt[:120])


if __name__ == "__main__":
    main()
