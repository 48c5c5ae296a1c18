#!/usr/bin/env python3
"""tests/golden/mfcc_chirp_exact.npz: what the MFCCs of the noiseless chirp fixture (tests/golden/mfcc_chirp_c0.npz) are in exact
arithmetic, and how far the REFERENCE's own float32 output is from that -- the reference's noise floor on this input.

The chirp is the one input on which float32 implementations of the same MFCC formula disagree by ~0.1: most mel bins of a pure
tone hold no signal, only the FFT's rounding residue (power ~1e-7 of the peak), and log() turns the residue's relative differences
between two FFT algorithms into absolute cepstral differences.  A fixed tolerance cannot separate "wrong" from "differently
rounded" there; the distance of the reference itself from the exact value can.  The exact value is computed by the oracle's own
MFCC code (oracle/vb_oracle_feat.c, pinned to the reference on every other fixture at <= 1e-4) compiled with every float a double
(-Dfloat=double and the double forms of the math functions): same formula, ~1e-16 relative rounding instead of ~1e-7.

    python oracle/gen_golden_chirp_exact.py        (build container or anywhere: needs only gcc)
"""
import ctypes as C
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
from oracle import vbo  # noqa: E402


def exact_mfcc(wave, opts32):
    d = tempfile.mkdtemp(prefix="vbo_f64_")
    so = os.path.join(d, "libvbo_f64.so")
    subprocess.check_call(["gcc", "-O2", "-std=gnu99", "-fPIC", "-shared", "-Dfloat=double", "-Dlogf=log", "-Dexpf=exp", "-Dsqrtf=sqrt", "-Dcosf=cos", "-Dsinf=sin",
                           "-Dpowf=pow", "-Dfloorf=floor", "-Dfabsf=fabs", "-Dfmaxf=fmax", "-Dfminf=fmin", "-o", so, os.path.join(HERE, "vb_oracle_feat.c"), "-lm"])
    lib = C.CDLL(so)

    class Opts64(C.Structure):
        _fields_ = [(n, C.c_double if t is C.c_float else t) for n, t in vbo.MfccOpts._fields_]
    o = Opts64()
    for n, _t in vbo.MfccOpts._fields_:
        setattr(o, n, getattr(opts32, n))
    w = np.ascontiguousarray(wave, np.float64)
    lib.vbo_num_frames.argtypes = [C.c_int64, C.c_void_p]
    T = lib.vbo_num_frames(C.c_int64(len(w)), C.byref(o))
    out = np.zeros((T, o.num_ceps), np.float64)
    rc = lib.vbo_mfcc(C.byref(o), w.ctypes.data_as(C.c_void_p), C.c_int64(len(w)), C.c_double(1.0), out.ctypes.data_as(C.c_void_p), None, None, None)
    assert rc == 0, rc
    return out


def main():
    g = np.load(os.path.join(ROOT, "tests", "golden", "mfcc_chirp_c0.npz"))
    opts = vbo.mfcc_opts(use_energy=0)
    exact = exact_mfcc(g["wave"], opts)
    ref = g["feats"].astype(np.float64)
    orc = vbo.mfcc(g["wave"].astype(np.float32), opts).astype(np.float64)
    e_ref, e_orc = np.abs(ref - exact), np.abs(orc - exact)
    print("reference vs exact: max %.4f (per coefficient %s)" % (e_ref.max(), np.round(e_ref.max(0), 3)))
    print("oracle    vs exact: max %.4f; oracle vs reference: max %.4f" % (e_orc.max(), np.abs(orc - ref).max()))
    # sanity of the exact value: on a noisy input (where every bin holds signal) float32 and exact agree to ~1e-4
    s = np.load(os.path.join(ROOT, "tests", "golden", "mfcc_sine_nosnip_c0.npz"))
    from tests.util import mfcc_kwargs, parse_opts
    kw, _v = mfcc_kwargs(parse_opts(s["opts"]))
    o2 = vbo.mfcc_opts(**kw)
    d2 = np.abs(exact_mfcc(s["wave"], o2) - s["feats"]).max()
    print("control (sine + noise fixture): reference vs exact max %.2e" % d2)
    assert d2 < 1e-3
    path = os.path.join(ROOT, "tests", "golden", "mfcc_chirp_exact.npz")
    np.savez_compressed(path, exact=exact, reference_max_abs_err=e_ref.max(), reference_err_per_coeff=e_ref.max(0), control_max_abs_err=d2)
    print(path)


if __name__ == "__main__":
    main()
