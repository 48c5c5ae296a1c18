"""Builds libvoicebridge_hip.so (hipcc, gfx950) in-tree: voicebridge_amd/lib/.

hipcc cross-compiles without a GPU; the resulting .so travels to the GPU box with the
repo snapshot.  `python -m voicebridge_amd.build` or __graft_entry__.build().
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libvoicebridge_hip.so")
HIP_SOURCES = ["vbh_ctx.hip", "vbh_mfcc.hip", "vbh_featpost.hip", "vbh_gmm.hip", "vbh_decode.hip", "vbh_faster.hip", "vbh_fmllr.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I" + os.path.join(ROOT, "include"), "-I" + CSRC]
LAYOUT2_FLAGS = ["-DVBH_LAYOUT2=1", "-DVBH_DECODE_NT=512", "-DVBH_WAVES_PER_EU=4", "-DVBH_HBITS=13"]


def _newer(src, dst):
    return (not os.path.exists(dst)) or os.path.getmtime(src) > os.path.getmtime(dst)


def build(force=False, verbose=True):
    os.makedirs(LIBDIR, exist_ok=True)
    objdir = os.path.join(LIBDIR, "obj")
    os.makedirs(objdir, exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    deps = [os.path.join(CSRC, "vbh_internal.h"), os.path.join(ROOT, "include", "voicebridge_hip.h")]
    objs, procs = [], []
    for s in HIP_SOURCES:
        src = os.path.join(CSRC, s)
        if not os.path.exists(src):
            continue
        obj = os.path.join(objdir, s.replace(".hip", ".o"))
        objs.append(obj)
        if force or _newer(src, obj) or any(_newer(d, obj) for d in deps):
            extra = []
            if s in ("vbh_decode.hip", "vbh_faster.hip"):
                extra = ["-ffp-contract=off"]   # exact float32 / float64 cost expressions
            if s == "vbh_faster.hip" and os.environ.get("VBH_FASTER_PROF"):
                extra.append("-DVBH_FASTER_PROF")
            if s == "vbh_decode.hip":
                if os.environ.get("VBH_DECODE_NT"):
                    extra.append("-DVBH_DECODE_NT=" + os.environ["VBH_DECODE_NT"])
            cmd = [hipcc] + FLAGS + extra + ["-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd), flush=True)
            procs.append((s, subprocess.Popen(cmd)))
    # the search kernel's second layout (two 512-thread workgroups per CU, for light searches: csrc/vbh_decode.hip "the two builds of this
    # file"): the same source again, device code + launcher only
    src = os.path.join(CSRC, "vbh_decode.hip")
    obj2 = os.path.join(objdir, "vbh_decode_l2.o")
    objs.append(obj2)
    if force or _newer(src, obj2) or any(_newer(d, obj2) for d in deps):
        cmd = [hipcc] + FLAGS + ["-ffp-contract=off"] + LAYOUT2_FLAGS + ["-c", src, "-o", obj2]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append(("vbh_decode.hip (layout 2)", subprocess.Popen(cmd)))
    failed = [s for s, p in procs if p.wait() != 0]
    if failed:
        raise RuntimeError("hipcc failed for: " + ", ".join(failed))
    if procs or not os.path.exists(LIB):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    build_host(force=force, verbose=verbose)
    return LIB


HOST_LIB = os.path.join(LIBDIR, "libvoicebridge_host.so")
HOST_SOURCES = ["kaldi_io.cc", "lattice.cc", "device_feats.cc", "voicebridge_api.cc", "voicebridge_latbin.cc", "voicebridge_fmllrbin.cc", "voicebridge_alignbin.cc", "voicebridge_scoring.cc", "fmllr.cc", "voicebridge_steps.cc", "host_capi.cc"]


def build_host(force=False, verbose=True):
    """C++ host layer (file formats + the reference's argv-style bin functions) over the C ABI."""
    hdir = os.path.join(HERE, "host")
    srcs = [os.path.join(hdir, s) for s in HOST_SOURCES]
    deps = srcs + [os.path.join(hdir, "kaldi_io.h"), os.path.join(hdir, "voicebridge_api.h"), os.path.join(hdir, "voicebridge_steps.h"),
                   os.path.join(hdir, "lattice.h"), os.path.join(hdir, "fmllr.h"), os.path.join(hdir, "api_common.h"), os.path.join(hdir, "device_feats.h"),
                   os.path.join(ROOT, "include", "voicebridge_hip.h"), LIB]
    if force or any(_newer(d, HOST_LIB) for d in deps):
        cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wall", "-pthread", "-Wl,--no-undefined", "-I" + os.path.join(ROOT, "include"), "-I" + hdir,
               "-o", HOST_LIB] + srcs + ["-L" + LIBDIR, "-lvoicebridge_hip", "-Wl,-rpath,$ORIGIN"]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return HOST_LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
