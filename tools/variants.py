"""Experiment helper: build variants of one kernel source (vbh_decode.hip unless SRC is another of the library's sources) into voicebridge_amd/lib/variants/ and
(on the GPU box) time each with `bench.py --probe`.

  python tools/variants.py build NAME=SRC[:FLAG[:FLAG...]] ...   (here; SRC may live outside the repo)
  python tools/variants.py run [--batch B] NAME ...               (under gpurun)

The product library is untouched; a variant is selected with VBH_LIB_PATH (voicebridge_amd/_lib.py).
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from voicebridge_amd import build as B  # noqa: E402

VDIR = os.path.join(B.LIBDIR, "variants")


def build(specs):
    os.makedirs(VDIR, exist_ok=True)
    B.build(verbose=False)
    procs = []
    for spec in specs:
        name, rest = spec.split("=", 1)
        parts = rest.split(":")
        src, flags = parts[0], parts[1:]
        obj = os.path.join(VDIR, name + ".o")
        cmd = [os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")] + B.FLAGS + ["-ffp-contract=off"] + flags + ["-c", src, "-o", obj]
        print(" ".join(cmd), flush=True)
        procs.append((name, obj, subprocess.Popen(cmd)))
    for name, obj, p in procs:
        if p.wait() != 0:
            raise SystemExit("build failed: " + name)
        replaced = os.path.basename(src) if os.path.basename(src) in B.HIP_SOURCES else "vbh_decode.hip"      # (the product object the variant stands in for)
        objs = [os.path.join(B.LIBDIR, "obj", s.replace(".hip", ".o")) for s in B.HIP_SOURCES if s != replaced] + [obj, os.path.join(B.LIBDIR, "obj", "vbh_decode_l2.o")]
        lib = os.path.join(VDIR, "libvoicebridge_hip_%s.so" % name)
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs)
        os.remove(obj)
        print("built", lib)


def run(argv):
    extra = []
    while argv and argv[0].startswith("--"):
        extra += argv[:2]
        argv = argv[2:]
    for name in argv:
        env = dict(os.environ)
        env["VBH_LIB_PATH"] = os.path.join(VDIR, "libvoicebridge_hip_%s.so" % name)
        print("==== variant", name, " ".join(extra), flush=True)
        subprocess.call([sys.executable, os.path.join(ROOT, "bench.py"), "--probe"] + extra, env=env)


if __name__ == "__main__":
    if sys.argv[1] == "build":
        build(sys.argv[2:])
    else:
        run(sys.argv[2:])
