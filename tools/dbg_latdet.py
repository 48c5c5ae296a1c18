"""debug: GPU GmmLatgenFaster (determinized and raw) against the reference program on one case"""
import ctypes as C, os, sys, subprocess, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import gen_golden_lat as gg
from tests import util
from voicebridge_amd import build
lib = C.CDLL(build.HOST_LIB)
def call(fn, args, log=""):
    argv = (C.c_char_p * len(args))(*[a.encode() for a in args]); return fn(len(args), argv, log.encode())
d = tempfile.mkdtemp()
beam = sys.argv[1] if len(sys.argv) > 1 else "20"
gg.make_case(d, seed=41, n_words=80, n_cd=12, frames=(150, 60, 333, 1))
p = lambda n: os.path.join(d, n)
for det in ("false", "true"):
    opts = ["--beam=" + beam, "--lattice-beam=6", "--max-active=7000", "--acoustic-scale=0.083333", "--allow-partial=true", "--determinize-lattice=" + det]
    files = [p("final.mdl"), p("HCLG.fst"), "ark:" + p("feats.ark")]
    assert call(lib.vb_GmmLatgenFaster, ["x"] + opts + files + ["ark:" + p("m_%s.ark" % det)], p("log")) == 0, open(p("log")).read()
    subprocess.run([os.path.join(gg.B, "gmm-latgen-faster")] + opts + files + ["ark:" + p("r_%s.ark" % det)], stderr=subprocess.DEVNULL, check=True)
    A, B = util.read_lattice_ark(p("m_%s.ark" % det)), util.read_lattice_ark(p("r_%s.ark" % det))
    for k in B:
        a, b = A[k][2], B[k][2]
        if det == "false":
            # compare as sets of arcs keyed by labels+weights (state numbering differs)
            sa = sorted((x[0], x[1], x[2], x[3]) for st in a for x in st[1]); sb = sorted((x[0], x[1], x[2], x[3]) for st in b for x in st[1])
            print(det, k, "states", len(a), len(b), "arcs", len(sa), len(sb), "same multiset:", sa == sb)
            if sa != sb:
                import collections
                ca, cb = collections.Counter(sa), collections.Counter(sb)
                print("   only mine:", list((ca - cb).items())[:4]); print("   only ref:", list((cb - ca).items())[:4])
        else:
            print(det, k, "states", len(a), len(b), "equal:", a == b)
            if a != b:
                for s, (x, y) in enumerate(zip(a, b)):
                    if x != y:
                        print("  state", s, "final", x[0][:2], y[0][:2], "narcs", len(x[1]), len(y[1]))
                        for (u, v) in zip(x[1], y[1]):
                            if u != v:
                                print("    mine", u[0], u[1], u[2], len(u[3]), u[4]); print("    ref ", v[0], v[1], v[2], len(v[3]), v[4]); break
                        break
# and: my determinizer on the reference's raw archive vs the reference's determinized output
assert call(lib.vb_LatticeDeterminizePhonePruned, ["x", "--acoustic-scale=0.083333", "--beam=6", p("final.mdl"), "ark:" + p("r_false.ark"), "ark:" + p("m_from_r.ark")], p("log")) == 0
subprocess.run([os.path.join(gg.B, "lattice-determinize-phone-pruned"), "--acoustic-scale=0.083333", "--beam=6", p("final.mdl"), "ark:" + p("r_false.ark"), "ark:" + p("r_from_r.ark")], stderr=subprocess.DEVNULL, check=True)
A, B, R = util.read_lattice_ark(p("m_from_r.ark")), util.read_lattice_ark(p("r_from_r.ark")), util.read_lattice_ark(p("r_true.ark"))
print("from the reference's raw archive: mine == ref program:", A == B, "; ref program == ref in-decoder:", {k: B[k][2] == R[k][2] for k in B})
