#!/usr/bin/env python3
"""tests/golden/config3_refpaths.npz: what the REFERENCE decodes on utterances of the bench's config-3 workload
(30 s utterances = 2998 frames, P = 2500 / G = 15000 / D = 39, the 180 k-word lexicon graph of
bench.py (~2 M states), beam 13, lattice-beam 6, max-active 7000, acoustic scale 1/12), and how the order-independent
admission rule of the GPU search relates to the reference's lattices at that scale.  Run in the build container only:

    make -C oracle ref && python oracle/gen_golden_config3.py

Two things are produced.
(1) Best paths from the reference's own programs (oracle/_ref/bin: compute-mfcc-feats --dither=0 | compute-cmvn-stats |
    apply-cmvn | add-deltas | transform-feats | gmm-latgen-faster): words, alignment and (graph, acoustic) weight per
    utterance.  The GPU box has no reference binary; tests/test_gpu_config3.py runs the GPU path on the same seeded inputs
    and compares.
(2) Lattice relation at this scale, on IDENTICAL scores (the oracle's feature chain and all-pdf log-likelihoods, fed to
    the reference's LatticeFasterDecoder through oracle/_ref/vb_ref_harness and to the oracle's exact_cutoff=1 mode,
    which is what the GPU search computes bit for bit): arcs/tokens of the reference's raw lattice that the
    order-independent lattice lacks and vice versa, and whether the two DETERMINIZED lattices
    (lattice-determinize-phone-pruned semantics, host code of this build, itself bit-exact with the reference's program)
    are identical.  The numbers are stored in the fixture and quoted in DESIGN.md 5.
Inputs are regenerated from seeds by the test (voicebridge_amd/synth.py); the fixture holds the model's placement vectors
(mean/std of the features, which position the random GMM in feature space) and the outputs only.
"""
import ctypes as C
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
from voicebridge_amd import refio, synth  # noqa: E402

BIN = os.path.join(HERE, "_ref", "bin")
HARNESS = os.path.join(HERE, "_ref", "vb_ref_harness")
GOLD = os.path.join(ROOT, "tests", "golden")

N_UTT, N_SPK, SECONDS = 8, 2, 30.0
P, G, D, WORDS = 2500, 15000, 39, 180000
CFG = dict(beam=13.0, max_active=7000, min_active=200, lattice_beam=6.0, acwt=0.083333)


def workload():
    """the seeded inputs both sides rebuild (the test imports this module's constants and this function)"""
    waves = [synth.wave(i, SECONDS) for i in range(N_UTT)]
    utt2spk = np.arange(N_UTT, dtype=np.int32) % N_SPK
    xforms = np.stack([synth.affine(D, seed=s) for s in range(N_SPK)])
    return waves, utt2spk, xforms


def model_from(mu, std):
    offs, w, means, iv = synth.gmm_model(P, G, D, seed=1, mean_scale=1.5, dim_scale=std, var_spread=0.2)
    means = means + mu.astype(np.float32)
    return offs, w, (means * iv).astype(np.float32), iv


def link_rows(r):
    tf, ts = r["tok_frame"], r["tok_state"]
    return set(zip(tf[r["link_src"]].tolist(), ts[r["link_src"]].tolist(), tf[r["link_dst"]].tolist(), ts[r["link_dst"]].tolist(),
                   r["link_ilabel"].tolist(), r["link_olabel"].tolist(), r["link_graph"].view(np.int32).tolist(),
                   r["link_acoustic"].view(np.int32).tolist()))


def main():
    t00 = time.time()
    waves, utt2spk, xforms = workload()
    o = vbo.mfcc_opts()
    raws = [vbo.mfcc(w.astype(np.float32), o) for w in waves]
    stats = [sum(vbo.acc_cmvn_stats(raws[u]) for u in range(N_UTT) if utt2spk[u] == s) for s in range(N_SPK)]
    feats = [vbo.transform_feats(vbo.add_deltas(vbo.apply_cmvn(raws[u], stats[utt2spk[u]])), xforms[utt2spk[u]]) for u in range(N_UTT)]
    mu, std = feats[0][:20000].mean(0), feats[0][:20000].std(0) + 1e-3
    offs, w, miv, iv = model_from(mu, std)
    g, t2p = synth.lexicon_graph(WORDS, P, seed=11)
    print("inputs ready: %d states / %d arcs, %.0f s" % (len(g["final"]), len(g["arcs"]), time.time() - t00), flush=True)
    d = tempfile.mkdtemp(prefix="vb_c3_")
    env = dict(os.environ, OPENBLAS_NUM_THREADS="1", OMP_NUM_THREADS="1")
    p = lambda n: os.path.join(d, n)
    out = dict(mu=mu.astype(np.float32), std=std.astype(np.float32), n_utt=N_UTT, n_spk=N_SPK, seconds=SECONDS, words_in_graph=WORDS,
               cfg=np.array(["%s=%r" % kv for kv in sorted(CFG.items())]))
    try:
        # ---- (1) the reference's programs, end to end
        refio.write_mdl(p("final.mdl"), P // 3, 3, offs, w, miv, iv)
        refio.write_vector_fst(p("HCLG.fst"), g["final"], g["arc_offset"], g["arcs"], g["start"])
        with open(p("wav.scp"), "w") as f, open(p("utt2spk"), "w") as fu, open(p("spk2utt"), "w") as fs:
            for u in range(N_UTT):
                refio.write_wav(p("u%d.wav" % u), waves[u])
            order = sorted(range(N_UTT), key=lambda u: "s%d-u%d" % (utt2spk[u], u))
            for u in order:
                f.write("s%d-u%d %s\n" % (utt2spk[u], u, p("u%d.wav" % u)))
                fu.write("s%d-u%d s%d\n" % (utt2spk[u], u, utt2spk[u]))
            for s in range(N_SPK):
                fs.write("s%d %s\n" % (s, " ".join("s%d-u%d" % (s, u) for u in range(N_UTT) if utt2spk[u] == s)))
        refio.write_feats_ark(p("trans.ark"), {"s%d" % s: xforms[s] for s in range(N_SPK)})

        def run(name, *a):
            r = subprocess.run([os.path.join(BIN, name)] + list(a), env=env, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True)
            if r.returncode != 0:
                raise RuntimeError(name + ": " + r.stderr[-500:])
            return r.stderr
        run("compute-mfcc-feats", "--dither=0", "scp:" + p("wav.scp"), "ark:" + p("mfcc.ark"))
        run("compute-cmvn-stats", "--spk2utt=ark:" + p("spk2utt"), "ark:" + p("mfcc.ark"), "ark:" + p("cmvn.ark"))
        run("apply-cmvn", "--utt2spk=ark:" + p("utt2spk"), "ark:" + p("cmvn.ark"), "ark:" + p("mfcc.ark"), "ark:" + p("cm.ark"))
        run("add-deltas", "ark:" + p("cm.ark"), "ark:" + p("dl.ark"))
        run("transform-feats", "--utt2spk=ark:" + p("utt2spk"), "ark:" + p("trans.ark"), "ark:" + p("dl.ark"), "ark:" + p("feats.ark"))
        log = run("gmm-latgen-faster", "--beam=%g" % CFG["beam"], "--lattice-beam=%g" % CFG["lattice_beam"], "--max-active=%d" % CFG["max_active"],
                  "--min-active=%d" % CFG["min_active"], "--acoustic-scale=%g" % CFG["acwt"], "--allow-partial=true", "--determinize-lattice=false",
                  p("final.mdl"), p("HCLG.fst"), "ark:" + p("feats.ark"), "ark:" + p("lat.ark"), "ark,t:" + p("words.txt"), "ark,t:" + p("ali.txt"))
        for name, fn in (("words", "words.txt"), ("ali", "ali.txt")):
            for line in open(p(fn)):
                f_ = line.split()
                out["ref_%s_%d" % (name, int(f_[0].split("-u")[1]))] = np.array(f_[1:], np.int32)
        # "LOG ... Log-like per frame for utterance s0-u0 is -7.05 over 2998 frames." -> per-frame average of (graph + acoustic) / acwt... keep the log line values
        for m in re.finditer(r"Log-like per frame for utterance s\d+-u(\d+) is (\S+) over (\d+) frames", log):
            out["ref_like_per_frame_%d" % int(m.group(1))] = np.float32(m.group(2))
        print("reference programs done, %.0f s" % (time.time() - t00), flush=True)

        # ---- (2) lattice relation on identical scores
        gc = vbo.gmm_gconsts(w, miv, iv)
        fst = vbo.make_fst(g["final"], g["arc_offset"], g["arcs"], g["start"])
        hd = p("h")
        os.makedirs(hd)
        ai = np.stack([g["arcs"]["ilabel"], g["arcs"]["olabel"], g["arcs"]["nextstate"]], 1).astype(np.int32)
        np.save(os.path.join(hd, "fst_final.npy"), g["final"]); np.save(os.path.join(hd, "fst_off.npy"), g["arc_offset"].astype(np.int64))
        np.save(os.path.join(hd, "fst_arcs_i.npy"), ai); np.save(os.path.join(hd, "fst_arcs_w.npy"), g["arcs"]["weight"].astype(np.float32))
        np.save(os.path.join(hd, "tid2pdf.npy"), t2p.astype(np.int32))
        with open(os.path.join(hd, "opts.txt"), "w") as f:
            for k, v in dict(CFG, start=0, prune_interval=25).items():
                f.write("%s=%s\n" % (k, v))
        ref_lats, our_lats = {}, {}
        rel = np.zeros((N_UTT, 10), np.int64)
        same_best = 0
        for u in range(N_UTT):
            ll = vbo.gmm_loglikes(offs, gc, miv, iv, feats[u])
            np.save(os.path.join(hd, "loglikes.npy"), ll)
            subprocess.check_call([HARNESS, "decode", hd], env=env)
            L = lambda n: np.load(os.path.join(hd, "out_lat_%s.npy" % n))
            ref = dict(tok_frame=L("tok_frame"), tok_state=L("tok_state"), tok_cost=L("tok_cost"), tok_extra=L("tok_extra"),
                       link_src=L("link_src"), link_dst=L("link_dst"), link_ilabel=L("link_ilabel"), link_olabel=L("link_olabel"),
                       link_graph=L("link_graph"), link_acoustic=L("link_acoustic"))
            T = int(L("meta")[2])
            fin = np.full(len(ref["tok_frame"]), np.inf, np.float32)
            last = ref["tok_frame"] == T
            fc = g["final"][ref["tok_state"][last]]
            fin[last] = fc if np.isfinite(fc).any() else 0.0
            ref["tok_final"] = fin
            ours = vbo.decode_lattice_faster(fst, ll, t2p, exact_cutoff=True, **{k: v for k, v in CFG.items()})
            a, b = link_rows(ref), link_rows(ours)
            ta = set(zip(ref["tok_frame"].tolist(), ref["tok_state"].tolist()))
            tb = set(zip(ours["tok_frame"].tolist(), ours["tok_state"].tolist()))
            rel[u] = [len(a), len(b), len(a - b), len(b - a), len(ta), len(tb), len(ta - tb), len(tb - ta),
                      ours["n_frames_maxactive_bound"], ours["n_frames_minactive_bound"]]
            sb = (np.array_equal(ours["words"], L("words")) and np.array_equal(ours["ali"], L("ali")) and
                  np.array_equal(np.float32([ours["best_graph"], ours["best_acoustic"]]), L("best_weight")))
            same_best += int(sb)
            ref_lats["u%d" % u] = ref
            our_lats["u%d" % u] = ours
            print("utt %d: ref arcs %d, ours %d, ref-ours %d, ours-ref %d; tokens %d / %d (ref-ours %d, ours-ref %d); maxbound %d minbound %d; best path equal %s  [%.0f s]"
                  % ((u,) + tuple(rel[u]) + (sb, time.time() - t00)), flush=True)
        out["lattice_relation"] = rel
        out["lattice_relation_cols"] = np.array(["ref_arcs", "our_arcs", "ref_minus_ours", "ours_minus_ref", "ref_toks", "our_toks", "ref_toks_minus_ours",
                                                 "our_toks_minus_ref", "frames_maxactive_bound", "frames_minactive_bound"])
        out["same_best_path_on_identical_scores"] = same_best
        # determinized (phone-pruned + word-level, beam = lattice beam): the host code of this build
        from voicebridge_amd import build
        host = C.CDLL(build.HOST_LIB)

        def call(fn, args, log=""):
            argv = (C.c_char_p * len(args))(*[a.encode() for a in args])
            return fn(len(args), argv, str(log).encode())
        refio.write_raw_lattice_ark(p("ref_raw.ark"), ref_lats, g["start"])
        refio.write_raw_lattice_ark(p("our_raw.ark"), our_lats, g["start"])
        for n in ("ref", "our"):
            assert call(host.vb_LatticeDeterminizePhonePruned, ["lattice-determinize-phone-pruned", "--acoustic-scale=%g" % CFG["acwt"],
                                                                "--beam=%g" % CFG["lattice_beam"], p("final.mdl"), "ark:" + p(n + "_raw.ark"),
                                                                "ark:" + p(n + "_det.ark")], p("det.log")) == 0, open(p("det.log")).read()
        # the reference's own program on the reference's raw lattices must agree with the host code (bit-exact claim of DESIGN.md 1)
        run("lattice-determinize-phone-pruned", "--acoustic-scale=%g" % CFG["acwt"], "--beam=%g" % CFG["lattice_beam"], p("final.mdl"),
            "ark:" + p("ref_raw.ark"), "ark:" + p("ref_det_refprog.ark"))
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from tests import util
        A, B, R = util.read_lattice_ark(p("ref_det.ark")), util.read_lattice_ark(p("our_det.ark")), util.read_lattice_ark(p("ref_det_refprog.ark"))
        assert A == R, "host determinizer differs from the reference's program on the reference's raw lattices"
        det_equal = np.array([int(A[k] == B[k]) for k in sorted(A)], np.int32)
        det_sizes = np.array([[len(A[k][2]), sum(len(s[1]) for s in A[k][2]), len(B[k][2]), sum(len(s[1]) for s in B[k][2])] for k in sorted(A)], np.int32)
        # what differs where they are not identical: word labels / structure, transition-id strings, or weights only
        det_diff = []
        for k in sorted(A):
            if A[k] == B[k]:
                det_diff.append("identical")
                continue
            sa, sb = A[k][2], B[k][2]
            if len(sa) != len(sb) or any(len(x[1]) != len(y[1]) for x, y in zip(sa, sb)):
                det_diff.append("structure")
                continue
            labels = all([(a[0], a[4]) for a in x[1]] == [(a[0], a[4]) for a in y[1]] for x, y in zip(sa, sb))
            tids = all([a[3] for a in x[1]] == [a[3] for a in y[1]] and x[0][2] == y[0][2] for x, y in zip(sa, sb))
            wmax = max([abs(a[1] - b[1]) + abs(a[2] - b[2]) for x, y in zip(sa, sb) for a, b in zip(x[1], y[1])] +
                       [abs(x[0][0] - y[0][0]) + abs(x[0][1] - y[0][1]) for x, y in zip(sa, sb) if np.isfinite(x[0][0]) and np.isfinite(y[0][0])] + [0.0])
            det_diff.append("same words and next states: %s; same transition-id strings: %s; max weight difference %.3g" % (labels, tids, wmax))
        out["determinized_difference"] = np.array(det_diff)
        print("\n".join("  %s: %s" % kv for kv in zip(sorted(A), det_diff)), flush=True)
        out["determinized_equal"] = det_equal
        out["determinized_sizes"] = det_sizes      # ref states, ref arcs, our states, our arcs
        print("determinized lattices identical: %d / %d; sizes %s" % (det_equal.sum(), len(det_equal), det_sizes.tolist()), flush=True)
    finally:
        shutil.rmtree(d, ignore_errors=True)
    path = os.path.join(GOLD, "config3_refpaths.npz")
    np.savez_compressed(path, **out)
    print("%s  %.1f KB  (%.0f s)" % (path, os.path.getsize(path) / 1024.0, time.time() - t00))


if __name__ == "__main__":
    main()
