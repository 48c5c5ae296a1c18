#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by running the REFERENCE's own C++
(oracle/_ref/vb_ref_harness, built by `make -C oracle ref` from /root/reference) on seeded
synthetic inputs.  Run in the build container only:

    make -C oracle ref && python oracle/gen_golden.py

Each fixture is an .npz holding the inputs and the reference's outputs (data only).  The
reference's HTK feature fixtures (kaldi-master/src/feat/test_data/test.wav and
test.wav.fea_htk.1-6, data files of the reference's own feature-mfcc-test) are copied as
they are into tests/golden/htk/.
"""
import os
import shutil
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
from voicebridge_amd import synth  # noqa: E402

HARNESS = os.path.join(HERE, "_ref", "vb_ref_harness")
GOLD = os.path.join(ROOT, "tests", "golden")
REF_TEST_DATA = "/root/reference/kaldi-master/src/feat/test_data"


def run(cmd, inputs, opts):
    """Write inputs/opts into a temp dir, run the harness, return dict of outputs."""
    d = tempfile.mkdtemp(prefix="vbgold_")
    try:
        for k, v in inputs.items():
            np.save(os.path.join(d, k + ".npy"), v)
        with open(os.path.join(d, "opts.txt"), "w") as f:
            for k, v in opts.items():
                f.write("%s=%s\n" % (k, v))
        subprocess.check_call([HARNESS, cmd, d], env=dict(os.environ, OPENBLAS_NUM_THREADS="1"))
        out = {}
        for fn in sorted(os.listdir(d)):
            if fn.startswith("out_"):
                p = os.path.join(d, fn)
                if fn.endswith(".npy"):
                    out[fn[4:-4]] = np.load(p)
                else:
                    out["file_" + fn[4:].replace(".", "_")] = np.fromfile(p, dtype=np.uint8)
        return out
    finally:
        shutil.rmtree(d)


def save(name, **arrays):
    path = os.path.join(GOLD, name + ".npz")
    np.savez_compressed(path, **arrays)
    print("%-28s %7.1f KB" % (name + ".npz", os.path.getsize(path) / 1024.0))


def opts_str(o):
    return np.array(["%s=%s" % kv for kv in sorted(o.items())])


def gen_mfcc():
    rng = np.random.default_rng(11)
    n = 8000
    t = np.arange(n) / 16000.0
    sine = np.round((0.3 * np.sin(2 * np.pi * 440 * t) + 0.1 * rng.standard_normal(n)) * 8000)
    chirp = synth.chirp(0.5).astype(np.float64)
    dc = np.full(n, 1234.0)
    dc[:2000] = 0.0  # silence then DC: exercises the log floor and the DC removal
    speechy = synth.wave(3, seconds=0.6).astype(np.float64)
    cases = {
        "mfcc_sine_default": (sine, dict(), True),
        "mfcc_chirp_c0": (chirp, dict(use_energy=0), False),
        "mfcc_dc_nosnip": (dc, dict(snip_edges=0), False),
        "mfcc_sine_nosnip_c0": (sine, dict(snip_edges=0, use_energy=0), False),
        "mfcc_speechy_htk": (speechy, dict(htk_compat=1, raw_energy=0, window_type="hamming",
                                           energy_floor=1.0e+9, preemph_coeff=0.9), False),
        "mfcc_speechy_vtln": (speechy, dict(vtln_warp=1.1, use_energy=0, num_bins=26, num_ceps=15,
                                            low_freq=60, high_freq=-300, cepstral_lifter=0), False),
        "mfcc_short_hann": (sine[:999], dict(window_type="hanning", remove_dc_offset=0,
                                             frame_length_ms=20, frame_shift_ms=5), False),
        "mfcc_tooshort": (sine[:300], dict(), False),
    }
    for name, (w, o, stages) in cases.items():
        o = dict(o, dither=0)
        out = run("mfcc", dict(wave=w.astype(np.float32)), o)
        keep = dict(wave=w.astype(np.float32), opts=opts_str(o), feats=out["feats"], numframes=out["numframes"])
        if stages:
            keep.update(win=out["win"], pspec=out["pspec"], mel=out["mel"])
        save(name, **keep)


def gen_post():
    rng = np.random.default_rng(5)
    feats = (rng.standard_normal((60, 13)) * np.linspace(8, 1, 13) + np.linspace(-3, 3, 13)).astype(np.float32)
    spk = (rng.standard_normal((500, 13)) * np.linspace(8, 1, 13) + np.linspace(-3, 3, 13)).astype(np.float32)
    stats = np.zeros((2, 14))
    stats[0, :13] = spk.astype(np.float64).sum(0)
    stats[1, :13] = (spk.astype(np.float64) ** 2).sum(0)
    stats[0, 13] = len(spk)
    xform = synth.affine(39, seed=3)
    out = run("post", dict(feats=feats, cmvn_stats=stats, xform=xform), dict())
    save("post_13", feats=feats, cmvn_stats=stats, xform=xform, **{"ref_" + k: v for k, v in out.items()})
    # linear (LDA-shaped) transform on the deltas and a tiny matrix (rows < 5 column-header case
    # is only reachable with the explicit speech-feature method; rows = 9 is the smallest auto case)
    feats2 = feats[:9]
    lda = (rng.standard_normal((40, 39)) * 0.2).astype(np.float32)
    out = run("post", dict(feats=feats2, cmvn_stats=stats, xform=lda), dict(delta_window=3, splice_left=2, splice_right=1))
    save("post_9_lda", feats=feats2, cmvn_stats=stats, xform=lda,
         opts=opts_str(dict(delta_window=3, splice_left=2, splice_right=1)),
         **{"ref_" + k: v for k, v in out.items()})
    # the [300 x 13] compression case of SURVEY §8c(3), constant column included
    feats3 = (rng.standard_normal((300, 13)) * 5).astype(np.float32)
    feats3[:, 4] = 2.5
    out = run("post", dict(feats=feats3, cmvn_stats=stats, xform=xform), dict())
    save("post_300_cm", feats=feats3, ref_cm=out["cm"], ref_cm_back=out["cm_back"])


def gen_gmm():
    for name, P, G, D, T, zw in [("gmm_p8_d39", 8, 36, 39, 20, True), ("gmm_p64_d40", 64, 300, 40, 12, False)]:
        offs, w, means, iv = synth.gmm_model(P, G, D, seed=P, zero_weight=zw)
        rng = np.random.default_rng(P + 1)
        feats = (rng.standard_normal((T, D)) * 1.5).astype(np.float32)
        out = run("gmm", dict(offs=offs, weights=w, means=means, inv_vars=iv, feats=feats), dict())
        save(name, offs=offs, weights=w, means=means, inv_vars=iv, feats=feats, gconsts=out["gconsts"],
             miv=out["miv"], loglikes=out["loglikes"])


def decode_inputs(S, P, T, seed, **gkw):
    ntid = 3 * P
    g = synth.decoding_graph(S, ntid, seed=seed, num_words=50, **gkw)
    t2p = synth.tid2pdf(ntid, P, seed=seed)
    rng = np.random.default_rng(seed + 100)
    # loglikes with a realistic spread after the 1/12 acoustic scale
    ll = (rng.standard_normal((T, P)) * 30.0 - 80.0).astype(np.float32)
    arcs_i = np.stack([g["arcs"]["ilabel"], g["arcs"]["olabel"], g["arcs"]["nextstate"]], 1).astype(np.int32)
    inputs = dict(fst_final=g["final"], fst_off=g["arc_offset"], fst_arcs_i=np.ascontiguousarray(arcs_i),
                  fst_arcs_w=g["arcs"]["weight"].astype(np.float32), loglikes=ll, tid2pdf=t2p)
    return inputs


def gen_decode():
    graphs = {
        "g1": decode_inputs(400, 16, 60, seed=1),
        "g2": decode_inputs(3000, 40, 50, seed=2, eps_frac=0.5, final_frac=0.2),
        "g3_nofinal": decode_inputs(500, 16, 40, seed=3, unreachable_final=True),
    }
    cfgs = {
        "beam13": dict(beam=13.0, max_active=7000, min_active=200, lattice_beam=6.0),
        "beam10": dict(beam=10.0, max_active=2000, min_active=200, lattice_beam=6.0),
        "maxact": dict(beam=13.0, max_active=60, min_active=20, lattice_beam=6.0),
        "beamonly": dict(beam=8.0, max_active=2147483647, min_active=0, lattice_beam=4.0),
    }
    for gname, inputs in graphs.items():
        for cname, cfg in cfgs.items():
            o = dict(cfg, acwt=0.083333, start=0, prune_interval=25 if cname != "beam10" else 7)
            out = run("decode", inputs, o)
            save("decode_%s_%s" % (gname, cname), opts=opts_str(o), **inputs,
                 **{"ref_" + k: v for k, v in out.items()})


def gen_files():
    offs, w, means, iv = synth.gmm_model(6, 20, 39, seed=9)
    inp = decode_inputs(50, 6, 5, seed=4)
    rng = np.random.default_rng(8)
    feats = (rng.standard_normal((30, 39)) * 2).astype(np.float32)
    wave = synth.wave(1, seconds=0.3).astype(np.float32)
    inputs = dict(offs=offs, weights=w, means=means, inv_vars=iv, feats=feats, wave=wave,
                  fst_final=inp["fst_final"], fst_off=inp["fst_off"], fst_arcs_i=inp["fst_arcs_i"],
                  fst_arcs_w=inp["fst_arcs_w"])
    out = run("files", inputs, dict(start=0, seed=7))
    save("files", **inputs, **out)


def copy_htk():
    dst = os.path.join(GOLD, "htk")
    os.makedirs(dst, exist_ok=True)
    for fn in ["test.wav"] + ["test.wav.fea_htk.%d" % i for i in range(1, 7)]:
        shutil.copyfile(os.path.join(REF_TEST_DATA, fn), os.path.join(dst, fn))
        os.chmod(os.path.join(dst, fn), 0o644)


if __name__ == "__main__":
    os.makedirs(GOLD, exist_ok=True)
    which = sys.argv[1:] or ["mfcc", "post", "gmm", "decode", "files", "htk"]
    if "mfcc" in which:
        gen_mfcc()
    if "post" in which:
        gen_post()
    if "gmm" in which:
        gen_gmm()
    if "decode" in which:
        gen_decode()
    if "files" in which:
        gen_files()
    if "htk" in which:
        copy_htk()
