"""The step API (seam B1: MakeMfcc, ComputeCmvnStats, Decode incl. determinization and scoring) timed on a data directory, files in, files out
-- what an application calling the DLL sees, file I/O and per-job set-up included.  Synthetic directory: N utterances of 30 s, 4 per speaker,
the bench's model sizes and synthetic HCLG.  usage (under gpurun): python tools/bench_steps.py [--utts 256] [--nj 4] [--words 180000]"""
import argparse
import ctypes as C
import json
import os
import shutil
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from voicebridge_amd import build, refio, synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--utts", type=int, default=256)
    ap.add_argument("--seconds", type=float, default=30.0)
    ap.add_argument("--nj", type=int, default=4)
    ap.add_argument("--words", type=int, default=180000)
    ap.add_argument("--pdfs", type=int, default=2499)
    ap.add_argument("--gauss", type=int, default=15000)
    ap.add_argument("--job-threads", type=int, default=1, help="host threads the jobs of a step run on, all on GPU 0 (VB_STEP_DEVICES=0,0,...)")
    ap.add_argument("--fmllr", action="store_true", help="also time DecodeFmllr (SI pass, two fMLLR estimates, main search, rescoring, determinization)")
    a = ap.parse_args()
    if a.job_threads > 1:
        os.environ["VB_STEP_DEVICES"] = ",".join(["0"] * a.job_threads)
    host = C.CDLL(build.HOST_LIB)
    host.vb_steps_last_error.restype = C.c_char_p
    host.vb_DecodeEx.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_int, C.c_float, C.c_int, C.c_double, C.c_double,
                                 C.c_int, C.c_char_p, C.c_int, C.c_int]
    d = tempfile.mkdtemp(prefix="vb_steps_")
    try:
        data = os.path.join(d, "data", "test")
        os.makedirs(data)
        t0 = time.perf_counter()
        pool = [synth.wave(i, seconds=a.seconds) for i in range(8)]
        with open(os.path.join(data, "wav.scp"), "w") as fw, open(os.path.join(data, "utt2spk"), "w") as fu, open(os.path.join(data, "text"), "w") as ft:
            spk2utt = {}
            for u in range(a.utts):
                p = os.path.join(data, "u%05d.wav" % u)
                refio.write_wav(p, pool[u % 8])
                s = "s%04d" % (u // 4)
                fw.write("s%04d-u%05d %s\n" % (u // 4, u, p))
                fu.write("s%04d-u%05d %s\n" % (u // 4, u, s))
                ft.write("s%04d-u%05d w1 w2\n" % (u // 4, u))
                spk2utt.setdefault(s, []).append("s%04d-u%05d" % (u // 4, u))
        with open(os.path.join(data, "spk2utt"), "w") as f:
            for s, us in spk2utt.items():
                f.write(s + " " + " ".join(us) + "\n")
        conf = os.path.join(d, "mfcc.conf")
        open(conf, "w").write("--dither=0\n")
        exp = os.path.join(d, "exp", "tri3")
        graph = os.path.join(exp, "graph")
        os.makedirs(graph)
        H = 3
        n_cd = a.pdfs // H
        g, _ = synth.lexicon_graph(a.words, n_cd * H, seed=11)
        offs, w, means, iv = synth.gmm_model(n_cd * H, a.gauss, 39, seed=1, mean_scale=1.5, var_spread=0.2)
        refio.write_mdl(os.path.join(exp, "final.mdl"), n_cd, H, offs, w, (means * iv).astype(np.float32), iv)
        refio.write_vector_fst(os.path.join(graph, "HCLG.fst"), g["final"], g["arc_offset"], g["arcs"], g["start"])
        with open(os.path.join(graph, "words.txt"), "w") as f:
            f.write("<eps> 0\n" + "".join("w%d %d\n" % (i, i) for i in range(1, a.words + 2)))
        t_setup = time.perf_counter() - t0
        frames = a.utts * int(1 + (a.seconds * 16000 - 400) // 160)
        out = {"utterances": a.utts, "audio_seconds": a.utts * a.seconds, "frames": frames, "nj": a.nj, "graph_states": len(g["final"]), "setup_s": t_setup}
        t0 = time.perf_counter()
        assert host.vb_MakeMfcc(data.encode(), conf.encode(), a.nj, 1, 0) == 0, host.vb_steps_last_error()
        out["MakeMfcc_s"] = time.perf_counter() - t0
        t0 = time.perf_counter()
        assert host.vb_ComputeCmvnStatsDir(data.encode()) == 0, host.vb_steps_last_error()
        out["ComputeCmvnStats_s"] = time.perf_counter() - t0
        dec = os.path.join(exp, "decode_test")
        t0 = time.perf_counter()
        assert host.vb_DecodeEx(graph.encode(), data.encode(), dec.encode(), b"", b"", a.nj, 0.083333, 7000, 13.0, 6.0, 1, b"0.0,0.5,1.0", 7, 17) == 0, host.vb_steps_last_error()
        out["Decode_s (search + determinization, lattices and words written)"] = time.perf_counter() - t0
        t0 = time.perf_counter()
        host.vb_ScoreKaldiWER.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_int, C.c_char_p, C.c_int, C.c_int]
        assert host.vb_ScoreKaldiWER(data.encode(), graph.encode(), dec.encode(), a.nj, b"0.0,0.5,1.0", 7, 17) == 0, host.vb_steps_last_error()
        out["ScoreKaldiWER_s (33 LMWT x penalty combinations)"] = time.perf_counter() - t0
        if a.fmllr:
            os.makedirs(os.path.join(graph, "phones"), exist_ok=True)
            open(os.path.join(graph, "phones", "silence.csl"), "w").write("1:2\n")
            host.vb_DecodeFmllr.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_int, C.c_float, C.c_int, C.c_double, C.c_double, C.c_double, C.c_char_p,
                                            C.c_int, C.c_char_p, C.c_int, C.c_int]
            dec2 = os.path.join(exp, "decode_fmllr")
            t0 = time.perf_counter()
            assert host.vb_DecodeFmllr(graph.encode(), data.encode(), dec2.encode(), a.nj, 0.083333, 7000, 13.0, 6.0, 0.01, b"", 1, b"0.0", 9, 9) == 0, host.vb_steps_last_error()
            out["DecodeFmllr_s (SI pass + 2 fMLLR estimates + main search + rescoring + determinization)"] = time.perf_counter() - t0
        tot = out["MakeMfcc_s"] + out["ComputeCmvnStats_s"] + out["Decode_s (search + determinization, lattices and words written)"] + out["ScoreKaldiWER_s (33 LMWT x penalty combinations)"]
        out["total_s"] = tot
        out["frames_per_s"] = frames / tot
        out["rtf"] = tot / (a.utts * a.seconds)
        print(json.dumps(out), flush=True)
    finally:
        shutil.rmtree(d, ignore_errors=True)


if __name__ == "__main__":
    main()
