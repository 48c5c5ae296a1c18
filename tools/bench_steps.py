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
    ap.add_argument("--words", type=int, default=228000)
    ap.add_argument("--depth", type=float, default=0.25, help="spread of the pdfs' spectral envelopes (synth_speech: search difficulty)")
    ap.add_argument("--train-utts", type=int, default=128)
    ap.add_argument("--pdfs", type=int, default=2499)
    ap.add_argument("--gauss", type=int, default=15000)
    ap.add_argument("--job-threads", type=int, default=4, help="host threads the jobs of a step run on, all on GPU 0 (VB_STEP_DEVICES=0,0,...)")
    ap.add_argument("--repeat", type=int, default=2, help="calls of Decode / DecodeFmllr: the LAST one is reported (the process's graph, models and search workspaces exist: "
                    "the rate of a process that decodes set after set), the first one beside it as *_first_call_s (includes loading the graph and making the workspaces)")
    ap.add_argument("--fmllr", action="store_true", help="also time DecodeFmllr (SI pass, two fMLLR estimates, main search, rescoring, determinization)")
    a = ap.parse_args()
    os.environ["VB_STEP_DEVICES"] = ",".join(["0"] * max(a.job_threads, 1))      # (one entry: the jobs one after the other)
    import torch        # before the product's libraries: torch brings its own HIP runtime, and the one loaded first serves the process
    host = C.CDLL(build.HOST_LIB)
    host.vb_steps_last_error.restype = C.c_char_p
    host.vb_DecodeEx.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_int, C.c_float, C.c_int, C.c_double, C.c_double,
                                 C.c_int, C.c_char_p, C.c_int, C.c_int]
    d = tempfile.mkdtemp(prefix="vb_steps_")
    try:
        data = os.path.join(d, "data", "test")
        os.makedirs(data)
        t0 = time.perf_counter()
        # speech-like corpus (voicebridge_amd/synth_speech.py): word sequences of the lexicon, 4 utterances per speaker, transcripts in `text`
        from voicebridge_amd import api, synth_speech as ss
        H = 3
        n_cd = a.pdfs // H
        P = n_cd * H
        g, _t2p, lex = synth.lexicon_graph(a.words, P, seed=11, return_lexicon=True)
        ctx = api.Context(0)
        dev = ctx.device
        n_hops = int(round(a.seconds * 16000)) // ss.HOP
        gains = torch.from_numpy(ss.pdf_gains(P, a.depth)).to(dev)
        ids = np.arange(a.utts)
        hop_pdf, truth = ss.transcripts(lex, ids, n_hops)
        with open(os.path.join(data, "wav.scp"), "w") as fw, open(os.path.join(data, "utt2spk"), "w") as fu, open(os.path.join(data, "text"), "w") as ft:
            spk2utt = {}
            for c0 in range(0, a.utts, 64):
                c1 = min(c0 + 64, a.utts)
                sg = torch.from_numpy(ss.speaker_gains(ids[c0:c1] // 4)).to(dev)
                wv = ss.waves_device(hop_pdf[c0:c1], gains, ids[c0:c1], dev, spk_gain=sg).cpu().numpy()
                for u in range(c0, c1):
                    p = os.path.join(data, "u%05d.wav" % u)
                    refio.write_wav(p, wv[u - c0])
                    s = "s%04d" % (u // 4)
                    fw.write("s%04d-u%05d %s\n" % (u // 4, u, p))
                    fu.write("s%04d-u%05d %s\n" % (u // 4, u, s))
                    ft.write("s%04d-u%05d %s\n" % (u // 4, u, " ".join("w%d" % w for w in truth[u])))
                    spk2utt.setdefault(s, []).append("s%04d-u%05d" % (u // 4, u))
        with open(os.path.join(data, "spk2utt"), "w") as f:
            for s, us in spk2utt.items():
                f.write(s + " " + " ".join(us) + "\n")
        conf = os.path.join(d, "mfcc.conf")
        open(conf, "w").write("--dither=0\n")
        exp = os.path.join(d, "exp", "tri3")
        graph = os.path.join(exp, "graph")
        os.makedirs(graph)
        # the acoustic model: fit to other utterances by other speakers through the product's MFCC + CMVN + deltas (the features the
        # speaker-independent pass of Decode / DecodeFmllr sees)
        mf = api.Mfcc(ctx)
        post = api.FeaturePipeline(ctx, cmvn=True, mode="delta")
        fit = ss.GmmFit(P, 39, dev)
        n_smp = n_hops * ss.HOP
        for c0 in range(0, a.train_utts, 64):
            c1 = min(c0 + 64, a.train_utts)
            tid = (1 << 24) + np.arange(c0, c1)
            th, _tw = ss.transcripts(lex, tid, n_hops, uniform=True)
            sg = torch.from_numpy(ss.speaker_gains((1 << 20) + np.arange(c0, c1) // 4)).to(dev)
            wv = ss.waves_device(th, gains, tid, dev, spk_gain=sg).reshape(-1)
            ctx.torch_sync()
            raw, foff = mf.compute_packed(wv, np.arange(c1 - c0 + 1, dtype=np.int64) * n_smp, True)
            ctx.sync()
            u2s = ((np.arange(c0, c1) // 4) - c0 // 4).astype(np.int32)
            st = post.cmvn_stats(raw, foff, u2s, int(u2s.max()) + 1)
            f = post.apply(raw, foff, utt2spk=u2s, cmvn_stats=st)
            ctx.sync()
            fit.add(f, ss.frame_labels(th, int(foff[1] - foff[0])))
        (offs, w, miv, iv), _unseen = fit.model(a.gauss)
        refio.write_mdl(os.path.join(exp, "final.mdl"), n_cd, H, offs, w, miv, iv)
        del mf, post, fit, gains
        torch.cuda.empty_cache()
        refio.write_const_fst(os.path.join(graph, "HCLG.fst"), g["final"], g["arc_offset"], g["arcs"], g["start"])      # (a ConstFst, as the reference's mkgraph leaves it)
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
        import shutil
        for rep in range(max(1, a.repeat)):
            shutil.rmtree(dec, ignore_errors=True)
            t0 = time.perf_counter()
            assert host.vb_DecodeEx(graph.encode(), data.encode(), dec.encode(), b"", b"", a.nj, 0.083333, 7000, 13.0, 6.0, 1, b"0.0,0.5,1.0", 7, 17) == 0, host.vb_steps_last_error()
            out["Decode_s (search + determinization, lattices and words written)"] = time.perf_counter() - t0
            if rep == 0:
                out["Decode_first_call_s"] = out["Decode_s (search + determinization, lattices and words written)"]
        t0 = time.perf_counter()
        host.vb_ScoreKaldiWER.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_int, C.c_char_p, C.c_int, C.c_int]
        assert host.vb_ScoreKaldiWER(data.encode(), graph.encode(), dec.encode(), a.nj, b"0.0,0.5,1.0", 7, 17) == 0, host.vb_steps_last_error()
        out["ScoreKaldiWER_s (33 LMWT x penalty combinations)"] = time.perf_counter() - t0
        bw = os.path.join(dec, "scoring_kaldi", "best_wer")
        if os.path.exists(bw):
            out["Decode_best_wer"] = open(bw).read().split(dec)[0].strip()
        if a.fmllr:
            os.makedirs(os.path.join(graph, "phones"), exist_ok=True)
            open(os.path.join(graph, "phones", "silence.csl"), "w").write("1:2\n")
            host.vb_DecodeFmllr.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_int, C.c_float, C.c_int, C.c_double, C.c_double, C.c_double, C.c_char_p,
                                            C.c_int, C.c_char_p, C.c_int, C.c_int]
            dec2 = os.path.join(exp, "decode_fmllr")
            for rep in range(max(1, a.repeat)):
                shutil.rmtree(dec2, ignore_errors=True)
                t0 = time.perf_counter()
                rc = host.vb_DecodeFmllr(graph.encode(), data.encode(), dec2.encode(), a.nj, 0.083333, 7000, 13.0, 6.0, 0.01, b"", 0, b"0.0,0.5,1.0", 7, 17)
                if rc != 0:         # show what the jobs logged before the directory goes away
                    for root, _dirs, files in os.walk(dec2):
                        for fn in sorted(files):
                            if fn.endswith(".log"):
                                txt = open(os.path.join(root, fn), errors="replace").read()
                                sys.stderr.write("---- %s\n%s\n" % (os.path.join(root, fn), txt[-1500:]))
                assert rc == 0, host.vb_steps_last_error()
                out["DecodeFmllr_s (SI pass + 2 fMLLR estimates + main search + rescoring + determinization + scoring)"] = time.perf_counter() - t0
                if rep == 0:
                    out["DecodeFmllr_first_call_s"] = time.perf_counter() - t0
            bw = os.path.join(dec2, "scoring_kaldi", "best_wer")
            if os.path.exists(bw):
                out["DecodeFmllr_best_wer"] = open(bw).read().split(dec2)[0].strip()
        tot = out["MakeMfcc_s"] + out["ComputeCmvnStats_s"] + out["Decode_s (search + determinization, lattices and words written)"] + out["ScoreKaldiWER_s (33 LMWT x penalty combinations)"]
        out["total_s"] = tot
        out["frames_per_s"] = frames / tot
        out["rtf"] = tot / (a.utts * a.seconds)
        print(json.dumps(out), flush=True)
        host.vb_release_device_cache()        # the calling thread's model / graph / workspace; then its context: a normal exit follows
        ctx.close()
    finally:
        shutil.rmtree(d, ignore_errors=True)


if __name__ == "__main__":
    main()
    sys.stdout.flush()
    sys.stderr.flush()
    if os.environ.get("VBH_BENCH_HARD_EXIT"):        # (round 2's way out, kept for comparison: see bench.py)
        os._exit(0)
