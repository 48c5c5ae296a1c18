#!/usr/bin/env python3
"""bench.py -- throughput of the VoiceBridge hot path on MI355X.

One "step" = one pass of the hot path over one batch of synthetic 16 kHz utterances that are
already resident in HBM: MFCC -> CMVN+deltas+fMLLR -> all-pdf GMM log-likelihoods
[-> lattice beam search, when the decoder is enabled].  Sizes are the DELTA+SAT sizes of
BASELINE.md (13 MFCC -> 39 dims, 2500 pdfs / 15000 Gaussians, 39x40 per-speaker affine).

    python bench.py --gpus N --steps K --warmup W

For N > 1 it is launched one rank per GPU (torch.distributed.run); utterance batches are
independent, so ranks share nothing on the data path (weak scaling) and a single all-reduce
collects the per-rank max time and frame counts at the end.

Prints ONE JSON line (rank 0).  `value` is whole-job frames/s (100 frames = 1 s of audio, so
RTF = 100 / value), the host lattice stage of gmm-latgen-faster (determinization) inside the
timed step.  `roofline` prices the dominant kernel from HIP-event device time measured here;
`cpu_baseline` times the REFERENCE's own programs (oracle/_ref/bin, compiled from its sources;
kind "reference") on a bounded sample of the same workload on this host's cores, or, where
they are not there, the oracle's restatement (kind "port") -- a reported baseline, not the target.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

F32_MFMA_PEAK_TFLOPS = 157.3     # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: HBM3E spec peak


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=16)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=1024, help="utterances per step per GPU (--scaling strong: the most one batch may hold)")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak",
                    help="weak (default; the N = 1 headline): every GPU searches --batch utterances per step, the job grows with N.  strong (BASELINE "
                         "config 5): ONE corpus of --utterances utterances, sharded by speaker over the ranks (dist.shard_speakers, as "
                         "VB/scr/utils/split_data.cpp:20-100 splits a data directory); a step = one pass of every rank over its share, in equal batches "
                         "of at most --batch utterances (the corpus is rounded up by a few speakers so that the shares divide evenly)")
    ap.add_argument("--utterances", type=int, default=10000, help="--scaling strong: utterances of the corpus (config 5: 10 000 x 30 s)")
    ap.add_argument("--host-threads-per-rank", type=int, default=0,
                    help="host threads of a rank's lattice stage (same as --det-threads; 0 = the rank's share of the node's cores, 16 at most): what an "
                         "8-GPU node with few cores leaves a rank can be rehearsed on one GPU")
    ap.add_argument("--seconds", type=float, default=30.0, help="utterance length")
    ap.add_argument("--pdfs", type=int, default=2500)
    ap.add_argument("--gauss", type=int, default=15000)
    ap.add_argument("--no-decode", action="store_true", help="stop after GMM scoring (BASELINE config 2)")
    ap.add_argument("--feat", choices=("delta", "lda"), default="delta",
                    help="delta: CMVN + deltas (39) + per-speaker fMLLR 39x40 (DELTA+SAT, config 3); lda: CMVN + splice +-4 (117) + LDA 40x117 + "
                         "per-speaker fMLLR 40x41 (LDA+MLLT+SAT, config 4)")
    ap.add_argument("--words", type=int, default=228000,
                    help="vocabulary of the synthetic HCLG (~11 states and ~22 arcs per word: 228 k words = 2.5 M states / 5.0 M arcs, the graph size of "
                         "SURVEY.md 8d config 3)")
    ap.add_argument("--beam", type=float, default=13.0)
    ap.add_argument("--lattice-beam", type=float, default=6.0)
    ap.add_argument("--max-active", type=int, default=7000)
    ap.add_argument("--workload", choices=("speech", "tones"), default="speech",
                    help="speech: utterances are word sequences of the graph's lexicon, every HMM state a noise burst with its pdf's spectral "
                         "envelope, acoustic model fit to such data through the product's front end (voicebridge_amd/synth_speech.py); hypotheses are "
                         "scored against the transcripts.  tones: round 1's tone-plus-noise waveforms under a randomly placed GMM")
    ap.add_argument("--depth", type=float, default=0.25,
                    help="speech workload: standard deviation (nats) of the pdfs' cepstral envelopes = how far apart the pdfs sound (search difficulty)")
    ap.add_argument("--train-utts", type=int, default=128, help="speech workload: utterances (uniformly drawn words) the acoustic model is fit to")
    ap.add_argument("--mean-scale", type=float, default=1.5, help="tones workload: spread of the synthetic GMM means (search difficulty)")
    ap.add_argument("--var-spread", type=float, default=0.2, help="sigma of the log-variances of the synthetic GMM")
    ap.add_argument("--sub-batches", type=int, default=1,
                    help="a step's utterances go through the pipeline as this many slices of whole speakers (front end and search per slice): more, "
                         "smaller persistent search grids in flight, whose drains overlap; the step stays --batch utterances")
    ap.add_argument("--batch-sets", type=int, default=4,
                    help="distinct batches (other speakers, other utterances) that rotate through the timed steps, so that no step re-searches what the "
                         "previous one left in the caches; 1 = the same batch every step (rounds 1-2)")
    ap.add_argument("--ahead", type=int, default=1, help="items the front end runs ahead of the search that is enqueued next (ring of pipeline + ahead score buffers)")
    ap.add_argument("--pipeline", type=int, default=2,
                    help="search grids in flight: n contexts/streams whose decoders share one pool of search workspaces; the grid of one "
                         "batch drains while the next ones' workgroups fill the freed slots; 1 = one batch at a time")
    ap.add_argument("--fe-cus", type=int, default=-1,
                    help="confine the front end (MFCC, post, GMM) to this many CUs, the ones the searches do not use (-1: all the CUs the search mask leaves "
                         "when batches overlap; 0: not confined, round 1's behaviour).  Unconfined, the GMM grid of the batch after next takes every CU a "
                         "draining search grid frees -- short workgroups, three per CU, that keep coming -- and the next search's workgroups, which need a CU "
                         "to themselves, start only when the GMM grid is through: consecutive searches did not overlap (r02 device timeline)")
    ap.add_argument("--search-cus", type=int, default=-1,
                    help="confine the search contexts to this many CUs with a CU-masked stream (0 or -1 = not confined, the default since r02). "
                         "Round 1 kept 1/8 of the CUs free for the front end of the batches ahead, which otherwise only runs while a search grid "
                         "drains (a persistent search workgroup owns its CU's whole LDS); that paid when utterance times spread 1:5.  With the speech "
                         "workload's even utterance times the drain is short, the front end (90 ms of full-chip work per batch) fits into it, and a "
                         "mask of 224 CUs turned out to give the searches ~200 (device timeline): 2.50 M frames/s unmasked against 2.25 M masked")
    ap.add_argument("--no-determinize", action="store_true",
                    help="skip the second timed region, in which the host stage that follows every search in the reference's gmm-latgen-faster "
                         "(lattice determinization) is part of the step: a pool of host threads determinizes the lattices of batch k while the GPU "
                         "searches batch k+1")
    ap.add_argument("--det-steps", type=int, default=8, help="steps of that second timed region")
    ap.add_argument("--det-threads", type=int, default=0, help="host threads of the lattice stage (0: 16, the CPU share of a one-GPU box)")
    ap.add_argument("--save-waves", default="", help="write the rank's synthesised PCM (int16 .npy) to this file and go on")
    ap.add_argument("--load-waves", default="", help="read the PCM a --save-waves run of the same --batch / --seconds wrote instead of synthesising it "
                                                      "on the device (profiling runs: rocprofv3 --pmc does not survive torch's generator kernels)")
    ap.add_argument("--probe", action="store_true", help="print search statistics of the first utterances and exit")
    ap.add_argument("--sweep-depth", default="",
                    help="speech workload: comma-separated --depth values: for each, the batch is synthesised, the model fit, one batch scored and "
                         "searched (no pipeline) and a JSON line printed (search rate, tokens per frame, bound shares, WER against the transcripts); "
                         "then exit")
    ap.add_argument("--sweep-mean-scale", default="",
                    help="comma-separated --mean-scale values: for each, one batch is scored and searched (one batch at a time, no pipeline) and a JSON "
                         "line with the search's difficulty (tokens, arcs per frame, share of frames bound by max-/min-active) and its rate is printed; "
                         "then exit.  RTF as a function of search difficulty")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-step-api", action="store_true", help="skip the step-API measurement (MakeMfcc / Decode / DecodeFmllr on a data directory, in a child process)")
    ap.add_argument("--step-api-utts", type=int, default=1024)
    ap.add_argument("--step-api-threads", type=int, default=4, help="host threads the nj = 4 jobs of a step run on, all on the one GPU (VB_STEP_DEVICES=0,0,..; the reference runs its jobs side by side)")
    ap.add_argument("--cpu-sample-seconds", type=float, default=240.0,
                    help="audio seconds one process of the CPU baseline decodes (whole speakers; the default is two speakers of four 30 s utterances, "
                         "about 20-30 s of one core depending on how hard the search is)")
    ap.add_argument("--cpu-jobs", type=int, default=8,
                    help="processes of the CPU baseline run side by side, one job of whole speakers each (the reference runs its nj jobs side by side; "
                         "8 x 8 utterances = the 64-utterance sample; `cores` in the bench line says how many were used)")
    return ap.parse_args()


def build_graph(args, lexicon=False):
    """synthetic HCLG: lexicon prefix tree x 3-state HMMs x unigram loop (voicebridge_amd/synth.py);
    ~2.5 M states / 5 M arcs at the default 228 k words"""
    from voicebridge_amd import synth
    return synth.lexicon_graph(args.words, args.pdfs, seed=11, return_lexicon=lexicon)


UTT_PER_SPK = 4
SPEECH_XFORM_SCALE = 0.005


def shared_objects(tag, rank, world, dist, build):
    """What every rank needs and none has to make for itself (the 228 k-word graph: ~20 s of Python; the fitted acoustic model):
    rank 0 builds it and leaves the arrays in /dev/shm, the other ranks of the node wait at a barrier and map them (read-only
    numpy memory maps), rank 0 removes the files when everybody has them open.  `build` returns a tuple of dicts / arrays / ints;
    with one rank it is simply called."""
    if world <= 1:
        return build()
    import shutil
    d = "/dev/shm/vb_bench_%s_%s" % (os.environ.get("MASTER_PORT", "0"), tag)

    def save(obj, path):
        if isinstance(obj, dict):
            os.makedirs(path)
            for k, v in obj.items():
                save(v, os.path.join(path, "d_" + k))
        elif isinstance(obj, (tuple, list)):
            os.makedirs(path)
            for i, v in enumerate(obj):
                save(v, os.path.join(path, "t_%03d" % i))
        elif obj is None:
            os.makedirs(path + ".none")
        else:
            np.save(path + ".npy", np.asarray(obj))

    def load(path):
        if os.path.isdir(path + ".none"):
            return None
        if os.path.isdir(path):
            names = sorted(os.listdir(path))
            if names and names[0].startswith("t_"):
                return tuple(load(os.path.join(path, n.split(".")[0])) for n in sorted(set(n.split(".")[0] for n in names)))
            return {n.split(".")[0][2:]: load(os.path.join(path, n.split(".")[0])) for n in sorted(set(n.split(".")[0] for n in names))}
        a = np.load(path + ".npy", mmap_mode="r")
        return a[()] .item() if a.ndim == 0 else a
    if rank == 0:
        shutil.rmtree(d, ignore_errors=True)
        obj = build()
        save(obj, os.path.join(d, "root"))
    dist.barrier()
    if rank != 0:
        obj = load(os.path.join(d, "root"))
    dist.barrier()
    if rank == 0:
        shutil.rmtree(d, ignore_errors=True)          # (the maps of the other ranks keep the pages until they let go)
    return obj


def build_workload(args, rank, world, device, lex=None, set_index=0):
    """The job = world x batch utterances of world x batch/4 speakers (speakers of 4 utterances),
    sharded by speaker over the ranks as voicebridge_amd/dist.py does for a data directory (longest-processing-time
    assignment; all utterances have the same length here, so every rank gets batch/4 speakers).  Every utterance of the
    job is a distinct waveform, synthesised on the rank's device (synth.wave_batch_device).  Returns the packed int16
    samples on the device, the sample offsets, the rank-local utt2spk, the speaker count and the speakers' transforms."""
    import torch
    from voicebridge_amd import synth
    from voicebridge_amd import dist as vdist
    n_spk = max(1, args.batch // UTT_PER_SPK)
    upk = -(-args.batch // n_spk)
    frames_per_utt = 1 + (int(round(args.seconds * 16000)) - 400) // 160
    rank_of = vdist.shard_speakers(np.full(world * n_spk, upk * frames_per_utt, np.int64), world)
    my_spk = np.flatnonzero(rank_of == rank)
    assert len(my_spk) == n_spk
    my_spk = my_spk + set_index * world * n_spk        # (further batches of the job: other speakers, other utterances)
    utt2spk = (np.arange(args.batch, dtype=np.int32) // upk).astype(np.int32)      # rank-local speaker index: a speaker's utterances are adjacent
    gid = my_spk[utt2spk] * upk + np.arange(args.batch) % upk             # job-wide utterance id
    n = int(round(args.seconds * 16000))
    truth = None
    if lex is not None:
        from voicebridge_amd import synth_speech as ss
        n = (n // ss.HOP) * ss.HOP
        hop_pdf, truth = ss.transcripts(lex, gid, n // ss.HOP)
    if args.load_waves:
        h_wave = np.load(args.load_waves)
        if h_wave.dtype != np.int16 or h_wave.size != args.batch * n:
            sys.exit("bench.py: %s does not hold %d x %d int16 samples" % (args.load_waves, args.batch, n))
        d_wave = torch.from_numpy(h_wave.reshape(-1)).to(device)
    else:
        d_wave = torch.empty(args.batch * n, dtype=torch.int16, device=device)
        if lex is not None:
            gains = torch.from_numpy(ss.pdf_gains(args.pdfs, args.depth)).to(device)
            sgain = torch.from_numpy(ss.speaker_gains(my_spk)).to(device)
        for a in range(0, args.batch, 64):
            b_ = min(a + 64, args.batch)
            if lex is not None:
                d_wave[a * n:b_ * n] = ss.waves_device(hop_pdf[a:b_], gains, gid[a:b_], device, spk_gain=sgain[utt2spk[a:b_]]).reshape(-1)
            else:
                d_wave[a * n:b_ * n] = synth.wave_batch_device(gid[a:b_], args.seconds, device).reshape(-1)
    if args.save_waves:
        np.save(args.save_waves, d_wave.cpu().numpy())
    woff = np.arange(args.batch + 1, dtype=np.int64) * n
    # the speakers' fMLLR-like transforms: for the speech workload close to the identity (an adaptation transform that scrambled the
    # features by as much as the pdfs differ would make the model useless for every speaker but its own)
    xforms = np.stack([synth.affine(args.feat_dim, seed=int(s), scale=SPEECH_XFORM_SCALE if lex is not None else 0.05) for s in my_spk])
    return d_wave, woff, utt2spk, n_spk, xforms, gid, truth


def speech_model(args, lex, mf, post, ctx, device):
    """The acoustic model of the speech workload: --train-utts utterances of uniformly drawn words (so that every phone is
    heard), spoken by speakers of their own and taken through the bench's front end (MFCC, per-speaker CMVN, deltas or
    splice + LDA, the speakers' transforms); per-pdf mean / variance of the labelled frames, spread over the pdf's
    mixture components (synth_speech.GmmFit).  Every rank fits the same model (the seeds do not depend on the rank)."""
    import torch
    from voicebridge_amd import synth, synth_speech as ss
    n = (int(round(args.seconds * 16000)) // ss.HOP) * ss.HOP
    gains = torch.from_numpy(ss.pdf_gains(args.pdfs, args.depth)).to(device)
    fit = ss.GmmFit(args.pdfs, args.feat_dim, device)
    first_id, first_spk = 1 << 24, 1 << 20          # ids no job utterance / speaker has
    for a in range(0, args.train_utts, 64):
        b_ = min(a + 64, args.train_utts)
        ids = first_id + np.arange(a, b_)
        spk = first_spk + np.arange(a, b_) // UTT_PER_SPK
        u2s = (np.arange(a, b_) // UTT_PER_SPK - a // UTT_PER_SPK).astype(np.int32)
        hop_pdf, _w = ss.transcripts(lex, ids, n // ss.HOP, uniform=True)
        sg = torch.from_numpy(ss.speaker_gains(spk)).to(device)
        w = ss.waves_device(hop_pdf, gains, ids, device, spk_gain=sg).reshape(-1)
        woff = np.arange(b_ - a + 1, dtype=np.int64) * n
        ctx.torch_sync()
        raw, foff = mf.compute_packed(w, woff, True)
        ctx.sync()
        n_spk = int(u2s.max()) + 1
        xf = np.stack([synth.affine(args.feat_dim, seed=int(s_), scale=SPEECH_XFORM_SCALE) for s_ in np.unique(spk)])
        st = post.cmvn_stats(raw, foff, u2s, n_spk)
        f = post.apply(raw, foff, utt2spk=u2s, cmvn_stats=st, spk_xforms=xf)
        ctx.sync()
        nf = int(foff[1] - foff[0])
        fit.add(f, ss.frame_labels(hop_pdf, nf))
    model, n_unseen = fit.model(args.gauss)
    return model, n_unseen


def synth_lda(args):
    from voicebridge_amd import synth
    return synth.lda_matrix()


def feature_space_model(args, sample_feats):
    """random GMM placed in the feature space of the synthetic data (so that scores have a
    realistic spread): per-dimension scale = std of the post-processed features."""
    from voicebridge_amd import synth
    std = sample_feats.std(0) + 1e-3
    mu = sample_feats.mean(0)
    offs, w, means, iv = synth.gmm_model(args.pdfs, args.gauss, args.feat_dim, seed=1, mean_scale=args.mean_scale, dim_scale=std,
                                         var_spread=args.var_spread)
    means = means + mu.astype(np.float32)
    miv = (means * iv).astype(np.float32)
    return offs, w, miv, iv


def cpu_sample(args, utt2spk, n_spk):
    """utterances of the bounded CPU sample: whole speakers (CMVN statistics are per speaker), split into <= 4 jobs of whole
    speakers as the reference's split_data does (VB/scr/utils/split_data.cpp:20-100)"""
    per = max(1, int(round(args.cpu_sample_seconds / args.seconds)))          # utterances one process decodes (about 10-20 s of one core)
    upk = max(1, int(np.sum(np.asarray(utt2spk) == 0)))
    spk_per_job = max(1, per // upk)
    n_jobs = max(1, min(max(1, args.cpu_jobs), n_spk // spk_per_job))
    jobs = []
    for j in range(n_jobs):
        spk = list(range(j * spk_per_job, (j + 1) * spk_per_job))
        jobs.append([int(u) for u in np.flatnonzero(np.isin(utt2spk, spk))])
    return jobs


def cpu_baseline_reference(args, jobs, waves, utt2spk, xforms, lda, model, decode_setup):
    """The reference's own programs (oracle/_ref/bin: its sources compiled as they lie, OpenBLAS single-threaded in place of MKL) on a
    bounded sample of the bench workload: compute-mfcc-feats, compute-cmvn-stats, apply-cmvn, add-deltas (or splice-feats | transform-feats
    final.mat), transform-feats trans, gmm-latgen-faster --determinize-lattice=false (the raw-lattice search the GPU step is compared with).
    Two figures: len(jobs) processes side by side, one job each (the reference's nj jobs, VB/scr/steps/decode_gmm.cpp:300-324), and job 0
    alone on one core.  A process's time = wall-clock of its feature programs + the decoder's own "Time taken" (its loop over the
    utterances; model and graph loading excluded).  Also returns the words the reference decoded, per utterance, for the parity line.
    Returns None when the binaries are not there."""
    import re
    import shutil
    import subprocess
    import tempfile
    from concurrent.futures import ThreadPoolExecutor
    from voicebridge_amd import refio
    B = os.path.join(ROOT, "oracle", "_ref", "bin")
    need = ["compute-mfcc-feats", "compute-cmvn-stats", "apply-cmvn", "add-deltas", "splice-feats", "transform-feats", "gmm-latgen-faster"]
    if decode_setup is None or not all(os.access(os.path.join(B, b), os.X_OK) for b in need):
        return None
    offs, w, miv, iv = model
    g, _t2p, _cfg = decode_setup
    d = tempfile.mkdtemp(prefix="vb_cpu_ref_")
    env = dict(os.environ, OPENBLAS_NUM_THREADS="1", OMP_NUM_THREADS="1")
    try:
        p = lambda n: os.path.join(d, n)
        H = 3
        n_cd = args.pdfs // H
        refio.write_mdl(p("final.mdl"), n_cd, H, offs, w, miv, iv)
        refio.write_vector_fst(p("HCLG.fst"), g["final"], g["arc_offset"], g["arcs"], g["start"])
        if lda is not None:
            refio.write_feats_ark(p("final.mat.ark"), {"lda": lda})
            # a bare matrix file = the archive entry without its key
            blob = open(p("final.mat.ark"), "rb").read()
            open(p("final.mat"), "wb").write(blob[blob.index(b"\0B"):])
        for j, utts in enumerate(jobs):
            spks = sorted(set(int(utt2spk[u]) for u in utts))
            with open(p("wav.%d.scp" % j), "w") as f, open(p("utt2spk.%d" % j), "w") as fu, open(p("spk2utt.%d" % j), "w") as fs:
                for u in sorted(utts, key=lambda u_: "s%04d-u%05d" % (utt2spk[u_], u_)):
                    refio.write_wav(p("u%05d.wav" % u), waves[u])
                    f.write("s%04d-u%05d %s\n" % (utt2spk[u], u, p("u%05d.wav" % u)))
                    fu.write("s%04d-u%05d s%04d\n" % (utt2spk[u], u, utt2spk[u]))
                for s_ in spks:
                    fs.write("s%04d %s\n" % (s_, " ".join("s%04d-u%05d" % (s_, u) for u in sorted(utts) if utt2spk[u] == s_)))
            refio.write_feats_ark(p("trans.%d.ark" % j), {"s%04d" % s_: xforms[s_] for s_ in spks})

        def run(name, *a):
            t0 = time.perf_counter()
            r = subprocess.run([os.path.join(B, name)] + list(a), env=env, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True)
            if r.returncode != 0:
                raise RuntimeError(name + ": " + r.stderr[-300:])
            return time.perf_counter() - t0, r.stderr

        ref_like = {}

        def job(j, tag):
            """one job's program chain, as Decode() of the reference builds it (VB/scr/steps/decode_gmm.cpp:201-293)"""
            q = lambda n: p("%s.%d.%s" % (n, j, tag))
            t_feat = run("compute-mfcc-feats", "--dither=0", "scp:" + p("wav.%d.scp" % j), "ark:" + q("mfcc"))[0]
            t_feat += run("compute-cmvn-stats", "--spk2utt=ark:" + p("spk2utt.%d" % j), "ark:" + q("mfcc"), "ark:" + q("cmvn"))[0]
            t_feat += run("apply-cmvn", "--utt2spk=ark:" + p("utt2spk.%d" % j), "ark:" + q("cmvn"), "ark:" + q("mfcc"), "ark:" + q("cm"))[0]
            if lda is None:
                t_feat += run("add-deltas", "ark:" + q("cm"), "ark:" + q("dl"))[0]
            else:
                t_feat += run("splice-feats", "--left-context=4", "--right-context=4", "ark:" + q("cm"), "ark:" + q("sp"))[0]
                t_feat += run("transform-feats", p("final.mat"), "ark:" + q("sp"), "ark:" + q("dl"))[0]
            t_feat += run("transform-feats", "--utt2spk=ark:" + p("utt2spk.%d" % j), "ark:" + p("trans.%d.ark" % j), "ark:" + q("dl"), "ark:" + q("feats"))[0]
            t_wall, log = run("gmm-latgen-faster", "--beam=%g" % args.beam, "--lattice-beam=%g" % args.lattice_beam, "--max-active=%d" % args.max_active,
                              "--acoustic-scale=0.083333", "--allow-partial=true", "--determinize-lattice=false", p("final.mdl"), p("HCLG.fst"),
                              "ark:" + q("feats"), "ark:" + q("lat"), "ark,t:" + q("words"))
            m = re.search(r"Time taken ([0-9.eE+-]+)s", log)
            t_dec = float(m.group(1)) if m else t_wall
            m = re.findall(r"Overall log-likelihood per frame is \S+ over (\d+) frames", log)
            frames = int(m[-1]) if m else 0
            words = {}
            for line in open(q("words")):
                f_ = line.split()
                if f_:
                    words[int(f_[0].split("-u")[1])] = [int(x) for x in f_[1:]]
            for m_ in re.finditer(r"Log-like per frame for utterance \S+-u(\d+) is (\S+) over", log):
                ref_like[int(m_.group(1))] = float(m_.group(2))
            for n in ("mfcc", "cm", "dl", "sp", "feats", "lat"):
                if os.path.exists(q(n)):
                    os.remove(q(n))
            return t_feat, t_dec, frames, max(t_wall - t_dec, 0.0), words

        with ThreadPoolExecutor(len(jobs)) as ex:
            par = list(ex.map(lambda j: job(j, "par"), range(len(jobs))))
        one = job(0, "one")
        if one[2] <= 0 or any(r[2] <= 0 for r in par):
            return None
        words = {}
        for r in par:
            words.update(r[4])
        chain = ("compute-mfcc-feats, compute-cmvn-stats, apply-cmvn, %s, transform-feats, gmm-latgen-faster --determinize-lattice=false; OpenBLAS, "
                 "one thread per process, MKL not available" % ("add-deltas" if lda is None else "splice-feats, transform-feats final.mat"))
        t_par = max(r[0] + r[1] for r in par)
        out_par = dict(value=sum(r[2] for r in par) / t_par, unit="frames/s", cores=len(jobs), kind="reference",
                       sample="%d x %.0f s utterances (%d frames) in %d jobs of whole speakers run side by side, one process per job, through the "
                              "reference's own programs compiled from its sources (%s): slowest job %.2f s = features %.2f s + search %.2f s (its own "
                              "'Time taken'; loading the graph, %.1f s more per process, is left out)"
                              % (sum(len(j_) for j_ in jobs), args.seconds, sum(r[2] for r in par), len(jobs), chain, t_par,
                                 max(par, key=lambda r: r[0] + r[1])[0], max(par, key=lambda r: r[0] + r[1])[1], par[0][3]))
        out_one = dict(value=one[2] / (one[0] + one[1]), unit="frames/s", cores=1, kind="reference",
                       sample="job 0 of that sample alone (%d utterances, %d frames), one process: features %.2f s, search %.2f s" % (len(jobs[0]), one[2], one[0], one[1]))
        return out_par, out_one, words, ref_like
    finally:
        shutil.rmtree(d, ignore_errors=True)


def cpu_baseline(args, utts, waves, utt2spk, xforms, lda, model, decode_setup):
    """oracle (plain C restatement of the reference, 1 thread) on job 0 of the bounded sample; the fallback where the
    reference's programs are not there.  Returns the figure and the words it decoded."""
    from oracle import vbo
    offs, w, miv, iv = model
    o = vbo.mfcc_opts()
    gc = vbo.gmm_gconsts(w, miv, iv)
    fst = None
    t0 = time.perf_counter()
    frames = 0
    raws = {u: vbo.mfcc(waves[u].astype(np.float32), o) for u in utts}
    stats = {}
    for u in utts:
        st = vbo.acc_cmvn_stats(raws[u])
        stats[utt2spk[u]] = st if utt2spk[u] not in stats else stats[utt2spk[u]] + st
    words = {}
    for u in utts:
        f = vbo.apply_cmvn(raws[u], stats[utt2spk[u]])
        f = vbo.add_deltas(f) if lda is None else vbo.transform_feats(vbo.splice(f), lda)
        f = vbo.transform_feats(f, xforms[utt2spk[u]])
        ll = vbo.gmm_loglikes(offs, gc, miv, iv, f)
        if decode_setup is not None:
            g, t2p, cfg = decode_setup
            if fst is None:
                fst = vbo.make_fst(g["final"], g["arc_offset"], g["arcs"], g["start"])
            words[u] = [int(x) for x in vbo.decode_lattice_faster(fst, ll, t2p, **cfg)["words"]]
        frames += len(raws[u])
    dt = time.perf_counter() - t0
    return dict(value=frames / dt, unit="frames/s", cores=1, kind="port",
                sample="%d x %.0f s utterances (%d frames) through the oracle: MFCC, CMVN+%s+fMLLR, all-pdf "
                       "GMM scoring%s; on-demand scoring as in the reference would evaluate fewer pdfs"
                       % (len(utts), args.seconds, frames, "deltas" if lda is None else "splice+LDA",
                          ", lattice search" if decode_setup is not None else "")), words


def host_threads(args):
    """threads of the host stage: --det-threads, else the CPU share of a one-GPU box (16; more threads than cores only thrash)"""
    if args.det_threads > 0:
        return args.det_threads
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:
        n = os.cpu_count() or 16
    # (the ranks of a node share its cores: a rank takes its share, 16 at most)
    ranks_here = max(1, int(os.environ.get("LOCAL_WORLD_SIZE", os.environ.get("WORLD_SIZE", "1"))))
    return max(1, min(16, n // ranks_here))


def trans_info(args):
    """synth.lexicon_graph(): pdf p has the self-loop tid 2p+1 and the forward tid 2p+2; 3-state phones"""
    P = args.pdfs
    n_tid = 2 * P
    phone = np.zeros(n_tid + 1, np.int32); hmm = np.zeros(n_tid + 1, np.int32); sl = np.zeros(n_tid + 1, np.uint8)
    for p_ in range(P):
        for k_, tid in enumerate((2 * p_ + 1, 2 * p_ + 2)):
            phone[tid] = p_ // 3 + 1; hmm[tid] = p_ % 3; sl[tid] = 1 - k_
    return n_tid, phone, hmm, sl


class DetPool:
    """the lattice stage as part of the step: host threads determinize the lattices of the batch just searched
    (voicebridge_amd/host: vb_det_pool_*) while the GPU works on the next one"""

    def __init__(self, args):
        import ctypes as C
        from voicebridge_amd import build
        self.C = C
        self.host = C.CDLL(build.HOST_LIB)
        n_tid, phone, hmm, sl = trans_info(args)
        self.threads = host_threads(args)
        self.host.vb_det_pool_create.restype = C.c_void_p
        self.host.vb_det_pool_create.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_float]
        self.host.vb_det_pool_submit.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
        self.host.vb_det_pool_wait.argtypes = [C.c_void_p, C.c_void_p]
        self.host.vb_det_pool_destroy.argtypes = [C.c_void_p]
        self.h = self.host.vb_det_pool_create(self.threads, n_tid, phone.ctypes.data, hmm.ctypes.data, sl.ctypes.data,
                                              C.c_float(args.lattice_beam), C.c_float(0.083333))
        self.submit_s = 0.0
        self.totals = np.zeros(6)

    def submit(self, dec, n_utt):
        t0 = time.perf_counter()
        if self.host.vb_det_pool_submit(self.h, dec.h, n_utt, 0) != 0:
            raise RuntimeError("vb_det_pool_submit failed")
        self.submit_s += time.perf_counter() - t0

    def wait(self):
        out = (self.C.c_double * 6)()
        if self.host.vb_det_pool_wait(self.h, out) != 0:
            raise RuntimeError("vb_det_pool_wait failed")
        self.totals += np.array(list(out))

    def reset(self):
        self.submit_s = 0.0
        self.totals[:] = 0

    def close(self):
        if self.h:
            self.host.vb_det_pool_destroy(self.h)
            self.h = None


def determinize_measurement(args, dec, n_utt, T):
    """The step after the search in the reference's Decode (gmm-latgen-faster determinizes every
    utterance's lattice before writing it, K/decoder/decoder-wrappers.cc:262-284): host C++
    (voicebridge_amd/host/lattice.cc) over the lattices of the last batch, on the host's cores.
    Measured beside the GPU step, not inside it: the pipeline would hide it behind the next
    batch's search as long as its rate stays above the search's."""
    import ctypes as C
    from voicebridge_amd import build
    host = C.CDLL(build.HOST_LIB)
    n_tid, phone, hmm, sl = trans_info(args)
    threads = host_threads(args)
    out = (C.c_double * 6)()
    host.vb_determinize_decoder_results.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_float,
                                                    C.c_int, C.c_void_p]
    if os.environ.get("VBH_DET_SWEEP"):       # diagnostic: how many host threads does the box really give us?
        for th in (4, 8, 16, 24, 32, 48, 64):
            host.vb_determinize_decoder_results(dec.h, n_utt, 0, n_tid, phone.ctypes.data, hmm.ctypes.data, sl.ctypes.data,
                                                C.c_float(args.lattice_beam), C.c_float(0.083333), th, out)
            print("determinize %d lattices on %d threads: %.3f s" % (n_utt, th, out[0]), file=sys.stderr, flush=True)
    rc = host.vb_determinize_decoder_results(dec.h, n_utt, 0, n_tid, phone.ctypes.data, hmm.ctypes.data, sl.ctypes.data,
                                             C.c_float(args.lattice_beam), C.c_float(0.083333), threads, out)
    if rc != 0:
        return {"error": "vb_determinize_decoder_results failed"}
    sec = out[0]
    return {"what": "Connect + two-pass pruned determinization (phone level, word level) + un-scaling of the %d lattices of the last batch, "
                    "host C++ on %d threads" % (n_utt, threads),
            "seconds": sec, "utterances_per_s": n_utt / sec, "frames_per_s": T / sec, "threads": threads,
            "raw_states_per_utt": out[1] / n_utt, "raw_arcs_per_utt": out[2] / n_utt,
            "det_states_per_utt": out[3] / n_utt, "det_arcs_per_utt": out[4] / n_utt, "stopped_before_beam": int(out[5])}


F64_MFMA_PEAK_TFLOPS = 78.6      # vendor figure for MI355X FP64 matrix (the micro-architecture guide lists no f64 row): UNVERIFIED --
F64_MFMA_MEASURED_TFLOPS = 47.6  # what tools/micro/mfma_f64_peak.hip sustains on the box (profiles/r02_mfma_f64_peak.json: independent
                                 # v_mfma_f64_16x16x4_f64 chains on every SIMD)


def ondemand_gmm_measurement(args, ctx, gmm, model, feats, T):
    """The on-demand form of the GMM scores (vbh_gmm_loglikes_pairs; the reference's semantics, SURVEY.md 8d: per frame only the pdfs
    somebody asks for are evaluated -- here 3 per frame, what a lattice holds, and 32 per frame): wall time of the call (upload of the pair
    lists, kernel, download of the scores) and the algorithmic gather traffic = touched Gaussians x (2D + 1) x 4 B + frames x D x 4 B."""
    import ctypes as C
    offs = np.asarray(model[0], np.int64)
    ng = np.diff(offs)
    fn = gmm.lib.vbh_gmm_loglikes_pairs
    fn.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]
    out = {}
    rng = np.random.default_rng(9)
    Tm = min(T, 1 << 20)                       # a million frames is enough to time it
    for per in (3, 32):
        poff = np.arange(Tm + 1, dtype=np.int64) * per
        ppdf = rng.integers(0, gmm.num_pdfs, Tm * per).astype(np.int32)
        res = np.zeros(Tm * per, np.float32)
        best = None
        for _ in range(2):
            t0 = time.perf_counter()
            if fn(gmm.h, feats.data_ptr(), Tm, poff.ctypes.data, ppdf.ctypes.data, res.ctypes.data) != 0:
                return {"error": "vbh_gmm_loglikes_pairs failed"}
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        gauss = float(ng[ppdf].sum())
        bytes_ = gauss * (2 * gmm.dim + 1) * 4.0 + Tm * gmm.dim * 4.0 + Tm * per * 4.0
        out["%d_pdfs_per_frame" % per] = {"frames": int(Tm), "call_ms": 1e3 * best, "pdf_evals_per_s": Tm * per / best, "gather_GBs": bytes_ / best / 1e9,
                                           "frac_of_hbm_peak": bytes_ / best / 1e9 / HBM_PEAK_GBS}
    out["note"] = ("call time incl. the H2D of the pair lists and the D2H of the scores; the model (4.8 MB) is L2-resident, so this is a gather out of "
                   "L2, not an HBM stream; compare roofline_gmm: the batched all-pdf MFMA kernel scores all %d pdfs of a frame in the time this form "
                   "needs for a few dozen (kernel_ms / frames against call_ms / (frames x pdfs) here)" % gmm.num_pdfs)
    return out


def fmllr_measurement(args, ctx, gmm, feats, foff, utt2spk, n_spk, T):
    """fMLLR sufficient statistics of one batch (vbh_fmllr_accumulate): every frame carries two
    (pdf, weight) posterior pairs, as a pruned-determinized lattice typically leaves; the accumulation kernel is
    priced against the f64 MFMA peak with its own HIP-event duration."""
    import ctypes as C
    rng = np.random.default_rng(5)
    poff = np.arange(T + 1, dtype=np.int64) * 2
    ppdf = rng.integers(0, gmm.num_pdfs, 2 * T).astype(np.int32)
    pw = np.tile(np.array([0.7, 0.3], np.float32), T)
    best = None
    for _ in range(2):
        beta, K, G, ms = gmm.fmllr_stats(feats, foff, utt2spk, n_spk, poff, ppdf, pw)
        best = ms if best is None else min(best, ms)
    b, f = C.c_double(), C.c_double()
    gmm.lib.vbh_fmllr_work(gmm.dim, int(T), C.byref(b), C.byref(f))
    return {"kernel": "fmllr_accumulate_kernel", "bound": "mfma (f64)", "achieved": f.value / (best * 1e-3) / 1e12, "peak": F64_MFMA_PEAK_TFLOPS,
            "unit": "TFLOP/s", "frac": f.value / (best * 1e-3) / 1e12 / F64_MFMA_PEAK_TFLOPS, "kernel_ms": best, "algorithmic_flops": f.value,
            "peak_note": "unverified vendor figure; measured with independent MFMA chains on every SIMD: %.1f TFLOP/s (profiles/r02_mfma_f64_peak.json)" % F64_MFMA_MEASURED_TFLOPS,
            "frac_of_measured_peak": f.value / (best * 1e-3) / 1e12 / F64_MFMA_MEASURED_TFLOPS,
            "algorithmic_bytes": b.value, "speakers": int(n_spk), "frames": int(T), "count_check": float(beta.sum()),
            "note": "per-speaker G_d = sum_t b_t[d] x+ x+^T (lower-triangle tiles), K, beta for the whole batch; flops count the padded 48-row tiles issued"}


def launch_ranks(args):
    """`python bench.py --gpus N` without a launcher around it: start one rank per GPU as children of a process that has
    not touched the GPU (torch.distributed.run, rendezvous on 127.0.0.1), forward their output and exit code.  Nothing is
    re-executed in place: a process that has initialised HIP must never be replaced by another program."""
    import socket
    import subprocess
    port = os.environ.get("MASTER_PORT")
    if not port:
        with socket.socket() as s_:
            s_.bind(("127.0.0.1", 0))
            port = str(s_.getsockname()[1])
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", port, os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd)


def main():
    args = parse()
    # (multi-process GPU work on this driver stack needs dmabuf IPC: RCCL's handle exchange fails with the legacy mode)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if args.gpus > 1 and "RANK" not in os.environ:
        sys.exit(launch_ranks(args))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    args.feat_dim = 40 if args.feat == "lda" else 39
    if args.host_threads_per_rank > 0:
        args.det_threads = args.host_threads_per_rank
    # --scaling strong: the corpus' speakers are dealt to the ranks (longest processing time first; equal lengths here, so the shares
    # differ by one speaker at most); every rank walks its share as `sets_per_step` batches of `batch` utterances
    args.sets_per_step = 1
    args.corpus_utts = None
    if args.scaling == "strong":
        from voicebridge_amd import dist as vdist_
        n_spk_all = -(-args.utterances // UTT_PER_SPK)
        share = int(np.bincount(vdist_.shard_speakers(np.full(n_spk_all, UTT_PER_SPK, np.int64), world), minlength=world).max())     # speakers of the largest share
        n_sets_ = -(-share * UTT_PER_SPK // max(args.batch, UTT_PER_SPK))
        args.batch = -(-share // n_sets_) * UTT_PER_SPK
        args.batch_sets = n_sets_
        args.sets_per_step = n_sets_
        args.corpus_utts = args.batch * n_sets_ * world
    if world != max(args.gpus, 1):
        sys.exit("bench.py: --gpus %d but WORLD_SIZE=%d: launch one rank per GPU (or plain `python bench.py --gpus N`)" % (args.gpus, world))
    import torch
    import torch.distributed as dist
    # VBH_BENCH_BACKEND=gloo: rehearsal of the N > 1 control flow on a box with fewer GPUs than ranks (ranks share the
    # devices round-robin, the few scalars and the end gather go through CPU tensors); the driver's runs use RCCL
    backend = os.environ.get("VBH_BENCH_BACKEND", "nccl")
    if backend == "gloo":
        local_rank %= max(torch.cuda.device_count(), 1)
    if world > 1:
        torch.cuda.set_device(local_rank)
        if backend == "gloo":
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    from voicebridge_amd import api
    depth_ = max(1, args.pipeline) if not args.no_decode else 1
    ncu_ = torch.cuda.get_device_properties(local_rank).multi_processor_count
    if args.search_cus < 0:
        args.search_cus = 0          # (r02: not masked; round 1 masked the searches to 7/8 of the CUs, see --search-cus)
    if args.fe_cus < 0:
        args.fe_cus = 0
    ctx_full = api.Context(local_rank)                 # set-up and the kernels timed alone on the chip
    ctx = api.Context(local_rank, cus=args.fe_cus, first_cu=args.search_cus) if args.fe_cus > 0 else ctx_full      # the front end of the step
    dev = ctx.device
    cdev = torch.device("cpu") if backend == "gloo" else dev        # where the tensors of the collectives live

    lex = None
    if args.workload == "speech":
        g, t2p, lex = shared_objects("graph", rank, world, dist, lambda: build_graph(args, lexicon=True))
    d_wave, woff, utt2spk, n_spk, xforms, gid, truth = build_workload(args, rank, world, dev, lex)
    is_i16 = True
    mf = api.Mfcc(ctx)
    if args.feat == "lda":
        lda = synth_lda(args)
        post = api.FeaturePipeline(ctx, cmvn=True, mode="splice", splice_left=4, splice_right=4, lda=lda, spk_xform_shape=(40, 41))
    else:
        lda = None
        post = api.FeaturePipeline(ctx, cmvn=True, mode="delta", spk_xform_shape=(39, 40))
    ctx.torch_sync()
    raw, foff = mf.compute_packed(d_wave, woff, is_i16)
    ctx.sync()
    stats = post.cmvn_stats(raw, foff, utt2spk, n_spk)
    feats = post.apply(raw, foff, utt2spk=utt2spk, cmvn_stats=stats, spk_xforms=xforms)
    ctx.sync()
    T = int(foff[-1])
    # the random GMM is placed in the feature space of the whole batch: a sample of frames spread over all utterances
    sample_idx = torch.linspace(0, T - 1, min(T, 20000), device=dev).long()
    n_unseen_pdfs = None
    if lex is not None:
        model, n_unseen_pdfs = shared_objects("model", rank, world, dist, lambda: speech_model(args, lex, mf, post, ctx, dev))
        model = tuple(np.ascontiguousarray(m_) for m_ in model)
    else:
        model = feature_space_model(args, feats[sample_idx].cpu().numpy())
    gmm = api.DiagGmm(ctx, *model)
    # score rows are padded to a multiple of 32 floats (128-byte lines: the GMM kernel's stores then cover whole sectors, vbh_gmm_loglikes_ld);
    # the searches take the matrix as it is (their num_pdfs argument is the row length), the pad columns are never read
    NPAD = gmm.padded_pdfs
    ll = torch.empty((T, NPAD), dtype=torch.float32, device=dev)
    ctx.torch_sync()

    decode_setup = None
    dec = None
    if not args.no_decode:
        if lex is None:
            g, t2p = build_graph(args)
        graph = api.Graph(ctx, g["final"], g["arc_offset"], g["arcs"], 0)
        Tmax = int(np.max(np.diff(foff)))
        dec = api.LatticeDecoder(ctx, graph, args.batch, Tmax, beam=args.beam, max_active=args.max_active,
                                 min_active=200, lattice_beam=args.lattice_beam, acoustic_scale=0.083333)
        decode_setup = (g, t2p, dict(acwt=0.083333, beam=args.beam, max_active=args.max_active, min_active=200,
                                     lattice_beam=args.lattice_beam, exact_cutoff=3, adm_bucket_bits=6))      # ordered admission: the product default
        ctx.sync()

    dec_enabled = dec is not None
    if args.sweep_depth and dec is not None and lex is not None:
        from voicebridge_amd import dist as vdist
        for dp in [float(x) for x in args.sweep_depth.split(",") if x]:
            args.depth = dp
            gmm.close()
            d_wave = raw = feats = None
            d_wave, woff, utt2spk, n_spk, xforms, gid, truth = build_workload(args, rank, world, dev, lex)
            ctx.torch_sync()
            raw, foff = mf.compute_packed(d_wave, woff, is_i16)
            ctx.sync()
            stats = post.cmvn_stats(raw, foff, utt2spk, n_spk)
            feats = post.apply(raw, foff, utt2spk=utt2spk, cmvn_stats=stats, spk_xforms=xforms)
            ctx.sync()
            model, n_unseen_pdfs = speech_model(args, lex, mf, post, ctx, dev)
            gmm = api.DiagGmm(ctx, *model)
            gmm.loglikes(feats, out=ll)
            ctx.sync()
            # frame accuracy of the model: is the spoken pdf the best-scoring one?
            from voicebridge_amd import synth_speech as ss
            n_hops = (int(round(args.seconds * 16000)) // ss.HOP)
            hop_pdf, _tw = ss.transcripts(lex, gid, n_hops)
            lab = torch.as_tensor(np.concatenate([ss.frame_labels(hop_pdf[u:u + 1], int(foff[u + 1] - foff[u]))[0] for u in range(args.batch)]), device=dev).long()
            acc = 0
            for a_ in range(0, T, 1 << 18):
                acc += int((ll[a_:a_ + (1 << 18), :gmm.num_pdfs].argmax(1) == lab[a_:a_ + (1 << 18)]).sum())
            best = None
            for _ in range(2):
                dec.decode(ll, foff, t2p)
                tm = dec.times()
                best = tm["kernel_ms"] if best is None else min(best, tm["kernel_ms"])
            c = dict(tok=0, arc=0, new=0, link=0, maxb=0, minb=0, lat=0, words=0, fin=0)
            hyps = []
            for u in range(args.batch):
                r = dec.result(u, arrays=False)
                c["tok"] += r["n_tokens_expanded"]; c["arc"] += r["n_arcs_expanded"]; c["new"] += r["n_tokens_created"]; c["link"] += r["n_links_created"]
                c["maxb"] += r["n_frames_maxactive_bound"]; c["minb"] += r["n_frames_minactive_bound"]; c["lat"] += r["num_links"]; c["words"] += len(r["words"])
                c["fin"] += int(r["reached_final"])
                hyps.append(r["words"])
            wc = dict(zip(vdist.COUNTER_NAMES, [int(x) for x in vdist.score(truth, hyps)]))
            print(json.dumps({"depth": dp, "utterances": args.batch, "frames": T, "search_kernel_ms": best, "frames_per_s_search_only": T / (best * 1e-3),
                              "tokens_expanded_per_frame": c["tok"] / T, "arcs_per_frame": c["arc"] / T,
                              "tokens_created_per_frame": c["new"] / T, "links_per_frame": c["link"] / T, "share_frames_maxactive_bound": c["maxb"] / T,
                              "share_frames_minactive_bound": c["minb"] / T, "lattice_arcs_per_frame": c["lat"] / T, "words_per_utterance": c["words"] / args.batch,
                              "reached_final": c["fin"], "frame_accuracy_of_the_model": acc / T, "pdfs_unseen_in_training": n_unseen_pdfs,
                              "wer_percent": 100.0 * wc["word_errs"] / max(wc["num_words"], 1), "ref_words_per_utterance": wc["num_words"] / args.batch,
                              "ins_del_sub": [wc["ins"], wc["del"], wc["sub"]], "batch_stats": dec.batch_stats()}), flush=True)
        return
    if args.sweep_mean_scale and dec is not None:
        for ms_ in [float(x) for x in args.sweep_mean_scale.split(",") if x]:
            args.mean_scale = ms_
            gm = api.DiagGmm(ctx, *feature_space_model(args, feats[sample_idx].cpu().numpy()))
            gm.loglikes(feats, out=ll)
            ctx.sync()
            best = None
            for _ in range(2):
                dec.decode(ll, foff, t2p)
                tm = dec.times()
                best = tm["kernel_ms"] if best is None else min(best, tm["kernel_ms"])
            c = dict(tok=0, arc=0, new=0, link=0, maxb=0, minb=0, lat=0, words=0)
            for u in range(args.batch):
                r = dec.result(u, arrays=False)
                c["tok"] += r["n_tokens_expanded"]; c["arc"] += r["n_arcs_expanded"]; c["new"] += r["n_tokens_created"]; c["link"] += r["n_links_created"]
                c["maxb"] += r["n_frames_maxactive_bound"]; c["minb"] += r["n_frames_minactive_bound"]; c["lat"] += r["num_links"]; c["words"] += len(r["words"])
            print(json.dumps({"mean_scale": ms_, "utterances": args.batch, "frames": T, "search_kernel_ms": best, "frames_per_s_search_only": T / (best * 1e-3),
                              "rtf_search_only": 100.0 * best * 1e-3 / T, "tokens_expanded_per_frame": c["tok"] / T, "arcs_per_frame": c["arc"] / T,
                              "tokens_created_per_frame": c["new"] / T, "links_per_frame": c["link"] / T, "share_frames_maxactive_bound": c["maxb"] / T,
                              "share_frames_minactive_bound": c["minb"] / T, "lattice_arcs_per_frame": c["lat"] / T, "words_per_utterance": c["words"] / args.batch,
                              "batch_stats": dec.batch_stats()}), flush=True)
            gm.close()
        return
    if args.probe and dec is not None and os.environ.get("VBH_PROBE_FASTER"):
        # the 1-best search (FasterDecoder, vbh_faster_*) on the same scores
        gmm.loglikes(feats, out=ll)
        ctx.sync()
        fd = api.FasterDecoder(ctx, graph, args.batch, Tmax, beam=args.beam, max_active=args.max_active, min_active=20,
                               acoustic_scale=0.083333)
        for rep_ in range(2):
            t0 = time.perf_counter()
            fd.decode(ll, foff, t2p)
            dt = time.perf_counter() - t0
        r0 = fd.result(0)
        print("FasterDecoder: %d utterances in %.3f s = %.0f frames/s; utt 0: ok=%d final=%d words=%d tokens/frame=%.0f"
              % (args.batch, dt, T / dt, r0["ok"], r0["reached_final"], len(r0["words"]), r0["n_tokens"] / max(r0["num_frames"], 1)))
        fd.close()
    if args.probe and dec is not None:
        gmm.loglikes(feats, out=ll)
        if os.environ.get("VBH_PROBE_REPEAT"):
            # determinism: the same batch searched n times must give the same retries, counters and lattices every time
            first, diffs = None, 0
            for rep_ in range(int(os.environ["VBH_PROBE_REPEAT"])):
                dec.decode(ll, foff, t2p)
                sig = []
                for u in range(args.batch):
                    r = dec.result(u)
                    sig.append((r["n_tokens_expanded"], r["n_arcs_expanded"], r["n_tokens_created"], r["n_links_created"], r["n_frames_maxactive_bound"],
                                r["n_frames_minactive_bound"], r["num_toks"], r["num_links"],
                                hash(r["tok_state"].tobytes() + r["tok_cost"].tobytes() + r["tok_extra"].tobytes() + r["link_src"].tobytes() + r["link_dst"].tobytes() +
                                     r["link_graph"].tobytes() + r["link_acoustic"].tobytes() + np.asarray(r["words"]).tobytes())))
                st_ = dec.batch_stats()
                if first is None:
                    first = (sig, st_)
                else:
                    bad = [u for u in range(args.batch) if sig[u] != first[0][u]]
                    diffs += len(bad) + int(st_ != first[1])
                    print("repeat %d: %d utterances differ from the first run%s; batch stats %s (first run %s)" % (rep_, len(bad), (" (first: %d)" % bad[0]) if bad else "", st_, first[1]), flush=True)
            print("determinism: %d differences over %s repeats of %d utterances" % (diffs, os.environ["VBH_PROBE_REPEAT"], args.batch), flush=True)
        t0 = time.perf_counter()
        dec.decode(ll, foff, t2p)
        dt = time.perf_counter() - t0
        print("decode of %d utterances: %.3f s  %s" % (args.batch, dt, " ".join("%s=%.1f" % kv for kv in dec.times().items())))
        tl = np.array([dec.timeline(u) for u in range(args.batch)], np.int64)
        t0_, t1_ = tl[:, 0].min(), tl[:, 1].max()
        dur = (tl[:, 1] - tl[:, 0]) / 1e5      # ms
        n_wg = len(set(tl[:, 2].tolist()))
        print("search grid: %d workgroups, span %.1f ms, busy %.1f%% of workgroup-time; per-utterance ms: min %.1f mean %.1f max %.1f; "
              "last start at %.1f ms" % (n_wg, (t1_ - t0_) / 1e5, 100.0 * dur.sum() / (n_wg * (t1_ - t0_) / 1e5), dur.min(), dur.mean(),
                                         dur.max(), (tl[:, 0].max() - t0_) / 1e5))
        if os.environ.get("VBH_PROBE_PROXY"):
            # does a cheap statistic of the scores predict an utterance's search time?
            fo = torch.as_tensor(np.asarray(foff, np.int64), device=dev)
            ll = ll[:, :gmm.num_pdfs]
            mx = ll.max(dim=1, keepdim=True).values
            for theta in (30.0, 60.0, 100.0, 156.0, 300.0):
                cnt = torch.zeros(T, device=dev)
                for a in range(0, T, 1 << 18):
                    cnt[a:a + (1 << 18)] = (ll[a:a + (1 << 18)] > mx[a:a + (1 << 18)] - theta).sum(1).float()
                cs = torch.cat([torch.zeros(1, device=dev, dtype=torch.float64), cnt.double().cumsum(0)])
                prox = ((cs[fo[1:]] - cs[fo[:-1]]) / (fo[1:] - fo[:-1]).clamp(min=1)).cpu().numpy()
                rk = lambda x: np.argsort(np.argsort(x))
                rho = np.corrcoef(rk(prox), rk(dur))[0, 1]
                print("proxy theta=%g: spearman rho with per-utterance time = %.3f (pearson %.3f)" % (theta, rho, np.corrcoef(prox, dur)[0, 1]))
        agg = np.zeros(6)
        for u in range(args.batch):
            r = dec.result(u, arrays=False)
            agg += [r["n_tokens_expanded"], r["n_arcs_expanded"], r["n_tokens_created"], r["n_links_created"], r["n_frames_maxactive_bound"], r["n_frames_minactive_bound"]]
        print("all utterances: tok_expanded/frame=%.0f arcs/frame=%.0f created/frame=%.0f links/frame=%.0f maxbound %.1f%% minbound %.1f%% of frames; %s; "
              "search kernel %.0f frames/s" % (agg[0] / T, agg[1] / T, agg[2] / T, agg[3] / T, 100 * agg[4] / T, 100 * agg[5] / T, dec.batch_stats(),
                                             T / (dec.times()["kernel_ms"] * 1e-3)))
        for u in range(min(args.batch, 4)):
            r = dec.result(u, arrays=False)
            nf = max(r["num_frames"], 1)
            print("utt %d: ok=%d final=%d frames=%d tok_expanded/frame=%.0f arcs/frame=%.0f created/frame=%.0f links/frame=%.0f "
                  "maxbound=%d minbound=%d lattice toks=%d links=%d words=%d best=(%.2f, %.2f)" % (
                      u, r["ok"], r["reached_final"], r["num_frames"], r["n_tokens_expanded"] / nf, r["n_arcs_expanded"] / nf,
                      r["n_tokens_created"] / nf, r["n_links_created"] / nf, r["n_frames_maxactive_bound"],
                      r["n_frames_minactive_bound"], r["num_toks"], r["num_links"], len(r["words"]), r["best_graph"], r["best_acoustic"]))
            ph = dec.phase_cycles(u)
            tot = max(sum(ph[:7]), 1)
            print("   phase share: " + ", ".join("%s %.1f%%" % (nm, 100.0 * c / tot) for nm, c in zip(
                ["cutoff", "pass1", "admit+links", "eps", "prune", "compact", "output"], ph[:7])) + "  cycles/frame=%.0f prune-visited-frames=%d" % (tot / nf, ph[7]))
            dg = dec.diag(u)
            if os.environ.get("VBH_PROBE_P1"):
                print("   pass 1 of wave 0, kilocycles per frame: round top (token/state wait) %.1f, owner search %.1f, arc wait %.1f, score wait %.1f, "
                      "compute + alloc %.1f, stores %.1f, loop exit %.1f, reduction barrier %.1f" % tuple(x / nf / 1e3 for x in dg[:8]))
            elif not (dg[8] or dg[9]):      # (not a -DVBH_DIAG build: [12] = candidates pass 1 wrote)
                print("   candidates/frame=%.0f (of %.0f arcs expanded; %.0f admitted)" % (dg[12] / nf, r["n_arcs_expanded"] / nf, r["n_links_created"] / nf))
            else:
                tot += dg[8] + dg[9]
                print("   VBH_DIAG: admit %.1f%% links %.1f%% cutoff-up-to-first-reduction %.1f%% rest-of-cutoff %.1f%%; admissions/frame %.0f, to the HBM map %.1f%%; "
                      "frames creating > 3/4 of the LDS map %d, sweeps of the pruning passes %d"
                      % (100.0 * dg[8] / tot, 100.0 * dg[2] / tot, 100.0 * dg[9] / tot, 100.0 * dg[0] / tot, dg[11] / nf, 100.0 * dg[10] / max(dg[11], 1), dg[12], dg[13]))
                print("   frames by tokens at their start (< 512, < 2048, < 6144, more): frames %s, kilocycles per frame %s"
                      % ([(dg[15] >> (16 * i)) & 0xFFFF for i in range(4)], [(dg[14] >> (16 * i)) & 0xFFFF for i in range(4)]))
        return

    # ---- pipeline (depth > 1).  Front end (MFCC, post, GMM) on its own context/stream, two
    #      batches ahead of the search: ring of depth+1 score buffers.  Searches alternate between
    #      `depth` decoder contexts/streams, so that while the persistent grid of batch k drains (its
    #      last, heaviest utterances) the workgroups of batch k+1 fill the freed CUs, and the front end
    #      of batch k+2 takes its share of them too.  depth 1 = one batch at a time on one stream.
    class Lane:
        pass
    depth = max(1, args.pipeline) if dec is not None else 1
    # ---- slices of a step: contiguous ranges of whole speakers
    S = max(1, args.sub_batches)
    if args.batch % (S * UTT_PER_SPK) != 0:
        sys.exit("bench.py: --batch must be a multiple of --sub-batches x %d (slices hold whole speakers)" % UTT_PER_SPK)
    per = args.batch // S

    class Slice:
        pass
    # further batch sets (--batch-sets): waveforms, CMVN statistics and transforms of their own; set 0 is the batch built above
    n_sets = max(1, args.batch_sets) if (dec is not None and args.workload == "speech" and not args.load_waves) else 1
    if args.scaling == "strong" and n_sets < args.sets_per_step:
        # (the step would otherwise walk the SAME batch sets_per_step times and still call it a corpus of N distinct utterances)
        sys.exit("bench.py: --scaling strong needs the speech workload with its own synthesis (no --load-waves, no --no-decode): a step is one pass over "
                 "%d DISTINCT batches per rank, and this configuration holds %d" % (args.sets_per_step, n_sets))
    sets = [(d_wave, woff, foff, utt2spk, stats, xforms)]
    for k_ in range(1, n_sets):
        w_, wo_, u2s_, nspk_, xf_, _gid, _truth = build_workload(args, rank, world, dev, lex, set_index=k_)
        ctx.torch_sync()
        raw_k, foff_k = mf.compute_packed(w_, wo_, is_i16, out=raw)        # (raw is scratch here: the pipeline recomputes it per item)
        ctx.sync()
        st_ = post.cmvn_stats(raw_k, foff_k, u2s_, nspk_)
        sets.append((w_, wo_, foff_k, u2s_, st_, xf_))
    slices = []        # [set * S + slice]
    for (dw_, wo_, fo_, u2s_, st_, xf_) in sets:
      for k in range(S):
        sl = Slice()
        sl.a, sl.b = k * per, (k + 1) * per
        sl.wave = dw_[int(wo_[sl.a]):int(wo_[sl.b])]
        sl.woff = np.ascontiguousarray(wo_[sl.a:sl.b + 1] - wo_[sl.a])
        sl.r0, sl.r1 = int(fo_[sl.a]), int(fo_[sl.b])
        sl.foff = np.ascontiguousarray(fo_[sl.a:sl.b + 1] - fo_[sl.a])
        s0, s1 = int(u2s_[sl.a]), int(u2s_[sl.b - 1]) + 1
        sl.u2s = np.ascontiguousarray(u2s_[sl.a:sl.b] - s0)
        sl.stats, sl.xforms = np.ascontiguousarray(st_[s0:s1]), np.ascontiguousarray(xf_[s0:s1])
        slices.append(sl)
    item0 = [0]          # items run so far: the batch sets rotate across the timed regions

    def slice_of(i):
        """slice of pipeline item i of the current run_steps call: set ((item0 + i) // S) % n_sets, part i % S"""
        return (((item0[0] + i) // S) % n_sets) * S + (i % S)
    Ts = max(sl.r1 - sl.r0 for sl in slices)
    ahead = max(1, args.ahead) if depth > 1 else 1
    n_ring = depth + ahead if depth > 1 else 1
    if S == 1:
        ring = [(raw, feats, ll)] + [(torch.empty_like(raw), torch.empty_like(feats), torch.empty_like(ll)) for _ in range(n_ring - 1)]
    else:
        del ll
        ring = [(torch.empty((Ts, raw.shape[1]), dtype=torch.float32, device=dev), torch.empty((Ts, feats.shape[1]), dtype=torch.float32, device=dev),
                 torch.empty((Ts, NPAD), dtype=torch.float32, device=dev)) for _ in range(n_ring)]
    lanes = []
    if dec is not None and (depth > 1 or S > 1):
        dec.close()          # the probe/setup decoder on the front-end context is not used by the pipeline
        dec = None
    for k in range(depth):
        L = Lane()
        if depth == 1 and S == 1:
            L.ctx, L.dec = ctx, dec
        elif dec_enabled:
            L.ctx = api.Context(local_rank, priority=int(os.environ.get("VBH_BENCH_DEC_PRIO", "0")), cus=args.search_cus) if depth > 1 else ctx
            # the lanes' decoders draw their search workspaces from one pool (vbh_decoder_create_shared)
            L.dec = api.LatticeDecoder(L.ctx, graph, per, Tmax, beam=args.beam, max_active=args.max_active,
                                       min_active=200, lattice_beam=args.lattice_beam, acoustic_scale=0.083333,
                                       share_slots_with=lanes[0].dec if lanes else None)
        else:
            L.ctx, L.dec = ctx, None
        L.pending = False
        L.timed = False
        L.slice = None
        lanes.append(L)
    ctx.torch_sync()
    gmm_ms, feat_ms, dec_ms, dec_k_ms, dec_h_ms = [], [], [], [], []
    fe_timed = []            # front-end timer slots in flight: (slot_feat, slot_gmm)
    # per-utterance results of one complete step, gathered by an untimed pass after the timed regions (collect[0]):
    # words, search counters, device timeline
    collect = [False]
    col_words = [None] * args.batch
    col_like = np.zeros(args.batch)
    col_cnt = np.zeros((args.batch, 8), np.float64)
    col_tl = np.zeros((args.batch, 3), np.int64)
    col_stats = dict(retried_in_big_slot=0, pool_grown=0, failed=0)

    def front_end(i, timed):
        """MFCC -> post -> GMM of item i (slice i % S of a step) into ring[i % n_ring], on the front-end stream"""
        sl_ = slices[slice_of(i)]
        r_, f_, l_ = ring[i % n_ring]
        n_rows = sl_.r1 - sl_.r0
        tsl = 1 + 2 * (i % 7)                # timer slots rotate: several items may be in flight
        if timed:
            ctx.timer_start(tsl)
        dbg_ = os.environ.get("VBH_BENCH_STAGE_SYNC")
        mf.compute_packed(sl_.wave, sl_.woff, is_i16, out=r_[:n_rows])
        if dbg_:
            ctx.sync(); print("item %d: mfcc done" % i, file=sys.stderr, flush=True)
        post.apply(r_[:n_rows], sl_.foff, utt2spk=sl_.u2s, cmvn_stats=sl_.stats, spk_xforms=sl_.xforms, out=f_[:n_rows])
        if dbg_:
            ctx.sync(); print("item %d: featpost done" % i, file=sys.stderr, flush=True)
        if timed:
            ctx.timer_stop(tsl)
            ctx.timer_start(tsl + 1)
        gmm.loglikes(f_[:n_rows], out=l_[:n_rows])
        if dbg_:
            ctx.sync(); print("item %d: gmm done" % i, file=sys.stderr, flush=True)
        if os.environ.get("VBH_BENCH_CHECK_LL"):     # diagnostic: are the scores sane while search grids are running beside the kernel?
            ctx.sync()
            lv_ = l_[:n_rows, :gmm.num_pdfs]
            bad = int((~torch.isfinite(lv_)).sum().item())
            mn, mx = float(lv_.min().item()), float(lv_.max().item())
            print("item %d: scores non-finite %d, min %.1f max %.1f" % (i, bad, mn, mx), file=sys.stderr, flush=True)
            if check_gmm[0] is not None:
                worst = 0.0
                for a_ in range(0, n_rows, 1 << 18):
                    b_ = min(n_rows, a_ + (1 << 18))
                    ref_ = check_gmm[0].loglikes(f_[a_:b_])
                    ctx.sync()
                    worst = max(worst, float((ref_ - l_[a_:b_, :gmm.num_pdfs]).abs().max().item()))
                    del ref_
                print("   vs the other GMM kernel: max abs diff %.3g" % worst, file=sys.stderr, flush=True)
        if timed:
            ctx.timer_stop(tsl + 1)
            fe_timed.append(tsl)
            if len(fe_timed) >= 7:           # read the oldest before its slot comes round again
                o_ = fe_timed.pop(0)
                feat_ms.append(ctx.timer_ms(o_)); gmm_ms.append(ctx.timer_ms(o_ + 1))

    check_gmm = [None]
    if os.environ.get("VBH_BENCH_CHECK_LL") and not os.environ.get("VBH_GMM_V1"):
        os.environ["VBH_GMM_V1"] = "1"
        check_gmm[0] = api.DiagGmm(ctx, *model)
        del os.environ["VBH_GMM_V1"]
    det_pool = DetPool(args) if (dec_enabled and not args.no_determinize) else None
    det_on = [False]         # the headline region times the GPU path (SURVEY.md 8 a-e); a second region adds the host lattice stage

    tl_dump = []

    def finish(L):
        """complete the lane's search in flight: lattices on the host, times collected"""
        if L.pending:
            L.dec.decode_end()
            L.pending = False
            if det_pool is not None and det_on[0]:
                det_pool.submit(L.dec, per)             # raw lattices copied out; determinized by the pool under the next search
            if collect[0]:
                sl_ = slices[L.slice]
                for u in range(per):
                    r = L.dec.result(u, arrays=False)
                    col_words[sl_.a + u] = r["words"]
                    col_like[sl_.a + u] = -(r["best_graph"] + r["best_acoustic"]) / max(r["num_frames"], 1)
                    col_cnt[sl_.a + u] = [r["n_tokens_expanded"], r["n_arcs_expanded"], r["n_tokens_created"], r["n_links_created"],
                                          r["n_frames_maxactive_bound"], r["n_frames_minactive_bound"], r["num_toks"], r["num_links"]]
                    col_tl[sl_.a + u] = L.dec.timeline(u)
                for k_, v_ in L.dec.batch_stats().items():
                    col_stats[k_] += v_
            if L.timed and os.environ.get("VBH_BENCH_TIMELINE"):      # diagnostic: device timestamps of every utterance of every timed batch
                tl_dump.append(np.array([L.dec.timeline(u) for u in range(per)], np.int64))
            if L.timed:
                tm = L.dec.times()      # HIP events around the search kernel on its stream + host clocks
                dec_ms.append(tm["total_ms"]); dec_k_ms.append(tm["kernel_ms"])
                dec_h_ms.append(tm["copy_ms"] + tm["host_ms"])
                L.timed = False

    def run_steps(n, timed, sets_per_step=None):
        """n complete steps = n x sets_per_step x S items (front end + search each), the pipeline empty before and after"""
        items = n * S * (args.sets_per_step if sets_per_step is None else sets_per_step)
        if items <= 0:
            return
        if depth == 1:
            L = lanes[0]
            for i in range(items):
                front_end(i, timed)
                if L.dec is not None:
                    L.timed = timed
                    L.slice = slice_of(i)
                    L.dec.decode_begin(ring[0][2], slices[slice_of(i)].foff, t2p)
                    L.pending = True
                    finish(L)
            item0[0] += items
            return
        # the front end runs `ahead` items in front of the search enqueued next; a search waits (on the device) for the front
        # end of its own item only: an event recorded on the front-end stream right after that item's scores
        n_fe = 0
        for i in range(items):
            L = lanes[i % depth]
            finish(L)                                    # search i - depth
            while n_fe < items and n_fe <= i + ahead - 1:
                front_end(n_fe, timed)
                ctx.mark(n_fe % n_ring)                  # "scores of item n_fe are complete" on the front-end stream
                n_fe += 1
            L.ctx.wait_mark(ctx, i % n_ring)
            L.timed = timed
            L.slice = slice_of(i)
            L.dec.decode_begin(ring[i % n_ring][2], slices[slice_of(i)].foff, t2p)
            L.pending = True
        for L in lanes:
            finish(L)
        item0[0] += items

    def barrier():
        for L in lanes:
            finish(L)
            L.ctx.sync()
        ctx.sync()
        torch.cuda.synchronize(dev)
        if det_pool is not None and det_on[0]:
            det_pool.wait()                             # the step is complete when its lattices are determinized
        if world > 1:
            dist.barrier()

    # headline region: the step INCLUDES the host stage the reference's gmm-latgen-faster applies to every utterance (Connect, phone- and
    # word-level pruned determinization, un-scaling): a pool of host threads determinizes the lattices of batch k while the GPU searches
    # batch k+1, and the region ends when the last batch is determinized (rounds 1-2 timed the raw-lattice path here)
    det_in_headline = det_pool is not None
    det_on[0] = det_in_headline
    run_steps(args.warmup, False)
    barrier()
    if det_pool is not None:
        det_pool.reset()
    t0 = time.perf_counter()
    run_steps(args.steps, True)
    barrier()
    dt = time.perf_counter() - t0
    det_totals = det_pool.totals.copy() if det_pool is not None else None
    det_submit_s = det_pool.submit_s if det_pool is not None else 0.0
    det_on[0] = False
    for o_ in fe_timed:
        feat_ms.append(ctx.timer_ms(o_)); gmm_ms.append(ctx.timer_ms(o_ + 1))
    del fe_timed[:]
    # second timed region: the GPU path alone (raw state-level lattices delivered to the host, no determinization)
    raw_dt = None
    if det_pool is not None and args.det_steps > 0:
        n_dec_timed = len(dec_ms)
        run_steps(1, False, 1)
        barrier()
        t0 = time.perf_counter()
        run_steps(args.det_steps, False, 1)
        barrier()
        raw_dt = time.perf_counter() - t0
        del dec_ms[n_dec_timed:], dec_k_ms[n_dec_timed:], dec_h_ms[n_dec_timed:]
    # one more (untimed) step through the same pipeline to gather every utterance's words, counters and timeline
    if dec_enabled:
        collect[0] = True
        item0[0] = 0           # (the collection step searches batch set 0: its transcripts and ids are the ones scored and gathered)
        run_steps(1, False, 1)
        barrier()
        collect[0] = False
    # after the timed region: the GMM kernel alone on the whole chip (in the pipeline it shares the
    # CUs with the search grids, so its in-pipeline duration says little about the kernel itself)
    n_rows0 = slices[0].r1 - slices[0].r0
    # (with the front end's context confined to a few CUs these measurements get objects of their own on an unconfined context)
    ctx.sync()
    if ctx_full is not ctx:
        gmm_m, mf_m = api.DiagGmm(ctx_full, *model), api.Mfcc(ctx_full)
    else:
        gmm_m, mf_m = gmm, mf
    gmm_alone_ms = None
    for _ in range(3):
        ctx_full.timer_start(0)
        gmm_m.loglikes(ring[0][1][:n_rows0], out=ring[0][2][:n_rows0])
        ctx_full.timer_stop(0)
        ctx_full.sync()
        t_ = ctx_full.timer_ms(0) * (T / float(n_rows0))          # (scaled to the step's frame count when the step runs in slices)
        gmm_alone_ms = t_ if gmm_alone_ms is None else min(gmm_alone_ms, t_)
    # the MFCC kernel alone on the whole chip, the same way (its share of the pipelined step says little about the kernel)
    mfcc_alone_ms = None
    for _ in range(3):
        ctx_full.timer_start(0)
        mf_m.compute_packed(slices[0].wave, slices[0].woff, is_i16, out=ring[0][0][:n_rows0])
        ctx_full.timer_stop(0)
        ctx_full.sync()
        t_ = ctx_full.timer_ms(0) * (T / float(n_rows0))
        mfcc_alone_ms = t_ if mfcc_alone_ms is None else min(mfcc_alone_ms, t_)
    # ... and the post-processing kernel (CMVN + deltas / splice + transforms), the same way
    featpost_alone_ms = None
    post_m = post if ctx_full is ctx else (api.FeaturePipeline(ctx_full, cmvn=True, mode="splice", splice_left=4, splice_right=4, lda=lda, spk_xform_shape=(40, 41))
                                           if args.feat == "lda" else api.FeaturePipeline(ctx_full, cmvn=True, mode="delta", spk_xform_shape=(39, 40)))
    for _ in range(3):
        ctx_full.timer_start(0)
        post_m.apply(ring[0][0][:n_rows0], slices[0].foff, utt2spk=slices[0].u2s, cmvn_stats=slices[0].stats, spk_xforms=slices[0].xforms, out=ring[0][1][:n_rows0])
        ctx_full.timer_stop(0)
        ctx_full.sync()
        t_ = ctx_full.timer_ms(0) * (T / float(n_rows0))
        featpost_alone_ms = t_ if featpost_alone_ms is None else min(featpost_alone_ms, t_)
    if dec_enabled:
        dec = lanes[(S - 1) % depth].dec                 # holds the lattices of the collection step's last slice
        dur = (col_tl[:, 1] - col_tl[:, 0]).astype(np.float64)
        span = float(col_tl[:, 1].max() - col_tl[:, 0].min())
        grid_stats = {"what": "the %d utterances of one step (searched as %d slice(s), %d search grid(s) in flight)" % (args.batch, S, depth),
                      "workgroups": len(set(col_tl[:, 2].tolist())), "span_ms": span / 1e5,
                      "mean_utterances_in_search": float(dur.sum()) / span,
                      "utterance_ms_min_mean_max": [float(dur.min()) / 1e5, float(dur.mean()) / 1e5, float(dur.max()) / 1e5]}
    SPS = args.sets_per_step
    per_rank = [{"rank": 0, "seconds": dt, "frames_per_s": T * SPS * args.steps / dt, "device": local_rank}]
    if world > 1:
        # every rank's own time for its K steps (between the same two barriers): a straggler shows up here, not only in the maximum
        mine = torch.tensor([dt, float(T * SPS * args.steps), float(local_rank)], dtype=torch.float64, device=cdev)
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
        per_rank = [{"rank": r_, "seconds": float(e_[0]), "frames_per_s": float(e_[1]) / float(e_[0]), "device": int(e_[2])} for r_, e_ in enumerate(every)]
        tt = torch.tensor([dt], dtype=torch.float64, device=cdev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
        fr = torch.tensor([T * SPS * args.steps], dtype=torch.float64, device=cdev)
        dist.all_reduce(fr, op=dist.ReduceOp.SUM)
        total_frames = float(fr.item())
    else:
        total_frames = float(T * SPS * args.steps)

    # end-of-run gather (the only collective of the job): hypotheses + scoring counters -> rank 0
    gathered = None
    if dec is not None and world > 1:
        from voicebridge_amd import dist as vdist
        hyps = col_words
        ids = [int(x) for x in gid]
        counters = vdist.score(truth if truth is not None else hyps, hyps)     # (tones have no transcripts: self-scored)
        totals, merged = vdist.gather_results(ids, hyps, counters, device=cdev)
        if rank == 0:
            gathered = {"utterances": len(merged), "counters": dict(zip(vdist.COUNTER_NAMES, totals.tolist()))}

    if rank == 0:
        value = total_frames / dt
        b, f = gmm.work(T)
        g_ms = float(np.mean(gmm_ms))
        gmm_roof = {"kernel": "gmm_rows_kernel", "bound": "mfma", "achieved": f / (g_ms * 1e-3) / 1e12,
                    "peak": F32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": f / (g_ms * 1e-3) / 1e12 / F32_MFMA_PEAK_TFLOPS,
                    "traffic": None, "kernel_ms": g_ms, "algorithmic_flops": f, "algorithmic_bytes": b,
                    "hbm_GBs": b / (g_ms * 1e-3) / 1e9, "hbm_frac": b / (g_ms * 1e-3) / 1e9 / HBM_PEAK_GBS}
        if gmm_alone_ms is not None:
            gmm_roof["in_pipeline_ms"] = g_ms
            gmm_roof["note"] = ("kernel alone on the chip, HIP events on its stream, best of 3 after the timed region; in_pipeline_ms is its "
                                "duration inside the timed region, where it runs beside two search grids on the CUs they leave free")
            for k_, v_ in (("achieved", f / (gmm_alone_ms * 1e-3) / 1e12), ("kernel_ms", gmm_alone_ms),
                           ("hbm_GBs", b / (gmm_alone_ms * 1e-3) / 1e9)):
                gmm_roof[k_] = v_
            gmm_roof["frac"] = gmm_roof["achieved"] / F32_MFMA_PEAK_TFLOPS
            gmm_roof["hbm_frac"] = gmm_roof["hbm_GBs"] / HBM_PEAK_GBS
        # MFCC: SURVEY.md 8d: 160 new int16 samples in + 13 floats out = 372 B and ~15 kflop per frame.  What binds it is the LDS pipe (the
        # radix-4 exchanges of the FFT, the mel accumulation and the DCT all go through LDS): the counters of profiles/r04_frontend_counters.json
        # (rocprofv3 --pmc, this command with --no-decode) say the LDS arrays are busy 75 % of the CU-cycles and the waves wait for LDS 62 %
        # of theirs, VALU issue 17 %.  So the kernel is priced against the LDS pipe, with the fraction of the HBM roof beside it for the record.
        m_bytes, m_flops = 372.0 * T, 15.0e3 * T
        fe_cnt = {}
        try:
            fe_cnt = json.load(open(os.path.join(ROOT, "profiles", "r04_frontend_counters.json")))
        except Exception:
            pass
        mc = fe_cnt.get("mfcc512_kernel", {})
        mc_ok = all(mc.get(k_) for k_ in ("clock_hz", "cus", "lds_busy_cycles_per_frame"))      # (a counters file without them must not raise)
        extra_mfcc = {"kernel": "mfcc512_kernel", "bound": "lds", "achieved": mc.get("lds_busy_cycles_per_frame"), "unit": "LDS-array busy cycles per frame (all CUs)",
                      "peak": (mfcc_alone_ms * 1e-3 * mc["clock_hz"] * mc["cus"] / T) if mc_ok else None,
                      "frac": (mc["lds_busy_cycles_per_frame"] / (mfcc_alone_ms * 1e-3 * mc["clock_hz"] * mc["cus"] / T)) if mc_ok else None,
                      "traffic": None, "kernel_ms": mfcc_alone_ms,
                      "hbm_GBs": m_bytes / (mfcc_alone_ms * 1e-3) / 1e9, "hbm_frac": m_bytes / (mfcc_alone_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                      "algorithmic_bytes": m_bytes, "algorithmic_flops": m_flops, "gflops": m_flops / (mfcc_alone_ms * 1e-3) / 1e9,
                      "frames_per_s": T / (mfcc_alone_ms * 1e-3), "counters": mc or None,
                      "note": "kernel alone on the chip, HIP events on its stream, best of 3 after the timed region.  frac = the share of this launch's CU-cycles "
                              "(its duration x 2.4 GHz x 256 CUs: a lower bound, the chip clocks lower under load) during which the LDS arrays are busy, from SQ_LDS_IDX_ACTIVE per frame of the committed "
                              "counter pass; by arithmetic intensity (40 flop/B) the kernel would be HBM-bound, but a frame is a 512-point FFT plus three small "
                              "dot-product stages through LDS, so hbm_frac is small by construction"}
        # post-processing: 4 (13 + D) bytes per frame in and out, HBM-bound
        fp_bytes = 4.0 * (13 + args.feat_dim) * T
        extra_featpost = {"kernel": "featpost_kernel", "bound": "hbm", "achieved": fp_bytes / (featpost_alone_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                          "frac": fp_bytes / (featpost_alone_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": None, "kernel_ms": featpost_alone_ms,
                          "algorithmic_bytes": fp_bytes, "frames_per_s": T / (featpost_alone_ms * 1e-3), "counters": fe_cnt.get("featpost_kernel"),
                          "note": "kernel alone on the chip, best of 3 after the timed region: CMVN + deltas (or splice + LDA) + the speaker's transform, 64 frames "
                                  "(+ halo) per workgroup; 1 % of the step.  Far from the HBM roof: a workgroup's 64 frames are 3 KB in and 10 KB out between two "
                                  "barriers, and the waves spend their time waiting for those few loads (VALU issue 17 % of the wave-cycles, "
                                  "profiles/r04_frontend_counters.json)"}
        stage_ms = {"mfcc+featpost": float(np.mean(feat_ms)), "gmm": g_ms}
        roof = gmm_roof
        extra = {}
        if dec is not None:
            d_ms = float(np.mean(dec_ms))
            k_ms = float(np.mean(dec_k_ms))
            stage_ms["decode (begin to end on the host)"] = d_ms
            stage_ms["decode: search kernel only"] = k_ms
            stage_ms["decode: lattice D2H, host ordering + best path"] = float(np.mean(dec_h_ms))
            if depth > 1:
                stage_ms["note"] = ("per pipeline item (a step = %d slice(s) of %d utterances); %d items in flight on %d streams: stage times of "
                                    "consecutive items overlap and are not additive; a search kernel's duration includes the time its "
                                    "workgroups wait for CUs still held by the grids launched before it" % (S, per, depth, depth))
            if truth is not None:
                from voicebridge_amd import dist as vdist
                if gathered is not None:
                    wc = dict(gathered["counters"])
                else:
                    wc = dict(zip(vdist.COUNTER_NAMES, [int(x) for x in vdist.score(truth, col_words)]))
                extra["wer_vs_transcripts"] = {
                    "wer_percent": 100.0 * wc["word_errs"] / max(wc["num_words"], 1), "words": wc["num_words"], "ins": wc["ins"], "del": wc["del"],
                    "sub": wc["sub"], "utterances": wc["num_sent"], "words_per_utterance": wc["num_words"] / max(wc["num_sent"], 1),
                    "what": "best path of the raw lattice (acoustic scale 1/12, no LM-weight search) of the last timed step's utterances against "
                            "the word sequences the waveforms were synthesised from; the acoustic model was fit to other utterances by other speakers"
                            + (", %d pdfs unseen in the training data" % n_unseen_pdfs if n_unseen_pdfs else "")}
            extra["search_grid_last_batch"] = grid_stats
            extra["search_retries_last_batch"] = col_stats
            # algorithmic bytes of the search from the counters of one step:
            # N_tok*(20+16) + N_arc*(16+4) + N_new*16 + N_link*24; one launch searches one slice, i.e. 1/S of them
            cs = col_cnt.sum(0)
            c = dict(tok=cs[0], arc=cs[1], new=cs[2], link=cs[3], maxb=int(cs[4]), minb=int(cs[5]), lat_tok=cs[6], lat_link=cs[7])
            dbytes = (c["tok"] * 36.0 + c["arc"] * 20.0 + c["new"] * 16.0 + c["link"] * 24.0) / S
            dec_roof = {"kernel": "decode_kernel", "bound": "hbm", "achieved": dbytes / (k_ms * 1e-3) / 1e9,
                        "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": dbytes / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                        "traffic": None, "kernel_ms": k_ms, "algorithmic_bytes": dbytes, "launches_per_step": S,
                        "note": "random-access (token/arc/link) bytes of SURVEY.md 8d over the kernel's own duration (HIP events "
                                "on its stream); the search is bound by dependent-access latency, not by bandwidth (DESIGN.md)"}
            extra["search"] = {"tokens_expanded_per_frame": c["tok"] / T, "arcs_per_frame": c["arc"] / T,
                               "tokens_created_per_frame": c["new"] / T, "links_per_frame": c["link"] / T,
                               "frames_maxactive_bound": c["maxb"], "frames_minactive_bound": c["minb"],
                               "lattice_tokens_per_frame": c["lat_tok"] / T, "lattice_arcs_per_frame": c["lat_link"] / T,
                               "arc_expansions_per_s": c["arc"] / S / (k_ms * 1e-3)}
            extra["lattice_stage_host"] = determinize_measurement(args, dec, per, slices[(S - 1) % S].r1 - slices[(S - 1) % S].r0)
            if det_pool is not None and det_totals is not None:
                nb = args.steps * args.batch * SPS
                tt = det_totals
                extra["lattice_stage_in_step"] = {
                    "what": "the headline `value` includes it: every timed batch's lattices determinized (Connect, phone-level + word-level pruned "
                            "determinization, un-scaling: what gmm-latgen-faster does to every utterance) by a pool of host threads while the GPU "
                            "searches the next batch; the timed region ends when the last batch is determinized",
                    "threads": det_pool.threads, "utterances": nb, "worker_seconds": float(tt[0]), "worker_ms_per_utterance": 1e3 * float(tt[0]) / max(nb, 1),
                    "copy_out_seconds_on_the_step_thread": det_submit_s, "raw_arcs_per_utt": float(tt[2]) / max(nb, 1),
                    "det_arcs_per_utt": float(tt[4]) / max(nb, 1), "stopped_before_beam": int(tt[5])}
                if raw_dt is not None:
                    # does the host stage keep up with the GPU on this rank's thread budget?  (value = all ranks' frames / the slowest rank's
                    # time, the raw rate below is this rank's: compared per rank)
                    mine_ = T * SPS * args.steps / per_rank[0]["seconds"]
                    extra["lattice_stage_in_step"]["keeps_up"] = bool(mine_ >= 0.97 * (T * args.det_steps / raw_dt))
                    extra["lattice_stage_in_step"]["rate_with_over_without"] = mine_ / (T * args.det_steps / raw_dt)
                    extra["raw_lattice_rate"] = {"value": T * args.det_steps / raw_dt, "unit": "frames/s", "steps": args.det_steps,
                                                 "ms_per_step": 1e3 * raw_dt / args.det_steps,
                                                 "what": "a second timed region without the host lattice stage: the GPU path alone, raw state-level "
                                                         "lattices delivered to the host (what rounds 1-2 reported as `value`)"}
            # beside the step, on the scores of one batch: the 1-best search (FasterDecoder: gmm-decode-faster, and gmm-align-compiled's search;
            # K/gmmbin/gmm-decode-faster.cc:131-166) and the lattice search under the other admission rule (vbh_decode_cfg::admission)
            try:
                sl0 = slices[0]
                ll0 = ring[0][2][:n_rows0]
                fd = api.FasterDecoder(ctx_full, graph, per, Tmax, beam=args.beam, max_active=args.max_active, min_active=20, acoustic_scale=0.083333)
                best_f = None
                for _ in range(3):
                    t0_ = time.perf_counter()
                    fd.decode(ll0, sl0.foff, t2p)
                    dtf = time.perf_counter() - t0_
                    best_f = dtf if best_f is None else min(best_f, dtf)
                r0_ = fd.result(0)
                fd.close()
                extra["faster_rate"] = {"value": n_rows0 / best_f, "unit": "frames/s", "utterances": per, "seconds_per_call": best_f,
                                        "what": "vbh_faster_decode_batch on one batch's scores (the same utterances, graph, beam and max-active as the lattice "
                                                "search; min-active 20): kernel + back traces + results on the host, best of 3 calls",
                                        "words_utt0": len(r0_["words"]), "reached_final_utt0": int(r0_["reached_final"])}
            except Exception as e:
                extra["faster_rate"] = {"error": str(e)}
            try:
                dx = api.LatticeDecoder(ctx_full, graph, per, Tmax, beam=args.beam, max_active=args.max_active, min_active=200, lattice_beam=args.lattice_beam,
                                        acoustic_scale=0.083333, share_slots_with=dec, admission="exact")
                kx = None
                for _ in range(2):
                    dx.decode(ring[0][2][:n_rows0], slices[0].foff, t2p)
                    kx = dx.times()["kernel_ms"] if kx is None else min(kx, dx.times()["kernel_ms"])
                cx = np.zeros(4)
                for u in range(per):
                    r_ = dx.result(u, arrays=False)
                    cx += [r_["n_tokens_expanded"], r_["n_arcs_expanded"], r_["n_tokens_created"], r_["n_links_created"]]
                dx.close()
                extra["search_exact_admission"] = {"kernel_ms": kx, "frames_per_s_search_only": n_rows0 / (kx * 1e-3), "tokens_created_per_frame": cx[2] / n_rows0,
                                                   "links_per_frame": cx[3] / n_rows0,
                                                   "what": "the search kernel alone on one batch with admission = exact (tot <= the frame's final next_cutoff, rounds 1-3): "
                                                           "fewer tokens than the reference's rule admits; the step runs admission = ordered (DESIGN.md 5)"}
            except Exception as e:
                extra["search_exact_admission"] = {"error": str(e)}
            try:
                extra["gmm_on_demand"] = ondemand_gmm_measurement(args, ctx_full, gmm_m, model, ring[0][1][:n_rows0], n_rows0)
            except Exception as e:
                extra["gmm_on_demand"] = {"error": str(e)}
            try:
                extra["roofline_fmllr_stats"] = fmllr_measurement(args, ctx_full, gmm_m, ring[0][1][:n_rows0], slices[0].foff, slices[0].u2s,
                                                                  int(slices[0].u2s.max()) + 1, n_rows0)
            except Exception as e:        # a measurement beside the step, never a reason to lose the bench line
                extra["roofline_fmllr_stats"] = {"error": str(e)}
            if d_ms >= g_ms:
                roof, extra["roofline_gmm"] = dec_roof, gmm_roof
            else:
                extra["roofline_decode"] = dec_roof
        out = {
            "metric": "decode frames/sec/node (RTF = 100/value), DELTA+SAT sizes",
            "value": value, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": args.scaling,
            "vs_baseline": None, "dtype": "f32",
            "data": ("synthetic speech: word sequences of the graph's lexicon, one spectral envelope per pdf (depth %g), model fit to %d other utterances"
                     % (args.depth, args.train_utts)) if lex is not None else "synthetic",
            "rtf": 100.0 / value,
            "config": {"workload": ("config2: MFCC + CMVN/deltas/fMLLR + all-pdf GMM scoring (search not in the step)"
                                    if dec is None else
                                    ("config5: a fixed corpus of %d synthetic %g s utterances (%d asked for), speaker-sharded over %d rank(s), %d batch(es) of %d per rank and step; "
                                     % (args.corpus_utts, args.seconds, args.utterances, world, SPS, args.batch) if args.scaling == "strong" else "config3: ") +
                                    "DELTA+SAT full GPU decode: MFCC, CMVN+deltas+fMLLR, GMM scoring, lattice "
                                    "search beam %g lattice-beam %g max-active %d over a synthetic %d-state / %d-arc HCLG"
                                    % (args.beam, args.lattice_beam, args.max_active, graph.num_states, graph.num_arcs)),
                       "utterances_per_step_per_gpu": args.batch * SPS, "utterances_per_batch": args.batch, "batches_per_step_per_gpu": SPS,
                       "utterances_per_step_job": args.batch * SPS * world, "utterance_seconds": args.seconds,
                       "frames_per_step_per_gpu": T * SPS, "host_threads_per_rank": host_threads(args), "pdfs": gmm.num_pdfs, "gaussians": gmm.num_gauss, "dim": gmm.dim, "features": args.feat,
                       "batches_in_flight": depth, "slices_per_step": S, "search_cus": args.search_cus, "front_end_cus": args.fe_cus,
                       "lattices": "determinized on the host inside the timed step (lattice_stage_in_step), as gmm-latgen-faster writes them; raw_lattice_rate: the GPU path alone",
                       "batch_sets": n_sets,
                       "sharding": "utterances over ranks, no data-path collective"},
            "roofline": roof, "stage_ms": stage_ms,
            "ranks": {"world_size": world, "backend": backend if world > 1 else "none (one process)", "visible_devices": torch.cuda.device_count(),
                      "per_rank": per_rank, "timing": "every rank times its own K steps between the same two barriers; `value` = frames of all ranks / the slowest rank's time"},
        }
        out.update(extra)
        out["roofline_mfcc"] = extra_mfcc
        out["roofline_featpost"] = extra_featpost
        # HBM traffic of the dominant kernels from the committed rocprofv3 --pmc passes of this
        # same command (profiles/): only attached when the batch size matches that run
        try:
            import glob
            tr_path = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")))[-1]        # the latest round's passes
            tr = json.load(open(tr_path))
            if tr.get("batch") == args.batch and world == 1:
                for ro in [out["roofline"]] + [out[k] for k in ("roofline_gmm", "roofline_decode", "roofline_mfcc") if k in out]:
                    # (front-end kernels also run on small training batches in that command: their largest launch is the full batch)
                    b_ = tr.get("bytes_of_largest_launch", {}).get(ro["kernel"]) if ro["kernel"] != "decode_kernel" else None
                    b_ = b_ or tr["bytes_per_launch"].get(ro["kernel"])
                    if b_:
                        ro["traffic"] = b_
                        ro["traffic_source"] = tr.get("source", "profiles/")
                    if b_ and ro["kernel"] == "decode_kernel":
                        # A roof that fits a search made of 16-byte records gathered at random (VERDICT r4 "next" 2a): what the chip delivers to a kernel
                        # that does nothing else -- tools/micro/gather16_peak.hip, 256 workgroups x 16 waves on a table of the packed graph's size:
                        # 57 G random 64-byte lines per second chip-wide whatever the loads in flight per lane (223 M per CU; a CU alone reaches 850 M: the
                        # limit is the fabric's, not the CU's), i.e. 0.91 TB/s of useful 16-byte records = 11 % of the 8 TB/s this kernel's `frac` is
                        # priced against.  gather_frac = the 64-byte lines the kernel moves through the fabric per second (counted traffic / 64 B over its
                        # live duration) over that ceiling: an UPPER bound on how close to the random-access roof it runs, since half of its traffic
                        # (tokens, links, candidates) streams, which the fabric serves at up to 98 G lines/s (6.29 TB/s measured copy)
                        try:
                            gl_ = [json.loads(x) for x in open(os.path.join(ROOT, "profiles", "r05_gather16_126mb.jsonl")) if x.startswith("{")]
                            ceil_ = max(g_["records_per_s"] for g_ in gl_ if g_["what"].startswith("independent") and g_.get("cus", 256) >= 256 and " 1 lanes per" in g_["what"])
                            lines_ = b_ / 64.0 / (ro["kernel_ms"] * 1e-3)
                            ro["gather_frac"] = lines_ / ceil_
                            ro["gather"] = {"lines64_per_s": lines_, "ceiling_lines64_per_s": ceil_, "ceiling_useful_GBs_of_16_byte_records": ceil_ * 16.0 / 1e9,
                                            "ceiling_source": "profiles/r05_gather16_126mb.jsonl (tools/micro/gather16_peak.hip: independent random 16-byte loads, 126 MB table, all CUs)",
                                            "streaming_ceiling_lines64_per_s": 6.29e12 / 64.0,
                                            "note": "counted FETCH_SIZE + WRITE_SIZE of the committed profile of this command / 64 B, over this run's kernel time"}
                        except Exception:
                            pass
        except Exception:
            pass
        if gathered is not None:
            out["gather"] = gathered
        if not args.no_cpu_baseline and world == 1:      # (the reported CPU baseline belongs to the one-GPU line; the N > 1 runs do not repeat it)
            jobs = cpu_sample(args, utt2spk, n_spk)
            n_smp = int(round(args.seconds * 16000))
            waves = {u: d_wave[u * n_smp:(u + 1) * n_smp].cpu().numpy() for j_ in jobs for u in j_}
            ref_cb, cpu_words, cpu_like = None, {}, {}
            try:
                ref_cb = cpu_baseline_reference(args, jobs, waves, utt2spk, xforms, lda, model, decode_setup)
            except Exception as e:
                print("cpu_baseline (reference programs) failed: %s" % e, file=sys.stderr, flush=True)
            if ref_cb is not None:
                out["cpu_baseline"], out["cpu_baseline_1proc"], cpu_words, cpu_like = ref_cb
                if os.environ.get("VBH_BENCH_PORT_BASELINE"):
                    out["cpu_baseline_port"] = cpu_baseline(args, jobs[0], waves, utt2spk, xforms, lda, model, decode_setup)[0]
            else:
                out["cpu_baseline"], cpu_words = cpu_baseline(args, jobs[0], waves, utt2spk, xforms, lda, model, decode_setup)
            if dec is not None and cpu_words:
                # the utterances the CPU leg decoded anyway: are the GPU step's hypotheses the same words?
                same = sum(1 for u, w_ in cpu_words.items() if list(col_words[u]) == list(w_))
                out["parity_words_equal"] = "%d/%d" % (same, len(cpu_words))
                # where the words differ: the two best paths' log-likelihood per frame (a difference at the level of the front end's float
                # tolerance means two hypotheses of nearly equal score, which MFCC differences of 1e-3 are enough to swap)
                diff = [{"utt": int(u), "reference_words": list(w_), "gpu_words": [int(x) for x in col_words[u]],
                         "reference_like_per_frame": cpu_like.get(u), "gpu_like_per_frame": float(col_like[u])}
                        for u, w_ in cpu_words.items() if list(col_words[u]) != list(w_)]
                if diff:
                    out["parity_words_differ"] = [{k_: (v_ if "words" not in k_ else "%d words" % len(v_)) for k_, v_ in d_.items()} for d_ in diff]
                # which of the differing utterances the GPU's own best path scores better / worse on (like_per_frame: the path's own score)
                gb_ = sum(1 for d_ in diff if d_["reference_like_per_frame"] is not None and d_["gpu_like_per_frame"] > d_["reference_like_per_frame"])
                gw_ = sum(1 for d_ in diff if d_["reference_like_per_frame"] is not None and d_["gpu_like_per_frame"] < d_["reference_like_per_frame"])
                out["parity_paths_where_words_differ"] = {"gpu_path_scores_better": gb_, "gpu_path_scores_worse": gw_, "utterances": len(diff)}
                out["parity_note"] = ("best-path words of the GPU step against the %s's for the utterances of the CPU sample (same waveforms, model, graph, "
                                      "transforms).  A search with max-active deciding the cutoff in every sixth frame is sensitive to WHICH arcs above the "
                                      "final next_cutoff ProcessEmitting lets through, i.e. to its visiting order: the reference against itself under another "
                                      "--hash-ratio changes the words of 6-7 of 512 such utterances (tests/golden/speech_e2e512.npz).  The GPU search applies the "
                                      "reference's rule in a canonical order (vbh_decode_cfg::admission = ordered), so single utterances differ at about that rate "
                                      "plus the front end's float tolerance (MFCC 1e-3 abs, scores 1e-4 rel).  What THIS sample shows is in "
                                      "parity_wer_on_the_cpu_sample (error counts of both against the transcripts, utterances with more / fewer errors than the "
                                      "reference, the bootstrap probability that the GPU improves on the reference computed on these utterances) and "
                                      "parity_paths_where_words_differ (whose best path scores higher where the words differ: a lower GPU score is a search error "
                                      "of the GPU's); the 512-utterance measurement (tests/test_gpu_e2e512.py, DESIGN.md 5): WER within 0.03 abs of the "
                                      "reference's, bootstrap P(GPU improves on the reference) 0.34"
                                      % ("reference programs" if ref_cb is not None else "oracle"))
                if truth is not None:
                    from voicebridge_amd import dist as vdist
                    us_ = sorted(cpu_words)
                    wr_ = vdist.score([truth[u] for u in us_], [cpu_words[u] for u in us_])
                    wg_ = vdist.score([truth[u] for u in us_], [col_words[u] for u in us_])
                    # per-utterance errors and the two-system bootstrap of compute-wer-bootci (K/bin/compute-wer-bootci.cc:137-155: utterances
                    # resampled with replacement, "improves" = strictly fewer errors in the resample), on THIS sample
                    er_ = np.array([sum(vdist.edit_counts(truth[u], cpu_words[u])) for u in us_], np.int64)
                    eg_ = np.array([sum(vdist.edit_counts(truth[u], col_words[u])) for u in us_], np.int64)
                    rng_ = np.random.default_rng(0)
                    ix_ = rng_.integers(0, len(us_), (10000, len(us_)))
                    dd_ = (er_ - eg_)[ix_].sum(1)
                    out["parity_wer_on_the_cpu_sample"] = {"reference_percent": 100.0 * wr_[0] / max(wr_[1], 1), "gpu_percent": 100.0 * wg_[0] / max(wg_[1], 1),
                                                           "words": int(wr_[1]), "utterances": len(us_),
                                                           "reference_errors": int(er_.sum()), "gpu_errors": int(eg_.sum()),
                                                           "utterances_gpu_more_errors": int((eg_ > er_).sum()), "utterances_gpu_fewer_errors": int((eg_ < er_).sum()),
                                                           "bootstrap_p_gpu_improves_on_reference": float((dd_ > 0).mean()),
                                                           "bootstrap_p_reference_improves_on_gpu": float((dd_ < 0).mean()),
                                                           "bootstrap": "10000 resamples of the sample's utterances, strictly fewer word errors (compute-wer-bootci's two-system comparison)"}
    if tl_dump and os.environ.get("VBH_BENCH_TIMELINE"):
        np.save(os.environ["VBH_BENCH_TIMELINE"], np.stack(tl_dump))
    if det_pool is not None:
        det_pool.close()
    # tear down in order: decoders before their contexts (and before the interpreter's own finalisation)
    for L in lanes:
        if L.dec is not None:
            L.dec.close()
    for L in lanes:
        if L.ctx is not ctx:
            L.ctx.close()
    if rank == 0:
        if dec_enabled and world == 1 and not args.no_step_api:
            # The reference's own entry points on a data directory (MakeMfcc, ComputeCmvnStats, Decode, DecodeFmllr: wav files in, archives /
            # lattices / transforms / WER files out, per-job set-up and file I/O included), in a child process once this one has given its
            # GPU memory back: what an application that calls the step API sees, beside the kernels' own rate above
            try:
                ring = gmm = raw = feats = ll = d_wave = None      # (the tensors of this process: the child needs the memory)
                slices[:] = []
                torch.cuda.empty_cache()
                import subprocess
                r_ = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "bench_steps.py"), "--utts", str(args.step_api_utts), "--nj", "4", "--fmllr",
                                     "--job-threads", str(args.step_api_threads), "--words", str(args.words), "--depth", str(args.depth),
                                     "--train-utts", str(args.train_utts)], capture_output=True, text=True, timeout=600)
                st_ = json.loads([l_ for l_ in r_.stdout.splitlines() if l_.startswith("{")][-1])
                fr_ = float(st_["frames"])
                k_dec = [k_ for k_ in st_ if k_.startswith("Decode_s")][0]
                k_fm = [k_ for k_ in st_ if k_.startswith("DecodeFmllr_s")][0]
                out["step_api"] = {"what": "tools/bench_steps.py: %d utterances of 30 s in a data directory, nj = 4 jobs on %d host thread(s), one GPU; seconds per step "
                                           "of the reference's API and frames/s of the two decoding steps; Decode_s / DecodeFmllr_s = the SECOND call of the step in the process (graph, models and "
                                           "search workspaces exist: the rate of a process that decodes set after set), *_first_call_s = the first (loads the graph, makes the workspaces)" % (args.step_api_utts, args.step_api_threads),
                                   "job_threads": args.step_api_threads, "MakeMfcc_s": st_["MakeMfcc_s"], "ComputeCmvnStats_s": st_["ComputeCmvnStats_s"], "Decode_s": st_[k_dec],
                                   "DecodeFmllr_s": st_[k_fm], "Decode_first_call_s": st_.get("Decode_first_call_s"), "DecodeFmllr_first_call_s": st_.get("DecodeFmllr_first_call_s"), "Decode_step_frames_per_s": fr_ / st_[k_dec], "DecodeFmllr_step_frames_per_s": fr_ / st_[k_fm],
                                   "Decode_best_wer": st_.get("Decode_best_wer"), "DecodeFmllr_best_wer": st_.get("DecodeFmllr_best_wer")}
            except Exception as e:
                out["step_api"] = {"error": str(e)[:300]}
        print(json.dumps(out), flush=True)
    # device objects go before their contexts, contexts before the interpreter's own finalisation (no exit handler has to guess an order)
    ring = raw = feats = ll = d_wave = None
    slices[:] = []
    for obj in (locals().get("gmm_m"), locals().get("mf_m"), gmm, mf, post, locals().get("graph")):
        try:
            if obj is not None:
                obj.close()
        except Exception:
            pass
    torch.cuda.synchronize(dev)
    for c_ in (ctx, ctx_full):
        try:
            c_.close()
        except Exception:
            pass
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
    # A normal exit: main() has closed its decoders, contexts and device objects in order.  (Round 2 left here with os._exit(0) after one
    # run of an earlier build had ended with a GPU memory fault AFTER its last output line; with the explicit tear-down at the end of
    # main() the soft exit ran clean in every run since -- VBH_BENCH_HARD_EXIT=1 brings the old behaviour back for comparison.)
    sys.stdout.flush()
    sys.stderr.flush()
    if os.environ.get("VBH_BENCH_HARD_EXIT"):
        os._exit(0)
