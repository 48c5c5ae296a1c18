"""BASELINE config 1 at its stated size: the Yes-No example's stand-in -- 60 utterances of
6 s at 16 kHz, a monophone model of P = 11 pdfs (SIL: 5 states, YES and NO: 3 states each, as the yesno recipe's topology), a 2-word
ConstFst -- through the step API as an application calls it: vb_MakeMfcc -> vb_ComputeCmvnStatsDir -> vb_Decode (nj = 4, scoring on),
VB/scr/steps/{make_mfcc,compute_cmvn_stats,decode_gmm}.cpp.  The words are easy to tell apart, so the expected WER is below 1 % (the reference
quotes 2 % on the real recordings, README.md:48); what is checked beside it is every file the reference's Decode leaves behind
(VB/scr/steps/decode_gmm.cpp:300-390, steps/scoring/score_kaldi_wer.cpp): num_jobs, lat.JOB, w.JOB, transcriptionJOB.trn,
log/decode.JOB.log, wer_LMWT_WIP, scoring_kaldi/{best_wer, penalty_*/LMWT.txt, wer_details/{lmwt, wip, per_utt, per_spk, ops, wer_bootci}}."""
import ctypes as C
import os
import re
import wave as wavmod

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

N_UTT, N_SPK, SECONDS = 60, 15, 6.0
HS = [5, 3, 3]                      # SIL, YES, NO
SIL, YES, NO = 0, 1, 2
P = sum(HS)
FIRST_PDF = np.concatenate([[0], np.cumsum(HS)])


def yesno_graph():
    """loop state 0 (start, final); from it one input-epsilon arc per phone carrying the word (YES = 1, NO = 2, SIL none) into the phone's
    left-to-right HMM (self-loop tid 2p+1, forward tid 2p+2 on pdf p), whose last forward arc returns to the loop state"""
    from voicebridge_amd import synth
    arcs, off, final = [], [0], []
    self_w, fwd_w = -np.log(0.6), -np.log(0.4)
    first_state = {}
    s = 1
    for ph, h in enumerate(HS):
        first_state[ph] = s
        s += h
    n_states = s
    # state 0
    for ph, cost in ((SIL, 0.5), (YES, 4.0), (NO, 4.0)):        # (a word costs more than silence: the graph's own word insertion penalty)
        arcs.append((0, {SIL: 0, YES: 1, NO: 2}[ph], cost, first_state[ph]))
    off.append(len(arcs)); final.append(0.0)
    for ph, h in enumerate(HS):
        for k in range(h):
            p = int(FIRST_PDF[ph]) + k
            arcs.append((2 * p + 1, 0, self_w, first_state[ph] + k))
            arcs.append((2 * p + 2, 0, fwd_w, first_state[ph] + k + 1 if k + 1 < h else 0))
            off.append(len(arcs)); final.append(np.inf)
    a = np.zeros(len(arcs), synth.ARC_DTYPE)
    for i, (il, ol, w, ns) in enumerate(arcs):
        a[i] = (il, ol, w, ns)
    assert len(final) == n_states
    return np.array(final, np.float32), np.array(off, np.int64), a


def utterances(ids, rng_seed=7):
    """hop labels [n x 600] and transcripts: SIL, then eight words with a short SIL between them, SIL to the end"""
    from voicebridge_amd import synth_speech as ss
    n_hops = int(SECONDS * 16000) // ss.HOP
    hop = np.zeros((len(ids), n_hops), np.int64)
    words = []

    def phone(seq, ph, dur):
        h = HS[ph]
        cuts = np.linspace(0, dur, h + 1).round().astype(int)
        for k in range(h):
            seq.extend([int(FIRST_PDF[ph]) + k] * max(1, int(cuts[k + 1] - cuts[k])))
    for i, u in enumerate(ids):
        rng = np.random.default_rng(rng_seed + 1000 * int(u))
        seq, ws = [], []
        phone(seq, SIL, int(rng.integers(25, 45)))
        for _ in range(8):
            w = YES if rng.random() < 0.5 else NO
            ws.append(1 if w == YES else 2)
            phone(seq, w, int(rng.integers(30, 46)))
            phone(seq, SIL, int(rng.integers(15, 26)))
        while len(seq) < n_hops:
            phone(seq, SIL, min(40, n_hops - len(seq)) if n_hops - len(seq) >= 5 else 5)
        hop[i] = seq[:n_hops]
        # (the last phone may be cut by the end of the utterance: only SIL is, the eight words fit in 600 hops by construction)
        words.append(ws)
    return hop, words


def test_config1_yesno_through_the_step_api(tmp_path):
    from voicebridge_amd import api, build, refio, synth_speech as ss
    host = C.CDLL(build.HOST_LIB)
    host.vb_steps_last_error.restype = C.c_char_p
    host.vb_Decode.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_int, C.c_float, C.c_int, C.c_double, C.c_double, C.c_int]
    d = tmp_path
    gains = ss.pdf_gains(P, float(os.environ.get("VB_C1_DEPTH", "0.9")), dtype=np.float64)
    # ---- the acoustic model: per-pdf mean / variance of labelled training frames through the product's front end, 4 Gaussians per pdf
    ctx = api.Context(0)
    mf = api.Mfcc(ctx)
    post = api.FeaturePipeline(ctx, cmvn=True, mode="delta")
    fit = ss.GmmFit(P, 39, ctx.device)
    tr_ids = 10000 + np.arange(16)
    tr_spk = 500 + np.arange(16) // 4
    hop_tr, _ = utterances(tr_ids)
    waves_tr = ss.waves_numpy(hop_tr, gains, tr_ids, spk_gain=ss.speaker_gains(tr_spk, dtype=np.float64))
    raw, foff = mf.compute([w for w in waves_tr])
    u2s = (np.arange(16) // 4).astype(np.int32)
    st = post.cmvn_stats(raw, foff, u2s, 4)
    f = post.apply(raw, foff, utt2spk=u2s, cmvn_stats=st)
    ctx.sync()
    for u in range(16):
        fu = f[int(foff[u]):int(foff[u + 1])]
        fit.add(fu, ss.frame_labels(hop_tr[u:u + 1], fu.shape[0]))
    (offs, w, miv, iv), unseen = fit.model(4 * P)
    assert unseen == 0 and int(np.diff(offs).max()) <= 8          # (Gp <= 8)
    ctx.close()
    # ---- model and graph directories
    exp = d / "exp" / "mono0a"
    graph = exp / "graph"
    graph.mkdir(parents=True)
    refio.write_mdl(str(exp / "final.mdl"), 3, HS, offs, w, miv, iv)
    final, off, arcs = yesno_graph()
    refio.write_const_fst(str(graph / "HCLG.fst"), final, off, arcs, 0)
    (graph / "words.txt").write_text("<eps> 0\nYES 1\nNO 2\n")
    # ---- the data directory: 60 utterances by 15 speakers
    data = d / "data" / "test_yesno"
    data.mkdir(parents=True)
    ids = np.arange(N_UTT)
    spk = ids // (N_UTT // N_SPK)
    hop, truth = utterances(ids)
    waves = ss.waves_numpy(hop, gains, ids, spk_gain=ss.speaker_gains(spk, dtype=np.float64))
    names = ["spk%02d_utt%02d" % (s, u) for u, s in zip(ids, spk)]
    with open(data / "wav.scp", "w") as fw:
        for nm, wv in zip(names, waves):
            pth = data / (nm + ".wav")
            with wavmod.open(str(pth), "wb") as wf:
                wf.setnchannels(1); wf.setsampwidth(2); wf.setframerate(16000)
                wf.writeframes(np.asarray(wv, "<i2").tobytes())
            fw.write("%s %s\n" % (nm, pth))
    (data / "utt2spk").write_text("".join("%s spk%02d\n" % (nm, s) for nm, s in zip(names, spk)))
    (data / "spk2utt").write_text("".join("spk%02d %s\n" % (s, " ".join(nm for nm, s2 in zip(names, spk) if s2 == s)) for s in range(N_SPK)))
    (data / "text").write_text("".join("%s %s\n" % (nm, " ".join("YES" if x == 1 else "NO" for x in ws)) for nm, ws in zip(names, truth)))
    conf = d / "mfcc.conf"
    conf.write_text("--use-energy=false\n--dither=0\n")
    # ---- the steps
    nj = 4
    assert host.vb_MakeMfcc(str(data).encode(), str(conf).encode(), nj, 1, 1) == 0, host.vb_steps_last_error()
    assert [l.split()[0] for l in (data / "feats.scp").read_text().split("\n")[:-1]] == names
    nf = dict(l.split() for l in (data / "utt2num_frames").read_text().split("\n")[:-1])
    assert all(int(nf[nm]) == 598 for nm in names)                    # 6 s: 1 + (96000 - 400) // 160 frames
    for j in range(1, nj + 1):
        assert (data / "mfcc" / ("raw_mfcc_test_yesno.%d.ark" % j)).stat().st_size > 0 and (data / "log" / ("make_mfcc_test_yesno.%d.log" % j)).exists()
    assert host.vb_ComputeCmvnStatsDir(str(data).encode()) == 0, host.vb_steps_last_error()
    assert len((data / "cmvn.scp").read_text().split("\n")[:-1]) == N_SPK
    dec = exp / "decode_test_yesno"
    rc = host.vb_Decode(str(graph).encode(), str(data).encode(), str(dec).encode(), b"", nj, 0.083333, 7000, 13.0, 6.0, 0)
    logs = "".join((lg.read_text()[-1500:] for lg in sorted((dec / "log").glob("decode.*.log")))) if (dec / "log").exists() else ""
    assert rc == 0, (host.vb_steps_last_error(), logs)
    # ---- what Decode leaves behind (decode_gmm.cpp:300-390)
    assert (dec / "num_jobs").read_text().strip() == str(nj)
    hyp, n_done = {}, 0
    for j in range(1, nj + 1):
        assert (dec / ("lat.%d" % j)).stat().st_size > 0
        log = (dec / "log" / ("decode.%d.log" % j)).read_text()
        assert "Overall log-likelihood" in log and ", failed for 0" in log
        n_done += int(re.search(r"Done (\d+) utterances, failed for 0", log).group(1))
        for line in (dec / ("transcription%d.trn" % j)).read_text().split("\n"):
            if line.strip():
                hyp[line.split()[0]] = line.split()[1:]
        assert (dec / ("w.%d" % j)).stat().st_size > 0
        assert not (data / ("split%d" % nj) / str(j) / "apply_cmvn.temp").exists()      # (no temp archives: the features stay on the device)
    assert sorted(hyp) == sorted(names) and n_done == N_UTT
    # ---- scoring (score_kaldi_wer.cpp): one WER file per (LMWT, penalty), the best one recorded, the per-utterance details
    wers = {}
    for wip in ("0.0", "0.5", "1.0"):
        for lmwt in range(7, 18):
            fl = (dec / ("wer_%d_%s" % (lmwt, wip))).read_text().split()      # WER errors words ins del sub SER sentence-errors sentences
            assert len(fl) == 9 and int(fl[8]) == N_UTT and int(fl[2]) == 8 * N_UTT
            wers[(lmwt, wip)] = (float(fl[6]), float(fl[0]))
            assert (dec / "scoring_kaldi" / ("penalty_" + wip) / ("%d.txt" % lmwt)).exists()
    best = (dec / "scoring_kaldi" / "best_wer").read_text().split()
    assert (float(best[6]), float(best[0])) == min(wers.values())             # (smallest %SER, then smallest %WER: VB/scr/utils/best_wer.cpp:25-71)
    det = dec / "scoring_kaldi" / "wer_details"
    for fn in ("lmwt", "wip", "per_utt", "per_spk", "ops", "wer_bootci"):
        assert (det / fn).stat().st_size > 0, fn
    print("\nconfig 1 stand-in (60 x 6 s, P = 11, 2 words): best WER %.2f %% [%s], raw best path WER over all utterances: %d / %d words wrong"
          % (float(best[0]), " ".join(best[1:6]), sum(int(hyp[nm] != ["YES" if x == 1 else "NO" for x in ws]) for nm, ws in zip(names, truth)), N_UTT))
    # two easy words: next to no errors (the reference quotes 2 % on the real recordings; this synthetic speech leaves a couple of
    # insertions at phone boundaries in 480 words), and the transcripts of the step are the words of the best scoring
    assert float(best[0]) <= 1.0
    assert sum(int(hyp[nm] != ["YES" if x == 1 else "NO" for x in ws]) for nm, ws in zip(names, truth)) <= 5
    host.vb_release_device_cache()
