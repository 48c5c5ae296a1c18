"""GPU: the step API (voicebridge_amd/host/voicebridge_steps.cc; seam B1 of SURVEY.md 8b) on a data
directory: MakeMfcc -> ComputeCmvnStats -> Decode, checked against the bin functions run by hand and
against the reference's own programs where they are present (oracle/_ref/bin)."""
import ctypes as C
import os

import numpy as np
import pytest

from tests.test_gpu_binfuncs import REFBIN, call, have_ref, ref, write_wavs
from tests.test_host_io import read_fm_ark
from tests.util import load

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def host():
    from voicebridge_amd import build
    lib = C.CDLL(build.HOST_LIB)
    lib.vb_steps_last_error.restype = C.c_char_p
    lib.vb_Decode.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_int, C.c_float, C.c_int, C.c_double, C.c_double, C.c_int]
    lib.vb_DecodeFmllr.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_int, C.c_float, C.c_int, C.c_double, C.c_double, C.c_double, C.c_char_p,
                                   C.c_int, C.c_char_p, C.c_int, C.c_int]
    lib.vb_DecodeEx.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_int, C.c_float, C.c_int, C.c_double, C.c_double,
                                C.c_int, C.c_char_p, C.c_int, C.c_int]
    return lib


def make_datadir(d, n_utt=6, n_spk=3, seconds=None):
    from voicebridge_amd import synth
    data = d / "data" / "test"
    data.mkdir(parents=True)
    waves = [synth.wave(100 + i, seconds=(0.6 + 0.15 * i) if seconds is None else seconds + 0.1 * i) for i in range(n_utt)]
    write_wavs(data, waves)
    utts = ["utt%02d" % i for i in range(n_utt)]
    spk = ["spk%d" % (i * n_spk // n_utt) for i in range(n_utt)]
    (data / "utt2spk").write_text("".join("%s %s\n" % (u, s) for u, s in zip(utts, spk)))
    s2u = {}
    for u, s in zip(utts, spk):
        s2u.setdefault(s, []).append(u)
    (data / "spk2utt").write_text("".join("%s %s\n" % (s, " ".join(us)) for s, us in s2u.items()))
    conf = d / "mfcc.conf"
    conf.write_text("--use-energy=false\n--dither=0\n")
    return data, conf, utts


def test_make_mfcc_cmvn_and_decode_steps(host, tmp_path):
    d = tmp_path
    data, conf, utts = make_datadir(d)
    # ---- MakeMfcc: nj = 2, compressed archives, utt2num_frames
    rc = host.vb_MakeMfcc(str(data).encode(), str(conf).encode(), 2, 1, 1)
    assert rc == 0, host.vb_steps_last_error()
    feats_scp = (data / "feats.scp").read_text().split("\n")[:-1]
    assert [l.split()[0] for l in feats_scp] == utts
    for j in (1, 2):
        assert (data / "mfcc" / ("raw_mfcc_test.%d.ark" % j)).exists() and (data / "log" / ("make_mfcc_test.%d.log" % j)).exists()
    assert not list((data / "log").glob("*.mfctemp"))
    n_frames = dict(l.split() for l in (data / "utt2num_frames").read_text().split("\n")[:-1])
    # the same features as the bin functions by hand (uncompressed), up to the codec's quantisation
    assert call(host.vb_ComputeMFCCFeats, ["compute-mfcc-feats", "--config=%s" % conf, "scp:%s" % (data / "wav.scp"), "ark:%s" % (d / "raw.ark")], "") == 0
    assert call(host.vb_CopyFeats, ["copy-feats", "scp:%s" % (data / "feats.scp"), "ark:%s" % (d / "back.ark")], "") == 0
    raw, back = read_fm_ark(str(d / "raw.ark")), read_fm_ark(str(d / "back.ark"))
    assert list(raw) == utts and list(back) == utts
    for u in utts:
        assert raw[u].shape == back[u].shape and int(n_frames[u]) == raw[u].shape[0]
        rng = raw[u].max(0) - raw[u].min(0)
        assert np.all(np.abs(raw[u] - back[u]) <= rng / 40.0 + 1e-4)      # (quartile-wise 8-bit codec)
    if have_ref("copy-feats"):      # the reference reads our compressed archives to the same numbers
        ref("copy-feats", "scp:%s" % (data / "feats.scp"), "ark:%s" % (d / "ref_back.ark"))
        rb = read_fm_ark(str(d / "ref_back.ark"))
        for u in utts:
            np.testing.assert_array_equal(rb[u], back[u])
    # a second run moves feats.scp to backup/
    assert host.vb_MakeMfcc(str(data).encode(), str(conf).encode(), 3, 0, 0) == 0
    assert (data / "backup" / "feats.scp").exists()
    # ---- ComputeCmvnStats
    assert host.vb_ComputeCmvnStatsDir(str(data).encode()) == 0, host.vb_steps_last_error()
    cm = (data / "cmvn.scp").read_text().split("\n")[:-1]
    assert [l.split()[0] for l in cm] == ["spk0", "spk1", "spk2"] and (data / "data" / "cmvn_test.ark").exists()
    # ---- SplitData keeps speakers together
    assert host.vb_SplitData(str(data).encode(), 2, 0) == 0
    parts = [(data / "split2" / str(j) / "utt2spk").read_text().split("\n")[:-1] for j in (1, 2)]
    assert sum(len(p) for p in parts) == len(utts) and all(parts)
    assert not ({l.split()[1] for l in parts[0]} & {l.split()[1] for l in parts[1]})
    # ... and leaves an up-to-date split alone (split_data.cpp:64-82): the files keep their modification times; a newer source splits again
    stamp = lambda: [os.stat(data / "split2" / str(j) / f).st_mtime_ns for j in (1, 2) for f in ("utt2spk", "feats.scp")]
    before = stamp()
    assert host.vb_SplitData(str(data).encode(), 2, 0) == 0 and stamp() == before
    os.utime(data / "feats.scp", ns=(before[0] + 10**9, before[0] + 10**9))
    assert host.vb_SplitData(str(data).encode(), 2, 0) == 0 and stamp() != before
    # ---- Decode: model directory exp/tri with final.mdl, graph directory with HCLG.fst
    g = load("files")
    exp = d / "exp" / "tri"
    graph = exp / "graph"
    graph.mkdir(parents=True)
    (exp / "final.mdl").write_bytes(g["file_final_mdl"].tobytes())
    (graph / "HCLG.fst").write_bytes(g["file_HCLG_fst"].tobytes())
    dec = exp / "decode_test"
    # scoring needs data/text and graph/words.txt: without them it is an error, not a silent skip
    assert host.vb_Decode(str(graph).encode(), str(data).encode(), str(dec).encode(), b"", 2, 0.083333, 7000, 13.0, 6.0, 0) == -1
    assert b"text" in host.vb_steps_last_error()
    rc = host.vb_Decode(str(graph).encode(), str(data).encode(), str(dec).encode(), b"", 2, 0.083333, 7000, 13.0, 6.0, 1)
    assert rc == 0, host.vb_steps_last_error()
    assert (dec / "num_jobs").read_text().strip() == "2"
    for j in (1, 2):
        assert (dec / ("lat.%d" % j)).stat().st_size > 0 and "Overall log-likelihood" in (dec / "log" / ("decode.%d.log" % j)).read_text()
        assert not (data / "split2" / str(j) / "apply_cmvn.temp").exists()
    if have_ref("lattice-copy"):    # the reference reads the lattice archives the step wrote
        for j in (1, 2):
            ref("lattice-copy", "ark:%s" % (dec / ("lat.%d" % j)), "ark,t:%s" % (d / ("lat%d.txt" % j)))
        got = (d / "lat1.txt").read_text() + (d / "lat2.txt").read_text()
        assert all(u in got for u in utts)


def test_decode_with_fmllr_transforms_and_scoring(host, tmp_path):
    """Decode(..., trans_dir) -- the reference's "decode a SAT model with supplied fMLLR transforms"
    path (VB/scr/steps/decode_gmm.cpp:127-172,243-267; config 3 of SURVEY.md 8d) -- followed by its
    scoring (ScoreKaldiWER): determinized lattices, words, transcripts, one WER file per
    (LM weight, word insertion penalty), the best of them recorded."""
    from tests import util
    from tests.test_host_lattice import assert_same_archives
    from voicebridge_amd import synth
    d = tmp_path
    data, conf, utts = make_datadir(d, n_utt=6, n_spk=3)
    assert host.vb_MakeMfcc(str(data).encode(), str(conf).encode(), 2, 1, 0) == 0, host.vb_steps_last_error()
    assert host.vb_ComputeCmvnStatsDir(str(data).encode()) == 0
    # model + lexicon graph with words; words.txt
    H, n_cd, n_words = 3, 10, 30
    P = n_cd * H
    g, _ = synth.lexicon_graph(n_words, P, seed=3, n_phones=n_cd)
    offs, w, means, iv = synth.gmm_model(P, P * 3, 39, seed=4, mean_scale=1.5, var_spread=0.3)
    exp = d / "exp" / "tri3"
    graph = exp / "graph"
    graph.mkdir(parents=True)
    util.write_mdl(str(exp / "final.mdl"), n_cd, H, offs, w, (means * iv).astype(np.float32), iv)
    util.write_vector_fst(str(graph / "HCLG.fst"), g["final"], g["arc_offset"], g["arcs"])
    (graph / "words.txt").write_text("<eps> 0\n" + "".join("w%d %d\n" % (i, i) for i in range(1, n_words + 2)))
    # per-speaker fMLLR transforms written by an "earlier pass" that ran with ONE job (the decode runs with two)
    trans = d / "exp" / "tri3" / "decode_si"
    trans.mkdir()
    (trans / "num_jobs").write_text("1\n")
    with open(trans / "trans.1", "wb") as f:
        for k in range(3):
            m = synth.affine(39, seed=10 + k)
            f.write(b"spk%d \0BFM " % k + bytes([4]) + np.int32(39).tobytes() + bytes([4]) + np.int32(40).tobytes() + m.astype("<f4").tobytes())
    dec = exp / "decode_test"
    args = [str(graph).encode(), str(data).encode(), str(dec).encode(), b"", str(trans).encode(), 2, 0.083333, 7000, 13.0, 6.0]
    # first without scoring (no data/text yet): lattices, words, transcripts
    assert host.vb_DecodeEx(*args, 1, b"0.0,0.5", 8, 10) == 0, host.vb_steps_last_error()
    lats = {}
    for j in (1, 2):
        lats.update(util.read_lattice_ark(str(dec / ("lat.%d" % j))))
        assert (dec / ("transcription%d.trn" % j)).exists()
    assert sorted(lats) == utts and all(v[0] == "compactlattice44" for v in lats.values())
    words = dict(l.split(None, 1) if " " in l.strip() else (l.strip(), "") for j in (1, 2) for l in (dec / ("w.%d" % j)).read_text().split("\n") if l.strip())
    trn = dict(l.split(None, 1) if " " in l.strip() else (l.strip(), "") for j in (1, 2) for l in (dec / ("transcription%d.trn" % j)).read_text().split("\n") if l.strip())
    for u in utts:
        assert [("w" + x) for x in words[u].split()] == trn[u].split()
    # the same features by hand through the bin functions give the same lattices (transforms really applied per speaker)
    sd = data / "split2" / "1"
    log = str(d / "hand.log")
    assert call(host.vb_ApplyCmvn, ["apply-cmvn", "--utt2spk=ark:%s" % (sd / "utt2spk"), "scp:%s" % (sd / "cmvn.scp"), "scp:%s" % (sd / "feats.scp"), "ark:%s" % (d / "c.ark")], log) == 0
    assert call(host.vb_AddDeltas, ["add-deltas", "ark:%s" % (d / "c.ark"), "ark:%s" % (d / "dl.ark")]) == 0
    assert call(host.vb_TransformFeats, ["transform-feats", "--utt2spk=ark:%s" % (sd / "utt2spk"), "ark:%s" % (trans / "trans.1"), "ark:%s" % (d / "dl.ark"), "ark:%s" % (d / "tf.ark")], log) == 0
    assert call(host.vb_GmmLatgenFaster, ["gmm-latgen-faster", "--max-active=7000", "--beam=13.000000", "--lattice-beam=6.000000", "--acoustic-scale=0.083333",
                                          "--allow-partial=true", str(exp / "final.mdl"), str(graph / "HCLG.fst"), "ark:%s" % (d / "tf.ark"), "ark:%s" % (d / "hand.lat")], log) == 0
    assert_same_archives(str(d / "hand.lat"), str(dec / "lat.1"))
    untransformed = d / "untr.lat"
    assert call(host.vb_GmmLatgenFaster, ["gmm-latgen-faster", "--max-active=7000", "--beam=13.000000", "--lattice-beam=6.000000", "--acoustic-scale=0.083333",
                                          "--allow-partial=true", str(exp / "final.mdl"), str(graph / "HCLG.fst"), "ark:%s" % (d / "dl.ark"), "ark:%s" % untransformed], log) == 0
    assert util.read_lattice_ark(str(untransformed)) != util.read_lattice_ark(str(dec / "lat.1"))
    # ---- now with scoring: reference text = the hypotheses with two edits
    lines = []
    for i, u in enumerate(utts):
        ws = trn[u].split()
        if i == 0:
            ws = ws[1:]
        if i == 1:
            ws = ws + ["w1"]
        lines.append(" ".join([u] + ws))
    (data / "text").write_text("\n".join(lines) + "\n")
    assert host.vb_DecodeEx(*args, 0, b"0.0,0.5", 8, 10) == 0, host.vb_steps_last_error()
    for wip in ("0.0", "0.5"):
        for lmwt in (8, 9, 10):
            f = (dec / ("wer_%d_%s" % (lmwt, wip))).read_text().split()
            assert len(f) == 9 and int(f[8]) == len(utts)
            assert (dec / "scoring_kaldi" / ("penalty_" + wip) / ("%d.txt" % lmwt)).exists()
    best = (dec / "scoring_kaldi" / "best_wer").read_text().split()
    assert (dec / "scoring_kaldi" / "wer_details" / "lmwt").read_text().strip() in ("8", "9", "10")
    assert float(best[6]) == min(float((dec / ("wer_%d_%s" % (l, w))).read_text().split()[6]) for w in ("0.0", "0.5") for l in (8, 9, 10))
    # one (LMWT, penalty) pair by hand through the bin functions: same hypotheses
    with open(d / "all.lat", "wb") as f:
        f.write((dec / "lat.1").read_bytes() + (dec / "lat.2").read_bytes())
    assert call(host.vb_LatticeScale, ["lattice-scale", "--inv-acoustic-scale=9", "ark:%s" % (d / "all.lat"), "ark:%s" % (d / "s.lat")], log) == 0
    assert call(host.vb_LatticeAddPenalty, ["lattice-add-penalty", "--word-ins-penalty=0.5", "ark:%s" % (d / "s.lat"), "ark:%s" % (d / "p.lat")], log) == 0
    assert call(host.vb_LatticeBestPath, ["lattice-best-path", "ark:%s" % (d / "p.lat"), "ark,t:%s" % (d / "bp.txt")], log) == 0
    hand = {l.split()[0]: ["w" + x for x in l.split()[1:]] for l in (d / "bp.txt").read_text().split("\n") if l.strip()}
    got = {l.split()[0]: l.split()[1:] for l in (dec / "scoring_kaldi" / "penalty_0.5" / "9.txt").read_text().split("\n") if l.strip()}
    assert hand == got
    if have_ref("lattice-scale") and have_ref("lattice-best-path") and have_ref("lattice-add-penalty"):
        # and through the reference's own programs on the lattices this build wrote
        ref("lattice-scale", "--inv-acoustic-scale=9", "ark:%s" % (d / "all.lat"), "ark:%s" % (d / "rs.lat"))
        ref("lattice-add-penalty", "--word-ins-penalty=0.5", "ark:%s" % (d / "rs.lat"), "ark:%s" % (d / "rp.lat"))
        ref("lattice-best-path", "ark:%s" % (d / "rp.lat"), "ark,t:%s" % (d / "rbp.txt"))
        assert (d / "rbp.txt").read_text().split() == (d / "bp.txt").read_text().split()


def test_decode_fmllr_end_to_end(host, tmp_path):
    """DecodeFmllr (VB/scr/steps/decode_fmllr.cpp:91-760; config 3 of SURVEY.md 8d end to end): SI pass,
    fMLLR from its lattices, main search on the adapted features, second estimate, rescoring,
    determinization, best path, scoring.  Per-stage parity with the reference's programs is
    tests/test_host_fmllr.py; here: the files the reference leaves, transforms that are real updates,
    and the adapted features fitting the model better than the speaker-independent ones."""
    from tests import util
    from voicebridge_amd import synth
    d = tmp_path
    data, conf, utts = make_datadir(d, n_utt=4, n_spk=2, seconds=4.5)      # well over 500 non-silence frames per speaker (fMLLR min-count)
    assert host.vb_MakeMfcc(str(data).encode(), str(conf).encode(), 2, 1, 0) == 0, host.vb_steps_last_error()
    assert host.vb_ComputeCmvnStatsDir(str(data).encode()) == 0
    H, n_cd, n_words = 3, 10, 30
    P = n_cd * H
    g, _ = synth.lexicon_graph(n_words, P, seed=3, n_phones=n_cd)
    offs, w, means, iv = synth.gmm_model(P, P * 3, 39, seed=4, mean_scale=1.5, var_spread=0.3)
    exp = d / "exp" / "tri3"
    graph = exp / "graph"
    (graph / "phones").mkdir(parents=True)
    util.write_mdl(str(exp / "final.mdl"), n_cd, H, offs, w, (means * iv).astype(np.float32), iv)
    util.write_const_fst(str(graph / "HCLG.fst"), g["final"], g["arc_offset"], g["arcs"])      # (a ConstFst, as the reference's mkgraph writes HCLG: the mapped reader)
    (graph / "words.txt").write_text("<eps> 0\n" + "".join("w%d %d\n" % (i, i) for i in range(1, n_words + 2)))
    (graph / "phones" / "silence.csl").write_text("1:2\n")
    dec = exp / "decode_test"
    args = [str(graph).encode(), str(data).encode(), str(dec).encode(), 2, 0.083333, 7000, 13.0, 6.0, 0.01, b""]
    assert host.vb_DecodeFmllr(*args, 1, b"0.0,0.5", 9, 10) == 0, host.vb_steps_last_error()
    si = exp / "decode_test.si"
    assert (si / "lat.1").exists() and (si / "num_jobs").read_text().strip() == "2"
    lats = {}
    for j in (1, 2):
        lats.update(util.read_lattice_ark(str(dec / ("lat.%d" % j))))
        assert (dec / ("w.%d" % j)).exists() and (dec / ("transcription%d.trn" % j)).exists()
        assert not (dec / ("lat.tmp.%d" % j)).exists() and not (dec / ("pre_trans.%d" % j)).exists()
        assert not list((data / "split2" / str(j)).glob("*.temp"))
    assert sorted(lats) == utts and all(v[0] == "compactlattice44" for v in lats.values())
    from tests.test_host_io import read_fm_ark
    trans = {}
    for j in (1, 2):
        trans.update(read_fm_ark(str(dec / ("trans.%d" % j))))
    assert sorted(trans) == ["spk0", "spk1"]
    for m in trans.values():
        assert m.shape == (39, 40) and np.abs(m - np.eye(39, 40)).max() > 0.05 and np.isfinite(m).all()
    # likelihood of the best path per frame: adapted pass above the speaker-independent pass
    def like(logfile):
        l = [x for x in open(logfile) if "Overall log-likelihood per frame is" in x][-1]
        return float(l.split("per frame is")[1].split("over")[0])
    si_like = np.mean([like(si / "log" / ("decode.%d.log" % j)) for j in (1, 2)])
    ad_like = np.mean([like(dec / "log" / ("fmllr.%d.log" % j)) for j in (1, 2)])
    assert ad_like > si_like
    # second call with scoring, reusing the SI directory (si_dir argument): WER files of both LM weights
    trn = dict(l.split(None, 1) if " " in l.strip() else (l.strip(), "") for j in (1, 2) for l in (dec / ("transcription%d.trn" % j)).read_text().split("\n") if l.strip())
    (data / "text").write_text("".join("%s %s\n" % (u, trn[u].strip()) for u in utts))
    args[-1] = str(si).encode()
    assert host.vb_DecodeFmllr(*args, 0, b"0.0,0.5", 9, 10) == 0, host.vb_steps_last_error()
    for wip in ("0.0", "0.5"):
        for lmwt in (9, 10):
            f = (dec / ("wer_%d_%s" % (lmwt, wip))).read_text().split()
            assert len(f) == 9 and int(f[8]) == len(utts)
    assert (dec / "scoring_kaldi" / "best_wer").exists()
    # a missing silence list is an error
    os.remove(graph / "phones" / "silence.csl")
    assert host.vb_DecodeFmllr(*args, 1, b"0.0", 9, 9) == -1 and b"silence.csl" in host.vb_steps_last_error()


def test_decode_lda_mllt_sat_model_against_the_reference_chain(host, tmp_path):
    """Config 4 of SURVEY.md 8d: an LDA+MLLT+SAT model directory (final.mat 40 x 117, splice_opts, D = 40, per-speaker 40 x 41
    transforms from trans_dir).  Decode() must build apply-cmvn | splice-feats | transform-feats final.mat | transform-feats
    trans.JOB | gmm-latgen-faster (VB/scr/steps/decode_gmm.cpp:102-106, 243-267): checked against this build's bin functions
    run by hand (identical lattices) and against the reference's own programs run on the same files (features within float
    tolerance, same best-path words)."""
    from tests import util
    from tests.test_host_lattice import assert_same_archives
    from voicebridge_amd import synth
    d = tmp_path
    data, conf, utts = make_datadir(d, n_utt=6, n_spk=3, seconds=1.2)
    assert host.vb_MakeMfcc(str(data).encode(), str(conf).encode(), 2, 0, 0) == 0, host.vb_steps_last_error()      # uncompressed: same inputs for both sides
    assert host.vb_ComputeCmvnStatsDir(str(data).encode()) == 0
    H, n_cd, n_words, D = 3, 10, 30, 40
    P = n_cd * H
    g, _ = synth.lexicon_graph(n_words, P, seed=3, n_phones=n_cd)
    rng = np.random.default_rng(12)
    q, _r = np.linalg.qr(rng.standard_normal((117, D)))
    lda = np.ascontiguousarray(q.T, np.float32)                      # final.mat: 40 x 117, orthonormal rows
    # place the model in the space of the spliced + projected features of this data
    assert call(host.vb_ApplyCmvn, ["apply-cmvn", "--utt2spk=ark:%s" % (data / "utt2spk"), "scp:%s" % (data / "cmvn.scp"), "scp:%s" % (data / "feats.scp"),
                                    "ark:%s" % (d / "all_c.ark")], "") == 0
    sample = np.concatenate([np.concatenate([np.pad(m, ((4, 4), (0, 0)), mode="edge")[k:k + len(m)] for k in range(9)], 1) @ lda.T
                             for m in read_fm_ark(str(d / "all_c.ark")).values()])
    offs, w, means, iv = synth.gmm_model(P, P * 3, D, seed=4, mean_scale=1.5, dim_scale=sample.std(0) + 1e-3, var_spread=0.3)
    means = means + sample.mean(0).astype(np.float32)
    exp = d / "exp" / "tri3b"
    graph = exp / "graph"
    graph.mkdir(parents=True)
    util.write_mdl(str(exp / "final.mdl"), n_cd, H, offs, w, (means * iv).astype(np.float32), iv)
    util.write_vector_fst(str(graph / "HCLG.fst"), g["final"], g["arc_offset"], g["arcs"])
    (graph / "words.txt").write_text("<eps> 0\n" + "".join("w%d %d\n" % (i, i) for i in range(1, n_words + 2)))
    with open(exp / "final.mat", "wb") as f:
        f.write(b"\0BFM " + bytes([4]) + np.int32(D).tobytes() + bytes([4]) + np.int32(117).tobytes() + lda.astype("<f4").tobytes())
    (exp / "splice_opts").write_text("--left-context=4 --right-context=4\n")
    trans = exp / "decode_si"
    trans.mkdir()
    (trans / "num_jobs").write_text("1\n")
    with open(trans / "trans.1", "wb") as f:
        for k in range(3):
            m = synth.affine(D, seed=20 + k)
            f.write(b"spk%d \0BFM " % k + bytes([4]) + np.int32(D).tobytes() + bytes([4]) + np.int32(D + 1).tobytes() + m.astype("<f4").tobytes())
    dec = exp / "decode_test"
    assert host.vb_DecodeEx(str(graph).encode(), str(data).encode(), str(dec).encode(), b"", str(trans).encode(), 2, 0.083333, 7000, 13.0, 6.0,
                            1, b"0.0", 9, 9) == 0, host.vb_steps_last_error()
    words = {l.split()[0]: l.split()[1:] for j in (1, 2) for l in (dec / ("w.%d" % j)).read_text().split("\n") if l.strip()}
    assert sorted(words) == utts
    log = str(d / "hand.log")
    latgen = ["--max-active=7000", "--beam=13.000000", "--lattice-beam=6.000000", "--acoustic-scale=0.083333", "--allow-partial=true",
              str(exp / "final.mdl"), str(graph / "HCLG.fst")]
    for j in (1, 2):
        sd = data / "split2" / str(j)
        # ---- this build's bin functions, archive to archive
        assert call(host.vb_ApplyCmvn, ["apply-cmvn", "--utt2spk=ark:%s" % (sd / "utt2spk"), "scp:%s" % (sd / "cmvn.scp"), "scp:%s" % (sd / "feats.scp"), "ark:%s" % (d / "c.ark")], log) == 0
        assert call(host.vb_SpliceFeats, ["splice-feats", "--left-context=4", "--right-context=4", "ark:%s" % (d / "c.ark"), "ark:%s" % (d / "sp.ark")]) == 0
        assert call(host.vb_TransformFeats, ["transform-feats", str(exp / "final.mat"), "ark:%s" % (d / "sp.ark"), "ark:%s" % (d / "ld.ark")], log) == 0
        assert call(host.vb_TransformFeats, ["transform-feats", "--utt2spk=ark:%s" % (sd / "utt2spk"), "ark:%s" % (trans / "trans.1"), "ark:%s" % (d / "ld.ark"), "ark:%s" % (d / "tf.ark")], log) == 0
        assert call(host.vb_GmmLatgenFaster, ["gmm-latgen-faster"] + latgen + ["ark:%s" % (d / "tf.ark"), "ark:%s" % (d / "hand.lat")], log) == 0
        assert_same_archives(str(d / "hand.lat"), str(dec / ("lat.%d" % j)))
        mine = read_fm_ark(str(d / "tf.ark"))
        assert all(m.shape[1] == D for m in mine.values())
        if have_ref("splice-feats") and have_ref("gmm-latgen-faster"):
            # ---- the reference's programs on the same split directory
            ref("apply-cmvn", "--utt2spk=ark:%s" % (sd / "utt2spk"), "scp:%s" % (sd / "cmvn.scp"), "scp:%s" % (sd / "feats.scp"), "ark:%s" % (d / "r_c.ark"))
            ref("splice-feats", "--left-context=4", "--right-context=4", "ark:%s" % (d / "r_c.ark"), "ark:%s" % (d / "r_sp.ark"))
            ref("transform-feats", str(exp / "final.mat"), "ark:%s" % (d / "r_sp.ark"), "ark:%s" % (d / "r_ld.ark"))
            ref("transform-feats", "--utt2spk=ark:%s" % (sd / "utt2spk"), "ark:%s" % (trans / "trans.1"), "ark:%s" % (d / "r_ld.ark"), "ark:%s" % (d / "r_tf.ark"))
            theirs = read_fm_ark(str(d / "r_tf.ark"))
            assert list(theirs) == list(mine)
            for u in mine:
                np.testing.assert_allclose(mine[u], theirs[u], rtol=1e-5, atol=2e-4)
            ref("gmm-latgen-faster", *latgen, "ark:%s" % (d / "r_tf.ark"), "ark:%s" % (d / "r.lat"), "ark,t:%s" % (d / "r_w.txt"))
            rw = {l.split()[0]: l.split()[1:] for l in (d / "r_w.txt").read_text().split("\n") if l.strip()}
            for u in rw:
                assert rw[u] == words[u], u


def test_job_threads_leave_the_same_files(host, tmp_path):
    """vb_set_step_devices: the jobs of a step on host threads of their own (here three threads sharing the one GPU and ONE
    pool of search workspaces between their decoders), as VB/scr/steps/decode_gmm.cpp:300-324 runs its nj jobs side by side.
    Same files as the jobs one after the other; a second round on the same (persistent) job threads too."""
    from tests import util
    from voicebridge_amd import synth
    H, n_cd, n_words = 3, 10, 30
    P = n_cd * H
    g, _ = synth.lexicon_graph(n_words, P, seed=3, n_phones=n_cd)
    offs, w, means, iv = synth.gmm_model(P, P * 3, 39, seed=4, mean_scale=1.5, var_spread=0.3)
    host.vb_set_step_devices.argtypes = [C.POINTER(C.c_int), C.c_int]

    def read_fm_bytes(b):
        f = tmp_path / "m.ark"
        f.write_bytes(b)
        return read_fm_ark(str(f))
    out = {}
    try:
        # (combined: VB_STEP_COMBINED_SEARCH=1, the searches of all jobs of a step as one batch -- same files)
        # (file_temps: VB_STEP_FILE_TEMPS=1, the temporaries between the programs as the reference's files instead of in-memory tables --
        #  posteriors as objects, lattices as bytes; host_update: VB_FMLLR_HOST_UPDATE=1, the fMLLR row update on the host's threads instead
        #  of the device -- same files again)
        for tag, devs in (("serial", [0]), ("threads", [0, 0, 0]), ("threads_again", []), ("combined", [0, 0, 0]), ("file_temps", [0, 0, 0]), ("host_update", [0, 0, 0])):      # ([]: the default, four job threads on device 0)
            for k in ("VB_STEP_COMBINED_SEARCH", "VB_STEP_FILE_TEMPS", "VB_FMLLR_HOST_UPDATE"):
                os.environ.pop(k, None)
            if tag == "combined":
                os.environ["VB_STEP_COMBINED_SEARCH"] = "1"
            if tag == "file_temps":
                os.environ["VB_STEP_FILE_TEMPS"] = "1"
            if tag == "host_update":
                os.environ["VB_FMLLR_HOST_UPDATE"] = "1"
            d = tmp_path / tag
            d.mkdir()
            data, conf, utts = make_datadir(d, n_utt=6, n_spk=3, seconds=4.5)
            assert host.vb_set_step_devices((C.c_int * max(len(devs), 1))(*devs), len(devs)) == 0
            assert host.vb_MakeMfcc(str(data).encode(), str(conf).encode(), 3, 1, 0) == 0, host.vb_steps_last_error()
            assert host.vb_ComputeCmvnStatsDir(str(data).encode()) == 0
            exp = d / "exp" / "tri3"
            graph = exp / "graph"
            (graph / "phones").mkdir(parents=True)
            util.write_mdl(str(exp / "final.mdl"), n_cd, H, offs, w, (means * iv).astype(np.float32), iv)
            util.write_vector_fst(str(graph / "HCLG.fst"), g["final"], g["arc_offset"], g["arcs"])
            (graph / "words.txt").write_text("<eps> 0\n" + "".join("w%d %d\n" % (i, i) for i in range(1, n_words + 2)))
            (graph / "phones" / "silence.csl").write_text("1:2\n")
            dec = exp / "decode_test"
            assert host.vb_DecodeFmllr(str(graph).encode(), str(data).encode(), str(dec).encode(), 3, 0.083333, 7000, 13.0, 6.0, 0.01, b"",
                                       1, b"0.0", 9, 9) == 0, host.vb_steps_last_error()
            # which worker ran which job on which device: the fan-out is one code path whatever the list holds (worker t sets devs[t] and
            # runs jobs t+1, t+1+nt, ...), so [0, 0, 0] exercises everything a three-GPU list does except the device index
            dv, wk = (C.c_int * 8)(), (C.c_int * 8)()
            nj_ = host.vb_last_step_jobs(dv, wk, 8)
            assert nj_ == 3 and list(dv[:3]) == [0, 0, 0]
            assert list(wk[:3]) == ([-1, -1, -1] if tag == "serial" else [0, 1, 2])
            logs = sorted(p_.name for p_ in (dec / "log").iterdir())
            assert ("fmllr.all.log" in logs) == (tag == "combined")
            files = {}
            for j in (1, 2, 3):
                for n in ("lat.%d", "trans.%d", "w.%d"):
                    files[n % j] = (dec / (n % j)).read_bytes()
                files["si_lat.%d" % j] = (exp / "decode_test.si" / ("lat.%d" % j)).read_bytes()
                files["mfcc.%d" % j] = (data / "mfcc" / ("raw_mfcc_test.%d.ark" % j)).read_bytes()
            assert all(len(v) > 0 for v in files.values())
            out[tag] = files
    finally:
        host.vb_set_step_devices((C.c_int * 1)(), 0)
        for k in ("VB_STEP_COMBINED_SEARCH", "VB_STEP_FILE_TEMPS", "VB_FMLLR_HOST_UPDATE"):
            os.environ.pop(k, None)
    # The searches, the feature files and the words are the same byte for byte.  The fMLLR statistics are summed with f64 atomics
    # (vbh_fmllr.hip), whose order depends on what else the device is running: transforms agree to rounding, and so do the costs
    # of the final lattices, whose states, arcs and labels are the same.
    def close_lattices(a, b):
        la, lb = util.read_lattice_ark_bytes(a), util.read_lattice_ark_bytes(b)
        assert sorted(la) == sorted(lb)
        for k in la:
            (ka, sa, sta), (kb, sb, stb) = la[k], lb[k]
            assert ka == kb and sa == sb and len(sta) == len(stb)
            for (fa, aa), (fb, ab) in zip(sta, stb):
                assert fa[2:] == fb[2:] and np.allclose(fa[:2], fb[:2], rtol=1e-4, atol=2e-3) and len(aa) == len(ab)
                for x, y in zip(aa, ab):
                    assert (x[0], x[3], x[4]) == (y[0], y[3], y[4]) and np.allclose(x[1:3], y[1:3], rtol=1e-4, atol=2e-3)
    for tag in ("threads", "threads_again", "combined", "file_temps", "host_update"):
        assert sorted(out[tag]) == sorted(out["serial"])
        for n, v in out["serial"].items():
            if n.startswith("trans."):
                ta, tb = read_fm_bytes(out[tag][n]), read_fm_bytes(v)
                assert sorted(ta) == sorted(tb) and all(np.allclose(ta[k], tb[k], rtol=1e-5, atol=1e-6) for k in ta), (tag, n)
            elif n.startswith("lat."):
                close_lattices(out[tag][n], v)
            else:
                assert out[tag][n] == v, (tag, n)


def test_make_mfcc_with_speaker_warp_factors(host, tmp_path):
    """MakeMfcc on a data directory with spk2warp (VB/scr/steps/make_mfcc.cpp:63-76: --vtln-map=ark:spk2warp --utt2spk=ark:utt2spk):
    the archives hold what compute-mfcc-feats computes with each speaker's warp factor"""
    from oracle import vbo
    data, conf, utts = make_datadir(tmp_path, n_utt=4, n_spk=2, seconds=0.8)
    (data / "spk2warp").write_text("spk0 0.92\nspk1 1.08\n")
    assert host.vb_MakeMfcc(str(data).encode(), str(conf).encode(), 2, 0, 0) == 0, host.vb_steps_last_error()
    feats = {}
    for j in (1, 2):
        feats.update(read_fm_ark(str(data / "mfcc" / ("raw_mfcc_test.%d.ark" % j))))
    assert sorted(feats) == utts
    import wave as wavmod
    o = vbo.mfcc_opts()
    o.use_energy = 0
    for i, u in enumerate(utts):
        with wavmod.open(str(data / (u + ".wav")), "rb") as f:
            x = np.frombuffer(f.readframes(f.getnframes()), "<i2").astype(np.float32)
        np.testing.assert_allclose(feats[u], vbo.mfcc(x, o, vtln_warp=0.92 if i < 2 else 1.08), atol=1e-3, rtol=1e-5)
    assert np.abs(feats[utts[0]] - vbo.mfcc(np.frombuffer(wavmod.open(str(data / (utts[0] + ".wav")), "rb").readframes(10 ** 7), "<i2").astype(np.float32), o)).max() > 0.05


def test_job_threads_report_a_failing_job_and_go_on(host, tmp_path):
    """a job that fails on its thread fails the step with its message (the first one), the other jobs finish, and the job threads
    serve the next step"""
    host.vb_set_step_devices.argtypes = [C.POINTER(C.c_int), C.c_int]
    data, conf, utts = make_datadir(tmp_path, n_utt=4, n_spk=2, seconds=0.6)
    try:
        assert host.vb_set_step_devices((C.c_int * 2)(0, 0), 2) == 0
        bad = tmp_path / "bad.conf"
        bad.write_text("--use-energy=false\n--dither=0\n--no-such-option=1\n")
        assert host.vb_MakeMfcc(str(data).encode(), str(bad).encode(), 2, 1, 0) == -1
        assert b"compute-mfcc-feats failed for job" in host.vb_steps_last_error()
        assert host.vb_MakeMfcc(str(data).encode(), str(conf).encode(), 2, 1, 0) == 0, host.vb_steps_last_error()
        assert len((data / "feats.scp").read_text().strip().split("\n")) == len(utts)
    finally:
        host.vb_set_step_devices((C.c_int * 1)(), 0)
