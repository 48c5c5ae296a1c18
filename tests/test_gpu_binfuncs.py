"""GPU: the reference's argv-style bin functions (voicebridge_amd/host/voicebridge_api.cc) run end
to end on files, checked against the REFERENCE'S OWN command-line programs (oracle/_ref/bin,
compiled from /root/reference; they travel to the GPU box as binaries) when present, and
against the oracle otherwise."""
import ctypes as C
import os
import subprocess
import wave as wavmod

import numpy as np
import pytest

from tests.test_host_io import read_fm_ark
from tests.util import load

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REFBIN = os.path.join(ROOT, "oracle", "_ref", "bin")


def have_ref(name):
    return os.access(os.path.join(REFBIN, name), os.X_OK)


def ref(name, *args):
    subprocess.check_call([os.path.join(REFBIN, name)] + list(args), env=dict(os.environ, OPENBLAS_NUM_THREADS="1"),
                          stderr=subprocess.DEVNULL)


@pytest.fixture(scope="module")
def host():
    from voicebridge_amd import build
    lib = C.CDLL(build.HOST_LIB)
    return lib


def call(fn, args, log=None):
    argv = (C.c_char_p * len(args))(*[a.encode() for a in args])
    if log is None and fn.__name__ in ("vb_AddDeltas", "vb_SpliceFeats"):
        return fn(len(args), argv)
    return fn(len(args), argv, (log or "").encode())


def write_wavs(d, waves):
    lines = []
    for i, w in enumerate(waves):
        p = d / ("utt%02d.wav" % i)
        with wavmod.open(str(p), "wb") as f:
            f.setnchannels(1); f.setsampwidth(2); f.setframerate(16000)
            f.writeframes(np.asarray(w, "<i2").tobytes())
        lines.append("utt%02d %s" % (i, p))
    (d / "wav.scp").write_text("\n".join(lines) + "\n")


def test_feature_pipeline_vs_reference_programs(host, tmp_path):
    from oracle import vbo
    from voicebridge_amd import synth
    d = tmp_path
    waves = [synth.wave(i, 0.8 + 0.3 * i) for i in range(4)] + [synth.wave(9, 0.01)]   # the last is too short: skipped
    write_wavs(d, waves)
    log = str(d / "log.txt")
    # --- compute-mfcc-feats
    assert call(host.vb_ComputeMFCCFeats, ["compute-mfcc-feats", "--dither=0", "scp,p:%s" % (d / "wav.scp"),
                                           "ark,scp:%s,%s" % (d / "mfcc.ark", d / "mfcc.scp")], log) == 0
    # a stock configuration (no --dither: the reference's default 1.0) runs; on these loud waves the dithered cepstra stay
    # within a few tenths of the undithered ones, the reference's own dithered output does the same
    assert call(host.vb_ComputeMFCCFeats, ["compute-mfcc-feats", "scp:%s" % (d / "wav.scp"), "ark:%s" % (d / "x.ark")], log) == 0
    dith = read_fm_ark(d / "x.ark")
    mine = read_fm_ark(d / "mfcc.ark")
    for k in mine:
        if len(mine[k]):
            dd = np.abs(dith[k] - mine[k]).max()
            assert 0 < dd < 0.5, dd
    if have_ref("compute-mfcc-feats"):
        ref("compute-mfcc-feats", "scp:%s" % (d / "wav.scp"), "ark:%s" % (d / "ref_dith.ark"))
        rd = read_fm_ark(d / "ref_dith.ark")
        for k in mine:
            if len(mine[k]):
                assert np.abs(rd[k] - mine[k]).max() < 0.5
    assert sorted(mine) == ["utt00", "utt01", "utt02", "utt03", "utt04"]   # snip-edges: 0 rows for the short one? see below
    if have_ref("compute-mfcc-feats"):
        ref("compute-mfcc-feats", "--dither=0", "scp:%s" % (d / "wav.scp"), "ark:%s" % (d / "ref_mfcc.ark"))
        want = read_fm_ark(d / "ref_mfcc.ark")
    else:
        want = {"utt%02d" % i: vbo.mfcc(w.astype(np.float32), vbo.mfcc_opts()) for i, w in enumerate(waves)}
    for k in want:
        assert mine[k].shape == want[k].shape
        if len(want[k]):
            np.testing.assert_allclose(mine[k], want[k], atol=1e-3, rtol=1e-5)
    # --- compute-cmvn-stats --spk2utt | apply-cmvn --utt2spk | add-deltas | transform-feats --utt2spk
    # (on the four real utterances: the reference's apply-cmvn aborts on the 0 x 0 matrix of the short one)
    write_wavs(d, waves[:4])
    assert call(host.vb_ComputeMFCCFeats, ["compute-mfcc-feats", "--dither=0", "scp:%s" % (d / "wav.scp"),
                                           "ark,scp:%s,%s" % (d / "mfcc.ark", d / "mfcc.scp")], log) == 0
    (d / "spk2utt").write_text("spkA utt00 utt02\nspkB utt01 utt03\n")
    (d / "utt2spk").write_text("utt00 spkA\nutt01 spkB\nutt02 spkA\nutt03 spkB\n")
    # feed both sides the SAME mfcc archive (only the 4 real utterances)
    assert call(host.vb_ComputeCmvnStats, ["compute-cmvn-stats", "--spk2utt=ark:%s" % (d / "spk2utt"), "scp:%s" % (d / "mfcc.scp"),
                                           "ark:%s" % (d / "cmvn.ark")], log) == 0
    assert call(host.vb_ApplyCmvn, ["apply-cmvn", "--utt2spk=ark:%s" % (d / "utt2spk"), "ark:%s" % (d / "cmvn.ark"),
                                    "scp:%s" % (d / "mfcc.scp"), "ark:%s" % (d / "cm.ark")], log) in (0,)
    assert call(host.vb_AddDeltas, ["add-deltas", "ark:%s" % (d / "cm.ark"), "ark:%s" % (d / "dl.ark")]) == 0
    xf = {"spkA": synth.affine(39, seed=1), "spkB": synth.affine(39, seed=2)}
    with open(d / "trans.ark", "wb") as f:
        for k, m in xf.items():
            f.write(k.encode() + b" \0BFM " + bytes([4]) + np.int32(39).tobytes() + bytes([4]) + np.int32(40).tobytes() + m.astype("<f4").tobytes())
    assert call(host.vb_TransformFeats, ["transform-feats", "--utt2spk=ark:%s" % (d / "utt2spk"), "ark:%s" % (d / "trans.ark"),
                                         "ark:%s" % (d / "dl.ark"), "ark:%s" % (d / "final.ark")], log) == 0
    mine = read_fm_ark(d / "final.ark")
    if all(have_ref(b) for b in ("compute-cmvn-stats", "apply-cmvn", "add-deltas", "transform-feats")):
        ref("compute-cmvn-stats", "--spk2utt=ark:%s" % (d / "spk2utt"), "ark:%s" % (d / "mfcc.ark"), "ark:%s" % (d / "r_cmvn.ark"))
        ref("apply-cmvn", "--utt2spk=ark:%s" % (d / "utt2spk"), "ark:%s" % (d / "r_cmvn.ark"), "ark:%s" % (d / "mfcc.ark"), "ark:%s" % (d / "r_cm.ark"))
        ref("add-deltas", "ark:%s" % (d / "r_cm.ark"), "ark:%s" % (d / "r_dl.ark"))
        ref("transform-feats", "--utt2spk=ark:%s" % (d / "utt2spk"), "ark:%s" % (d / "trans.ark"), "ark:%s" % (d / "r_dl.ark"), "ark:%s" % (d / "r_final.ark"))
        want = read_fm_ark(d / "r_final.ark")
    else:
        raw = read_fm_ark(d / "mfcc.ark")
        st = {"spkA": vbo.acc_cmvn_stats(raw["utt00"]) + vbo.acc_cmvn_stats(raw["utt02"]),
              "spkB": vbo.acc_cmvn_stats(raw["utt01"]) + vbo.acc_cmvn_stats(raw["utt03"])}
        spk = {"utt00": "spkA", "utt01": "spkB", "utt02": "spkA", "utt03": "spkB"}
        want = {u: vbo.transform_feats(vbo.add_deltas(vbo.apply_cmvn(raw[u], st[spk[u]])), xf[spk[u]]) for u in spk}
    for k in want:
        np.testing.assert_allclose(mine[k], want[k], atol=5e-3, rtol=1e-4)
    # --- splice-feats
    assert call(host.vb_SpliceFeats, ["splice-feats", "--left-context=3", "--right-context=2", "ark:%s" % (d / "cm.ark"), "ark:%s" % (d / "sp.ark")]) == 0
    if have_ref("splice-feats"):
        ref("splice-feats", "--left-context=3", "--right-context=2", "ark:%s" % (d / "cm.ark"), "ark:%s" % (d / "r_sp.ark"))
        a, b = read_fm_ark(d / "sp.ark"), read_fm_ark(d / "r_sp.ark")
        for k in b:
            np.testing.assert_array_equal(a[k], b[k])          # pure copy: bit-exact


def test_gmm_latgen_faster_vs_reference_program(host, tmp_path):
    """final.mdl / HCLG.fst written by the reference's writers; decode with our GmmLatgenFaster and
    with the reference's gmm-latgen-faster; compare words and alignments."""
    from oracle import vbo
    from voicebridge_amd import synth
    d = tmp_path
    g = load("files")
    (d / "final.mdl").write_bytes(g["file_final_mdl"].tobytes())
    (d / "HCLG.fst").write_bytes(g["file_HCLG_fst"].tobytes())
    rng = np.random.default_rng(0)
    feats = {"uttA": (rng.standard_normal((40, 39)) * 2).astype(np.float32), "uttB": (rng.standard_normal((25, 39)) * 2).astype(np.float32),
             "uttC": (rng.standard_normal((1, 39)) * 2).astype(np.float32)}
    with open(d / "feats.ark", "wb") as f:
        for k, m in feats.items():
            f.write(k.encode() + b" \0BFM " + bytes([4]) + np.int32(m.shape[0]).tobytes() + bytes([4]) + np.int32(39).tobytes() + m.tobytes())
    opts = ["--beam=13", "--lattice-beam=6", "--max-active=7000", "--acoustic-scale=0.083333", "--allow-partial=true",
            "--determinize-lattice=false"]
    log = str(d / "dec.log")
    rc = call(host.vb_GmmLatgenFaster, ["gmm-latgen-faster"] + opts + [str(d / "final.mdl"), str(d / "HCLG.fst"), "ark:%s" % (d / "feats.ark"),
                                        "ark:%s" % (d / "lat.ark"), "ark,t:%s" % (d / "words.txt"), "ark,t:%s" % (d / "ali.txt")], log)
    assert rc == 0, open(log).read()
    # --minimize=true (determinized output pushed and minimized, K/lat/determinize-lattice-pruned.cc:1460-1466; bit-exactness of that
    # stage against the reference's programs: tests/test_host_lattice.py): never more states, the same best path
    common = ["--beam=13", "--lattice-beam=6", "--acoustic-scale=0.083333", str(d / "final.mdl"), str(d / "HCLG.fst"), "ark:%s" % (d / "feats.ark")]
    assert call(host.vb_GmmLatgenFaster, ["gmm-latgen-faster", "--minimize=true"] + common + ["ark:%s" % (d / "lat_min.ark"), "ark,t:%s" % (d / "w_min.txt")], log) == 0
    assert call(host.vb_GmmLatgenFaster, ["gmm-latgen-faster"] + common + ["ark:%s" % (d / "lat_det.ark"), "ark,t:%s" % (d / "w_det.txt")], log) == 0
    from tests import util
    n_min = sum(len(v[2]) for v in util.read_lattice_ark(str(d / "lat_min.ark")).values())
    n_det = sum(len(v[2]) for v in util.read_lattice_ark(str(d / "lat_det.ark")).values())
    assert 0 < n_min <= n_det and (d / "w_min.txt").read_text() == (d / "w_det.txt").read_text()
    mine_w, mine_a = (d / "words.txt").read_text(), (d / "ali.txt").read_text()
    if have_ref("gmm-latgen-faster"):
        ref("gmm-latgen-faster", *opts, str(d / "final.mdl"), str(d / "HCLG.fst"), "ark:%s" % (d / "feats.ark"), "ark:%s" % (d / "r_lat.ark"),
            "ark,t:%s" % (d / "r_words.txt"), "ark,t:%s" % (d / "r_ali.txt"))
        assert mine_w.split() == (d / "r_words.txt").read_text().split()
        assert mine_a.split() == (d / "r_ali.txt").read_text().split()
        if have_ref("lattice-copy"):
            # our raw-lattice archive must be readable by the reference and describe a lattice with
            # the same best path cost as the reference's (the reference's may hold extra arcs)
            ref("lattice-copy", "ark:%s" % (d / "lat.ark"), "ark,t:%s" % (d / "lat.txt"))
            assert "uttA" in (d / "lat.txt").read_text()
    else:
        assert "uttA" in mine_w and "uttA" in mine_a


def test_gmm_decode_faster_vs_reference_program(host, tmp_path):
    """the same files through GmmDecodeFaster (FasterDecoder on the GPU) and the reference's
    gmm-decode-faster: words and alignments must agree"""
    d = tmp_path
    g = load("files")
    (d / "final.mdl").write_bytes(g["file_final_mdl"].tobytes())
    (d / "HCLG.fst").write_bytes(g["file_HCLG_fst"].tobytes())
    rng = np.random.default_rng(1)
    feats = {"uttA": (rng.standard_normal((40, 39)) * 2).astype(np.float32), "uttB": (rng.standard_normal((25, 39)) * 2).astype(np.float32),
             "uttC": (rng.standard_normal((1, 39)) * 2).astype(np.float32)}
    with open(d / "feats.ark", "wb") as f:
        for k, m in feats.items():
            f.write(k.encode() + b" \0BFM " + bytes([4]) + np.int32(m.shape[0]).tobytes() + bytes([4]) + np.int32(39).tobytes() + m.tobytes())
    opts = ["--beam=13", "--max-active=7000", "--acoustic-scale=0.083333"]
    log = str(d / "dec.log")
    rc = call(host.vb_GmmDecodeFaster, ["gmm-decode-faster"] + opts + [str(d / "final.mdl"), str(d / "HCLG.fst"), "ark:%s" % (d / "feats.ark"),
                                        "ark,t:%s" % (d / "words.txt"), "ark,t:%s" % (d / "ali.txt")], log)
    assert rc == 0, open(log).read()
    assert "Overall log-likelihood per frame" in open(log).read()
    # the 6th argument: the best path as a CompactLattice (GetBestPath's linear lattice after RemoveEpsLocal, acoustic costs un-scaled)
    assert call(host.vb_GmmDecodeFaster, ["gmm-decode-faster"] + opts + [str(d / "final.mdl"), str(d / "HCLG.fst"), "ark:%s" % (d / "feats.ark"),
                                          "ark,t:%s" % (d / "w2.txt"), "ark,t:%s" % (d / "a2.txt"), "ark:%s" % (d / "lat.ark")], log) == 0, open(log).read()
    mine_w, mine_a = (d / "words.txt").read_text(), (d / "ali.txt").read_text()
    assert (d / "w2.txt").read_text() == mine_w and (d / "a2.txt").read_text() == mine_a
    # --admission=ordered|exact (an extension: vbh_faster_cfg::admission); anything else is an error, as a bad option is in the reference
    for adm in ("ordered", "exact"):
        assert call(host.vb_GmmDecodeFaster, ["gmm-decode-faster", "--admission=" + adm] + opts + [str(d / "final.mdl"), str(d / "HCLG.fst"), "ark:%s" % (d / "feats.ark"),
                                              "ark,t:%s" % (d / "w3.txt"), "ark,t:%s" % (d / "a3.txt")], log) == 0, open(log).read()
        assert (d / "w3.txt").read_text() == mine_w and (d / "a3.txt").read_text() == mine_a      # (beam-limited toy case: the rule does not decide)
    assert call(host.vb_GmmDecodeFaster, ["gmm-decode-faster", "--admission=sometimes"] + opts + [str(d / "final.mdl"), str(d / "HCLG.fst"), "ark:%s" % (d / "feats.ark"),
                                          "ark,t:%s" % (d / "w3.txt")], log) == -1
    assert call(host.vb_GmmLatgenFaster, ["gmm-latgen-faster", "--admission=sometimes"] + opts + [str(d / "final.mdl"), str(d / "HCLG.fst"), "ark:%s" % (d / "feats.ark"),
                                          "ark:%s" % (d / "l3.ark")], log) == -1
    from tests import util
    lats = util.read_lattice_ark(str(d / "lat.ark"))
    ali_of = {l.split()[0]: [int(x) for x in l.split()[1:]] for l in mine_a.splitlines()}
    words_of = {l.split()[0]: [int(x) for x in l.split()[1:]] for l in mine_w.splitlines()}
    for k, (_typ, _start, states) in lats.items():
        # a linear chain whose arcs spell the words and whose strings concatenate to the alignment
        labels = [a[0] for st in states for a in st[1] if a[0] != 0]
        tids = [t for st in states for a in st[1] for t in a[3]] + [t for st in states for t in st[0][2]]
        assert labels == words_of[k] and tids == ali_of[k], k
        assert all(len(st[1]) <= 1 for st in states)
    if have_ref("gmm-decode-faster"):
        ref("gmm-decode-faster", *opts, str(d / "final.mdl"), str(d / "HCLG.fst"), "ark:%s" % (d / "feats.ark"),
            "ark,t:%s" % (d / "r_words.txt"), "ark,t:%s" % (d / "r_ali.txt"), "ark:%s" % (d / "r_lat.ark"))
        assert mine_w.split() == (d / "r_words.txt").read_text().split()
        assert mine_a.split() == (d / "r_ali.txt").read_text().split()
        from tests.test_host_lattice import assert_same_archives
        assert_same_archives(str(d / "lat.ark"), str(d / "r_lat.ark"), rtol=2e-4, atol=2e-2)
    else:
        assert "uttA" in mine_w and "uttA" in mine_a


def test_gmm_latgen_faster_determinized_vs_reference_program(host, tmp_path):
    """--determinize-lattice=true (the reference's default): GPU search + host determinization
    (DeterminizeLatticePhonePrunedWrapper restated in host/lattice.cc) against the reference's
    gmm-latgen-faster on a lexicon graph with words.  The beam is wide, so that the raw lattices of
    the two searches hold the same arcs (they can only differ in arcs beyond the final
    beam cutoff); the determinized CompactLattice archives must then have identical structure -- states, arcs,
    word labels, transition-id strings -- and weights equal within the GMM score tolerance (the
    acoustic costs come from two implementations of the log-likelihoods; the determinizer itself is
    bit-exact on equal input, tests/test_host_lattice.py)."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import gen_golden_lat as gg
    from tests import util
    from tests.test_host_lattice import assert_same_archives
    d = tmp_path
    gg.make_case(str(d), seed=41, n_words=80, n_cd=12, frames=(150, 60, 333, 1))
    opts = ["--beam=20", "--lattice-beam=6", "--max-active=7000", "--acoustic-scale=0.083333", "--allow-partial=true"]
    log = str(d / "dec.log")
    files = [str(d / "final.mdl"), str(d / "HCLG.fst"), "ark:%s" % (d / "feats.ark")]
    rc = call(host.vb_GmmLatgenFaster, ["gmm-latgen-faster"] + opts + files + ["ark:%s" % (d / "lat.ark"), "ark,t:%s" % (d / "words.txt"),
                                                                              "ark,t:%s" % (d / "ali.txt")], log)
    assert rc == 0, open(log).read()
    mine = util.read_lattice_ark(str(d / "lat.ark"))
    assert list(mine) == ["uttA", "uttB", "uttC", "uttD"] and all(v[0] == "compactlattice44" for v in mine.values())
    # the table in batches of one utterance (the decoder is kept and grown across batches): the same archive, bit for bit
    assert call(host.vb_GmmLatgenFaster, ["gmm-latgen-faster", "--decode-batch=1"] + opts + files + ["ark:%s" % (d / "lat_b1.ark")], log) == 0, open(log).read()
    assert (d / "lat_b1.ark").read_bytes() == (d / "lat.ark").read_bytes()
    # the best path of the determinized lattice is the decoder's best path
    assert call(host.vb_LatticeBestPath, ["lattice-best-path", "--acoustic-scale=0.083333", "ark:%s" % (d / "lat.ark"), "ark,t:%s" % (d / "bp_words.txt"),
                                          "ark,t:%s" % (d / "bp_ali.txt")], log) == 0
    assert (d / "bp_words.txt").read_text().split() == (d / "words.txt").read_text().split()
    if have_ref("gmm-latgen-faster"):
        ref("gmm-latgen-faster", *opts, *files, "ark:%s" % (d / "r_lat.ark"), "ark,t:%s" % (d / "r_words.txt"), "ark,t:%s" % (d / "r_ali.txt"))
        assert (d / "words.txt").read_text().split() == (d / "r_words.txt").read_text().split()
        assert (d / "ali.txt").read_text().split() == (d / "r_ali.txt").read_text().split()
        assert_same_archives(str(d / "lat.ark"), str(d / "r_lat.ark"), rtol=1e-4, atol=0.05)


def test_compute_mfcc_feats_with_segments(host, tmp_path):
    """--segments (extension: the reference's MakeMfcc pipes extract-segments into compute-mfcc-feats when the data directory has
    a segments file, VB/scr/steps/make_mfcc.cpp:93-190): the pieces of the recordings, with extract-segments' rules for bad
    lines, overshoot and short segments (K/featbin/extract-segments.cc:60-195)"""
    from oracle import vbo
    from voicebridge_amd import synth
    d = tmp_path
    recs = [synth.wave(20, 3.0), synth.wave(21, 2.0)]
    write_wavs(d, recs)                                  # utt00, utt01 are the recordings
    (d / "segments").write_text(
        "segA utt00 0.50 1.75\n"
        "segB utt00 1.75 -1\n"                          # to the end of the recording
        "segC utt01 0.10 2.30\n"                        # 0.3 s beyond the end: truncated (max-overshoot 0.5)
        "segD utt01 0.10 2.90\n"                        # 0.9 s beyond: skipped
        "segE utt01 1.00 1.05\n"                        # shorter than 0.1 s: skipped
        "segF nosuchrec 0.0 1.0\n"                      # unknown recording: skipped
        "segG utt01 1.2 0.4\n"                          # start after end: invalid line
        "segH utt00 0.25 0.95 0\n")                     # explicit channel
    log = str(d / "log.txt")
    assert call(host.vb_ComputeMFCCFeats, ["compute-mfcc-feats", "--dither=0", "--segments=%s" % (d / "segments"), "scp:%s" % (d / "wav.scp"),
                                           "ark:%s" % (d / "seg.ark")], log) == 0, open(log).read()
    got = read_fm_ark(d / "seg.ark")
    assert list(got) == ["segA", "segB", "segC", "segH"]
    text = open(log).read()
    for needle in ("too far out of range", "too short", "Could not find recording nosuchrec", "empty or invalid segment"):
        assert needle in text, needle
    sr = 16000
    cuts = {"segA": recs[0][int(0.50 * sr):int(1.75 * sr)], "segB": recs[0][int(1.75 * sr):], "segC": recs[1][int(0.10 * sr):],
            "segH": recs[0][int(0.25 * sr):int(0.95 * sr)]}
    for k, w in cuts.items():
        ref_ = vbo.mfcc(w.astype(np.float32), vbo.mfcc_opts())
        assert got[k].shape == ref_.shape, k
        np.testing.assert_allclose(got[k], ref_, atol=1e-3, rtol=1e-5)
    if have_ref("extract-segments") and have_ref("compute-mfcc-feats"):
        ref("extract-segments", "scp:%s" % (d / "wav.scp"), str(d / "segments"), "ark:%s" % (d / "r_seg_wav.ark"))
        ref("compute-mfcc-feats", "--dither=0", "ark:%s" % (d / "r_seg_wav.ark"), "ark:%s" % (d / "r_seg.ark"))
        want = read_fm_ark(d / "r_seg.ark")
        assert list(want) == list(got)
        for k in want:
            assert want[k].shape == got[k].shape
            np.testing.assert_allclose(got[k], want[k], atol=1e-3, rtol=1e-5)


def test_compute_mfcc_feats_with_vtln_map(host, tmp_path):
    """--vtln-map [--utt2spk] (K/featbin/compute-mfcc-feats.cc:82-86,127-137; what MakeMfcc passes when the data directory has
    spk2warp or utt2warp, VB/scr/steps/make_mfcc.cpp:63-76): a warp factor per speaker or per utterance, an utterance without
    one is skipped with the reference's warning; the text and the binary form of the table."""
    import struct
    from oracle import vbo
    from voicebridge_amd import synth
    d = tmp_path
    waves = [synth.wave(40 + i, 0.7 + 0.2 * i) for i in range(6)]
    write_wavs(d, waves)
    spk = {"utt%02d" % i: "spk%d" % (i // 2) for i in range(6)}
    (d / "utt2spk").write_text("".join("%s %s\n" % kv for kv in spk.items()))
    (d / "spk2warp").write_text("spk0 0.9\nspk1 1.1\n")                 # no entry for spk2: utt04, utt05 are skipped
    with open(d / "utt2warp.bin", "wb") as f:                            # binary BaseFloat table, one utterance left out
        for i, wv in enumerate((0.85, 0.95, 1.0, 1.05, 1.2)):
            f.write(b"utt%02d \0B\x04" % i + struct.pack("<f", wv))
    log = str(d / "log.txt")
    opts = ["compute-mfcc-feats", "--dither=0", "--vtln-low=60", "--vtln-high=7200"]
    assert call(host.vb_ComputeMFCCFeats, opts + ["--vtln-map=ark:%s" % (d / "spk2warp"), "--utt2spk=ark:%s" % (d / "utt2spk"),
                                                  "scp:%s" % (d / "wav.scp"), "ark:%s" % (d / "a.ark")], log) == 0, open(log).read()
    assert call(host.vb_ComputeMFCCFeats, opts + ["--vtln-map=ark:%s" % (d / "utt2warp.bin"), "scp:%s" % (d / "wav.scp"), "ark:%s" % (d / "b.ark")], log) == 0
    text = open(log).read()
    assert "No vtln-map entry for utterance-id (or speaker-id) utt04" in text and "No vtln-map entry for utterance-id (or speaker-id) utt05" in text
    o = vbo.mfcc_opts()
    o.vtln_low, o.vtln_high = 60.0, 7200.0
    a, b = read_fm_ark(d / "a.ark"), read_fm_ark(d / "b.ark")
    assert list(a) == ["utt00", "utt01", "utt02", "utt03"] and list(b) == ["utt00", "utt01", "utt02", "utt03", "utt04"]
    for got, warps in ((a, [0.9, 0.9, 1.1, 1.1]), (b, [0.85, 0.95, 1.0, 1.05, 1.2])):
        for i, wv in enumerate(warps):
            want = vbo.mfcc(waves[i].astype(np.float32), o, vtln_warp=wv)
            np.testing.assert_allclose(got["utt%02d" % i], want, atol=1e-3, rtol=1e-5)
    assert np.abs(a["utt00"] - b["utt02"][:1].mean()).max() > 0.1      # (the warp does change the features)
    assert np.abs(a["utt02"] - vbo.mfcc(waves[2].astype(np.float32), o)).max() > 0.05
    # --utt2spk without --vtln-map is an error, as upstream
    assert call(host.vb_ComputeMFCCFeats, opts + ["--utt2spk=ark:%s" % (d / "utt2spk"), "scp:%s" % (d / "wav.scp"), "ark:%s" % (d / "c.ark")], log) == -1
    if have_ref("compute-mfcc-feats"):
        ref("compute-mfcc-feats", "--dither=0", "--vtln-low=60", "--vtln-high=7200", "--vtln-map=ark:%s" % (d / "spk2warp"), "--utt2spk=ark:%s" % (d / "utt2spk"),
            "scp:%s" % (d / "wav.scp"), "ark:%s" % (d / "r.ark"))
        want = read_fm_ark(d / "r.ark")
        assert list(want) == list(a)
        for k in want:
            np.testing.assert_allclose(a[k], want[k], atol=1e-3, rtol=1e-5)
