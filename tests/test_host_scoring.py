"""CPU: the per-utterance scoring statistics of ScoreKaldiWER (voicebridge_amd/host/voicebridge_scoring.cc; reference:
VB/scr/steps/scoring/score_kaldi_wer.cpp:165-254 and the programs it runs).  align-text and compute-wer-bootci are pinned to
the reference's own programs (tests/golden/scoring_stats.json, made by oracle/gen_golden_scoring.py; live as well where
oracle/_ref/bin exists); the three table writers are checked on the examples the reference's sources document."""
import ctypes as C
import json
import os
import subprocess

import pytest

from tests.test_gpu_binfuncs import REFBIN, have_ref

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "scoring_stats.json")


@pytest.fixture(scope="module")
def host():
    from voicebridge_amd import build
    if not os.path.exists(build.HOST_LIB):
        pytest.skip("host library not built")
    return C.CDLL(build.HOST_LIB)


def argv(*a):
    arr = (C.c_char_p * len(a))(*[x.encode() for x in a])
    return len(a), arr


def write(path, lines):
    with open(path, "w") as f:
        f.write("".join(l + "\n" for l in lines))


@pytest.mark.parametrize("case", ["a", "b", "c"])
def test_align_text_and_bootstrap_interval_equal_the_reference_programs(host, tmp_path, case):
    g = json.load(open(GOLD))[case]
    write(tmp_path / "ref", g["ref"]); write(tmp_path / "hyp", g["hyp"])
    assert host.vb_AlignText(*argv("align-text", "--special-symbol=" + g["special_symbol"], "ark:%s/ref" % tmp_path, "ark:%s/hyp" % tmp_path,
                                   "ark,t:%s/ali" % tmp_path)) == 0
    assert (tmp_path / "ali").read_text() == g["align_text"]
    assert host.vb_ComputeWerBootci(*argv("compute-wer-bootci", "--mode=" + g["mode"], "ark:%s/ref" % tmp_path, "ark:%s/hyp" % tmp_path,
                                          str(tmp_path / "ci")), str(tmp_path / "log").encode()) == 0
    assert (tmp_path / "log").read_text() == g["bootci_stdout"]
    # the file of the reference's wrapper (VB/src/bin/compute-wer-bootci.cpp:277-288): mean and half-width, 0xB1 between them
    f = g["bootci_stdout"].split()
    mean, lo = float(f[2]), float(f[7].rstrip(","))
    got = (tmp_path / "ci").read_bytes()
    assert got.startswith(b"Set1: WER %.2f%% \xb1 " % mean) and got.endswith(b"%\n")
    assert abs(float(got.split(b"\xb1 ")[1].rstrip(b"%\n")) - (mean - lo)) < 0.011
    if have_ref("align-text"):
        subprocess.run([os.path.join(REFBIN, "align-text"), "--special-symbol=" + g["special_symbol"], "ark:%s/ref" % tmp_path, "ark:%s/hyp" % tmp_path,
                        "ark,t:%s/ali_ref" % tmp_path], check=True, capture_output=True)
        assert (tmp_path / "ali").read_text() == (tmp_path / "ali_ref").read_text()


def test_bootstrap_comparison_of_two_systems(host, tmp_path):
    g = json.load(open(GOLD))["dual"]
    write(tmp_path / "ref", g["ref"]); write(tmp_path / "hyp", g["hyp"]); write(tmp_path / "hyp2", g["hyp2"])
    assert host.vb_ComputeWerBootci(*argv("compute-wer-bootci", "--replications=%d" % g["replications"], "ark:%s/ref" % tmp_path, "ark:%s/hyp" % tmp_path,
                                          str(tmp_path / "ci"), "ark:%s/hyp2" % tmp_path), str(tmp_path / "log").encode()) == 0
    assert (tmp_path / "log").read_text() == g["bootci_stdout"]
    assert (tmp_path / "ci").read_bytes().count(b"\n") == 3
    # strict mode without a hypothesis is an error, as upstream
    write(tmp_path / "hyp_short", g["hyp"][:-1])
    assert host.vb_ComputeWerBootci(*argv("compute-wer-bootci", "ark:%s/ref" % tmp_path, "ark:%s/hyp_short" % tmp_path, ""), str(tmp_path / "log2").encode()) == -1
    assert "strict" in (tmp_path / "log2").read_text()


def test_details_per_utterance_speaker_and_word_pair(host, tmp_path):
    """the example of VB/scr/utils/scoring/wer_per_utt_details.cpp:30-44 (ref A <eps> B C D E / hyp A A B <eps> D X) and a second
    utterance of another speaker"""
    write(tmp_path / "at", ["UTT-A word-A word-A ; <eps> word-A ; word-B word-B ; word-C <eps> ; word-D word-D ; word-E word-X ",
                            "UTT-B a a ; longword b ", "UTT-C "])
    write(tmp_path / "utt2spk", ["UTT-A spkA", "UTT-B speaker-with-a-long-name", "UTT-C spkA", "UTT-D silent"])
    assert host.vb_WerPerUttDetails(str(tmp_path / "at").encode(), str(tmp_path / "per_utt").encode(), b"<eps>") == 0
    assert (tmp_path / "per_utt").read_text().split("\n") == [
        "UTT-A ref  word-A   <eps>  word-B  word-C  word-D  word-E",
        "UTT-A hyp  word-A  word-A  word-B   <eps>  word-D  word-X",
        "UTT-A op      C       I       C       D       C       S  ",
        "UTT-A #csid 3 1 1 1",
        "UTT-B ref  a  longword",
        "UTT-B hyp  a      b   ",
        "UTT-B op   C      S   ",
        "UTT-B #csid 1 1 0 0",
        "UTT-C ref  ", "UTT-C hyp  ", "UTT-C op   ", "UTT-C #csid 0 0 0 0", ""]
    assert host.vb_WerPerSpkDetails(str(tmp_path / "per_utt").encode(), str(tmp_path / "utt2spk").encode(), str(tmp_path / "per_spk").encode()) == 0
    rows = [l.split() for l in (tmp_path / "per_spk").read_text().strip().split("\n")]
    assert rows[0] == ["SPEAKER", "id", "#SENT", "#WORD", "Corr", "Sub", "Ins", "Del", "Err", "S.Err"]
    # speakers in sorted order; one without scored utterances has a raw row of zeros and no percentages
    assert rows[1:] == [["silent", "raw", "0", "0", "0", "0", "0", "0", "0", "0"],
                        ["speaker-with-a-long-name", "raw", "1", "2", "1", "1", "0", "0", "1", "1"],
                        ["speaker-with-a-long-name", "sys", "1", "2", "50.00", "50.00", "0.00", "0.00", "50.00", "100.00"],
                        ["spkA", "raw", "2", "5", "3", "1", "1", "1", "3", "1"],
                        ["spkA", "sys", "2", "5", "60.00", "20.00", "20.00", "20.00", "60.00", "50.00"],
                        ["SUM", "raw", "3", "7", "4", "2", "1", "1", "4", "2"],
                        ["SUM", "sys", "3", "7", "57.14", "28.57", "14.29", "14.29", "57.14", "66.67"]]
    lines = (tmp_path / "per_spk").read_text().split("\n")
    w = len("speaker-with-a-long-name")
    assert lines[0].startswith(" " * (w - 7) + "SPEAKER id  #SENT      #WORD      ") and lines[-3].startswith(" " * (w - 3) + "SUM raw 3          7          ")
    assert host.vb_WerOpsDetails(str(tmp_path / "per_utt").encode(), str(tmp_path / "ops").encode(), b"<eps>") == 0
    assert (tmp_path / "ops").read_text().split("\n") == [
        "correct a a 1", "correct word-A word-A 1", "correct word-B word-B 1", "correct word-D word-D 1",
        "deletion word-C <eps> 1", "insertion <eps> word-A 1", "substitution longword b 1", "substitution word-E word-X 1", ""]
    # an utterance that utt2spk does not know, and a malformed alignment line, are errors
    write(tmp_path / "utt2spk_short", ["UTT-A spkA"])
    assert host.vb_WerPerSpkDetails(str(tmp_path / "per_utt").encode(), str(tmp_path / "utt2spk_short").encode(), str(tmp_path / "x").encode()) == -1
    write(tmp_path / "at_bad", ["UTT-A a a ; b "])
    assert host.vb_WerPerUttDetails(str(tmp_path / "at_bad").encode(), str(tmp_path / "x").encode(), b"<eps>") == -1


def test_word_pairs_by_falling_count(host, tmp_path):
    write(tmp_path / "per_utt", ["u1 ref  a  b  a  c", "u1 hyp  a  x  a  c", "u1 op  C S C C", "u1 #csid 3 1 0 0",
                                 "u2 ref  a  b  ***", "u2 hyp  a  x  c", "u2 #csid 1 1 1 0"])
    assert host.vb_WerOpsDetails(str(tmp_path / "per_utt").encode(), str(tmp_path / "ops").encode(), b"***") == 0
    assert (tmp_path / "ops").read_text().split("\n") == ["correct a a 3", "correct c c 1", "insertion *** c 1", "substitution b x 2", ""]
