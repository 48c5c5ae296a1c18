"""Host: the lattice stage after the search (voicebridge_amd/host/lattice.cc, voicebridge_latbin.cc;
SURVEY.md 8f rank 1 and 3) -- determinization, scaling, word insertion penalty, best path, WER --
against golden archives produced by the reference's own programs (tests/golden/lattices.npz, made
by oracle/gen_golden_lat.py) and, where oracle/_ref/bin is present, against those programs run on
fresh cases.  The bar is bit-exact: same states, arcs, float32 weights and transition-id strings.
No GPU is involved: these functions work on lattice archives."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from tests import util

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REFBIN = os.path.join(ROOT, "oracle", "_ref", "bin")
NEEDED = ("gmm-latgen-faster", "lattice-determinize-phone-pruned", "lattice-determinize-pruned", "lattice-scale",
          "lattice-add-penalty", "lattice-best-path", "compute-wer")


@pytest.fixture(scope="module")
def host():
    from voicebridge_amd import build
    if not os.path.exists(build.HOST_LIB):
        build.build()
    return C.CDLL(build.HOST_LIB)


def call(fn, args, log=""):
    argv = (C.c_char_p * len(args))(*[a.encode() for a in args])
    return fn(len(args), argv, str(log).encode())


def unpack_golden(d):
    g = util.load("lattices")
    for k in g.files:
        if k.startswith("file_"):
            name = k[5:].rsplit("_", 1)
            (d / (name[0] + "." + name[1])).write_bytes(g[k].tobytes())
        elif k.startswith("text_"):
            name = k[5:].rsplit("_", 1)
            (d / (name[0] + "." + name[1])).write_text(str(g[k]))
    return util.parse_opts(g["opts"])


def mine_chain(host, d, o, prefix="m_"):
    """the reference's scoring chain through this build's bin functions"""
    p = lambda n: str(d / n)
    log = p(prefix + "log.txt")
    acwt, beam = "--acoustic-scale=%s" % o["acwt"], "--beam=%s" % o["beam"]
    assert call(host.vb_LatticeDeterminizePhonePruned, ["lattice-determinize-phone-pruned", acwt, beam, p("final.mdl"), "ark:" + p("raw.ark"),
                                                        "ark:" + p(prefix + "det_pp.ark")], log) == 0, open(log).read()
    assert call(host.vb_LatticeDeterminizePruned, ["lattice-determinize-pruned", acwt, beam, "ark:" + p("raw.ark"), "ark:" + p(prefix + "det_p.ark")], log) == 0
    assert call(host.vb_LatticeScale, ["lattice-scale", "--inv-acoustic-scale=%s" % o["lmwt"], "ark:" + p(prefix + "det_pp.ark"),
                                       "ark:" + p(prefix + "scaled.ark")], log) == 0
    assert call(host.vb_LatticeAddPenalty, ["lattice-add-penalty", "--word-ins-penalty=%s" % o["wip"], "ark:" + p(prefix + "scaled.ark"),
                                            "ark:" + p(prefix + "pen.ark")], log) == 0
    assert call(host.vb_LatticeBestPath, ["lattice-best-path", "ark:" + p(prefix + "pen.ark"), "ark,t:" + p(prefix + "best_words.txt"),
                                          "ark,t:" + p(prefix + "best_ali.txt")], log) == 0
    return log


def assert_same_archives(a, b, rtol=0.0, atol=1e-4):
    """rtol = 0: identical, bit for bit.  rtol > 0: identical structure (states, arcs, labels,
    transition-id strings, next states), weights within rtol -- for lattices whose acoustic costs
    come from two implementations of the GMM scores (tolerance 1e-4 relative, tests/test_gpu_gmm.py);
    normalised weights are differences of sums over many frames, hence the absolute floor atol"""
    A, B = util.read_lattice_ark(a), util.read_lattice_ark(b)
    assert list(A) == list(B)
    for k in B:
        assert A[k][0] == B[k][0] and A[k][1] == B[k][1], k
        assert len(A[k][2]) == len(B[k][2]), "%s: %d vs %d states" % (k, len(A[k][2]), len(B[k][2]))
        for s, (x, y) in enumerate(zip(A[k][2], B[k][2])):
            if rtol == 0.0:
                assert x == y, "%s state %d" % (k, s)
                continue
            (xf, xa), (yf, ya) = x, y
            assert xf[2:] == yf[2:] and len(xa) == len(ya), "%s state %d" % (k, s)
            np.testing.assert_allclose(xf[:2], yf[:2], rtol=rtol, atol=atol)
            for u, v in zip(xa, ya):
                assert (u[0], u[3], u[4]) == (v[0], v[3], v[4]), "%s state %d" % (k, s)
                np.testing.assert_allclose(u[1:3], v[1:3], rtol=rtol, atol=atol)


def test_chain_matches_golden(host, tmp_path):
    d = tmp_path
    o = unpack_golden(d)
    mine_chain(host, d, o)
    for fn in ("det_pp.ark", "det_p.ark", "scaled.ark", "pen.ark"):
        assert_same_archives(str(d / ("m_" + fn)), str(d / fn))
    assert (d / "m_best_words.txt").read_text().split() == (d / "best_words.txt").read_text().split()
    assert (d / "m_best_ali.txt").read_text().split() == (d / "best_ali.txt").read_text().split()
    # compute-wer: the three result lines, and the counters the multi-GPU gather sums
    log = str(d / "wer_log.txt")
    assert call(host.vb_ComputeWer, ["compute-wer", "--text", "--mode=present", "ark:" + str(d / "ref_text.txt"), "ark:" + str(d / "m_best_words.txt")], log) == 0
    assert open(log).read() == (d / "wer.txt").read_text()
    c = (C.c_longlong * 8)()
    host.vb_ComputeWerCounts(c)
    assert list(c)[:2] == [3, 6] and c[5] == 3
    # strict mode with a missing hypothesis is an error (-1), "all" scores it as empty
    (d / "hyp2.txt").write_text("uttA 27 12\n")
    assert call(host.vb_ComputeWer, ["compute-wer", "ark:" + str(d / "ref_text.txt"), "ark:" + str(d / "hyp2.txt")], log) == -1
    assert call(host.vb_ComputeWer, ["compute-wer", "--mode=all", "ark:" + str(d / "ref_text.txt"), "ark:" + str(d / "hyp2.txt")], log) == 0
    assert "[PARTIAL]" in open(log).read()


def test_in_memory_tables_between_programs(host, tmp_path):
    """"ark:mem:NAME": a table handed from one bin function to the next without a file (what DecodeFmllr does with its 100 MB temporaries,
    voicebridge_steps.cc): the chain through memory ends in the same bytes as the chain through files; a table that was never
    written, and one that was dropped, are errors like a missing file"""
    d = tmp_path
    o = unpack_golden(d)
    p = lambda n: str(d / n)
    log = p("log.txt")
    acwt, beam = "--acoustic-scale=%s" % o["acwt"], "--beam=%s" % o["beam"]
    assert call(host.vb_LatticeDeterminizePhonePruned, ["lattice-determinize-phone-pruned", acwt, beam, p("final.mdl"), "ark:" + p("raw.ark"), "ark:mem:" + p("det_pp")], log) == 0
    assert not os.path.exists(p("det_pp")) and not os.path.exists("mem:" + p("det_pp"))
    assert call(host.vb_LatticeScale, ["lattice-scale", "--inv-acoustic-scale=%s" % o["lmwt"], "ark:mem:" + p("det_pp"), "ark:mem:" + p("scaled")], log) == 0
    assert call(host.vb_LatticeAddPenalty, ["lattice-add-penalty", "--word-ins-penalty=%s" % o["wip"], "ark:mem:" + p("scaled"), "ark:" + p("m_pen.ark")], log) == 0
    assert_same_archives(p("m_pen.ark"), p("pen.ark"))
    # a raw (state-level) table through memory into the word-level determinization
    assert call(host.vb_LatticeDeterminizePruned, ["lattice-determinize-pruned", acwt, beam, "ark:" + p("raw.ark"), "ark:mem:" + p("det_p")], log) == 0
    assert call(host.vb_LatticeScale, ["lattice-scale", "--inv-acoustic-scale=1", "ark:mem:" + p("det_p"), "ark:" + p("m_det_p.ark")], log) == 0
    assert_same_archives(p("m_det_p.ark"), p("det_p.ark"))
    assert call(host.vb_LatticeScale, ["lattice-scale", "ark:mem:" + p("never_written"), "ark:" + p("x.ark")], log) == -1
    host.vbio_mem_unlink(("mem:" + p("scaled")).encode())
    assert call(host.vb_LatticeAddPenalty, ["lattice-add-penalty", "ark:mem:" + p("scaled"), "ark:" + p("y.ark")], log) == -1
    assert call(host.vb_LatticeScale, ["lattice-scale", "ark:" + p("raw.ark"), "ark,t:mem:" + p("text")], log) == -1      # no text-mode tables in memory


def test_determinized_lattice_properties(host, tmp_path):
    """size-independent properties: the output is deterministic on word labels, its best path has
    the cost of the raw lattice's best path, and every path's alignment has one transition-id per frame"""
    d = tmp_path
    o = unpack_golden(d)
    mine_chain(host, d, o)
    det = util.read_lattice_ark(str(d / "m_det_pp.ark"))
    raw = util.read_lattice_ark(str(d / "raw.ark"))
    acwt = float(o["acwt"])
    for k, (_t, start, states) in det.items():
        for fin, arcs in states:
            labels = [a[0] for a in arcs]
            assert len(labels) == len(set(labels)), "state with two arcs of the same word label"
        # best cost by dynamic programming over the (top-sorted on write) compact lattice and the raw one
        def best(states, cost_arc, cost_final, nxt):
            n = len(states)
            order = list(range(n))
            INF = float("inf")
            dist = [INF] * n
            dist[0] = 0.0
            # relax repeatedly (lattices are acyclic; raw ones are not numbered in order)
            changed = True
            while changed:
                changed = False
                for s in order:
                    if dist[s] == INF:
                        continue
                    for a in states[s][1]:
                        c = dist[s] + cost_arc(a)
                        if c < dist[nxt(a)] - 1e-9:
                            dist[nxt(a)] = c
                            changed = True
            return min(dist[s] + cost_final(states[s][0]) for s in range(n))
        b_det = best(states, lambda a: a[1] + acwt * a[2], lambda f: f[0] + acwt * f[1], lambda a: a[4])
        b_raw = best(raw[k][2], lambda a: a[2] + acwt * a[3], lambda f: f[0] + acwt * f[1], lambda a: a[4])
        assert abs(b_det - b_raw) < 1e-3 * max(1.0, abs(b_raw))
        # frames: follow first arcs
        s, frames = start, 0
        while True:
            fin, arcs = states[s]
            if not arcs:
                frames += len(fin[2])
                break
            frames += len(arcs[0][3])
            s = arcs[0][4]
        n_raw_frames = sum(1 for l in open(d / "best_ali.txt") if l.split()[0] == k for _ in l.split()[1:])
        assert frames == n_raw_frames


def test_errors_are_reported(host, tmp_path):
    log = str(tmp_path / "log.txt")
    assert call(host.vb_LatticeScale, ["lattice-scale", "ark:/nonexistent/x.ark", "ark:" + str(tmp_path / "o.ark")], log) == -1
    assert "ERROR (lattice-scale)" in open(log).read()
    assert call(host.vb_LatticeBestPath, ["lattice-best-path", "--no-such-option=1", "ark:x"], log) == -1
    assert call(host.vb_LatticeDeterminizePruned, ["lattice-determinize-pruned", "--acoustic-scale=0", "ark:a", "ark:b"], log) == -1


def test_damaged_and_script_addressed_archives(host, tmp_path):
    """the archive readers parse a mapping of the file: a truncated or corrupt entry is an error message, never a read past
    the end; entries addressed through an .scp (key path:offset) give the same lattices as the archive itself"""
    d = tmp_path
    o = unpack_golden(d)
    p = lambda n: str(d / n)
    log = p("log.txt")
    acwt, beam = "--acoustic-scale=%s" % o["acwt"], "--beam=%s" % o["beam"]
    raw = open(p("raw.ark"), "rb").read()
    assert call(host.vb_LatticeDeterminizePruned, ["lattice-determinize-pruned", acwt, beam, "ark:" + p("raw.ark"), "ark:" + p("whole.ark")], log) == 0
    det = open(p("whole.ark"), "rb").read()
    rng = np.random.default_rng(0)
    for name, blob, fn, pre in (("raw", raw, host.vb_LatticeDeterminizePruned, ["lattice-determinize-pruned", acwt, beam]),
                                ("det", det, host.vb_LatticeScale, ["lattice-scale", "--inv-acoustic-scale=9"])):
        cuts = sorted(set([5, 9, 20, 40, 60, 90, len(blob) - 1, len(blob) - 7] + [int(x) for x in rng.integers(10, len(blob) - 1, 12)]))
        for cut in cuts:
            open(p("cut.ark"), "wb").write(blob[:cut])
            rc = call(fn, pre + ["ark:" + p("cut.ark"), "ark:" + p("cut_out.ark")], log)
            assert rc != 0 or cut < 6, "%s truncated at %d of %d accepted" % (name, cut, len(blob))      # a cut inside the first key reads as an empty table
        # sizes that claim more than the file holds (state count, arc count, string length fields set to huge values)
        for pos in [int(x) for x in rng.integers(40, len(blob) - 8, 40)]:
            bad = bytearray(blob)
            bad[pos:pos + 4] = (0x7fffff00).to_bytes(4, "little")
            open(p("bad.ark"), "wb").write(bytes(bad))
            call(fn, pre + ["ark:" + p("bad.ark"), "ark:" + p("bad_out.ark")], log)       # any return code; must not crash
    # script-addressed entries: offsets of the objects right after "key "
    offs, pos = [], 0
    lats = util.read_lattice_ark(p("raw.ark"))
    for key in lats:
        at = raw.index(key.encode() + b" ", pos)
        offs.append((key, at + len(key) + 1))
        pos = at + len(key) + 1
    open(p("raw.scp"), "w").write("".join("%s %s:%d\n" % (k, p("raw.ark"), off) for k, off in offs))
    assert call(host.vb_LatticeDeterminizePruned, ["lattice-determinize-pruned", acwt, beam, "scp:" + p("raw.scp"), "ark:" + p("via_scp.ark")], log) == 0, open(log).read()
    assert open(p("via_scp.ark"), "rb").read() == det


@pytest.mark.skipif(not all(os.access(os.path.join(REFBIN, b), os.X_OK) for b in NEEDED), reason="oracle/_ref/bin not built")
@pytest.mark.parametrize("seed,n_words,n_cd,frames,beam", [(11, 60, 10, (200, 31, 1), "6"), (23, 150, 16, (400, 90), "8"), (31, 25, 6, (300,), "4")])
def test_fresh_cases_vs_reference_programs(host, tmp_path, seed, n_words, n_cd, frames, beam):
    import sys
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import gen_golden_lat as gg
    d = tmp_path
    gg.make_case(str(d), seed=seed, n_words=n_words, n_cd=n_cd, frames=frames)
    gg.run_reference_chain(str(d), beam=beam, lmwt="11", wip="1.0")
    mine_chain(host, d, dict(acwt="0.083333", beam=beam, lmwt="11", wip="1.0"))
    for fn in ("det_pp.ark", "det_p.ark", "scaled.ark", "pen.ark"):
        assert_same_archives(str(d / ("m_" + fn)), str(d / fn))
    assert (d / "m_best_words.txt").read_text().split() == (d / "best_words.txt").read_text().split()
    assert (d / "m_best_ali.txt").read_text().split() == (d / "best_ali.txt").read_text().split()
    # and with --minimize=true, at a wider beam (more to merge)
    p = lambda n: str(d / n)
    for prog, fn, extra in (("lattice-determinize-pruned", host.vb_LatticeDeterminizePruned, []),
                            ("lattice-determinize-phone-pruned", host.vb_LatticeDeterminizePhonePruned, [p("final.mdl")])):
        a = ["--acoustic-scale=0.083333", "--beam=%s" % (float(beam) + 3.0), "--minimize=true"] + extra + ["ark:" + p("raw.ark")]
        gg.ref(prog, *a, "ark:" + p("r_min.ark"))
        assert call(fn, [prog] + a + ["ark:" + p("m_min.ark")], p("m_log.txt")) == 0
        assert_same_archives(p("m_min.ark"), p("r_min.ark"))
    # the phone-level pass alone, and no determinization at all (K/lat/determinize-lattice-pruned.cc:1423-1429,1445-1450)
    for flags in (["--word-determinize=false"], ["--phone-determinize=false", "--word-determinize=false"]):
        a = ["--acoustic-scale=0.083333", "--beam=%s" % beam] + flags + [p("final.mdl"), "ark:" + p("raw.ark")]
        gg.ref("lattice-determinize-phone-pruned", *a, "ark:" + p("r_part.ark"))
        assert call(host.vb_LatticeDeterminizePhonePruned, ["lattice-determinize-phone-pruned"] + a + ["ark:" + p("m_part.ark")], p("m_log.txt")) == 0
        assert_same_archives(p("m_part.ark"), p("r_part.ark"))


def test_score_kaldi_wer_on_golden_lattices(host, tmp_path):
    """ScoreKaldiWER (VB/scr/steps/scoring/score_kaldi_wer.cpp) over a decode directory holding the
    reference's determinized lattices: the hypotheses for (LMWT 9, penalty 0.5) are the reference's
    lattice-scale | lattice-add-penalty | lattice-best-path output, the WER file holds the counts
    of the reference's compute-wer on them."""
    d = tmp_path
    unpack_golden(d)
    dec = d / "exp" / "decode"
    dec.mkdir(parents=True)
    (dec / "lat.1").write_bytes((d / "det_pp.ark").read_bytes())
    (dec / "num_jobs").write_text("1\n")
    graph = d / "graph"
    graph.mkdir()
    (graph / "words.txt").write_text("<eps> 0\n" + "".join("%d %d\n" % (i, i) for i in range(1, 1000)))   # identity: symbols are the ids
    data = d / "data"
    data.mkdir()
    (data / "text").write_text((d / "ref_text.txt").read_text())
    utts = [l.split()[0] for l in (d / "ref_text.txt").read_text().split("\n") if l.strip()]
    (data / "utt2spk").write_text("".join("%s spk%d\n" % (u, i % 2) for i, u in enumerate(utts)))
    host.vb_steps_last_error.restype = C.c_char_p
    rc = host.vb_ScoreKaldiWER(str(data).encode(), str(graph).encode(), str(dec).encode(), 2, b"0.0, 0.5", 8, 10)
    assert rc == 0, host.vb_steps_last_error()
    got = (dec / "scoring_kaldi" / "penalty_0.5" / "9.txt").read_text().split()
    assert got == (d / "best_words.txt").read_text().split()
    # "%WER 50.00 [ 3 / 6, 1 ins, 2 del, 0 sub ]", "%SER 66.67 [ 2 / 3 ]" in the reference's one-line form
    assert (dec / "wer_9_0.5").read_text().split() == ["50.00", "3", "6", "1", "2", "0", "66.67", "2", "3"]
    assert len(list(dec.glob("wer_*"))) == 6
    det = dec / "scoring_kaldi" / "wer_details"
    assert det.joinpath("wip").read_text().strip() in ("0.0", "0.5")
    # the statistics of the best combination (score_kaldi_wer.cpp:165-236): per utterance, per speaker, per word pair, bootstrap
    # interval; their totals are those of the best WER file
    best = (dec / "scoring_kaldi" / "best_wer").read_text().split()
    csid = [list(map(int, l.split()[2:])) for l in det.joinpath("per_utt").read_text().split("\n") if " #csid " in l]
    assert len(csid) == int(best[8]) and sum(c[2] for c in csid) == int(best[3]) and sum(c[3] for c in csid) == int(best[4]) and sum(c[1] for c in csid) == int(best[5])
    rows = [l.split() for l in det.joinpath("per_spk").read_text().strip().split("\n")]
    assert rows[-2][:2] == ["SUM", "raw"] and int(rows[-2][3]) == int(best[2]) and int(rows[-2][8]) == int(best[1]) and rows[-1][8] == best[0]
    assert [r[0] for r in rows[1:-2:2]] == ["spk0", "spk1"]
    ops = [l.split() for l in det.joinpath("ops").read_text().strip().split("\n")]
    assert sum(int(o[3]) for o in ops if o[0] != "insertion") == int(best[2]) and sum(int(o[3]) for o in ops if o[0] == "insertion") == int(best[3])
    assert all(o[1] == "'***'" for o in ops if o[0] == "insertion") and [o[0] for o in ops] == sorted(o[0] for o in ops)
    assert det.joinpath("wer_bootci").read_bytes().startswith(b"Set1: WER ") and not list((dec / "scoring_kaldi").glob("penalty_*/*.at"))
    # a missing lattice file is an error
    os.remove(dec / "lat.1")
    assert host.vb_ScoreKaldiWER(str(data).encode(), str(graph).encode(), str(dec).encode(), 2, b"0.0", 8, 8) == -1


def test_text_mode_lattice_archives(host, tmp_path):
    """"ark,t:" lattice tables (K/lat/kaldi-lattice.cc:62-90,94-262,388-415): what this build writes is what the reference's lattice-copy
    writes, byte for byte, and both read each other's text the same way -- CompactLattice and state-level Lattice entries"""
    have = all(os.access(os.path.join(REFBIN, b), os.X_OK) for b in ("lattice-copy", "lattice-determinize-pruned"))
    if not have:
        pytest.skip("oracle/_ref/bin not built here")
    d = tmp_path
    o = unpack_golden(d)
    p = lambda n: str(d / n)
    log = p("log.txt")

    def ref(*a):
        r = subprocess.run([os.path.join(REFBIN, a[0])] + list(a[1:]), capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-400:]
    # ---- CompactLattice: write
    assert call(host.vb_LatticeScale, ["lattice-scale", "--inv-acoustic-scale=1", "ark:" + p("det_pp.ark"), "ark,t:" + p("m_c.txt")], log) == 0, open(log).read()
    ref("lattice-copy", "ark:" + p("det_pp.ark"), "ark,t:" + p("r_c.txt"))
    assert open(p("m_c.txt")).read() == open(p("r_c.txt")).read()
    assert "\t" in open(p("m_c.txt")).read() and "_" in open(p("m_c.txt")).read()
    # ---- CompactLattice: read (the text has 6 significant digits, so compare reader with reader)
    assert call(host.vb_LatticeScale, ["lattice-scale", "--inv-acoustic-scale=1", "ark,t:" + p("r_c.txt"), "ark:" + p("m_c.ark")], log) == 0, open(log).read()
    ref("lattice-copy", "ark,t:" + p("r_c.txt"), "ark:" + p("r_c.ark"))
    assert_same_archives(p("m_c.ark"), p("r_c.ark"))
    # ---- state-level Lattice entries: read (lattice-determinize-pruned takes either kind) and write
    ref("lattice-copy", "--write-compact=false", "ark:" + p("raw.ark"), "ark,t:" + p("r_l.txt"))
    acwt, beam = "--acoustic-scale=%s" % o["acwt"], "--beam=%s" % o["beam"]
    assert call(host.vb_LatticeDeterminizePruned, ["lattice-determinize-pruned", acwt, beam, "ark,t:" + p("r_l.txt"), "ark:" + p("m_l.ark")], log) == 0, open(log).read()
    ref("lattice-determinize-pruned", acwt, beam, "ark,t:" + p("r_l.txt"), "ark:" + p("r_l.ark"))
    assert_same_archives(p("m_l.ark"), p("r_l.ark"))


@pytest.mark.parametrize("case", ["a", "b", "c"])
def test_minimize_equals_the_reference_programs(host, tmp_path, case):
    """--minimize=true (K/lat/push-lattice.cc, K/lat/minimize-lattice.cc behind K/latbin/lattice-determinize-pruned.cc:111-115 and
    K/lat/determinize-lattice-pruned.cc:1460-1466): transition-id strings and weights pushed towards the start, equivalent states
    merged.  Expected: the archives the reference's own programs wrote (tests/golden/lattices_min.npz, oracle/gen_golden_lat_min.py),
    bit for bit; cases a and c lose a quarter / half of their states to the merge."""
    g = util.load("lattices_min")
    p = lambda n: str(tmp_path / n)
    (tmp_path / "raw.ark").write_bytes(g[case + "_raw"].tobytes())
    (tmp_path / "final.mdl").write_bytes(g[case + "_mdl"].tobytes())
    acwt, beam = "--acoustic-scale=%s" % str(g["acwt"]), "--beam=%s" % str(g[case + "_beam"])
    for fn, tag, extra in ((host.vb_LatticeDeterminizePruned, "p", []), (host.vb_LatticeDeterminizePhonePruned, "pp", [p("final.mdl")])):
        (tmp_path / ("ref_%s.ark" % tag)).write_bytes(g["%s_det_%s_min" % (case, tag)].tobytes())
        assert call(fn, ["prog", acwt, beam, "--minimize=true"] + extra + ["ark:" + p("raw.ark"), "ark:" + p("min_%s.ark" % tag)], p("log")) == 0, open(p("log")).read()
        assert_same_archives(p("min_%s.ark" % tag), p("ref_%s.ark" % tag))
        assert call(fn, ["prog", acwt, beam] + extra + ["ark:" + p("raw.ark"), "ark:" + p("det_%s.ark" % tag)], p("log")) == 0
        n_det = sum(len(v[2]) for v in util.read_lattice_ark(p("det_%s.ark" % tag)).values())
        n_min = sum(len(v[2]) for v in util.read_lattice_ark(p("min_%s.ark" % tag)).values())
        assert [n_det, n_min] == g["%s_states_%s" % (case, tag)].tolist()
        # the best path is what it was
        for a in ("det", "min"):
            assert call(host.vb_LatticeBestPath, ["lattice-best-path", acwt, "ark:" + p("%s_%s.ark" % (a, tag)), "ark,t:" + p("w_" + a), "ark,t:" + p("ali_" + a)], p("log")) == 0
        assert open(p("w_det")).read() == open(p("w_min")).read() and open(p("ali_det")).read() == open(p("ali_min")).read()
