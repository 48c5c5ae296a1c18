"""GmmAlignCompiled (host/voicebridge_alignbin.cc over the batched FasterDecoder, csrc/vbh_faster.hip) against the REFERENCE's
gmm-align-compiled (tests/golden/align.npz, oracle/gen_golden_align.py: kaldi-master/src/gmmbin/gmm-align-compiled.cc over
AlignUtteranceWrapper, decoder/decoder-wrappers.cc:424-516): 12 utterances, each against its own graph (linear transcripts with
second pronunciations behind epsilon arcs), the reference's alignment options, one utterance that only aligns with the retry
beam and one that cannot align at all.  Inputs are regenerated from seeds; files go through the C ABI of the host library."""
import ctypes as C
import os

import numpy as np
import pytest

from tests.util import load

pytestmark = pytest.mark.gpu


def argv(*a):
    arr = (C.c_char_p * len(a))(*[x.encode() for x in a])
    return len(a), arr


@pytest.fixture(scope="module")
def case(tmp_path_factory):
    from oracle import gen_golden_align as ga
    from tests.test_gpu_config3 import lexicon_graph
    from voicebridge_amd import build
    _g, _t2p, lex = lexicon_graph()
    case_, model = ga.align_case(lex)
    d = str(tmp_path_factory.mktemp("align"))
    ga.write_case(d, case_, model)
    return dict(d=d, fx=load("align"), ga=ga, host=C.CDLL(build.HOST_LIB), keys=sorted(case_))


def run(c, tag, *extra):
    d, ga = c["d"], c["ga"]
    p = lambda n: os.path.join(d, n + "." + tag)
    a = ["gmm-align-compiled"] + list(ga.OPTS) + list(extra) + ["--write-per-frame-acoustic-loglikes=ark,t:" + p("pf"), os.path.join(d, "final.mdl"),
                                                                 "ark:" + os.path.join(d, "graphs.fsts"), "ark:" + os.path.join(d, "feats.ark"),
                                                                 "ark,t:" + p("ali"), "ark,t:" + p("scores")]
    rc = c["host"].vb_GmmAlignCompiled(*argv(*a), p("log").encode())
    return rc, ga.read_int_table(p("ali")), ga.read_float_vec_table(p("scores")), ga.read_float_vec_table(p("pf")), open(p("log")).read()


def check(c, ali, sc, pf, log):
    fx = c["fx"]
    failed, retried = set(str(x) for x in fx["failed"]), set(str(x) for x in fx["retried"])
    assert set(ali) == set(c["keys"]) - failed
    for k in ali:
        np.testing.assert_array_equal(ali[k], fx["ali_" + k], err_msg="alignment of " + k)
        # scores: -(graph + acoustic) of the path; the acoustic part comes from two float pipelines (scores 1e-4 rel)
        np.testing.assert_allclose(sc[k][0], float(fx["score_" + k]), rtol=2e-4)
        np.testing.assert_allclose(pf[k], fx["per_frame_" + k], rtol=2e-4, atol=2e-3)
    for k in retried:
        assert "Retrying utterance %s with beam" % k in log
    for k in failed:
        assert "Did not successfully decode file %s" % k in log
    assert "Retried %d out of %d utterances." % (len(retried), len(c["keys"])) in log
    assert "Done %d, errors on %d" % (int(fx["done_err"][0]), int(fx["done_err"][1])) in log
    over = float(log.split("Overall log-likelihood per frame is ")[1].split()[0])
    np.testing.assert_allclose(over, float(fx["overall"][0]), rtol=2e-4)


def test_alignments_equal_the_reference_program(case):
    rc, ali, sc, pf, log = run(case, "a")
    assert rc == 0
    check(case, ali, sc, pf, log)


def test_batches_of_any_size_give_the_same_files(case):
    """--decode-batch=5: three launches, the retry of each batch on its own"""
    rc, ali, sc, pf, log = run(case, "b", "--decode-batch=5")
    assert rc == 0
    check(case, ali, sc, pf, log)


def test_option_errors_are_the_reference_s(case):
    d = case["d"]
    rc = case["host"].vb_GmmAlignCompiled(*argv("gmm-align-compiled", "--beam=10", "--retry-beam=5", os.path.join(d, "final.mdl"), "ark:" + os.path.join(d, "graphs.fsts"),
                                                "ark:" + os.path.join(d, "feats.ark"), "ark,t:" + os.path.join(d, "x")), os.path.join(d, "log.e").encode())
    assert rc == -1 and "Beams do not make sense" in open(os.path.join(d, "log.e")).read()
    rc = case["host"].vb_GmmAlignCompiled(*argv("gmm-align-compiled", os.path.join(d, "final.mdl")), os.path.join(d, "log.e2").encode())
    assert rc == -1 and "wrong arguments" in open(os.path.join(d, "log.e2")).read()
