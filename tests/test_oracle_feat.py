"""Pins oracle/vb_oracle_feat.c against the reference: golden vectors produced by the
reference's own C++ (oracle/gen_golden.py) and the reference's HTK fixtures
(kaldi-master/src/feat/feature-mfcc-test.cc:112-615)."""
import os

import numpy as np
import pytest

from oracle import vbo
from tests.util import GOLD, load, mfcc_kwargs, parse_opts, read_htk, read_wav_pcm16

MFCC_CASES = ["mfcc_sine_default", "mfcc_chirp_c0", "mfcc_dc_nosnip", "mfcc_sine_nosnip_c0",
              "mfcc_speechy_htk", "mfcc_speechy_vtln", "mfcc_short_hann", "mfcc_tooshort"]

# float32 FFT + BLAS summation-order noise of the reference itself; cepstra are O(10..100).
# Measured worst case on the noisy signals: 1e-4.
MFCC_ATOL = 1e-3
# A noiseless chirp leaves most mel bins at the float32 FFT noise floor of the reference's
# split-radix transform (srfft.cc), so their logs are implementation noise: measured 0.09.
MFCC_ATOL_CASE = {"mfcc_chirp_c0": 0.25}


@pytest.mark.parametrize("name", MFCC_CASES)
def test_mfcc_matches_reference(name):
    g = load(name)
    kw, vtln = mfcc_kwargs(parse_opts(g["opts"]))
    o = vbo.mfcc_opts(**kw)
    # integer part of the path must be exact (feature-window.cc:30-87)
    assert vbo.num_frames(len(g["wave"]), o) == int(g["numframes"][0]) == g["feats"].shape[0]
    if g["feats"].shape[0] == 0:
        return
    out = vbo.mfcc(g["wave"], o, vtln)
    assert out.shape == g["feats"].shape
    np.testing.assert_allclose(out, g["feats"], atol=MFCC_ATOL_CASE.get(name, MFCC_ATOL), rtol=1e-5)


def test_mfcc_stages_match_reference():
    g = load("mfcc_sine_default")
    o = vbo.mfcc_opts(**mfcc_kwargs(parse_opts(g["opts"]))[0])
    out, win, ps, mel = vbo.mfcc(g["wave"], o, stages=True)
    np.testing.assert_allclose(win, g["win"], rtol=1e-6, atol=1e-3)        # samples are O(1e3)
    np.testing.assert_allclose(ps, g["pspec"], rtol=2e-4, atol=ps.max() * 1e-9)
    np.testing.assert_allclose(mel, g["mel"], rtol=0, atol=1e-4)


HTK_OPTS = {
    1: dict(preemph_coeff=0.0, window_type="hamming", remove_dc_offset=0, low_freq=0.0, htk_mode=1, htk_compat=1, use_energy=0),
    2: dict(preemph_coeff=0.0, window_type="hamming", remove_dc_offset=0, low_freq=0.0, htk_mode=1, htk_compat=1, use_energy=1),
    3: dict(preemph_coeff=0.0, window_type="hamming", remove_dc_offset=0, low_freq=20.0, htk_mode=1, htk_compat=1, use_energy=1),
    4: dict(window_type="hamming", remove_dc_offset=0, low_freq=0.0, htk_mode=1, htk_compat=1, use_energy=1),
    5: dict(window_type="hamming", remove_dc_offset=0, low_freq=0.0, vtln_low=100.0, vtln_high=7500.0, htk_mode=1, htk_compat=1, use_energy=1),
    6: dict(preemph_coeff=0.97, window_type="hamming", remove_dc_offset=0, num_bins=24, low_freq=125.0, high_freq=7800.0, htk_compat=1, use_energy=0),
}


@pytest.mark.parametrize("case", [1, 2, 3, 4, 5, 6])
def test_htk_compare(case):
    """The reference's own known-answer test: MFCC + deltas vs HTK output, abs tol 1.0,
    10 edge frames ignored (feature-mfcc-test.cc:161-175)."""
    sr, wav = read_wav_pcm16(os.path.join(GOLD, "htk", "test.wav"))
    htk = read_htk(os.path.join(GOLD, "htk", "test.wav.fea_htk.%d" % case))
    o = vbo.mfcc_opts(**HTK_OPTS[case])
    raw = vbo.mfcc(wav.astype(np.float32), o, 1.1 if case == 5 else 1.0)
    feats = vbo.add_deltas(raw, 2, 2)
    assert feats.shape == htk.shape
    assert np.abs(feats[10:-10] - htk[10:-10]).max() <= 1.0


@pytest.mark.parametrize("name", ["post_13", "post_9_lda"])
def test_post_matches_reference(name):
    g = load(name)
    o = parse_opts(g["opts"]) if "opts" in g.files else {}
    feats = g["feats"]
    d = vbo.add_deltas(feats, 2, int(o.get("delta_window", 2)))
    np.testing.assert_allclose(d, g["ref_deltas"], rtol=1e-5, atol=1e-5)
    s = vbo.splice(feats, int(o.get("splice_left", 4)), int(o.get("splice_right", 4)))
    np.testing.assert_array_equal(s, g["ref_splice"])            # pure copy: bit-exact
    np.testing.assert_allclose(vbo.acc_cmvn_stats(feats), g["ref_stats"], rtol=1e-12)
    np.testing.assert_allclose(vbo.apply_cmvn(feats, g["cmvn_stats"], False), g["ref_cmvn"], rtol=0, atol=1e-6)
    np.testing.assert_allclose(vbo.apply_cmvn(feats, g["cmvn_stats"], True), g["ref_cmvn_var"], rtol=1e-6, atol=1e-6)
    x = vbo.transform_feats(g["ref_deltas"], g["xform"])
    np.testing.assert_allclose(x, g["ref_xform"], rtol=1e-5, atol=1e-4)


@pytest.mark.parametrize("name", ["post_13", "post_9_lda", "post_300_cm"])
def test_compressed_matrix_bit_exact(name):
    """CompressedMatrix bytes are integer work: must be identical (compressed-matrix.cc:129-187)."""
    g = load(name)
    ref = g["ref_cm"]
    assert bytes(ref[:3]) == b"CM "
    blob = vbo.compress(g["feats"])
    np.testing.assert_array_equal(blob, ref[3:])
    np.testing.assert_array_equal(vbo.uncompress(blob), g["ref_cm_back"])
