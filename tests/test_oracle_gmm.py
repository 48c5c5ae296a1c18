"""Pins oracle/vb_oracle_gmm.c against the reference (DecodableAmDiagGmmUnmapped, diag-gmm.cc)."""
import numpy as np
import pytest

from oracle import vbo
from tests.util import load

# gmm/diag-gmm-test.cc:141 uses rel tol 0.01; sgemv order noise is ~1e-6 relative here
RTOL = 2e-5


@pytest.mark.parametrize("name", ["gmm_p8_d39", "gmm_p64_d40"])
def test_gmm_matches_reference(name):
    g = load(name)
    miv = g["miv"]
    np.testing.assert_allclose(g["means"] * g["inv_vars"], miv, rtol=1e-6)
    gc = vbo.gmm_gconsts(g["weights"], miv, g["inv_vars"])
    fin = np.isfinite(g["gconsts"])
    np.testing.assert_array_equal(np.isneginf(gc), np.isneginf(g["gconsts"]))
    np.testing.assert_allclose(gc[fin], g["gconsts"][fin], rtol=1e-6, atol=1e-4)
    ll = vbo.gmm_loglikes(g["offs"], g["gconsts"], miv, g["inv_vars"], g["feats"])
    np.testing.assert_allclose(ll, g["loglikes"], rtol=RTOL, atol=1e-3)


def test_log_sum_exp_cutoff():
    """kaldi-vector.cc:757-775: terms below max + log(FLT_EPSILON) are dropped."""
    import ctypes as C
    v = np.array([0.0, -15.0, -16.5, -100.0], np.float32)
    got = vbo.lib().vbo_log_sum_exp(v.ctypes.data_as(C.POINTER(C.c_float)), 4, C.c_float(-1.0))
    want = np.float32(np.log(1.0 + float(np.exp(np.float32(-15.0)))))
    assert abs(got - want) < 1e-7
