"""Host-compiled twins of the device building blocks (vb_math.h) against the oracle / numpy.
Runs without a GPU: the same source is compiled for gfx950 inside the kernels."""
import math

import numpy as np
import pytest

import oracle
from fabber_core_amd import hiplib, vbabi

pytestmark = pytest.mark.skipif(not hiplib.available(), reason="libfabber_vb_hip.so not built")


@pytest.mark.parametrize("x", [1e-6, 0.5, 1.0, 4.5, 10.0, 50.0, 52.500001, 99.5, 1000.0])
def test_special_functions(x):
    L = hiplib.lib()
    assert math.isclose(L.fabber_vb_gammaln(x), oracle.lib().oracle_gammaln(x), rel_tol=1e-14, abs_tol=1e-14)
    assert math.isclose(L.fabber_vb_digamma(x), oracle.lib().oracle_digamma(x), rel_tol=1e-14, abs_tol=1e-14)
    # independent check of both against scipy
    from scipy import special
    assert math.isclose(L.fabber_vb_digamma(x), special.digamma(x), rel_tol=1e-12, abs_tol=1e-12)
    # the reference's Lanczos series is only good to ~1e-10 relative
    assert math.isclose(L.fabber_vb_gammaln(x), special.gammaln(x), rel_tol=1e-9, abs_tol=1e-9)


@pytest.mark.parametrize("tr", range(5))
@pytest.mark.parametrize("x", [-3.0, -0.5, 0.25, 0.75, 2.0, 11.0])
def test_transforms(tr, x):
    L, O = hiplib.lib(), oracle.lib()
    pairs = [(0, O.oracle_transform_to_model), (1, O.oracle_transform_to_fabber),
             (2, O.oracle_transform_to_model_var), (3, O.oracle_transform_to_fabber_var)]
    for which, ofn in pairs:
        a, b = L.fabber_vb_transform(which, tr, x), ofn(tr, x)
        assert (math.isnan(a) and math.isnan(b)) or a == b or math.isclose(a, b, rel_tol=1e-15), (which, tr, x, a, b)


@pytest.mark.parametrize("P", [1, 2, 3, 4, 5, 6])
def test_ldl_inverse_matches_lu_and_numpy(P):
    rng = np.random.default_rng(P)
    for trial in range(20):
        B = rng.standard_normal((P + 3, P))
        a = B.T @ B + 1e-3 * np.eye(P)
        inv, logabs, sign, ok = hiplib.ldl_inverse(a)
        assert ok and sign == 1
        ref = np.linalg.inv(a)
        assert np.max(np.abs(inv - ref)) <= 1e-10 * np.max(np.abs(ref))
        sgn, ld = np.linalg.slogdet(a)
        assert abs(logabs - ld) < 1e-10 * max(1, abs(ld))
        # oracle's LU inverse (the restatement of NEWMAT .i())
        oinv = np.zeros((P, P))
        assert oracle.lib().oracle_inverse(P, np.ascontiguousarray(a).ctypes.data, oinv.ctypes.data) == 0
        assert np.max(np.abs(inv - oinv)) <= 1e-10 * np.max(np.abs(ref))


def test_ldl_singular_gets_ridge():
    # exactly singular -> dist_mvn.cc:211-224 retry with 1e-10 on the diagonal
    a = np.zeros((3, 3))
    inv, logabs, sign, ok = hiplib.ldl_inverse(a)
    assert ok
    assert np.allclose(np.diag(inv), 1e10)
    assert not math.isfinite(logabs)  # log|det| of the unridged matrix: F becomes non-finite


def test_ldl_indefinite_sign():
    a = np.diag([2.0, -3.0, 4.0])
    inv, logabs, sign, ok = hiplib.ldl_inverse(a)
    assert ok and sign == -1
    assert math.isclose(logabs, math.log(24.0), rel_tol=1e-14)
    assert np.allclose(inv, np.diag([0.5, -1 / 3.0, 0.25]))


def test_exp_acc_is_good_to_about_half_an_ulp():
    """vb_math.h exp_acc: what the spatial path's pointwise linearisations use instead of the device library's exp
    (1 ulp), because the first central differences of a run amplify every rounding of the model prediction
    (DESIGN 5.4). Against the x87 long double exp (64-bit mantissa): 0.52 ulp at most, glibc's own exp 0.51."""
    if np.finfo(np.longdouble).nmant < 63:
        pytest.skip("no extended precision on this host")
    L = hiplib.lib()
    rng = np.random.default_rng(7)
    xs = np.concatenate([rng.uniform(-40, 40, 150000), rng.uniform(-1, 1, 50000), rng.uniform(-700, 700, 20000),
                         [0.0, -0.0, 1e-300, -1e-300, 689.9, -689.9, 709.0, -745.0, 1e-10, -1e-10]])
    worst = 0.0
    for x in xs:
        got = L.fabber_vb_exp_acc(float(x))
        want = np.exp(np.longdouble(x))
        if not np.isfinite(want) or want == 0 or float(want) < 2.3e-308:
            assert got == float(want) or math.isclose(got, float(want), rel_tol=1e-15)
            continue
        ulp = np.longdouble(math.ulp(float(want)))
        worst = max(worst, float(abs(np.longdouble(got) - want) / ulp))
    assert worst < 0.53, worst
    assert math.isnan(L.fabber_vb_exp_acc(float("nan"))) and L.fabber_vb_exp_acc(float("inf")) == float("inf")
    assert L.fabber_vb_exp_acc(float("-inf")) == 0.0
