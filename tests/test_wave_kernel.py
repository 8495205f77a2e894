"""The wave-per-voxel kernel (vb_wave_kernel.h: Jacobian in LDS, lanes over timepoints) against
the CPU oracle: the cases only it can run (several noise precisions, parameter counts without a
lane instantiation, long series) and the voxelwise feature set re-run with the variant forced."""
import numpy as np
import pytest

import cases
import hipengine
import oracle
import parity
from fabber_core_amd import hiplib, vbabi

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _wave_variant():
    assert hiplib.available() and hiplib.device_count() > 0
    hiplib.set_variant("wave")
    yield
    hiplib.set_variant("auto")


def check(h, y, **kw):
    assert hiplib.kernel_name(h) == "wave"
    return parity.strict(h, oracle.run(h, y), hipengine.run(h, y), cpu2=oracle.run_fma(h, y), **kw)


@pytest.mark.parametrize("case", cases.ALL_CASES, ids=lambda c: c.__name__)
def test_reference_known_answers(case):
    case(hipengine.run)


def test_models_against_oracle():
    h, y = cases.poly_problem(512, 10, 2, seed=20260101)
    assert check(h, y, what="C1 poly")["err_means"] < 1e-6
    h, y = cases.exp_problem(1000 - 37, 50, 1, 0.04, seed=20260102, max_iterations=10, need_f=True)
    assert check(h, y, what="C2 exp", check_f=True)["err_means"] < 1e-6
    h, y = cases.linear_problem(300, 200, seed=20260104)
    assert check(h, y, what="linear")["rel_means"] < parity.NORTH_STAR


@pytest.mark.parametrize("variant", ["wave", "lane"])
@pytest.mark.parametrize("pattern", ["12", "123", "1122"])
def test_noise_patterns_with_several_precisions(pattern, variant):
    """noise-pattern (noisemodel_white.cc:166-226): each digit selects the noise precision of
    that timepoint, repeating; here with a masked timepoint as well. Both mappings: the wave kernel and
    the lane kernel with per-precision moments (vb_lane_pattern_kernel.h)."""
    hiplib.set_variant(variant)
    V, T = 400, 24
    rng = np.random.default_rng(5)
    h, y = cases.poly_problem(V, T, 2, seed=31, max_iterations=10, noise_pattern=pattern, masked_timepoints=(5,), need_f=True)
    y = y.astype(np.float64)
    pat = [int(c) for c in pattern]
    for t in range(T):  # a different noise level per group so that the precisions differ
        y[t] += rng.normal(0, 0.05 * pat[t % len(pat)], V)
    if variant == "lane":
        assert hiplib.kernel_name(h) == "lane_phis<poly,3,%d>" % (2 if max(int(c) for c in pattern) == 2 else 4)
        r = parity.strict(h, oracle.run(h, y), hipengine.run(h, y), cpu2=oracle.run_fma(h, y), what="pattern " + pattern,
                          check_f=True, allow_floor=(pattern == "1122"))
    else:
        r = check(h, y, what="pattern " + pattern, check_f=True, allow_floor=(pattern == "1122"))  # four noise precisions: floor 1e-5
    n_phis = max(pat)
    assert h.cfg.n_phis == n_phis
    P = 3
    nm = oracle.run(h, y)["mvn"][(P + n_phis) * (P + n_phis + 1) // 2 + P:][:n_phis]
    assert nm.shape[0] == n_phis and np.all(np.median(nm, axis=1)[:-1] > np.median(nm, axis=1)[1:])


def test_parameter_counts_without_a_lane_instantiation():
    # 8 regressors: a cosine basis (well conditioned; a monomial basis of this size is not, and
    # then even two CPU builds of the oracle disagree)
    V, T = 300, 30
    rng = np.random.default_rng(8)
    t = (np.arange(T) + 0.5) / T
    X = np.stack([np.cos(np.pi * k * t) for k in range(8)], axis=1)
    coef = rng.uniform(-2, 2, (8, V))
    h = vbabi.build_config(vbabi.MODEL_LINEAR, V, T, design=X, max_iterations=6, need_f=True)
    y = X @ coef + rng.normal(0, 0.05, (T, V))
    assert h.cfg.n_params == 8
    hiplib.set_variant("auto")
    assert hiplib.kernel_name(h) == "wave"  # (300 voxels: below the size at which "auto" takes lane<linear,8>)
    hiplib.set_variant("wave")
    check(h, y, what="linear P=8", check_f=True)
    # 16 regressors, 400 timepoints: 87 KB of LDS per voxel (above the 64 KB default limit)
    V, T, P = 96, 400, 16
    X = rng.normal(0, 1, (T, P))
    theta = rng.normal(0, 3, (P, V))
    h = vbabi.build_config(vbabi.MODEL_LINEAR, V, T, design=X, max_iterations=5)
    check(h, X @ theta + rng.normal(0, 0.5, (T, V)), what="linear P=16 T=400")
    # three exponentials (P = 6), one iteration from the oracle's state (the fit is chaotic)
    h, y = cases.exp_problem(500, 100, 3, 0.02, seed=3, max_iterations=3)
    state = oracle.run(h, y)["mvn"]
    h1, _ = cases.exp_problem(500, 100, 3, 0.02, seed=3, max_iterations=1, init_mvn=state, need_f=True)
    a, b = oracle.run(h1, y), hipengine.run(h1, y)
    ok = np.isfinite(a["mvn"]).all(axis=0) & (a["status"] == 0)
    e_mean, e_cov, _ = parity.voxel_errors(h1, a, b, ok)
    assert np.median(e_mean) < 1e-7 and np.quantile(e_mean, 0.95) < 1e-4


def test_automatic_choice_between_the_two_mappings():
    hiplib.set_variant("auto")
    small, _ = cases.exp_problem(4095, 50, 1, 0.04, seed=1)
    large, _ = cases.exp_problem(8192, 50, 1, 0.04, seed=1)
    assert hiplib.kernel_name(small) == "wave" and hiplib.kernel_name(large) == "lane<exp,2>"
    # a series too long for the LDS stays on the lane kernel however few voxels there are
    rng = np.random.default_rng(0)
    h = vbabi.build_config(vbabi.MODEL_LINEAR, 64, 6000, design=rng.normal(0, 1, (6000, 4)))
    assert hiplib.kernel_name(h) == "lane<linear,4>"
    # AR(1) noise has lane kernels only
    h, _ = cases.poly_problem(100, 20, 2, seed=1, noise=vbabi.NOISE_AR1)
    assert hiplib.kernel_name(h).startswith("lane_ar1<")


def test_series_too_long_for_lds_fails_loudly():
    V, T, P = 8, 4000, 16
    rng = np.random.default_rng(0)
    h = vbabi.build_config(vbabi.MODEL_LINEAR, V, T, design=rng.normal(0, 1, (T, P)))
    with pytest.raises(hiplib.HipEngineError, match="LDS"):
        hipengine.run(h, rng.normal(0, 1, (T, V)))


@pytest.mark.parametrize("conv", ["pointzeroone", "freduce", "trialmode", "lm"])
def test_free_energy_convergence_detectors(conv):
    V = 1500
    h, y = cases.exp_problem(V, 50, 1, 0.04, seed=7, convergence=conv, max_iterations=30, min_fchange=0.01)
    check(h, y, what=conv, allow_iter_mismatch=V // 200)


def test_priors_transforms_restart_and_history():
    V = 600
    rng = np.random.default_rng(3)
    img = rng.normal(0.5, 0.1, V)
    h, y = cases.poly_problem(V, 20, 2, seed=4, max_iterations=12, need_f=True,
                              param_overrides={"c1": dict(type="I", prec=4.0)}, image_priors={"c1": img})
    check(h, y, check_f=True, what="image prior")
    # (ARD on the cubic polynomial: against the binary128 ground truth, as the lane kernels in tests/test_hip_parity.py)
    for name in ("ARD last", "ARD middle", "masked"):
        parity.cubic_case_against_truth(name, hipengine.run, base_tolerance=False)
    h, y = cases.exp_problem(V, 50, 1, 0.04, seed=9, max_iterations=10,
                             param_overrides={"amp1": dict(transform="S"), "r1": dict(transform="A", mean=1.0, prec=1e-2)})
    check(h, y, what="softplus/abs")
    h, y = cases.exp_problem(V, 50, 1, 0.04, seed=13, max_iterations=5)
    first = oracle.run(h, y)
    h2, _ = cases.exp_problem(V, 50, 1, 0.04, seed=13, max_iterations=5, init_mvn=first["mvn"])
    check(h2, y, what="continue-from-mvn")
    h, y = cases.exp_problem(V, 50, 1, 0.04, seed=11, max_iterations=12, need_f=True, f_history_rows=14)
    a, b = oracle.run(h, y), hipengine.run(h, y)
    parity.strict(h, a, b, check_f=True, what="F history", cpu2=oracle.run_fma(h, y))
    assert np.array_equal(a["f_history_len"], b["f_history_len"])
    Fa, Fb = a["f_history"][:13], b["f_history"][:13]
    # the first iterations take a big step from the initial posterior; F is sensitive there
    assert np.max(np.abs(Fa - Fb) / np.maximum(1.0, np.abs(Fa))) < 1e-5
    assert np.max(np.abs(Fa[4:] - Fb[4:]) / np.maximum(1.0, np.abs(Fa[4:]))) < parity.TOL_F


def test_bad_voxels_are_flagged():
    h, y = cases.exp_problem(256, 50, 1, 0.04, seed=17, max_iterations=10)
    y = y.copy()
    y[7, 5] = np.nan
    y[:, 9] = 0.0
    a, b = oracle.run(h, y), hipengine.run(h, y)
    assert np.array_equal(a["status"], b["status"])
    assert b["status"][5] != 0 and b["status"][9] != 0 and b["setup_failed"][9]
    ok = a["status"] == 0
    e_mean, _, _ = parity.voxel_errors(h, a, b, ok)
    assert e_mean.max() < 1e-6


def test_biexponential_population_and_agreement_with_lane_kernel():
    V = 2000
    h, y = cases.exp_problem(V, 100, 2, 0.02, seed=20260103, max_iterations=50)
    cpu, cpu2, wave = oracle.run(h, y), oracle.run_fma(h, y), hipengine.run(h, y)
    floor = parity.population_stats(h, cpu, cpu2)
    parity.population(h, cpu, wave, floor, what="C3 wave")
    # well-conditioned model: each mapping is held to 1e-6 of the oracle (strict parity), so they are
    # within 2e-6 of each other (measured 1.1e-6 on the worst voxel; the lane kernel forms k'k from
    # moments and advances its exponentials geometrically, this one evaluates everything directly)
    h, y = cases.exp_problem(V, 50, 1, 0.04, seed=2, max_iterations=10, need_f=True)
    wave = hipengine.run(h, y)
    hiplib.set_variant("lane")
    lane = hipengine.run(h, y)
    e_mean, e_cov, _ = parity.voxel_errors(h, lane, wave)
    assert e_mean.max() < 2e-6 and e_cov.max() < 1e-5
    assert np.max(np.abs(lane["free_energy"] - wave["free_energy"]) / np.abs(lane["free_energy"])) < 1e-6
