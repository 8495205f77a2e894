"""Parity of the HIP voxelwise-VB path (through its C ABI) with the CPU oracle, the reference's
known-answer tests and the reference's stored outputs. Needs a real MI355X: `pytest -m gpu`.

Tolerances are defined and justified in tests/parity.py: strict per-voxel comparison wherever
the reference algorithm is well-conditioned, population statistics against a measured
CPU-vs-CPU floor for the (chaotic) bi-exponential fit, and one-iteration-from-identical-state
comparison for every model so that each formula is still checked per voxel.
"""
import numpy as np
import pytest

import os
import sys

import cases
import golden_utils as gu
import hipengine
import oracle
import parity
from fabber_core_amd import hiplib, vbabi

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def _require_gpu():
    # Fail loudly (not skip) if the native path is not there: a silent fallback is not acceptable
    assert hiplib.available(), "libfabber_vb_hip.so is not built"
    assert hiplib.device_count() > 0, "no HIP device visible"
    # These problems are oracle-sized, i.e. below the voxel count at which "auto" prefers the
    # wave-per-voxel kernel (tests/test_wave_kernel.py); this module is about the lane kernels.
    hiplib.set_variant("lane")
    yield
    hiplib.set_variant("auto")


def both(h, y):
    return oracle.run(h, y), hipengine.run(h, y)


def check(h, y, **kw):
    """Strict per-voxel parity. The second CPU build measures the CPU-vs-CPU floor; a comparison may
    use it (10 x the floor, capped at 1e-4) only where the call says allow_floor=True - the cubic
    polynomial over 20-24 timepoints (columns 1 ... t^3 ~ 1e4: two CPU builds are 3e-6 - 1e-5 apart on
    the means) and a continued single-exponential run; everything else holds the base 1e-6."""
    return parity.strict(h, oracle.run(h, y), hipengine.run(h, y), cpu2=oracle.run_fma(h, y), **kw)


@pytest.mark.parametrize("case", cases.ALL_CASES, ids=lambda c: c.__name__)
def test_reference_known_answers_on_gpu(case):
    case(hipengine.run)


def test_c1_poly_volume():
    """BASELINE config 1: poly degree 2, white noise, T=10, 8x8x8."""
    h, y = cases.poly_problem(512, 10, 2, seed=20260101)
    assert hiplib.kernel_name(h).startswith("lane<poly,3")
    r = check(h, y, what="C1")
    assert r["err_means"] < 1e-6


def test_c2_single_exponential():
    """BASELINE config 2 model (exp, 1 exponential, T=50, dt=0.04, 10 iterations); ragged voxel
    count (not a multiple of the 64-lane wavefront)."""
    h, y = cases.exp_problem(4096 - 37, 50, 1, 0.04, seed=20260102, max_iterations=10)
    assert hiplib.kernel_name(h) == "lane<exp,2>"
    r = check(h, y, what="C2")
    assert r["err_means"] < 1e-6


def test_linear_design_model():
    h, y = cases.linear_problem(1000, 200, seed=20260104)
    r = check(h, y, what="linear")
    assert r["rel_means"] < parity.NORTH_STAR


@pytest.mark.parametrize("k", [0, 1, 2, 3, 5, 8, 13, 21, 34, 49])
def test_c3_biexponential_single_iteration_from_identical_state(k):
    """BASELINE config 3 model. The CPU oracle's state after k iterations is handed to both
    engines (continue-from-mvn) which then do ONE iteration: every formula of the loop is
    compared per voxel without the chaotic amplification of the full trajectory."""
    V = 1024 + 13
    h, y = cases.exp_problem(V, 100, 2, 0.02, seed=20260103, max_iterations=max(k, 1))
    state = oracle.run(h, y)["mvn"] if k > 0 else None
    h1, _ = cases.exp_problem(V, 100, 2, 0.02, seed=20260103, max_iterations=1, init_mvn=state, need_f=True)
    assert hiplib.kernel_name(h1) == "lane<exp,4,F>"
    a, b = both(h1, y)
    ok = np.isfinite(a["mvn"]).all(axis=0) & (a["status"] == 0)
    assert ok.mean() > 0.99
    e_mean, e_cov, _ = parity.voxel_errors(h1, a, b, ok)
    # The comparison is split by the conditioning of the step's posterior precision. Where cond(Lambda)
    # < 1e10 the lane kernel is held to the bound the wave kernel holds everywhere: 99th percentile of the
    # scaled error below 1e-6 (2 x the 99th percentile between the two CPU builds where that is larger:
    # in the very first iterations log r is within 1e-5 of its initial 0, the finite-difference step
    # sits on its 1e-10 floor and the CPU builds are themselves ~1e-6 apart). The rest - voxels passing
    # through astronomically large parameter values around iteration 8, up to ~17 % there - are
    # one-step problems that are themselves ill-conditioned: 99th percentile of the parameter means
    # 5e-3 (or 2 x what the two CPU builds differ by on the same voxels), 90th percentile 1e-5 over
    # everything. The share of each class is printed.
    n = h1.cfg.n_params + 1
    P = h1.cfg.n_params
    ca, ma = oracle.unpack_mvn(a["mvn"][:, ok], n)
    cb, mb = oracle.unpack_mvn(b["mvn"][:, ok], n)
    cond = np.linalg.cond(ca[:, :P, :P])
    well = np.isfinite(cond) & (cond < 1e10)
    floor_mean, _, _ = parity.voxel_errors(h1, a, oracle.run_fma(h1, y), ok)
    sd = np.sqrt(np.abs(np.einsum("vii->vi", ca)))
    e_par = (np.abs(ma - mb) / np.maximum(np.abs(ma), sd))[:, :P].max(axis=1)
    q99 = lambda x: float(np.quantile(x, 0.99)) if len(x) else 0.0
    print("one step from the oracle's state after %d iterations: %.1f %% of the voxels have cond < 1e10; 99th pct of the "
          "error there %.2e (CPU vs CPU %.2e), elsewhere %.2e" % (k, 100 * well.mean(), q99(e_mean[well]), q99(floor_mean[well]),
                                                             q99(e_par[~well])))
    assert well.mean() > 0.75
    assert np.array_equal(a["status"], b["status"])
    assert q99(e_mean[well]) < max(1e-6, 2 * q99(floor_mean[well])), (q99(e_mean[well]), q99(floor_mean[well]))
    assert np.median(e_mean) < max(1e-7, 2 * np.median(floor_mean)), (np.median(e_mean), np.median(floor_mean))
    assert np.quantile(e_mean, 0.90) < 1e-5, np.quantile(e_mean, 0.90)
    cc, mc = oracle.unpack_mvn(oracle.run_fma(h1, y)["mvn"][:, ok], n)
    f_par = (np.abs(ma - mc) / np.maximum(np.abs(ma), sd))[:, :P].max(axis=1)
    assert q99(e_par[~well]) < max(5e-3, 2 * q99(f_par[~well])), (q99(e_par[~well]), q99(f_par[~well]))
    Fa, Fb, Fc = (r["free_energy"][ok][well] for r in (a, b, oracle.run_fma(h1, y)))
    rel_f = lambda x: q99(np.abs(Fa - x) / np.maximum(1, np.abs(Fa)))
    assert rel_f(Fb) < max(1e-6, 2 * rel_f(Fc)), (rel_f(Fb), rel_f(Fc))


@pytest.mark.parametrize("need_f", [False, True])
@pytest.mark.parametrize("variant", ["lane", "wave"])
def test_c3_error_against_the_ground_truth(variant, need_f):
    """BASELINE config 3 model, 50 iterations, 4096 seeded voxels, against the binary128 evaluation of
    the reference algorithm (tests/golden/c3_truth_binary128.npz). The fit is chaotic, so "within 1e-4
    of the CPU" is not a property any fp64 implementation has per voxel (two CPU builds: 73 %); what can
    be measured is each implementation's error against what the algorithm computes exactly. The kernels
    must be NO WORSE than the worse of the two CPU builds - no slack (parity.no_worse_than_the_cpu_builds) -
    in the share of voxels within 1e-4 / 1e-6 of the truth, in the 75th / 90th / 99th percentile of the error
    and in the share of failed voxels, and after 1, 2, 3, 5 iterations (where the rounding noise is amplified
    ~1e5-fold) in the median relative error of the means.
    need_f: the kernels that evaluate the free energy four times per iteration - lane<exp,4,F> is what the
    reference's command line tool runs by default (rundata.cc:221-231) - against the same posterior (the counting
    detector does not look at F) AND against the truth's free energy."""
    import make_c3_truth as mt
    truth = parity.load_c3_truth()
    V = truth["n_voxels"]

    def runs(engine):
        h, y = mt.problem(V, need_f=need_f)
        final = parity.truth_stats(h, truth, engine(h, y), with_f=need_f)
        by_it = {}
        for k, it in enumerate(truth["its"]):
            if it <= 5:
                hk, _ = mt.problem(V, need_f=need_f)
                hk.cfg.max_iterations = it
                by_it[it] = parity.truth_trace_stats(hk, truth["trace_means"][k], engine(hk, y))["median"]
        return final, by_it

    hiplib.set_variant(variant)
    try:
        gpu, gpu_it = runs(hipengine.run)
    finally:
        hiplib.set_variant("lane")
    (c1, c1_it), (c2, c2_it) = runs(oracle.run), runs(oracle.run_fma)
    print("C3 vs binary128 truth [%s, F=%s]: gpu %s | cpu %s | cpu_fma %s | by iteration gpu %s cpu %s cpu_fma %s"
          % (variant, need_f, gpu, c1, c2, gpu_it, c1_it, c2_it))
    parity.no_worse_than_the_cpu_builds(gpu, c1, c2, what="C3 %s F=%s" % (variant, need_f), with_f=need_f)
    for it in gpu_it:
        assert gpu_it[it] <= 1.1 * max(c1_it[it], c2_it[it]), (it, gpu_it[it], c1_it[it], c2_it[it])


def test_c3_status_matches_the_oracle_with_the_same_inverse():
    """Which of the bi-exponential voxels end in a non-finite prediction (status != 0, the voxel stops as
    inference_vb.cc:529-544 prescribes) follows the algorithm that inverts the numerically singular
    precision matrices: the oracle's default restates NEWMAT .i() as LAPACK-style LU and
    loses ~0.1 % of the voxels, the kernels' unpivoted symmetric sweep a few in a hundred thousand. With
    the oracle switched to the sweep as well (oracle.set_inverse) the two implementations must agree:
    measured on 32768 voxels of the benchmark problem - kernel 0 failed, oracle with the sweep 1, oracle
    with LU 24. The failing voxels are the extremes of the chaotic phase of the fit (parameter values
    ~e^700 on their way to overflow), so WHICH voxel crosses the line is as irreproducible as the
    trajectory itself (two CPU builds of the LU oracle do not fail the same voxels either); what is
    asserted is the rate - equal within counting noise with the same inverse, and far below the LU
    oracle's - and, per voxel, that every voxel the sweep oracle carries through is carried through
    by the kernel unless it is one of those few."""
    V = 32768
    h, y = cases.exp_problem(V, 100, 2, 0.02, seed=20260103, max_iterations=50)
    gpu = hipengine.run(h, y)
    lu, lu2 = oracle.run(h, y), oracle.run_fma(h, y)
    oracle.set_inverse("sweep")
    try:
        sweep = oracle.run(h, y)
    finally:
        oracle.set_inverse("lu")
    bad = lambda r: r["status"] != 0
    n_gpu, n_sweep, n_lu = (int(np.count_nonzero(bad(r))) for r in (gpu, sweep, lu))
    both_lu = int(np.count_nonzero(bad(lu) & bad(lu2)))
    print("C3 failed voxels of %d: kernel %d, oracle with the sweep inverse %d, oracle with LU %d (FMA build %d, %d in common)"
          % (V, n_gpu, n_sweep, n_lu, int(np.count_nonzero(bad(lu2))), both_lu))
    assert abs(n_gpu - n_sweep) <= 3 * np.sqrt(max(n_gpu, n_sweep, 1)) + 1
    assert int(np.count_nonzero(bad(gpu) != bad(sweep))) <= n_gpu + n_sweep
    assert n_gpu <= n_lu / 4 and n_lu >= 10


@pytest.mark.parametrize("need_f", [False, True])
def test_c3_biexponential_population(need_f):
    """Full 50-iteration bi-exponential fits: GPU-vs-CPU agreement must match the CPU-vs-CPU
    reproducibility floor of the algorithm (two builds of the oracle)."""
    V = 3000
    h, y = cases.exp_problem(V, 100, 2, 0.02, seed=20260103, max_iterations=50, need_f=need_f)
    cpu, cpu2, gpu = oracle.run(h, y), oracle.run_fma(h, y), hipengine.run(h, y)
    floor = parity.population_stats(h, cpu, cpu2)
    s = parity.population(h, cpu, gpu, floor, what="C3 need_f=%s" % need_f)
    print("C3 population: floor", floor, "gpu", s)


@pytest.mark.parametrize("conv", ["pointzeroone", "freduce", "trialmode", "lm"])
def test_free_energy_convergence_detectors(conv):
    """F-driven detectors on a well-conditioned model: per-voxel iteration counts, save/revert
    and the final F must agree (a voxel whose |dF| sits within rounding of the threshold may stop
    one iteration apart: at most 0.5 % of the voxels)."""
    V = 2000
    h, y = cases.exp_problem(V, 50, 1, 0.04, seed=7, convergence=conv, max_iterations=30, min_fchange=0.01)
    r = check(h, y, what=conv, allow_iter_mismatch=V // 200)
    assert len(np.unique(oracle.run(h, y)["iterations"])) > 1


@pytest.mark.parametrize("conv", ["pointzeroone", "trialmode"])
def test_detectors_on_biexponential_population(conv):
    V = 2000
    h, y = cases.exp_problem(V, 100, 2, 0.02, seed=7, convergence=conv, max_iterations=30, min_fchange=0.01)
    cpu, cpu2, gpu = oracle.run(h, y), oracle.run_fma(h, y), hipengine.run(h, y)
    floor = parity.population_stats(h, cpu, cpu2)
    parity.population(h, cpu, gpu, floor, what=conv)


def test_free_energy_values_and_history():
    h, y = cases.exp_problem(700, 50, 1, 0.04, seed=11, max_iterations=12, need_f=True, f_history_rows=14)
    a, b = both(h, y)
    parity.strict(h, a, b, check_f=True, what="F", cpu2=oracle.run_fma(h, y))
    assert np.array_equal(a["f_history_len"], b["f_history_len"])
    assert np.all(a["f_history_len"] == 13)
    Fa, Fb = a["f_history"][:13], b["f_history"][:13]
    # the first iterations take a big step from the initial posterior: the moments form of
    # k'k loses ~8 digits there (noise scale b off by ~1e-8 relative), visible in F at 1e-5
    assert np.max(np.abs(Fa - Fb) / np.maximum(1.0, np.abs(Fa))) < 1e-5
    assert np.max(np.abs(Fa[4:] - Fb[4:]) / np.maximum(1.0, np.abs(Fa[4:]))) < parity.TOL_F


def test_priors_ard_and_image():
    V = 900
    rng = np.random.default_rng(3)
    img = rng.normal(0.5, 0.1, V)
    opts = dict(param_overrides={"c1": dict(type="I", prec=4.0)}, image_priors={"c1": img})
    h, y = cases.poly_problem(V, 20, 2, seed=4, max_iterations=12, need_f=True, **opts)
    check(h, y, check_f=True, what="image prior")
    # ARD on the LAST parameter (its free-energy term is the one that survives 'Fprior =') and on a middle one (updates
    # the prior, contributes nothing to F) on the cubic polynomial: test_cubic_cases_against_the_ground_truth


def test_transform_overrides():
    h, y = cases.exp_problem(640, 50, 1, 0.04, seed=9, max_iterations=10,
                             param_overrides={"amp1": dict(transform="S"), "r1": dict(transform="A", mean=1.0, prec=1e-2)})
    check(h, y, what="softplus/abs")
    h, y = cases.exp_problem(640, 50, 1, 0.04, seed=9, max_iterations=10,
                             param_overrides={"amp1": dict(transform="I", mean=1.0, prec=1e-2), "r1": dict(transform="I", mean=1.0, prec=1e-2)})
    check(h, y, what="identity")
    h, y = cases.exp_problem(640, 50, 1, 0.04, seed=9, max_iterations=10,
                             param_overrides={"amp1": dict(transform="F", mean=0.5, prec=1.0)})
    check(h, y, what="fractional")


@pytest.mark.parametrize("name", ["ARD last", "ARD middle", "masked", "prior-noise-stddev", "locked-noise-stdev"])
def test_cubic_cases_against_the_ground_truth(name):
    """The cubic polynomial over 20 - 24 timepoints (design columns 1 ... t^3 ~ 1e4) with ARD priors, masked timepoints
    and the noise options: two fp64 CPU builds of the oracle are 3e-6 - 3e-5 apart on single voxels, so a per-voxel
    bound against ONE of them says little (rounds 1 - 3 raised it to a multiple of that distance). Instead every
    implementation is measured against what the algorithm computes - the oracle's statements in binary128
    (tests/golden/make_cubic_truth.py). The kernels turn out to be the closest of the three (worst voxel 3e-9 - 2e-7;
    the CPU builds 2e-7 - 4e-5: the raised bounds absorbed the oracle's LU inverse, not the kernels), so they are held
    to the BASE tolerances against the truth, per voxel; status and iteration counts identical."""
    parity.cubic_case_against_truth(name, hipengine.run)


def test_noise_options_on_a_well_conditioned_problem():
    """masked timepoints, prior-noise-stddev and locked-noise-stdev, strict per voxel against the oracle (a quadratic
    over 12 timepoints; the cubic cases are test_cubic_cases_against_the_ground_truth)"""
    h, y = cases.poly_problem(333, 12, 2, seed=21, max_iterations=15, masked_timepoints=(3, 7, 12), need_f=True)
    check(h, y, check_f=True, what="masked (quadratic)")
    h, y = cases.poly_problem(333, 12, 2, seed=21, max_iterations=15, prior_noise_stddev=0.5)
    check(h, y, what="prior-noise-stddev (quadratic)")
    h, y = cases.poly_problem(333, 12, 2, seed=21, max_iterations=15, locked_noise_stdev=0.07)
    check(h, y, what="locked-noise-stdev (quadratic)")


def test_continue_from_mvn_and_float64_data():
    h, y = cases.exp_problem(500, 50, 1, 0.04, seed=13, max_iterations=5)
    first = oracle.run(h, y)
    h2, _ = cases.exp_problem(500, 50, 1, 0.04, seed=13, max_iterations=5, init_mvn=first["mvn"])
    # (continuing from the CPU's intermediate state restarts the pointwise model evaluation of the first two
    # linearisations on the kernel's side: observed 1.9e-6 where two CPU builds are 5.6e-7 apart - a stated bound of
    # 3e-6 of max(|mean|, sd), 30 x below the north star, instead of a multiple of the floor)
    check(h2, y, what="continue-from-mvn", tol_mean=3e-6)
    y64 = y.astype(np.float64) + 1e-9
    check(h, y64, what="float64 data")


def test_bad_voxels_are_flagged_not_fatal():
    """Non-finite data -> ReCentre's non-finite check (fwdmodel_linear.cc:134-140,174-181);
    a zero series under a LOG transform -> log(0) initial posterior."""
    h, y = cases.exp_problem(256, 50, 1, 0.04, seed=17, max_iterations=10)
    y = y.copy()
    y[:, 5] = 0.0       # data_max = 0 -> amp = log(0) = -inf
    y[:, 77] = -1.0     # data_max < 0 -> log(negative) = NaN
    y[3, 100] = np.nan
    a, b = both(h, y)
    assert a["status"][5] != 0 and a["status"][77] != 0 and a["status"][100] != 0
    assert np.array_equal(a["status"], b["status"])
    assert np.count_nonzero(a["status"]) == 3
    assert np.all(b["setup_failed"][[5, 77]]) and not b["setup_failed"][100]
    parity.strict(h, a, b, what="bad voxels")


@pytest.mark.parametrize("name", ["poly", "linear_vb"])
def test_gpu_reproduces_reference_stored_fixed_point(name):
    ref = gu.load_reference_outdata()
    if name == "poly":
        J, mvn = gu.poly_design(106, 2), ref["poly/finalMVN"].astype(np.float64)
        h = vbabi.build_config(vbabi.MODEL_POLY, 147, 106, degree=2)
    else:
        J, mvn = ref["linear_design"], ref["linear_vb/finalMVN"].astype(np.float64)
        h = vbabi.build_config(vbabi.MODEL_LINEAR, 147, 106, design=J)
    P = J.shape[1]
    cov, means = gu.unpack(mvn, P + 1)
    y = gu.data_with_same_sufficient_statistics(J, cov, means)
    res = hipengine.run(h, y)
    assert np.all(res["status"] == 0)
    got_cov, got_means = gu.unpack(res["mvn"], P + 1)
    sd = np.sqrt(np.einsum("vii->vi", cov))
    assert np.max(np.abs(got_means - means) / np.maximum(sd, np.abs(means))) < 1e-3
    assert np.max(np.abs(got_cov - cov) / (sd[:, :, None] * sd[:, None, :])) < 1e-3
    assert np.max(np.abs(got_means[:, P] / means[:, P] - 1)) < 1e-5


@pytest.mark.parametrize("P", [7, 8])
def test_seven_and_eight_parameters_on_the_lane_kernel(P):
    """vb_lane_wide.hip: design matrices with 7 / 8 regressors (a cosine basis: well conditioned; a monomial
    basis of this size is not, and then even two CPU builds of the oracle disagree) run one lane per voxel -
    with heavy spills, and still 18 - 26 x the wave kernel's rate at volume sizes
    (profiles/r2_lane_vs_wave_wide.jsonl). Strict per-voxel parity, with and without F, ragged voxel count."""
    V, T = 4096 + 21, 40
    rng = np.random.default_rng(8)
    t = (np.arange(T) + 0.5) / T
    X = np.stack([np.cos(np.pi * k * t) for k in range(P)], axis=1)
    coef = rng.uniform(-2, 2, (P, V))
    y = X @ coef + rng.normal(0, 0.05, (T, V))
    for need_f in (False, True):
        h = vbabi.build_config(vbabi.MODEL_LINEAR, V, T, design=X, max_iterations=6, need_f=need_f)
        assert hiplib.kernel_name(h) == "lane<linear,%d%s>" % (P, ",F" if need_f else "")
        check(h, y, what="linear P=%d" % P, check_f=need_f)
    # the polynomial model of this size shares the template (PolyModel's design is the monomials 1 .. t^(P-1), over
    # t = 1..12 cond(J'J) > 1e16: two CPU builds of the oracle end a median 13 posterior sd apart, so there is no
    # value to compare) - the dispatch and a finite result are what is checked
    h, yp = cases.poly_problem(3000, 12, P - 1, seed=77, max_iterations=4)
    assert hiplib.kernel_name(h) == "lane<poly,%d>" % P
    r = hipengine.run(h, yp)
    assert np.isfinite(r["mvn"][:, r["status"] == 0]).all() and (r["status"] == 0).mean() > 0.9


def test_four_exponentials_on_the_lane_kernel():
    """exp(4) = 8 parameters: the fit is chaotic like the bi-exponential one (more so): one iteration from the
    oracle's state per voxel, and population parity of a short run"""
    V = 2000
    rng = np.random.default_rng(9)
    t = np.arange(100) * 0.02
    y = sum(a * np.exp(-r * t[:, None]) for a, r in ((1.0, 0.5), (0.7, 2.0), (0.5, 6.0), (0.3, 15.0))) + rng.normal(0, 0.05, (100, V))
    y = y.astype(np.float32)
    h = vbabi.build_config(vbabi.MODEL_EXP, V, 100, num_exps=4, dt=0.02, max_iterations=3)
    assert hiplib.kernel_name(h) == "lane<exp,8>"
    state = oracle.run(h, y)
    h1 = vbabi.build_config(vbabi.MODEL_EXP, V, 100, num_exps=4, dt=0.02, max_iterations=1, init_mvn=state["mvn"])
    a, b = both(h1, y)
    ok = np.isfinite(a["mvn"]).all(axis=0) & (a["status"] == 0) & (state["status"] == 0)
    assert ok.mean() > 0.95 and np.array_equal(a["status"][ok], b["status"][ok])
    e_mean, _, _ = parity.voxel_errors(h1, a, b, ok)
    floor, _, _ = parity.voxel_errors(h1, a, oracle.run_fma(h1, y), ok)
    print("exp(4) one step: median %.2e (CPU vs CPU %.2e), 90th pct %.2e (%.2e)"
          % (np.median(e_mean), np.median(floor), np.quantile(e_mean, 0.9), np.quantile(floor, 0.9)))
    assert np.median(e_mean) < max(1e-7, 3 * np.median(floor))
    assert np.quantile(e_mean, 0.9) < max(1e-5, 3 * np.quantile(floor, 0.9))


def _echo_noise(y, pattern, seed, sd=0.05):
    """a different noise level per class of the pattern, so that the precisions differ"""
    rng = np.random.default_rng(seed)
    y = y.astype(np.float64)
    pat = [int(c) for c in pattern]
    for t in range(y.shape[0]):
        y[t] += rng.normal(0, sd * pat[t % len(pat)], y.shape[1])
    return y


PATTERN_RUNS = {
    "exp, two precisions, ragged": lambda: (cases.exp_problem(4096 + 77, 50, 1, 0.04, seed=3, max_iterations=8, noise_pattern="12", need_f=True), "12", "lane_phis<exp,2,2>", {}),
    "bi-exponential start, three precisions": lambda: (cases.exp_problem(4200, 60, 2, 0.03, seed=4, max_iterations=2, noise_pattern="1231"), "1231", "lane_phis<exp,4,4>", {}),
    "linear, four precisions, float series": lambda: (cases.linear_problem(5000, 64, seed=5, noise_pattern="1234", max_iterations=5, need_f=True), "1234", "lane_phis<linear,4,4>", dict(as_float=True)),
    "F-reduction detector (save / revert)": lambda: (cases.poly_problem(4500, 24, 2, seed=6, noise_pattern="12", convergence="freduce", max_iterations=12, need_f=True), "12", "lane_phis<poly,3,2>", {}),
    "trial mode + ARD": lambda: (cases.poly_problem(4500, 24, 1, seed=7, noise_pattern="112", convergence="trialmode", max_trials=3, max_iterations=12, need_f=True, param_overrides={"c1": dict(type="A")}), "112", "lane_phis<poly,2,2>", {}),
    "Levenberg-Marquardt": lambda: (cases.exp_problem(4300, 50, 1, 0.04, seed=8, noise_pattern="12", convergence="lm", max_iterations=10, need_f=True), "12", "lane_phis<exp,2,2>", {}),
    "masked timepoints, locked noise": lambda: (cases.poly_problem(4400, 20, 2, seed=9, noise_pattern="12", masked_timepoints=(3, 4, 11), locked_noise_stdev=0.08, max_iterations=5), "12", "lane_phis<poly,3,2>", {}),
}


@pytest.mark.parametrize("name", list(PATTERN_RUNS), ids=lambda s: s.replace(" ", "_"))
def test_several_noise_precisions_on_the_lane_kernel(name):
    """vb_lane_pattern_kernel.h (noise-pattern with 2 .. 4 precisions, noisemodel_white.cc:166-273) at sizes where
    "auto" takes it, against the oracle - strict per-voxel parity, F where asked"""
    (h, y), pattern, kernel, opts = PATTERN_RUNS[name]()
    assert hiplib.kernel_name(h) == kernel
    y = _echo_noise(y, pattern, seed=11)
    if opts.get("as_float"):
        y = y.astype(np.float32)
    # (detectors that watch F: a voxel whose F change sits on the threshold may stop an iteration apart, as in
    # test_convergence_detectors above)
    uses_f = h.cfg.convergence != 0
    check(h, y, what=name, check_f=bool(h.cfg.need_f), allow_floor=True, allow_iter_mismatch=h.cfg.n_voxels // 200 if uses_f else 0)


def test_several_noise_precisions_continue_from_mvn():
    """continue-from-mvn with the noise rows of a P + n_phis posterior (WhiteParams::InputFromMVN per precision)"""
    h, y = cases.poly_problem(4300, 24, 2, seed=12, noise_pattern="123", max_iterations=3)
    y = _echo_noise(y, "123", seed=13)
    first = hipengine.run(h, y)
    ref = oracle.run(h, y)
    parity.strict(h, ref, first, what="first leg")
    h2, _ = cases.poly_problem(4300, 24, 2, seed=12, noise_pattern="123", max_iterations=3, init_mvn=ref["mvn"])
    assert hiplib.kernel_name(h2) == "lane_phis<poly,3,4>"
    check(h2, y, what="continued", allow_floor=True)


def test_postproc_images():
    h, y = cases.exp_problem(777, 100, 2, 0.02, seed=23, max_iterations=10)
    res = hipengine.run(h, y)
    a = oracle.postproc(h, y, res["mvn"])
    b = hiplib.postproc_host(h, y, res["mvn"])
    for k in a:
        assert np.allclose(a[k], b[k], rtol=1e-12, atol=1e-12, equal_nan=True), k


def test_properties_at_full_size():
    """Size-independent properties at BASELINE config-3 size (1e6 voxels, bi-exponential,
    T=100, 50 iterations): (1) voxels are independent, so a sub-block run alone gives
    bit-identical rows; (2) replicated voxels give bit-identical results wherever they sit in
    the volume; (3) a random sample agrees with the CPU oracle as well as two CPU builds agree
    with each other; (4) failures are rare."""
    V = 1_000_000
    h, y = cases.exp_problem(V, 100, 2, 0.02, seed=20260103, max_iterations=50)
    y[:, V - 1] = y[:, 0]
    y[:, 123457] = y[:, 0]
    res = hipengine.run(h, y)
    assert np.mean(res["status"] != 0) < 1e-3
    assert np.all(res["iterations"][res["status"] == 0] == 50)
    assert np.array_equal(res["mvn"][:, 0], res["mvn"][:, V - 1])
    assert np.array_equal(res["mvn"][:, 0], res["mvn"][:, 123457])
    lo, hi = 500_000 - 31, 500_000 + 4097
    hs, _ = cases.exp_problem(hi - lo, 100, 2, 0.02, seed=1, max_iterations=50)
    sub = hipengine.run(hs, y[:, lo:hi])
    assert np.array_equal(sub["mvn"], res["mvn"][:, lo:hi], equal_nan=True)
    idx = np.sort(np.random.default_rng(0).choice(V, 2048, replace=False))
    ho, _ = cases.exp_problem(len(idx), 100, 2, 0.02, seed=1, max_iterations=50)
    ys = np.ascontiguousarray(y[:, idx])
    cpu, cpu2 = oracle.run(ho, ys), oracle.run_fma(ho, ys)
    gpu = {k: (v[:, idx] if v.ndim == 2 else v[idx]) for k, v in res.items() if isinstance(v, np.ndarray) and k != "f_history"}
    floor = parity.population_stats(ho, cpu, cpu2)
    parity.population(ho, cpu, gpu, floor, what="1e6 sample")


def _full_size_properties(h, y, make_small, what, sample=4096):
    """replicated voxels and a sub-block run on its own are bit-identical; a random sample agrees
    per voxel with the CPU oracle (well-conditioned models)"""
    V = h.cfg.n_voxels
    y[:, V - 1] = y[:, 0]
    y[:, V // 3 + 5] = y[:, 0]
    res = hipengine.run(h, y)
    assert np.all(res["status"] == 0)
    assert np.array_equal(res["mvn"][:, 0], res["mvn"][:, V - 1])
    assert np.array_equal(res["mvn"][:, 0], res["mvn"][:, V // 3 + 5])
    lo, hi = V // 2 - 31, V // 2 + 8200
    sub = hipengine.run(make_small(hi - lo), np.ascontiguousarray(y[:, lo:hi]))
    assert np.array_equal(sub["mvn"], res["mvn"][:, lo:hi])
    idx = np.sort(np.random.default_rng(0).choice(V, sample, replace=False))
    hs = make_small(len(idx))
    ys = np.ascontiguousarray(y[:, idx])
    gpu = {k: (v[:, idx] if v.ndim == 2 else v[idx]) for k, v in res.items() if isinstance(v, np.ndarray) and k != "f_history"}
    # (strict() bounds |d mean| by 1e-6 max(|mean|, sd), or 10x the CPU-vs-CPU floor; a purely relative
    # figure is meaningless for the Fabber-space means that sit at ~0, e.g. log r of a rate of 1)
    return parity.strict(hs, oracle.run(hs, ys), gpu, what=what, cpu2=oracle.run_fma(hs, ys))


def test_c2_properties_at_full_size():
    """BASELINE config 2 at its size: 128 x 128 x 64 voxels, single exponential, T = 50."""
    V = 128 * 128 * 64
    h, y = cases.exp_problem(V, 50, 1, 0.04, seed=20260102, max_iterations=10)
    assert hiplib.kernel_name(h) == "lane<exp,2>"
    _full_size_properties(h, y, lambda n: cases.exp_problem(n, 50, 1, 0.04, seed=1, max_iterations=10)[0], "C2 full size")


def test_c4_model_properties_at_two_million_voxels():
    """BASELINE config 4 model (linear design, AR(1) noise, T = 200) at 2e6 voxels (the 256^3 volume
    itself is run by bench.py --workload c4 --voxels 16777216)."""
    V = 2_000_000
    h, y = cases.linear_problem(V, 200, seed=20260104, max_iterations=10, noise=vbabi.NOISE_AR1)
    assert hiplib.kernel_name(h) == "lane_ar1<linear,4>"
    _full_size_properties(h, y, lambda n: cases.linear_problem(n, 200, seed=1, max_iterations=10, noise=vbabi.NOISE_AR1)[0],
                          "C4 model 2e6", sample=2048)


def test_math_building_blocks_as_the_device_compiles_them():
    """The kernels are compiled with floating-point contraction allowed. exp_acc's compensated reduction
    and the frexp-product log-determinant of the symmetric sweep are error-free-transform idioms that contraction can
    change, and tests/test_math_host.py only sees their HOST twin: here the device code itself, against long double /
    NumPy and against the twin."""
    rng = np.random.default_rng(77)
    x = np.concatenate([rng.uniform(-40, 40, 20000), rng.uniform(-1e-3, 1e-3, 2000), [0.0, -0.0, 700.0, -700.0, 709.0, -745.0]])
    got = hiplib.device_math("exp_acc", x)
    truth = np.exp(x.astype(np.longdouble))
    ulp = np.spacing(np.abs(truth.astype(np.float64)))
    err = np.abs((got.astype(np.longdouble) - truth) / ulp.astype(np.longdouble)).astype(np.float64)
    assert err.max() < 0.55, err.max()  # (0.52 ulp measured for the host twin; the device library's exp: ~1 ulp)
    L = hiplib.lib()
    twin = np.array([L.fabber_vb_exp_acc(float(v)) for v in x[:4000]])
    assert np.array_equal(twin, got[:4000])  # the same bits as the host twin
    lib_err = np.abs((hiplib.device_math("exp", x).astype(np.longdouble) - truth) / ulp.astype(np.longdouble)).astype(np.float64)
    assert 0.5 < lib_err.max() <= 1.0 + 1e-9  # (what DESIGN 5.4 says of the device library's exp: a 1-ulp exp)
    special = hiplib.device_math("exp_acc", np.array([np.nan, np.inf, -np.inf]))
    assert np.isnan(special[0]) and special[1] == np.inf and special[2] == 0.0
    # the symmetric sweep and its log-determinant (one logarithm of the pivots' product, exponents apart)
    mats, packed = [], []
    for i in range(500):
        B = rng.standard_normal((7, 4)) * 10.0 ** rng.uniform(-3, 3, 4)
        a = B.T @ B + 1e-9 * np.eye(4)
        mats.append(a)
        packed.append([a[r, c] for r in range(4) for c in range(r + 1)])
    inv, logdet = hiplib.device_math("invert4", np.array(packed))
    for a, ip, ld in zip(mats, inv, logdet):
        ref = np.linalg.inv(a)
        full = np.zeros((4, 4))
        k = 0
        for r in range(4):
            for c in range(r + 1):
                full[r, c] = full[c, r] = ip[k]
                k += 1
        cond = np.linalg.cond(a)
        assert np.max(np.abs(full - ref)) <= 1e-13 * cond * np.max(np.abs(ref)) + 1e-300
        assert abs(ld - np.linalg.slogdet(a)[1]) <= 1e-12 * max(1.0, abs(ld))
        tw_inv, tw_ld, _, ok = hiplib.ldl_inverse(a)
        # (the device's log against glibc's: a few ulp - and the pivots themselves, which the device forms with fused
        # multiply-adds and the host twin without: their relative difference grows with the condition number)
        assert ok and abs(tw_ld - ld) <= 1e-14 * max(1.0, abs(ld)) + 2e-16 * cond, (tw_ld, ld, cond)
