"""method=nlls (inference_nlls.cc): the reference's own known-answer tests run with this method
too (test/test_inference.cc:79-238,353-561, parametrised over "vb", "nlls", "spatialvb"), so
they are applied to the CPU oracle (oracle/vb_oracle_nlls.inc) and, on the GPU, to the HIP kernel
(csrc/vb_nlls_kernel.h) - followed by oracle-vs-HIP parity on seeded problems.

The minimiser (FSL MISCMATHS nonlin) is outside the reference's tree: the oracle restates it and
is pinned on these known answers and on an independent SciPy least-squares solution only."""
import numpy as np
import pytest
import scipy.optimize

import cases
import oracle
from fabber_core_amd import hiplib, vbabi


def oracle_engine(h, data, **kw):
    return oracle.run_nlls(h, data, **kw)


def hip_engine(variant):
    """The HIP path with the lane-per-voxel or the wave-per-voxel kernel forced (the automatic
    choice takes the wave kernel below 4096 voxels and where no lane instantiation exists)."""
    def run(h, data, **kw):
        hiplib.set_variant(variant)
        try:
            return hiplib.nlls_run_host(h, data, **kw)
        finally:
            hiplib.set_variant("auto")
    return run


ENGINES = [pytest.param(oracle_engine, id="oracle"), pytest.param(hip_engine("lane"), id="hip-lane", marks=pytest.mark.gpu),
           pytest.param(hip_engine("wave"), id="hip-wave", marks=pytest.mark.gpu)]


def means(res, h):
    P = h.cfg.n_params
    off = P * (P + 1) // 2
    m = res["mvn"][off:off + P].copy()
    for p in range(P):
        if h.cfg.transform[p] != vbabi.TRANSFORM_IDENTITY:
            m[p] = [vbabi.to_model(h.cfg.transform[p], x) for x in m[p]]
    return m


def covariance(res, h):
    P, V = h.cfg.n_params, h.cfg.n_voxels
    cov = np.zeros((P, P, V))
    row = 0
    for r in range(P):
        for c in range(r + 1):
            cov[r, c] = cov[c, r] = res["mvn"][row]
            row += 1
    return cov


# ---- the reference's known answers ------------------------------------------------------------
@pytest.mark.parametrize("engine", ENGINES)
def test_one_param_one_voxel_one_timeslice(engine):
    """test_inference.cc:79-106: only NLLS can fit one sample with one parameter."""
    h = vbabi.build_config(vbabi.MODEL_POLY, 1, 1, degree=0)
    res = engine(h, np.full((1, 1), cases.VAL, dtype=np.float32))
    assert cases.float_eq(means(res, h)[0, 0], cases.VAL)
    assert res["status"][0] == 0


@pytest.mark.parametrize("engine", ENGINES)
@pytest.mark.parametrize("n_voxels", [1, 125])
def test_constant_data(engine, n_voxels):
    """test_inference.cc:108-186."""
    h = vbabi.build_config(vbabi.MODEL_POLY, n_voxels, 10, degree=0)
    res = engine(h, np.full((10, n_voxels), cases.VAL, dtype=np.float32))
    m = means(res, h)
    assert all(cases.float_eq(x, cases.VAL) for x in m[0])
    assert np.all(res["status"] == 0)
    assert np.all(res["mvn"][-1] == 1.0)


@pytest.mark.parametrize("engine", ENGINES)
def test_alternating_data(engine):
    """test_inference.cc:190-238."""
    h = vbabi.build_config(vbabi.MODEL_POLY, 125, 10, degree=0)
    data = np.empty((10, 125), dtype=np.float32)
    data[0::2] = cases.VAL
    data[1::2] = cases.VAL * np.float32(3)
    m = means(engine(h, data), h)
    assert all(cases.float_eq(x, cases.VAL * np.float32(2)) for x in m[0])


@pytest.mark.parametrize("engine", ENGINES)
@pytest.mark.parametrize("lm", [False, True])
def test_polynomial_fit(engine, lm):
    """test_inference.cc:353-429: cubic, degree 3 -> coefficients within 1e-3."""
    val = 2.0
    h = vbabi.build_config(vbabi.MODEL_POLY, 125, 10, degree=3)
    m = means(engine(h, cases.cubic_data(125, 10, val), lm=lm), h)
    assert np.all(np.abs(m[0] - val) < 1e-3) and np.all(np.abs(m[1]) < 1e-3)
    assert np.all(np.abs(m[2] - 1.5 * val) < 1e-3) and np.all(np.abs(m[3] + 2 * val) < 1e-3)


@pytest.mark.parametrize("engine", ENGINES)
def test_masked_timepoints(engine):
    """test_inference.cc:485-561."""
    val = np.float32(2)
    data = np.full((10, 125), val, dtype=np.float32)
    h = vbabi.build_config(vbabi.MODEL_POLY, 125, 10, degree=1)
    assert np.all(np.abs(means(engine(h, data), h)[0] - val) < 1e-3)
    data[2] = val * 2
    data[6] = val * 2
    assert np.all(means(engine(h, data), h)[0] > val)
    h = vbabi.build_config(vbabi.MODEL_POLY, 125, 10, degree=1, masked_timepoints=(3, 7))
    res = engine(h, data)
    assert np.all(np.abs(means(res, h)[0] - val) < 1e-3)
    assert np.all(np.abs(means(res, h)[1]) < 1e-3)


# ---- against an independent solver -------------------------------------------------------------
@pytest.mark.parametrize("engine", ENGINES)
def test_exponential_fit_matches_scipy_least_squares(engine):
    """Means = the least-squares solution; covariance = mse (J'J)^-1 (inference_nlls.cc:160-173)."""
    h, data = cases.exp_problem(16, 50, 1, 0.04, seed=4, noise_sd=0.05)
    res = engine(h, data)
    assert np.all(res["status"] == 0)
    P = 2
    theta = res["mvn"][3:5]      # Fabber space: log amp, log rate
    cov = covariance(res, h)
    t = np.arange(50) * 0.04
    for v in range(16):
        y = data[:, v].astype(np.float64)
        f = lambda q: np.exp(q[0]) * np.exp(-np.exp(q[1]) * t) - y
        sol = scipy.optimize.least_squares(f, [0.0, 0.0], xtol=1e-14, ftol=1e-14, gtol=1e-14)
        assert np.allclose(theta[:, v], sol.x, rtol=0, atol=2e-5), (v, theta[:, v], sol.x)
        assert np.isclose(res["cost"][v], 2 * sol.cost, rtol=1e-8)
        J = sol.jac
        expect = np.linalg.inv(J.T @ J) * (2 * sol.cost / (50 - P))
        assert np.allclose(cov[:, :, v], expect, rtol=2e-3), (v, cov[:, :, v], expect)


def test_non_finite_data_gives_the_uninformative_precision():
    """inference_nlls.cc:186-207: a voxel whose model / Jacobian cannot be evaluated keeps its
    parameters and gets precisions 1e-12 I; with halt_bad_voxel the run stops there."""
    h, data = cases.exp_problem(4, 20, 1, 0.04, seed=1)
    h.cfg.transform[1] = vbabi.TRANSFORM_IDENTITY
    res = oracle.run_nlls(h, data, start=[0.0, -1e6])     # exp(+1e6 t) overflows
    assert np.all(res["status"] != 0)
    assert np.all(res["mvn"][0] == 1e12) and np.all(res["mvn"][1] == 0) and np.all(res["mvn"][2] == 1e12)
    assert np.all(res["mvn"][3] == 0.0) and np.all(res["mvn"][4] == -1e6)
    assert oracle.run_nlls(h, data, start=[0.0, -1e6], halt_bad_voxel=True)["first_bad_voxel"] == 1


# ---- oracle vs HIP --------------------------------------------------------------------------------
def assert_parity(h, data, tol=1e-4, variant="lane", **kw):
    """Means within `tol` of max(|mean|, sd) - the north star's 1e-4 - and far better in practice
    (checked at 1e-6 for 99 % of the voxels). What is NOT compared is the number of iterations: at
    the minimum a step changes the cost by rounding only, so whether `ncf < cf` holds - and with it
    how many damping increases pass before the minimiser stops - is decided by the last bit in
    either implementation; the stopping rule (relative cost change 1e-8) leaves the parameters
    themselves defined to ~1e-4 sd."""
    ref = oracle.run_nlls(h, data, **kw)
    got = hip_engine(variant)(h, data, **kw)
    assert np.array_equal(ref["status"], got["status"])
    ok = ref["status"] == 0
    P = h.cfg.n_params
    off = P * (P + 1) // 2
    sd = np.sqrt(np.abs(np.stack([ref["mvn"][p * (p + 1) // 2 + p] for p in range(P)])))
    scale = np.maximum(np.abs(ref["mvn"][off:off + P]), sd)
    err = (np.abs(got["mvn"][off:off + P] - ref["mvn"][off:off + P]) / np.maximum(scale, 1e-300))[:, ok]
    assert err.max() < tol, err.max()
    assert np.quantile(err.max(axis=0), 0.99) < 1e-6, np.quantile(err.max(axis=0), 0.99)
    assert np.allclose(got["cost"][ok], ref["cost"][ok], rtol=1e-7, atol=1e-12)
    # covariance entries against the scale of their row and column
    row = 0
    for r in range(P):
        for c in range(r + 1):
            d = np.abs(got["mvn"][row] - ref["mvn"][row])[ok] / (sd[r] * sd[c])[ok]
            assert d.max() < 1e-3 and np.quantile(d, 0.99) < 1e-5, (r, c, d.max())
            row += 1
    return ref, got


@pytest.mark.gpu
@pytest.mark.parametrize("variant", ["lane", "wave"])
@pytest.mark.parametrize("lm", [False, True])
def test_parity_polynomial(lm, variant):
    h, data = cases.poly_problem(512, 10, 2, seed=20260101)
    assert_parity(h, data, lm=lm, variant=variant)


@pytest.mark.gpu
@pytest.mark.parametrize("variant", ["lane", "wave"])
def test_parity_linear_model_with_masked_timepoints(variant):
    h, data = cases.linear_problem(300, 200, seed=20260104, masked_timepoints=(5, 17, 100))
    assert_parity(h, data, variant=variant)


@pytest.mark.gpu
@pytest.mark.parametrize("variant", ["lane", "wave"])
@pytest.mark.parametrize("lm", [False, True])
def test_parity_single_exponential(lm, variant):
    h, data = cases.exp_problem(2048, 50, 1, 0.04, seed=20260102)
    assert_parity(h, data, lm=lm, variant=variant)


@pytest.mark.gpu
def test_biexponential_population():
    """Started from identical exponentials (all-zero Fabber-space start) the Gauss-Newton matrix
    is singular and the first steps are decided by rounding: compare what the fit achieves, the
    final cost, over the population instead of per-voxel paths."""
    h, data = cases.exp_problem(2048, 100, 2, 0.02, seed=20260103)
    ref = oracle.run_nlls(h, data)
    got = hiplib.nlls_run_host(h, data)
    ok = (ref["status"] == 0) & (got["status"] == 0)
    assert ok.mean() > 0.99
    assert abs(np.median(got["cost"][ok]) / np.median(ref["cost"][ok]) - 1) < 0.01
    assert np.mean(np.isclose(got["cost"][ok], ref["cost"][ok], rtol=1e-3)) > 0.9


@pytest.mark.gpu
def test_non_finite_model_on_the_gpu():
    h, data = cases.exp_problem(70, 20, 1, 0.04, seed=1)
    h.cfg.transform[1] = vbabi.TRANSFORM_IDENTITY
    ref = oracle.run_nlls(h, data, start=[0.0, -1e6])
    for variant in ("lane", "wave"):
        got = hip_engine(variant)(h, data, start=[0.0, -1e6])
        assert np.array_equal(ref["status"] != 0, got["status"] != 0)
        assert np.array_equal(ref["mvn"], got["mvn"])


@pytest.mark.gpu
def test_parameter_counts_without_a_lane_kernel_run_on_the_wave_kernel():
    rng = np.random.default_rng(5)
    T, P, V = 64, 9, 200
    t = np.arange(T)
    X = np.stack([np.cos(np.pi * (t + 0.5) * k / T) for k in range(P)], axis=1)
    data = (X @ rng.normal(0, 3, (P, V)) + rng.normal(0, 0.5, (T, V))).astype(np.float32)
    h = vbabi.build_config(vbabi.MODEL_LINEAR, V, T, design=X)
    assert_parity(h, data, variant="auto")


# ---- models evaluated by the caller (fabber_nlls_run_hostmodel_host) --------------------------------
def exp_model_in_numpy(h):
    """examples/fwdmodel_exp.cc:65-85 with the holder's transforms (fwdmodel.cc:375-379): the model a library would hold"""
    T, P, dt = h.cfg.n_times, h.cfg.n_params, h.cfg.model_dopt[0]
    t = np.arange(T) * dt

    def model(params, ids):
        with np.errstate(over="ignore", invalid="ignore"):
            q = np.stack([np.exp(params[:, i]) if h.cfg.transform[i] == vbabi.TRANSFORM_LOG else params[:, i] for i in range(P)], axis=1)
            return sum(q[:, 2 * k:2 * k + 1] * np.exp(-q[:, 2 * k + 1:2 * k + 2] * t[None, :]) for k in range(P // 2))
    return model


def assert_close_to_the_oracle(h, ref, got, tol=1e-6):
    assert np.array_equal(ref["status"] != 0, got["status"] != 0)
    ok = ref["status"] == 0
    P = h.cfg.n_params
    off = P * (P + 1) // 2
    sd = np.sqrt(np.abs(np.stack([ref["mvn"][p * (p + 1) // 2 + p] for p in range(P)])))
    scale = np.maximum(np.abs(ref["mvn"][off:off + P]), sd)
    err = (np.abs(got["mvn"][off:off + P] - ref["mvn"][off:off + P]) / np.maximum(scale, 1e-300))[:, ok]
    assert err.max() < 1e-4 and np.quantile(err.max(axis=0), 0.99) < tol, (err.max(), np.quantile(err.max(axis=0), 0.99))
    assert np.allclose(got["cost"][ok], ref["cost"][ok], rtol=1e-7, atol=1e-12)


@pytest.mark.gpu
@pytest.mark.parametrize("lm", [False, True])
def test_host_evaluated_model_against_the_oracle(lm, monkeypatch):
    """inference_nlls.cc:94-214 works with any FwdModel: here the single exponential evaluated by the caller (NumPy),
    the minimiser on the device one trial point per launch, against the oracle's run of the built-in model; several
    batches per step (FVB_HOSTMODEL_BATCH) and masked timepoints."""
    h, data = cases.exp_problem(700, 50, 1, 0.04, seed=20260110, masked_timepoints=(3, 30))
    ref = oracle.run_nlls(h, data, lm=lm)
    monkeypatch.setenv("FVB_HOSTMODEL_BATCH", "256")
    got = hiplib.nlls_run_hostmodel_host(h, data, exp_model_in_numpy(h), lm=lm)
    assert_close_to_the_oracle(h, ref, got)
    # ... and a starting point where the model overflows: the voxel's catch branch (inference_nlls.cc:186-207)
    h2, d2 = cases.exp_problem(70, 20, 1, 0.04, seed=1)
    h2.cfg.transform[1] = vbabi.TRANSFORM_IDENTITY
    ref = oracle.run_nlls(h2, d2, start=[0.0, -1e6])
    got = hiplib.nlls_run_hostmodel_host(h2, d2, exp_model_in_numpy(h2), start=[0.0, -1e6])
    assert np.array_equal(ref["status"] != 0, got["status"] != 0)
    assert np.array_equal(ref["mvn"], got["mvn"])


@pytest.mark.gpu
def test_host_evaluated_model_callback_failure_is_reported():
    h, data = cases.exp_problem(64, 20, 1, 0.04, seed=2)

    def broken(params, ids):
        raise RuntimeError("model library failure")
    with pytest.raises(hiplib.HipEngineError, match="callback failed"):
        hiplib.nlls_run_hostmodel_host(h, data, broken)


@pytest.mark.gpu
def test_twenty_parameters():
    """More parameters than the lane kernels are built for (and than the 16 the engine stopped at before): the
    wave-per-voxel kernels take any count up to FVB_MAX_PARAMS = 32 (the reference has no limit: stated in DESIGN.md)."""
    rng = np.random.default_rng(6)
    T, P, V = 120, 20, 300
    t = np.arange(T)
    X = np.stack([np.cos(np.pi * (t + 0.5) * k / T) for k in range(P)], axis=1)
    data = (X @ rng.normal(0, 3, (P, V)) + rng.normal(0, 0.5, (T, V))).astype(np.float32)
    h = vbabi.build_config(vbabi.MODEL_LINEAR, V, T, design=X)
    assert_parity(h, data, variant="auto")


# ---- through the reference's API (fabber_capi.h, method=nlls; setup.cc:31-33) ------------------
from fabber_core_amd import fabber  # noqa: E402


def test_nlls_is_a_registered_method_with_the_reference_options():
    with fabber.Fabber() as f:
        assert {"vb", "spatialvb", "nlls"} <= set(f.get_methods())
        desc, opts = f.get_options("method", "nlls")
        assert "least squares" in desc
        assert {"vb-init", "lm"} <= {o["name"] for o in opts}


@pytest.mark.gpu
def test_capi_nlls_known_answers():
    """test_inference.cc:108-186,353-429 with method=nlls, through fabber_dorun."""
    shape = (5, 5, 5)
    data = np.full(shape + (10,), cases.VAL, dtype=np.float32)
    out = fabber.run(data, {"model": "poly", "degree": 0, "method": "nlls", "noise": "white", "save-mean": True, "save-mvn": True,
                            "save-std": True, "save-model-fit": True, "save-residuals": True})
    assert all(cases.float_eq(x, cases.VAL) for x in out["mean_c0"].ravel())
    assert out["finalMVN"].shape == shape + (3,) and np.all(out["finalMVN"][..., 2] == 1.0)
    assert "noise_means" not in out and "freeEnergy" not in out
    assert np.allclose(out["modelfit"], data, rtol=1e-6) and np.allclose(out["residuals"], 0, atol=1e-5)
    cubic = cases.cubic_data(125, 10).T.reshape(5, 5, 5, 10, order="F")
    out = fabber.run(cubic, {"model": "poly", "degree": 3, "method": "nlls", "noise": "white", "save-mean": True, "lm": True})
    for name, want in (("mean_c0", 2.0), ("mean_c1", 0.0), ("mean_c2", 3.0), ("mean_c3", -4.0)):
        assert np.all(np.abs(out[name] - want) < 1e-3), name


@pytest.mark.gpu
def test_capi_nlls_matches_the_engine_call_and_honours_masked_timepoints():
    h, data = cases.exp_problem(64, 50, 1, 0.04, seed=9, noise_sd=0.05)
    vol = data.T.reshape(4, 4, 4, 50, order="F")
    opts = {"model": "exp", "num-exps": 1, "dt": 0.04, "method": "nlls", "noise": "white", "save-mean": True, "save-mvn": True,
            "save-var": True}
    out = fabber.run(vol, opts)
    eng = hiplib.nlls_run_host(h, data)
    got = out["finalMVN"].reshape(64, -1, order="F").T
    assert np.allclose(got, eng["mvn"], rtol=1e-6, atol=1e-12)      # (the C ABI hands back float32)
    assert np.allclose(out["mean_amp1"].ravel(order="F"), np.exp(eng["mvn"][3]), rtol=1e-6)
    masked = fabber.run(vol, dict(opts, mt1=3, mt2=7))
    hm, _ = cases.exp_problem(64, 50, 1, 0.04, seed=9, noise_sd=0.05, masked_timepoints=(3, 7))
    eng = hiplib.nlls_run_host(hm, data)
    assert np.allclose(masked["finalMVN"].reshape(64, -1, order="F").T, eng["mvn"], rtol=1e-6, atol=1e-12)
    assert not np.allclose(masked["mean_amp1"], out["mean_amp1"], rtol=1e-9)


@pytest.mark.gpu
def test_capi_nlls_voxel_with_a_nan_sample():
    vol = np.ones((2, 2, 1, 10), dtype=np.float32)
    vol[1, 1, 0, 4] = np.nan
    opts = {"model": "poly", "degree": 1, "method": "nlls", "noise": "white", "save-mean": True, "save-mvn": True}
    # a NaN sample makes every cost comparison false: the minimiser gives up, the precision is not
    # finite and the voxel gets the uninformative one - no exception in the reference either
    out = fabber.run(vol, opts)
    assert np.isclose(out["mean_c0"][0, 0, 0], 1.0, atol=1e-6)
    assert out["finalMVN"][1, 1, 0, 0] == np.float32(1e12)
