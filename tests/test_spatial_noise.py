"""Spatial VB under the noise models beyond "white noise, one precision": AR(1) noise and noise patterns.

Vb::DoCalculationsSpatial calls the noise model through its virtuals (inference_vb.cc:645, :688), and the reference's
own suite runs its AR fit under method=spatialvb (test/test_vb.cc:617-694, instantiated at :847). CPU part: the
oracle's spatial loop with these models is pinned on the voxelwise loop (with non-spatial priors and the counting
detector the two loops are the same arithmetic). GPU part: HIP path against the oracle, strict per voxel, for first-
and second-neighbour priors, with F, failing voxels, continue-from-mvn and locked linearisation centres; the
reference's VbTest.ArNoise through the C ABI under both methods."""
import numpy as np
import pytest

import oracle
import parity
from fabber_core_amd import hiplib, vbabi

gpu = pytest.mark.gpu
AR = vbabi.NOISE_AR1


def masked_volume(shape, seed, keep=0.85):
    rng = np.random.default_rng(seed)
    mask = rng.random(shape) < keep
    return mask, vbabi.grid_coords(shape, mask)


def ar_noise(T, V, rho, sd, seed):
    rng = np.random.default_rng(seed)
    e = rng.normal(0, sd, (T, V))
    for t in range(1, T):
        e[t] += rho * e[t - 1]
    return e


def smooth_line_data(coords, T, seed, rho=0.0, sd=0.2):
    """c0 (smooth in space) + 0.3 t, with white or AR(1) noise"""
    V = coords.shape[1]
    t = np.arange(1, T + 1.0)
    c0 = 2.0 + np.sin(coords[0] / 2.0) * np.cos(coords[1] / 3.0) + 0.2 * coords[2]
    return c0, c0[None, :] + 0.3 * t[:, None] + ar_noise(T, V, rho, sd, seed)


def smooth_exp_data(coords, T, dt, seed, rho=0.0, sd=0.05):
    t = np.arange(T) * dt
    amp = 1.0 + 0.3 * np.sin(coords[0] / 3.0) * np.cos(coords[1] / 4.0) + 0.1 * np.sin(coords[2] / 2.0)
    return amp, amp[None, :] * np.exp(-1.0 * t[:, None]) + ar_noise(T, coords.shape[1], rho, sd, seed)


# ---------------------------------------------------------------------------------------------
# CPU: the oracle's spatial loop with the other noise models
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("noise_kw", [dict(noise=AR), dict(noise_pattern="12"), dict(noise_pattern="1231"),
                                      dict(noise=AR, num_echoes=2, ar_cross_terms="dual")])
def test_oracle_spatial_loop_with_nonspatial_priors_is_the_voxelwise_loop(noise_kw):
    """method=spatialvb with all-N priors (what test_vb.cc:847 runs) is the voxelwise loop with the counting
    detector, whatever the noise model: same posterior, same F."""
    _, coords = masked_volume((6, 5, 4), seed=1)
    V, T = coords.shape[1], 24
    _, y = smooth_line_data(coords, T, seed=2, rho=0.3)
    h = vbabi.build_config(vbabi.MODEL_POLY, V, T, degree=2, max_iterations=7, need_f=True, **noise_kw)
    rs = oracle.run_spatial(h, vbabi.SpatialHolder(coords), y)
    rv = oracle.run(h, y)
    assert np.all(rs["status"] == 0) and np.all(rv["status"] == 0)
    # (one-echo AR: the voxelwise oracle is the stencil restatement vb_oracle_ar.inc, the spatial one the general
    # form vb_oracle_arn.inc - the same numbers up to the order of their sums)
    if noise_kw.get("noise") != AR or noise_kw.get("num_echoes", 1) == 2:
        assert np.array_equal(rs["mvn"], rv["mvn"])
        assert np.array_equal(rs["free_energy"], rv["free_energy"])
    else:
        e_mean, e_cov, _ = parity.voxel_errors(h, rv, rs)
        assert e_mean.max() < parity.TOL_MEAN and e_cov.max() < parity.TOL_COV
        assert np.max(np.abs(rs["free_energy"] - rv["free_energy"]) / np.abs(rv["free_energy"])) < parity.TOL_F


@pytest.mark.parametrize("noise_kw", [dict(noise=AR), dict(noise_pattern="12")])
def test_oracle_spatial_prior_smooths_under_other_noise_models(noise_kw):
    _, coords = masked_volume((8, 7, 6), seed=0)
    V, T = coords.shape[1], 30
    c0, y = smooth_line_data(coords, T, seed=1, rho=0.3, sd=1.5)
    kw = dict(degree=1, max_iterations=10, **noise_kw)
    rs = oracle.run_spatial(vbabi.build_config(vbabi.MODEL_POLY, V, T, param_overrides={"c0": dict(type="M")}, **kw),
                            vbabi.SpatialHolder(coords), y)
    h0 = vbabi.build_config(vbabi.MODEL_POLY, V, T, **kw)
    rv = oracle.run(h0, y)
    n = 2 + h0.n_noise_outputs
    off = n * (n + 1) // 2
    rmse = lambda r: np.sqrt(np.mean((r["mvn"][off] - c0) ** 2))
    assert np.isfinite(rs["mvn"]).all() and np.all(rs["status"] == 0)
    assert rmse(rs) < 0.9 * rmse(rv)


def test_oracle_locked_linearisation_centres():
    """locked-linear-from-mvn (inference_vb.cc:171-181,225-232,695-696): the set-up re-centre uses the given centres
    and the spatial loop never re-centres. For a model that is linear in its parameters the linearisation is exact
    about any centre, so the fit is the unlocked one; for the exponential model it is not."""
    _, coords = masked_volume((5, 5, 4), seed=3)
    V, T = coords.shape[1], 20
    _, y = smooth_line_data(coords, T, seed=4)
    h = vbabi.build_config(vbabi.MODEL_POLY, V, T, degree=1, max_iterations=6, param_overrides={"c0": dict(type="M")})
    free = oracle.run_spatial(h, vbabi.SpatialHolder(coords), y)
    centres = np.stack([np.full(V, 1.5), np.full(V, -0.2)])
    locked = oracle.run_spatial(h, vbabi.SpatialHolder(coords, locked_centres=centres), y)
    assert np.allclose(free["mvn"], locked["mvn"], rtol=1e-6, atol=1e-9)
    _, ye = smooth_exp_data(coords, 40, 0.05, seed=5)
    he = vbabi.build_config(vbabi.MODEL_EXP, V, 40, num_exps=1, dt=0.05, max_iterations=6, param_overrides={"amp1": dict(type="M")})
    free = oracle.run_spatial(he, vbabi.SpatialHolder(coords), ye)
    centres = np.stack([np.full(V, np.log(0.8)), np.full(V, np.log(1.3))])
    locked = oracle.run_spatial(he, vbabi.SpatialHolder(coords, locked_centres=centres), ye)
    assert np.all(locked["status"] == 0) and np.isfinite(locked["mvn"]).all()
    assert np.max(np.abs(free["mvn"][6] - locked["mvn"][6])) > 1e-3  # the means differ: another (fixed) linearisation


# ---------------------------------------------------------------------------------------------
# GPU
# ---------------------------------------------------------------------------------------------
def spatial_check(h, sp, y, what, **kw):
    cpu = oracle.run_spatial(h, sp, y)
    got = hiplib.run_spatial_host(h, sp, y)
    cpu2 = [oracle.run_spatial_fma(h, sp, y)]
    if h.cfg.model == vbabi.MODEL_EXP and kw.get("allow_floor"):
        # (where a test measures the CPU-vs-CPU floor on the exponential model: the build whose exp has the device's
        # accuracy class is one of the CPU builds - DESIGN 5.4, tests/test_random_configs.py)
        cpu2.append(oracle.run_spatial_exp1ulp(h, sp, y))
    for r in [cpu] + cpu2:
        r.setdefault("f_history_len", np.zeros(h.cfg.n_voxels, dtype=np.int32))
    return parity.strict(h, cpu, got, what=what, cpu2=cpu2, **kw), got


NOISES = {"ar": dict(noise=AR), "pattern12": dict(noise_pattern="12"), "pattern1231": dict(noise_pattern="1231"),
          "pattern1234": dict(noise_pattern="1234"), "pattern12345": dict(noise_pattern="1234512345"),
          "pattern8": dict(noise_pattern="1234567812345678"),
          # two interleaved echoes, 2 / 3 / 4 AR coefficients (round 4: the SpArN policy, vb_spatial_noise.h)
          "ar2none": dict(noise=AR, num_echoes=2, ar_cross_terms="none"), "ar2same": dict(noise=AR, num_echoes=2, ar_cross_terms="same"),
          "ar2dual": dict(noise=AR, num_echoes=2, ar_cross_terms="dual")}


def is_ar(noise):
    return noise.startswith("ar")


@gpu
@pytest.mark.parametrize("noise", sorted(NOISES))
@pytest.mark.parametrize("typ", ["M", "m", "P", "p"])
def test_spatial_priors_under_other_noise_models(typ, noise):
    """strict per-voxel parity for every spatial prior type: the split first sweep (M, m) and the per-level launches
    (P, p) read the effective moments the second sweep left, whatever the noise model"""
    _, coords = masked_volume((11, 9, 7), seed=3)
    V, T = coords.shape[1], 40
    _, y = smooth_line_data(coords, T, seed=4, rho=0.3 if is_ar(noise) else 0.0)
    h = vbabi.build_config(vbabi.MODEL_POLY, V, T, degree=1, max_iterations=8, param_overrides={"c0": dict(type=typ)}, **NOISES[noise])
    spatial_check(h, vbabi.SpatialHolder(coords), y, "spatial %s %s" % (typ, noise))


@gpu
@pytest.mark.parametrize("noise", ["ar", "ar2none", "ar2same", "ar2dual"])
def test_alpha_distributions_from_file_under_spatial_vb(noise):
    """noise-initial-prior / -posterior for the AR(1) coefficients (Ar1cParams::InputFromMVN, noisemodel_ar.cc:302-316)
    inside the spatial loop: an informative prior with covariance between the alphas and a given initial posterior,
    with F (the prior's log-determinant, quadratic form and trace are terms of it)"""
    from test_ar_general import alpha_distributions
    n_alphas = 2 + {"none": 0, "same": 1, "dual": 2}[NOISES[noise].get("ar_cross_terms", "none")]
    prior, post = alpha_distributions(n_alphas, seed=40 + n_alphas)
    _, coords = masked_volume((9, 8, 6), seed=21)
    V, T = coords.shape[1], 40
    _, y = smooth_line_data(coords, T, seed=22, rho=0.3)
    for typ, kw in (("M", dict(ar_alpha_prior=prior, ar_alpha_post=post)), ("P", dict(ar_alpha_prior=prior)), ("m", dict(ar_alpha_post=post))):
        h = vbabi.build_config(vbabi.MODEL_POLY, V, T, degree=1, max_iterations=6, need_f=True, param_overrides={"c0": dict(type=typ)},
                               **NOISES[noise], **kw)
        plain = vbabi.build_config(vbabi.MODEL_POLY, V, T, degree=1, max_iterations=6, need_f=True, param_overrides={"c0": dict(type=typ)},
                                   **NOISES[noise])
        _, got = spatial_check(h, vbabi.SpatialHolder(coords), y, "alpha from file %s %s" % (typ, noise), check_f=True)
        assert np.abs(got["free_energy"] - hiplib.run_spatial_host(plain, vbabi.SpatialHolder(coords), y)["free_energy"]).max() > 1e-3


@gpu
@pytest.mark.parametrize("noise", ["ar", "pattern12", "ar2dual", "ar2none"])
def test_nonlinear_model_with_free_energy_ard_and_two_spatial_parameters(noise):
    _, coords = masked_volume((10, 8, 6), seed=5)
    V = coords.shape[1]
    _, y = smooth_exp_data(coords, 50, 0.04, seed=6, rho=0.3 if is_ar(noise) else 0.0)
    h = vbabi.build_config(vbabi.MODEL_EXP, V, 50, num_exps=1, dt=0.04, max_iterations=7, need_f=True,
                           param_overrides={"amp1": dict(type="M"), "r1": dict(type="A")}, **NOISES[noise])
    # (the two CPU builds are 1.0e-6 apart on this fit already: the bound follows the measured floor)
    spatial_check(h, vbabi.SpatialHolder(coords, update_first_iter=True), y, "exp M+ARD F " + noise, check_f=True, allow_floor=True)
    h = vbabi.build_config(vbabi.MODEL_EXP, V, 50, num_exps=1, dt=0.04, max_iterations=7, need_f=True,
                           param_overrides={"amp1": dict(type="P"), "r1": dict(type="m")}, **NOISES[noise])
    spatial_check(h, vbabi.SpatialHolder(coords, spatial_speed=1.5, q1=5.0, q2=2.0, spatial_dims=2), y, "exp P+m F " + noise, check_f=True, allow_floor=True)


@gpu
@pytest.mark.parametrize("noise", ["ar", "pattern12", "ar2dual", "ar2same"])
def test_linear_design_and_float32_series(noise):
    """C4's model (design matrix, four regressors) under a spatial prior, the series as the C ABI hands it over"""
    _, coords = masked_volume((9, 8, 6), seed=7)
    V, T = coords.shape[1], 60
    t = np.arange(T)
    X = np.stack([np.ones(T), t / T, np.sin(2 * np.pi * t / 25), np.cos(2 * np.pi * t / 25)], axis=1)
    rng = np.random.default_rng(8)
    beta = rng.normal(0, 3, (4, V)) + np.stack([np.sin(coords[0] / 2.0) * 4, 0 * coords[0], 0 * coords[0], 0 * coords[0]])
    y = (X @ beta + ar_noise(T, V, 0.3 if is_ar(noise) else 0.0, 1.0, 9)).astype(np.float32)
    h = vbabi.build_config(vbabi.MODEL_LINEAR, V, T, design=X, max_iterations=6, param_overrides={"beta_1": dict(type="M")} if False else None,
                           **NOISES[noise])
    names = [p["name"] for p in h.params]
    h = vbabi.build_config(vbabi.MODEL_LINEAR, V, T, design=X, max_iterations=6, param_overrides={names[0]: dict(type="M")}, **NOISES[noise])
    spatial_check(h, vbabi.SpatialHolder(coords), y, "linear M " + noise)


@gpu
@pytest.mark.parametrize("noise", ["ar", "pattern12", "ar2dual"])
@pytest.mark.parametrize("typ", ["M", "P"])
def test_failing_voxels_under_other_noise_models(typ, noise):
    """Vb::IgnoreVoxel: the voxels with a non-finite sample fail (with F: at the first CalculateF of the sweep) and
    leave their neighbours' lists; the split sweep hands such a run to the per-level launches"""
    _, coords = masked_volume((9, 8, 6), seed=11, keep=0.9)
    V = coords.shape[1]
    _, y = smooth_exp_data(coords, 50, 0.04, seed=12)
    bad = [V // 2, V // 2 + 1, V - 1]
    for v in bad:
        y[7, v] = np.nan
    for need_f in (True, False):
        h = vbabi.build_config(vbabi.MODEL_EXP, V, 50, num_exps=1, dt=0.04, max_iterations=6, need_f=need_f,
                               param_overrides={"amp1": dict(type=typ)}, **NOISES[noise])
        # (two CPU builds are up to 6e-7 apart here: the bound follows 3 x that floor)
        _, got = spatial_check(h, vbabi.SpatialHolder(coords), y, "IgnoreVoxel %s %s F=%d" % (typ, noise, need_f), check_f=need_f,
                               allow_floor=True)
        if need_f:
            assert sorted(np.flatnonzero(got["status"]).tolist()) == sorted(bad)


@gpu
@pytest.mark.parametrize("noise", ["ar", "pattern12", "ar2dual", "ar2same"])
def test_continue_from_mvn_under_other_noise_models(noise):
    """3 + 3 iterations through continue-from-mvn: the noise posterior (alpha and phi; every precision) is taken from
    the MVN (Ar1cParams / WhiteParams::InputFromMVN)"""
    _, coords = masked_volume((8, 7, 5), seed=13)
    V, T = coords.shape[1], 30
    _, y = smooth_line_data(coords, T, seed=14, rho=0.3 if is_ar(noise) else 0.0)
    kw = dict(degree=1, param_overrides={"c0": dict(type="M")}, **NOISES[noise])
    first = hiplib.run_spatial_host(vbabi.build_config(vbabi.MODEL_POLY, V, T, max_iterations=3, **kw), vbabi.SpatialHolder(coords), y)
    h = vbabi.build_config(vbabi.MODEL_POLY, V, T, max_iterations=3, init_mvn=first["mvn"], **kw)
    spatial_check(h, vbabi.SpatialHolder(coords), y, "continue-from-mvn " + noise)


@gpu
@pytest.mark.parametrize("noise", ["white", "ar", "pattern12", "ar2dual"])
def test_locked_linearisation_centres(noise):
    # (a full block: with fixed centres nothing ever looks at the means again - ReCentre is what throws on non-finite
    # values - so the NaN prior mean of a voxel without neighbours, priors.cc:448-451, would spread through the
    # whole volume, in the reference as here)
    _, coords = masked_volume((8, 7, 5), seed=15, keep=1.0)
    V = coords.shape[1]
    _, y = smooth_exp_data(coords, 40, 0.05, seed=16, rho=0.3 if is_ar(noise) else 0.0)
    centres = np.stack([np.full(V, np.log(0.8)), np.full(V, np.log(1.3))])
    for need_f in (False, True):
        h = vbabi.build_config(vbabi.MODEL_EXP, V, 40, num_exps=1, dt=0.05, max_iterations=5, need_f=need_f,
                               param_overrides={"amp1": dict(type="M")}, **NOISES.get(noise, {}))
        spatial_check(h, vbabi.SpatialHolder(coords, locked_centres=centres), y, "locked centres %s F=%d" % (noise, need_f), check_f=need_f)


@gpu
def test_nine_noise_precisions_are_refused_everywhere_through_the_c_abi():
    """round 4: two-echo AR(1) and 5 - 8 noise precisions run under spatial VB too; what is left is what the engine does
    not take at all (more than FVB_MAX_PHIS = 8 precisions), with a message, voxelwise and spatial alike"""
    from fabber_core_amd import fabber
    rng = np.random.default_rng(0)
    data = rng.normal(2, 0.1, (4, 4, 3, 27)).astype(np.float32)
    for method in ("vb", "spatialvb"):
        with pytest.raises(Exception, match="(?i)precision|pattern|phis"):
            fabber.run(data, {"model": "poly", "degree": 1, "method": method, "noise": "white", "noise-pattern": "123456789", "max-iterations": 2})


# ---- the reference's own test through the C ABI -----------------------------------------------------
@gpu
@pytest.mark.parametrize("method", ["vb", "spatialvb"])
@pytest.mark.parametrize("noise", ["ar", "white"])
def test_reference_vbtest_noise_fits(method, noise):
    """test/test_vb.cc:617-694 (ArNoise) and :697-774 (WhiteNoise), instantiated for "vb" and "spatialvb" (:847):
    a cubic VAL + 1.5 VAL n^2 - 2 VAL n^3 with uniform noise, poly degree 3, 50 iterations; every coefficient
    within 0.2."""
    from fabber_core_amd import fabber
    NT, VS, VAL = 10, 5, 2.0
    rng = np.random.default_rng(1)
    n = np.arange(1, NT + 1.0)
    clean = VAL + 1.5 * VAL * n ** 2 - 2 * VAL * n ** 3
    amp = VAL / 200 if noise == "ar" else VAL / 100
    data = (clean[None, None, None, :] + (rng.random((VS, VS, VS, NT)) - 0.5) * amp).astype(np.float32)
    out = fabber.run(data, {"noise": noise, "model": "poly", "max-iterations": 50, "degree": 3, "method": method, "save-mean": True})
    for name, want in (("mean_c0", VAL), ("mean_c1", 0.0), ("mean_c2", 1.5 * VAL), ("mean_c3", -2 * VAL)):
        assert out[name].shape == (VS, VS, VS)
        assert np.all(np.abs(out[name] - want) < 0.2), (name, float(np.abs(out[name] - want).max()))


@gpu
@pytest.mark.parametrize("echoes,cross", [(1, "none"), (2, "dual"), (2, "none")])
def test_spatialvb_with_ar_noise_and_a_spatial_prior_through_the_c_abi(echoes, cross):
    """noise=ar + PSP_byname1_type=M through fabber_dorun: the result images carry the AR block (the AR coefficients,
    then one precision per echo) and match the engine's own entry point; one echo, and two with and without the cross
    terms (num-echoes=2, ar1-cross-terms: noisemodel_ar.cc:322-332)"""
    from fabber_core_amd import fabber
    shape = (7, 6, 5)
    coords = vbabi.grid_coords(shape)
    V, T = coords.shape[1], 30
    _, y = smooth_line_data(coords, T, seed=21, rho=0.3)
    data = np.ascontiguousarray(y.T.reshape(shape[2], shape[1], shape[0], T).transpose(2, 1, 0, 3)).astype(np.float32)
    opts = {"noise": "ar", "model": "poly", "degree": 1, "method": "spatialvb", "max-iterations": 5, "save-mvn": True, "save-mean": True,
            "PSP_byname1": "c0", "PSP_byname1_type": "M"}
    if echoes == 2:
        opts.update({"num-echoes": 2, "ar1-cross-terms": cross})
    out = fabber.run(data, opts)
    n = 2 + (2 + {"none": 0, "same": 1, "dual": 2}[cross] + echoes)
    assert out["finalMVN"].shape[3] == vbabi.mvn_rows(n)
    h = vbabi.build_config(vbabi.MODEL_POLY, V, T, degree=1, max_iterations=5, noise=AR, num_echoes=echoes, ar_cross_terms=cross,
                           param_overrides={"c0": dict(type="M")})
    y32 = data.transpose(2, 1, 0, 3).reshape(V, T).T.copy()
    ref = hiplib.run_spatial_host(h, vbabi.SpatialHolder(coords), y32)
    got = out["finalMVN"].transpose(2, 1, 0, 3).reshape(V, -1).T
    assert np.allclose(got, ref["mvn"], rtol=1e-5, atol=1e-7)  # (the images are float32)


@gpu
def test_locked_linear_from_mvn_under_spatialvb_through_the_c_abi():
    from fabber_core_amd import fabber
    shape = (6, 5, 4)
    coords = vbabi.grid_coords(shape)
    V, T = coords.shape[1], 40
    _, y = smooth_exp_data(coords, T, 0.05, seed=22)
    data = np.ascontiguousarray(y.T.reshape(shape[2], shape[1], shape[0], T).transpose(2, 1, 0, 3)).astype(np.float32)
    n = 3
    mvn = np.zeros(shape + (vbabi.mvn_rows(n),), dtype=np.float32)
    for i in range(n):
        mvn[..., i * (i + 1) // 2 + i] = 1.0
    nCov = n * (n + 1) // 2
    mvn[..., nCov] = np.log(0.8)
    mvn[..., nCov + 1] = np.log(1.3)
    mvn[..., nCov + 2] = 1.0
    mvn[..., nCov + n] = 1.0
    opts = {"noise": "white", "model": "exp", "dt": 0.05, "method": "spatialvb", "max-iterations": 5, "save-mvn": True,
            "PSP_byname1": "amp1", "PSP_byname1_type": "M", "locked-linear-from-mvn": "lockmvn"}
    out = fabber.run(data, opts, extra_data={"lockmvn": mvn})
    h = vbabi.build_config(vbabi.MODEL_EXP, V, T, num_exps=1, dt=0.05, max_iterations=5, param_overrides={"amp1": dict(type="M")})
    centres = np.stack([np.full(V, np.float32(np.log(0.8))), np.full(V, np.float32(np.log(1.3)))]).astype(np.float64)
    y32 = data.transpose(2, 1, 0, 3).reshape(V, T).T.copy()
    ref = hiplib.run_spatial_host(h, vbabi.SpatialHolder(coords, locked_centres=centres), y32)
    got = out["finalMVN"].transpose(2, 1, 0, 3).reshape(V, -1).T
    assert np.allclose(got, ref["mvn"], rtol=1e-5, atol=1e-7)
