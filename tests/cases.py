"""Known-answer cases of the reference's own tests, as engine-independent functions.

Every case takes ``engine`` - a callable ``engine(holder, data_float32) -> results dict`` with
keys mvn / free_energy / status / iterations (tests/oracle.py:run has that shape, and so does
tests/hipengine.py:run) - so that the same assertions are applied to the CPU oracle and to the
HIP path. Sources: /test/test_inference.cc and /test/test_vb.cc of the reference.
"""
import numpy as np

from fabber_core_amd import vbabi


def float_eq(a, b):
    """gtest ASSERT_FLOAT_EQ: within 4 ULPs as float32."""
    a32, b32 = np.float32(a), np.float32(b)
    if a32 == b32:
        return True
    ia = np.frombuffer(np.float32(a32).tobytes(), dtype=np.int32)[0]
    ib = np.frombuffer(np.float32(b32).tobytes(), dtype=np.int32)[0]
    return abs(int(ia) - int(ib)) <= 4


def means_of(res, holder):
    """Posterior means in model space per parameter -> [P][V] (inference.cc:139-147)."""
    cfg = holder.cfg
    n = cfg.n_params + holder.n_noise_outputs
    off = n * (n + 1) // 2
    m = res["mvn"][off:off + cfg.n_params].copy()
    for p in range(cfg.n_params):
        tr = cfg.transform[p]
        if tr != vbabi.TRANSFORM_IDENTITY:
            m[p] = [vbabi.to_model(tr, x) for x in m[p]]
    return m


VAL = np.float32(7.32)


def constant_data(engine, n_voxels=125, n_times=10):
    """test_inference.cc:108-186 OneParam*MultiTimeslice: constant series -> mean_c0 == VAL."""
    h = vbabi.build_config(vbabi.MODEL_POLY, n_voxels, n_times, degree=0)
    data = np.full((n_times, n_voxels), VAL, dtype=np.float32)
    res = engine(h, data)
    m = means_of(res, h)
    assert m.shape == (1, n_voxels)
    assert np.all(res["status"] == 0)
    for v in range(n_voxels):
        assert float_eq(m[0, v], VAL), (m[0, v], VAL)
    return res


def alternating_data(engine, n_voxels=125, n_times=10):
    """test_inference.cc:190-238: VAL / 3 VAL alternating -> mean_c0 == 2 VAL."""
    h = vbabi.build_config(vbabi.MODEL_POLY, n_voxels, n_times, degree=0)
    data = np.empty((n_times, n_voxels), dtype=np.float32)
    data[0::2] = VAL
    data[1::2] = VAL * np.float32(3)
    res = engine(h, data)
    m = means_of(res, h)
    for v in range(n_voxels):
        assert float_eq(m[0, v], VAL * np.float32(2)), (m[0, v], VAL * 2)
    return res


def cubic_data(n_voxels, n_times, val=2.0):
    n = np.arange(1, n_times + 1, dtype=np.float64)
    y = val + (1.5 * val) * n * n - 2 * val * n * n * n
    return np.repeat(y[:, None], n_voxels, axis=1).astype(np.float32)


def polynomial_fit(engine, n_voxels=125, n_times=10):
    """test_inference.cc:353-429: cubic, degree 3, 50 iterations -> coefficients within 1e-3."""
    val = 2.0
    h = vbabi.build_config(vbabi.MODEL_POLY, n_voxels, n_times, degree=3, max_iterations=50)
    res = engine(h, cubic_data(n_voxels, n_times, val))
    m = means_of(res, h)
    assert np.all(np.abs(m[0] - val) < 1e-3)
    assert np.all(np.abs(m[1]) < 1e-3)
    assert np.all(np.abs(m[2] - 1.5 * val) < 1e-3)
    assert np.all(np.abs(m[3] + 2 * val) < 1e-3)
    return res


def masked_timepoints(engine, n_voxels=125, n_times=10):
    """test_inference.cc:485-561."""
    val = np.float32(2)
    data = np.full((n_times, n_voxels), val, dtype=np.float32)
    h = vbabi.build_config(vbabi.MODEL_POLY, n_voxels, n_times, degree=1, max_iterations=10)
    m = means_of(engine(h, data), h)
    assert np.all(np.abs(m[0] - val) < 1e-3)
    data[2] = val * 2
    data[6] = val * 2
    m = means_of(engine(h, data), h)
    assert np.all(m[0] > val)
    h = vbabi.build_config(vbabi.MODEL_POLY, n_voxels, n_times, degree=1, max_iterations=10, masked_timepoints=(3, 7))
    res = engine(h, data)
    m = means_of(res, h)
    assert np.all(np.abs(m[0] - val) < 1e-3)
    return res


def quadratic_data(n_voxels, n_times):
    n = np.arange(1, n_times + 1, dtype=np.float64)
    y = float(VAL) + (1.5 * float(VAL)) * n * n
    # float64: the reference's test fills a NEWMAT::Matrix (double) directly (test_vb.cc:326-333)
    return np.repeat(y[:, None], n_voxels, axis=1)


def restart_chain(engine, n_voxels=125, n_times=10, repeats=50, degree=5):
    """test_vb.cc:305-409: 51 chained 1-iteration runs through continue-from-mvn converge."""
    data = quadratic_data(n_voxels, n_times)
    h = vbabi.build_config(vbabi.MODEL_POLY, n_voxels, n_times, degree=degree, max_iterations=1)
    res = engine(h, data)
    m = means_of(res, h)
    assert m[0, 0] != VAL
    assert m[2, 0] != VAL * np.float32(1.5)
    for _ in range(repeats):
        h = vbabi.build_config(vbabi.MODEL_POLY, n_voxels, n_times, degree=degree, max_iterations=1, init_mvn=res["mvn"])
        res = engine(h, data)
    m = means_of(res, h)
    for v in range(n_voxels):
        assert float_eq(m[0, v], VAL), (m[0, v], VAL)
        assert abs(m[1, v]) < 1e-5
        assert float_eq(m[2, v], VAL * np.float32(1.5)), (m[2, v], VAL * 1.5)
    return res


def image_prior_data(n_voxels, n_times):
    data = np.empty((n_times, n_voxels), dtype=np.float32)
    data[0::2] = VAL
    data[1::2] = VAL * np.float32(3)
    return data


def image_prior_high_precision(engine, n_voxels=125, n_times=10):
    """test_vb.cc:118-175: image prior with huge precision pins the mean near the prior."""
    iprior = np.full(n_voxels, float(VAL) * 1.5)
    h = vbabi.build_config(vbabi.MODEL_POLY, n_voxels, n_times, degree=0,
                           param_overrides={"c0": dict(type="I", prec=1e12)}, image_priors={"c0": iprior})
    res = engine(h, image_prior_data(n_voxels, n_times))
    m = means_of(res, h)
    assert np.all(np.abs(m[0] - iprior) < 0.1 * float(VAL))
    return res


def image_prior_low_precision(engine, n_voxels=125, n_times=10):
    """test_vb.cc:177-232: image prior with tiny precision leaves the data mean."""
    iprior = np.full(n_voxels, float(VAL) * 1.5)
    h = vbabi.build_config(vbabi.MODEL_POLY, n_voxels, n_times, degree=0,
                           param_overrides={"c0": dict(type="I", prec=1e-5)}, image_priors={"c0": iprior})
    res = engine(h, image_prior_data(n_voxels, n_times))
    m = means_of(res, h)
    assert np.all(np.abs(m[0] - 2 * float(VAL)) < 0.1 * float(VAL))
    return res


ALL_CASES = [constant_data, alternating_data, polynomial_fit, masked_timepoints, restart_chain,
             image_prior_high_precision, image_prior_low_precision]


# ---------------------------------------------------------------------------------------------
# Seeded synthetic problems shared by the oracle-vs-HIP parity tests and by bench.py
# ---------------------------------------------------------------------------------------------
def exp_problem(n_voxels, n_times, num_exps, dt, seed, noise_sd=0.1, **cfg_opts):
    """examples/test_single.py / test_biexp.py style data: patches of amp1 in {1, .5},
    r1 in {1, .8} (+ amp2 = .5, r2 = 6 for the bi-exponential), N(0, noise_sd^2) noise."""
    rng = np.random.default_rng(seed)
    t = np.arange(n_times, dtype=np.float64) * dt
    amp1 = np.where(rng.integers(0, 2, n_voxels) == 0, 1.0, 0.5)
    r1 = np.where(rng.integers(0, 2, n_voxels) == 0, 1.0, 0.8)
    y = amp1[None, :] * np.exp(-r1[None, :] * t[:, None])
    if num_exps == 2:
        y += 0.5 * np.exp(-6.0 * t[:, None])
    y += rng.normal(0.0, noise_sd, size=y.shape)
    h = vbabi.build_config(vbabi.MODEL_EXP, n_voxels, n_times, num_exps=num_exps, dt=dt, **cfg_opts)
    return h, y.astype(np.float32)


def poly_problem(n_voxels, n_times, degree, seed, noise_sd=0.05, **cfg_opts):
    rng = np.random.default_rng(seed)
    coef = rng.uniform(-5, 5, size=(degree + 1, n_voxels))
    t = np.arange(1, n_times + 1, dtype=np.float64)
    y = sum(coef[n][None, :] * (t[:, None] ** n) for n in range(degree + 1))
    y = y + rng.normal(0.0, noise_sd, size=y.shape)
    h = vbabi.build_config(vbabi.MODEL_POLY, n_voxels, n_times, degree=degree, **cfg_opts)
    return h, y.astype(np.float32)


def linear_problem(n_voxels, n_times, seed, noise_sd=1.0, **cfg_opts):
    """BASELINE config 4 style design: columns {1, t/T, sin 2 pi t/50, cos 2 pi t/50}."""
    rng = np.random.default_rng(seed)
    t = np.arange(n_times, dtype=np.float64)
    X = np.stack([np.ones(n_times), t / n_times, np.sin(2 * np.pi * t / 50), np.cos(2 * np.pi * t / 50)], axis=1)
    theta = rng.normal(0, 10, size=(4, n_voxels))
    y = X @ theta + rng.normal(0, noise_sd, size=(n_times, n_voxels))
    h = vbabi.build_config(vbabi.MODEL_LINEAR, n_voxels, n_times, design=X, **cfg_opts)
    return h, y.astype(np.float32)


def c5_problem(shape, seed=20260105, noise_sd=0.1, n_times=100, dt=0.02, max_iterations=10, sigma=4.0, **cfg_opts):
    """BASELINE configs[4] (SURVEY 8d "C5 inputs"): the bi-exponential model of config 3 on a full
    shape[0] x shape[1] x shape[2] grid with a 6-neighbour MRF prior (type M) on amp1; the ground-truth
    amp1 is a smooth field (white noise, Gaussian-filtered with sigma voxels, rescaled to [0.5, 1]).
    Returns (holder, coords [3][V], data float32 [T][V], amp1 [V])."""
    import scipy.ndimage
    nx, ny, nz = shape
    rng = np.random.default_rng(seed)
    field = scipy.ndimage.gaussian_filter(rng.standard_normal((nz, ny, nx)), sigma, mode="nearest")
    field = 0.5 + 0.5 * (field - field.min()) / max(field.max() - field.min(), 1e-30)
    coords = vbabi.grid_coords(shape)
    amp1 = field.reshape(-1).astype(np.float32)  # z slowest, x fastest: the voxel order of grid_coords
    V = amp1.size
    t = np.arange(n_times, dtype=np.float64) * dt
    e1 = np.exp(-1.0 * t).astype(np.float32)
    e2 = (0.5 * np.exp(-6.0 * t)).astype(np.float32)
    y = np.empty((n_times, V), dtype=np.float32)
    for i in range(n_times):  # row by row: 128^3 x 100 floats are 840 MB, no float64 temporaries of that size
        y[i] = amp1 * e1[i] + e2[i] + rng.standard_normal(V, dtype=np.float32) * np.float32(noise_sd)
    opts = dict(param_overrides={"amp1": dict(type="M")})
    opts.update(cfg_opts)
    h = vbabi.build_config(vbabi.MODEL_EXP, V, n_times, num_exps=2, dt=dt, max_iterations=max_iterations, **opts)
    return h, coords, y, amp1


def cubic_cases():
    """The cubic-polynomial problems whose fp64 builds differ by more than the 1e-6 base tolerance (columns 1 ... t^3
    ~ 1e4 over 20 - 24 timepoints: ARD, masked timepoints, the noise options): name -> (holder, data). Held against a
    binary128 ground truth (tests/golden/make_cubic_truth.py -> cubic_truth_binary128.npz) instead of a raised bound."""
    out = {}
    out["ARD last"] = poly_problem(900, 20, 3, seed=5, max_iterations=12, need_f=True, param_overrides={"c3": dict(type="A")})
    out["ARD middle"] = poly_problem(900, 20, 3, seed=6, max_iterations=4, need_f=True,
                                     param_overrides={"c1": dict(type="A"), "c2": dict(mean=1.0, prec=0.5)})
    out["masked"] = poly_problem(333, 24, 3, seed=21, max_iterations=15, masked_timepoints=(3, 7, 24), need_f=True)
    out["prior-noise-stddev"] = poly_problem(333, 24, 3, seed=21, max_iterations=15, prior_noise_stddev=0.5)
    out["locked-noise-stdev"] = poly_problem(333, 24, 3, seed=21, max_iterations=15, locked_noise_stdev=0.07)
    return out
