"""AR(1) noise with two interleaved echoes and the cross-term variants (num-echoes = 2,
ar1-cross-terms = none / same / dual; noisemodel_ar.cc:83-769).

The oracle keeps the reference's alpha matrices as lines (oracle/vb_oracle_arn.inc). It is checked
here against (1) its own one-echo stencil restatement (vb_oracle_ar.inc), which the reference's
stored outputs pin, and (2) an independent NumPy transcription that builds the DENSE matrices with
the reference's loops and runs the update equations on them. GPU: the wave-per-voxel AR kernel
against the oracle."""
import os

import numpy as np
import pytest

import cases
import oracle
from fabber_core_amd import vbabi

AR = vbabi.NOISE_AR1


# ---- dense NumPy transcription (linear model: J = design, g = J theta) -------------------------
def alpha_matrix(n, a12, a34, n_phis, n_times):
    """Ar1cMatrixCache::Update, noisemodel_ar.cc:112-181 (1-based rows and columns)."""
    row, col = {0: (1 + n_phis, 1 + n_phis), 10: (1, 1 + n_phis), 20: (1, 1), 1: (4, 3), 11: (4, 1), 2: (4, 4)}[a12 * 10 + a34]
    value = -1.0 if a12 + a34 == 1 else 1.0
    if n == 2:
        row = row - 1 + 2 * (row % 2)
        col = col - 1 + 2 * (col % 2)
    m = np.zeros((n_times * n_phis, n_times * n_phis))
    for _ in range(n_times - 1):
        m[row - 1, col - 1] = m[col - 1, row - 1] = value
        row += n_phis
        col += n_phis
    return m


def dense_ar_vb(X, y, n_phis, n_alphas, iterations, alpha_prior=None, alpha_post=None):
    """Vb::DoCalculationsVoxelwise with Ar1cNoiseModel for one voxel of a linear model with the
    default priors (precision 1e-12 on theta): UpdateTheta, UpdateAlpha, UpdatePhi per iteration."""
    T, P = X.shape
    nT = T // n_phis
    M = lambda n, a, b: alpha_matrix(n, a, b, n_phis, nT)
    L0, m0 = np.eye(P) * 1e-12, np.zeros(P)
    theta = np.zeros(P)
    A0, a0 = np.eye(n_alphas) * 1e-4, np.zeros(n_alphas)
    a_mean, a_prec = np.zeros(n_alphas), A0.copy()
    if alpha_prior is not None:                         # noise-initial-prior: InputFromMVN, :302-316
        a0, A0 = np.array(alpha_prior[0], dtype=float), np.array(alpha_prior[1], dtype=float)
    if alpha_post is not None:                          # noise-initial-posterior
        a_mean, a_prec = np.array(alpha_post[0], dtype=float), np.linalg.inv(alpha_post[1])
    b0, c0 = 1e6, 1e-6
    b = np.full(n_phis, 1e-8)
    c = np.full(n_phis, c0 + (nT - 1) * 0.5)          # Precalculate, :765-768

    def marginals():
        cp = np.linalg.inv(a_prec) + np.outer(a_mean, a_mean)
        out = []
        for n in range(1, n_phis + 1):
            q = M(n, 0, 0) + M(n, 1, 0) * a_mean[n - 1] + M(n, 2, 0) * cp[n - 1, n - 1]
            if n_alphas >= 3:
                t = (2 + n if n_alphas == 4 else 3) - 1
                q = q + M(n, 0, 1) * a_mean[t] + M(n, 1, 1) * cp[n - 1, t] + M(n, 0, 2) * cp[t, t]
            out.append(q)
        return out

    Q = marginals()
    for _ in range(iterations):
        ml = theta.copy()                              # linearisation centre; g = X ml, J = X
        Xq = sum(b[i] * c[i] * Q[i] for i in range(n_phis))
        Lam = L0 + X.T @ Xq @ X                        # :589-591
        theta = np.linalg.solve(Lam, X.T @ Xq @ (y - X @ ml + X @ ml) + L0 @ m0)   # :604-609
        k = y - X @ ml + X @ (ml - theta)
        Li = np.linalg.inv(Lam)
        op = lambda mat: k @ mat @ k + np.trace(Li @ X.T @ mat @ X)   # OperatorKLJ, :433-445
        sc = b * c
        prec = A0.copy()
        for i in range(1, n_phis + 1):
            prec[i - 1, i - 1] += sc[i - 1] * op(M(i, 2, 0))
        Tn = n_alphas
        if Tn > 2:                                     # :476-485
            prec[2, 0] += 0.5 * sc[0] * op(M(1, 1, 1)); prec[0, 2] = prec[2, 0]
            prec[Tn - 1, 1] += 0.5 * sc[1] * op(M(2, 1, 1)); prec[1, Tn - 1] = prec[Tn - 1, 1]
            prec[2, 2] += sc[0] * op(M(1, 0, 2))
            prec[Tn - 1, Tn - 1] += sc[1] * op(M(2, 0, 2))
        tmp = A0 @ a0                                  # :501-502
        for i in range(1, n_phis + 1):
            tmp[i - 1] += -0.5 * sc[i - 1] * op(M(i, 1, 0))
        if Tn > 2:
            tmp[2] += -0.5 * sc[0] * op(M(1, 0, 1))
            tmp[Tn - 1] += -0.5 * sc[1] * op(M(2, 0, 1))
        a_prec = prec
        a_mean = np.linalg.solve(prec, tmp)
        Q = marginals()
        for i in range(n_phis):                        # UpdatePhi, :530-556
            t2 = k @ Q[i] @ k + np.trace(Li @ X.T @ Q[i] @ X)
            b[i] = 1 / (t2 * 0.5 + 1 / b0)
            c[i] = (nT - 1) * 0.5 + c0
    return theta, Li, a_mean, np.linalg.inv(a_prec), b * c


def two_echo_problem(n_voxels, n_times, seed, cross, **opts):
    """Two interleaved series with different noise levels and AR coefficients, linear model."""
    rng = np.random.default_rng(seed)
    t = np.arange(n_times, dtype=np.float64)
    X = np.stack([np.ones(n_times), (t % 2), np.sin(2 * np.pi * t / 17), np.cos(2 * np.pi * t / 17)], axis=1)
    theta = rng.normal(0, 5, size=(4, n_voxels))
    noise = np.zeros((n_times, n_voxels))
    for echo, (rho, sd) in enumerate(((0.4, 1.0), (-0.2, 0.5))):
        e = rng.normal(0, sd, size=(n_times // 2, n_voxels))
        for i in range(1, n_times // 2):
            e[i] += rho * e[i - 1]
        noise[echo::2] = e
    y = X @ theta + noise
    h = vbabi.build_config(vbabi.MODEL_LINEAR, n_voxels, n_times, design=X, noise=AR, num_echoes=2, ar_cross_terms=cross, **opts)
    return h, y, X


def unpack(res, h, v):
    n = h.cfg.n_params + h.n_noise_outputs
    cov = np.zeros((n, n))
    row = 0
    for r in range(n):
        for c in range(r + 1):
            cov[r, c] = cov[c, r] = res["mvn"][row, v]
            row += 1
    return res["mvn"][row:row + n, v], cov


def test_general_form_reproduces_the_one_echo_restatement(monkeypatch):
    h, y = cases.linear_problem(30, 50, seed=3, max_iterations=8, noise=AR, need_f=True)
    a = oracle.run(h, y)
    monkeypatch.setenv("ORACLE_AR_GENERAL", "1")
    b = oracle.run(h, y)
    assert np.all(a["status"] == 0) and np.all(b["status"] == 0)
    assert np.allclose(a["mvn"], b["mvn"], rtol=1e-7, atol=1e-9)
    assert np.allclose(a["free_energy"], b["free_energy"], rtol=1e-9)


@pytest.mark.parametrize("cross,n_alphas", [("none", 2), ("same", 3), ("dual", 4)])
def test_two_echoes_against_the_dense_transcription(cross, n_alphas):
    h, y, X = two_echo_problem(6, 40, seed=5, cross=cross, max_iterations=5)
    res = oracle.run(h, y.astype(np.float64))
    assert np.all(res["status"] == 0)
    assert res["mvn"].shape[0] == vbabi.mvn_rows(4 + n_alphas + 2)
    for v in range(6):
        theta, cov, a_mean, a_cov, phi = dense_ar_vb(X, y[:, v], 2, n_alphas, 5)
        means, full = unpack(res, h, v)
        assert np.allclose(means[:4], theta, rtol=1e-6, atol=1e-8), (cross, v)
        assert np.allclose(full[:4, :4], cov, rtol=1e-5, atol=1e-12)
        assert np.allclose(means[4:4 + n_alphas], a_mean, rtol=1e-6, atol=1e-9)
        assert np.allclose(full[4:4 + n_alphas, 4:4 + n_alphas], a_cov, rtol=1e-5, atol=1e-12)
        assert np.allclose(means[4 + n_alphas:], phi, rtol=1e-6)


def alpha_distributions(n_alphas, seed):
    """A prior and an initial posterior over the AR(1) coefficients with off-diagonal terms, as a noise-initial-prior /
    noise-initial-posterior file would give them: (mean, PRECISION matrix), (mean, COVARIANCE matrix)."""
    rng = np.random.default_rng(seed)
    a = rng.normal(size=(n_alphas, n_alphas))
    prec = 0.5 * (a @ a.T) + 3.0 * np.eye(n_alphas)     # an informative prior: its mean must show in the posterior
    b = rng.normal(size=(n_alphas, n_alphas))
    cov = 0.01 * (b @ b.T) + 0.05 * np.eye(n_alphas)
    return (rng.uniform(-0.3, 0.3, n_alphas), prec), (rng.uniform(-0.2, 0.2, n_alphas), cov)


@pytest.mark.parametrize("cross,n_alphas", [("none", 2), ("same", 3), ("dual", 4)])
def test_alpha_distributions_from_file_against_the_dense_transcription(cross, n_alphas):
    """Ar1cParams::InputFromMVN (noisemodel_ar.cc:302-316): the prior's mean and precision matrix enter UpdateAlpha
    (:468, :501-502) and the free energy, the initial posterior the first marginals"""
    prior, post = alpha_distributions(n_alphas, seed=n_alphas)
    h, y, X = two_echo_problem(5, 40, seed=8, cross=cross, max_iterations=4, ar_alpha_prior=prior, ar_alpha_post=post)
    res = oracle.run(h, y.astype(np.float64))
    plain = oracle.run(two_echo_problem(5, 40, seed=8, cross=cross, max_iterations=4)[0], y.astype(np.float64))
    assert np.all(res["status"] == 0)
    n = 4 + n_alphas + 2
    off = n * (n + 1) // 2
    assert np.abs(res["mvn"][off + 4:off + 4 + n_alphas] - plain["mvn"][off + 4:off + 4 + n_alphas]).max() > 1e-3
    for v in range(5):
        theta, cov, a_mean, a_cov, phi = dense_ar_vb(X, y[:, v], 2, n_alphas, 4, alpha_prior=prior, alpha_post=post)
        means, full = unpack(res, h, v)
        assert np.allclose(means[:4], theta, rtol=1e-6, atol=1e-8), (cross, v)
        assert np.allclose(means[4:4 + n_alphas], a_mean, rtol=1e-6, atol=1e-9)
        assert np.allclose(full[4:4 + n_alphas, 4:4 + n_alphas], a_cov, rtol=1e-5, atol=1e-12)
        assert np.allclose(means[4 + n_alphas:], phi, rtol=1e-6)


def test_one_echo_alpha_distributions_in_both_restatements(monkeypatch):
    prior, post = alpha_distributions(2, seed=11)
    # (three iterations: the two restatements agree to 1e-14 after one and drift apart by a factor of ~50 per iteration
    # on this design, with the hard-coded distributions as well)
    h, y = cases.linear_problem(20, 50, seed=3, max_iterations=3, noise=AR, need_f=True, ar_alpha_prior=prior, ar_alpha_post=post)
    a = oracle.run(h, y)
    monkeypatch.setenv("ORACLE_AR_GENERAL", "1")
    b = oracle.run(h, y)
    assert np.all(a["status"] == 0) and np.all(b["status"] == 0)
    assert np.allclose(a["mvn"], b["mvn"], rtol=1e-6, atol=1e-9)
    assert np.allclose(a["free_energy"], b["free_energy"], rtol=1e-9)
    hp, _ = cases.linear_problem(20, 50, seed=3, max_iterations=3, noise=AR, need_f=True)
    assert not np.allclose(oracle.run(hp, y)["free_energy"], a["free_energy"], rtol=1e-6)


def test_two_echoes_recover_the_two_noise_levels():
    h, y, _ = two_echo_problem(40, 200, seed=6, cross="none", max_iterations=10)
    res = oracle.run(h, y.astype(np.float32))
    n = 4 + 2 + 2
    off = n * (n + 1) // 2
    alpha, phi = res["mvn"][off + 4:off + 6], res["mvn"][off + 6:off + 8]
    assert abs(np.median(alpha[0]) - 0.4) < 0.1 and abs(np.median(alpha[1]) + 0.2) < 0.1
    assert abs(np.median(phi[0]) - 1.0) < 0.25 and abs(np.median(phi[1]) - 4.0) < 1.0


def test_invalid_echo_settings_are_refused():
    h, y = cases.linear_problem(4, 21, seed=1, noise=AR, num_echoes=2)     # odd length
    with pytest.raises(RuntimeError):
        oracle.run(h, y)
    h, y = cases.linear_problem(4, 20, seed=1, noise=AR, num_echoes=1, ar_cross_terms="dual")
    with pytest.raises(RuntimeError):
        oracle.run(h, y)


# ---- HIP: the wave-per-voxel AR kernel (csrc/vb_wave_ar_kernel.h) against the oracle ------------
from fabber_core_amd import hiplib  # noqa: E402
import hipengine  # noqa: E402


def assert_close_to_oracle(h, y, ref=None, rtol=1e-6):
    ref = ref or oracle.run(h, y)
    got = hipengine.run(h, y)
    assert np.array_equal(ref["status"], got["status"])
    ok = ref["status"] == 0
    n = h.cfg.n_params + h.n_noise_outputs
    off = n * (n + 1) // 2
    sd = np.sqrt(np.abs(np.stack([ref["mvn"][p * (p + 1) // 2 + p] for p in range(n)])))
    scale = np.maximum(np.abs(ref["mvn"][off:off + n]), sd)
    err = np.abs(got["mvn"][off:off + n] - ref["mvn"][off:off + n]) / scale
    assert err[:, ok].max() < rtol, err[:, ok].max()
    row = 0
    for r in range(n):
        for c in range(r + 1):
            d = np.abs(got["mvn"][row] - ref["mvn"][row])[ok] / np.maximum(sd[r] * sd[c], 1e-300)[ok]
            assert d.max() < 10 * rtol, (r, c, d.max())
            row += 1
    if h.cfg.need_f:
        assert np.allclose(got["free_energy"][ok], ref["free_energy"][ok], rtol=1e-7, atol=1e-6)
    return ref, got


@pytest.mark.gpu
@pytest.mark.parametrize("echoes,cross", [(1, "none"), (2, "none"), (2, "same"), (2, "dual")])
def test_alpha_distributions_from_file_on_the_gpu(echoes, cross):
    """noise-initial-prior / -posterior under AR(1) noise: the lane kernels from 4096 voxels (a full 2 x 2 prior in
    lane_ar1, the general precision matrix in lane_ar2), the wave-per-voxel kernel below"""
    n_alphas = 2 + {"none": 0, "same": 1, "dual": 2}[cross]
    prior, post = alpha_distributions(n_alphas, seed=20 + n_alphas)
    for V, only in ((300, "both"), (5000, "prior"), (64, "post")):
        opts = dict(max_iterations=5, need_f=True)
        if only in ("both", "prior"):
            opts["ar_alpha_prior"] = prior
        if only in ("both", "post"):
            opts["ar_alpha_post"] = post
        if echoes == 2:
            h, y, _ = two_echo_problem(V, 40, seed=9, cross=cross, **opts)
        else:
            h, y = cases.linear_problem(V, 50, seed=4, noise=AR, **opts)
        assert_close_to_oracle(h, y.astype(np.float32))
        # (one echo: lane_ar1 at every size; two echoes: lane_ar2 from 4096 voxels, the wave-per-voxel kernel below)
        assert ("lane_ar" if V >= 4096 or echoes == 1 else "wave") in hiplib.kernel_name(h), hiplib.kernel_name(h)
        if echoes == 1 and V == 300:
            hiplib.set_variant("wave")
            try:
                assert "wave" in hiplib.kernel_name(h)
                assert_close_to_oracle(h, y.astype(np.float32))
            finally:
                hiplib.set_variant("auto")


@pytest.mark.gpu
@pytest.mark.parametrize("cross", ["none", "same", "dual"])
@pytest.mark.parametrize("need_f", [False, True])
def test_two_echoes_on_the_gpu(cross, need_f):
    h, y, _ = two_echo_problem(200, 60, seed=7, cross=cross, max_iterations=8, need_f=need_f)
    assert_close_to_oracle(h, y.astype(np.float32))


# ---- the lane-per-voxel kernel for two echoes (csrc/vb_lane_arn_kernel.h: two streaming passes per iteration) ----
def run_on_the_lane_kernel(h, y):
    hiplib.set_variant("lane")
    try:
        assert "lane_ar2" in hiplib.kernel_name(h), hiplib.kernel_name(h)
        return hipengine.run(h, y)
    finally:
        hiplib.set_variant("auto")


def assert_lane_close_to_oracle(h, y, rtol=1e-6):
    ref = oracle.run(h, y)
    got = run_on_the_lane_kernel(h, y)
    assert np.array_equal(ref["status"], got["status"])
    assert np.array_equal(ref["iterations"], got["iterations"])
    ok = ref["status"] == 0
    n = h.cfg.n_params + h.n_noise_outputs
    off = n * (n + 1) // 2
    sd = np.sqrt(np.abs(np.stack([ref["mvn"][p * (p + 1) // 2 + p] for p in range(n)])))
    scale = np.maximum(np.abs(ref["mvn"][off:off + n]), sd)
    err = np.abs(got["mvn"][off:off + n] - ref["mvn"][off:off + n]) / scale
    assert err[:, ok].max() < rtol, err[:, ok].max()
    row = 0
    for r in range(n):
        for c in range(r + 1):
            d = np.abs(got["mvn"][row] - ref["mvn"][row])[ok] / np.maximum(sd[r] * sd[c], 1e-300)[ok]
            assert d.max() < 10 * rtol, (r, c, d.max())
            row += 1
    if h.cfg.need_f:
        assert np.allclose(got["free_energy"][ok], ref["free_energy"][ok], rtol=1e-7, atol=1e-6)
    return ref, got


@pytest.mark.gpu
@pytest.mark.parametrize("cross", ["none", "same", "dual"])
@pytest.mark.parametrize("need_f", [False, True])
def test_two_echoes_on_the_lane_kernel(cross, need_f):
    """strict per-voxel parity with vb_oracle_arn.inc: every cross-term variant (2 / 3 / 4 AR coefficients), with and
    without the free energy (evaluated four times per iteration, inference_vb.cc:468-495); also the automatic choice
    from 4096 voxels up"""
    h, y, _ = two_echo_problem(200, 60, seed=7, cross=cross, max_iterations=8, need_f=need_f)
    assert_lane_close_to_oracle(h, y.astype(np.float32))
    h, y, _ = two_echo_problem(4100, 40, seed=17, cross=cross, max_iterations=4, need_f=need_f)
    assert "lane_ar2" in hiplib.kernel_name(h)
    ref, got = oracle.run(h, y.astype(np.float32)), hipengine.run(h, y.astype(np.float32))
    n = 4 + 2 + {"none": 0, "same": 1, "dual": 2}[cross] + 2
    off = n * (n + 1) // 2
    assert np.array_equal(ref["status"], got["status"])
    assert np.allclose(got["mvn"][off:off + n], ref["mvn"][off:off + n], rtol=1e-5, atol=1e-7)


@pytest.mark.gpu
@pytest.mark.parametrize("conv", ["pointzeroone", "freduce", "trialmode", "lm"])
def test_two_echoes_lane_kernel_with_free_energy_detectors(conv):
    """detectors that save and revert (the alpha posterior and both precisions travel with the saved state)"""
    h, y, _ = two_echo_problem(150, 60, seed=8, cross="dual", max_iterations=20, convergence=conv, min_fchange=0.01)
    ref = oracle.run(h, y.astype(np.float32))
    got = run_on_the_lane_kernel(h, y.astype(np.float32))
    same = ref["iterations"] == got["iterations"]
    assert same.mean() > 0.97
    n = 4 + 4 + 2
    off = n * (n + 1) // 2
    assert np.allclose(got["mvn"][off:off + n, same], ref["mvn"][off:off + n, same], rtol=1e-5, atol=1e-7)
    assert np.allclose(got["free_energy"][same], ref["free_energy"][same], rtol=1e-6, atol=1e-5)


@pytest.mark.gpu
def test_two_echoes_lane_kernel_continue_from_mvn_models_and_float64_series():
    h, y, X = two_echo_problem(64, 60, seed=9, cross="same", max_iterations=3)
    first = run_on_the_lane_kernel(h, y.astype(np.float32))
    h2 = vbabi.build_config(vbabi.MODEL_LINEAR, 64, 60, design=X, noise=AR, num_echoes=2, ar_cross_terms="same",
                            max_iterations=3, init_mvn=first["mvn"])
    assert_lane_close_to_oracle(h2, y.astype(np.float32))
    he, ye = cases.exp_problem(128, 60, 1, 0.04, seed=3, noise_sd=0.05, noise=AR, num_echoes=2, max_iterations=8, need_f=True)
    assert_lane_close_to_oracle(he, ye, rtol=1e-5)
    hp, yp = cases.poly_problem(100, 24, 2, seed=5, noise=AR, num_echoes=2, ar_cross_terms="dual", max_iterations=6, need_f=True)
    assert_lane_close_to_oracle(hp, yp, rtol=1e-5)
    h64, y64, _ = two_echo_problem(90, 60, seed=10, cross="dual", max_iterations=5, need_f=True)
    assert_lane_close_to_oracle(h64, y64.astype(np.float64))
    # a failing voxel (non-finite sample) stops alone
    yb = y.astype(np.float32).copy()
    yb[5, 3] = np.nan
    ref, got = oracle.run(h, yb), run_on_the_lane_kernel(h, yb)
    assert np.array_equal(ref["status"] != 0, got["status"] != 0) and got["status"][3] != 0


@pytest.mark.gpu
def test_one_echo_wave_kernel_matches_oracle_and_lane_kernel():
    h, y = cases.linear_problem(300, 80, seed=3, max_iterations=8, noise=AR, need_f=True)
    hiplib.set_variant("wave")
    try:
        assert "wave" in hiplib.kernel_name(h)
        _, wave = assert_close_to_oracle(h, y)
    finally:
        hiplib.set_variant("auto")
    lane = hipengine.run(h, y)
    assert "lane_ar1" in hiplib.kernel_name(h)
    assert np.allclose(wave["mvn"], lane["mvn"], rtol=1e-6, atol=1e-9)
    assert np.allclose(wave["free_energy"], lane["free_energy"], rtol=1e-8)


@pytest.mark.gpu
def test_one_echo_with_a_parameter_count_the_lane_kernels_do_not_cover():
    rng = np.random.default_rng(11)
    T, P, V = 64, 9, 100
    t = np.arange(T)
    X = np.stack([np.cos(np.pi * (t + 0.5) * k / T) for k in range(P)], axis=1)
    y = X @ rng.normal(0, 3, (P, V)) + rng.normal(0, 0.5, (T, V))
    h = vbabi.build_config(vbabi.MODEL_LINEAR, V, T, design=X, noise=AR, max_iterations=6, need_f=True)
    assert "wave" in hiplib.kernel_name(h)
    assert_close_to_oracle(h, y.astype(np.float32))


@pytest.mark.gpu
@pytest.mark.parametrize("conv", ["pointzeroone", "freduce", "trialmode"])
def test_two_echoes_with_free_energy_detectors(conv):
    h, y, _ = two_echo_problem(150, 60, seed=8, cross="dual", max_iterations=20, convergence=conv, min_fchange=0.01)
    ref = oracle.run(h, y.astype(np.float32))
    got = hipengine.run(h, y.astype(np.float32))
    same = ref["iterations"] == got["iterations"]
    assert same.mean() > 0.97
    n = 4 + 4 + 2
    off = n * (n + 1) // 2
    assert np.allclose(got["mvn"][off:off + 4, same], ref["mvn"][off:off + 4, same], rtol=1e-5, atol=1e-7)
    assert np.allclose(got["free_energy"][same], ref["free_energy"][same], rtol=1e-6, atol=1e-5)


@pytest.mark.gpu
def test_two_echoes_continue_from_mvn_and_exponential_model():
    h, y, X = two_echo_problem(64, 60, seed=9, cross="same", max_iterations=3)
    first = hipengine.run(h, y.astype(np.float32))
    h2 = vbabi.build_config(vbabi.MODEL_LINEAR, 64, 60, design=X, noise=AR, num_echoes=2, ar_cross_terms="same",
                            max_iterations=3, init_mvn=first["mvn"])
    assert_close_to_oracle(h2, y.astype(np.float32))
    he, ye = cases.exp_problem(128, 60, 1, 0.04, seed=3, noise_sd=0.05, noise=AR, num_echoes=2, max_iterations=8, need_f=True)
    assert_close_to_oracle(he, ye, rtol=1e-5)


# ---- through the reference's API: noise=ar num-echoes=2 ar1-cross-terms=... (fabber_capi.h) -----
from fabber_core_amd import fabber  # noqa: E402


@pytest.mark.gpu
@pytest.mark.parametrize("cross,n_alphas", [("none", 2), ("dual", 4)])
def test_two_echoes_through_the_c_abi(cross, n_alphas, tmp_path):
    h, y, X = two_echo_problem(48, 60, seed=10, cross=cross, max_iterations=6)
    design = tmp_path / "design.mat"
    np.savetxt(design, X)
    vol = y.astype(np.float32).T.reshape(4, 4, 3, 60, order="F")
    out = fabber.run(vol, {"model": "linear", "basis": str(design), "noise": "ar", "num-echoes": 2, "ar1-cross-terms": cross,
                           "method": "vb", "max-iterations": 6, "save-mean": True, "save-mvn": True, "save-noise-mean": True,
                           "save-noise-std": True})
    eng = hipengine.run(h, y.astype(np.float32))
    n = 4 + n_alphas + 2
    assert out["finalMVN"].shape[3] == vbabi.mvn_rows(n)
    assert np.allclose(out["finalMVN"].reshape(48, -1, order="F").T, eng["mvn"], rtol=2e-6, atol=1e-10)
    # noise_means holds the first NumParams() = num-echoes noise entries, i.e. alpha_1, alpha_2
    # (noisemodel_ar.cc:362-365, inference_vb.cc:981-989)
    off = n * (n + 1) // 2
    assert out["noise_means"].shape == (4, 4, 3, 2)
    assert np.allclose(out["noise_means"].reshape(48, 2, order="F").T, eng["mvn"][off + 4:off + 6], rtol=2e-6, atol=1e-9)


@pytest.mark.gpu
def test_two_echoes_through_the_c_abi_at_a_volume_that_takes_the_lane_kernel(tmp_path):
    """from 4096 voxels up fabber_dorun runs the same options on lane_ar2 (vb_lane_arn_kernel.h): the logfile names the
    kernel, the images are the engine call's, and a few voxels are checked against the oracle"""
    V = 17 * 16 * 16
    h, y, X = two_echo_problem(V, 60, seed=12, cross="dual", max_iterations=5, need_f=True)
    design = tmp_path / "design.mat"
    np.savetxt(design, X)
    vol = y.astype(np.float32).T.reshape(17, 16, 16, 60, order="F")
    out = fabber.run(vol, {"model": "linear", "basis": str(design), "noise": "ar", "num-echoes": 2, "ar1-cross-terms": "dual",
                           "method": "vb", "max-iterations": 5, "save-mean": True, "save-mvn": True, "save-free-energy": True})
    assert "lane_ar2<linear,4,4,F>" in out["log"]
    eng = hipengine.run(h, y.astype(np.float32))
    assert np.allclose(out["finalMVN"].reshape(V, -1, order="F").T, eng["mvn"], rtol=2e-6, atol=1e-10)
    assert np.allclose(out["freeEnergy"].reshape(V, order="F"), eng["free_energy"], rtol=1e-6)
    hs, ys, _ = two_echo_problem(V, 60, seed=12, cross="dual", max_iterations=5, need_f=True)
    pick = np.arange(0, V, 97)
    hs.cfg.n_voxels = len(pick)
    ref = oracle.run(hs, np.ascontiguousarray(ys.astype(np.float32)[:, pick]))
    n = 4 + 4 + 2
    off = n * (n + 1) // 2
    assert np.allclose(eng["mvn"][off:off + n, pick], ref["mvn"][off:off + n], rtol=1e-5, atol=1e-7)
    assert np.allclose(eng["free_energy"][pick], ref["free_energy"], rtol=1e-7, atol=1e-6)


def test_echo_options_are_validated():
    with fabber.Fabber() as f:
        f.set_extent((2, 2, 1))
        f.set_options({"model": "poly", "degree": 1, "noise": "ar", "method": "vb", "num-echoes": 1, "ar1-cross-terms": "dual"})
        f.set_data("data", np.ones((2, 2, 1, 10), dtype=np.float32))
        with pytest.raises(fabber.FabberError, match="ar1-cross-terms=none with num-echoes=1"):
            f.run()
    with fabber.Fabber() as f:
        f.set_extent((2, 2, 1))
        f.set_options({"model": "poly", "degree": 1, "noise": "ar", "method": "vb", "num-echoes": 3})
        f.set_data("data", np.ones((2, 2, 1, 10), dtype=np.float32))
        with pytest.raises(fabber.FabberError, match="Must be 1 or 2"):
            f.run()


@pytest.mark.gpu
def test_long_series_many_parameters_use_the_large_lds_path():
    """T = 400, P = 8, two echoes: ~125 KB of LDS per voxel (above the 64 KB default limit)."""
    rng = np.random.default_rng(12)
    T, P, V = 400, 8, 24
    t = np.arange(T)
    X = np.stack([np.cos(np.pi * (t + 0.5) * k / T) for k in range(P)], axis=1)
    y = X @ rng.normal(0, 3, (P, V)) + rng.normal(0, 0.5, (T, V))
    h = vbabi.build_config(vbabi.MODEL_LINEAR, V, T, design=X, noise=AR, num_echoes=2, ar_cross_terms="dual", max_iterations=4)
    assert_close_to_oracle(h, y.astype(np.float32))
