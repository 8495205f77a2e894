"""Forward models that exist only as host code (model libraries written for the reference): the
model is evaluated on the host, the rest of the VB loop on the GPU (vb_hostmodel.h,
fabber_vb_run_hostmodel_host). A small third-party-style library (tests/plugins/fwdmodel_bump.cc)
is compiled against the public headers and loaded with fabber_load_models."""
import os
import shutil
import subprocess

import numpy as np
import pytest

from fabber_core_amd import fabber, hiplib, vbabi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "fabber_core_amd", "csrc", "host")
LIBDIR = os.path.join(ROOT, "fabber_core_amd", "lib")
SRC = os.path.join(ROOT, "tests", "plugins", "fwdmodel_bump.cc")

pytestmark = [
    pytest.mark.skipif(shutil.which("g++") is None, reason="no g++"),
    pytest.mark.skipif(not os.path.exists(os.path.join(LIBDIR, "libfabbercore_amd.so")), reason="host library not built"),
]


@pytest.fixture(scope="module")
def plugin(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("plugin") / "libfabber_models_bump.so")
    cmd = ["g++", "-std=c++17", "-shared", "-fPIC", "-I", HOST, SRC, "-o", out, "-L", LIBDIR, "-lfabbercore_amd",
           "-Wl,-rpath," + LIBDIR, "-Wl,--no-undefined"]
    p = subprocess.run(cmd, capture_output=True, text=True)
    assert p.returncode == 0, p.stderr[-3000:]
    return out


def volume(shape, series):
    return np.broadcast_to(np.asarray(series, dtype=np.float32), tuple(shape) + (len(series),)).copy()


def test_plugin_loads_and_describes_itself(plugin):
    f = fabber.Fabber(model_libs=[plugin])
    assert {"bump", "mypoly"} <= set(f.get_models())
    f.set_options({"model": "bump", "noise": "white", "method": "vb"})
    assert f.get_model_params() == ["amp", "mu", "width"]
    y = np.asarray(f.model_evaluate([2.0, 5.0, 2.0], 9))
    assert np.allclose(y, 2.0 * np.exp(-(np.arange(1, 10) - 5.0) ** 2 / 8.0), rtol=1e-6)


@pytest.mark.gpu
def test_host_evaluated_copy_of_a_builtin_model_matches_the_device_run(plugin):
    """The library's 'mypoly' is the built-in polynomial model without a device body: same
    problem, model evaluated on the host vs in the kernel."""
    rng = np.random.default_rng(1)
    shape, T = (6, 5, 4), 12
    t = np.arange(1, T + 1)
    c = rng.uniform(-3, 3, shape + (3,))
    data = (c[..., 0:1] + c[..., 1:2] * t + c[..., 2:3] * t * t + rng.normal(0, 0.1, shape + (T,))).astype(np.float32)
    opts = {"degree": 2, "noise": "white", "method": "vb", "max-iterations": 8, "save-mean": True, "save-mvn": True,
            "save-free-energy": True, "save-model-fit": True, "save-residuals": True, "save-noise-mean": True}
    dev = fabber.run(data, dict(opts, model="poly"))
    host = fabber.run(data, dict(opts, model="mypoly"), model_libs=[plugin])
    assert "evaluated on the host" in host["log"]
    for k in ("mean_c0", "mean_c1", "mean_c2", "noise_means", "modelfit", "residuals"):
        assert np.allclose(host[k], dev[k], rtol=2e-5, atol=1e-5), k
    assert np.allclose(host["finalMVN"], dev["finalMVN"], rtol=1e-4, atol=1e-7)
    assert np.allclose(host["freeEnergy"], dev["freeEnergy"], rtol=1e-5)
    # the same for a built-in model forced onto the host route
    forced = fabber.run(data, dict(opts, model="poly", **{"host-model": True}))
    assert np.allclose(forced["finalMVN"], dev["finalMVN"], rtol=1e-4, atol=1e-7)


@pytest.mark.gpu
@pytest.mark.parametrize("conv", ["pointzeroone", "freduce", "trialmode", "lm"])
def test_every_convergence_detector_on_the_host_route(conv):
    """Save / revert / trial iterations survive the cut of the loop at its re-centres: same
    per-voxel results as the device route of the same (nonlinear) model."""
    rng = np.random.default_rng(2)
    shape, T = (8, 8, 4), 50
    t = np.arange(T) * 0.04
    amp = np.where(rng.random(shape) < 0.5, 1.0, 0.5)
    data = (amp[..., None] * np.exp(-t) + rng.normal(0, 0.1, shape + (T,))).astype(np.float32)
    opts = {"model": "exp", "num-exps": 1, "dt": 0.04, "noise": "white", "method": "vb", "convergence": conv, "max-iterations": 30,
            "min-fchange": 0.01, "save-mean": True, "save-mvn": True, "save-free-energy": True, "save-free-energy-history": True}
    dev = fabber.run(data, opts)
    host = fabber.run(data, dict(opts, **{"host-model": True}))
    assert host["freeEnergyHistory"].shape == dev["freeEnergyHistory"].shape
    close = np.isclose(host["freeEnergy"], dev["freeEnergy"], rtol=1e-4, atol=1e-3)
    assert close.mean() > 0.99      # a voxel whose |dF| sits on the threshold may stop one iteration apart
    sel = close
    assert np.allclose(host["mean_amp1"][sel], dev["mean_amp1"][sel], rtol=1e-3)
    assert np.allclose(host["mean_r1"][sel], dev["mean_r1"][sel], rtol=1e-3)


@pytest.mark.gpu
def test_third_party_model_fits_its_data(plugin):
    rng = np.random.default_rng(3)
    shape, T = (5, 5, 3), 24
    t = np.arange(1, T + 1)
    amp = rng.uniform(2, 4, shape)
    mu = rng.uniform(8, 16, shape)
    width = rng.uniform(2.5, 4, shape)
    clean = amp[..., None] * np.exp(-(t - mu[..., None]) ** 2 / (2 * width[..., None] ** 2))
    data = (clean + rng.normal(0, 0.02, shape + (T,))).astype(np.float32)
    out = fabber.run(data, {"model": "bump", "noise": "white", "method": "vb", "max-iterations": 20, "save-mean": True,
                            "save-model-fit": True, "save-mvn": True}, model_libs=[plugin])
    assert np.allclose(out["mean_amp"], amp, rtol=0.03)
    assert np.allclose(out["mean_mu"], mu, atol=0.1)
    assert np.allclose(np.abs(out["mean_width"]), width, rtol=0.05)
    assert np.sqrt(np.mean((out["modelfit"] - clean) ** 2)) < 0.02
    # bad voxels: a series of zeros (log amplitude prior: the model still evaluates) stays finite,
    # a NaN sample stops that voxel only when allow-bad-voxels is set
    data[1, 1, 1, 5] = np.nan
    with pytest.raises(fabber.FabberError):
        fabber.run(data, {"model": "bump", "noise": "white", "method": "vb", "save-mean": True}, model_libs=[plugin])
    out = fabber.run(data, {"model": "bump", "noise": "white", "method": "vb", "save-mean": True, "allow-bad-voxels": True},
                     model_libs=[plugin])
    assert np.isfinite(out["mean_amp"][0, 0, 0])


@pytest.mark.gpu
@pytest.mark.parametrize("prior", ["M", "P"])
def test_host_evaluated_model_under_spatial_vb(plugin, prior):
    """Any FwdModel runs under method=spatialvb in the reference (inference_vb.cc:578-767): here the library's
    'mypoly' (no device body) and a built-in model forced onto the host route, with a spatial prior on one
    parameter, ARD on another and F, against the device route of the same problem - the model's two re-centres
    per iteration run on the host, both sweeps on the device (fabber_vb_run_spatial_hostmodel_host)."""
    rng = np.random.default_rng(4)
    shape, T = (9, 8, 5), 16
    t = np.arange(1, T + 1)
    x = np.arange(shape[0])[:, None, None]
    c0 = 2.0 + np.sin(x / 2.0) + np.zeros(shape)
    data = (c0[..., None] + 0.3 * t + 0.01 * t * t + rng.normal(0, 0.2, shape + (T,))).astype(np.float32)
    opts = {"degree": 2, "noise": "white", "method": "spatialvb", "max-iterations": 6, "param-spatial-priors": prior + "AN",
            "save-mean": True, "save-mvn": True, "save-free-energy": True, "save-noise-mean": True, "save-model-fit": True}
    dev = fabber.run(data, dict(opts, model="poly"))
    host = fabber.run(data, dict(opts, model="mypoly"), model_libs=[plugin])
    assert "the model is evaluated on the host" in host["log"]
    forced = fabber.run(data, dict(opts, model="poly", **{"host-model": True}))
    for other in (host, forced):
        for k in ("mean_c0", "mean_c1", "mean_c2", "noise_means", "modelfit"):
            assert np.allclose(other[k], dev[k], rtol=2e-5, atol=1e-5), k
        assert np.allclose(other["finalMVN"], dev["finalMVN"], rtol=1e-4, atol=1e-7)
        assert np.allclose(other["freeEnergy"], dev["freeEnergy"], rtol=1e-5)
    # the spatial prior did something: the fit differs from the voxelwise one
    plain = fabber.run(data, {"model": "poly", "degree": 2, "noise": "white", "method": "vb", "max-iterations": 6, "save-mean": True})
    assert np.abs(plain["mean_c0"] - dev["mean_c0"]).max() > 1e-3


@pytest.mark.gpu
def test_nonlinear_host_model_under_spatial_vb_with_a_masked_volume(plugin):
    """the third-party model (nonlinear, transforms) on a masked volume with a voxel that fails: the host route
    skips it as the device route does; against the exponential model's device route for a built-in model"""
    rng = np.random.default_rng(5)
    shape, T = (7, 6, 4), 40
    t = np.arange(T) * 0.04
    amp = 0.5 + 0.5 * (np.arange(shape[0])[:, None, None] > 3) + np.zeros(shape)
    data = (amp[..., None] * np.exp(-t) + rng.normal(0, 0.05, shape + (T,))).astype(np.float32)
    mask = (rng.random(shape) < 0.85).astype(np.int32)
    opts = {"model": "exp", "num-exps": 1, "dt": 0.04, "noise": "white", "method": "spatialvb", "max-iterations": 5,
            "param-spatial-priors": "MN", "save-mean": True, "save-mvn": True}
    dev = fabber.run(data, opts, mask=mask)
    host = fabber.run(data, dict(opts, **{"host-model": True}), mask=mask)
    sel = mask > 0
    assert np.allclose(host["mean_amp1"][sel], dev["mean_amp1"][sel], rtol=1e-5)
    assert np.allclose(host["mean_r1"][sel], dev["mean_r1"][sel], rtol=1e-5)
    assert np.allclose(host["finalMVN"][sel], dev["finalMVN"][sel], rtol=1e-4, atol=1e-8)


@pytest.mark.gpu
@pytest.mark.parametrize("conv", ["maxits", "freduce"])
def test_host_evaluated_model_with_ar1_noise(plugin, conv):
    """noise=ar with a model that has no device body (Ar1cNoiseModel works with any FwdModel in the reference):
    vb_hostmodel_ar.h carries every voxel from re-centre to re-centre with the AR(1) steps of the wave kernel;
    against the device route of the same model, also with a detector that saves and reverts"""
    rng = np.random.default_rng(6)
    shape, T = (6, 5, 4), 40
    t = np.arange(1, T + 1)
    c = rng.uniform(-2, 2, shape + (3,))
    e = rng.normal(0, 0.3, shape + (T,))
    for k in range(1, T):  # AR(1) noise
        e[..., k] += 0.4 * e[..., k - 1]
    data = (c[..., 0:1] + c[..., 1:2] * t / 10 + c[..., 2:3] * (t / 10) ** 2 + e).astype(np.float32)
    opts = {"degree": 2, "noise": "ar", "method": "vb", "max-iterations": 8, "convergence": conv, "save-mean": True, "save-mvn": True,
            "save-free-energy": True, "save-noise-mean": True}
    dev = fabber.run(data, dict(opts, model="poly"))
    host = fabber.run(data, dict(opts, model="mypoly"), model_libs=[plugin])
    assert "evaluated on the host" in host["log"]
    for k in ("mean_c0", "mean_c1", "mean_c2", "noise_means"):
        assert np.allclose(host[k], dev[k], rtol=2e-5, atol=1e-5), k
    assert np.allclose(host["finalMVN"], dev["finalMVN"], rtol=1e-4, atol=1e-7)
    assert np.allclose(host["freeEnergy"], dev["freeEnergy"], rtol=1e-5)
    # two interleaved echoes with the dual cross terms (the general AR form), built-in model forced onto the host route
    opts2 = dict(opts, model="poly", **{"num-echoes": 2, "ar1-cross-terms": "dual"})
    dev2 = fabber.run(data, opts2)
    host2 = fabber.run(data, dict(opts2, **{"host-model": True}))
    assert np.allclose(host2["finalMVN"], dev2["finalMVN"], rtol=1e-4, atol=1e-7)


@pytest.mark.gpu
@pytest.mark.parametrize("noise", ["white", "ar"])
def test_host_route_in_batches_is_the_host_route_in_one(plugin, noise, monkeypatch):
    """the voxels still running are worked through in batches (two buffers each side, the host's next batch
    overlapping the device's current one): 7 voxels per batch here, with a detector that makes voxels finish at
    different steps - every output identical to the run with one batch"""
    rng = np.random.default_rng(7)
    shape, T = (5, 4, 3), 40
    t = np.arange(T) * 0.04
    amp = np.where(rng.random(shape) < 0.5, 1.0, 0.5)
    data = (amp[..., None] * np.exp(-t) + rng.normal(0, 0.1, shape + (T,))).astype(np.float32)
    opts = {"model": "exp", "num-exps": 1, "dt": 0.04, "noise": noise, "method": "vb", "convergence": "pointzeroone", "max-iterations": 20,
            "save-mean": True, "save-mvn": True, "save-free-energy": True, "host-model": True}
    one = fabber.run(data, opts)
    monkeypatch.setenv("FVB_HOSTMODEL_BATCH", "7")
    many = fabber.run(data, opts)
    for k in ("finalMVN", "freeEnergy", "mean_amp1", "mean_r1"):
        assert np.array_equal(one[k], many[k]), k


@pytest.mark.gpu
def test_spatial_vb_without_device_kernels_falls_back_to_the_host_model():
    """no spatial kernels are built for a polynomial of degree 6 (7 parameters): fabber_dorun lets the model's host
    code do the re-centres instead of refusing; against the same run with host-model set"""
    rng = np.random.default_rng(9)
    shape, T = (6, 5, 4), 16
    t = np.arange(1, T + 1) / 8.0
    data = (1.0 + 0.5 * t - 0.3 * t ** 2 + rng.normal(0, 0.05, shape + (T,))).astype(np.float32)
    opts = {"model": "poly", "degree": 6, "noise": "white", "method": "spatialvb", "max-iterations": 3, "param-spatial-priors": "M+",
            "save-mean": True, "save-mvn": True}
    auto = fabber.run(data, opts)
    assert "no device kernels for spatial VB" in auto["log"]
    forced = fabber.run(data, dict(opts, **{"host-model": True}))
    assert np.array_equal(auto["finalMVN"], forced["finalMVN"])


# ---- method=nlls with a model evaluated on the host (fabber_nlls_run_hostmodel_host) ----------------
@pytest.mark.gpu
def test_nlls_with_a_host_evaluated_copy_of_a_builtin_model(plugin):
    """NLLSInferenceTechnique works with any FwdModel (inference_nlls.cc:94-214): the library's 'mypoly' and a
    built-in nonlinear model forced onto the host route against the device route of the same model"""
    rng = np.random.default_rng(11)
    shape, T = (6, 5, 4), 12
    t = np.arange(1, T + 1)
    c = rng.uniform(-3, 3, shape + (3,))
    data = (c[..., 0:1] + c[..., 1:2] * t + c[..., 2:3] * t * t + rng.normal(0, 0.1, shape + (T,))).astype(np.float32)
    opts = {"degree": 2, "noise": "white", "method": "nlls", "save-mean": True, "save-mvn": True, "save-model-fit": True}
    dev = fabber.run(data, dict(opts, model="poly"))
    host = fabber.run(data, dict(opts, model="mypoly"), model_libs=[plugin])
    assert "evaluated on the host" in host["log"]
    for k in ("mean_c0", "mean_c1", "mean_c2", "modelfit"):
        assert np.allclose(host[k], dev[k], rtol=1e-5, atol=1e-5), k
    assert np.allclose(host["finalMVN"], dev["finalMVN"], rtol=1e-4, atol=1e-7)
    te = np.arange(50) * 0.04
    amp = np.where(rng.random(shape) < 0.5, 1.0, 0.5)
    data = (amp[..., None] * np.exp(-te) + rng.normal(0, 0.05, shape + (50,))).astype(np.float32)
    opts = {"model": "exp", "num-exps": 1, "dt": 0.04, "noise": "white", "method": "nlls", "save-mean": True, "save-mvn": True, "mt1": 7}
    dev = fabber.run(data, opts)
    host = fabber.run(data, dict(opts, **{"host-model": True, "host-model-threads": 3}))
    assert np.allclose(host["mean_amp1"], dev["mean_amp1"], rtol=1e-5)
    assert np.allclose(host["mean_r1"], dev["mean_r1"], rtol=1e-5)
    assert np.allclose(host["finalMVN"], dev["finalMVN"], rtol=1e-3, atol=1e-9)


@pytest.mark.gpu
def test_nlls_third_party_model_against_an_independent_least_squares_fit(plugin, tmp_path):
    """tests/plugins/fwdmodel_bump.cc under method=nlls: the minimum of the sum of squares is what SciPy's
    least-squares solver finds for the same model. The reference starts NLLS from HardcodedInitialDists' posterior
    (inference_nlls.cc:68-82), which models that describe themselves through GetParameterDefaults leave at zero
    (fwdmodel.h:305) - a zero width here, where the model and its Jacobian are identically zero: no step lowers the
    cost, the damping runs up to its limit, the start comes back with the floor precision 1e-6 (:164-170). With a
    starting estimate from a file (fwd-inital-posterior) the fit runs."""
    import scipy.optimize
    rng = np.random.default_rng(12)
    shape, T = (4, 4, 3), 24
    t = np.arange(1, T + 1)
    amp = rng.uniform(2, 4, shape)
    mu = rng.uniform(9, 12, shape)
    width = rng.uniform(2.5, 4, shape)
    clean = amp[..., None] * np.exp(-(t - mu[..., None]) ** 2 / (2 * width[..., None] ** 2))
    data = (clean + rng.normal(0, 0.02, shape + (T,))).astype(np.float32)
    opts = {"model": "bump", "noise": "white", "method": "nlls", "save-mean": True, "save-mvn": True, "save-model-fit": True}
    out = fabber.run(data, opts, model_libs=[plugin])
    assert "evaluated on the host" in out["log"]
    assert np.all(out["mean_mu"] == 0) and np.all(out["mean_width"] == 0) and np.all(out["mean_amp"] == 1)
    assert np.allclose(out["finalMVN"][..., 0], 1e6)  # J'J = 0: the diagonal is raised to 1e-6
    start = np.zeros((4, 4))
    start[:3, :3] = np.eye(3)
    start[:3, 3] = start[3, :3] = [np.log(2.0), 10.0, 3.0]  # Fabber space: amp is LOG-transformed
    start[3, 3] = 1.0
    np.savetxt(str(tmp_path / "start.mat"), start)
    out = fabber.run(data, dict(opts, **{"fwd-inital-posterior": str(tmp_path / "start.mat")}), model_libs=[plugin])
    for idx in np.ndindex(shape):
        y = data[idx].astype(np.float64)

        def resid(p):
            return p[0] * np.exp(-(t - p[1]) ** 2 / (2 * p[2] ** 2)) - y
        sol = scipy.optimize.least_squares(resid, [amp[idx], mu[idx], width[idx]], xtol=1e-14, ftol=1e-14, gtol=1e-14)
        got = np.array([out["mean_amp"][idx], out["mean_mu"][idx], abs(out["mean_width"][idx])])
        assert np.allclose(got, sol.x, rtol=2e-4), (idx, got, sol.x)
    assert np.sqrt(np.mean((out["modelfit"] - clean) ** 2)) < 0.02


@pytest.mark.gpu
def test_twenty_parameter_linear_model_through_the_c_api(tmp_path):
    """the engine took 16 parameters at most: a 20-column design under vb and nlls against the closed-form
    least-squares solution (flat priors: the VB means are the least-squares means, test_inference.cc:353-429 style)"""
    rng = np.random.default_rng(13)
    T, P, shape = 120, 20, (5, 4, 3)
    tt = np.arange(T)
    X = np.stack([np.cos(np.pi * (tt + 0.5) * k / T) for k in range(P)], axis=1)
    np.savetxt(str(tmp_path / "design.mat"), X)
    theta = rng.normal(0, 3, shape + (P,))
    data = (theta @ X.T + rng.normal(0, 0.3, shape + (T,))).astype(np.float32)
    want = np.linalg.lstsq(X, data.reshape(-1, T).T.astype(np.float64), rcond=None)[0].T.reshape(shape + (P,))
    for method in ("vb", "nlls"):
        out = fabber.run(data, {"model": "linear", "basis": str(tmp_path / "design.mat"), "noise": "white", "method": method,
                                "save-mean": True, "max-iterations": 10})
        for k in range(P):
            assert np.allclose(out["mean_Parameter_%d" % (k + 1)], want[..., k], rtol=1e-4, atol=1e-4), (method, k)


@pytest.mark.gpu
def test_forty_parameter_linear_model():
    """More than FVB_MAX_PARAMS = 32 parameters (the reference has no limit): the per-parameter entries travel as a table
    (fvb_config.params_ext) and the wave-per-voxel kernel takes the problem. A 40-column cosine design with an ARD prior
    and an image prior among the parameters: the engine against the oracle, strict per voxel, and through fabber_dorun
    against the closed-form least-squares solution, result images (mean, std, zstat, model fit) included."""
    import oracle
    import parity
    rng = np.random.default_rng(14)
    T, P, V = 100, 40, 300  # (the wave kernel keeps J, the moments and the P x P work areas of a voxel in LDS: 160 KB hold T = 100 at P = 40)
    tt = np.arange(T)
    X = np.stack([np.cos(np.pi * (tt + 0.5) * k / T) for k in range(P)], axis=1)
    theta = rng.normal(0, 3, (P, V))
    y = (X @ theta + rng.normal(0, 0.3, (T, V))).astype(np.float32)
    img = rng.normal(0.0, 1.0, V)
    h = vbabi.build_config(vbabi.MODEL_LINEAR, V, T, design=X, max_iterations=6, need_f=True,
                           param_overrides={"Parameter_38": dict(type="A"), "Parameter_5": dict(type="I", prec=0.5)},
                           image_priors={"Parameter_5": img})
    assert h.cfg.n_params == 40 and h.cfg.params_ext
    assert hiplib.kernel_name(h) == "wave"
    cpu, got = oracle.run(h, y), hiplib.run_host(h, y)
    parity.strict(h, cpu, got, what="forty parameters", check_f=True, cpu2=oracle.run_fma(h, y))
    # the C ABI: 5 x 4 x 3 voxels, flat priors -> the least-squares means
    shape = (5, 4, 3)
    import tempfile
    with tempfile.TemporaryDirectory() as tmp:
        np.savetxt(os.path.join(tmp, "design.mat"), X)
        th = rng.normal(0, 3, shape + (P,))
        data = (th @ X.T + rng.normal(0, 0.3, shape + (T,))).astype(np.float32)
        want = np.linalg.lstsq(X, data.reshape(-1, T).T.astype(np.float64), rcond=None)[0].T.reshape(shape + (P,))
        out = fabber.run(data, {"model": "linear", "basis": os.path.join(tmp, "design.mat"), "noise": "white", "method": "vb",
                                "save-mean": True, "save-std": True, "save-zstat": True, "save-model-fit": True, "max-iterations": 10})
        for k in range(P):
            assert np.allclose(out["mean_Parameter_%d" % (k + 1)], want[..., k], rtol=1e-4, atol=1e-4), k
            assert np.all(out["std_Parameter_%d" % (k + 1)] > 0)
        assert np.sqrt(np.mean((out["modelfit"] - data) ** 2)) < 0.4
        # method=nlls with the same 40 columns: the least-squares solution again (the wave-per-voxel minimiser reads the table)
        out = fabber.run(data, {"model": "linear", "basis": os.path.join(tmp, "design.mat"), "noise": "white", "method": "nlls",
                                "save-mean": True, "save-std": True, "max-iterations": 10})
        for k in range(P):
            assert np.allclose(out["mean_Parameter_%d" % (k + 1)], want[..., k], rtol=1e-4, atol=1e-4), k
        # what has no wide form says so: spatial VB
        with pytest.raises(Exception, match="parameters"):
            fabber.run(data, {"model": "linear", "basis": os.path.join(tmp, "design.mat"), "noise": "white", "method": "spatialvb",
                              "param-spatial-priors": "M+", "max-iterations": 3})


@pytest.mark.gpu
def test_more_than_32_parameters_under_ar_noise_and_nlls():
    """The parameter table (fvb_config.params_ext) is read by the wave-per-voxel AR(1) kernel and by the wave-per-voxel
    minimiser of method=nlls as well: 36 cosine regressors over 60 timepoints (what the AR kernel's three [T][P] work
    areas leave room for in 160 KB of LDS), one and two echoes; 40 regressors over 100 timepoints for NLLS."""
    import oracle
    import parity
    rng = np.random.default_rng(15)
    T, P, V = 60, 36, 200
    tt = np.arange(T)
    X = np.stack([np.cos(np.pi * (tt + 0.5) * k / T) for k in range(P)], axis=1)
    theta = rng.normal(0, 3, (P, V))
    e = rng.normal(0, 0.3, (T, V))
    for t in range(1, T):
        e[t] += 0.3 * e[t - 1]
    y = (X @ theta + e).astype(np.float32)
    for kw in (dict(), dict(num_echoes=2, ar_cross_terms="dual")):
        h = vbabi.build_config(vbabi.MODEL_LINEAR, V, T, design=X, max_iterations=5, need_f=True, noise=vbabi.NOISE_AR1,
                               param_overrides={"Parameter_30": dict(type="A")}, **kw)
        assert h.cfg.n_params == 36 and h.cfg.params_ext and "wave" in hiplib.kernel_name(h)
        cpu, got = oracle.run(h, y), hiplib.run_host(h, y)
        parity.strict(h, cpu, got, what="36 parameters, AR(1) %s" % (kw or "one echo"), check_f=True, cpu2=oracle.run_fma(h, y), allow_floor=True)
    T, P = 100, 40
    tt = np.arange(T)
    X = np.stack([np.cos(np.pi * (tt + 0.5) * k / T) for k in range(P)], axis=1)
    y = (X @ rng.normal(0, 3, (P, V)) + rng.normal(0, 0.3, (T, V))).astype(np.float32)
    h = vbabi.build_config(vbabi.MODEL_LINEAR, V, T, design=X)
    import test_nlls
    test_nlls.assert_parity(h, y, variant="auto")  # (as every NLLS parity test: means within 1e-4 sd, 99 % within 1e-6, costs 1e-7)
    got = hiplib.nlls_run_host(h, y)
    rows = vbabi.mvn_rows(P)
    off = P * (P + 1) // 2
    want = np.linalg.lstsq(X, y.astype(np.float64), rcond=None)[0]
    assert np.allclose(got["mvn"][off:off + P], want, rtol=1e-4, atol=1e-5)  # (the minimiser stops at its own tolerance)
    assert got["mvn"].shape[0] == rows
