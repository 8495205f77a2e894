"""The drop-in boundary: libfabbercore_amd.so through the reference's C ABI (include/fabber_capi.h).

CPU part (-m "not gpu"): the library loads, exports every declared symbol, honours the error
conventions of the reference's fabber_capi.cc, resolves options / parameters on the host, and
refuses loudly to run without a GPU. GPU part (-m gpu): the reference's known-answer tests
(test/test_inference.cc, test/test_vb.cc) driven through the C ABI like its Python wrapper does.
"""
import ctypes as C
import os
import re

import numpy as np
import pytest

import oracle
from fabber_core_amd import fabber, hiplib, vbabi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
needs_lib = pytest.mark.skipif(not os.path.exists(fabber.DEFAULT_LIB), reason="libfabbercore_amd.so not built")
VAL = np.float32(7.32)


def volume(shape, series):
    """Constant-in-space volume [x,y,z,t] from one time series."""
    return np.broadcast_to(np.asarray(series, dtype=np.float32), tuple(shape) + (len(series),)).copy()


# ---------------------------------------------------------------------------------------------
# CPU: symbols, conventions, host logic
# ---------------------------------------------------------------------------------------------
@needs_lib
def test_library_exports_every_declared_symbol():
    for header, libname in (("fabber_capi.h", "libfabbercore_amd.so"), ("fabber_vb.h", "libfabber_vb_hip.so")):
        text = open(os.path.join(ROOT, "include", header)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        names = set(re.findall(r"\b(fabber_[a-z_]+)\s*\(", text))
        assert len(names) >= 10
        L = C.CDLL(os.path.join(ROOT, "fabber_core_amd", "lib", libname))
        for n in sorted(names):
            assert hasattr(L, n), (libname, n)
    assert set(fabber.CAPI_SYMBOLS) <= set(re.findall(r"\b(fabber_[a-z_]+)\s*\(", open(os.path.join(ROOT, "include", "fabber_capi.h")).read()))


@needs_lib
def test_error_conventions():
    L = fabber.load_library()
    err = C.create_string_buffer(255)
    assert L.fabber_set_opt(None, b"a", b"b", err) == fabber.ERR_FATAL and b"NULL" in err.value
    with fabber.Fabber() as fab:
        assert L.fabber_set_extent(fab.handle, 2, 2, 2, None, err) == fabber.ERR_FATAL  # NULL mask rejected
        assert L.fabber_set_extent(fab.handle, 0, 2, 2, (C.c_int * 8)(), err) == fabber.ERR_FATAL
        assert L.fabber_set_opt(fab.handle, None, b"x", err) == fabber.ERR_FATAL
        assert L.fabber_get_data_size(fab.handle, b"nonexistent", err) == -1
        assert b"Voxel data not found" in err.value
        small = C.create_string_buffer(4)
        assert L.fabber_get_models(fab.handle, 4, small, err) == -1 and b"Buffer too small" in err.value
        assert L.fabber_dorun(fab.handle, 10, None, err, None) == fabber.ERR_FATAL
        assert L.fabber_dorun(fab.handle, 10, C.create_string_buffer(10), None, None) == fabber.ERR_FATAL
        # err_buf is optional everywhere else
        assert L.fabber_set_opt(fab.handle, b"model", b"poly", None) == 0
        L.fabber_destroy(None)  # ignored


@needs_lib
def test_registry_and_option_tables():
    with fabber.Fabber() as fab:
        assert {"poly", "linear", "exp"} <= set(fab.get_models())
        assert {"vb", "spatialvb"} <= set(fab.get_methods())
        desc, opts = fab.get_options()
        assert any(o["name"] == "save-mvn" and o["type"] == "BOOL" for o in opts)
        desc, opts = fab.get_options("model", "exp")
        assert "exponential" in desc
        assert [o["name"] for o in opts] == ["dt", "num-exps"] and opts[1]["default"] == "1" and opts[1]["optional"]
        desc, opts = fab.get_options("method", "vb")
        assert any(o["name"] == "max-iterations" for o in opts)
        with pytest.raises(fabber.FabberError):
            fab.get_options("model", "nosuchmodel")


@needs_lib
def test_model_parameters_and_prior_grammar():
    with fabber.Fabber() as fab:
        fab.set_options({"model": "exp", "num-exps": 2, "dt": 0.02})
        assert fab.get_model_params() == ["amp1", "r1", "amp2", "r2"]
        assert fab.get_model_outputs() == []
    with fabber.Fabber() as fab:
        fab.set_options({"model": "poly", "degree": 2, "param-spatial-priors": "NA+"})
        assert fab.get_model_params() == ["c0", "c1", "c2"]
    for bad in ("NNNN", "N++"):  # too many types / two '+': priors.cc:35-92, test_priors.cc:125-142
        with fabber.Fabber() as fab:
            fab.set_options({"model": "poly", "degree": 2, "param-spatial-priors": bad})
            with pytest.raises(fabber.FabberError):
                fab.get_model_params()
    with fabber.Fabber() as fab:
        fab.set_options({"model": "poly"})  # mandatory option missing
        with pytest.raises(fabber.FabberError, match="degree"):
            fab.get_model_params()


@needs_lib
def test_model_evaluate_matches_oracle_models():
    with fabber.Fabber() as fab:
        fab.set_options({"model": "exp", "num-exps": 2, "dt": 0.02})
        got = fab.model_evaluate([1.0, 0.8, 0.5, 6.0], 100)
    h = vbabi.build_config(vbabi.MODEL_EXP, 1, 100, num_exps=2, dt=0.02,
                           param_overrides={k: dict(transform="I") for k in ("amp1", "r1", "amp2", "r2")})
    ref = np.zeros(100)
    p = np.array([1.0, 0.8, 0.5, 6.0])
    assert oracle.lib().oracle_evaluate_fabber(C.byref(h.cfg), p.ctypes.data, ref.ctypes.data) == 0
    assert np.allclose(got, ref, rtol=1e-6)
    with fabber.Fabber() as fab:
        fab.set_options({"model": "poly", "degree": 3})
        got = fab.model_evaluate([2, 0, 3, -4], 10, indata=np.zeros(10))
    n = np.arange(1, 11)
    assert np.allclose(got, 2 + 3 * n ** 2 - 4 * n ** 3)


@needs_lib
def test_data_round_trip_through_mask():
    """set_data gathers through the mask, get_data scatters back with zeros outside
    (rundata_array.cc:68-133); voxel order is x fastest."""
    rng = np.random.default_rng(0)
    shape = (4, 3, 2)
    mask = rng.integers(0, 2, shape)
    vol = rng.normal(size=shape + (5,)).astype(np.float32)
    with fabber.Fabber() as fab:
        fab.set_extent(shape, mask)
        fab.set_data("thing", vol)
        assert fab.data_size("thing") == 5
        back = fab.get_data("thing")
        assert np.array_equal(back, vol * (mask != 0)[..., None])
        coords = fab.get_data("coords")
        assert fab.data_size("coords") == 3
        for d in range(3):  # 0-based grid indices (rundata_array.cc:54-56)
            assert np.array_equal(coords[..., d][mask != 0], np.nonzero(mask)[d])


@needs_lib
@pytest.mark.skipif(hiplib.available() and hiplib.device_count() > 0, reason="a GPU is present")
def test_run_without_gpu_fails_loudly():
    with fabber.Fabber() as fab:
        fab.set_extent((2, 2, 2))
        fab.set_options({"model": "poly", "degree": 0, "noise": "white", "method": "vb"})
        fab.set_data("data", volume((2, 2, 2), [VAL] * 10))
        with pytest.raises(fabber.FabberError) as e:
            fab.run()
        assert e.value.code == fabber.ERR_FATAL
        assert "no HIP device" in e.value.message and "no CPU fallback" in e.value.message


def test_noise_distribution_files_are_checked_before_the_engine_runs(tmp_path):
    """noise-initial-prior / -posterior: what Ar1cParams / WhiteParams::InputFromMVN reject (noisemodel_ar.cc:302-316,
    noisemodel_white.cc:70-79, dist_mvn.cc:157-165) is rejected while the configuration is built - no device needed"""
    data = volume((2, 2, 2), [VAL + 0.1 * i for i in range(20)])

    def write(path, mean, cov):
        n = len(mean)
        m = np.zeros((n + 1, n + 1))
        m[:n, :n], m[:n, n], m[n, :n], m[n, n] = cov, mean, mean, 1
        path.write_text("\n".join(" ".join("%.17g" % x for x in row) for row in m) + "\n")

    def run(noise_opts, f):
        with fabber.Fabber() as fab:
            fab.set_extent((2, 2, 2))
            fab.set_options(dict({"model": "poly", "degree": 1, "method": "vb", "noise-initial-prior": str(f)}, **noise_opts))
            fab.set_data("data", data)
            fab.run()

    f = tmp_path / "dist.mat"
    ar2 = {"noise": "ar", "num-echoes": 2, "ar1-cross-terms": "same"}       # 3 alphas + 2 precisions
    cov = np.diag([0.1, 0.1, 0.1, 2.0, 2.0])
    mean = [0.1, -0.1, 0.0, 4.0, 4.0]
    cov[0, 3] = cov[3, 0] = 0.01
    write(f, mean, cov)
    with pytest.raises(fabber.FabberError, match="independent"):
        run(ar2, f)
    cov[0, 3] = cov[3, 0] = 0
    cov[3, 4] = cov[4, 3] = 0.5
    write(f, mean, cov)
    with pytest.raises(fabber.FabberError, match="zero covariance"):
        run(ar2, f)
    cov[3, 4] = cov[4, 3] = 0
    write(f, mean[:4], cov[:4, :4])
    with pytest.raises(fabber.FabberError, match="entries"):
        run(ar2, f)
    cov[1, 1] = -0.1
    write(f, mean, cov)
    with pytest.raises(fabber.FabberError, match="positive definite"):
        run(ar2, f)
    cov[1, 1] = 0.1
    write(f, mean[:3] + [-4.0, 4.0], cov)
    with pytest.raises(fabber.FabberError, match="positive mean and a positive variance"):
        run(ar2, f)
    write(f, [4.0, 4.0], np.array([[2.0, 0.3], [0.3, 2.0]]))
    with pytest.raises(fabber.FabberError, match="zero covariance"):
        run({"noise": "white", "noise-pattern": "12"}, f)
    # a well-formed file gets as far as the engine
    write(f, mean, cov)
    if not hiplib_has_device():
        with pytest.raises(fabber.FabberError, match="no HIP device"):
            run(ar2, f)


def hiplib_has_device():
    from fabber_core_amd import hiplib
    L = hiplib.lib()
    L.fabber_vb_device_count.restype = C.c_int32
    return L.fabber_vb_device_count() > 0


# ---------------------------------------------------------------------------------------------
# GPU: the reference's known-answer tests through the C ABI
# ---------------------------------------------------------------------------------------------
def run_poly(data, degree, mask=None, extra=None, **options):
    opts = {"model": "poly", "degree": degree, "noise": "white", "method": "vb", "save-mean": True, "save-mvn": True}
    opts.update({(k if k.startswith("PSP_") else k.replace("_", "-")): v for k, v in options.items()})
    return fabber.run(data, opts, mask=mask, extra_data=extra)


@pytest.mark.gpu
def test_spatial_priors_through_the_c_abi():
    """method=spatialvb with param-spatial-priors / PSP_byname types (test_inference.cc runs every
    case for vb and spatialvb; test_priors.cc grammar): same numbers as the engine called directly
    on the masked voxels, one progress call per spatial iteration (inference_vb.cc:610)."""
    rng = np.random.default_rng(11)
    shape, T = (9, 8, 5), 50
    t = np.arange(T) * 0.04
    gx = np.arange(shape[0])[:, None, None]
    amp = 1.0 + 0.3 * np.sin(gx / 2.0) + np.zeros(shape)
    data = (amp[..., None] * np.exp(-t) + rng.normal(0, 0.1, shape + (T,))).astype(np.float32)
    mask = rng.random(shape) < 0.85
    opts = {"model": "exp", "dt": 0.04, "noise": "white", "method": "spatialvb", "max-iterations": 6, "save-mean": True,
            "save-mvn": True, "param-spatial-priors": "MN", "save-free-energy": True}
    calls = []
    out = fabber.run(data, opts, mask=mask, progress_cb=lambda i, n: calls.append((i, n)))
    assert [c for c in calls if c[1] == 6] == [(i, 6) for i in range(6)]
    sel = mask.transpose(2, 1, 0).ravel()
    y = data.transpose(3, 2, 1, 0).reshape(T, -1)[:, sel].astype(np.float64)
    h = vbabi.build_config(vbabi.MODEL_EXP, int(mask.sum()), T, num_exps=1, dt=0.04, max_iterations=6, need_f=True,
                           param_overrides={"amp1": dict(type="M")})
    direct = hiplib.run_spatial_host(h, vbabi.SpatialHolder(vbabi.grid_coords(shape, mask)), y)
    got = out["finalMVN"].transpose(3, 2, 1, 0).reshape(10, -1)[:, sel]
    assert np.allclose(got, direct["mvn"].astype(np.float32), rtol=1e-6, atol=0)
    F = out["freeEnergy"].transpose(2, 1, 0).ravel()[sel]
    assert np.allclose(F, direct["free_energy"], rtol=1e-6)
    # the same prior chosen by name, without method=spatialvb: Vb::IsSpatial switches loops
    byname = dict(opts, method="vb", PSP_byname1="amp1", PSP_byname1_type="M")
    del byname["param-spatial-priors"]
    out2 = fabber.run(data, byname, mask=mask)
    assert np.array_equal(out2["finalMVN"], out["finalMVN"])
    # and it differs from the voxelwise answer
    plain = fabber.run(data, dict(byname, PSP_byname1_type="N"), mask=mask)
    assert not np.allclose(plain["mean_amp1"], out["mean_amp1"])


@pytest.mark.gpu
@pytest.mark.parametrize("method", ["vb", "spatialvb"])
def test_constant_and_alternating_data(method):
    """test_inference.cc:108-238"""
    out = run_poly(volume((5, 5, 5), [VAL] * 10), 0, method=method)
    assert out["mean_c0"].shape == (5, 5, 5)
    assert np.all(np.abs(out["mean_c0"] - VAL) <= 4 * np.spacing(VAL))
    series = [VAL if n % 2 == 0 else VAL * np.float32(3) for n in range(10)]
    out = run_poly(volume((5, 5, 5), series), 0, method=method)
    assert np.all(np.abs(out["mean_c0"] - VAL * 2) <= 4 * np.spacing(VAL * 2))


@pytest.mark.gpu
def test_no_voxels():
    """test_inference.cc:57-74 / 'some chunks have no ROI': an empty mask gives empty outputs."""
    data = volume((3, 3, 3), [VAL] * 10)
    out = run_poly(data, 0, mask=np.zeros((3, 3, 3), dtype=int))
    assert np.all(out["mean_c0"] == 0)


@pytest.mark.gpu
def test_polynomial_fit_and_model_fit_output():
    """test_inference.cc:353-482"""
    n = np.arange(1, 11, dtype=np.float64)
    series = 2 + 3 * n * n - 4 * n * n * n
    out = run_poly(volume((5, 5, 5), series), 3, max_iterations=50, save_model_fit=True, save_residuals=True)
    for name, want in (("mean_c0", 2), ("mean_c1", 0), ("mean_c2", 3), ("mean_c3", -4)):
        assert np.all(np.abs(out[name] - want) < 1e-3), name
    assert out["modelfit"].shape == (5, 5, 5, 10)
    assert np.allclose(out["modelfit"][0, 0, 0], series, rtol=1e-5)
    assert np.all(np.abs(out["residuals"]) < 0.05)


@pytest.mark.gpu
def test_masked_timepoints():
    """test_inference.cc:485-561"""
    series = np.full(10, 2.0)
    out = run_poly(volume((5, 5, 5), series), 1)
    assert np.all(np.abs(out["mean_c0"] - 2) < 1e-3)
    series[2] = series[6] = 4.0
    assert np.all(run_poly(volume((5, 5, 5), series), 1)["mean_c0"] > 2)
    out = run_poly(volume((5, 5, 5), series), 1, mt1=3, mt2=7)
    assert np.all(np.abs(out["mean_c0"] - 2) < 1e-3)


@pytest.mark.gpu
def test_restart_chain_through_float_mvn():
    """test_vb.cc:305-409 through the C ABI: the MVN travels as float32 between the 51 runs."""
    n = np.arange(1, 11, dtype=np.float64)
    data = volume((3, 3, 3), float(VAL) + 1.5 * float(VAL) * n * n)
    out = run_poly(data, 5, max_iterations=1)
    assert out["mean_c0"][0, 0, 0] != VAL
    for _ in range(50):
        out = run_poly(data, 5, max_iterations=1, extra={"mvns": out["finalMVN"]}, continue_from_mvn="mvns")
    assert np.all(np.abs(out["mean_c0"] - VAL) < 2e-3)
    assert np.all(np.abs(out["mean_c1"]) < 2e-3)
    assert np.all(np.abs(out["mean_c2"] - 1.5 * VAL) < 1e-4)


@pytest.mark.gpu
def test_output_only_from_mvn():
    """test_vb.cc:412-498"""
    n = np.arange(1, 11, dtype=np.float64)
    data = volume((3, 3, 3), float(VAL) + 1.5 * float(VAL) * n * n)
    first = run_poly(data, 2, max_iterations=100)
    out = fabber.run(data, {"model": "poly", "degree": 2, "noise": "white", "method": "vb", "save-mean": True,
                            "save-model-fit": True, "output-only": True, "continue-from-mvn": "mvns"},
                     extra_data={"mvns": first["finalMVN"]})
    assert "finalMVN" not in out
    assert np.all(np.abs(out["mean_c0"] - VAL) < 1e-4)
    assert np.all(np.abs(out["mean_c1"]) < 1e-4)
    assert np.all(np.abs(out["mean_c2"] - 1.5 * VAL) < 1e-4)
    assert out["modelfit"].shape == (3, 3, 3, 10)


@pytest.mark.gpu
def test_spatialvb_restart_output_only():
    """test_spatialvb.cc:585-670 (RestartOutputOnly)"""
    n = np.arange(1, 11, dtype=np.float64)
    data = volume((5, 5, 5), float(VAL) + 1.5 * float(VAL) * n * n)
    first = fabber.run(data, {"model": "poly", "degree": 2, "noise": "white", "method": "spatialvb", "max-iterations": 50,
                              "save-mvn": True})
    out = fabber.run(data, {"model": "poly", "degree": 2, "noise": "white", "method": "spatialvb", "save-mean": True,
                            "save-model-fit": True, "output-only": True, "continue-from-mvn": "mvns"},
                     extra_data={"mvns": first["finalMVN"]})
    assert "finalMVN" not in out
    assert out["mean_c0"].shape == (5, 5, 5)
    assert np.all(np.abs(out["mean_c0"] - VAL) < 1e-4)
    assert np.all(np.abs(out["mean_c1"]) < 1e-4)
    assert np.all(np.abs(out["mean_c2"] - 1.5 * VAL) < 1e-4)
    assert out["modelfit"].shape == (5, 5, 5, 10)


@pytest.mark.gpu
def test_image_priors():
    """test_vb.cc:71-232"""
    series = [VAL if n % 2 == 0 else VAL * np.float32(3) for n in range(10)]
    data = volume((5, 5, 5), series)
    img = np.full((5, 5, 5), VAL * 1.5, dtype=np.float32)
    base = {"PSP_byname1": "c0", "PSP_byname1_type": "I"}
    out = run_poly(data, 0, extra={"PSP_byname1_image": img}, **base, PSP_byname1_prec="1e12")
    assert np.all(np.abs(out["mean_c0"] - VAL * 1.5) < VAL * 0.1)
    out = run_poly(data, 0, extra={"PSP_byname1_image": img}, **base, PSP_byname1_prec="1e-5")
    assert np.all(np.abs(out["mean_c0"] - VAL * 2) < VAL * 0.1)


@pytest.mark.gpu
def test_outputs_gated_by_save_options_and_match_engine():
    rng = np.random.default_rng(5)
    shape, T = (6, 5, 4), 50
    t = np.arange(T) * 0.04
    data = (np.exp(-t)[None, None, None, :] + rng.normal(0, 0.05, shape + (T,))).astype(np.float32)
    mask = rng.integers(0, 4, shape) > 0
    opts = {"model": "exp", "dt": 0.04, "noise": "white", "method": "vb", "max-iterations": 10, "save-mean": True,
            "save-std": True, "save-zstat": True, "save-var": True, "save-mvn": True, "save-noise-mean": True,
            "save-noise-std": True, "save-free-energy": True, "save-free-energy-history": True}
    out = fabber.run(data, opts, mask=mask)
    assert out["finalMVN"].shape == shape + (10,)
    assert np.all(out["finalMVN"][~mask] == 0) and np.all(out["finalMVN"][mask][:, -1] == 1)
    assert out["freeEnergyHistory"].shape[3] == 11
    assert np.allclose(out["zstat_amp1"][mask], out["mean_amp1"][mask] / out["std_amp1"][mask], rtol=1e-5)
    assert np.allclose(out["var_r1"][mask], out["std_r1"][mask] ** 2, rtol=1e-5)
    assert "modelfit" not in out and "residuals" not in out
    # same numbers as the engine called directly on the masked voxels
    from fabber_core_amd import hiplib
    V = int(mask.sum())
    y = data.transpose(3, 2, 1, 0).reshape(T, -1)[:, mask.transpose(2, 1, 0).ravel()]
    h = vbabi.build_config(vbabi.MODEL_EXP, V, T, num_exps=1, dt=0.04, max_iterations=10, need_f=True)
    direct = hiplib.run_host(h, y.astype(np.float64))
    got = out["finalMVN"].transpose(3, 2, 1, 0).reshape(10, -1)[:, mask.transpose(2, 1, 0).ravel()]
    assert np.allclose(got, direct["mvn"].astype(np.float32), rtol=1e-6, atol=0)
    F = out["freeEnergy"].transpose(2, 1, 0).ravel()[mask.transpose(2, 1, 0).ravel()]
    assert np.allclose(F, direct["free_energy"], rtol=1e-6)


@pytest.mark.gpu
def test_bad_voxels_halt_or_continue():
    """inference.cc:93-109, inference_vb.cc:529-544: a numerical failure is fatal unless
    allow-bad-voxels is set; a failure of the very first re-linearisation is always fatal."""
    rng = np.random.default_rng(1)
    T = 50
    t = np.arange(T) * 0.04
    data = (np.exp(-t)[None, None, None, :] + rng.normal(0, 0.05, (4, 4, 2, T))).astype(np.float32)
    data[1, 1, 1, 3] = np.nan  # becomes non-finite inside the loop (iteration 1)
    opts = {"model": "exp", "dt": 0.04, "noise": "white", "method": "vb", "save-mean": True}
    with pytest.raises(fabber.FabberError, match="Non-finite"):
        fabber.run(data, opts)
    out = fabber.run(data, dict(opts, **{"allow-bad-voxels": True}))
    assert "numerical errors" in out["log"]
    assert np.isfinite(out["mean_amp1"][0, 0, 0])
    data[2, 2, 0, :] = -1.0  # log of a negative amplitude: fails in setup -> fatal regardless
    with pytest.raises(fabber.FabberError, match="Non-finite"):
        fabber.run(data, dict(opts, **{"allow-bad-voxels": True}))


@pytest.mark.gpu
def test_ar_noise_through_the_c_abi():
    """test_inference.cc:564-633 (MaskedTimepointsArNoise) + noise image quirk: with the AR
    model `noise_means` holds alpha_1 (noisemodel_ar.cc:362-365, inference_vb.cc:981-989)."""
    series = np.full(10, 2.0)
    opts = {"model": "poly", "degree": 1, "noise": "ar", "method": "vb", "max-iterations": 10, "save-mean": True,
            "save-mvn": True, "save-noise-mean": True}
    out = fabber.run(volume((5, 5, 5), series), opts)
    assert np.all(np.abs(out["mean_c0"] - 2) < 1e-3)
    assert out["finalMVN"].shape[3] == vbabi.mvn_rows(2 + 3)
    assert out["noise_means"].shape == (5, 5, 5)
    n = 5
    assert np.allclose(out["noise_means"], out["finalMVN"][..., n * (n + 1) // 2 + 2])
    series[2] = series[6] = 4.0
    assert np.all(fabber.run(volume((5, 5, 5), series), opts)["mean_c0"] > 2)
    with pytest.raises(fabber.FabberError, match="Masked time points are not supported"):
        fabber.run(volume((5, 5, 5), series), dict(opts, mt1=3, mt2=7))


@pytest.mark.gpu
def test_progress_callback_and_unused_option_warning():
    calls = []
    out = fabber.run(volume((2, 2, 2), [VAL] * 10), {"model": "poly", "degree": 0, "noise": "white", "method": "vb",
                                                     "save-mean": True, "bogus-option": "1"},
                     progress_cb=lambda v, n: calls.append((v, n)))
    assert calls[0] == (0, 8) and calls[-1] == (8, 8)
    assert "Unused option specified: bogus-option" in out["log"]


@pytest.mark.gpu
def test_noise_distributions_from_matrix_files(tmp_path):
    """noise-initial-prior / noise-initial-posterior (Vb::InitializeNoiseFromParam, inference_vb.cc:132-142): one MVN
    file for every voxel, a Gamma per precision from its mean and variance. prior-noise-stddev=0.5 is the Gamma with
    c = 0.5, b = 8 as prior AND initial posterior (noisemodel_white.cc:151-162): mean 4, variance 32 - the same run
    from files."""
    rng = np.random.default_rng(10)
    shape, T = (5, 4, 3), 20
    t = np.arange(1, T + 1)
    data = (2.0 + 0.3 * t + rng.normal(0, 0.5, shape + (T,))).astype(np.float32)
    opts = {"model": "poly", "degree": 1, "noise": "white", "method": "vb", "max-iterations": 6, "save-mean": True, "save-mvn": True,
            "save-noise-mean": True}
    by_option = fabber.run(data, dict(opts, **{"prior-noise-stddev": 0.5}))
    f = tmp_path / "noise_mvn.mat"
    f.write_text("32 4\n4 1\n")
    by_file = fabber.run(data, dict(opts, **{"noise-initial-prior": str(f), "noise-initial-posterior": str(f)}))
    assert "Loading noise-initial-prior distribution" in by_file["log"]
    assert np.allclose(by_file["finalMVN"], by_option["finalMVN"], rtol=1e-12, atol=0)
    default = fabber.run(data, opts)
    assert np.abs(default["noise_means"] - by_file["noise_means"]).max() > 1e-3
    # only the prior from the file: a different run again
    only_prior = fabber.run(data, dict(opts, **{"noise-initial-prior": str(f)}))
    assert np.abs(only_prior["finalMVN"] - by_file["finalMVN"]).max() > 0
    with pytest.raises(fabber.FabberError):
        fabber.run(data, dict(opts, noise="ar", **{"noise-initial-prior": str(f)}))
    # what WhiteParams::InputFromMVN rejects (noisemodel_white.cc:70-79): precisions with a covariance between them;
    # and what no Gamma distribution has: a non-positive mean or variance
    two = tmp_path / "two.mat"
    two.write_text("32 1 4\n1 32 4\n4 4 1\n")
    with pytest.raises(fabber.FabberError, match="zero covariance"):
        fabber.run(data, dict(opts, **{"noise-pattern": "12", "noise-initial-prior": str(two)}))
    two.write_text("32 0 4\n0 32 4\n4 4 1\n")
    assert np.isfinite(fabber.run(data, dict(opts, **{"noise-pattern": "12", "noise-initial-prior": str(two)}))["mean_c0"]).all()
    for text in ("32 -4\n-4 1\n", "0 4\n4 1\n"):
        f.write_text(text)
        with pytest.raises(fabber.FabberError, match="positive mean and a positive variance"):
            fabber.run(data, dict(opts, **{"noise-initial-prior": str(f)}))


@pytest.mark.gpu
@pytest.mark.parametrize("echoes,cross", [(1, "none"), (2, "dual")])
def test_ar_noise_distributions_from_matrix_files(tmp_path, echoes, cross):
    """noise-initial-prior / noise-initial-posterior under AR(1) noise (Ar1cParams::InputFromMVN, noisemodel_ar.cc:302-316):
    the file holds the alphas' MVN block first, then one Gamma per echo by mean and variance. (1) The hard-coded
    distributions (noisemodel_ar.cc:391-401) written to files give the default run; (2) an informative prior with
    covariance between the alphas gives what the oracle computes from the same distributions; (3) what InputFromMVN
    rejects is rejected."""
    import oracle
    rng = np.random.default_rng(12)
    shape, T = (4, 3, 3), 40
    t = np.arange(1, T + 1)
    data = (2.0 + 0.3 * t + rng.normal(0, 0.5, shape + (T,))).astype(np.float32)
    nA = 2 + {"none": 0, "same": 1, "dual": 2}[cross]
    n = nA + echoes
    opts = {"model": "poly", "degree": 1, "noise": "ar", "method": "vb", "max-iterations": 5, "save-mean": True, "save-mvn": True,
            "num-echoes": echoes, "ar1-cross-terms": cross}

    def write(path, mean, cov):
        m = np.zeros((n + 1, n + 1))
        m[:n, :n], m[:n, n], m[n, :n], m[n, n] = cov, mean, mean, 1
        path.write_text("\n".join(" ".join("%.17g" % x for x in row) for row in m) + "\n")

    default = fabber.run(data, opts)
    prior_f, post_f = tmp_path / "prior.mat", tmp_path / "post.mat"
    # Gamma(b, c): mean bc, variance b^2 c; prior b = 1e6, c = 1e-6, posterior b = 1e-8, c = 1e-6
    write(prior_f, [0] * nA + [1.0] * echoes, np.diag([1e4] * nA + [1e6] * echoes))
    write(post_f, [0] * nA + [1e-14] * echoes, np.diag([1e4] * nA + [1e-22] * echoes))
    same = fabber.run(data, dict(opts, **{"noise-initial-prior": str(prior_f), "noise-initial-posterior": str(post_f)}))
    assert "Loading noise-initial-prior distribution" in same["log"]
    assert np.allclose(same["finalMVN"], default["finalMVN"], rtol=2e-5, atol=1e-9)

    a = rng.normal(size=(nA, nA))
    cov_alpha = 0.02 * (a @ a.T) + 0.1 * np.eye(nA)
    mean_alpha = rng.uniform(-0.3, 0.3, nA)
    cov = np.zeros((n, n))
    cov[:nA, :nA] = cov_alpha
    cov[nA:, nA:] = np.diag([2.0] * echoes)                   # Gamma with mean 4, variance 2: b = 0.5, c = 8
    write(prior_f, list(mean_alpha) + [4.0] * echoes, cov)
    out = fabber.run(data, dict(opts, **{"noise-initial-prior": str(prior_f)}))
    assert np.abs(out["finalMVN"] - default["finalMVN"]).max() > 1e-3
    V = int(np.prod(shape))
    y = data.transpose(3, 2, 1, 0).reshape(T, -1).astype(np.float64)
    prec_alpha = np.linalg.inv(cov_alpha)
    prec_alpha = 0.5 * (prec_alpha + prec_alpha.T)
    h = vbabi.build_config(vbabi.MODEL_POLY, V, T, degree=1, max_iterations=5, noise=vbabi.NOISE_AR1, num_echoes=echoes,
                           ar_cross_terms=cross, ar_alpha_prior=(mean_alpha, prec_alpha))
    for e in range(echoes):
        h.cfg.noise_prior_b[e], h.cfg.noise_prior_c[e] = 0.5, 8.0
    ref = oracle.run(h, y)
    rows = vbabi.mvn_rows(2 + n)
    got = out["finalMVN"].transpose(3, 2, 1, 0).reshape(rows, -1)
    assert np.all(ref["status"] == 0)
    assert np.allclose(got, ref["mvn"], rtol=2e-5, atol=1e-7)

    cov[0, nA] = cov[nA, 0] = 0.01                              # alpha block not independent of the precisions (dist_mvn.cc:157-165)
    write(prior_f, list(mean_alpha) + [4.0] * echoes, cov)
    with pytest.raises(fabber.FabberError, match="independent"):
        fabber.run(data, dict(opts, **{"noise-initial-prior": str(prior_f)}))
    if echoes == 2:
        cov[0, nA] = cov[nA, 0] = 0
        cov[nA, nA + 1] = cov[nA + 1, nA] = 0.5
        write(prior_f, list(mean_alpha) + [4.0] * echoes, cov)
        with pytest.raises(fabber.FabberError, match="zero covariance"):
            fabber.run(data, dict(opts, **{"noise-initial-prior": str(prior_f)}))
    short = tmp_path / "short.mat"
    short.write_text("32 4\n4 1\n")                              # one entry: the white noise model's file
    with pytest.raises(fabber.FabberError, match="entries"):
        fabber.run(data, dict(opts, **{"noise-initial-prior": str(short)}))
    # under spatial VB too: the hard-coded distributions from files give the default spatial run
    write(prior_f, [0] * nA + [1.0] * echoes, np.diag([1e4] * nA + [1e6] * echoes))
    sp_opts = dict(opts, method="spatialvb", **{"param-spatial-priors": "MN"})
    sp_default = fabber.run(data, sp_opts)
    sp_file = fabber.run(data, dict(sp_opts, **{"noise-initial-prior": str(prior_f), "noise-initial-posterior": str(post_f)}))
    assert np.allclose(sp_file["finalMVN"], sp_default["finalMVN"], rtol=2e-5, atol=1e-9)
