"""ctypes binding of fabber_core_amd/lib/libfabber_vb_hip.so (the HIP voxelwise-VB engine).

There is deliberately no fallback: if the shared library is missing or no GPU is visible the
functions raise. Build the library with ``python -m fabber_core_amd.build``.
"""
import ctypes as C
import os

import numpy as np

from . import vbabi

_HERE = os.path.dirname(os.path.abspath(__file__))
# (FVB_LIB_PATH: an experiment build of the engine, see fabber_core_amd/build.py --variant)
LIB_PATH = os.environ.get("FVB_LIB_PATH") or os.path.join(_HERE, "lib", "libfabber_vb_hip.so")
_LIB = None


class HipEngineError(RuntimeError):
    pass


def available():
    """True if the native library has been built (says nothing about a GPU being present)."""
    return os.path.exists(LIB_PATH)


def lib():
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise HipEngineError("%s not built: run `python -m fabber_core_amd.build` (no CPU fallback exists)" % LIB_PATH)
        from . import single_hip_runtime
        single_hip_runtime()
        L = C.CDLL(LIB_PATH)
        cfgp, outp, ppp = C.POINTER(vbabi.FvbConfig), C.POINTER(vbabi.FvbOutputs), C.POINTER(vbabi.FvbPostproc)
        L.fabber_vb_mvn_rows.restype = C.c_int32
        L.fabber_vb_mvn_rows.argtypes = [C.c_int32]
        L.fabber_vb_abi_version.restype = C.c_int32
        L.fabber_vb_device_count.restype = C.c_int32
        L.fabber_vb_last_error.restype = C.c_char_p
        L.fabber_vb_kernel_name.restype = C.c_char_p
        L.fabber_vb_kernel_name.argtypes = [cfgp]
        L.fabber_vb_set_variant.argtypes = [C.c_int32]
        L.fabber_vb_set_residual_mode.argtypes = [C.c_int32]
        L.fabber_vb_set_residual_tolerance.argtypes = [C.c_double]
        L.fabber_vb_run_device.restype = C.c_int32
        L.fabber_vb_run_device.argtypes = [cfgp, C.c_void_p, outp, C.c_void_p]
        L.fabber_vb_run_device_ex.restype = C.c_int32
        L.fabber_vb_run_device_ex.argtypes = [cfgp, C.c_void_p, outp, C.c_void_p, C.c_int32]
        L.fabber_vb_run_host.restype = C.c_int32
        L.fabber_vb_run_host.argtypes = [cfgp, C.c_void_p, outp, C.c_int32]
        L.fabber_vb_run_host_multi.restype = C.c_int32
        L.fabber_vb_run_host_multi.argtypes = [cfgp, C.c_void_p, outp, C.POINTER(C.c_int32), C.c_int32, C.POINTER(vbabi.FvbSummary)]
        L.fabber_vb_postproc_device.restype = C.c_int32
        L.fabber_vb_postproc_device.argtypes = [cfgp, C.c_void_p, C.c_void_p, ppp, C.c_void_p]
        L.fabber_vb_postproc_host.restype = C.c_int32
        L.fabber_vb_postproc_host.argtypes = [cfgp, C.c_void_p, C.c_void_p, ppp, C.c_int32]
        L.fabber_vb_convergence_trace.restype = C.c_int32
        L.fabber_vb_convergence_trace.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.c_double, C.c_void_p, C.c_int32,
                                                  C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32]
        for name in ("fabber_vb_gammaln", "fabber_vb_digamma", "fabber_vb_exp_acc"):
            getattr(L, name).restype = C.c_double
            getattr(L, name).argtypes = [C.c_double]
        L.fabber_vb_transform.restype = C.c_double
        L.fabber_vb_transform.argtypes = [C.c_int32, C.c_int32, C.c_double]
        L.fabber_vb_ldl_inverse.restype = C.c_int32
        L.fabber_vb_ldl_inverse.argtypes = [C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.fabber_nlls_run_host.restype = C.c_int32
        L.fabber_nlls_run_host.argtypes = [cfgp, C.POINTER(vbabi.FvbNlls), C.c_void_p, outp, C.c_int32]
        L.fabber_nlls_run_device.restype = C.c_int32
        L.fabber_nlls_run_device.argtypes = [cfgp, C.POINTER(vbabi.FvbNlls), C.c_void_p, outp, C.c_void_p, C.c_int32]
        L.fabber_nlls_defaults.restype = None
        L.fabber_nlls_defaults.argtypes = [C.POINTER(vbabi.FvbNlls)]
        if L.fabber_vb_abi_version() != vbabi.FVB_ABI_VERSION:
            raise HipEngineError("libfabber_vb_hip.so ABI version mismatch: rebuild")
        _LIB = L
    return _LIB


def _check(rc):
    if rc != 0:
        raise HipEngineError("fabber_vb error %d: %s" % (rc, lib().fabber_vb_last_error().decode()))


def device_count():
    return lib().fabber_vb_device_count()


def kernel_name(holder):
    return lib().fabber_vb_kernel_name(C.byref(holder.cfg)).decode()


def set_variant(variant):
    lib().fabber_vb_set_variant({"auto": 0, "lane": 1, "wave": 2}.get(variant, variant))


def set_residual_mode(mode):
    """k'Qk: 'auto' (moments + exact fallback), 'exact' (always direct), 'moments' (never)."""
    lib().fabber_vb_set_residual_mode({"auto": 0, "exact": 1, "moments": 2}.get(mode, mode))


def set_residual_tolerance(tol):
    lib().fabber_vb_set_residual_tolerance(float(tol))


def n_unmasked(holder):
    phi = holder.keep.get("phi_index")
    return int(holder.cfg.n_times if phi is None else np.count_nonzero(phi != 255))


def _prepare_data(holder, data):
    cfg = holder.cfg
    data = np.ascontiguousarray(data)
    if data.dtype == np.float64:
        cfg.data_f64 = 1
    else:
        data = np.ascontiguousarray(data, dtype=np.float32)
        cfg.data_f64 = 0
    assert data.shape == (cfg.n_times, cfg.n_voxels), (data.shape, cfg.n_times, cfg.n_voxels)
    return data


def run_host(holder, data, device=0, devices=None, into=None):
    """Voxelwise VB on the GPU from host arrays (config pointers are host numpy arrays).
    Returns the same dict of arrays as tests/oracle.py:run. devices: "all" or a list of device indices
    (fabber_vb_run_host_multi: contiguous voxel blocks, one per entry); the result then also holds
    `summary` = (sum of F, sum of iterations, bad voxels). into: the dict a previous call returned - its arrays are
    written again instead of new ones being allocated and filled (176 MB per million voxels of the bi-exponential
    configuration: as long as the engine's share of the call)."""
    cfg = holder.cfg
    data = _prepare_data(holder, data)
    V = cfg.n_voxels
    if into is not None:
        arrs = {k: into[k] for k in ("mvn", "free_energy", "status", "iterations", "f_history_len", "f_history") if k in into}
    else:
        arrs = dict(
            mvn=np.full((holder.n_mvn_rows, V), np.nan),
            free_energy=np.full(V, np.nan),
            status=np.full(V, -1, dtype=np.int32),
            iterations=np.full(V, -1, dtype=np.int32),
            f_history_len=np.zeros(V, dtype=np.int32),
        )
        if cfg.f_history_rows > 0:
            arrs["f_history"] = np.full((cfg.f_history_rows, V), np.nan)
    out = vbabi.FvbOutputs()
    for k, a in arrs.items():
        setattr(out, k, a.ctypes.data)
    if devices is None:
        _check(lib().fabber_vb_run_host(C.byref(cfg), data.ctypes.data, C.byref(out), device))
    else:
        summary = vbabi.FvbSummary()
        if devices == "all":
            _check(lib().fabber_vb_run_host_multi(C.byref(cfg), data.ctypes.data, C.byref(out), None, 0, C.byref(summary)))
        else:
            ids = (C.c_int32 * len(devices))(*devices)
            _check(lib().fabber_vb_run_host_multi(C.byref(cfg), data.ctypes.data, C.byref(out), ids, len(devices), C.byref(summary)))
        arrs["summary"] = (summary.sum_free_energy, summary.sum_iterations, summary.bad_voxels)
    arrs["setup_failed"] = (arrs["status"] & 0x100) != 0
    arrs["status"] = arrs["status"] & 0xFF
    return arrs


def nlls_run_host(holder, data, lm=False, start=None, settings=None, device=0):
    """method=nlls on the GPU from host arrays. `start` = Fabber-space starting estimate
    (default zeros, what the built-in models' HardcodedInitialDists leaves). Returns mvn
    [mvn_rows(P)][V] (parameters only), status, iterations, cost (final sum of squares)."""
    cfg = holder.cfg
    data = _prepare_data(holder, data)
    V, P = cfg.n_voxels, cfg.n_params
    holder.set_post_mean([0.0] * P if start is None else start)
    nl = settings or vbabi.FvbNlls.defaults(lm)
    arrs = dict(mvn=np.full((vbabi.mvn_rows(P), V), np.nan), status=np.full(V, -1, dtype=np.int32),
                iterations=np.full(V, -1, dtype=np.int32), free_energy=np.full(V, np.nan))
    out = vbabi.FvbOutputs()
    for k, a in arrs.items():
        setattr(out, k, a.ctypes.data)
    _check(lib().fabber_nlls_run_host(C.byref(cfg), C.byref(nl), data.ctypes.data, C.byref(out), device))
    arrs["cost"] = arrs.pop("free_energy")
    return arrs


LINEARISE_FN = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_double), C.POINTER(C.c_double))


def recentre_callback(model, T, P):
    """fvb_linearise_fn from a Python model: model(params [n][P] Fabber space, voxel ids [n]) -> [n][T]. The offset and
    the Jacobian as LinearizedFwdModel::ReCentre computes them (fwdmodel_linear.cc:126-182): central differences,
    delta = max(|centre| 1e-5, 1e-10). Keep the returned object alive for the duration of the call."""
    def cb(user, n, ids, means, lin):
        try:
            v = np.ctypeslib.as_array(ids, (n,))
            m = np.ctypeslib.as_array(means, (n * P,)).reshape(n, P)
            out = np.ctypeslib.as_array(lin, (n * T * (P + 1),)).reshape(n, T * (P + 1))
            out[:, :T] = model(m, v)
            J = np.empty((n, T, P))
            for i in range(P):
                delta = np.maximum(np.abs(m[:, i] * 1e-5), 1e-10)
                up, dn = m.copy(), m.copy()
                up[:, i] = m[:, i] + delta
                dn[:, i] = m[:, i] - delta
                J[:, :, i] = (model(up, v) - model(dn, v)) / (up[:, i] - dn[:, i])[:, None]
            out[:, T:] = J.reshape(n, -1)
            return 0
        except Exception:  # noqa: BLE001 (the engine reports -54)
            return 1
    return LINEARISE_FN(cb)


def nlls_run_hostmodel_host(holder, data, model, lm=False, start=None, settings=None, device=0):
    """method=nlls with the model evaluated by the caller (fabber_nlls_run_hostmodel_host): `model` as for
    recentre_callback. The holder's model fields are not read."""
    cfg = holder.cfg
    data = _prepare_data(holder, data)
    V, P, T = cfg.n_voxels, cfg.n_params, cfg.n_times
    holder.set_post_mean([0.0] * P if start is None else start)
    nl = settings or vbabi.FvbNlls.defaults(lm)
    arrs = dict(mvn=np.full((vbabi.mvn_rows(P), V), np.nan), status=np.full(V, -1, dtype=np.int32),
                iterations=np.full(V, -1, dtype=np.int32), free_energy=np.full(V, np.nan))
    out = vbabi.FvbOutputs()
    for k, a in arrs.items():
        setattr(out, k, a.ctypes.data)
    cb = recentre_callback(model, T, P)
    L = lib()
    L.fabber_nlls_run_hostmodel_host.restype = C.c_int32
    L.fabber_nlls_run_hostmodel_host.argtypes = [C.POINTER(vbabi.FvbConfig), C.POINTER(vbabi.FvbNlls), C.c_void_p,
                                                 C.POINTER(vbabi.FvbOutputs), C.c_int32, LINEARISE_FN, C.c_void_p]
    saved = cfg.model
    cfg.model = vbabi.MODEL_HOSTJAC
    try:
        _check(L.fabber_nlls_run_hostmodel_host(C.byref(cfg), C.byref(nl), data.ctypes.data, C.byref(out), device, cb, None))
    finally:
        cfg.model = saved
    arrs["cost"] = arrs.pop("free_energy")
    return arrs


def postproc_host(holder, data, mvn, want=("mean", "var", "std", "zstat", "modelfit", "residuals", "noise_mean", "noise_std"),
                  device=0):
    cfg = holder.cfg
    V, T, P, N = cfg.n_voxels, cfg.n_times, cfg.n_params, holder.n_noise_outputs
    shapes = dict(mean=(P, V), var=(P, V), std=(P, V), zstat=(P, V), modelfit=(T, V), residuals=(T, V),
                  noise_mean=(N, V), noise_std=(N, V))
    data = _prepare_data(holder, data)
    mvn = np.ascontiguousarray(mvn, dtype=np.float64)
    pp = vbabi.FvbPostproc()
    arrs = {}
    for k in want:
        arrs[k] = np.full(shapes[k], np.nan)
        setattr(pp, k, arrs[k].ctypes.data)
    _check(lib().fabber_vb_postproc_host(C.byref(cfg), data.ctypes.data, mvn.ctypes.data, C.byref(pp), device))
    return arrs


def convergence_trace(conv, F, max_iterations=10, max_trials=10, min_fchange=0.01, stop_at_done=True):
    F = np.ascontiguousarray(F, dtype=np.float64)
    n = len(F)
    done = np.zeros(n, dtype=np.int32)
    save = np.zeros(n, dtype=np.int32)
    revert = np.zeros(n, dtype=np.int32)
    alpha = np.zeros(n)
    conv = vbabi.CONV_NAMES[conv] if isinstance(conv, str) else conv
    m = lib().fabber_vb_convergence_trace(conv, max_iterations, max_trials, min_fchange, F.ctypes.data, n,
                                          done.ctypes.data, save.ctypes.data, revert.ctypes.data, alpha.ctypes.data,
                                          int(stop_at_done))
    return done[:m].astype(bool), save[:m].astype(bool), revert[:m].astype(bool), alpha[:m]


def device_math(what, values):
    """Building blocks of vb_math.h evaluated on the device (fabber_vb_device_math): what = "exp_acc" | "exp" on a
    vector, or "invert4" on packed 4 x 4 symmetric matrices [n][10] -> (packed inverses [n][10], log|det| [n])."""
    L = lib()
    L.fabber_vb_device_math.restype = C.c_int32
    L.fabber_vb_device_math.argtypes = [C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]
    x = np.ascontiguousarray(values, dtype=np.float64)
    code = {"exp_acc": 0, "exp": 1, "invert4": 2}[what]
    n = x.shape[0]
    out = np.empty((n, 11) if code == 2 else n)
    _check(L.fabber_vb_device_math(code, n, x.ctypes.data, out.ctypes.data))
    return (out[:, :10], out[:, 10]) if code == 2 else out


def ldl_inverse(a):
    """Host twin of the in-register LDL^T inverse; a: symmetric PxP. Returns (inv, logabs, sign, ok)."""
    a = np.asarray(a, dtype=np.float64)
    P = a.shape[0]
    packed = np.array([a[i, j] for i in range(P) for j in range(i + 1)])
    inv = np.zeros_like(packed)
    logabs = C.c_double()
    sign = C.c_int32()
    rc = lib().fabber_vb_ldl_inverse(P, packed.ctypes.data, inv.ctypes.data, C.byref(logabs), C.byref(sign))
    out = np.zeros((P, P))
    k = 0
    for i in range(P):
        for j in range(i + 1):
            out[i, j] = out[j, i] = inv[k]
            k += 1
    return out, logabs.value, sign.value, rc == 0


# ---- spatial VB --------------------------------------------------------------------------------
PROGRESS_CB = C.CFUNCTYPE(None, C.c_int, C.c_int)


def run_spatial_host(holder, spatial, data, device=0, progress_cb=None, devices=None):
    """Spatial VB on the GPU from host arrays. spatial: vbabi.SpatialHolder. devices: "all" or a list of device
    indices - z-slabs of the one volume on several devices, driven by this process
    (fabber_vb_run_spatial_host_multi)."""
    cfg = holder.cfg
    data = _prepare_data(holder, data)
    V = cfg.n_voxels
    arrs = dict(
        mvn=np.full((holder.n_mvn_rows, V), np.nan),
        free_energy=np.full(V, np.nan),
        status=np.full(V, -1, dtype=np.int32),
        iterations=np.full(V, -1, dtype=np.int32),
    )
    out = vbabi.FvbOutputs()
    for k, a in arrs.items():
        setattr(out, k, a.ctypes.data)
    L = lib()
    L.fabber_vb_run_spatial_host.restype = C.c_int32
    L.fabber_vb_run_spatial_host.argtypes = [C.POINTER(vbabi.FvbConfig), C.POINTER(vbabi.FvbSpatial), C.c_void_p,
                                             C.POINTER(vbabi.FvbOutputs), C.c_int32, C.c_void_p]
    cb = PROGRESS_CB(progress_cb) if progress_cb else None
    if devices is not None:
        L.fabber_vb_run_spatial_host_multi.restype = C.c_int32
        L.fabber_vb_run_spatial_host_multi.argtypes = [C.POINTER(vbabi.FvbConfig), C.POINTER(vbabi.FvbSpatial), C.c_void_p,
                                                       C.POINTER(vbabi.FvbOutputs), C.c_void_p, C.c_int32, C.c_void_p]
        ids = None if devices == "all" else (C.c_int32 * len(devices))(*devices)
        _check(L.fabber_vb_run_spatial_host_multi(C.byref(cfg), C.byref(spatial.sp), data.ctypes.data, C.byref(out), ids,
                                                  0 if ids is None else len(devices), C.cast(cb, C.c_void_p) if cb else None))
    else:
        _check(L.fabber_vb_run_spatial_host(C.byref(cfg), C.byref(spatial.sp), data.ctypes.data, C.byref(out), device,
                                            C.cast(cb, C.c_void_p) if cb else None))
    arrs["setup_failed"] = (arrs["status"] & 0x100) != 0
    arrs["status"] = arrs["status"] & 0xFF
    return arrs


class SpatialMultiRun:
    """One spatial VB problem resident on several devices (fabber_vb_spatial_multi_*): open uploads, run() is one
    complete run on the resident data (may be repeated), results() downloads, close() frees."""

    def __init__(self, holder, spatial, data, devices, want=("free_energy", "status", "iterations")):
        self.holder, self.spatial = holder, spatial
        cfg = holder.cfg
        self._data = _prepare_data(holder, data)
        V = cfg.n_voxels
        self.arrs = dict(mvn=np.full((holder.n_mvn_rows, V), np.nan))
        if "free_energy" in want:
            self.arrs["free_energy"] = np.full(V, np.nan)
        if "status" in want:
            self.arrs["status"] = np.full(V, -1, dtype=np.int32)
        if "iterations" in want:
            self.arrs["iterations"] = np.full(V, -1, dtype=np.int32)
        self.out = vbabi.FvbOutputs()
        for k, a in self.arrs.items():
            setattr(self.out, k, a.ctypes.data)
        L = self.L = lib()
        L.fabber_vb_spatial_multi_open.restype = C.c_int32
        L.fabber_vb_spatial_multi_open.argtypes = [C.POINTER(vbabi.FvbConfig), C.POINTER(vbabi.FvbSpatial), C.c_void_p,
                                                   C.POINTER(vbabi.FvbOutputs), C.c_void_p, C.c_int32, C.POINTER(C.c_void_p)]
        for name in ("fabber_vb_spatial_multi_run", "fabber_vb_spatial_multi_results"):
            getattr(L, name).restype = C.c_int32
            getattr(L, name).argtypes = [C.c_void_p, C.c_void_p]
        L.fabber_vb_spatial_multi_slabs.restype = C.c_int32
        L.fabber_vb_spatial_multi_slabs.argtypes = [C.c_void_p, C.POINTER(C.c_int32), C.c_char_p, C.c_int32]
        L.fabber_vb_spatial_multi_close.restype = C.c_int32
        L.fabber_vb_spatial_multi_close.argtypes = [C.c_void_p]
        ids = None if devices == "all" else (C.c_int32 * len(devices))(*devices)
        self.h = C.c_void_p()
        _check(L.fabber_vb_spatial_multi_open(C.byref(cfg), C.byref(spatial.sp), self._data.ctypes.data, C.byref(self.out), ids,
                                              0 if ids is None else len(devices), C.byref(self.h)))

    def run(self):
        _check(self.L.fabber_vb_spatial_multi_run(self.h, None))

    def slabs(self):
        n = C.c_int32(0)
        buf = C.create_string_buffer(64)
        _check(self.L.fabber_vb_spatial_multi_slabs(self.h, C.byref(n), buf, 64))
        return n.value, buf.value.decode()

    def results(self):
        _check(self.L.fabber_vb_spatial_multi_results(self.h, C.byref(self.out)))
        r = dict(self.arrs)
        if "status" in r:
            r["setup_failed"] = (r["status"] & 0x100) != 0
            r["status"] = r["status"] & 0xFF
        return r

    def close(self):
        if self.h:
            self.L.fabber_vb_spatial_multi_close(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def test_unlink_slab_pair(pair):
    """Fault injection for tests (fabber_vb_test_unlink_slab_pair): the next multi-device spatial run of this thread
    leaves the slab above `pair` without its inboxes' writer."""
    L = lib()
    L.fabber_vb_test_unlink_slab_pair.restype = None
    L.fabber_vb_test_unlink_slab_pair.argtypes = [C.c_int32]
    L.fabber_vb_test_unlink_slab_pair(pair)


def neighbours(coords, spatial_dims=3):
    """First-neighbour table of the spatial driver (host code, no GPU): [V][6], -1 = none."""
    coords = np.ascontiguousarray(coords, dtype=np.int32)
    V = coords.shape[1]
    nn = np.full((V, 6), -1, dtype=np.int32)
    L = lib()
    L.fabber_vb_neighbours.restype = C.c_int32
    L.fabber_vb_neighbours.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]
    _check(L.fabber_vb_neighbours(coords.ctypes.data, V, spatial_dims, nn.ctypes.data))
    return nn
