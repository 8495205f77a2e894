"""Loader for the CPU oracle (oracle/liboracle.so). Test infrastructure only.

Builds the library on first use with oracle/Makefile (g++ only, a few seconds).
"""
import ctypes as C
import os
import subprocess

import numpy as np

from fabber_core_amd import vbabi

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_LIB = None
_LIB_FMA = None
_LIB_QUAD = None


class OracleTrace(C.Structure):
    _fields_ = [("means", C.c_void_p), ("noise_b", C.c_void_p), ("max_rows", C.c_int32)]


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_ROOT, "oracle", "liboracle.so")
        src = os.path.join(_ROOT, "oracle", "vb_oracle.cc")
        subprocess.check_call(["make", "-s", "-C", os.path.join(_ROOT, "oracle"), "liboracle.so"])  # (make knows the dependencies: the source, the .inc files, include/fabber_vb.h)
        L = C.CDLL(path)
        L.oracle_vb_run.restype = C.c_int32
        L.oracle_vb_run.argtypes = [C.POINTER(vbabi.FvbConfig), C.c_void_p, C.POINTER(vbabi.FvbOutputs),
                                    C.c_int32, C.c_int32, C.c_int32, C.c_void_p]
        L.oracle_vb_postproc.restype = C.c_int32
        L.oracle_vb_postproc.argtypes = [C.POINTER(vbabi.FvbConfig), C.c_void_p, C.c_void_p, C.POINTER(vbabi.FvbPostproc)]
        for name in ("oracle_gammaln", "oracle_digamma"):
            getattr(L, name).restype = C.c_double
            getattr(L, name).argtypes = [C.c_double]
        for name in ("oracle_transform_to_model", "oracle_transform_to_fabber", "oracle_transform_to_model_var",
                     "oracle_transform_to_fabber_var"):
            getattr(L, name).restype = C.c_double
            getattr(L, name).argtypes = [C.c_int32, C.c_double]
        L.oracle_evaluate_fabber.restype = C.c_int32
        L.oracle_evaluate_fabber.argtypes = [C.POINTER(vbabi.FvbConfig), C.c_void_p, C.c_void_p]
        L.oracle_convergence_trace.restype = C.c_int32
        L.oracle_convergence_trace.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.c_double, C.c_void_p, C.c_int32,
                                               C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32]
        L.oracle_inverse.restype = C.c_int32
        L.oracle_inverse.argtypes = [C.c_int32, C.c_void_p, C.c_void_p]
        L.oracle_logdet.restype = C.c_double
        L.oracle_logdet.argtypes = [C.c_int32, C.c_void_p, C.c_void_p]
        _LIB = L
    return _LIB


def lib_fma():
    """Second CPU build of the same source with FMA contraction (oracle/Makefile)."""
    global _LIB_FMA
    if _LIB_FMA is None:
        path = os.path.join(_ROOT, "oracle", "liboracle_fma.so")
        src = os.path.join(_ROOT, "oracle", "vb_oracle.cc")
        subprocess.check_call(["make", "-s", "-C", os.path.join(_ROOT, "oracle"), "liboracle_fma.so"])  # (make knows the dependencies: the source, the .inc files, include/fabber_vb.h)
        L = C.CDLL(path)
        L.oracle_vb_run.restype = C.c_int32
        L.oracle_vb_run.argtypes = [C.POINTER(vbabi.FvbConfig), C.c_void_p, C.POINTER(vbabi.FvbOutputs),
                                    C.c_int32, C.c_int32, C.c_int32, C.c_void_p]
        _LIB_FMA = L
    return _LIB_FMA


_LIB_EXP1ULP = None


def lib_exp1ulp():
    """CPU build whose exp has the device's accuracy class (1 ulp; oracle/Makefile, -DORACLE_EXP_1ULP)."""
    global _LIB_EXP1ULP
    if _LIB_EXP1ULP is None:
        path = os.path.join(_ROOT, "oracle", "liboracle_exp1ulp.so")
        src = os.path.join(_ROOT, "oracle", "vb_oracle.cc")
        subprocess.check_call(["make", "-s", "-C", os.path.join(_ROOT, "oracle"), "liboracle_exp1ulp.so"])  # (make knows the dependencies: the source, the .inc files, include/fabber_vb.h)
        L = C.CDLL(path)
        L.oracle_vb_run.restype = C.c_int32
        L.oracle_vb_run.argtypes = [C.POINTER(vbabi.FvbConfig), C.c_void_p, C.POINTER(vbabi.FvbOutputs),
                                    C.c_int32, C.c_int32, C.c_int32, C.c_void_p]
        _LIB_EXP1ULP = L
    return _LIB_EXP1ULP


def lib_quad():
    """The white-noise voxelwise loop evaluated in IEEE binary128 (oracle/Makefile, -DORACLE_QUAD):
    the ground truth for the rounding error of any fp64 build of the algorithm."""
    global _LIB_QUAD
    if _LIB_QUAD is None:
        path = os.path.join(_ROOT, "oracle", "liboracle_quad.so")
        src = os.path.join(_ROOT, "oracle", "vb_oracle.cc")
        subprocess.check_call(["make", "-s", "-C", os.path.join(_ROOT, "oracle"), "liboracle_quad.so"])  # (make knows the dependencies: the source, the .inc files, include/fabber_vb.h)
        L = C.CDLL(path)
        L.oracle_vb_run.restype = C.c_int32
        L.oracle_vb_run.argtypes = [C.POINTER(vbabi.FvbConfig), C.c_void_p, C.POINTER(vbabi.FvbOutputs),
                                    C.c_int32, C.c_int32, C.c_int32, C.c_void_p]
        _LIB_QUAD = L
    return _LIB_QUAD


def alloc_outputs(holder):
    """Host result arrays for a config, plus the FvbOutputs struct pointing at them."""
    cfg = holder.cfg
    V = cfg.n_voxels
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
    return arrs, out


def prepare_data(holder, data):
    """float32 data is the C-ABI route, float64 the in-memory NEWMAT::Matrix route."""
    cfg = holder.cfg
    data = np.ascontiguousarray(data)
    if data.dtype == np.float64:
        cfg.data_f64 = 1
    else:
        data = np.ascontiguousarray(data, dtype=np.float32)
        cfg.data_f64 = 0
    assert data.shape == (cfg.n_times, cfg.n_voxels), (data.shape, cfg.n_times, cfg.n_voxels)
    return data


def run_fma(holder, data, **kw):
    """The oracle compiled with FMA contraction: same algorithm, different rounding."""
    return run(holder, data, _lib=lib_fma(), **kw)


def set_inverse(mode, _lib=None):
    """'lu' (default: LAPACK-style LU with partial pivoting) or 'sweep' (the kernels' unpivoted symmetric
    sweep) as the restatement of NEWMAT's .i(); see inverse() in oracle/vb_oracle.cc."""
    (_lib or lib()).oracle_set_inverse({"lu": 0, "sweep": 1}[mode])


def run_quad(holder, data, **kw):
    """Ground truth: the same statements in binary128 (white noise, voxelwise)."""
    return run(holder, data, _lib=lib_quad(), **kw)


def run(holder, data, v_begin=0, v_end=None, halt_bad_voxel=False, trace_rows=0, _lib=None):
    """Run the oracle. data: float32 [n_times][n_voxels]. Returns dict of result arrays."""
    cfg = holder.cfg
    data = prepare_data(holder, data)
    arrs, out = alloc_outputs(holder)
    tr = None
    trp = None
    if trace_rows:
        arrs["trace_means"] = np.full((trace_rows, cfg.n_params, cfg.n_voxels), np.nan)
        arrs["trace_noise_b"] = np.full((trace_rows, cfg.n_phis, cfg.n_voxels), np.nan)
        tr = OracleTrace(arrs["trace_means"].ctypes.data, arrs["trace_noise_b"].ctypes.data, trace_rows)
        trp = C.addressof(tr)
    if v_end is None:
        v_end = cfg.n_voxels
    rc = (_lib or lib()).oracle_vb_run(C.byref(cfg), data.ctypes.data, C.byref(out), v_begin, v_end, int(halt_bad_voxel), trp)
    if rc < 0:
        raise RuntimeError("oracle_vb_run failed: %d" % rc)
    arrs["first_bad_voxel"] = rc
    return arrs


def run_nlls(holder, data, lm=False, start=None, settings=None, halt_bad_voxel=False, _lib=None):
    """method=nlls on the CPU (oracle/vb_oracle_nlls.inc); same result dict as hiplib.nlls_run_host."""
    cfg = holder.cfg
    data = prepare_data(holder, data)
    V, P = cfg.n_voxels, cfg.n_params
    holder.set_post_mean([0.0] * P if start is None else start)
    nl = settings or vbabi.FvbNlls.defaults(lm)
    arrs = dict(mvn=np.full((vbabi.mvn_rows(P), V), np.nan), status=np.full(V, -1, dtype=np.int32),
                iterations=np.full(V, -1, dtype=np.int32), free_energy=np.full(V, np.nan))
    out = vbabi.FvbOutputs()
    for k, a in arrs.items():
        setattr(out, k, a.ctypes.data)
    L = _lib or lib()
    L.oracle_nlls_run.restype = C.c_int32
    L.oracle_nlls_run.argtypes = [C.POINTER(vbabi.FvbConfig), C.POINTER(vbabi.FvbNlls), C.c_void_p,
                                  C.POINTER(vbabi.FvbOutputs), C.c_int32, C.c_int32, C.c_int32]
    rc = L.oracle_nlls_run(C.byref(cfg), C.byref(nl), data.ctypes.data, C.byref(out), 0, V, int(halt_bad_voxel))
    if rc < 0:
        raise RuntimeError("oracle_nlls_run failed: %d" % rc)
    arrs["first_bad_voxel"] = rc
    arrs["cost"] = arrs.pop("free_energy")
    return arrs


def postproc(holder, data, mvn, want=("mean", "var", "std", "zstat", "modelfit", "residuals", "noise_mean", "noise_std")):
    cfg = holder.cfg
    V, T, P, N = cfg.n_voxels, cfg.n_times, cfg.n_params, cfg.n_phis
    shapes = dict(mean=(P, V), var=(P, V), std=(P, V), zstat=(P, V), modelfit=(T, V), residuals=(T, V),
                  noise_mean=(N, V), noise_std=(N, V))
    data = prepare_data(holder, data)
    mvn = np.ascontiguousarray(mvn, dtype=np.float64)
    pp = vbabi.FvbPostproc()
    arrs = {}
    for k in want:
        arrs[k] = np.full(shapes[k], np.nan)
        setattr(pp, k, arrs[k].ctypes.data)
    rc = lib().oracle_vb_postproc(C.byref(cfg), data.ctypes.data, mvn.ctypes.data, C.byref(pp))
    assert rc == 0
    return arrs


def unpack_mvn(mvn, n):
    """Packed MVN rows -> (cov [V][n][n], means [V][n])."""
    V = mvn.shape[1]
    cov = np.zeros((V, n, n))
    k = 0
    for r in range(n):
        for c in range(r + 1):
            cov[:, r, c] = cov[:, c, r] = mvn[k]
            k += 1
    means = mvn[k:k + n].T.copy()
    return cov, means


def convergence_trace(conv, F, max_iterations=10, max_trials=10, min_fchange=0.01, stop_at_done=True):
    F = np.ascontiguousarray(F, dtype=np.float64)
    n = len(F)
    done = np.zeros(n, dtype=np.int32)
    save = np.zeros(n, dtype=np.int32)
    revert = np.zeros(n, dtype=np.int32)
    alpha = np.zeros(n)
    conv = vbabi.CONV_NAMES[conv] if isinstance(conv, str) else conv
    m = lib().oracle_convergence_trace(conv, max_iterations, max_trials, min_fchange, F.ctypes.data, n,
                                       done.ctypes.data, save.ctypes.data, revert.ctypes.data, alpha.ctypes.data,
                                       int(stop_at_done))
    return done[:m].astype(bool), save[:m].astype(bool), revert[:m].astype(bool), alpha[:m]


def run_spatial(holder, spatial, data, _lib=None):
    """Spatial VB oracle. spatial: vbabi.SpatialHolder."""
    cfg = holder.cfg
    data = prepare_data(holder, data)
    arrs, out = alloc_outputs(holder)
    L = _lib or lib()
    L.oracle_vb_run_spatial.restype = C.c_int32
    L.oracle_vb_run_spatial.argtypes = [C.POINTER(vbabi.FvbConfig), C.POINTER(vbabi.FvbSpatial), C.c_void_p,
                                        C.POINTER(vbabi.FvbOutputs)]
    L.oracle_last_error.restype = C.c_char_p
    rc = L.oracle_vb_run_spatial(C.byref(cfg), C.byref(spatial.sp), data.ctypes.data, C.byref(out))
    if rc != 0:
        raise RuntimeError("oracle_vb_run_spatial failed: %d %s" % (rc, L.oracle_last_error().decode()))
    return arrs


def run_spatial_fma(holder, spatial, data):
    return run_spatial(holder, spatial, data, _lib=lib_fma())


def run_spatial_exp1ulp(holder, spatial, data):
    return run_spatial(holder, spatial, data, _lib=lib_exp1ulp())


def run_exp1ulp(holder, data, **kw):
    return run(holder, data, _lib=lib_exp1ulp(), **kw)


def run_spatial_quad(holder, spatial, data):
    """The spatial loop (white noise) with every internal variable in IEEE binary128"""
    return run_spatial(holder, spatial, data, _lib=lib_quad())


def calc_neighbours(coords, spatial_dims=3):
    """Reference neighbour lists: (nn [V][6], nn2 [V][30], n2count [V]) with 1-based ids."""
    coords = np.ascontiguousarray(coords, dtype=np.int32)
    V = coords.shape[1]
    nn = np.zeros((V, 6), dtype=np.int32)
    nn2 = np.zeros((V, 30), dtype=np.int32)
    n2c = np.zeros(V, dtype=np.int32)
    L = lib()
    L.oracle_calc_neighbours.restype = C.c_int32
    L.oracle_calc_neighbours.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]
    L.oracle_last_error.restype = C.c_char_p
    rc = L.oracle_calc_neighbours(coords.ctypes.data, V, spatial_dims, nn.ctypes.data, nn2.ctypes.data, n2c.ctypes.data)
    if rc != 0:
        raise RuntimeError(L.oracle_last_error().decode())
    return nn, nn2, n2c


def spatial_prior_apply(coords, typ, means, mean0, prec0, aK, spatial_dims=3):
    """SpatialPrior::ApplyToMVN (priors.cc:346-488) for one parameter on a grid: (prior mean [V], prior precision [V])
    every voxel gets from the posterior means `means` and the smoothing precision aK. typ: "M", "m", "P", "p"."""
    coords = np.ascontiguousarray(coords, dtype=np.int32)
    V = coords.shape[1]
    means = np.ascontiguousarray(means, dtype=np.float64)
    assert means.shape == (V,)
    pm, pp = np.zeros(V), np.zeros(V)
    L = lib()
    L.oracle_spatial_prior_apply.restype = C.c_int32
    L.oracle_spatial_prior_apply.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_double, C.c_double, C.c_double,
                                             C.c_void_p, C.c_void_p, C.c_void_p]
    L.oracle_last_error.restype = C.c_char_p
    rc = L.oracle_spatial_prior_apply(coords.ctypes.data, V, spatial_dims, vbabi.PRIOR_CODES[typ], mean0, prec0, aK,
                                      means.ctypes.data, pm.ctypes.data, pp.ctypes.data)
    if rc != 0:
        raise RuntimeError(L.oracle_last_error().decode())
    return pm, pp
