"""ctypes mirror of include/fabber_vb.h (the C ABI of the HIP voxelwise-VB engine).

Only plain data lives here: structure layouts, enum values and a helper that resolves the
built-in models' parameter defaults into an ``fvb_config`` the same way the reference's
``FwdModel::GetParameters`` (fwdmodel.cc:210-282) and ``WhiteNoiseModel::HardcodedInitialDists``
(noisemodel_white.cc:127-164) do. The C++ host library does the same resolution natively for
the fabber_capi path; this module is what bench.py and the tests use to talk to the kernel
library directly with device pointers.
"""
import ctypes as C
import math

import numpy as np

FVB_MAX_PARAMS = 32
FVB_MAX_PHIS = 8
FVB_MAX_ALPHAS = 4
FVB_ABI_VERSION = 9

MODEL_POLY, MODEL_LINEAR, MODEL_EXP, MODEL_HOSTJAC = 0, 1, 2, 100
TRANSFORM_IDENTITY, TRANSFORM_LOG, TRANSFORM_SOFTPLUS, TRANSFORM_FRACTIONAL, TRANSFORM_ABS = range(5)
TRANSFORM_CODES = {"I": 0, "L": 1, "S": 2, "F": 3, "A": 4}
PRIOR_NORMAL, PRIOR_IMAGE, PRIOR_ARD, PRIOR_SPATIAL_M, PRIOR_SPATIAL_m, PRIOR_SPATIAL_P, PRIOR_SPATIAL_p = range(7)
PRIOR_CODES = {"N": 0, "-": 0, "I": 1, "A": 2, "M": 3, "m": 4, "P": 5, "p": 6}
CONV_MAXITS, CONV_FCHANGE, CONV_FREDUCE, CONV_TRIALMODE, CONV_LM = range(5)
CONV_NAMES = {"maxits": 0, "pointzeroone": 1, "freduce": 2, "trialmode": 3, "lm": 4}
NOISE_WHITE, NOISE_AR1 = 0, 1
STATUS_OK, STATUS_BAD_OFFSET, STATUS_BAD_JACOBIAN, STATUS_BAD_FREE_ENERGY, STATUS_BAD_RESULT, STATUS_BAD_AR_ALPHA = range(6)

_dp = C.POINTER(C.c_double)


class FvbParamTable(C.Structure):
    """fvb_param_table: the per-parameter entries of a problem with more than FVB_MAX_PARAMS parameters"""
    _fields_ = [("transform", C.c_void_p), ("prior_type", C.c_void_p), ("prior_mean", C.c_void_p), ("prior_var", C.c_void_p),
                ("prior_prec", C.c_void_p), ("post_mean", C.c_void_p), ("post_var", C.c_void_p), ("image_prior", C.c_void_p)]


class FvbConfig(C.Structure):
    _fields_ = [
        ("abi_version", C.c_int32),
        ("n_voxels", C.c_int32),
        ("n_times", C.c_int32),
        ("n_params", C.c_int32),
        ("n_phis", C.c_int32),
        ("noise", C.c_int32),
        ("model", C.c_int32),
        ("model_iopt", C.c_int32 * 4),
        ("model_dopt", C.c_double * 4),
        ("design", C.c_void_p),
        ("transform", C.c_int32 * FVB_MAX_PARAMS),
        ("prior_type", C.c_int32 * FVB_MAX_PARAMS),
        ("prior_mean", C.c_double * FVB_MAX_PARAMS),
        ("prior_var", C.c_double * FVB_MAX_PARAMS),
        ("prior_prec", C.c_double * FVB_MAX_PARAMS),
        ("post_mean", C.c_double * FVB_MAX_PARAMS),
        ("post_var", C.c_double * FVB_MAX_PARAMS),
        ("image_prior", C.c_void_p * FVB_MAX_PARAMS),
        ("noise_prior_b", C.c_double * FVB_MAX_PHIS),
        ("noise_prior_c", C.c_double * FVB_MAX_PHIS),
        ("noise_post_b", C.c_double * FVB_MAX_PHIS),
        ("noise_post_c", C.c_double * FVB_MAX_PHIS),
        ("locked_noise_stdev", C.c_double),
        ("phi_index", C.c_void_p),
        ("convergence", C.c_int32),
        ("max_iterations", C.c_int32),
        ("max_trials", C.c_int32),
        ("need_f", C.c_int32),
        ("min_fchange", C.c_double),
        ("init_mvn", C.c_void_p),
        ("f_history_rows", C.c_int32),
        ("data_f64", C.c_int32),
        ("ar_cross_terms", C.c_int32),
        ("ar_alpha_given", C.c_int32),
        ("ar_alpha_prior_mean", C.c_double * FVB_MAX_ALPHAS),
        ("ar_alpha_prior_prec", (C.c_double * FVB_MAX_ALPHAS) * FVB_MAX_ALPHAS),
        ("ar_alpha_post_mean", C.c_double * FVB_MAX_ALPHAS),
        ("ar_alpha_post_cov", (C.c_double * FVB_MAX_ALPHAS) * FVB_MAX_ALPHAS),
        ("params_ext", C.c_void_p),
    ]


class FvbOutputs(C.Structure):
    _fields_ = [
        ("mvn", C.c_void_p),
        ("free_energy", C.c_void_p),
        ("f_history", C.c_void_p),
        ("f_history_len", C.c_void_p),
        ("status", C.c_void_p),
        ("iterations", C.c_void_p),
    ]


class FvbSummary(C.Structure):
    _fields_ = [("sum_free_energy", C.c_double), ("sum_iterations", C.c_int64), ("bad_voxels", C.c_int64)]


class FvbPostproc(C.Structure):
    _fields_ = [
        ("mean", C.c_void_p),
        ("var", C.c_void_p),
        ("std", C.c_void_p),
        ("zstat", C.c_void_p),
        ("modelfit", C.c_void_p),
        ("residuals", C.c_void_p),
        ("noise_mean", C.c_void_p),
        ("noise_std", C.c_void_p),
    ]


class FvbNlls(C.Structure):
    """fvb_nlls (include/fabber_vb.h): settings of the method=nlls minimiser."""
    _fields_ = [
        ("lm", C.c_int32),
        ("max_iterations", C.c_int32),
        ("cf_tolerance", C.c_double),
        ("lambda0", C.c_double),
        ("lambda_max", C.c_double),
    ]

    @classmethod
    def defaults(cls, lm=False):
        return cls(int(lm), 200, 1e-8, 0.1, 1e20)


def mvn_rows(n):
    """Rows of the packed MVN image for an n-dimensional MVN (dist_mvn.cc:408)."""
    return n * (n + 1) // 2 + n + 1


# ---- transforms.h:114-242, transforms.cc:17-25 (host-side, used only to resolve priors) ----
def to_model(tr, x):
    if tr == TRANSFORM_LOG:
        return math.exp(x)
    if tr == TRANSFORM_SOFTPLUS:
        return math.log(1 + math.exp(x)) if x < 10 else x
    if tr == TRANSFORM_FRACTIONAL:
        return 1 / (1 + math.exp(x))
    if tr == TRANSFORM_ABS:
        return abs(x)
    return x


def to_fabber(tr, x):
    if tr == TRANSFORM_LOG:
        return math.log(x)
    if tr == TRANSFORM_SOFTPLUS:
        return math.log(math.exp(x) - 1) if x < 10 else x
    if tr == TRANSFORM_FRACTIONAL:
        return math.log(1 / x - 1)
    return x


def to_fabber_var(tr, v):
    if tr in (TRANSFORM_IDENTITY, TRANSFORM_FRACTIONAL):
        return v
    if tr == TRANSFORM_LOG:
        return math.log(v)
    return to_fabber(tr, to_model(tr, 0) + math.sqrt(v)) ** 2


def model_parameter_defaults(model, **opts):
    """GetParameterDefaults of the built-in models: list of dicts with name, prior (mean, var),
    post (mean, var), prior_type, transform. fwdmodel_poly.cc:53-60, fwdmodel_linear.cc:83-90,
    examples/fwdmodel_exp.cc:49-63."""
    params = []
    if model == MODEL_POLY:
        for i in range(int(opts["degree"]) + 1):
            params.append(dict(name="c%d" % i, prior=(0.0, 1e12), post=(0.0, 1e12), prior_type="N", transform=TRANSFORM_IDENTITY))
    elif model == MODEL_LINEAR:
        for i in range(int(opts["n_basis"])):
            params.append(dict(name="Parameter_%d" % (i + 1), prior=(0.0, 1e12), post=(0.0, 1e12), prior_type="N", transform=TRANSFORM_IDENTITY))
    elif model == MODEL_EXP:
        for i in range(int(opts.get("num_exps", 1))):
            for nm in ("amp", "r"):
                params.append(dict(name="%s%d" % (nm, i + 1), prior=(1.0, 1e5), post=(1.0, 1.5), prior_type="N", transform=TRANSFORM_LOG))
    else:
        raise ValueError("unknown model %r" % (model,))
    return params


class ConfigHolder:
    """An FvbConfig plus the numpy arrays its pointer members refer to (kept alive here)."""

    def __init__(self, cfg, params, keep):
        self.cfg = cfg
        self.params = params
        self.keep = keep

    def set_post_mean(self, values):
        """The initial posterior means (method=nlls: the Fabber-space starting estimate), wherever the per-parameter
        entries live - the fixed arrays or the table of cfg.params_ext."""
        dst = self.keep["param_table"][0]["post_mean"] if self.cfg.params_ext else self.cfg.post_mean
        for k in range(self.cfg.n_params):
            dst[k] = float(values[k])

    @property
    def n_mvn_rows(self):
        return mvn_rows(self.cfg.n_params + self.n_noise_outputs)

    @property
    def n_noise_outputs(self):
        return self.cfg.n_phis if self.cfg.noise == NOISE_WHITE else 2 + self.cfg.ar_cross_terms + self.cfg.n_phis


def build_config(model, n_voxels, n_times, *, degree=None, design=None, num_exps=1, dt=1.0,
                 convergence="maxits", max_iterations=10, min_fchange=0.01, max_trials=10,
                 need_f=None, f_history_rows=0, noise_pattern="1", masked_timepoints=(),
                 prior_noise_stddev=-1.0, locked_noise_stdev=-1.0, param_overrides=None,
                 image_priors=None, init_mvn=None, noise=NOISE_WHITE, num_echoes=1, ar_cross_terms="none",
                 ar_alpha_prior=None, ar_alpha_post=None):
    """Resolve options into an fvb_config whose pointer members are HOST numpy arrays.

    param_overrides: {name: dict(type=, mean=, prec=, transform=)} == PSP_byname options
    (fwdmodel.cc:238-266). image_priors: {name: float64 array [n_voxels]}.
    """
    keep = {}
    cfg = FvbConfig()
    cfg.abi_version = FVB_ABI_VERSION
    cfg.n_voxels = n_voxels
    cfg.n_times = n_times
    cfg.noise = noise
    cfg.model = model
    if model == MODEL_POLY:
        cfg.model_iopt[0] = int(degree)
        params = model_parameter_defaults(model, degree=degree)
    elif model == MODEL_LINEAR:
        design = np.ascontiguousarray(design, dtype=np.float64)
        assert design.shape[0] == n_times
        keep["design"] = design
        cfg.design = design.ctypes.data
        params = model_parameter_defaults(model, n_basis=design.shape[1])
    elif model == MODEL_EXP:
        cfg.model_iopt[0] = int(num_exps)
        cfg.model_dopt[0] = float(dt)
        params = model_parameter_defaults(model, num_exps=num_exps)
    else:
        raise ValueError(model)
    P = len(params)
    cfg.n_params = P
    param_overrides = param_overrides or {}
    image_priors = image_priors or {}
    # more than FVB_MAX_PARAMS parameters: the per-parameter entries go into an fvb_param_table (cfg.params_ext)
    wide = P > FVB_MAX_PARAMS
    tab = None
    if wide:
        tab = dict(transform=np.zeros(P, dtype=np.int32), prior_type=np.zeros(P, dtype=np.int32), prior_mean=np.zeros(P),
                   prior_var=np.zeros(P), prior_prec=np.zeros(P), post_mean=np.zeros(P), post_var=np.zeros(P),
                   image_prior=np.zeros(P, dtype=np.uint64))
    dst = tab if wide else {f: getattr(cfg, f) for f in ("transform", "prior_type", "prior_mean", "prior_var", "prior_prec", "post_mean",
                                                         "post_var", "image_prior")}
    for k, p in enumerate(params):
        ov = param_overrides.get(p["name"], {})
        tr = p["transform"]
        if "transform" in ov:
            tr = TRANSFORM_CODES[ov["transform"]] if isinstance(ov["transform"], str) else ov["transform"]
        ptype = ov.get("type", p["prior_type"])
        mean, var = p["prior"]
        if "mean" in ov or "prec" in ov:
            # fwdmodel.cc:258-263: DistParams(mean, 1/prec)
            mean = float(ov.get("mean", mean))
            prec = float(ov.get("prec", 1.0 / var))
            var = 1.0 / prec
        if 1.0 / var > 1e12:  # fwdmodel.cc:268-271
            var = 1e-12
        # Prior to Fabber space (fwdmodel.cc:277, transforms.cc:10-15)
        fmean = to_fabber(tr, mean)
        fvar = to_fabber_var(tr, var)
        dst["transform"][k] = tr
        dst["prior_type"][k] = PRIOR_CODES[ptype]
        dst["prior_mean"][k] = fmean
        dst["prior_var"][k] = fvar
        dst["prior_prec"][k] = 1.0 / fvar if fvar != 0 else math.inf
        dst["post_mean"][k] = p["post"][0]
        dst["post_var"][k] = p["post"][1]
        p.update(transform=tr, prior_type=ptype)
        if PRIOR_CODES[ptype] == PRIOR_IMAGE:
            img = np.ascontiguousarray(image_priors[p["name"]], dtype=np.float64)
            assert img.shape == (n_voxels,)
            keep["image_%d" % k] = img
            dst["image_prior"][k] = img.ctypes.data
    if wide:
        ext = FvbParamTable()
        for f, a in tab.items():
            setattr(ext, f, a.ctypes.data)
        keep["param_table"] = (tab, ext)
        cfg.params_ext = C.addressof(ext)

    # noise pattern -> phi index per timepoint (noisemodel_white.cc:166-226)
    pat = []
    for ch in noise_pattern:
        if "1" <= ch <= "9":
            pat.append(ord(ch) - ord("0"))
        elif "A" <= ch <= "Z":
            pat.append(ord(ch) - ord("A") + 10)
        elif "a" <= ch <= "z":
            pat.append(ord(ch) - ord("a") + 10)
        else:
            raise ValueError("noise-pattern: invalid character %r" % ch)
    if len(pat) > n_times:
        raise ValueError("noise-pattern: Pattern length exceeds data length")
    n_phis = max(pat)
    assert n_phis <= FVB_MAX_PHIS
    cfg.n_phis = n_phis if noise == NOISE_WHITE else 1
    phi_index = np.zeros(n_times, dtype=np.uint8)
    for t in range(n_times):
        phi_index[t] = pat[t % len(pat)] - 1
    for mt in masked_timepoints:  # 1-based (noisemodel.cc:34-40)
        phi_index[mt - 1] = 255
    keep["phi_index"] = phi_index
    cfg.phi_index = phi_index.ctypes.data
    for i in range(FVB_MAX_PHIS):
        if prior_noise_stddev == -1.0:  # noisemodel_white.cc:141-150
            cfg.noise_prior_b[i], cfg.noise_prior_c[i] = 1e6, 1e-6
            cfg.noise_post_b[i], cfg.noise_post_c[i] = 1e-8, 50.0
        else:  # :151-162
            c = 0.5
            b = 1 / (prior_noise_stddev * prior_noise_stddev * c)
            cfg.noise_prior_b[i] = cfg.noise_post_b[i] = b
            cfg.noise_prior_c[i] = cfg.noise_post_c[i] = c
    if noise == NOISE_AR1:  # noisemodel_ar.cc:322-403
        cfg.n_phis = num_echoes
        cfg.ar_cross_terms = {"none": 0, "same": 1, "dual": 2}[ar_cross_terms]
        # noise-initial-prior / -posterior for the AR(1) coefficients: (mean, precision matrix) and (mean, covariance)
        nA = 2 + cfg.ar_cross_terms
        for bit, given, mean_f, mat_f in ((1, ar_alpha_prior, cfg.ar_alpha_prior_mean, cfg.ar_alpha_prior_prec),
                                          (2, ar_alpha_post, cfg.ar_alpha_post_mean, cfg.ar_alpha_post_cov)):
            if given is None:
                continue
            mean, mat = np.asarray(given[0], dtype=np.float64), np.asarray(given[1], dtype=np.float64)
            assert mean.shape == (nA,) and mat.shape == (nA, nA) and np.array_equal(mat, mat.T)
            cfg.ar_alpha_given |= bit
            for i in range(nA):
                mean_f[i] = mean[i]
                for j in range(nA):
                    mat_f[i][j] = mat[i, j]
        for i in range(num_echoes):
            cfg.noise_prior_b[i], cfg.noise_prior_c[i] = 1e6, 1e-6
            cfg.noise_post_b[i], cfg.noise_post_c[i] = 1e-8, 1e-6
    cfg.locked_noise_stdev = locked_noise_stdev

    cfg.convergence = CONV_NAMES[convergence] if isinstance(convergence, str) else convergence
    cfg.max_iterations = max_iterations
    cfg.max_trials = max_trials
    cfg.min_fchange = min_fchange
    uses_f = cfg.convergence != CONV_MAXITS
    cfg.need_f = int(uses_f if need_f is None else (need_f or uses_f))
    cfg.f_history_rows = f_history_rows
    if init_mvn is not None:
        init_mvn = np.ascontiguousarray(init_mvn, dtype=np.float64)
        keep["init_mvn"] = init_mvn
        cfg.init_mvn = init_mvn.ctypes.data
    return ConfigHolder(cfg, params, keep)


# ---- spatial VB (include/fabber_vb.h: fvb_spatial) --------------------------------------------
class FvbSpatial(C.Structure):
    _fields_ = [
        ("coords", C.c_void_p),
        ("spatial_dims", C.c_int32),
        ("update_first_iter", C.c_int32),
        ("spatial_speed", C.c_double),
        ("q1", C.c_double),
        ("q2", C.c_double),
        ("owned_begin", C.c_int32),
        ("owned_end", C.c_int32),
        ("n_voxels_global", C.c_int32),
        ("locked_centres", C.c_void_p),
    ]


def grid_coords(shape, mask=None):
    """[3][V] int32 grid coordinates of the masked voxels, x fastest then y then z
    (rundata_array.cc:42-63)."""
    nx, ny, nz = shape
    z, y, x = np.meshgrid(np.arange(nz), np.arange(ny), np.arange(nx), indexing="ij")
    keep = np.ones(nx * ny * nz, dtype=bool) if mask is None else (np.asarray(mask).transpose(2, 1, 0).ravel() != 0)
    return np.ascontiguousarray(np.stack([x.ravel()[keep], y.ravel()[keep], z.ravel()[keep]]).astype(np.int32))


class SpatialHolder:
    def __init__(self, coords, spatial_dims=3, spatial_speed=-1.0, q1=10.0, q2=1.0, update_first_iter=False,
                 owned=None, n_voxels_global=0, locked_centres=None):
        self.coords = np.ascontiguousarray(coords, dtype=np.int32)
        assert self.coords.ndim == 2 and self.coords.shape[0] == 3
        self.sp = FvbSpatial()
        self.sp.coords = self.coords.ctypes.data
        self.sp.spatial_dims = spatial_dims
        self.sp.update_first_iter = int(update_first_iter)
        self.sp.spatial_speed = spatial_speed
        self.sp.q1 = q1
        self.sp.q2 = q2
        self.sp.owned_begin, self.sp.owned_end = owned if owned is not None else (0, 0)
        self.sp.n_voxels_global = n_voxels_global
        # [P][V] fixed linearisation centres (locked-linear-from-mvn), kept alive with the holder
        self.locked_centres = None if locked_centres is None else np.ascontiguousarray(locked_centres, dtype=np.float64)
        self.sp.locked_centres = None if self.locked_centres is None else self.locked_centres.ctypes.data
