"""Python 3 client of the fabber C ABI (include/fabber_capi.h), for
fabber_core_amd/lib/libfabbercore_amd.so.

Mirrors the reference's ctypes wrapper (py/fabber.py:489-771, which is Python 2): create a
context, set extent/options/data, run, read outputs back as numpy arrays. Any library that
exports the reference's 15 fabber_* symbols can be driven with it (pass its path), so the same
script runs against the reference's libfabbercore_shared and against this library.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
DEFAULT_LIB = os.path.join(_HERE, "lib", "libfabbercore_amd.so")

ERR_MAXC = 255
ERR_FATAL = -255
ERR_NEWMAT = -254

CAPI_SYMBOLS = [
    "fabber_new", "fabber_load_models", "fabber_set_extent", "fabber_destroy", "fabber_set_opt", "fabber_set_data",
    "fabber_get_data_size", "fabber_get_data", "fabber_dorun", "fabber_get_options", "fabber_get_models",
    "fabber_get_methods", "fabber_get_model_params", "fabber_get_model_param_descs", "fabber_get_model_outputs",
    "fabber_model_evaluate", "fabber_model_evaluate_output",
]

PROGRESS_CB = C.CFUNCTYPE(None, C.c_int, C.c_int)


class FabberError(RuntimeError):
    def __init__(self, code, message, log=""):
        RuntimeError.__init__(self, "fabber error %d: %s" % (code, message))
        self.code = code
        self.message = message
        self.log = log


def load_library(path=None):
    path = path or DEFAULT_LIB
    if not os.path.exists(path):
        raise RuntimeError("%s not found: run `python -m fabber_core_amd.build`" % path)
    from . import single_hip_runtime
    single_hip_runtime()
    L = C.CDLL(path)
    cp, vp = C.c_char_p, C.c_void_p
    L.fabber_new.restype = vp
    L.fabber_new.argtypes = [cp]
    L.fabber_load_models.argtypes = [vp, cp, cp]
    L.fabber_set_extent.argtypes = [vp, C.c_uint, C.c_uint, C.c_uint, vp, cp]
    L.fabber_destroy.restype = None
    L.fabber_destroy.argtypes = [vp]
    L.fabber_set_opt.argtypes = [vp, cp, cp, cp]
    L.fabber_set_data.argtypes = [vp, cp, C.c_uint, vp, cp]
    L.fabber_get_data_size.argtypes = [vp, cp, cp]
    L.fabber_get_data.argtypes = [vp, cp, vp, cp]
    L.fabber_dorun.argtypes = [vp, C.c_uint, cp, cp, vp]
    L.fabber_get_options.argtypes = [vp, cp, cp, C.c_uint, cp, cp]
    for name in ("fabber_get_models", "fabber_get_methods", "fabber_get_model_params", "fabber_get_model_param_descs",
                 "fabber_get_model_outputs"):
        getattr(L, name).argtypes = [vp, C.c_uint, cp, cp]
    L.fabber_model_evaluate.argtypes = [vp, C.c_uint, vp, C.c_uint, vp, vp, cp]
    L.fabber_model_evaluate_output.argtypes = [vp, C.c_uint, vp, C.c_uint, vp, cp, vp, cp]
    for name in CAPI_SYMBOLS:
        if name not in ("fabber_new", "fabber_destroy"):
            getattr(L, name).restype = C.c_int
    return L


class Fabber:
    """One fabber context. Volumes are numpy arrays indexed [x, y, z] or [x, y, z, t]."""

    def __init__(self, lib_path=None, model_libs=()):
        self.lib = load_library(lib_path)
        self.err = C.create_string_buffer(ERR_MAXC)
        self.handle = self.lib.fabber_new(self.err)
        if not self.handle:
            raise FabberError(ERR_FATAL, self.err.value.decode())
        self.shape = None
        self.log = ""
        for ml in model_libs:
            self._check(self.lib.fabber_load_models(self.handle, ml.encode(), self.err))

    def close(self):
        if self.handle:
            self.lib.fabber_destroy(self.handle)
            self.handle = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def _check(self, rc):
        if rc < 0:
            raise FabberError(rc, self.err.value.decode(errors="replace"), self.log)
        return rc

    # ---- configuration ----
    def set_extent(self, shape, mask=None):
        nx, ny, nz = (int(s) for s in shape)
        if mask is None:
            mask = np.ones((nx, ny, nz), dtype=np.int32)
        mask = np.asarray(mask)
        assert mask.shape == (nx, ny, nz)
        flat = np.ascontiguousarray(mask.astype(np.int32).transpose(2, 1, 0)).ravel()  # x fastest
        self._check(self.lib.fabber_set_extent(self.handle, nx, ny, nz, flat.ctypes.data, self.err))
        self.shape = (nx, ny, nz)
        self.mask = mask != 0

    def set_options(self, options):
        for key, value in options.items():
            if value is True:
                value = ""
            elif value is False or value is None:
                continue
            self._check(self.lib.fabber_set_opt(self.handle, str(key).encode(), str(value).encode(), self.err))

    def set_data(self, name, volume):
        volume = np.asarray(volume, dtype=np.float32)
        if volume.ndim == 3:
            volume = volume[..., None]
        assert volume.shape[:3] == self.shape, (volume.shape, self.shape)
        nt = volume.shape[3]
        flat = np.ascontiguousarray(volume.transpose(3, 2, 1, 0)).ravel()  # [t][z][y][x]
        self._check(self.lib.fabber_set_data(self.handle, name.encode(), nt, flat.ctypes.data, self.err))

    # ---- run / outputs ----
    def run(self, progress_cb=None, log_size=1 << 20):
        log = C.create_string_buffer(log_size)
        cb = PROGRESS_CB(progress_cb) if progress_cb else None
        rc = self.lib.fabber_dorun(self.handle, log_size, log, self.err, C.cast(cb, C.c_void_p) if cb else None)
        self.log = log.value.decode(errors="replace")
        self._check(rc)
        return self.log

    def data_size(self, name):
        return self.lib.fabber_get_data_size(self.handle, name.encode(), self.err)

    def get_data(self, name):
        n = self.data_size(name)
        if n < 0:
            raise FabberError(n, self.err.value.decode(errors="replace"))
        nx, ny, nz = self.shape
        buf = np.zeros(n * nz * ny * nx, dtype=np.float32)
        self._check(self.lib.fabber_get_data(self.handle, name.encode(), buf.ctypes.data, self.err))
        vol = buf.reshape(n, nz, ny, nx).transpose(3, 2, 1, 0)
        return vol[..., 0] if n == 1 else vol

    # ---- introspection ----
    def _text(self, fn, *args, size=1 << 16):
        out = C.create_string_buffer(size)
        self._check(fn(self.handle, *args, size, out, self.err))
        return out.value.decode()

    def get_models(self):
        return self._text(self.lib.fabber_get_models).split()

    def get_methods(self):
        return self._text(self.lib.fabber_get_methods).split()

    def get_model_params(self):
        return self._text(self.lib.fabber_get_model_params).split()

    def get_model_outputs(self):
        return self._text(self.lib.fabber_get_model_outputs).split()

    def get_options(self, key=None, value=None):
        """-> (description, [dict(name, description, type, optional, default)])"""
        text = self._text(self.lib.fabber_get_options, key.encode() if key else None, value.encode() if value else None)
        lines = text.split("\n")
        opts = []
        for line in lines[1:]:
            f = line.split("\t")
            if len(f) >= 5:
                opts.append(dict(name=f[0], description=f[1], type=f[2], optional=f[3] == "1", default=f[4]))
        return lines[0], opts

    def model_evaluate(self, params, nt, indata=None, output_name=""):
        p = np.ascontiguousarray(params, dtype=np.float32)
        out = np.zeros(nt, dtype=np.float32)
        ind = None if indata is None else np.ascontiguousarray(indata, dtype=np.float32)
        self._check(self.lib.fabber_model_evaluate_output(
            self.handle, len(p), p.ctypes.data, nt, ind.ctypes.data if ind is not None else None,
            output_name.encode(), out.ctypes.data, self.err))
        return out


def run(data, options, mask=None, extra_data=None, outputs=None, lib_path=None, model_libs=(), progress_cb=None):
    """One-call interface: data [x,y,z,t], options dict -> dict of output volumes + log."""
    data = np.asarray(data)
    with Fabber(lib_path, model_libs) as fab:
        fab.set_extent(data.shape[:3], mask)
        fab.set_options(options)
        fab.set_data("data", data)
        for name, vol in (extra_data or {}).items():
            fab.set_data(name, vol)
        log = fab.run(progress_cb)
        result = {"log": log}
        if outputs is None:
            names = ["finalMVN", "freeEnergy", "modelfit", "residuals", "noise_means", "noise_stdevs", "freeEnergyHistory"]
            for p in fab.get_model_params():
                names += ["mean_" + p, "std_" + p, "zstat_" + p, "var_" + p]
        else:
            names = outputs
        for name in names:
            if fab.data_size(name) >= 0:
                result[name] = fab.get_data(name)
        return result


# ---- model validation workflow (py/fabber.py:41-176, doc/models.rst "testing your model") ----

def _value_list(values):
    return [float(v) for v in values] if np.ndim(values) else [float(values)]


def generate_test_data(options, param_testvalues, nt=10, patchsize=10, noise=None, seed=None, lib_path=None, model_libs=()):
    """Synthetic volume made of cubic patches, one per combination of the parameters that were
    given several test values (at most three such parameters: one per axis); every voxel of a
    patch holds the model's prediction for that combination (py/fabber.py:105-176).

    -> dict(data, clean, patch_rois, param_rois): `data` = clean + N(0, noise²) when `noise` is
    given, `patch_rois` labels the patches 1.., `param_rois[p]` holds 1 + the index of p's value."""
    with Fabber(lib_path, model_libs) as fab:
        fab.set_options(options)
        names = fab.get_model_params()
        unknown = set(param_testvalues) - set(names)
        if unknown:
            raise ValueError("not parameters of model %s: %s" % (options.get("model"), sorted(unknown)))
        values = {p: _value_list(param_testvalues.get(p, 0.0)) for p in names}
        axes = [p for p in param_testvalues if len(values[p]) > 1]
        if len(axes) > 3:
            raise ValueError("at most 3 parameters may vary, got %d" % len(axes))
        counts = [len(values[p]) for p in axes] + [1] * (3 - len(axes))
        shape = tuple(c * patchsize for c in counts)
        clean = np.zeros(shape + (nt,), dtype=np.float64)
        patch_rois = np.zeros(shape, dtype=np.int32)
        param_rois = {p: np.zeros(shape, dtype=np.int32) for p in axes}
        label = 0
        for pos in np.ndindex(*counts):
            block = tuple(slice(i * patchsize, (i + 1) * patchsize) for i in pos)
            theta = [values[p][pos[axes.index(p)]] if p in axes else values[p][0] for p in names]
            clean[block] = fab.model_evaluate(theta, nt)
            label += 1
            patch_rois[block] = label
            for a, p in enumerate(axes):
                param_rois[p][block] = pos[a] + 1
    data = clean
    if noise is not None:
        data = clean + np.random.default_rng(seed).normal(0.0, noise, clean.shape)
    return {"data": data, "clean": clean, "patch_rois": patch_rois, "param_rois": param_rois}


def self_test(model, options, param_testvalues, invert=True, disp=False, lib_path=None, model_libs=(), **kwargs):
    """Generate test data from the model, fit it and report, per varied parameter, the mean
    estimate over the voxels generated with each test value (py/fabber.py:41-103).

    -> (report, log): report[param][test value] = mean output; report["noise"][input sd] =
    1/sqrt(mean noise precision)."""
    options = dict(options, model=model)
    test = generate_test_data(options, param_testvalues, lib_path=lib_path, model_libs=model_libs, **kwargs)
    report, log = {}, None
    if not invert:
        return report, log
    options.setdefault("method", "vb")
    options.setdefault("noise", "white")
    options.update({"save-mean": True, "save-noise-mean": True, "save-noise-std": True, "save-model-fit": True,
                    "allow-bad-voxels": True})
    out = run(test["data"], options, lib_path=lib_path, model_libs=model_libs)
    log = out["log"]
    for param, values in param_testvalues.items():
        values = _value_list(values)
        if len(values) < 2:
            continue
        roi = test["param_rois"][param]
        report[param] = {v: float(np.mean(out["mean_" + param][roi == i + 1])) for i, v in enumerate(values)}
        if disp:
            for v, got in report[param].items():
                print("%s: input %g -> %g output" % (param, v, got))
    noise_in = kwargs.get("noise") or 0.0
    report["noise"] = {noise_in: float(1.0 / np.sqrt(np.mean(out["noise_means"])))}
    if disp:
        print("noise: input %g -> %g output" % (noise_in, report["noise"][noise_in]))
    return report, log
