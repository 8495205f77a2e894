"""NIfTI-1 I/O (rundata_newimage.cc) and the `fabber` command line tool (rank 2:
fabber_core.cc:88-323). CPU: the reader / writer against an independent Python implementation,
the information queries of the CLI. GPU: a complete run from NIfTI files, compared with the same
problem through the C API."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import nifti_utils as nu
from fabber_core_amd import fabber, hiplib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "fabber_core_amd", "bin", "fabber")
CORE = os.path.join(ROOT, "fabber_core_amd", "lib", "libfabbercore_amd.so")
pytestmark = pytest.mark.skipif(not (os.path.exists(EXE) and os.path.exists(CORE)), reason="host library / CLI not built")


def core():
    lib = C.CDLL(CORE)
    lib.fabber_nifti_read.argtypes = [C.c_char_p, C.POINTER(C.c_int), C.c_void_p, C.c_ulonglong, C.c_char_p]
    lib.fabber_nifti_write.argtypes = [C.c_char_p, C.POINTER(C.c_int), C.c_void_p, C.c_int, C.c_void_p, C.c_char_p]
    return lib


def cxx_read(path):
    lib, dims, err = core(), (C.c_int * 4)(), C.create_string_buffer(256)
    assert lib.fabber_nifti_read(path.encode(), dims, None, 0, err) == 0, err.value
    nx, ny, nz, nt = dims
    buf = np.empty((nt, nz, ny, nx), dtype=np.float32)
    assert lib.fabber_nifti_read(path.encode(), dims, buf.ctypes.data, buf.size, err) == 0, err.value
    return buf.transpose(3, 2, 1, 0)


def run_cli(*args, cwd=None):
    return subprocess.run([EXE] + list(args), capture_output=True, text=True, cwd=cwd, timeout=600)


# ---------------------------------------------------------------------------------------------
# CPU
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype", [np.uint8, np.int16, np.int32, np.float32, np.float64, np.uint16])
@pytest.mark.parametrize("ext,order", [(".nii", "<"), (".nii.gz", "<"), (".nii.gz", ">")])
def test_reader_against_independent_writer(tmp_path, dtype, ext, order):
    rng = np.random.default_rng(1)
    vol = (rng.random((5, 4, 3, 6)) * 100).astype(dtype)
    path = str(tmp_path / ("vol" + ext))
    nu.write(path, vol, byteorder=order)
    got = cxx_read(path)
    assert got.shape == vol.shape and np.array_equal(got, vol.astype(np.float32))
    # without the extension (fsl_imageexists behaviour)
    assert np.array_equal(cxx_read(str(tmp_path / "vol")), got)


def test_reader_applies_scaling_and_reads_3d(tmp_path):
    vol = np.arange(24, dtype=np.int16).reshape(4, 3, 2)
    path = str(tmp_path / "scaled.nii.gz")
    nu.write(path, vol, scl=(0.5, 10.0))
    got = cxx_read(path)
    assert got.shape == (4, 3, 2, 1) and np.allclose(got[..., 0], vol * 0.5 + 10.0)


def test_writer_against_independent_reader(tmp_path):
    rng = np.random.default_rng(2)
    vol = rng.normal(0, 1, (6, 5, 4, 3)).astype(np.float32)
    lib, err = core(), C.create_string_buffer(256)
    for name in ("out.nii.gz", "out.nii"):
        path = str(tmp_path / name)
        data = np.ascontiguousarray(vol.transpose(3, 2, 1, 0))
        pix = (C.c_float * 3)(2.0, 2.5, 3.0)
        assert lib.fabber_nifti_write(path.encode(), (C.c_int * 4)(6, 5, 4, 3), data.ctypes.data, 1005, pix, err) == 0, err.value
        got, hdr = nu.read(path)
        assert np.array_equal(got.astype(np.float32), vol)
        assert hdr["intent_code"] == 1005 and hdr["datatype"] == 16 and hdr["magic"] == b"n+1\0"
        assert hdr["pixdim"][1:4] == (2.0, 2.5, 3.0) and hdr["vox_offset"] == 352


def test_reader_errors_are_messages(tmp_path):
    lib, dims, err = core(), (C.c_int * 4)(), C.create_string_buffer(256)
    assert lib.fabber_nifti_read(str(tmp_path / "missing").encode(), dims, None, 0, err) == -1
    assert b"no such image" in err.value
    bad = tmp_path / "bad.nii"
    bad.write_bytes(b"\0" * 400)
    assert lib.fabber_nifti_read(str(bad).encode(), dims, None, 0, err) == -1
    assert b"not a NIfTI-1" in err.value


def test_cli_information_queries(tmp_path):
    r = run_cli("--version")
    assert r.returncode == 0 and r.stdout.startswith("Fabber ")
    assert run_cli("--listmodels").stdout.split() == ["exp", "linear", "poly"]
    assert run_cli("--listmethods").stdout.split() == ["nlls", "spatialvb", "vb"]
    assert run_cli("--listparams", "--model=poly", "--degree=2").stdout.split() == ["c0", "c1", "c2"]
    assert run_cli("--listparams", "--model=exp", "--num-exps=2", "--dt=0.1").stdout.split() == ["amp1", "r1", "amp2", "r2"]
    r = run_cli()
    assert r.returncode == 0 and "Usage: fabber" in r.stdout and "--output" in r.stdout
    r = run_cli("--help", "--method=vb")
    assert r.returncode == 0 and "max-iterations" in r.stdout
    r = run_cli("--help", "--model=poly")
    assert r.returncode == 0 and "degree" in r.stdout
    # --evaluate: model prediction for given parameters (fabber_core.cc:229-263)
    (tmp_path / "params.txt").write_text("2\n0\n3\n")
    r = run_cli("--model=poly", "--degree=2", "--evaluate=", "--evaluate-params=" + str(tmp_path / "params.txt"), "--evaluate-nt=4")
    assert r.returncode == 0, r.stderr
    assert [float(x) for x in r.stdout.split()] == [5.0, 14.0, 29.0, 50.0]
    # errors: exit code 1 and a message
    r = run_cli("--listparams", "--model=nosuchmodel")
    assert r.returncode == 1 and "nosuchmodel" in r.stderr
    r = run_cli("--model=poly", "--degree=1", "--method=vb", "--noise=white", "--data=" + str(tmp_path / "nothere"),
                "--output=" + str(tmp_path / "o"))
    assert r.returncode == 1 and "nothere" in r.stderr


# ---------------------------------------------------------------------------------------------
# GPU
# ---------------------------------------------------------------------------------------------
@pytest.mark.gpu
def test_cli_run_from_nifti_matches_the_c_api(tmp_path):
    rng = np.random.default_rng(3)
    shape, T = (7, 6, 5), 12
    t = np.arange(1, T + 1)
    c = rng.uniform(-3, 3, shape + (3,))
    data = (c[..., 0:1] + c[..., 1:2] * t + c[..., 2:3] * t * t + rng.normal(0, 0.1, shape + (T,))).astype(np.float32)
    mask = (rng.random(shape) < 0.8).astype(np.uint8)
    nu.write(str(tmp_path / "data.nii.gz"), data, pixdim=(2.0, 2.0, 3.0, 1.5))
    nu.write(str(tmp_path / "mask.nii.gz"), mask, pixdim=(2.0, 2.0, 3.0, 1.0))
    (tmp_path / "opts.txt").write_text("# options file, -f form\nmodel=poly\ndegree=2\nnoise=white\nmethod=vb\nmax-iterations=8\nsave-model-fit\n")
    out = str(tmp_path / "out")
    r = run_cli("-f", str(tmp_path / "opts.txt"), "--data=" + str(tmp_path / "data"), "--mask=" + str(tmp_path / "mask.nii.gz"),
                "--output=" + out, "--save-residuals")
    assert r.returncode == 0, r.stdout + r.stderr
    assert "Final logfile" in r.stdout
    assert (tmp_path / "out" / "logfile").exists()
    assert (tmp_path / "out" / "paramnames.txt").read_text().split() == ["c0", "c1", "c2"]
    ref = fabber.run(data, {"model": "poly", "degree": 2, "noise": "white", "method": "vb", "max-iterations": 8, "save-mean": True,
                            "save-std": True, "save-zstat": True, "save-mvn": True, "save-noise-mean": True, "save-noise-std": True,
                            "save-free-energy": True, "save-model-fit": True, "save-residuals": True}, mask=mask)
    # the CLI's backwards-compatible default outputs (rundata.cc compat options) plus the requested ones
    for name in ("mean_c0", "mean_c2", "std_c1", "zstat_c0", "finalMVN", "noise_means", "noise_stdevs", "freeEnergy", "modelfit", "residuals"):
        got, hdr = nu.read(os.path.join(out, name + ".nii.gz"))
        want = ref[name] if ref[name].ndim == 4 else ref[name][..., None]
        assert got.shape == want.shape, name
        assert np.array_equal(got.astype(np.float32), want.astype(np.float32)), name
        assert hdr["pixdim"][1:4] == (2.0, 2.0, 3.0), name          # geometry of the mask
        assert np.all(got[mask == 0] == 0), name
        assert hdr["intent_code"] == (1005 if name == "finalMVN" else 0)
    # a second run into the same --output appends '+' unless --overwrite is given (rundata.cc GetOutputDir)
    r = run_cli("-f", str(tmp_path / "opts.txt"), "--data=" + str(tmp_path / "data"), "--mask=" + str(tmp_path / "mask.nii.gz"), "--output=" + out)
    assert r.returncode == 0 and (tmp_path / "out+").is_dir()


@pytest.mark.gpu
def test_cli_spatialvb_without_a_mask(tmp_path):
    rng = np.random.default_rng(4)
    shape, T = (6, 6, 4), 50
    t = np.arange(T) * 0.04
    amp = 1.0 + 0.3 * np.sin(np.arange(shape[0]) / 2.0)[:, None, None] + np.zeros(shape)
    data = (amp[..., None] * np.exp(-t) + rng.normal(0, 0.1, shape + (T,))).astype(np.float32)
    nu.write(str(tmp_path / "data.nii"), data)
    out = str(tmp_path / "sp")
    r = run_cli("--model=exp", "--num-exps=1", "--dt=0.04", "--noise=white", "--method=spatialvb", "--param-spatial-priors=MN",
                "--max-iterations=5", "--data=" + str(tmp_path / "data.nii"), "--output=" + out, "--simple-output")
    assert r.returncode == 0, r.stdout + r.stderr
    ref = fabber.run(data, {"model": "exp", "num-exps": 1, "dt": 0.04, "noise": "white", "method": "spatialvb",
                            "param-spatial-priors": "MN", "max-iterations": 5, "save-mean": True})
    got, _ = nu.read(os.path.join(out, "mean_amp1.nii.gz"))
    assert np.array_equal(got[..., 0].astype(np.float32), ref["mean_amp1"].astype(np.float32))
