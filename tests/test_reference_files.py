"""The data files the reference's own tests hold (tests/golden/reference_files/, see its README) through the product's
readers: the C++ NIfTI-1 reader (csrc/host/nifti_io.cc) and the VEST / ASCII matrix readers behind `basis=` parse
files the reference's toolchain (FSL NEWIMAGE / miscmaths) wrote, and must give the numbers tests/golden/make_golden.py
extracted with its own independent Python code. GPU part: the command line tests of test/test_commandline.cc that run
on test_data_small.nii.gz (:236-345)."""
import ctypes as C
import os
import shutil
import subprocess

import numpy as np
import pytest

import golden_utils as gu
import nifti_utils as nu
from fabber_core_amd import fabber

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FILES = os.path.join(ROOT, "tests", "golden", "reference_files")
EXE = os.path.join(ROOT, "fabber_core_amd", "bin", "fabber")
CORE = os.path.join(ROOT, "fabber_core_amd", "lib", "libfabbercore_amd.so")
pytestmark = pytest.mark.skipif(not (os.path.exists(EXE) and os.path.exists(CORE)), reason="host library / CLI not built")


def cxx_read(path):
    lib, dims, err = C.CDLL(CORE), (C.c_int * 4)(), C.create_string_buffer(256)
    lib.fabber_nifti_read.argtypes = [C.c_char_p, C.POINTER(C.c_int), C.c_void_p, C.c_ulonglong, C.c_char_p]
    assert lib.fabber_nifti_read(path.encode(), dims, None, 0, err) == 0, err.value
    nx, ny, nz, nt = dims
    buf = np.empty((nt, nz, ny, nx), dtype=np.float32)
    assert lib.fabber_nifti_read(path.encode(), dims, buf.ctypes.data, buf.size, err) == 0, err.value
    return buf  # [t][z][y][x]


def test_reader_on_the_reference_input_volume():
    """test_data_small.nii.gz: 3 x 3 x 2 x 106, int16"""
    want = np.load(os.path.join(ROOT, "tests", "golden", "reference_data_small.npz"))["data"]
    got = cxx_read(os.path.join(FILES, "test_data_small.nii.gz"))
    assert got.shape == want.shape == (106, 2, 3, 3)
    assert np.array_equal(got, want.astype(np.float32))
    assert np.array_equal(cxx_read(os.path.join(FILES, "test_data_small")), got)  # (extension optional, as FSL's reader)


def test_reader_on_the_reference_masks():
    ref = gu.load_reference_outdata()
    mask = cxx_read(os.path.join(FILES, "test_mask_small.nii.gz"))
    assert mask.shape[0] == 1 and tuple(mask.shape[:0:-1]) == tuple(int(s) for s in ref["mask_shape"])
    # rundata_newimage.cc:80 binarises with > 1e-16
    assert np.array_equal(np.flatnonzero(mask.reshape(-1) > 1e-16), ref["mask_index"])
    assert np.count_nonzero(cxx_read(os.path.join(FILES, "test_mask_empty.nii.gz")) > 1e-16) == 0
    assert np.count_nonzero(cxx_read(os.path.join(FILES, "test_mask.nii.gz")) > 1e-16) > 0


def test_reader_on_outputs_the_reference_binary_wrote():
    """outdata_poly/finalMVN.nii.gz carries the SYMMATRIX intent (code 1005, rundata_newimage.cc:163-181): 15 rows =
    the packed 4 x 4 MVN + 1; the masked voxels are the golden vectors of tests/test_oracle_golden.py"""
    ref = gu.load_reference_outdata()
    idx = ref["mask_index"]
    path = os.path.join(FILES, "outdata_poly_finalMVN.nii.gz")
    got = cxx_read(path)
    assert got.shape[0] == 15
    assert np.array_equal(got.reshape(15, -1)[:, idx], ref["poly/finalMVN"])
    _, hdr = nu.read(path)  # (independent header read)
    assert hdr["intent_code"] == 1005 and hdr["datatype"] == 16
    got = cxx_read(os.path.join(FILES, "outdata_poly_mean_c0.nii.gz"))
    assert np.array_equal(got.reshape(1, -1)[:, idx], ref["poly/mean_c0"])


@pytest.mark.parametrize("name", ["test_linear_design.mat", "test_linear_design_ascii.mat"])
def test_matrix_readers_on_the_reference_design_files(name):
    """basis=<file>: VEST (/NumWaves, /NumPoints, /Matrix) and plain ASCII (tools.cc:27-45 -> MISCMATHS::read_vest /
    read_ascii_matrix). The linear model's prediction for unit parameter vectors is the design matrix column by
    column (fwdmodel_linear.cc:53-81), through fabber_model_evaluate."""
    design = gu.load_reference_outdata()["linear_design"]
    with fabber.Fabber() as fab:
        fab.set_options({"model": "linear", "basis": os.path.join(FILES, name)})
        assert len(fab.get_model_params()) == 4
        for k in range(4):
            unit = np.zeros(4)
            unit[k] = 1.0
            col = fab.model_evaluate(unit, 106)
            assert np.array_equal(col, design[:, k].astype(np.float32)), (name, k)
        mixed = fab.model_evaluate([1.0, -2.0, 0.5, 3.0], 106)
        assert np.allclose(mixed, design @ np.array([1.0, -2.0, 0.5, 3.0]), rtol=1e-6, atol=1e-5)


# ---------------------------------------------------------------------------------------------
# GPU: the reference's command line tests on test_data_small.nii.gz (test_commandline.cc:236-345)
# ---------------------------------------------------------------------------------------------
def run_cli(*args, cwd=None):
    return subprocess.run([EXE] + list(args), capture_output=True, text=True, cwd=cwd, timeout=600)


BASE = ["--model=poly", "--output=out.tmp", "--method=vb", "--noise=white", "--data=" + os.path.join(FILES, "test_data_small.nii.gz")]


def logfile(cwd, name="out.tmp"):
    return open(os.path.join(str(cwd), name, "logfile")).read()


@pytest.mark.gpu
def test_cl_poly_model_without_a_mask_and_output_properties(tmp_path):
    """PolyModelNoMask (:250-262) and OutputCopiesPropsNoMask (:236-247): the output images carry the input's voxel
    sizes; the fit equals the same volume through the C API"""
    r = run_cli(*BASE, "--degree=2", cwd=tmp_path)  # (the tool saves the means by default, rundata.cc:221-231)
    assert r.returncode == 0, r.stderr
    log = logfile(tmp_path)
    for s in ("model=poly", "method=vb", "noise=white", "test_data_small.nii.gz"):
        assert s in log
    _, hin = nu.read(os.path.join(FILES, "test_data_small.nii.gz"))
    mean, hout = nu.read(str(tmp_path / "out.tmp" / "mean_c0.nii.gz"))
    assert hout["pixdim"][1:4] == hin["pixdim"][1:4] and hout["dim"][1:4] == hin["dim"][1:4]
    vol = np.load(os.path.join(ROOT, "tests", "golden", "reference_data_small.npz"))["data"]  # [t][z][y][x]
    out = fabber.run(vol.transpose(3, 2, 1, 0).astype(np.float32), {"model": "poly", "degree": 2, "method": "vb", "noise": "white", "save-mean": True})
    assert np.array_equal(mean[..., 0].astype(np.float32), out["mean_c0"])


@pytest.mark.gpu
def test_cl_no_overwrite_overwrite_and_unused_options(tmp_path):
    """NoOverwrite (:265-294): a second run into an existing directory goes to "<name>+"; Overwrite (:297-319);
    UnusedParams (:338-352): an option nobody read is a WARNING in the logfile"""
    assert run_cli(*BASE, "--degree=2", cwd=tmp_path).returncode == 0
    assert "degree=2" in logfile(tmp_path)
    assert run_cli(*BASE, "--degree=1", cwd=tmp_path).returncode == 0
    assert "degree=1" in logfile(tmp_path, "out.tmp+") and "degree=2" in logfile(tmp_path)
    shutil.rmtree(str(tmp_path / "out.tmp+"))
    assert run_cli(*BASE, "--degree=1", "--overwrite", cwd=tmp_path).returncode == 0
    assert "degree=1" in logfile(tmp_path) and not os.path.exists(str(tmp_path / "out.tmp+"))
    assert run_cli(*BASE, "--degree=2", "--overwrite", cwd=tmp_path).returncode == 0
    assert "WARNING" not in logfile(tmp_path)
    assert run_cli(*BASE, "--degree=2", "--overwrite", "--squaffle", cwd=tmp_path).returncode == 0
    log = logfile(tmp_path)
    assert "WARNING" in log and "Unused option" in log


@pytest.mark.gpu
def test_cl_mask_from_the_reference_file(tmp_path):
    """--mask with a file whose grid is not the data's is refused with a message (the reference's masks belong to the
    missing full-size volume; what can be checked with them here is the refusal), and a mask cut to the small
    volume's grid restricts the fit to its voxels"""
    r = run_cli(*BASE, "--degree=2", "--mask=" + os.path.join(FILES, "test_mask_small.nii.gz"), cwd=tmp_path)
    assert r.returncode == 1 and r.stderr.strip() != ""
    mask = np.zeros((3, 3, 2), dtype=np.int16)
    mask[1:, :2, :] = 1
    nu.write(str(tmp_path / "m.nii.gz"), mask)
    r = run_cli(*BASE, "--degree=2", "--overwrite", "--mask=" + str(tmp_path / "m.nii.gz"), cwd=tmp_path)
    assert r.returncode == 0, r.stderr
    mean, _ = nu.read(str(tmp_path / "out.tmp" / "mean_c0.nii.gz"))
    assert np.all(mean[..., 0][mask == 0] == 0) and np.all(mean[..., 0][mask != 0] != 0)
