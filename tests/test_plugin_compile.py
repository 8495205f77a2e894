"""A model library written for the reference builds unchanged against this package's headers and
loads through fabber_load_models (fwdmodel.cc:25-27 hooks, factories.h registration).

Needs the reference's example sources, which exist only in the development container (they are
not copied into this repository): skipped elsewhere."""
import os
import shutil
import subprocess

import numpy as np
import pytest

from fabber_core_amd import fabber, hiplib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXAMPLES = "/root/reference/examples"
HOST = os.path.join(ROOT, "fabber_core_amd", "csrc", "host")
LIBDIR = os.path.join(ROOT, "fabber_core_amd", "lib")

pytestmark = [
    pytest.mark.skipif(not os.path.exists(os.path.join(EXAMPLES, "fwdmodel_exp.cc")), reason="reference examples not present"),
    pytest.mark.skipif(shutil.which("g++") is None, reason="no g++"),
    pytest.mark.skipif(not os.path.exists(os.path.join(LIBDIR, "libfabbercore_amd.so")), reason="host library not built"),
]


@pytest.fixture(scope="module")
def plugin(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("plugin") / "libfabber_models_exp.so")
    cmd = ["g++", "-std=c++17", "-shared", "-fPIC", "-Wno-deprecated-declarations", "-I", HOST, "-I", os.path.join(HOST, "fabber_core"),
           os.path.join(EXAMPLES, "fwdmodel_exp.cc"), os.path.join(EXAMPLES, "exp_models.cc"), "-o", out,
           "-L", LIBDIR, "-lfabbercore_amd", "-Wl,-rpath," + LIBDIR, "-Wl,--no-undefined"]
    p = subprocess.run(cmd, capture_output=True, text=True)
    assert p.returncode == 0, p.stderr[-3000:]
    return out


def test_reference_example_model_builds_loads_and_evaluates(plugin):
    f = fabber.Fabber(model_libs=[plugin])
    assert "exp" in f.get_models()
    f.set_options({"model": "exp", "num-exps": 2, "dt": 0.02, "noise": "white", "method": "vb"})
    assert f.get_model_params() == ["amp1", "r1", "amp2", "r2"]
    y = np.asarray(f.model_evaluate([1.0, 1.0, 0.5, 6.0], 10))
    t = np.arange(10) * 0.02
    assert np.allclose(y, np.exp(-t) + 0.5 * np.exp(-6 * t), rtol=1e-6)


@pytest.mark.skipif(hiplib.available() and hiplib.device_count() > 0, reason="a GPU is present")
def test_host_evaluated_model_still_needs_the_gpu(plugin):
    """The plugin's class knows nothing about FwdModel::GetDeviceModel, so its Evaluate runs on the
    host - but only that: the rest of the loop is the GPU engine, and without a device the run fails
    with a message instead of falling back to a CPU implementation."""
    f = fabber.Fabber(model_libs=[plugin])
    f.set_options({"model": "exp", "num-exps": 1, "dt": 0.04, "noise": "white", "method": "vb"})
    f.set_extent((2, 2, 1))
    f.set_data("data", np.ones((2, 2, 1, 10), dtype=np.float32))
    with pytest.raises(fabber.FabberError, match="no HIP device"):
        f.run()
