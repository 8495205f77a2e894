"""The C++ class API (FabberRunData, exceptions, factories) that model libraries and embedding
programs compile against: a small C++ program with the cases of the reference's test_rundata.cc
is built against fabber_core_amd/csrc/host and libfabbercore_amd.so and run. CPU only."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "fabber_core_amd", "csrc", "host")
LIBDIR = os.path.join(ROOT, "fabber_core_amd", "lib")

pytestmark = [
    pytest.mark.skipif(shutil.which("g++") is None, reason="no g++"),
    pytest.mark.skipif(not os.path.exists(os.path.join(LIBDIR, "libfabbercore_amd.so")), reason="host library not built"),
]


def build_and_run(tmp_path, name):
    exe = str(tmp_path / name)
    cmd = ["g++", "-std=c++17", "-I", HOST, os.path.join(ROOT, "tests", "cpp", name + ".cc"), "-o", exe,
           "-L", LIBDIR, "-lfabbercore_amd", "-lfabber_vb_hip", "-Wl,-rpath," + LIBDIR]
    p = subprocess.run(cmd, capture_output=True, text=True)
    assert p.returncode == 0, p.stderr[-3000:]
    r = subprocess.run([exe], capture_output=True, text=True, cwd=str(tmp_path), timeout=300)
    assert r.returncode == 0 and "all checks passed" in r.stdout, r.stdout + r.stderr


def test_rundata_class_api(tmp_path):
    build_and_run(tmp_path, "test_rundata_api")


@pytest.mark.gpu
def test_run_through_the_class_api(tmp_path):
    build_and_run(tmp_path, "test_run_class_api")
