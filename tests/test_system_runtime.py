"""The host entry points under the SYSTEM's HIP runtime. Every other GPU test runs with the runtime PyTorch bundles (the
Python loaders import torch first, fabber_core_amd/__init__.py); a C or C++ caller of the C ABI has the system's. Round 4
found that the two differ where it matters: with ROCm 7.2's runtime the pipelined host entry point returned wrong results
while several streams took their buffers from one stream-ordered memory pool (ROCm 7.0's, the bundled one, did not show it) -
the blocks' buffers now come from plain allocations kept with the call's streams (BlockSlot, csrc/vb_api.hip). These tests
run the calls in a child process that loads nothing but the C libraries."""
import json
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def child(what, voxels):
    p = subprocess.run([sys.executable, os.path.join(HERE, "system_runtime_child.py"), what, str(voxels)], capture_output=True, text=True, timeout=600)
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert p.returncode == 0 and lines, (p.returncode, p.stdout[-500:], p.stderr[-1500:])
    out = json.loads(lines[-1])
    assert not out["torch_loaded"] and out["runtime"] and all("torch" not in r for r in out["runtime"]), out
    return out


@pytest.mark.gpu
def test_pipelined_host_call_under_the_system_runtime():
    """fabber_vb_run_host on C3 (1e6 voxels: five blocks on four streams) against the same call as one block, the caller's
    arrays used again and again, pageable and then page-locked: identical every time"""
    out = child("engine", 1_000_000)
    assert out["identical"] == [True] * 6, out
    assert 0 < out["bad"] < 200


@pytest.mark.gpu
def test_c_abi_handle_after_handle_under_the_system_runtime():
    """fabber_new .. fabber_destroy four times in one process (the image buffers of a handle go to the next one): the same
    finalMVN every time, and the one the engine's own entry point gives"""
    out = child("capi", 600_000)
    assert out["identical"] == [True] * 4 and out["matches_engine"], out


def test_the_e2e_child_of_the_bench_leaves_pytorch_alone():
    """bench.py --e2e-child stands for a C caller: it must not import torch (the system's HIP runtime is then the one the
    engine library loads), and without a GPU it fails loudly like every compute entry point."""
    root = os.path.dirname(HERE)
    code = ("import sys; sys.argv = ['bench.py', '--e2e-child']; import runpy\n"
            "try:\n    runpy.run_path(%r, run_name='not_main')\nfinally:\n    print('TORCH', 'torch' in sys.modules)\n" % os.path.join(root, "bench.py"))
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert "TORCH False" in p.stdout, (p.stdout[-300:], p.stderr[-800:])
