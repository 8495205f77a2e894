"""fabber_vb_release_cached_memory (the pools trimmed to nothing) between two identical calls, under the system's runtime
(FVB_NO_TORCH=1) or PyTorch's (WITH_TORCH=1): the second call must give the first one's bits."""
import os, sys
if os.environ.get("WITH_TORCH") != "1":
    os.environ["FVB_NO_TORCH"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import cases
from fabber_core_amd import hiplib
h, y = cases.exp_problem(8192, 50, 1, 0.04, seed=3, max_iterations=5)
lib = hiplib.lib()
lib.fabber_vb_release_cached_memory.restype = None
if len(sys.argv) > 1 and sys.argv[1] == "multi":  # the state tests/test_multi_device.py leaves behind
    hm, ym = cases.exp_problem(20000, 50, 1, 0.04, seed=4, max_iterations=5)
    hiplib.run_host(hm, ym, devices=[0, 0, 0])
first = hiplib.run_host(h, y)
for rep in range(4):
    again = hiplib.run_host(h, y)
    print("repeat", rep, bool(np.array_equal(first["mvn"], again["mvn"], equal_nan=True)))
for rep in range(10):
    lib.fabber_vb_release_cached_memory()
    again = hiplib.run_host(h, y)
    d = np.flatnonzero(~np.all(first["mvn"] == again["mvn"], axis=0))
    rel = np.nanmax(np.abs(first["mvn"] - again["mvn"]) / np.maximum(np.abs(first["mvn"]), 1e-300))
    print("after release", rep, bool(np.array_equal(first["mvn"], again["mvn"], equal_nan=True)), "voxels that differ:", d.size, d[:6],
          "max rel diff", float(rel), "status equal", bool(np.array_equal(first["status"], again["status"])), "iterations equal",
          bool(np.array_equal(first["iterations"], again["iterations"])), "rows that differ", np.flatnonzero(np.any(first["mvn"] != again["mvn"], axis=1))[:12],
          "kernel", hiplib.kernel_name(h))
