"""fabber_vb_run_host under the SYSTEM's HIP runtime (FVB_NO_TORCH=1: what a C caller links) against the same call as one
block: are the pipelined copies right there too?"""
import os, sys
if os.environ.get("WITH_TORCH") != "1":
    os.environ["FVB_NO_TORCH"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import cases
from fabber_core_amd import hiplib
V = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
h, y = cases.exp_problem(V, 100, 2, 0.02, seed=20260103, max_iterations=50)
os.environ["FVB_HOST_BLOCK_VOXELS"] = "0"
one = hiplib.run_host(h, y)
os.environ.pop("FVB_HOST_BLOCK_VOXELS")
piped = hiplib.run_host(h, y)
if os.environ.get("PIN") == "1":
    import ctypes as C
    L = hiplib.lib()
    L.fabber_vb_pin_host_buffer.restype = C.c_int32
    L.fabber_vb_pin_host_buffer.argtypes = [C.c_void_p, C.c_size_t]
    y = np.ascontiguousarray(y)
    print("pinned:", [L.fabber_vb_pin_host_buffer(a.ctypes.data, a.nbytes) for a in [y] + [v for v in piped.values() if isinstance(v, np.ndarray)]])
for rep in range(4):
    piped = hiplib.run_host(h, y, into=piped)
    same = {k: bool(np.array_equal(one[k], piped[k], equal_nan=True)) for k in ("mvn", "status", "iterations")}
    bad = np.flatnonzero(~np.all(np.isclose(one["mvn"], piped["mvn"], rtol=0, atol=0, equal_nan=True), axis=0))
    print(rep, same, "status!=0:", int((one["status"] != 0).sum()), int((piped["status"] != 0).sum()), "voxels that differ:", bad.size,
          bad[:8], bad[-3:] if bad.size else "")
libs = {l.split()[-1] for l in open("/proc/self/maps") if "amdhip64" in l}
print(libs)
