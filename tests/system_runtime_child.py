"""Child process of tests/test_system_runtime.py: the host entry points with NOTHING but the C libraries loaded - the system's
HIP runtime, as a C / C++ caller has it (the test process itself has PyTorch's bundled runtime). Prints one JSON line."""
import json
import os
import sys

os.environ["FVB_NO_TORCH"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import cases  # noqa: E402
from fabber_core_amd import fabber, hiplib  # noqa: E402

what, V = sys.argv[1], int(sys.argv[2])
h, y = cases.exp_problem(V, 100, 2, 0.02, seed=20260103, max_iterations=50)
out = {}
if what == "engine":
    os.environ["FVB_HOST_BLOCK_VOXELS"] = "0"
    one = hiplib.run_host(h, y)
    os.environ.pop("FVB_HOST_BLOCK_VOXELS")
    piped = hiplib.run_host(h, y)
    same = []
    for rep in range(3):  # (the same host arrays again and again: what a caller that keeps its buffers does)
        piped = hiplib.run_host(h, y, into=piped)
        same.append(all(bool(np.array_equal(one[k], piped[k], equal_nan=True)) for k in ("mvn", "status", "iterations")))
    L = hiplib.lib()
    y = np.ascontiguousarray(y)
    pinned = [a for a in [y] + [v for v in piped.values() if isinstance(v, np.ndarray)]]
    import ctypes as C
    L.fabber_vb_pin_host_buffer.argtypes = [C.c_void_p, C.c_uint64]
    L.fabber_vb_unpin_host_buffer.argtypes = [C.c_void_p]
    for a in pinned:
        assert L.fabber_vb_pin_host_buffer(a.ctypes.data, a.nbytes) == 0
    for rep in range(3):
        piped = hiplib.run_host(h, y, into=piped)
        same.append(all(bool(np.array_equal(one[k], piped[k], equal_nan=True)) for k in ("mvn", "status", "iterations")))
    for a in pinned:
        L.fabber_vb_unpin_host_buffer(a.ctypes.data)
    out.update(identical=same, bad=int((one["status"] != 0).sum()))
else:
    data = np.ascontiguousarray(y.T.reshape(V, 1, 1, 100))
    opts = {"model": "exp", "num-exps": 2, "dt": 0.02, "max-iterations": 50, "noise": "white", "method": "vb", "save-mean": True,
            "save-mvn": True, "allow-bad-voxels": True}
    first, same = None, []
    for rep in range(4):  # (the host library hands the previous handle's image buffers out again)
        mvn = fabber.run(data, opts)["finalMVN"]
        first = mvn if first is None else first
        same.append(bool(np.array_equal(first, mvn, equal_nan=True)))
    eng = hiplib.run_host(h, y)
    rows = eng["mvn"].shape[0]
    got = first.transpose(3, 2, 1, 0).reshape(rows, -1)
    out.update(identical=same, matches_engine=bool(np.allclose(got, eng["mvn"].astype(np.float32), rtol=1e-6, atol=0, equal_nan=True)))
out.update(runtime=sorted({l.split()[-1] for l in open("/proc/self/maps") if "amdhip64" in l}), torch_loaded="torch" in sys.modules)
print(json.dumps(out))
