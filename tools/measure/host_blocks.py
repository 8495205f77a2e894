"""fabber_vb_run_host / fabber_vb_run_host_multi through host pointers (PCIe-inclusive), C3 problem, 1e6 voxels:
the C call alone (inputs and result arrays allocated and touched beforehand), cut into 1, 2, 4, 8 blocks on
streams of one device."""
import ctypes as C
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import numpy as np
import cases
from fabber_core_amd import hiplib, vbabi
h, y = cases.exp_problem(1000000, 100, 2, 0.02, seed=20260103, max_iterations=50)
cfg = h.cfg
V = cfg.n_voxels
arrs = dict(mvn=np.full((h.n_mvn_rows, V), np.nan), free_energy=np.full(V, np.nan), status=np.full(V, -1, dtype=np.int32),
            iterations=np.full(V, -1, dtype=np.int32))
out = vbabi.FvbOutputs()
for k, a in arrs.items():
    setattr(out, k, a.ctypes.data)
y = np.ascontiguousarray(y, dtype=np.float32)
lib = hiplib.lib()
def call(devs):
    if devs is None:
        rc = lib.fabber_vb_run_host(C.byref(cfg), y.ctypes.data, C.byref(out), 0)
    else:
        ids = (C.c_int32 * len(devs))(*devs)
        s = vbabi.FvbSummary()
        rc = lib.fabber_vb_run_host_multi(C.byref(cfg), y.ctypes.data, C.byref(out), ids, len(devs), C.byref(s))
    assert rc == 0, hiplib.last_error()
for devs in (None, [0], [0, 0], [0, 0, 0, 0], [0] * 8):
    call(devs)
    t0 = time.perf_counter()
    for _ in range(5):
        call(devs)
    print("fabber_vb_run_host" if devs is None else "%d block(s)" % len(devs), "%.1f ms" % ((time.perf_counter() - t0) / 5 * 1e3), flush=True)
