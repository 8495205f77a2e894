import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import cases
from fabber_core_amd import hiplib
h, y = cases.exp_problem(1000000, 100, 2, 0.02, seed=20260103, max_iterations=50)
hiplib.run_host(h, y)
ts = []
for _ in range(3):
    t0 = time.perf_counter(); hiplib.run_host(h, y); ts.append(time.perf_counter() - t0)
print("fabber_vb_run_host C3 1e6 voxels (host buffers in, host buffers out): %.1f ms -> %.2f Mvox/s" % (min(ts) * 1e3, 1.0 / min(ts)))
