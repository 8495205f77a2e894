"""Which voxels of the bi-exponential fit end with a non-zero status, on the GPU and in the two
CPU builds of the oracle (the fit is chaotic: the count differs between any two builds)."""
import sys, os, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import cases, oracle, hipengine
from fabber_core_amd import hiplib
hiplib.set_variant("lane")
V = int(sys.argv[1]) if len(sys.argv) > 1 else 40000
h, y = cases.exp_problem(V, 100, 2, 0.02, seed=20260103, max_iterations=50)
gpu = hipengine.run(h, y)
out = {"V": V, "gpu_bad": float(np.mean(gpu["status"] != 0)), "gpu_status_hist": np.bincount(gpu["status"], minlength=6).tolist(),
       "gpu_bad_iterations": np.bincount(gpu["iterations"][gpu["status"] != 0], minlength=51).tolist()}
if V <= 100000:
    a, b = oracle.run(h, y), oracle.run_fma(h, y)
    out.update(cpu_bad=float(np.mean(a["status"] != 0)), cpu_fma_bad=float(np.mean(b["status"] != 0)),
               cpu_status_hist=np.bincount(a["status"], minlength=6).tolist(),
               cpu_bad_iterations=np.bincount(a["iterations"][a["status"] != 0], minlength=51).tolist(),
               both_bad=int(np.sum((a["status"] != 0) & (gpu["status"] != 0))))
print(json.dumps(out))
