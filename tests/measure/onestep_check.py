import sys, os, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import cases, oracle, parity, hipengine
from fabber_core_amd import hiplib
from fabber_core_amd.device import DeviceProblem
hiplib.set_variant("lane")
V = 1024 + 13
out = {}
for k in (0, 1, 2, 5, 13):
    h, y = cases.exp_problem(V, 100, 2, 0.02, seed=20260103, max_iterations=max(k, 1))
    state = oracle.run(h, y)["mvn"] if k > 0 else None
    h1, _ = cases.exp_problem(V, 100, 2, 0.02, seed=20260103, max_iterations=1, init_mvn=state, need_f=True)
    a, a2, b = oracle.run(h1, y), oracle.run_fma(h1, y), hipengine.run(h1, y)
    ok = np.isfinite(a["mvn"]).all(axis=0) & (a["status"] == 0)
    e, _, _ = parity.voxel_errors(h1, a, b, ok)
    f, _, _ = parity.voxel_errors(h1, a, a2, ok)
    out["k=%d" % k] = dict(gpu_median=float(np.median(e)), gpu_q90=float(np.quantile(e, .9)), cpu_floor_median=float(np.median(f)), cpu_floor_q90=float(np.quantile(f, .9)))
h, y = cases.exp_problem(1000000, 100, 2, 0.02, seed=20260103, max_iterations=50)
prob = DeviceProblem(h, y, "cuda:0")
prob.run(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(3): prob.run()
e1.record(); torch.cuda.synchronize()
out["c3 ms"] = e0.elapsed_time(e1) / 3
print(json.dumps(out, indent=1))
