import sys, os, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import cases, oracle, parity, hipengine
from fabber_core_amd import hiplib
from fabber_core_amd.device import DeviceProblem
hiplib.set_variant("lane")
out = {}
# strict problems
for name, (h, y) in {"exp1 T50 10its": cases.exp_problem(4000, 50, 1, 0.04, seed=20260102, max_iterations=10, need_f=True),
                     "exp1 T50 30its": cases.exp_problem(4000, 50, 1, 0.04, seed=7, max_iterations=30, need_f=True)}.items():
    a, a2, b = oracle.run(h, y), oracle.run_fma(h, y), hipengine.run(h, y)
    ok = (a["status"] == 0)
    e_mean, e_cov, rel = parity.voxel_errors(h, a, b, ok)
    f_mean, f_cov, _ = parity.voxel_errors(h, a, a2, ok)
    out[name] = dict(gpu_err_mean_max=float(e_mean.max()), gpu_err_mean_med=float(np.median(e_mean)), gpu_err_cov_max=float(e_cov.max()),
                     cpu_floor_mean_max=float(f_mean.max()), cpu_floor_med=float(np.median(f_mean)),
                     F_rel=float(np.max(np.abs(a["free_energy"][ok]-b["free_energy"][ok])/np.maximum(1,np.abs(a["free_energy"][ok])))))
h, y = cases.exp_problem(4000, 100, 2, 0.02, seed=20260103, max_iterations=50)
cpu, cpu2, gpu = oracle.run(h, y), oracle.run_fma(h, y), hipengine.run(h, y)
tl = lambda d: {k: (v.tolist() if hasattr(v, "tolist") else v) for k, v in d.items()}
out["biexp floor"] = tl(parity.population_stats(h, cpu, cpu2))
out["biexp gpu"] = tl(parity.population_stats(h, cpu, gpu))
for mode in ("auto", "moments", "exact"):
    hiplib.set_residual_mode(mode)
    h, y = cases.exp_problem(1000000, 100, 2, 0.02, seed=20260103, max_iterations=50)
    prob = DeviceProblem(h, y, "cuda:0")
    prob.run(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3): prob.run()
    e1.record(); torch.cuda.synchronize()
    out["c3 1e6 ms " + mode] = e0.elapsed_time(e1) / 3
print(json.dumps(out, indent=1))
