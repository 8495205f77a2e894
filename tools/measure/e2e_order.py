"""Does the order in which torch and the engine first touch the GPU change fabber_vb_run_host? (bench.py's e2e leg reads
21 ms where the same function in a process without torch reads 18.7)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
mode = sys.argv[1]
if mode == "no_torch":  # the system's HIP runtime, torch never imported: a C caller's process
    os.environ["FVB_NO_TORCH"] = "1"
    sys.argv.append("--e2e-child")  # (bench.py then leaves torch alone)
if mode in ("torch_first", "torch_first_problem"):
    import torch
    torch.zeros(16, device="cuda").sum().item()
import bench
from fabber_core_amd import hiplib
w = bench.WORKLOADS["c3"]
holder, y = bench.make_problem(w, w["voxels"], 20260103, False)
if mode == "torch_first_problem":
    from fabber_core_amd.device import DeviceProblem
    prob = DeviceProblem(holder, y, torch.device("cuda:0"))
    for _ in range(3):
        prob.run()
    torch.cuda.synchronize()
res = hiplib.run_host(holder, y)
ts = []
for _ in range(6):
    t0 = time.perf_counter()
    hiplib.run_host(holder, y, into=res)
    ts.append(round((time.perf_counter() - t0) * 1e3, 2))
print(mode, ts)
