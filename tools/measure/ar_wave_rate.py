"""Throughput of the wave-per-voxel AR(1) kernel (two echoes) against the one-echo lane kernel."""
import sys, os, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import cases
from fabber_core_amd import hiplib, vbabi
from fabber_core_amd.device import DeviceProblem
V, T = 262144, 200
for label, kw, variant in (("1 echo, lane", dict(), "auto"), ("1 echo, wave", dict(), "wave"),
                           ("2 echoes none, wave", dict(num_echoes=2), "auto"), ("2 echoes dual, wave", dict(num_echoes=2, ar_cross_terms="dual"), "auto")):
    h, y = cases.linear_problem(V, T, seed=1, max_iterations=10, noise=vbabi.NOISE_AR1, **kw)
    hiplib.set_variant(variant)
    prob = DeviceProblem(h, y, "cuda:0")
    prob.run(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        prob.run()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 3
    print(json.dumps({"case": label, "kernel": prob.kernel, "voxels": V, "T": T, "ms": ms, "Mvox_s": V / ms / 1e3}), flush=True)
    del prob
