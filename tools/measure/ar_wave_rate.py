"""Throughput of the AR(1) kernels: one echo (lane kernel with stored lag moments, wave kernel) and two echoes
(lane kernel with two streaming passes per iteration - vb_lane_arn_kernel.h - against the wave kernel), with and
without the free energy."""
import sys, os, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import cases
from fabber_core_amd import hiplib, vbabi
from fabber_core_amd.device import DeviceProblem
V, T = 262144, 200
CASES = [("1 echo, lane", dict(), "auto"), ("1 echo, wave", dict(), "wave")]
for cross in ("none", "same", "dual"):
    for need_f in (False, True):
        CASES.append(("2 echoes %s%s, lane" % (cross, ", F" if need_f else ""), dict(num_echoes=2, ar_cross_terms=cross, need_f=need_f), "auto"))
CASES += [("2 echoes none, wave", dict(num_echoes=2), "wave"), ("2 echoes dual, wave", dict(num_echoes=2, ar_cross_terms="dual"), "wave")]
for label, kw, variant in CASES:
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
