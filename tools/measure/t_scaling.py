"""Kernel time of the single-exponential fit against the series length: fixed cost per iteration
(serial algebra, pass start-up) vs cost per timepoint."""
import sys, os, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import cases
from fabber_core_amd import hiplib
from fabber_core_amd.device import DeviceProblem
hiplib.set_variant("lane")
V, its = 1 << 20, 10
for nexp, dt in ((1, 0.04), (2, 0.02)):
    for T in (25, 50, 100, 200):
        h, y = cases.exp_problem(V, T, nexp, dt, seed=1, max_iterations=its)
        prob = DeviceProblem(h, y, "cuda:0")
        prob.run(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            prob.run()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        print(json.dumps({"num_exps": nexp, "T": T, "ms": ms, "us_per_wave_iteration": ms * 1e3 / its / (V / 64) * 1024 * (3 if nexp == 1 else 2)}), flush=True)
        del prob
