"""lane-per-voxel vs wave-per-voxel on the same problems, across voxel counts."""
import sys, os, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import cases
from fabber_core_amd import hiplib
from fabber_core_amd.device import DeviceProblem
rows = []
for name, mk in (("c3 bi-exp T=100 50 its", lambda V: cases.exp_problem(V, 100, 2, 0.02, seed=1, max_iterations=50)),
                 ("c2 exp T=50 10 its", lambda V: cases.exp_problem(V, 50, 1, 0.04, seed=1, max_iterations=10))):
    for V in (1024, 4096, 16384, 65536, 262144, 1048576):
        h, y = mk(V)
        rec = {"workload": name, "voxels": V}
        for variant in ("lane", "wave"):
            hiplib.set_variant(variant)
            prob = DeviceProblem(h, y, "cuda:0")
            prob.run(); torch.cuda.synchronize()
            n = 3
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(n):
                prob.run()
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / n
            rec[variant + "_ms"] = ms
            rec[variant + "_Mvox_s"] = V / ms / 1e3
        rows.append(rec)
        print(json.dumps(rec), flush=True)
