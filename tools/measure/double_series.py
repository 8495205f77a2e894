"""C3 with the series as float64 - what every fabber_dorun caller hands over (NEWMAT::Matrix is double) - against
float32 (the C ABI's image type): kernel time with the series resident."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import numpy as np, torch
import cases
from fabber_core_amd.device import DeviceProblem
h, y = cases.exp_problem(1000000, 100, 2, 0.02, seed=20260103, max_iterations=50)
for dt in (np.float32, np.float64):
    prob = DeviceProblem(h, y.astype(dt), "cuda:0")
    prob.run(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        prob.run()
    e1.record(); torch.cuda.synchronize()
    print(np.dtype(dt).name, prob.kernel, "%.2f ms" % (e0.elapsed_time(e1) / 5), flush=True)
