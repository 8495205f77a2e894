import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import numpy as np
import cases, oracle, parity, hipengine
h, y = cases.exp_problem(20000, 50, 1, 0.04, seed=20260102, max_iterations=10)
cpu, cpu2 = oracle.run(h, y), oracle.run_fma(h, y)
gpu = hipengine.run(h, y)
ok = (cpu["status"] == 0) & (gpu["status"] == 0)
e, _, _ = parity.voxel_errors(h, cpu, gpu, ok)
f, _, _ = parity.voxel_errors(h, cpu, cpu2, ok)
print("passes", os.environ.get("FVB_PRECISE_PASSES", "default"), "GPU vs CPU: median %.2e p99 %.2e max %.2e | CPU vs CPU-FMA: median %.2e p99 %.2e max %.2e"
      % (np.median(e), np.quantile(e, 0.99), e.max(), np.median(f), np.quantile(f, 0.99), f.max()))
