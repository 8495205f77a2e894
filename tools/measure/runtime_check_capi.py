"""The C ABI run repeatedly in one process under the SYSTEM's HIP runtime (FVB_NO_TORCH=1): fabber_new .. fabber_destroy with
the host library's buffer cache handing the same blocks out again."""
import os, sys
if os.environ.get("WITH_TORCH") != "1":
    os.environ["FVB_NO_TORCH"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import cases
from fabber_core_amd import fabber
V = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
h, y = cases.exp_problem(V, 100, 2, 0.02, seed=20260103, max_iterations=50)
data = np.ascontiguousarray(y.T.reshape(V, 1, 1, 100))
opts = {"model": "exp", "num-exps": 2, "dt": 0.02, "max-iterations": 50, "noise": "white", "method": "vb", "save-mean": True,
        "save-mvn": True, "allow-bad-voxels": True}
first = None
for rep in range(5):
    try:
        out = fabber.run(data, opts)
    except Exception as e:
        print(rep, "FAILED:", str(e)[:200])
        continue
    mvn = out["finalMVN"]
    if first is None:
        first = mvn
    print(rep, "ok; identical to the first run:", bool(np.array_equal(first, mvn, equal_nan=True)), "non-finite entries:", int((~np.isfinite(mvn)).sum()),
          [l for l in out["log"].splitlines() if "numerical errors" in l][-1:])
