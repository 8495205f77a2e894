"""CPU experiment: the oracle with the exponential model's difference quotient formed without the
cancellation of the unperturbed terms (ORACLE_STRUCTURED_J=1) against the reference's arithmetic:
fraction of voxels of the C3 problem that end with a non-finite prediction. ORACLE_SWEEP_INVERSE=1 swaps the
oracle's LU inverse for the kernels' symmetric sweep (the other factor that decides that fraction)."""
import sys, os, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import cases, oracle
V = 20000
h, y = cases.exp_problem(V, 100, 2, 0.02, seed=20260103, max_iterations=50)
a = oracle.run(h, y)
bad = a["status"] != 0
print(json.dumps({"structured": os.environ.get("ORACLE_STRUCTURED_J"), "bad": float(bad.mean()), "hist": np.bincount(a["status"], minlength=4).tolist(),
                  "its": np.bincount(a["iterations"][bad], minlength=30).tolist()[:30]}))
