"""Spatial VB (set-up on a stream of its own beside the run's stream; the multi-slab form with all slabs on one device)
under the SYSTEM's HIP runtime against the same calls under the runtime PyTorch bundles: results written by one process,
compared bit for bit by the other.   WITH_TORCH=1 python ... write <file>;  python ... check <file>"""
import os, sys
if os.environ.get("WITH_TORCH") != "1":
    os.environ["FVB_NO_TORCH"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import cases
from fabber_core_amd import hiplib, vbabi
mode, path = sys.argv[1], sys.argv[2]
n = 48
holder, coords, y, _ = cases.c5_problem((n, n, n), max_iterations=6, need_f=True)
sp = vbabi.SpatialHolder(coords)
out = {}
for rep in range(3):
    r = hiplib.run_spatial_host(holder, sp, y)
    out["one_%d" % rep] = r["mvn"]
    out["one_F_%d" % rep] = r["free_energy"]
for rep in range(2):
    r = hiplib.run_spatial_host(holder, sp, y, devices=[0, 0, 0])
    out["slabs_%d" % rep] = r["mvn"]
if mode == "write":
    np.savez(path, **out)
    print("written", sorted(out))
else:
    ref = np.load(path)
    for k in sorted(out):
        print(k, "identical:", bool(np.array_equal(ref[k], out[k], equal_nan=True)), "max |d|:", float(np.nanmax(np.abs(ref[k] - out[k]))))
print({l.split()[-1] for l in open("/proc/self/maps") if "amdhip64" in l})
