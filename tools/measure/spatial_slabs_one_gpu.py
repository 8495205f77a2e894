"""fabber_vb_run_spatial_host_multi rehearsed on one GPU: the device listed several times, against the one-device
run (must be identical), with where a difference starts if there is one."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import numpy as np
import cases
from fabber_core_amd import hiplib, vbabi
n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
its = int(sys.argv[2]) if len(sys.argv) > 2 else 10
h, coords, y, _ = cases.c5_problem((n, n, n), max_iterations=its)
sp = vbabi.SpatialHolder(coords)
ref = hiplib.run_spatial_host(h, sp, y)
t0 = time.perf_counter()
ref = hiplib.run_spatial_host(h, sp, y)
print("1 device: %.1f ms (the same Python call: result arrays allocated and filled inside it)" % ((time.perf_counter() - t0) * 1e3), flush=True)
for devs in ([0, 0], [0, 0, 0], [0, 0, 0, 0], [0] * 8):
    hiplib.run_spatial_host(h, sp, y, devices=devs)
    t0 = time.perf_counter()
    r = hiplib.run_spatial_host(h, sp, y, devices=devs)
    ms = (time.perf_counter() - t0) * 1e3
    same = np.array_equal(ref["mvn"], r["mvn"], equal_nan=True)
    print(len(devs), "slabs: %.1f ms" % ms, "identical" if same else "DIFFERENT", flush=True)
    if not same:
        bad = np.flatnonzero(np.any(ref["mvn"] != r["mvn"], axis=0))
        z = coords[2][bad]
        lev = coords[0] + coords[1] + coords[2]
        zb = np.unique(z)[0]
        on = bad[z == zb]
        print("   plane %d: %d differ, levels %d..%d; voxels of that plane with level >= %d: %d" % (
            zb, len(on), lev[on].min(), lev[on].max(), lev[on].min(), np.count_nonzero((coords[2] == zb) & (lev >= lev[on].min()))))
        print("   %d voxels differ, z planes %s, first voxel %d (x %d y %d z %d), max |d| %.3g" % (
            len(bad), np.unique(z)[:12], bad[0], coords[0][bad[0]], coords[1][bad[0]], coords[2][bad[0]],
            np.nanmax(np.abs(ref["mvn"][:, bad] - r["mvn"][:, bad]))))
