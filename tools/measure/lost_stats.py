import sys, os, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import cases
from fabber_core_amd import hiplib
from fabber_core_amd.device import DeviceProblem
hiplib.set_variant("lane")
for tol in (1e-10, 1e-12, 1e-14):
    hiplib.set_residual_tolerance(tol)
    h, y = cases.exp_problem(262144, 100, 2, 0.02, seed=20260103, max_iterations=50)
    prob = DeviceProblem(h, y, "cuda:0")
    prob.iterations.zero_()
    prob.run(); torch.cuda.synchronize()
    raw = prob.iterations.cpu().numpy().astype(np.int64) & 0xFFFFFFFF
    its = raw & 0xFF; lost = (raw >> 8) & 0xFFF; wave = (raw >> 20) & 0xFFF
    mvn = prob.mvn.cpu().numpy()
    print("tol %g: mean its %.2f; voxel-iterations lost %.4f; wave-iterations taking the exact pass %.4f" % (tol, its.mean(), lost.sum() / its.sum(), wave.sum() / its.sum()))
    print("   voxels never lost %.4f; lost >= 10 times %.4f; lost >= 40 times %.4f" % ((lost == 0).mean(), (lost >= 10).mean(), (lost >= 40).mean()))
    if tol == 1e-10:
        idx = np.flatnonzero(lost >= 40)[:5]
        n = 5; ncov = n*(n+1)//2
        print("   persistent-lost voxels means (Fabber space):", np.round(mvn[ncov:ncov+4, idx].T, 4).tolist())
        idx = np.flatnonzero(lost == 0)[:3]
        print("   never-lost voxels means:", np.round(mvn[ncov:ncov+4, idx].T, 4).tolist())
