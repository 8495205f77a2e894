import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from fabber_core_amd import vbabi, hiplib
from fabber_core_amd.device import DeviceProblem
n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
its = int(sys.argv[2]) if len(sys.argv) > 2 else 10
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 2
shape = (n, n, n)
coords = vbabi.grid_coords(shape)
V = coords.shape[1]
rng = np.random.default_rng(0)
T = 100
t = np.arange(T) * 0.02
amp1 = (0.75 + 0.25 * np.sin(coords[0] / 8.0) * np.cos(coords[2] / 6.0)).astype(np.float32)
y = np.empty((T, V), dtype=np.float32)
for i in range(T):
    y[i] = amp1 * np.float32(np.exp(-1.0 * t[i])) + np.float32(0.5 * np.exp(-6.0 * t[i])) + rng.standard_normal(V, dtype=np.float32) * np.float32(0.1)
h = vbabi.build_config(vbabi.MODEL_EXP, V, T, num_exps=2, dt=0.02, max_iterations=its, param_overrides={"amp1": dict(type="M")})
sp = vbabi.SpatialHolder(coords)
prob = DeviceProblem(h, y, "cuda:0")
for r in range(reps):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    prob.run_spatial(sp)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("rep %d: %d voxels, %d its: %.1f ms total -> %.2f Mvox/s (%.2f ms/iteration)" % (r, V, its, dt * 1e3, V / dt / 1e6, dt * 1e3 / its), flush=True)
res = prob.results()
print("bad", int((res["status"] != 0).sum()), "amp1 mean", float(np.exp(res["mvn"][15]).mean()) if False else "")
