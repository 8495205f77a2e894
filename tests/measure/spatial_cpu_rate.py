import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle
from fabber_core_amd import vbabi
shape = (24, 24, 24)
coords = vbabi.grid_coords(shape); V = coords.shape[1]
rng = np.random.default_rng(0); T = 100; t = np.arange(T) * 0.02
amp1 = 0.75 + 0.25 * np.sin(coords[0] / 8.0) * np.cos(coords[2] / 6.0)
y = amp1[None, :] * np.exp(-t[:, None]) + 0.5 * np.exp(-6 * t[:, None]) + rng.normal(0, 0.1, (T, V))
h = vbabi.build_config(vbabi.MODEL_EXP, V, T, num_exps=2, dt=0.02, max_iterations=10, param_overrides={"amp1": dict(type="M")})
t0 = time.perf_counter(); oracle.run_spatial(h, vbabi.SpatialHolder(coords), y); dt = time.perf_counter() - t0
print("CPU oracle spatial: %d voxels, 10 iterations: %.2f s -> %.0f voxels/s" % (V, dt, V / dt))
