import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from fabber_core_amd import fabber
rng = np.random.default_rng(0)
shape, T = (64, 64, 16), 50
t = np.arange(T) * 0.04
data = (np.exp(-t) + rng.normal(0, 0.1, shape + (T,))).astype(np.float32)
opts = {"model": "exp", "num-exps": 1, "dt": 0.04, "noise": "white", "method": "vb", "max-iterations": 10, "save-mean": True}
V = np.prod(shape)
for label, extra in (("device model", {}), ("host model, 1 thread", {"host-model": True, "host-model-threads": 1}),
                     ("host model, 16 threads", {"host-model": True, "host-model-threads": 16})):
    t0 = time.perf_counter(); out = fabber.run(data, dict(opts, **extra)); dt = time.perf_counter() - t0
    print("%-26s %7.2f s  %9.0f voxels/s  mean amp %.4f" % (label, dt, V / dt, out["mean_amp1"].mean()), flush=True)
