"""lane-per-voxel vs wave-per-voxel for 7 and 8 parameters (vb_lane_wide.hip): design matrices with 7 / 8
cosine regressors (T = 100, 10 iterations) and four exponentials (T = 100, 10 iterations)."""
import sys, os, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import cases
from fabber_core_amd import hiplib, vbabi
from fabber_core_amd.device import DeviceProblem


def linear(V, P, T=100):
    rng = np.random.default_rng(8)
    t = (np.arange(T) + 0.5) / T
    X = np.stack([np.cos(np.pi * k * t) for k in range(P)], axis=1)
    y = (X @ rng.uniform(-2, 2, (P, V)) + rng.normal(0, 0.05, (T, V))).astype(np.float32)
    return vbabi.build_config(vbabi.MODEL_LINEAR, V, T, design=X, max_iterations=10), y


def exp4(V, T=100):
    rng = np.random.default_rng(9)
    t = np.arange(T) * 0.02
    y = sum(a * np.exp(-r * t[:, None]) for a, r in ((1.0, 0.5), (0.7, 2.0), (0.5, 6.0), (0.3, 15.0))) + rng.normal(0, 0.05, (T, V))
    return vbabi.build_config(vbabi.MODEL_EXP, V, T, num_exps=4, dt=0.02, max_iterations=10), y.astype(np.float32)


for name, mk in (("linear P=7", lambda V: linear(V, 7)), ("linear P=8", lambda V: linear(V, 8)), ("exp x4 (P=8)", exp4)):
    for V in (16384, 262144):
        h, y = mk(V)
        rec = {"workload": name, "voxels": V}
        res = {}
        for variant in ("lane", "wave"):
            hiplib.set_variant(variant)
            prob = DeviceProblem(h, y, "cuda:0")
            rec[variant + "_kernel"] = prob.kernel
            prob.run(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3):
                prob.run()
            e1.record(); torch.cuda.synchronize()
            rec[variant + "_ms"] = e0.elapsed_time(e1) / 3
            res[variant] = prob.results()
        ok = (res["lane"]["status"] == 0) & (res["wave"]["status"] == 0)
        n = h.cfg.n_params
        off = (n + 1) * (n + 2) // 2
        m1, m2 = res["lane"]["mvn"][off:off + n][:, ok], res["wave"]["mvn"][off:off + n][:, ok]
        rec["max_rel_diff_means"] = float(np.max(np.abs(m1 - m2) / np.maximum(np.abs(m2), 1e-3)))
        rec["bad_lane"], rec["bad_wave"] = int((res["lane"]["status"] != 0).sum()), int((res["wave"]["status"] != 0).sum())
        print(json.dumps(rec), flush=True)
