"""Lane kernel with per-precision moments (vb_lane_pattern_kernel.h) against the wave kernel on noise-pattern
problems: kernel time (events on the launch stream, series resident) at volume sizes.
    python tools/measure/bench_patterns.py > profiles/r2_lane_vs_wave_patterns.jsonl"""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import cases  # noqa: E402
from fabber_core_amd import device, hiplib  # noqa: E402


def timed(h, y, variant, reps=3):
    hiplib.set_variant(variant)
    prob = device.DeviceProblem(h, y, "cuda:0")
    name = prob.kernel
    prob.run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        prob.run()
    e1.record()
    torch.cuda.synchronize()
    out = prob.results()
    hiplib.set_variant("auto")
    return name, e0.elapsed_time(e1) / reps, out


def main():
    runs = [
        ("single exponential, pattern 12", lambda V: cases.exp_problem(V, 50, 1, 0.04, seed=2, max_iterations=10, noise_pattern="12")),
        ("bi-exponential (C3 shape), pattern 12", lambda V: cases.exp_problem(V, 100, 2, 0.02, seed=3, max_iterations=10, noise_pattern="12")),
        ("linear 4 regressors T=200 (C4 design), pattern 1234", lambda V: cases.linear_problem(V, 200, seed=4, max_iterations=10, noise_pattern="1234")),
    ]
    for what, make in runs:
        for V in (16384, 262144):
            h, y = make(V)
            ln, lms, lo = timed(h, y, "lane")
            wn, wms, wo = timed(h, y, "wave")
            n = h.cfg.n_params + h.cfg.n_phis
            m0 = lo["mvn"][n * (n + 1) // 2:][:n]
            m1 = wo["mvn"][n * (n + 1) // 2:][:n]
            good = (lo["status"] == 0) & (wo["status"] == 0)
            rel = np.abs(m0 - m1)[:, good] / np.maximum(np.abs(m1[:, good]), 1e-3)
            print(json.dumps({"workload": what, "voxels": V, "lane_kernel": ln, "lane_ms": lms, "wave_kernel": wn, "wave_ms": wms,
                              "median_rel_diff_means": float(np.median(rel.max(axis=0))), "bad_lane": int((lo["status"] != 0).sum()),
                              "bad_wave": int((wo["status"] != 0).sum())}), flush=True)


if __name__ == "__main__":
    main()
