#!/usr/bin/env python3
"""Rounding error of every fp64 implementation of the bi-exponential fit against the binary128
ground truth (tests/golden/c3_truth_binary128.npz, see tests/golden/make_c3_truth.py).

    python tools/measure/c3_truth.py [--out profiles/r2_c3_truth.json]

Prints, for the CPU oracle, its FMA build and the HIP kernels (lane and wave mapping), the
distribution of |posterior mean - truth| per voxel: after 1, 2, 3, 5, 10, 20, 35, 50 iterations
(relative error of the parameter means, SURVEY 8d metric) and for the final posterior (scaled and
relative metrics, share within 1e-4 / 1e-6, quantiles, failed voxels).
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=None)
    ap.add_argument("--no-gpu", action="store_true")
    ap.add_argument("--no-cpu", action="store_true", help="the HIP kernels only (FVB_LIB_PATH chooses an experiment build)")
    a = ap.parse_args()
    import make_c3_truth as mt
    import oracle
    import parity
    truth = parity.load_c3_truth()
    V = truth["n_voxels"]
    h, y = mt.problem(V)
    engines = {} if a.no_cpu else {"cpu": lambda hh: oracle.run(hh, y), "cpu_fma": lambda hh: oracle.run_fma(hh, y)}
    if not a.no_gpu:
        from fabber_core_amd import hiplib

        def gpu(variant):
            def f(hh):
                hiplib.set_variant(variant)
                try:
                    return hiplib.run_host(hh, y)
                finally:
                    hiplib.set_variant("auto")
            return f
        engines["hip_lane"] = gpu("lane")
        engines["hip_wave"] = gpu("wave")
    report = {"n_voxels": V, "engines": {}}
    for name, run in engines.items():
        final = run(h)
        if not np.any(final["status"] == 0):
            print(name, "every voxel failed")
            report["engines"][name] = {"final": {"failed": 1.0}}
            continue
        rep = {"final": parity.truth_stats(h, truth, final), "by_iteration": {}}
        for k, it in enumerate(truth["its"]):
            hk, _ = mt.problem(V)
            hk.cfg.max_iterations = int(it)
            r = run(hk)
            rep["by_iteration"][int(it)] = parity.truth_trace_stats(hk, truth["trace_means"][k], r)
        report["engines"][name] = rep
        print(name, json.dumps(rep["final"]))
        for it, s in rep["by_iteration"].items():
            print("   it %2d  median %.2e  p90 %.2e  within 1e-4 %.3f" % (it, s["median"], s["p90"], s["within_1e4"]))
    if a.out:
        with open(a.out, "w") as fh:
            json.dump(report, fh, indent=1)


if __name__ == "__main__":
    main()
