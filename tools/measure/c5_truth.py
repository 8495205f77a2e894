#!/usr/bin/env python3
"""Rounding error of every fp64 implementation of the spatial bi-exponential fit against the binary128 ground truth
(tests/golden/c5_truth_binary128.npz, see tests/golden/make_c5_truth.py): the CPU oracle, its FMA build and the HIP
spatial path, for the final posterior (10 iterations) and after 1, 2, 3, 5 iterations.

    [FVB_PRECISE_PASSES=n] python tools/measure/c5_truth.py [--out profiles/r3_c5_truth.json] [--no-cpu]
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
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--residual", default="auto", choices=["auto", "exact", "moments"], help="how the kernels obtain k'Qk")
    ap.add_argument("--residual-tol", type=float, default=None, help="auto: fall back to the direct sum below this share of the terms' magnitude")
    a = ap.parse_args()
    import make_c5_truth as mt
    import oracle
    import parity
    truth, (h, sp, y) = parity.load_c5_truth()
    engines = {}
    if not a.no_cpu:
        engines["cpu"] = lambda hh: oracle.run_spatial(hh, sp, y)
        engines["cpu_fma"] = lambda hh: oracle.run_spatial_fma(hh, sp, y)
    if not a.no_gpu:
        from fabber_core_amd import hiplib
        hiplib.set_residual_mode(a.residual)
        if a.residual_tol is not None:
            hiplib.set_residual_tolerance(a.residual_tol)
        engines["hip"] = lambda hh: hiplib.run_spatial_host(hh, sp, y)
    report = {"precise_passes": os.environ.get("FVB_PRECISE_PASSES", "default (2)"), "residual": a.residual,
              "residual_tol": a.residual_tol, "engines": {}}
    for name, run in engines.items():
        rep = {"final": parity.truth_stats(h, truth, run(h), with_f=True), "by_iteration": {}}
        for k, it in enumerate(truth["its"]):
            hk, _, _ = mt.problem(max_iterations=int(it))
            rep["by_iteration"][int(it)] = parity.truth_trace_stats(hk, truth["trace_means"][k], run(hk))
        report["engines"][name] = rep
        print(name, "final median %.3e p75 %.3e p90 %.3e | by iteration (median): %s" % (
            rep["final"]["median"], rep["final"]["p75"], rep["final"]["p90"],
            {it: "%.3e" % v["median"] for it, v in rep["by_iteration"].items()}), flush=True)
    if a.out:
        json.dump(report, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
