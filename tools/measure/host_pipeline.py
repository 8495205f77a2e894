#!/usr/bin/env python3
"""fabber_vb_run_host on C3 (1e6 voxels, host pointers in and out) as a function of the block size of its pipeline
(FVB_HOST_BLOCK_VOXELS; 0 = one block, no overlap), best and mean of a few calls.

    python tools/measure/host_pipeline.py [--blocks 0,65536,131072,262144,524288] [--calls 5] [--once BLOCK]

--once BLOCK: a single warm call + one measured call at that block size (for a rocprofv3 --kernel-trace
--memory-copy-trace run: the copies' and kernels' timestamps show what overlaps)."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--blocks", default="0,65536,131072,262144,524288")
    ap.add_argument("--calls", type=int, default=5)
    ap.add_argument("--voxels", type=int, default=1_000_000)
    ap.add_argument("--once", type=int, default=None)
    ap.add_argument("--pinned", action="store_true", help="pin the caller's buffers first (fabber_vb_pin_host_buffer)")
    ap.add_argument("--schedules", default="", help="semicolon-separated FVB_HOST_BLOCK_SCHEDULE values to try after the block sizes")
    a = ap.parse_args()
    import cases
    from fabber_core_amd import hiplib
    h, y = cases.exp_problem(a.voxels, 100, 2, 0.02, seed=20260103, max_iterations=50)
    res = hiplib.run_host(h, y)
    if a.pinned:
        import ctypes as C
        L = hiplib.lib()
        L.fabber_vb_pin_host_buffer.restype = C.c_int32
        L.fabber_vb_pin_host_buffer.argtypes = [C.c_void_p, C.c_size_t]
        y = np.ascontiguousarray(y)
        for arr in [y] + [v for v in res.values() if isinstance(v, np.ndarray)]:
            print("pin", arr.nbytes, L.fabber_vb_pin_host_buffer(arr.ctypes.data, arr.nbytes))
    blocks = [a.once] if a.once is not None else [int(b) for b in a.blocks.split(",")]
    out = {}
    for b in blocks:
        os.environ["FVB_HOST_BLOCK_VOXELS"] = str(b)
        hiplib.run_host(h, y, into=res)
        ts = []
        for _ in range(1 if a.once is not None else a.calls):
            t0 = time.perf_counter()
            hiplib.run_host(h, y, into=res)
            ts.append((time.perf_counter() - t0) * 1e3)
        out[b] = {"min_ms": float(np.min(ts)), "mean_ms": float(np.mean(ts))}
        print(b, out[b], flush=True)
    for sched in [x for x in a.schedules.split(";") if x]:
        os.environ["FVB_HOST_BLOCK_SCHEDULE"] = sched
        hiplib.run_host(h, y, into=res)
        ts = []
        for _ in range(a.calls):
            t0 = time.perf_counter()
            hiplib.run_host(h, y, into=res)
            ts.append((time.perf_counter() - t0) * 1e3)
        out[sched] = {"min_ms": float(np.min(ts)), "mean_ms": float(np.mean(ts))}
        print("schedule", sched, out[sched], flush=True)
    os.environ.pop("FVB_HOST_BLOCK_SCHEDULE", None)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
