"""The reference's C ABI on C3 (fabber_new .. fabber_destroy, bench.py's e2e leg) with FVB_HOST_TIMING=1: the host side's stage timers on stderr."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["FVB_HOST_TIMING"] = "1"
import bench, cases
w = bench.WORKLOADS["c3"]
V = w["voxels"]
holder, y = bench.make_problem(w, V, 20260103, False)
out = bench.bench_boundary(w, V, holder, y, False)
print(json.dumps(out["fabber_capi_ms"]))
print(json.dumps(out["fabber_vb_run_host_ms"]))

# the bare engine call again, after the C ABI runs and after 20 s of idling (what bench.py's CPU baseline leaves the GPU in)
import time
import numpy as np
from fabber_core_amd import hiplib
os.environ.pop("FVB_HOST_TIMING", None)
res = hiplib.run_host(holder, y)


def timed(n=5):
    ts = []
    for _ in range(n):
        t0 = time.perf_counter()
        hiplib.run_host(holder, y, into=res)
        ts.append((time.perf_counter() - t0) * 1e3)
    return [round(t, 2) for t in ts]


print("after the C ABI runs:", timed())
time.sleep(20)
print("after 20 s idle:", timed())
import torch
x = torch.zeros(1 << 28, device="cuda")
torch.cuda.synchronize()
print("with torch's context and 1 GB allocated:", timed())
