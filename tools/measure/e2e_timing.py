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
