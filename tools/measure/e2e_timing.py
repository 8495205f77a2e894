import os, sys, json
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
os.environ["FVB_HOST_TIMING"] = "1"
import bench, cases
w = bench.WORKLOADS["c3"]
V = w["voxels"]
holder, y = bench.make_problem(w, V, 20260103, False)
out = bench.bench_boundary(w, V, holder, y, False)
print(json.dumps(out["fabber_capi_ms"]))
print(json.dumps(out["fabber_vb_run_host_ms"]))
