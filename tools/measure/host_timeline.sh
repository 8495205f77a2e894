#!/bin/bash
# Timeline of ONE fabber_vb_run_host call on C3 (kernels, copies, HIP runtime calls): what overlaps and what waits.
#   bash tools/measure/host_timeline.sh [block_voxels]     -> gpurun_out/host_timeline/ + a merged listing on stdout
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/host_timeline
rm -rf $OUT
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# PROG=e2e: the reference's C ABI (fabber_new .. fabber_destroy, tools/measure/e2e_timing.py) instead of the bare engine call
if [ "$PROG" = e2e ]; then
    timeout -k 10 300 rocprofv3 --kernel-trace --memory-copy-trace --hip-runtime-trace --output-format csv -d $OUT -- python3 $ROOT/tools/measure/e2e_timing.py > $OUT.log 2>&1 < /dev/null
else
    timeout -k 10 300 rocprofv3 --kernel-trace --memory-copy-trace --hip-runtime-trace --output-format csv -d $OUT -- python3 $ROOT/tools/measure/host_pipeline.py --once ${1:-262144} ${PIN:+--pinned} > $OUT.log 2>&1 < /dev/null
fi
tail -3 $OUT.log
python3 - "$OUT" <<'PY'
import csv, glob, sys
root = sys.argv[1]
ev = []
for f in glob.glob(root + "/*/*_kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "K", r["Kernel_Name"][:60]))
for f in glob.glob(root + "/*/*_memory_copy_trace.csv"):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "C", "%s %s bytes" % (r.get("Direction", ""), r.get("Bytes", r.get("Size", "")))))
for f in glob.glob(root + "/*/*_hip_api_trace.csv"):
    for r in csv.DictReader(open(f)):
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        if e - s > 30000:
            ev.append((s, e, "A", r["Function"] + " tid " + r.get("Thread_Id", "")))
ev.sort()
# the last call: from the last retile/ first upload after the previous call's last download ... take the last 45 ms
if not ev:
    sys.exit("no events")
t_end = max(e for _, e, _, _ in ev)
sel = [x for x in ev if x[0] > t_end - float(__import__('os').environ.get('WINDOW_MS', '45')) * 1e6]
t0 = sel[0][0]
for s, e, k, name in sel:
    print("%8.3f %8.3f %7.3f %s %s" % ((s - t0) / 1e6, (e - t0) / 1e6, (e - s) / 1e6, k, name))
PY
