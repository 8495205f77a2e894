#!/bin/bash
# quick look: kernel time + FETCH/WRITE per launch for the bench workload
set -e
TAG=${1:-q}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmcq_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $ROOT/bench.py --steps 3 --warmup 1 --cpu-sample 0 ${BENCH_ARGS}"
$BENCH > $OUT/bench.json 2> $OUT/bench.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- $BENCH > $OUT/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -- $BENCH > $OUT/write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES --output-format csv -d $OUT/sq -- $BENCH > $OUT/sq.log 2>&1
python3 $ROOT/tools/pmc_summary.py $OUT ${KMATCH:-vb_lane_kernel} > $OUT/summary.json
python3 - <<PY
import json
b = json.loads(open("$OUT/bench.json").read().strip().splitlines()[-1])
s = json.load(open("$OUT/summary.json"))
print("ms_per_step", b["ms_per_step"], "kernel_ms", b["roofline"]["kernel_ms"])
for k, v in s["counters"].items(): print(k, "%.4g" % v["mean_per_launch"])
PY
