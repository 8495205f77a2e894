#!/bin/bash
# instruction mix and wait breakdown of the VB kernel (two PMC passes):
#   bash tools/pmc_mix.sh <tag>   -> gpurun_out/pmcm_<tag>/{mix,wait}
set -e
TAG=${1:-m}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmcm_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $ROOT/bench.py --steps 3 --warmup 1 --cpu-sample 0 ${BENCH_ARGS}"
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_SALU SQ_INSTS_FLAT --output-format csv -d $OUT/mix -- $BENCH > $OUT/mix.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM --output-format csv -d $OUT/wait -- $BENCH > $OUT/wait.log 2>&1
python3 $ROOT/tools/pmc_summary.py $OUT ${KMATCH:-vb_lane_kernel} > $OUT/summary.json || true
python3 - <<PY
import json
s = json.load(open("$OUT/summary.json"))
for k, v in s["counters"].items(): print(k, "%.4g" % v["mean_per_launch"])
PY
