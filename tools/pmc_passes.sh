#!/bin/bash
# PMC passes for the roofline `traffic` figure and the issue statistics (run on the GPU box):
#   bash tools/pmc_passes.sh <tag>            -> gpurun_out/pmc_<tag>/{fetch,write,sq,calib_fetch,calib_write,trace}
# One counter group per pass, kernel-trace only, the program directly after `--`
# (MI355X_MICROARCH.md: rocprofv3 PMC slots; FETCH_SIZE and WRITE_SIZE do not fit in one pass).
set -e
TAG=${1:-run}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $ROOT/bench.py --steps 3 --warmup 1 --cpu-sample 0 ${BENCH_ARGS}"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $BENCH > $OUT/trace.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- $BENCH > $OUT/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -- $BENCH > $OUT/write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES --output-format csv -d $OUT/sq -- $BENCH > $OUT/sq.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/calib_fetch -- $ROOT/tools/pmc_calib > $OUT/calib_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/calib_write -- $ROOT/tools/pmc_calib > $OUT/calib_write.log 2>&1
find $OUT -name "*.csv" | head -30
