#!/bin/bash
# PMC passes for the roofline `traffic` figure and the issue statistics (run on the GPU box):
#   bash tools/pmc_passes.sh <tag>            -> gpurun_out/pmc_<tag>/{fetch,write,sq,fp64,mem,calib_fetch,calib_write,trace}
# One counter group per pass, kernel-trace only, the program directly after `--`
# (MI355X_MICROARCH.md: rocprofv3 PMC slots; FETCH_SIZE and WRITE_SIZE do not fit in one pass).
set -e
TAG=${1:-run}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $ROOT/bench.py --steps ${STEPS:-20} --warmup ${WARMUP:-2} --cpu-sample 0 ${BENCH_ARGS}"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $BENCH > $OUT/trace.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- $BENCH > $OUT/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -- $BENCH > $OUT/write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $OUT/sq -- $BENCH > $OUT/sq.log 2>&1
# executed fp64 operations by kind (for FLOP/s = (2 FMA + ADD + MUL) x 64 / time) and the shader clock the chip
# held during the kernel: GRBM_GUI_ACTIVE is summed over the 8 XCDs -> clock = value / 8 / kernel time
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --output-format csv -d $OUT/fp64 -- $BENCH > $OUT/fp64.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_VMEM SQ_IFETCH --output-format csv -d $OUT/mem -- $BENCH > $OUT/mem.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/calib_fetch -- $ROOT/tools/pmc_calib > $OUT/calib_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/calib_write -- $ROOT/tools/pmc_calib > $OUT/calib_write.log 2>&1
find $OUT -name "*.csv" | head -30
