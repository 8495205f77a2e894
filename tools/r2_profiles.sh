#!/bin/bash
# Round-2 profiles (run on the GPU box): PMC passes of the bench command for C3 (with and without F), C2, C4,
# the rocprofv3 kernel summary of C5, and the bench lines themselves. Summaries -> gpurun_out/r2p/, to be
# copied into profiles/ (see profiles/README.md).
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r2p
mkdir -p $OUT
cd $ROOT
for w in c3 c3_needf c2 c4; do
    case $w in
        c3) A="";;
        c3_needf) A="--need-f";;
        c2) A="--workload c2";;
        c4) A="--workload c4";;
    esac
    BENCH_ARGS="$A" bash tools/pmc_passes.sh r2$w > $OUT/pmc_$w.log 2>&1
    python3 tools/pmc_summary.py gpurun_out/pmc_r2$w vb_lane > $OUT/r2_pmc_$w.json
    cp gpurun_out/pmc_r2$w/trace/*/*kernel_stats.csv $OUT/r2_kernel_stats_$w.csv
    echo "[r2_profiles] $w done"
done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_c5 -- python3 $ROOT/bench.py --workload c5 --steps 3 --warmup 1 --cpu-sample 0 > $OUT/prof_c5.log 2>&1
cp $OUT/prof_c5/*/*kernel_stats.csv $OUT/r2_kernel_stats_c5.csv
cd $ROOT
python3 bench.py --steps 20 --warmup 5 > $OUT/r2_bench_c3.json 2> $OUT/bench_c3.err
python3 bench.py --steps 10 --need-f --cpu-sample 0 > $OUT/r2_bench_c3_needf.json 2>/dev/null
python3 bench.py --steps 20 --workload c2 > $OUT/r2_bench_c2.json 2>/dev/null
python3 bench.py --steps 10 --workload c4 > $OUT/r2_bench_c4.json 2>/dev/null
python3 bench.py --steps 5 --workload c5 > $OUT/r2_bench_c5.json 2>/dev/null
python3 bench.py --steps 5 --workload c1 > $OUT/r2_bench_c1.json 2>/dev/null
python3 bench.py --steps 3 --gpus 1 --single-process --cpu-sample 0 > $OUT/r2_bench_c3_single_process.json 2>/dev/null
echo "[r2_profiles] all done"
