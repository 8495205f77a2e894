#!/bin/bash
# Round-3 profiles (run on the GPU box): PMC passes of the bench command for C3 (with and without F) and C5 (prep /
# slab sweep / second-sweep kernels), the rocprofv3 kernel summaries, and the bench lines. Summaries ->
# gpurun_out/r3p/, to be copied into profiles/ (see profiles/README.md).
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r3p
mkdir -p $OUT
cd $ROOT
for w in c3 c3_needf c5; do
    case $w in
        c3) A="--no-e2e";;
        c3_needf) A="--need-f --no-e2e";;
        c5) A="--workload c5";;
    esac
    BENCH_ARGS="$A" bash tools/pmc_passes.sh r3$w > $OUT/pmc_$w.log 2>&1
    if [ $w = c5 ]; then
        for k in vb_spatial_prep_kernel vb_spatial_slab_sweep_kernel vb_spatial_setup_kernel; do
            python3 tools/pmc_summary.py gpurun_out/pmc_r3$w $k > $OUT/r3_pmc_c5_$k.json
        done
        # the second sweep's two instances: the streaming one (9 of 10 iterations) and the one with the half-ulp exp
        python3 tools/pmc_summary.py gpurun_out/pmc_r3$w "4, false, true, false>" > $OUT/r3_pmc_c5_vb_spatial_noise_kernel.json
        python3 tools/pmc_summary.py gpurun_out/pmc_r3$w "4, false, true, true>" > $OUT/r3_pmc_c5_vb_spatial_noise_kernel_acc.json
        python3 tools/pmc_c5_merge.py $OUT/r3_pmc_c5_vb_*.json > $OUT/r3_pmc_c5.json
    else
        python3 tools/pmc_summary.py gpurun_out/pmc_r3$w vb_lane > $OUT/r3_pmc_$w.json
    fi
    cp gpurun_out/pmc_r3$w/trace/*/*kernel_stats.csv $OUT/r3_kernel_stats_$w.csv
    echo "[r3_profiles] $w done"
done
python3 bench.py --steps 20 --warmup 5 > $OUT/r3_bench_c3.json 2> $OUT/bench_c3.err
python3 bench.py --steps 10 --need-f --cpu-sample 0 --no-e2e > $OUT/r3_bench_c3_needf.json 2>/dev/null
python3 bench.py --steps 20 --workload c2 --no-e2e > $OUT/r3_bench_c2.json 2>/dev/null
python3 bench.py --steps 10 --workload c4 --no-e2e > $OUT/r3_bench_c4.json 2>/dev/null
python3 bench.py --steps 5 --workload c5 > $OUT/r3_bench_c5.json 2>/dev/null
python3 bench.py --steps 5 --workload c1 --no-e2e > $OUT/r3_bench_c1.json 2>/dev/null
echo "[r3_profiles] all done"
