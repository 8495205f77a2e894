#!/bin/bash
# Round-4 profiles (run on the GPU box, on the build that is committed): for C3, C3 with F and C5, the rocprofv3 kernel
# summary and the counter passes of the bench command - >= 20 launches of every kernel profiled (STEPS=20: the round-3
# verdict's "PMC profile of 5 launches") - then the bench lines of all workloads, written AFTER the profiles are in place
# so that each line's `traffic` and the committed counter file are the same numbers.
#   bash tools/r4_profiles.sh            -> gpurun_out/r4p/   (copy into profiles/, see profiles/README.md)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r4p
mkdir -p $OUT
cd $ROOT
for w in ${WORKLOADS:-c3 c3_needf c5}; do
    case $w in
        c3) A="--no-e2e";;
        c3_needf) A="--need-f --no-e2e";;
        c5) A="--workload c5";;
    esac
    STEPS=20 BENCH_ARGS="$A" bash tools/pmc_passes.sh r4$w > $OUT/pmc_$w.log 2>&1
    if [ $w = c5 ]; then
        for k in vb_spatial_prep_kernel vb_spatial_slab_sweep_kernel vb_spatial_setup_kernel vb_spatial_ak_partial_kernel; do
            python3 tools/pmc_summary.py gpurun_out/pmc_r4$w $k > $OUT/r4_pmc_c5_$k.json
        done
        # the second sweep's instances (the streaming one and the one with the half-ulp exp), by their full names
        i=0
        grep -h -o '"void fvb::vb_spatial_noise_kernel[^"]*"' gpurun_out/pmc_r4$w/trace/*/*kernel_stats.csv | tr -d '"' | sort -u | while read -r name; do
            python3 tools/pmc_summary.py gpurun_out/pmc_r4$w "$name" > $OUT/r4_pmc_c5_vb_spatial_noise_kernel_$i.json
            i=$((i+1))
        done
        python3 tools/pmc_c5_merge.py $OUT/r4_pmc_c5_vb_*.json > $OUT/r4_pmc_c5.json
    else
        python3 tools/pmc_summary.py gpurun_out/pmc_r4$w vb_lane > $OUT/r4_pmc_$w.json
    fi
    cp gpurun_out/pmc_r4$w/trace/*/*kernel_stats.csv $OUT/r4_kernel_stats_$w.csv
    echo "[r4_profiles] $w done"
done
# bench.py reads profiles/r4_pmc_*.json: put this run's files there before the bench lines are taken
cp $OUT/r4_pmc_c3.json $OUT/r4_pmc_c3_needf.json $OUT/r4_pmc_c5.json profiles/ 2>/dev/null
python3 bench.py --steps 20 --warmup 5 > $OUT/r4_bench_c3.json 2> $OUT/bench_c3.err
python3 bench.py --steps 20 --need-f --cpu-sample 0 --no-e2e > $OUT/r4_bench_c3_needf.json 2>/dev/null
python3 bench.py --steps 20 --workload c2 --no-e2e > $OUT/r4_bench_c2.json 2>/dev/null
python3 bench.py --steps 10 --workload c4 --no-e2e > $OUT/r4_bench_c4.json 2>/dev/null
python3 bench.py --steps 10 --workload c5 > $OUT/r4_bench_c5.json 2>/dev/null
python3 bench.py --steps 10 --workload c5 --prior-type P --cpu-sample 0 > $OUT/r4_bench_c5_typeP.json 2>/dev/null
python3 bench.py --steps 5 --workload c1 --no-e2e > $OUT/r4_bench_c1.json 2>/dev/null
echo "[r4_profiles] all done"
