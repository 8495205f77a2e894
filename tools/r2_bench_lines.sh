ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r2p
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/prof_c5
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_c5 -- python3 $ROOT/bench.py --workload c5 --steps 3 --warmup 1 --cpu-sample 0 > $OUT/prof_c5.log 2>&1 < /dev/null
cp $OUT/prof_c5/*/*kernel_stats.csv $OUT/r2_kernel_stats_c5.csv
echo "[bench] c5 profile done"
cd $ROOT
python3 bench.py --steps 20 --warmup 5 > $OUT/r2_bench_c3.json 2> $OUT/bench_c3.err
echo "[bench] c3 done"
python3 bench.py --steps 10 --need-f --cpu-sample 0 > $OUT/r2_bench_c3_needf.json 2>/dev/null
python3 bench.py --steps 20 --workload c2 > $OUT/r2_bench_c2.json 2>/dev/null
echo "[bench] c2 done"
python3 bench.py --steps 10 --workload c4 > $OUT/r2_bench_c4.json 2>/dev/null
echo "[bench] c4 done"
python3 bench.py --steps 5 --workload c5 > $OUT/r2_bench_c5.json 2>/dev/null
python3 bench.py --steps 5 --workload c1 > $OUT/r2_bench_c1.json 2>/dev/null
python3 bench.py --steps 3 --gpus 1 --single-process --cpu-sample 0 > $OUT/r2_bench_c3_single_process.json 2>$OUT/sp.err
echo "[bench] all done"
