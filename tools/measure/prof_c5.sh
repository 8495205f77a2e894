#!/bin/bash
# per-kernel times of the C5 bench for several slab thicknesses of the spatial sweep
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd /tmp && export TMPDIR=/tmp
for cfg in ${CFGS:-1 2 4}; do
  export FVB_SPATIAL_SLAB_DZ=$cfg
  OUT=$ROOT/gpurun_out/c5prof_$cfg
  rm -rf $OUT
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $ROOT/bench.py --workload c5 --cpu-sample 0 --steps 3 > $OUT.log 2>&1 < /dev/null
  f=$(find $OUT -name "*kernel_stats.csv" | head -1)
  if [ -n "$f" ]; then echo "cfg=$cfg $(grep -h "sweep_kernel" "$f" | cut -d, -f1-4 | cut -c1-120)"; else echo "cfg=$cfg no kernel_stats.csv"; fi
done
