#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
for map in 0 1 2; do
  export MTMC_FEW_L0_MAP=$map
  timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/prof/map$map -o s02 --output-format csv -- python3 $ROOT/tools/fwd_loop.py s02 200 > $OUT/prof_map$map.log 2>&1
  echo "== MAP $map: $(grep few_l0 $OUT/prof/map$map/s02_kernel_stats.csv | cut -d, -f1-4)"
  timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE -d $OUT/prof/mapf$map -o s02 --output-format csv -- python3 $ROOT/tools/fwd_loop.py s02 20 > $OUT/prof_mapf$map.log 2>&1
  python3 $ROOT/tools/pmc_summary.py $OUT/prof/mapf$map/s02_counter_collection.csv FETCH_SIZE | grep few_l0
done 2>&1 | tee $OUT/r05_few_l0_map_ab.txt
