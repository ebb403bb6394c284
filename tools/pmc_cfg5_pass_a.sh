#!/bin/bash
# FETCH_SIZE and L2 hit / miss counts of pass A at config 5, in edge order and by column blocks (separate rocprofv3 passes).
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for mode in edge blocked; do
  if [ $mode = edge ]; then export MTMC_NO_COL_BLOCKS=1; else unset MTMC_NO_COL_BLOCKS; fi
  for ctr in FETCH_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
    tag=$(echo $ctr | tr ' ' '_')
    timeout -k 10 400 rocprofv3 --pmc $ctr -d $OUT/c5_${mode}_$tag -o c5 --output-format csv -- python3 $ROOT/tools/fwd_loop.py cfg5 2 > $OUT/c5_${mode}_$tag.log 2>&1 || { echo "$mode $ctr: failed"; tail -3 $OUT/c5_${mode}_$tag.log; continue; }
    for c in $ctr; do
      echo "== $mode $c"
      python3 $ROOT/tools/pmc_summary.py $OUT/c5_${mode}_$tag/c5_counter_collection.csv $c | grep "pass_a\|colblock"
    done
  done
done
