#!/bin/bash
# Regenerates profiles/r01_* on a GPU box:  bash tools/refresh_profiles.sh   (run through gpurun)
# kernel-trace/stats and each PMC counter are separate rocprofv3 passes, as the guide prescribes.
set -e
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for wl in s02 cfg4; do
  it=200; [ $wl = cfg4 ] && it=20
  timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/${wl}_stats -o $wl --output-format csv -- python3 $ROOT/tools/fwd_loop.py $wl $it > $OUT/${wl}_stats.log 2>&1
  cp $OUT/${wl}_stats/${wl}_kernel_stats.csv $ROOT/gpurun_out/r01_${wl}_kernel_stats.csv
  for ctr in FETCH_SIZE WRITE_SIZE; do
    it2=20; [ $wl = cfg4 ] && it2=5
    timeout -k 10 300 rocprofv3 --pmc $ctr -d $OUT/${wl}_$ctr -o $wl --output-format csv -- python3 $ROOT/tools/fwd_loop.py $wl $it2 > $OUT/${wl}_$ctr.log 2>&1
    python3 $ROOT/tools/pmc_summary.py $OUT/${wl}_$ctr/${wl}_counter_collection.csv $ctr > $ROOT/gpurun_out/r01_${wl}_pmc_$ctr.txt
  done
  echo "$wl done"
done
