#!/bin/bash
# Regenerates profiles/${ROUND}_* on a GPU box:  ROUND=r05 bash tools/refresh_profiles.sh   (run through gpurun; copies land in
# gpurun_out/, move them to profiles/ afterwards).  kernel-trace/stats and each PMC counter are separate rocprofv3
# passes, as the guide prescribes; the program after `--` is python3 itself.
set -e
ROUND=${ROUND:-r05}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
trim() { python3 $ROOT/tools/trim_stats.py "$1" "$2"; }
for wl in ${WORKLOADS:-s02 cfg4 cfg5}; do
  it=200; [ $wl = cfg4 ] && it=20; [ $wl = cfg5 ] && it=5
  timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $OUT/${wl}_stats -o $wl --output-format csv -- python3 $ROOT/tools/fwd_loop.py $wl $it > $OUT/${wl}_stats.log 2>&1
  trim $OUT/${wl}_stats/${wl}_kernel_stats.csv $ROOT/gpurun_out/${ROUND}_${wl}_kernel_stats.csv
  for ctr in FETCH_SIZE WRITE_SIZE; do
    it2=20; [ $wl = cfg4 ] && it2=5; [ $wl = cfg5 ] && it2=2
    timeout -k 10 400 rocprofv3 --pmc $ctr -d $OUT/${wl}_$ctr -o $wl --output-format csv -- python3 $ROOT/tools/fwd_loop.py $wl $it2 > $OUT/${wl}_$ctr.log 2>&1
    python3 $ROOT/tools/pmc_summary.py $OUT/${wl}_$ctr/${wl}_counter_collection.csv $ctr | grep -v "at::native\|rocprim\|rocclr" > $ROOT/gpurun_out/${ROUND}_${wl}_pmc_$ctr.txt
  done
  echo "$wl done"
done
# the bench command itself (headline workload only) and the training step
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $OUT/bench_stats -o bench --output-format csv -- python3 $ROOT/bench.py --no-stress --no-cpu > $OUT/bench_stats.log 2>&1
trim $OUT/bench_stats/bench_kernel_stats.csv $ROOT/gpurun_out/${ROUND}_bench_s02_kernel_stats.csv
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/train -o train --output-format csv -- python3 $ROOT/tools/train_loop.py 100 > $OUT/train.log 2>&1
trim $OUT/train/train_kernel_stats.csv $ROOT/gpurun_out/${ROUND}_train_kernel_stats.csv
grep -o "steps profiled: [0-9]*" $OUT/train.log | grep -o "[0-9]*" > $ROOT/gpurun_out/${ROUND}_train_kernel_stats.steps || echo 205 > $ROOT/gpurun_out/${ROUND}_train_kernel_stats.steps
echo "all done"
