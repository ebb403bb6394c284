#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
cd $ROOT
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_regimes.py tests/test_gpu_training.py tests/test_gpu_sharded_forward.py -x -q > $OUT/r05_t11.log 2>&1; echo "tests: $?"
tail -n 3 $OUT/r05_t11.log
MTMC_MPN_LIB=$ROOT/build_ab/stamp/pkg/csrc/libmtmc_mpn.so timeout -k 10 300 python3 tools/edge_stamps.py s02 2>&1 | grep -v amdgpu.ids > $OUT/r05_edge_stamps3.txt
cat $OUT/r05_edge_stamps3.txt
cd /tmp && export TMPDIR=/tmp
for which in before new; do
  if [ $which = new ]; then unset MTMC_MPN_LIB; else export MTMC_MPN_LIB=$ROOT/build_ab/$which/pkg/csrc/libmtmc_mpn.so; fi
  timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/prof/s02_$which -o s02 --output-format csv -- python3 $ROOT/tools/fwd_loop.py s02 200 > $OUT/prof_s02_$which.log 2>&1
  python3 $ROOT/tools/trim_stats.py $OUT/prof/s02_$which/s02_kernel_stats.csv $OUT/r05c_s02_kernel_stats_$which.csv
  echo "== $which"; python3 $ROOT/tools/kstats.py $OUT/r05c_s02_kernel_stats_$which.csv 210
done
