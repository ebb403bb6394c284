#!/bin/bash
# round 5: device seed resolved once per kernel: tests, A/B against the build without the indirection, the eval headline's kernels
set -e
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
cd $ROOT
python -m pytest tests/test_gpu_training.py tests/test_gpu_torch_ops.py tests/test_gpu_parity.py -x -q > $OUT/r05_t36.log 2>&1 || { tail -30 $OUT/r05_t36.log; exit 1; }
tail -2 $OUT/r05_t36.log
bash tools/r05_35.sh > $OUT/r05_seed_ab2.txt 2>&1
grep -A1 "^==" $OUT/r05_seed_ab2.txt
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/prof/s02c -o s02 --output-format csv -- python3 $ROOT/tools/fwd_loop.py s02 200 > $OUT/s02c.log 2>&1
python3 $ROOT/tools/trim_stats.py $OUT/prof/s02c/s02_kernel_stats.csv $OUT/r05g_s02_kernel_stats.csv
tail -1 $OUT/s02c.log
