#!/bin/bash
# rocprofv3 kernel statistics of the training step (run through gpurun):  bash tools/train_profile.sh
set -e
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $ROOT/gpurun_out/prof/train -o train --output-format csv -- python3 $ROOT/tools/train_loop.py 100 > $ROOT/gpurun_out/train_prof.log 2>&1
cp $ROOT/gpurun_out/prof/train/train_kernel_stats.csv $ROOT/gpurun_out/r01_train_kernel_stats.csv
python3 $ROOT/tools/kstats.py $ROOT/gpurun_out/r01_train_kernel_stats.csv 205 24
