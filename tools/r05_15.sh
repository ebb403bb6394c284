#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/prof/train_few -o train --output-format csv -- python3 $ROOT/tools/train_loop.py 100 > $OUT/prof_train_few.log 2>&1
python3 $ROOT/tools/trim_stats.py $OUT/prof/train_few/train_kernel_stats.csv $OUT/r05e_train_kernel_stats.csv
grep "steps profiled\|train step" $OUT/prof_train_few.log
python3 $ROOT/tools/kstats.py $OUT/r05e_train_kernel_stats.csv 205 40
