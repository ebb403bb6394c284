#!/bin/bash
set -e
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --memory-copy-trace --stats -d $OUT/prof/traing -o train --output-format csv -- python3 $ROOT/tools/train_graph_loop.py 100 > $OUT/traing.log 2>&1
tail -3 $OUT/traing.log
ls $OUT/prof/traing
