#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
cd $ROOT
timeout -k 10 1100 python3 -m pytest tests -m gpu -x -q > $OUT/r05_full2.log 2>&1; echo "full gpu suite: $?"
tail -n 4 $OUT/r05_full2.log
ROUND=r05 timeout -k 10 1500 bash tools/refresh_profiles.sh > $OUT/r05_refresh.log 2>&1; echo "refresh: $?"
tail -n 5 $OUT/r05_refresh.log
python3 tools/kstats.py $OUT/r05_s02_kernel_stats.csv 210
python3 tools/kstats.py $OUT/r05_cfg4_kernel_stats.csv 30 14
