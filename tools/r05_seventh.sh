#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
cd $ROOT
timeout -k 10 900 python3 -m pytest tests/test_gpu_regimes.py tests/test_gpu_training.py -x -q > $OUT/r05_t8.log 2>&1; echo "tests: $?"
tail -n 4 $OUT/r05_t8.log
MTMC_MPN_LIB=$ROOT/build_ab/stamp/pkg/csrc/libmtmc_mpn.so timeout -k 10 300 python3 tools/edge_stamps.py s02 2>&1 | grep -v amdgpu.ids > $OUT/r05_edge_stamps.txt
cat $OUT/r05_edge_stamps.txt
