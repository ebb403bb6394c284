#!/bin/bash
# round 5: long seeded sweeps of the forward (400 configurations) and of the backward (60) against the fp64 oracle
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
cd $ROOT
MTMC_FUZZ_SEEDS=400 timeout -k 10 900 python -m pytest tests/test_gpu_random_configs.py -q -x > $OUT/r05_fuzz_fwd.log 2>&1
tail -3 $OUT/r05_fuzz_fwd.log
MTMC_FUZZ_BWD_SEEDS=60 timeout -k 10 600 python -m pytest tests/test_gpu_training.py -q -x -k random_configurations > $OUT/r05_fuzz_bwd.log 2>&1
tail -3 $OUT/r05_fuzz_bwd.log
