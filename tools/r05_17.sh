#!/bin/bash
# round 5, evidence pass: the whole GPU suite, the profiles of the three workloads, the default bench line
set -e
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q > gpurun_out/r05_gpu_tests.log 2>&1
tail -3 gpurun_out/r05_gpu_tests.log
ROUND=r05 bash tools/refresh_profiles.sh
cd $GRAFT_REPO_ROOT
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r05_bench_default.json 2> gpurun_out/r05_bench_default.err
tail -c 600 gpurun_out/r05_bench_default.json
