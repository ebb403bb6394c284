#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
cd $ROOT
timeout -k 10 1000 python3 -m pytest tests/test_gpu_training.py tests/test_gpu_parity.py tests/test_gpu_torch_ops.py tests/test_gpu_ddp.py tests/test_gpu_random_configs.py tests/test_gpu_regimes.py -x -q > $OUT/r05_t14.log 2>&1; echo "tests: $?"
tail -n 4 $OUT/r05_t14.log
for rep in 1 2 3; do
  for nf in 1 0; do
    if [ $nf = 1 ]; then export MTMC_GEMM_NO_FEW=1; else unset MTMC_GEMM_NO_FEW; fi
    echo "NO_FEW=$nf $(python3 tools/train_loop.py 200 2>/dev/null | head -1)"
  done
done | tee $OUT/r05_train_ab.txt
unset MTMC_GEMM_NO_FEW
