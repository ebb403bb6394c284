#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
cd $ROOT
for rep in 1 2 3 4; do
  for nf in 1 0; do
    if [ $nf = 1 ]; then export MTMC_GEMM_NO_FEW=1; else unset MTMC_GEMM_NO_FEW; fi
    echo "NO_FEW=$nf $(python3 tools/train_step_ab.py 2>/dev/null | tail -1)"
  done
done | tee $OUT/r05_train_ab2.txt
