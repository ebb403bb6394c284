#!/bin/bash
# round 5: Dropout seed on the device + the training step as one HIP graph
set -e
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
cd $ROOT
python -m pytest tests/test_gpu_training.py tests/test_gpu_torch_ops.py tests/test_gpu_ddp.py -x -q > $OUT/r05_t20.log 2>&1 || { tail -40 $OUT/r05_t20.log; exit 1; }
tail -2 $OUT/r05_t20.log
for rep in 1 2 3; do
  echo "$(python3 tools/train_step_ab.py 2>/dev/null | tail -1)"
done | tee $OUT/r05_train_graph.txt
