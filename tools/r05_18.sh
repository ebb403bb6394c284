#!/bin/bash
# round 5: the training step after the host-/launch-side trims (per-step outputs, no materialised zero gradients, one zeroing
# launch, one transpose launch, tiled recomputation kernels): tests, kernel stats, wall
set -e
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
cd $ROOT
python -m pytest tests/test_gpu_training.py tests/test_gpu_torch_ops.py tests/test_gpu_ddp.py tests/test_gpu_loss.py tests/test_gpu_abi_from_c.py tests/test_gpu_integration_doc.py -x -q > $OUT/r05_t18.log 2>&1 || { tail -30 $OUT/r05_t18.log; exit 1; }
tail -2 $OUT/r05_t18.log
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/prof/train18 -o train --output-format csv -- python3 $ROOT/tools/train_loop.py 100 > $OUT/train18.log 2>&1
python3 $ROOT/tools/trim_stats.py $OUT/prof/train18/train_kernel_stats.csv $OUT/r05f_train_kernel_stats.csv
tail -4 $OUT/train18.log
cd $ROOT
for rep in 1 2 3; do
  echo "$(python3 tools/train_step_ab.py 2>/dev/null | tail -1)"
done | tee $OUT/r05_train_ab3.txt
