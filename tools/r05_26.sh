#!/bin/bash
# round 5: (1) the rest of the backward sweep, every seed; (2) few-edge / many-edge regime threshold at the tracker-scale graph
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
cd $ROOT
MTMC_FUZZ_BWD_SEEDS=80 timeout -k 10 600 python -m pytest tests/test_gpu_training.py -q -k random_configurations > $OUT/r05_fuzz_bwd2.log 2>&1
tail -4 $OUT/r05_fuzz_bwd2.log
{
for rep in 1 2 3; do
  for v in default se1m se1m_cap4k; do
    if [ $v = default ]; then unset MTMC_MPN_LIB; else export MTMC_MPN_LIB=$ROOT/build_ab/$v/pkg/csrc/libmtmc_mpn.so; fi
    echo "$v $(python3 tools/fwd_loop.py s02_tracker 300 2>/dev/null | tail -1)"
  done
done
} | tee $OUT/r05_small_edges_ab.txt
