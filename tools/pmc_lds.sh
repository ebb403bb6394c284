#!/bin/bash
# LDS bank-conflict share per kernel at config 4 (two separate rocprofv3 --pmc passes):  bash tools/pmc_lds.sh
set -e
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for ctr in SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE; do
  timeout -k 10 300 rocprofv3 --pmc $ctr -d $OUT/lds_$ctr -o lds --output-format csv -- python3 $ROOT/tools/fwd_loop.py cfg4 3 > $OUT/lds_$ctr.log 2>&1
  python3 $ROOT/tools/pmc_summary.py $OUT/lds_$ctr/lds_counter_collection.csv $ctr > $ROOT/gpurun_out/r01_cfg4_pmc_$ctr.txt
done
head -12 $ROOT/gpurun_out/r01_cfg4_pmc_SQ_LDS_BANK_CONFLICT.txt
head -12 $ROOT/gpurun_out/r01_cfg4_pmc_SQ_LDS_IDX_ACTIVE.txt
