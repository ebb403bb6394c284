#!/bin/bash
# SQ wait / active counters of the two forms of the role-split GEMM on layer 1 of config 4 ("report
# SQ_WAIT_ANY / SQ_WAVE_CYCLES before / after"): the product's gemm_staged_kernel<256,2> and the laboratory's
# gemm_staged_w_kernel (STAGED_LAB=1).  One rocprofv3 --pmc pass each.   bash tools/pmc_staged_forms.sh
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for lab in 0 1; do
  export STAGED_LAB=$lab
  timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU -d $OUT/stg_$lab -o stg --output-format csv -- python3 $ROOT/tools/staged_time.py 100000 1024 512 > $OUT/stg_$lab.log 2>&1 || { echo "lab=$lab failed"; tail -3 $OUT/stg_$lab.log; continue; }
  echo "== STAGED_LAB=$lab"
  for c in SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU; do
    python3 $ROOT/tools/pmc_summary.py $OUT/stg_$lab/stg_counter_collection.csv $c | grep "gemm_staged"
  done
done
