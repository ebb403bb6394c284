#!/bin/bash
# one rocprofv3 --pmc pass per counter named on the command line, config-4 forward:  bash tools/pmc_pass.sh CTR [CTR...]
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for ctr in "$@"; do
  timeout -k 10 300 rocprofv3 --pmc $ctr -d $OUT/p_$ctr -o p --output-format csv -- python3 $ROOT/tools/fwd_loop.py cfg4 3 > $OUT/p_$ctr.log 2>&1 || { echo "$ctr: failed"; tail -3 $OUT/p_$ctr.log; continue; }
  python3 $ROOT/tools/pmc_summary.py $OUT/p_$ctr/p_counter_collection.csv $ctr | grep -v "rocprim\|at::native\|rocclr" | head -14
done
