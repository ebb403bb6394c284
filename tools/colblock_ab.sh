#!/bin/bash
# A/B of the column-blocked pass A (MTMC_NO_COL_BLOCKS=1: edge order; MTMC_COL_BLOCKS=B: block count) on config 5.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
run() {
  env "$@" python3 $ROOT/bench.py --workload cfg5 --steps 6 --warmup 3 --no-cpu 2>/dev/null | python3 -c "
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('cfg5 $*: %.4f ms  (median %.4f)  pass_a %.3f  begin %.3f' % (d['ms_per_step'], d['step_ms']['median'], d['phase_ms'].get('pass_a_kernel', 0), d['phase_ms'].get('memset+prep_kernel(+split_rows_kernel)', 0)))
"
}
run MTMC_NO_COL_BLOCKS=1
run MTMC_COL_BLOCKS=0
run MTMC_COL_BLOCKS=16
run MTMC_COL_BLOCKS=32
run MTMC_COL_BLOCKS=0
run MTMC_NO_COL_BLOCKS=1
