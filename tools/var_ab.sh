#!/bin/bash
# Same-box comparison of the in-tree library with variant builds under build/var_<name>/ (made by hand: edit, `make`, copy the
# .so) on config 5 -- how profiles/r04_cfg5_pass_a_blocked_variants.txt was made (blocked pass A: plain z1 stores, 2048
# workgroups, 2 / 8 slots per lane and trip; none better than the shipped 1024 workgroups x 4 slots with streaming stores).
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
run() {
  python3 $ROOT/bench.py --workload cfg5 --steps 6 --warmup 3 --no-cpu --no-stress 2>/dev/null | python3 -c "
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1: %.4f ms (median %.4f)  pass_a %.3f' % (d['ms_per_step'], d['step_ms']['median'], d['phase_ms'].get('pass_a_kernel', 0)))
"
}
for rep in 1 2; do
  unset MTMC_MPN_LIB; run base
  for v in plainstore grid2048 u2 u8; do
    export MTMC_MPN_LIB=$ROOT/build/var_$v/libmtmc_mpn.so; run $v
  done
done
unset MTMC_MPN_LIB; run base
