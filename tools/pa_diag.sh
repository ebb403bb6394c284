#!/bin/bash
# What bounds the in-order pass A at config 5?  Applies tools/pa_diag.patch (timing variants, results garbage), rebuilds the
# library with -DPA_DIAG=n and times the edge passes:  1 = Pc gathers from a 64 KB window (no random access),
# 2 = no z1 store, 3 = no load of the previous z1, 4 = Pc gathered at the ROW (sequential).  Restores the source and the
# library afterwards.  Run on a GPU box:  bash tools/pa_diag.sh
set -e
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
CS=$ROOT/graph-convolutional-network-for-multi-camera-vehicle-tracking_amd/csrc
cd $ROOT && patch -p1 < tools/pa_diag.patch > /dev/null
export MTMC_NO_COL_BLOCKS=1 MTMC_SKIP_ISA_LINT=1
for d in 0 1 2 3 4; do
  (cd $CS && rm -f edge_kernels.o && make -s EXTRA="-DPA_DIAG=$d" libmtmc_mpn.so > /dev/null 2>&1)
  python3 $ROOT/bench.py --workload cfg5 --steps 4 --warmup 2 --no-cpu 2>/dev/null | python3 -c "
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('PA_DIAG=$d: forward %.3f ms  pass_a (3 launches) %.3f ms  pass_b %.3f  pass_c %.3f' % (d['ms_per_step'], d['phase_ms'].get('pass_a_kernel', 0), d['phase_ms'].get('pass_b_kernel', 0), d['phase_ms'].get('pass_c_kernel', 0)))
"
done
cd $ROOT && patch -R -p1 < tools/pa_diag.patch > /dev/null
(cd $CS && rm -f edge_kernels.o && make -s libmtmc_mpn.so > /dev/null 2>&1)
echo restored
