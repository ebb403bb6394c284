#!/bin/bash
# A/B of the pipelined layer 0 (MTMC_L0_PIPELINE: 0 off, 1 panels of 1,2,4,8,.. rounds, 2 / 3 uniform 1 / 2 rounds) on the
# many-row workloads; prints ms_per_step and the phase split per mode.  Run on a GPU box:  bash tools/l0_pipeline_ab.sh
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
for wl in cfg4 cfg5; do
  for mode in 0 1 2 3 1 0; do
    steps=20; [ $wl = cfg5 ] && steps=6
    MTMC_L0_PIPELINE=$mode python3 $ROOT/bench.py --workload $wl --steps $steps --warmup 3 --no-cpu 2>/dev/null | python3 -c "
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$wl mode $mode: %.4f ms  (median %.4f)  begin %.3f  gemm0 %.3f' % (d['ms_per_step'], d['step_ms']['median'], d['phase_ms'].get('memset+prep_kernel(+split_rows_kernel)', 0), d['phase_ms'].get('gemm_f16p_m16_kernel[0]', 0)))
"
  done
done
