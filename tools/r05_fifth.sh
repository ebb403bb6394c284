#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
cd $ROOT
timeout -k 10 1100 python3 -m pytest tests -m gpu -x -q > $OUT/r05_full1.log 2>&1; echo "full gpu suite: $?"
tail -n 15 $OUT/r05_full1.log
for rep in 1 2; do
  for nf in 1 0; do
    if [ $nf = 1 ]; then export MTMC_GEMM_NO_FEW=1; else unset MTMC_GEMM_NO_FEW; fi
    python3 bench.py --workload s02_tracker --steps 50 --warmup 10 --no-cpu --no-stress 2>$OUT/r05_bench_err.log | python3 -c "
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('s02_tracker NO_FEW=$nf: %.4f ms  (median %.4f, p10 %.4f)' % (d['ms_per_step'], d['step_ms']['median'], d['step_ms']['p10']), {k: v for k, v in d['phase_ms'].items() if 'few' in k or 'prep' in k or 'gemm' in k or 'combine' in k})
"
  done
done | tee $OUT/r05_tracker_ab.txt
unset MTMC_GEMM_NO_FEW
