#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
cd $ROOT
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_regimes.py -x -q > $OUT/r05_t9.log 2>&1; echo "tests: $?"
tail -n 3 $OUT/r05_t9.log
MTMC_MPN_LIB=$ROOT/build_ab/projmfma/pkg/csrc/libmtmc_mpn.so timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -x -q > $OUT/r05_t10.log 2>&1; echo "tests projmfma: $?"
tail -n 3 $OUT/r05_t10.log
for rep in 1 2 3; do
  for which in nohoist base projmfma; do
    if [ $which = base ]; then unset MTMC_MPN_LIB; else export MTMC_MPN_LIB=$ROOT/build_ab/$which/pkg/csrc/libmtmc_mpn.so; fi
    python3 bench.py --workload s02 --steps 100 --warmup 10 --no-cpu --no-stress 2>$OUT/r05_bench_err.log | python3 -c "
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('s02 $which: %.4f ms  (median %.4f, p10 %.4f)' % (d['ms_per_step'], d['step_ms']['median'], d['step_ms']['p10']), {k: v for k, v in d['phase_ms'].items() if 'node_proj' in k or 'pass' in k})
"
  done
done | tee $OUT/r05_edge_ab.txt
unset MTMC_MPN_LIB
tail -n 3 $OUT/r05_bench_err.log
MTMC_MPN_LIB=$ROOT/build_ab/stamp/pkg/csrc/libmtmc_mpn.so timeout -k 10 300 python3 tools/edge_stamps.py s02 2>&1 | grep -v amdgpu.ids > $OUT/r05_edge_stamps2.txt
cat $OUT/r05_edge_stamps2.txt
