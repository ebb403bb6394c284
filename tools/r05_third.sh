#!/bin/bash
# Round 5, third GPU call: few_l0 tile width (WN) and few_wave row blocks (RB), stamps + same-box A/B on the headline graph.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
cd $ROOT
timeout -k 10 600 python3 -m pytest tests/test_gpu_gemm_few.py tests/test_gpu_weight_cache.py -x -q > $OUT/r05_t4.log 2>&1; echo "tests: $?"
tail -3 $OUT/r05_t4.log
MTMC_FEW_L0_WN=2 timeout -k 10 600 python3 -m pytest tests/test_gpu_gemm_few.py -x -q -k layer0 > $OUT/r05_t5.log 2>&1; echo "tests WN=2: $?"
tail -3 $OUT/r05_t5.log
for wn in 1 2; do
  echo "== MTMC_FEW_L0_WN=$wn"
  MTMC_FEW_L0_WN=$wn MTMC_MPN_LIB=$ROOT/build_ab/stamp/pkg/csrc/libmtmc_mpn.so timeout -k 10 300 python3 tools/few_stamps.py 450 2>&1 | grep -v amdgpu.ids
done > $OUT/r05_few_stamps2.txt
cat $OUT/r05_few_stamps2.txt
for rep in 1 2 3; do
  for cfg in "1 1" "2 1" "1 2" "2 2"; do
    set -- $cfg
    MTMC_FEW_L0_WN=$1 MTMC_FEW_WAVE_RB=$2 python3 bench.py --workload s02 --steps 100 --warmup 10 --no-cpu --no-stress 2>$OUT/r05_bench_err.log | python3 -c "
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('s02 WN=$1 RB=$2: %.4f ms  (median %.4f, p10 %.4f)' % (d['ms_per_step'], d['step_ms']['median'], d['step_ms']['p10']), {k: v for k, v in d['phase_ms'].items() if 'few' in k or 'prep' in k})
"
  done
done | tee $OUT/r05_few_variants_ab.txt
tail -3 $OUT/r05_bench_err.log
