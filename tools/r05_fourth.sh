#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
cd $ROOT
timeout -k 10 600 python3 -m pytest tests/test_gpu_gemm_few.py tests/test_gpu_weight_cache.py tests/test_gpu_parity.py -x -q > $OUT/r05_t6.log 2>&1; echo "tests: $?"
tail -n 3 $OUT/r05_t6.log
MTMC_MPN_LIB=$ROOT/build_ab/stamp/pkg/csrc/libmtmc_mpn.so timeout -k 10 300 python3 tools/few_stamps.py 450 2>&1 | grep -v amdgpu.ids > $OUT/r05_few_stamps3.txt
cat $OUT/r05_few_stamps3.txt
for rep in 1 2 3; do
  for cfg in "0 1" "0 0" "1 0"; do
    set -- $cfg
    if [ $1 = 1 ]; then export MTMC_GEMM_NO_FEW=1; else unset MTMC_GEMM_NO_FEW; fi
    MTMC_FEW_WAVE_RB=$2 python3 bench.py --workload s02 --steps 100 --warmup 10 --no-cpu --no-stress 2>$OUT/r05_bench_err.log | python3 -c "
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('s02 NO_FEW=$1 RB=$2: %.4f ms  (median %.4f, p10 %.4f)' % (d['ms_per_step'], d['step_ms']['median'], d['step_ms']['p10']), {k: v for k, v in d['phase_ms'].items() if 'few' in k or 'prep' in k})
"
  done
done | tee $OUT/r05_few_variants_ab2.txt
unset MTMC_GEMM_NO_FEW
tail -n 3 $OUT/r05_bench_err.log
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/prof/s02_stats -o s02 --output-format csv -- python3 $ROOT/tools/fwd_loop.py s02 200 > $OUT/prof_s02.log 2>&1
python3 $ROOT/tools/trim_stats.py $OUT/prof/s02_stats/s02_kernel_stats.csv $OUT/r05b_s02_kernel_stats.csv
python3 $ROOT/tools/kstats.py $OUT/r05b_s02_kernel_stats.csv 210
