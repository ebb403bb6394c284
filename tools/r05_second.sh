#!/bin/bash
# Round 5, second GPU call: timelines of the few-row kernels, bank conflicts old vs new plane swizzle, A/B of the swizzle on
# the many-row GEMMs and of the few-row path on the headline graph.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
cd $ROOT
timeout -k 10 600 python3 -m pytest tests/test_gpu_gemm_few.py tests/test_gpu_weight_cache.py tests/test_gpu_gemm_presplit.py tests/test_gpu_gemm_staged.py -x -q > $OUT/r05_t3.log 2>&1; echo "tests: $?"
tail -3 $OUT/r05_t3.log
MTMC_MPN_LIB=$ROOT/build_ab/stamp/pkg/csrc/libmtmc_mpn.so timeout -k 10 300 python3 tools/few_stamps.py 450 > $OUT/r05_few_stamps.txt 2>&1; echo "stamps: $?"
cat $OUT/r05_few_stamps.txt
ab() {  # workload steps
  for rep in 1 2 3; do
    for which in swzold new; do
      if [ $which = swzold ]; then export MTMC_MPN_LIB=$ROOT/build_ab/swzold/pkg/csrc/libmtmc_mpn.so; else unset MTMC_MPN_LIB; fi
      python3 bench.py --workload $1 --steps $2 --warmup 10 --no-cpu --no-stress 2>$OUT/r05_bench_err.log | python3 -c "
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1 $which: %.4f ms  (median %.4f, p10 %.4f)' % (d['ms_per_step'], d['step_ms']['median'], d['step_ms']['p10']), {k: v for k, v in d['phase_ms'].items() if 'gemm' in k or 'few' in k})
"
    done
  done
}
ab cfg4 20 | tee $OUT/r05_swz_ab.txt
unset MTMC_MPN_LIB
for rep in 1 2 3; do
  for which in old new; do
    if [ $which = old ]; then export MTMC_GEMM_NO_FEW=1; else unset MTMC_GEMM_NO_FEW; fi
    python3 bench.py --workload s02 --steps 100 --warmup 10 --no-cpu --no-stress 2>$OUT/r05_bench_err.log | python3 -c "
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('s02 $which: %.4f ms  (median %.4f, p10 %.4f)  eager %.4f  replay %s' % (d['ms_per_step'], d['step_ms']['median'], d['step_ms']['p10'], d['launch']['eager_ms'], d['launch']['graph_replay_ms']))
"
  done
done | tee $OUT/r05_few_ab.txt
unset MTMC_GEMM_NO_FEW
tail -5 $OUT/r05_bench_err.log
cd /tmp && export TMPDIR=/tmp
for which in swzold new; do
  if [ $which = swzold ]; then export MTMC_MPN_LIB=$ROOT/build_ab/swzold/pkg/csrc/libmtmc_mpn.so; else unset MTMC_MPN_LIB; fi
  timeout -k 10 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $OUT/prof/lds_$which -o lds --output-format csv -- python3 $ROOT/tools/fwd_loop.py cfg4 3 > $OUT/prof_lds_$which.log 2>&1
  python3 $ROOT/tools/pmc_summary.py $OUT/prof/lds_$which/lds_counter_collection.csv SQ_LDS_BANK_CONFLICT | grep "gemm\|few" > $OUT/r05_cfg4_lds_conflict_$which.txt
  python3 $ROOT/tools/pmc_summary.py $OUT/prof/lds_$which/lds_counter_collection.csv SQ_LDS_IDX_ACTIVE | grep "gemm\|few" >> $OUT/r05_cfg4_lds_conflict_$which.txt
  cat $OUT/r05_cfg4_lds_conflict_$which.txt
done
