#!/bin/bash
# Round 5, first GPU call: the few-row kernels and the weight-plane cache -- unit tests, parity, then an A/B against the
# split-K path on the same box and a kernel trace of the headline forward.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
cd $ROOT
timeout -k 10 900 python3 -m pytest tests/test_gpu_gemm_few.py tests/test_gpu_weight_cache.py -x -q > $OUT/r05_t1.log 2>&1; echo "tests new: $?" 
tail -5 $OUT/r05_t1.log
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_abi_from_c.py tests/test_gpu_integration_doc.py -x -q > $OUT/r05_t2.log 2>&1; echo "tests parity: $?"
tail -5 $OUT/r05_t2.log
for rep in 1 2; do
  for which in old new; do
    if [ $which = old ]; then export MTMC_GEMM_NO_FEW=1; else unset MTMC_GEMM_NO_FEW; fi
    python3 bench.py --workload s02 --steps 100 --warmup 10 --no-cpu --no-stress 2>/dev/null | python3 -c "
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('s02 $which: %.4f ms  (median %.4f, p10 %.4f)  eager %.4f  replay %s' % (d['ms_per_step'], d['step_ms']['median'], d['step_ms']['p10'], d['launch']['eager_ms'], d['launch']['graph_replay_ms']))
print('   phases', {k: v for k, v in d['phase_ms'].items()})
"
  done
done | tee $OUT/r05_few_ab.txt
unset MTMC_GEMM_NO_FEW
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/prof/s02_stats -o s02 --output-format csv -- python3 $ROOT/tools/fwd_loop.py s02 200 > $OUT/prof_s02.log 2>&1
python3 $ROOT/tools/trim_stats.py $OUT/prof/s02_stats/s02_kernel_stats.csv $OUT/r05a_s02_kernel_stats.csv
python3 $ROOT/tools/kstats.py $OUT/r05a_s02_kernel_stats.csv 210
