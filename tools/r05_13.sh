#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
cd $ROOT
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_regimes.py tests/test_gpu_weight_cache.py tests/test_gpu_gemm_few.py tests/test_gpu_torch_ops.py tests/test_gpu_sharded_forward.py -x -q > $OUT/r05_t13.log 2>&1; echo "tests: $?"
tail -n 3 $OUT/r05_t13.log
for rep in 1 2 3; do
  for nr in 1 0; do
    if [ $nr = 1 ]; then export MTMC_NO_PREP_RIDE=1; else unset MTMC_NO_PREP_RIDE; fi
    python3 bench.py --workload s02 --steps 100 --warmup 10 --no-cpu --no-stress 2>$OUT/r05_bench_err.log | python3 -c "
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('s02 NO_PREP_RIDE=$nr: %.4f ms  (median %.4f, p10 %.4f) replay %s' % (d['ms_per_step'], d['step_ms']['median'], d['step_ms']['p10'], d['launch']['graph_replay_ms']))
"
  done
done | tee $OUT/r05_prep_ride_ab.txt
unset MTMC_NO_PREP_RIDE
tail -n 3 $OUT/r05_bench_err.log
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/prof/s02_ride -o s02 --output-format csv -- python3 $ROOT/tools/fwd_loop.py s02 200 > $OUT/prof_s02_ride.log 2>&1
python3 $ROOT/tools/trim_stats.py $OUT/prof/s02_ride/s02_kernel_stats.csv $OUT/r05d_s02_kernel_stats.csv
python3 $ROOT/tools/kstats.py $OUT/r05d_s02_kernel_stats.csv 210
