#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
cd $ROOT
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_regimes.py tests/test_gpu_large_configs.py tests/test_gpu_weight_cache.py -x -q > $OUT/r05_t12.log 2>&1; echo "tests: $?"
tail -n 3 $OUT/r05_t12.log
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/prof/cfg4_stats -o cfg4 --output-format csv -- python3 $ROOT/tools/fwd_loop.py cfg4 20 > $OUT/prof_cfg4.log 2>&1
python3 $ROOT/tools/trim_stats.py $OUT/prof/cfg4_stats/cfg4_kernel_stats.csv $OUT/r05_cfg4_kernel_stats.csv
python3 $ROOT/tools/kstats.py $OUT/r05_cfg4_kernel_stats.csv 30 16
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $OUT/prof/cfg5_phase -o cfg5 --output-format csv -- python3 $ROOT/tools/phase_loop.py cfg5 5 > $OUT/prof_cfg5_phase.log 2>&1
python3 $ROOT/tools/trim_stats.py $OUT/prof/cfg5_phase/cfg5_kernel_stats.csv $OUT/r05_cfg5_phase_kernel_stats.csv
tail -n 1 $OUT/prof_cfg5_phase.log
python3 $ROOT/tools/kstats.py $OUT/r05_cfg5_phase_kernel_stats.csv 7 16
cp $OUT/r05_cfg5_phase_kernel_stats.csv $ROOT/profiles/
cd $ROOT
MTMC_BENCH_FORCE_DIST=1 timeout -k 10 600 python3 bench.py --gpus 1 --workload cfg5 --steps 5 --warmup 2 > $OUT/r05_bench_dist_world1.json 2> $OUT/r05_bench_dist_world1.err; echo "bench_dist: $?"
tail -n 2 $OUT/r05_bench_dist_world1.err
python3 - <<'PY'
import json
d = json.loads(open("gpurun_out/r05_bench_dist_world1.json").read().strip().splitlines()[-1])
print("world1 ms", d["ms_per_step"], "replicated", d["ms_per_step_with_replicated_node_state"], "pred/meas", d.get("prediction_over_measured"))
print(d["prediction"]["per_rank_kernel_ms_by_phase"], d["prediction"]["predicted_ms_per_step"])
print(d["headline_graph_on_these_ranks"])
PY
