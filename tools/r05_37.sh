#!/bin/bash
# round 5, end: smoke() and the world-size-1 rehearsal of the multi-GPU bench (every RCCL call issued in a group of one)
set -e
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
cd $ROOT
python3 __graft_entry__.py smoke > $OUT/r05_smoke.log 2>&1 || { tail -20 $OUT/r05_smoke.log; exit 1; }
grep "smoke ok" $OUT/r05_smoke.log
MTMC_BENCH_FORCE_DIST=1 timeout -k 10 600 python3 bench.py --gpus 1 --workload cfg5 --steps 5 --warmup 2 > $OUT/r05_bench_dist_world1.json 2> $OUT/r05_bench_dist_world1.err; echo "bench_dist: $?"
python3 - <<PY
import json
d = json.loads(open("$OUT/r05_bench_dist_world1.json").read().strip().splitlines()[-1])
print({k: d.get(k) for k in ("value", "ms_per_step", "n_gpus", "scaling")})
print("prediction", d.get("prediction", {}).get("predicted_ms_per_step_sharded_node_state"), d.get("prediction_over_measured"), d.get("ms_per_step_with_replicated_node_state"))
PY
