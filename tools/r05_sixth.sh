#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
cd $ROOT
timeout -k 10 1000 python3 -m pytest tests/test_gpu_training.py tests/test_gpu_regimes.py tests/test_gpu_large_configs.py tests/test_gpu_parity.py tests/test_gpu_gemm_few.py tests/test_gpu_weight_cache.py -x -q > $OUT/r05_t7.log 2>&1; echo "tests: $?"
tail -n 5 $OUT/r05_t7.log
for rep in 1 2; do
  MTMC_FEW_ROWS_MAX=4095 python3 tools/few_crossover.py 2>/dev/null
  MTMC_GEMM_NO_FEW=1 python3 tools/few_crossover.py 2>/dev/null
done | tee $OUT/r05_few_crossover.txt
timeout -k 10 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/r05_bench_default.json 2> $OUT/r05_bench_default.err; echo "bench: $?"
tail -n 3 $OUT/r05_bench_default.err
python3 - <<'PY'
import json
d = json.loads(open("gpurun_out/r05_bench_default.json").read().strip().splitlines()[-1])
print("headline", d["ms_per_step"], "uncached", d.get("ms_per_step_uncached"), "fp32", d.get("ms_per_step_exact_fp32"))
print("roofline", {k: d["roofline"][k] for k in ("kernel", "frac", "avg_kernel_ms", "launches_per_step")})
print("longest", {k: d.get("roofline_longest_launch", {}).get(k) for k in ("kernel", "frac", "avg_kernel_ms")})
for k in ("stress", "scale_base"):
    r = d[k]["roofline"]
    print(k, d[k]["ms_per_step"], "uncached", d[k].get("ms_per_step_uncached"), {x: r.get(x) for x in ("kernel", "frac", "avg_kernel_ms", "kernel_ms_per_step", "launches_per_step")}, r.get("phase_path_single_launch"))
print("extra", {k: (v["ms_per_step"], v["ms_per_step_uncached"]) for k, v in d["extra"].items()})
print("train", d["training_step"]["ms_per_step"])
PY
