#!/bin/bash
# round 5: where do the few-edge forms of the edge passes stop winning?  (default: many-edge forms above 524288 edges; se_inf: never)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
cd $ROOT
{
for rep in 1 2; do
  for v in default se_inf; do
    if [ $v = default ]; then unset MTMC_MPN_LIB; else export MTMC_MPN_LIB=$ROOT/build_ab/$v/pkg/csrc/libmtmc_mpn.so; fi
    echo "== $v"
    timeout -k 10 400 python3 tools/regime_sweep.py 200 250 300 350 400 500 650 800 1000 2>/dev/null
  done
done
} | tee $OUT/r05_regime_sweep.txt
