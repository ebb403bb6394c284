#!/bin/bash
# round 5: what does the device-seed indirection in drop_keep / drop_apply cost the kernels that hash per element?  (same box)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
for v in default noseed default noseed; do
  if [ $v = default ]; then unset MTMC_MPN_LIB; else export MTMC_MPN_LIB=$ROOT/build_ab/$v/pkg/csrc/libmtmc_mpn.so; fi
  rm -rf $OUT/prof/seedab
  timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/prof/seedab -o t --output-format csv -- python3 $ROOT/tools/train_loop.py 60 > $OUT/seedab.log 2>&1
  echo "== $v"
  python3 - <<PY
import csv
rows=list(csv.DictReader(open("$OUT/prof/seedab/t_kernel_stats.csv")))
tot=sum(float(r['TotalDurationNs']) for r in rows)
n=[int(r['Calls']) for r in rows if 'prep_kernel' in r['Name']][0]
print('kernel us/step', round(tot/n/1000,1))
for r in rows:
    if any(k in r['Name'] for k in ('few_wave','pass_c_kernel','bwd_node_upd','pass_b_kernel','node_proj')):
        print('  ', r['Name'][:60], round(float(r['AverageNs'])/1000,2))
PY
done
