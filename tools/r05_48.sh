#!/bin/bash
# round 5 (as asked): the walk behind pass_c_sorted_kernel launched with ONE workgroup, config 4
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
for v in default walk1 default walk1; do
  if [ $v = default ]; then unset MTMC_MPN_LIB; else export MTMC_MPN_LIB=$ROOT/build_ab/$v/pkg/csrc/libmtmc_mpn.so; fi
  rm -rf $OUT/prof/walk1
  timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/prof/walk1 -o t --output-format csv -- python3 $ROOT/tools/fwd_loop.py cfg4 20 > $OUT/walk1.log 2>&1
  python3 - <<PY
import csv
rows=list(csv.DictReader(open("$OUT/prof/walk1/t_kernel_stats.csv")))
n=[int(r['Calls']) for r in rows if 'prep_kernel' in r['Name']][0]
tot=sum(float(r['TotalDurationNs']) for r in rows)
w=[r for r in rows if r['Name'].startswith('mtmc::pass_c_kernel')][0]
print("$v", 'kernel us/forward', round(tot/n/1000,1), '| pass_c_kernel (the walk behind the sorted kernel):', round(float(w['AverageNs'])/1000,2), 'us x', int(w['Calls'])//n)
PY
done
