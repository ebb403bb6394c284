#!/bin/bash
# A/B of non-temporal access in the edge passes (csrc/edge_kernels.hip: PA_NT / PC_NT / PB_NT): rebuilds the library per variant on a GPU box,
# prints the rocprofv3 kernel averages of config 4 and config 5 and the sum of all edge passes per forward; leaves the default build.
R=$GRAFT_REPO_ROOT; CS=$R/graph-convolutional-network-for-multi-camera-vehicle-tracking_amd/csrc
cd /tmp && export TMPDIR=/tmp
run() { rm -rf /tmp/ks; timeout -k 10 300 rocprofv3 --kernel-trace --stats -d /tmp/ks -o x --output-format csv -- python3 $R/tools/fwd_loop.py $1 $2 > /tmp/ks.log 2>&1; python3 -c "
import csv
d={r['Name'][6:36]:(int(r['Calls']),float(r['AverageNs'])/1000) for r in csv.DictReader(open('/tmp/ks/x_kernel_stats.csv')) if 'pass_' in r['Name']}
print(' | '.join(f'{k[5:22]} {t:7.1f}' for k,(c,t) in sorted(d.items())), '| edge passes per fwd', round(sum(c*t for c,t in d.values())/(max(c for c,_ in d.values())/3),1))"; }
for v in "-DPA_NT=11 -DPB_NT=1" "-DPA_NT=15 -DPB_NT=1" "-DPA_NT=11 -DPB_NT=1 -DPC_NT=1" "-DPA_NT=15 -DPB_NT=1 -DPC_NT=1"; do
  (cd $CS && rm -f edge_kernels.o && make -s EXTRA="$v" libmtmc_mpn.so > /dev/null 2>&1)
  echo "== $v"; run cfg4 20; run cfg5 3
done
(cd $CS && rm -f edge_kernels.o && make -s libmtmc_mpn.so > /dev/null 2>&1)
