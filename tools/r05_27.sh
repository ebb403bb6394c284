#!/bin/bash
# round 5: kernel time per tracker-scale forward (rocprofv3) for the regime-threshold variants; backward sweep seed 57
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
cd $ROOT
python3 tools/dbg_bwd_seed.py 57 > $OUT/r05_dbg57.txt 2>&1
cd /tmp && export TMPDIR=/tmp
for v in default se1m se1m_cap4k; do
  if [ $v = default ]; then unset MTMC_MPN_LIB; else export MTMC_MPN_LIB=$ROOT/build_ab/$v/pkg/csrc/libmtmc_mpn.so; fi
  timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/prof/se_$v -o t --output-format csv -- python3 $ROOT/tools/fwd_loop.py s02_tracker 100 > $OUT/se_$v.log 2>&1
  python3 $ROOT/tools/trim_stats.py $OUT/prof/se_$v/t_kernel_stats.csv $OUT/r05_tracker_${v}_kernel_stats.csv
  echo "$v done"
done
