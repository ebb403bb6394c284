set -e
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q -k "parity or training or random_configs or regimes or gemm" > gpurun_out/r3_t3.log 2>&1 || { tail -30 gpurun_out/r3_t3.log; exit 1; }
tail -3 gpurun_out/r3_t3.log
DETAIL=1 python tools/phase_ab.py cfg4 10 2>&1 | tee gpurun_out/r3_ab2.log
ROOT=$GRAFT_REPO_ROOT; OUT=$ROOT/gpurun_out/prof; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/cfg4_stats -o cfg4 --output-format csv -- python3 $ROOT/tools/fwd_loop.py cfg4 20 > $OUT/cfg4_stats.log 2>&1
python3 $ROOT/tools/trim_stats.py $OUT/cfg4_stats/cfg4_kernel_stats.csv $ROOT/gpurun_out/r03a_cfg4_kernel_stats.csv
cat $ROOT/gpurun_out/r03a_cfg4_kernel_stats.csv
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE -d $OUT/cfg4_W -o cfg4 --output-format csv -- python3 $ROOT/tools/fwd_loop.py cfg4 5 > $OUT/cfg4_W.log 2>&1
python3 $ROOT/tools/pmc_summary.py $OUT/cfg4_W/cfg4_counter_collection.csv WRITE_SIZE | grep -v "at::native\|rocprim\|rocclr" | tee $ROOT/gpurun_out/r03a_cfg4_pmc_WRITE_SIZE.txt
