cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/ks
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d /tmp/ks -o x --output-format csv -- python3 $R/tools/fwd_loop.py ${1:-cfg4} ${2:-20} > /tmp/ks.log 2>&1
python3 $R/tools/trim_stats.py /tmp/ks/x_kernel_stats.csv /tmp/ks/trim.csv && head -${3:-16} /tmp/ks/trim.csv | cut -c1-120
