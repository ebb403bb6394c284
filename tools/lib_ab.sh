#!/bin/bash
# Same-box A/B of two builds of the library (box-to-box variation is +-3-5 %, more than most single changes are worth):
#   bash tools/lib_ab.sh <old libmtmc_mpn.so> [workload ...]     (MTMC_MPN_LIB selects the build; same ABI required)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OLD=$1; shift
for wl in "${@:-s02}"; do
  steps=100; [ $wl = cfg4 ] && steps=20; [ $wl = cfg5 ] && steps=6
  for rep in 1 2 3; do
    for which in old new; do
      if [ $which = old ]; then export MTMC_MPN_LIB=$OLD; else unset MTMC_MPN_LIB; fi
      python3 $ROOT/bench.py --workload $wl --steps $steps --warmup 10 --no-cpu --no-stress 2>/dev/null | python3 -c "
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$wl $which: %.4f ms  (median %.4f, p10 %.4f)  eager %.4f  replay %s' % (d['ms_per_step'], d['step_ms']['median'], d['step_ms']['p10'], d['launch']['eager_ms'], d['launch']['graph_replay_ms']))
"
    done
  done
done
