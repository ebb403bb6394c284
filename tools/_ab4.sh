set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "parity or regimes or random_configs or sharded" 2>&1 | tail -3
for b in 768 1024 1536 2048; do
  echo "== MTMC_PASS_C_BLOCKS=$b"
  MTMC_PASS_C_BLOCKS=$b DETAIL=1 python tools/phase_ab.py cfg4 10 2>&1 | grep "per launch" | grep -o "'pass_c\[[0-9]\]': [0-9.]*"
done
for s in 4 16; do
  echo "== MTMC_PASS_C_SPAN=$s"
  MTMC_PASS_C_SPAN=$s DETAIL=1 python tools/phase_ab.py cfg4 10 2>&1 | grep "per launch" | grep -o "'pass_c\[[0-9]\]': [0-9.]*"
done
DETAIL=1 python tools/phase_ab.py s02 200 2>&1 | grep -v amdgpu.ids | head -3
DETAIL=1 python tools/phase_ab.py s02_tracker 100 2>&1 | grep -v amdgpu.ids | head -3
