set -e
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q -k "parity or training or random_configs or regimes" > gpurun_out/r3_t2.log 2>&1 || { tail -30 gpurun_out/r3_t2.log; exit 1; }
tail -3 gpurun_out/r3_t2.log
for e in 4 2 1; do
  (cd graph-convolutional-network-for-multi-camera-vehicle-tracking_amd/csrc && touch edge_kernels.hip && make EXTRA=-DMTMC_PASS_A_EPT=$e > /dev/null 2>&1)
  echo "== PASS_A_EPT=$e" | tee -a gpurun_out/r3_ab_a.log
  DETAIL=1 python tools/phase_ab.py cfg4 10 2>&1 | tee -a gpurun_out/r3_ab_a.log
done
(cd graph-convolutional-network-for-multi-camera-vehicle-tracking_amd/csrc && touch edge_kernels.hip && make > /dev/null 2>&1)
DETAIL=1 python tools/phase_ab.py s02 200 2>&1 | tee -a gpurun_out/r3_ab_a.log
DETAIL=1 python tools/phase_ab.py s02_tracker 100 2>&1 | tee -a gpurun_out/r3_ab_a.log
