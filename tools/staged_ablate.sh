#!/bin/bash
# Where does a k-tile of the role-split GEMM (csrc/gemm_staged.hip) go?  Rebuilds the kernel with one part removed at a time
# (-DSG_ABL=n: 1 no conversion arithmetic, 2 no global loads of A, 3 no MFMAs, 4 no LDS-DMA of W, 5 no fragment reads,
# 6 producers idle) and times the whole raw call (|A|max + W split + GEMM: only the GEMM changes).  Run on a GPU box:
#   bash tools/staged_ablate.sh "100000 1024 512" "100000 512 128"
# Leaves the product library rebuilt without ablation.
set -e
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
CS=$ROOT/graph-convolutional-network-for-multi-camera-vehicle-tracking_amd/csrc
for abl in 0 1 2 3 4 5 6 0; do
  (cd $CS && rm -f gemm_staged.o && make -s EXTRA="-DSG_ABL=$abl" libmtmc_mpn.so > /dev/null)
  for shape in "$@"; do
    echo "SG_ABL=$abl  $(python3 $ROOT/tools/staged_time.py $shape | tail -1)"
  done
done
