#!/bin/bash
# A/B of the two forms of the role-split GEMM -- the product's gemm_staged_kernel<256,2> and the laboratory's
# gemm_staged_w_kernel (csrc/lab/staged2_lab.hip; STAGED_LAB=1 makes tools/staged_time.py call it) -- on the layer-1 shapes of
# configs 4 / 5.  Run on a GPU box:  bash tools/staged2_ab.sh
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
for shape in "100000 1024 512" "1000000 1024 512" "125000 1024 512" "100000 2048 1024"; do
  for v in 0 1 0 1; do
    echo "STAGED_LAB=$v $shape: $(STAGED_LAB=$v python3 $ROOT/tools/staged_time.py $shape | tr '\n' ' ')"
  done
done
