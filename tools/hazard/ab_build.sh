#!/bin/bash
# Root-cause experiment for the gemm_bn_f16x3 LDS-store hazard: builds variants of gemm_bn.hip into
# tools/_hazard/<name>/libmtmc_mpn.so (the other objects are the product's) -- run `tools/hazard_run.sh` on the GPU.
set -e
cd "$(dirname "$0")/../../graph-convolutional-network-for-multi-camera-vehicle-tracking_amd/csrc"
make -j4 >/dev/null
OUT=../../tools/_hazard
mkdir -p $OUT
build() {  # name, flags...
  name=$1; shift
  mkdir -p $OUT/$name
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -Wno-unused-function "$@" -c gemm_bn.hip -o $OUT/$name/gemm_bn.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $(ls *.o | grep -v '^gemm_bn.o$') $OUT/$name/gemm_bn.o -o $OUT/$name/libmtmc_mpn.so
  rm $OUT/$name/gemm_bn.o
  echo built $name
}
build rne        -DMTMC_F16_CVT_RNE
build rne_nop0   -DMTMC_F16_CVT_RNE -DMTMC_DS_GUARD_NOP=0
build rne_nop1   -DMTMC_F16_CVT_RNE -DMTMC_DS_GUARD_NOP=1
build rne_nop3   -DMTMC_F16_CVT_RNE -DMTMC_DS_GUARD_NOP=3
build rne_nop7   -DMTMC_F16_CVT_RNE -DMTMC_DS_GUARD_NOP=7
build rne_split  -DMTMC_F16_CVT_RNE -DMTMC_DS_NO_WRITE2
build pk         
build pk_nop1    -DMTMC_DS_GUARD_NOP=1
