#!/bin/bash
# Build tools/_hazard/<name>/libmtmc_mpn.so whose gemm_bn device code comes from a patched ISA listing.
#   isa_build.sh NAME "EXTRA_CXXFLAGS" [PATCH_MODE ...]     (patch modes: see isa_patch.py; applied in order)
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
CSRC="$HERE/../../graph-convolutional-network-for-multi-camera-vehicle-tracking_amd/csrc"
OUT="$HERE/../_hazard"; NAME=$1; EXTRA=$2; shift 2
LLVM=/opt/rocm/lib/llvm/bin
W=$(mktemp -d); mkdir -p "$OUT/$NAME"
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -Wno-unused-function -Wno-unused-command-line-argument $EXTRA -I$CSRC"
KERN=${KERN:-gemm_bn_f16x3_kernelILi2ELi2ELi64E}
/opt/rocm/bin/hipcc $FLAGS -S --cuda-device-only "$CSRC/gemm_bn.hip" -o $W/dev0.s
i=0
for mode in "$@"; do python3 "$HERE/isa_patch.py" $W/dev$i.s $W/dev$((i+1)).s "$KERN" "$mode"; i=$((i+1)); done
$LLVM/clang -x assembler -target amdgcn-amd-amdhsa -mcpu=gfx950 -c $W/dev$i.s -o $W/dev.o
$LLVM/lld -flavor gnu -m elf64_amdgpu --no-undefined -shared -o $W/dev.out $W/dev.o
$LLVM/clang-offload-bundler -type=o -bundle-align=4096 -targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--gfx950 -input=/dev/null -input=$W/dev.out -output=$W/dev.hipfb
/opt/rocm/bin/hipcc $FLAGS --cuda-host-only -Xclang -fcuda-include-gpubinary -Xclang $W/dev.hipfb -c "$CSRC/gemm_bn.hip" -o $W/gemm_bn.o
(cd "$CSRC" && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $(ls *.o | grep -v '^gemm_bn.o$') $W/gemm_bn.o -o "$OUT/$NAME/libmtmc_mpn.so")
cp $W/dev$i.s "$OUT/$NAME/gemm_bn.s"
rm -rf $W
echo built $NAME
