#!/bin/bash
# A second build of the library with extra compiler flags, for same-box A/B runs (MTMC_MPN_LIB selects it):
#   bash tools/build_variant.sh NAME "-DFLAG=1 ..."   ->  build_ab/NAME/pkg/csrc/libmtmc_mpn.so
# (built in the container or on the GPU box; the .so is git-ignored and travels with gpurun)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; shift
DST=$ROOT/build_ab/$NAME
rm -rf $DST && mkdir -p $DST/pkg/csrc/lab $DST/tools $DST/include
cp $ROOT/graph-convolutional-network-for-multi-camera-vehicle-tracking_amd/csrc/*.hip $ROOT/graph-convolutional-network-for-multi-camera-vehicle-tracking_amd/csrc/*.h \
   $ROOT/graph-convolutional-network-for-multi-camera-vehicle-tracking_amd/csrc/Makefile $DST/pkg/csrc/
cp $ROOT/include/*.h $DST/include/
cp $ROOT/tools/check_isa.py $DST/tools/
make -C $DST/pkg/csrc -j${JOBS:-8} EXTRA="$*" libmtmc_mpn.so > $DST/build.log 2>&1 || { tail -20 $DST/build.log; exit 1; }
rm -f $DST/pkg/csrc/*.o
echo $DST/pkg/csrc/libmtmc_mpn.so
