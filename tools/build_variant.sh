#!/bin/bash
# Build a library variant for A/B runs on the GPU box (no GPU needed: hipcc cross-compiles gfx950):
#     tools/build_variant.sh NAME [-DSWITCH ...]          -> variants/NAME.so  (the working tree + the switches)
#     tools/build_variant.sh NAME --rev GITREV [-D...]    -> variants/NAME.so  (the sources of a commit, e.g. HEAD~3)
# variants/ is git-ignored and travels with the gpurun snapshot; time a variant with MPC_LQR_HIP_LIB=$PWD/variants/NAME.so.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; shift
SRC=$ROOT
if [ "$1" = "--rev" ]; then
  REV=$2; shift 2
  SRC=/tmp/variant_src_$NAME; rm -rf $SRC; mkdir -p $SRC
  git -C $ROOT archive $REV mpc.pytorch_amd/csrc include | tar -x -C $SRC
fi
BUILD=/tmp/variant_build_$NAME; rm -rf $BUILD; mkdir -p $BUILD/csrc $BUILD/mpc $BUILD/../include 2>/dev/null || true
mkdir -p $BUILD/mpc.pytorch_amd
rm -rf $BUILD/mpc.pytorch_amd/csrc; cp -r $SRC/mpc.pytorch_amd/csrc $BUILD/mpc.pytorch_amd/csrc
rm -rf $BUILD/include; cp -r $SRC/include $BUILD/include
mkdir -p $BUILD/mpc.pytorch_amd/mpc $ROOT/variants
rm -f $BUILD/mpc.pytorch_amd/csrc/*.o
make -C $BUILD/mpc.pytorch_amd/csrc -j8 OUT=$ROOT/variants/$NAME.so \
     CXXFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -ffp-contract=fast $*" > $BUILD/make.log 2>&1 \
  || { tail -20 $BUILD/make.log; exit 1; }
ls -la $ROOT/variants/$NAME.so
