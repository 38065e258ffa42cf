#!/bin/bash
# Builds a diagnostic variant of the library from the working tree:
#   tools/build_variant.sh NAME "-DCG_EXP=1"   ->  gpurun_ab/lib_NAME.so   (a laboratory build: -DCOMPEG_LAB, see lab.h)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
W=/tmp/variant_$1
rm -rf $W && mkdir -p $W/compeg_amd $W/include $ROOT/gpurun_ab
cp -r $ROOT/compeg_amd/csrc $W/compeg_amd/csrc && rm -rf $W/compeg_amd/csrc/build
cp $ROOT/include/*.h $W/include/
make -C $W/compeg_amd/csrc -s OUT=$ROOT/gpurun_ab/lib_$1.so \
  CXXFLAGS="${OPT:--O3} -std=c++17 -fPIC -ffp-contract=off -fno-signed-zeros -fvisibility=hidden -DCOMPEG_LAB $2"
