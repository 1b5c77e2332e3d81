#!/bin/bash
# Build a variant of libgmmvb.so with extra compiler flags next to the real one (for A/B runs with BAYESML_AMD_LIB):
#   tools/build_variant.sh rb80 -DGMMVB_RELEVANCE_BITS=80   ->   bayesml_amd/csrc/libgmmvb_rb80.so
set -eu
NAME=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
TMP=$(mktemp -d)
mkdir -p $TMP/bayesml_amd/csrc $TMP/include
cp $ROOT/bayesml_amd/csrc/*.h $ROOT/bayesml_amd/csrc/*.hip $ROOT/bayesml_amd/csrc/Makefile $TMP/bayesml_amd/csrc/
cp $ROOT/include/*.h $TMP/include/
make -C $TMP/bayesml_amd/csrc -j8 CXXFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function $*" > /dev/null
cp $TMP/bayesml_amd/csrc/libgmmvb.so $ROOT/bayesml_amd/csrc/libgmmvb_$NAME.so
rm -rf $TMP
echo built bayesml_amd/csrc/libgmmvb_$NAME.so
