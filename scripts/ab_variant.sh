#!/bin/bash
# A/B helper: builds the WORKING TREE kernels with extra defines into hnsw_rs_amd/libhnsw_<NAME>.so
#   scripts/ab_variant.sh B -DHX_MERGE_SHIFT_MAX=1
set -e
REPO=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; shift
TMP=$(mktemp -d)
mkdir -p $TMP/hnsw_rs_amd
cp -r $REPO/hnsw_rs_amd/csrc $TMP/hnsw_rs_amd/csrc
cp -r $REPO/include $TMP/include
rm -f $TMP/hnsw_rs_amd/csrc/*.o
make -C $TMP/hnsw_rs_amd/csrc -j4 OUT=$REPO/hnsw_rs_amd/libhnsw_$NAME.so EXTRA="$*" > /dev/null
rm -rf $TMP
ls -la $REPO/hnsw_rs_amd/libhnsw_$NAME.so
