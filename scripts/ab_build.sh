#!/bin/bash
# A/B helper: builds the committed (HEAD) kernels into hnsw_rs_amd/libhnsw_A.so next to the working-tree
# library, so that one gpurun call can time both (HNSW_MI355X_LIB selects the library).
set -e
REPO=$(cd "$(dirname "$0")/.." && pwd)
TMP=$(mktemp -d)
git -C $REPO archive HEAD hnsw_rs_amd/csrc include | tar -x -C $TMP
make -C $TMP/hnsw_rs_amd/csrc -j4 OUT=$REPO/hnsw_rs_amd/libhnsw_A.so > /dev/null
rm -rf $TMP
ls -la $REPO/hnsw_rs_amd/libhnsw_A.so
