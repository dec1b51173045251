#!/bin/bash
# A/B helper: builds the kernels of a commit (default HEAD) into hnsw_rs_amd/libhnsw_<NAME>.so (default A)
# next to the working-tree library, so that one gpurun call can time both (HNSW_MI355X_LIB selects the
# library).   scripts/ab_build.sh [commit] [name]
set -e
REPO=$(cd "$(dirname "$0")/.." && pwd)
COMMIT=${1:-HEAD}
NAME=${2:-A}
TMP=$(mktemp -d)
git -C $REPO archive $COMMIT hnsw_rs_amd/csrc include | tar -x -C $TMP
make -C $TMP/hnsw_rs_amd/csrc -j4 OUT=$REPO/hnsw_rs_amd/libhnsw_$NAME.so > /dev/null
rm -rf $TMP
ls -la $REPO/hnsw_rs_amd/libhnsw_$NAME.so
