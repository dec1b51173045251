#!/bin/bash
# Profiling recipe (run on the GPU box through gpurun):
#   bash scripts/profile.sh r01
# 1. plain bench (builds and caches the index under /tmp), 2. rocprofv3 kernel trace + stats of
# the same command, 3. PMC passes (FETCH_SIZE, WRITE_SIZE) in their own runs.
# Summaries land in gpurun_out/<tag>/ ; copy what should be judged into profiles/.
set -o pipefail
TAG=${1:-r01}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ARGS="--steps 100 --warmup 10"
cd $REPO
python bench.py $ARGS 2> $OUT/bench.err | tee $OUT/bench.json || exit 1
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python $REPO/bench.py $ARGS --no-cpu-baseline > $OUT/trace.json 2> $OUT/trace.err || { tail -5 $OUT/trace.err; exit 1; }
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python $REPO/bench.py $ARGS --no-cpu-baseline > $OUT/pmc_fetch.json 2> $OUT/pmc_fetch.err || { tail -5 $OUT/pmc_fetch.err; exit 1; }
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python $REPO/bench.py $ARGS --no-cpu-baseline > $OUT/pmc_write.json 2> $OUT/pmc_write.err || { tail -5 $OUT/pmc_write.err; exit 1; }
cd $REPO
python scripts/summarize_profile.py $OUT > $OUT/summary.txt
cat $OUT/summary.txt
