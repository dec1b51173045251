#!/bin/bash
# Profiling recipe (run on the GPU box through gpurun):
#   bash scripts/profile.sh r01
# 1. plain bench (builds and caches both indexes under /tmp); then per vector kind:
# 2. rocprofv3 kernel trace + stats of the same command, 3. PMC passes (FETCH_SIZE, WRITE_SIZE) in
# their own runs.  Summaries land in gpurun_out/<tag>/ ; copy what should be judged into profiles/.
set -o pipefail
TAG=${1:-r01}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ARGS="--steps 100 --warmup 10"
cd $REPO
python bench.py $ARGS 2> $OUT/bench.err | tee $OUT/bench.json || exit 1
for KIND in f32 quant8; do
  # the efSearch the plain run settled on; a 1024-query recall check keeps the traced launches uniform
  EF=$(python -c "import json,sys; b=json.load(open('$OUT/bench.json')); print(b['config']['efSearch'] if '$KIND'=='f32' else b['quant8_reference_default']['efSearch'])")
  P="$ARGS --kind $KIND --no-secondary --no-cpu-baseline --no-concurrent --no-extras --ef $EF --recall-queries 1024"
  cd /tmp
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$KIND -- python $REPO/bench.py $P > $OUT/trace_$KIND.json 2> $OUT/trace_$KIND.err || { tail -5 $OUT/trace_$KIND.err; exit 1; }
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch_$KIND -- python $REPO/bench.py $P > $OUT/pmc_fetch_$KIND.json 2> $OUT/pmc_fetch_$KIND.err || { tail -5 $OUT/pmc_fetch_$KIND.err; exit 1; }
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write_$KIND -- python $REPO/bench.py $P > $OUT/pmc_write_$KIND.json 2> $OUT/pmc_write_$KIND.err || { tail -5 $OUT/pmc_write_$KIND.err; exit 1; }
  cd $REPO
done
python scripts/summarize_profile.py $OUT > $OUT/summary.txt
cat $OUT/summary.txt
