#!/bin/bash
# Profiling recipe for BASELINE configs[3] / configs[4] on the one GPU of a gpurun box:
#   bash scripts/profile_cfg.sh r04 3                      (100M x 128d, the per-GPU workload of configs[3])
#   bash scripts/profile_cfg.sh r04 4 --n-points 16000000  (configs[4]'s build at a size whose passes fit the call)
# 1. the plain bench line; 2. FETCH_SIZE and WRITE_SIZE, each in its own pass (counters alone with --kernel-trace, as
# the pool requires), restricted to the kernels of interest: the timed search kernel, and for configs[4] the build's
# hx_insert_kernel (all its launches summed: the build is the "step" there).  Program directly after `--`.
# Output: gpurun_out/<tag>_c<N>/ ; traffic_c<N>.json holds the entries for profiles/traffic_latest.json.
set -o pipefail
TAG=${1:-r04}
CFG=${2:-3}
shift 2
EXTRA="$@"
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/${TAG}_c${CFG}
mkdir -p $OUT
export TMPDIR=/tmp
cd $REPO
python bench.py --config $CFG $EXTRA 2> $OUT/bench.err > $OUT/bench.json || { tail -5 $OUT/bench.err; exit 1; }
echo "[profile_cfg] plain line done" >&2
EF=$(python -c "import json; print(json.load(open('$OUT/bench.json'))['config']['efSearch'])")
P="--config $CFG $EXTRA --steps 20 --warmup 4 --no-cpu-baseline --no-concurrent --no-extras --recall-queries 256 --ef $EF"
RX="hx_lean_f32_kernel|hx_search_kernel"
if [ "$CFG" = "4" ]; then RX="$RX|hx_insert_kernel"; fi
cd /tmp
rocprofv3 --kernel-trace --pmc FETCH_SIZE --kernel-include-regex "$RX" --output-format csv -d $OUT/pmc_fetch -- python $REPO/bench.py $P > $OUT/pmc_fetch.json 2> $OUT/pmc_fetch.err || { tail -5 $OUT/pmc_fetch.err; exit 1; }
echo "[profile_cfg] FETCH_SIZE pass done" >&2
rocprofv3 --kernel-trace --pmc WRITE_SIZE --kernel-include-regex "$RX" --output-format csv -d $OUT/pmc_write -- python $REPO/bench.py $P > $OUT/pmc_write.json 2> $OUT/pmc_write.err || { tail -5 $OUT/pmc_write.err; exit 1; }
echo "[profile_cfg] WRITE_SIZE pass done" >&2
cd $REPO
python scripts/summarize_cfg_profile.py $OUT $CFG > $OUT/summary.txt
# the raw traces of a run with tens of thousands of build dispatches are hundreds of MB: gpurun merges back 64 MiB at most
rm -rf $OUT/pmc_fetch $OUT/pmc_write
cut -c1-600 $OUT/summary.txt
