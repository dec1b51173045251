#!/bin/bash
# Round 3, step 1: what the large-table regime really is.
#   (b) the footprint microbenchmark by access shape, plain and under --pmc FETCH_SIZE
#   (a) FETCH_SIZE / WRITE_SIZE of the search kernel on bench.py --config 2 (10M x 768d)
# Run on the GPU box:  bash scripts/r03/gather_shapes.sh [a|b|ab]
set -o pipefail
WHAT=${1:-ab}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r03_gather
mkdir -p $OUT
export TMPDIR=/tmp
BIN=$REPO/scripts/micro/bin/footprint_gather
if [[ $WHAT == *b* ]]; then
  $BIN 64 0 16 > $OUT/footprint_plain.txt 2>&1 || { tail -3 $OUT/footprint_plain.txt; exit 1; }
  tail -30 $OUT/footprint_plain.txt
  cd /tmp
  for GB in 16; do
    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch_micro_$GB -- $BIN $GB $GB 8 > $OUT/footprint_pmc_$GB.txt 2>&1 || { tail -5 $OUT/footprint_pmc_$GB.txt; exit 1; }
  done
  cd $REPO
  python - <<PY
import csv, glob, collections
for f in glob.glob("$OUT/pmc_fetch_micro_*/**/*counter_collection.csv", recursive=True):
    acc = collections.OrderedDict()
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != "FETCH_SIZE": continue
        acc.setdefault((r["Kernel_Name"], r["Grid_Size"]), []).append(float(r["Counter_Value"]))
    print(f)
    for (k, g), v in acc.items():
        # the timed launch is the last of each (kernel, grid); its algorithmic bytes: grid lanes x iters(8) x ROW
        row = 512 if "<512" in k or "512," in k.split("<")[1][:5] else 3072
        alg = int(g) * 8 * row
        print("%-60s grid %7s  FETCH_SIZE %10.1f MB raw (x2: %10.1f MB)  algorithmic %8.1f MB  raw/alg %.2f" % (k[:60], g, v[-1] / 1024, 2 * v[-1] / 1024, alg / 1e6, v[-1] * 1024 / alg))
PY
fi
if [[ $WHAT == *a* ]]; then
  ARGS="--config 2 --steps 10 --warmup 2 --no-cpu-baseline --no-concurrent --no-extras --recall-queries 256"
  cd /tmp
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --kernel-include-regex "hx_search_kernel|hx_lean|hx_wide" --output-format csv -d $OUT/pmc_fetch_c2 -- python $REPO/bench.py $ARGS > $OUT/c2_fetch.json 2> $OUT/c2_fetch.err || { tail -5 $OUT/c2_fetch.err; exit 1; }
  tail -c 600 $OUT/c2_fetch.json
  cd $REPO
  python - <<PY
import csv, glob, collections
for f in glob.glob("$OUT/pmc_fetch_c2/**/*counter_collection.csv", recursive=True):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        acc[(r["Kernel_Name"][:70], r["Counter_Name"], r["Grid_Size"])].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        print(k, "launches", len(v), "mean of last 10: %.1f MB raw" % (sum(v[-10:]) / len(v[-10:]) / 1024))
PY
fi
