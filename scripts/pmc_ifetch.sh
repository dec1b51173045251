#!/bin/bash
# Where does a lone wave wait for instructions?  Branch count, instruction fetches and I-cache hits / misses of the timed search
# kernel (diagnostic; gpurun):  bash scripts/pmc_ifetch.sh f32|quant8 [ef]
set -o pipefail
KIND=${1:-f32}
EF=${2:-68}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/ifetch_${KIND}_${EF}
mkdir -p $OUT
export TMPDIR=/tmp
ARGS="--steps 50 --warmup 5 --kind $KIND --no-secondary --no-cpu-baseline --no-concurrent --no-extras --recall-queries 1024 --ef $EF"
cd $REPO && python bench.py $ARGS > /dev/null 2> $OUT/warm.err
cd /tmp
rocprofv3 --kernel-trace --pmc SQ_INSTS_BRANCH SQ_IFETCH SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_SALU --output-format csv -d $OUT/p1 -- python $REPO/bench.py $ARGS > $OUT/p1.json 2> $OUT/p1.err || tail -3 $OUT/p1.err
cd $REPO
python - <<PY
import csv, glob, collections
f = glob.glob("$OUT/p1/**/*counter_collection.csv", recursive=True)
rows = [r for r in csv.DictReader(open(f[0])) if "hx_lean_" in r["Kernel_Name"] or "hx_search_kernel" in r["Kernel_Name"]]
cnt = collections.Counter(r["Kernel_Name"] for r in rows)
timed = cnt.most_common(1)[0][0]
acc = collections.defaultdict(list)
for r in rows:
    if r["Kernel_Name"] == timed: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
print(timed[:90])
for k, v in sorted(acc.items()):
    v = v[-50:]
    print("%-22s mean per launch %14.0f   per query %10.1f" % (k, sum(v)/len(v), sum(v)/len(v)/1024))
PY
rm -rf $OUT/p1
