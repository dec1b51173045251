#!/bin/bash
# SQ instruction / wait counters of the search kernel (diagnostic; gpurun)
set -o pipefail
KIND=${1:-quant8}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/sq_${KIND}_${2:-64}
mkdir -p $OUT
export TMPDIR=/tmp
EF=${2:-64}
ARGS="--steps 50 --warmup 5 --kind $KIND --no-secondary --no-cpu-baseline --no-concurrent --no-extras --recall-queries 1024 --ef $EF"
cd $REPO && python bench.py $ARGS > /dev/null 2> $OUT/warm.err   # builds + caches the index
cd /tmp
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD --output-format csv -d $OUT/p1 -- python $REPO/bench.py $ARGS > $OUT/p1.json 2> $OUT/p1.err || tail -3 $OUT/p1.err
rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS --output-format csv -d $OUT/p2 -- python $REPO/bench.py $ARGS > $OUT/p2.json 2> $OUT/p2.err || tail -3 $OUT/p2.err
cd $REPO
python - <<PY
import csv, glob, collections, json
doc = {}
for p in ("p1","p2"):
    f = glob.glob("$OUT/%s/**/*counter_collection.csv" % p, recursive=True)
    if not f: print("no csv for", p); continue
    rows = [r for r in csv.DictReader(open(f[0])) if ("hx_search_kernel" in r["Kernel_Name"] or "hx_lean_" in r["Kernel_Name"])]
    cnt = collections.Counter(r["Kernel_Name"] for r in rows)
    timed = cnt.most_common(1)[0][0]   # the timed efSearch's instantiation
    acc = collections.defaultdict(list)
    for r in rows:
        if r["Kernel_Name"] == timed:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        v = v[-50:]
        print("%-24s mean per launch %14.0f   per wave %10.1f" % (k, sum(v)/len(v), sum(v)/len(v)/1024))
        doc[k] = sum(v)/len(v)
    doc["kernel"] = timed[:80]
json.dump(doc, open("$OUT/sq.json", "w"), indent=1)
PY
