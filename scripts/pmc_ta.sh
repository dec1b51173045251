#!/bin/bash
# Texture-addresser / L1 (TA, TCP, TD) counters of the timed search kernel: how busy the CU's address path and tag
# lookup are at batch 1024 (diagnostic; gpurun).  Counters alone with --kernel-trace, a few per pass, program directly
# after `--`.   bash scripts/pmc_ta.sh f32 68
set -o pipefail
KIND=${1:-f32}
EF=${2:-68}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/ta_${KIND}_${EF}
mkdir -p $OUT
export TMPDIR=/tmp
ARGS="--steps 50 --warmup 5 --kind $KIND --no-secondary --no-cpu-baseline --no-concurrent --no-extras --recall-queries 1024 --ef $EF"
cd $REPO && python bench.py $ARGS > $OUT/plain.json 2> $OUT/warm.err   # builds + caches the index
cd /tmp
i=0
# (two counters of one block per pass: three TA counters are refused -- "exceeds the capabilities of the hardware" -- and
# rocprofv3 then hangs in its abort handler, hence the timeout around every pass)
for SET in "TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum GRBM_GUI_ACTIVE" \
           "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" \
           "TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_GATE_EN1_sum" \
           "TD_TD_BUSY_sum TD_TC_STALL_sum"; do
    i=$((i+1))
    timeout -k 5 150 rocprofv3 --kernel-trace --pmc $SET --kernel-include-regex "hx_lean_|hx_search_kernel" --output-format csv -d $OUT/p$i -- python $REPO/bench.py $ARGS > $OUT/p$i.json 2> $OUT/p$i.err || tail -3 $OUT/p$i.err
    echo "[pmc_ta] pass $i done" >&2
done
cd $REPO
python - <<PY
import csv, glob, collections, json
doc = {}
for p in range(1, 7):
    f = glob.glob("$OUT/p%d/**/*counter_collection.csv" % p, recursive=True)
    if not f: print("no csv for pass", p); continue
    rows = [r for r in csv.DictReader(open(f[0])) if r.get("Grid_Size", r.get("Grid_Size_X")) == str(64 * 1024)]
    cnt = collections.Counter(r["Kernel_Name"] for r in rows)
    if not cnt: print("no rows for pass", p); continue
    timed = cnt.most_common(1)[0][0]
    acc = collections.defaultdict(list)
    for r in rows:
        if r["Kernel_Name"] == timed:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        v = v[-50:]
        print("%-44s mean per launch %16.0f" % (k, sum(v)/len(v)))
        doc[k] = sum(v)/len(v)
    doc["kernel"] = timed[:80]
    try:
        doc.setdefault("kernel_ms_under_counters", {})["p%d" % p] = json.loads(open("$OUT/p%d.json" % p).read().strip().splitlines()[-1])["roofline"]["kernel_ms"]
    except Exception as e:
        print("no bench line for pass", p, e)
json.dump(doc, open("$OUT/ta.json", "w"), indent=1)
PY
rm -rf $OUT/p1 $OUT/p2 $OUT/p3 $OUT/p4 $OUT/p5 $OUT/p6
