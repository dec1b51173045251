#!/bin/bash
# SQ counters of the insert kernel over a 1M on-device build (gpurun)
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r03/build_pmc
mkdir -p $OUT
export TMPDIR=/tmp
export PYTHONPATH=$REPO
cd /tmp
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --kernel-include-regex "hx_insert_kernel" --output-format csv -d $OUT/p1 -- python $REPO/scripts/gpu_build_perf.py ${1:-1000000} ${2:-0} > $OUT/run1.log 2>&1 || tail -5 $OUT/run1.log
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VMEM_WR SQ_INSTS_SMEM --kernel-include-regex "hx_insert_kernel" --output-format csv -d $OUT/p2 -- python $REPO/scripts/gpu_build_perf.py ${1:-1000000} ${2:-0} > $OUT/run2.log 2>&1 || tail -5 $OUT/run2.log
cd $REPO
python - <<PY
import csv, glob, collections
for p in ("p1", "p2"):
    f = glob.glob("$OUT/%s/**/*counter_collection.csv" % p, recursive=True)
    if not f: print("no csv", p); continue
    acc = collections.defaultdict(float); waves = 0
    for r in csv.DictReader(open(f[0])):
        if "hx_insert_kernel" in r["Kernel_Name"] and r["Grid_Size"] == "524288":
            acc[r["Counter_Name"]] += float(r["Counter_Value"])
    n = acc.get("SQ_WAVES", 0) or None
    for k, v in sorted(acc.items()):
        print("%-24s total %16.0f" % (k, v) + ("   per wave %12.1f" % (v / n) if n else ""))
    if n: print("full batches counted:", n / 8192)
PY
