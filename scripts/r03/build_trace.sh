#!/bin/bash
# kernel trace of a 1M on-device build: insert-kernel duration per batch (gpurun)
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r03/build_trace
mkdir -p $OUT
export TMPDIR=/tmp
export PYTHONPATH=$REPO
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT -- python $REPO/scripts/gpu_build_perf.py ${1:-1000000} ${2:-0} > $OUT/run.log 2>&1 || tail -5 $OUT/run.log
cd $REPO
python - <<PY
import csv, glob
f = glob.glob("$OUT/**/*kernel_trace.csv", recursive=True)
rows = [r for r in csv.DictReader(open(f[0])) if "hx_insert_kernel" in r["Kernel_Name"]]
print(len(rows), "insert launches; VGPR", rows[0]["VGPR_Count"], "LDS", rows[0]["LDS_Block_Size"], "scratch", rows[0]["Scratch_Size"])
tot = 0
for i, r in enumerate(rows):
    ms = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    tot += ms
    if i % 8 == 0: print("batch %3d grid %8s  %.3f ms" % (i, r.get("Grid_Size", r.get("Grid_Size_X")), ms))
print("total insert kernel ms", tot)
PY
