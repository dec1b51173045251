"""Condense a scripts/profile.sh output directory into a small text summary (kernel stats +
PMC-derived HBM traffic per launch of the search kernel, per vector kind) and write
traffic.json in the format bench.py reads from profiles/traffic_latest.json."""
import csv
import glob
import json
import os
import sys

out = sys.argv[1]


def is_search(name):
    return "hx_search_kernel" in name or "hx_lean_f32_kernel" in name or "hx_lean_q8_kernel" in name


def find(pattern):
    r = glob.glob(os.path.join(out, pattern), recursive=True)
    return r[0] if r else None


print("== bench line (python bench.py --steps 100 --warmup 10) ==")
p = os.path.join(out, "bench.json")
if os.path.exists(p):
    print(open(p).read().strip())
entries = []
for kind in ("f32", "quant8"):
    print("\n==================== vector kind %s ====================" % kind)
    tj = os.path.join(out, "trace_%s.json" % kind)
    line = None
    if os.path.exists(tj):
        try:
            line = json.loads(open(tj).read().strip().splitlines()[-1])
            print("bench under rocprofv3: value %.0f q/s, kernel_ms %.5f, efSearch %d" % (
                line["value"], line["roofline"]["kernel_ms"], line["config"]["efSearch"]))
        except (ValueError, KeyError, IndexError):
            pass
    st = find("trace_%s/**/*kernel_stats.csv" % kind)
    if st:
        print("-- rocprofv3 --kernel-trace --stats (kernel_stats.csv) --")
        for i, row in enumerate(csv.reader(open(st))):
            if i < 8:
                print(", ".join(row))
    tr = find("trace_%s/**/*kernel_trace.csv" % kind)
    if tr:
        grid = str(64 * line["config"]["batch_per_gpu"]) if line is not None else None  # full-size launches only
        rows = [r for r in csv.DictReader(open(tr)) if is_search(r.get("Kernel_Name", ""))
                and (grid is None or r.get("Grid_Size", r.get("Grid_Size_X")) == grid)]
        # the timed efSearch is the kernel instantiation with the most dispatches (warm-up + timed + the
        # counter pass); the recall ladder and the efSearch-64 side measurement use other list widths
        names = {}
        for r in rows:
            names[r["Kernel_Name"]] = names.get(r["Kernel_Name"], 0) + 1
        if names:
            timed_name = max(names, key=names.get)
            rows = [r for r in rows if r["Kernel_Name"] == timed_name]
        if rows:
            durs = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows][-100:]
            print("timed search-kernel dispatches: %d, duration us: min %.1f median %.1f mean %.1f max %.1f" % (
                len(durs), min(durs), sorted(durs)[len(durs) // 2], sum(durs) / len(durs), max(durs)))
            r0 = rows[-1]
            print("kernel %s | VGPR_Count %s (rocprofv3 reports wave64 registers in units of two: x 2 = the allocation, which rounds the compiler's count of tests/test_kernel_resources.py up to a multiple of 8) SGPR %s LDS %s scratch %s" % (
                r0.get("Kernel_Name", "")[:80], r0.get("VGPR_Count"), r0.get("SGPR_Count"),
                r0.get("LDS_Block_Size"), r0.get("Scratch_Size")))
    res = {}
    for cname, dname in (("FETCH_SIZE", "pmc_fetch"), ("WRITE_SIZE", "pmc_write")):
        f = find("%s_%s/**/*counter_collection.csv" % (dname, kind))
        if not f:
            continue
        grid = str(64 * line["config"]["batch_per_gpu"]) if line is not None else None
        prow = [r for r in csv.DictReader(open(f))
                if is_search(r.get("Kernel_Name", "")) and r.get("Counter_Name") == cname
                and (grid is None or r.get("Grid_Size", r.get("Grid_Size_X")) == grid)]
        pn = {}
        for r in prow:
            pn[r["Kernel_Name"]] = pn.get(r["Kernel_Name"], 0) + 1
        vals = [float(r["Counter_Value"]) for r in prow if r["Kernel_Name"] == max(pn, key=pn.get)] if pn else []
        if vals:
            vals = vals[-100:]  # the timed launches
            res[cname] = sum(vals) / len(vals)
            print("%s per timed search launch: mean %.1f KB (raw counter) over %d launches" % (
                cname, res[cname], len(vals)))
    if "FETCH_SIZE" in res and line is not None:
        fetch_kb, write_kb = res["FETCH_SIZE"], res.get("WRITE_SIZE", 0.0)
        hbm = int((2 * fetch_kb + write_kb) * 1024)
        alg = line["roofline"]["algorithmic_bytes_per_launch"]
        print("HBM traffic per launch (MB = 1e6 bytes throughout): read %.1f MB raw counter -> %.1f MB after the gfx950 x2 "
              "correction for 16-B/lane loads (MI355X_MICROARCH.md, HBM), write %.2f MB; total %.1f MB; algorithmic %.1f MB; ratio %.2f" % (
                  fetch_kb * 1024 / 1e6, 2 * fetch_kb * 1024 / 1e6, write_kb * 1024 / 1e6, hbm / 1e6, alg / 1e6, hbm / alg))
        a = line["config"]
        entries.append({"workload": a.get("index_tag") or "n%d_d%d_m%d_efc%d_%s_r0" % (a["n_points"], a["dim"], a["M"],
                                                                                        a["ef_construction"], kind),
                        "ef": a["efSearch"], "batch": a["batch_per_gpu"], "fetch_size_kb_raw": fetch_kb,
                        "write_size_kb": write_kb, "hbm_bytes_per_launch": hbm})
import hashlib
import subprocess
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
hh = hashlib.sha256()
for f in ("search_kernels.hip", "search_lean.hip", "coop_rows.inc", "search_common.h", "device_index.h"):
    hh.update(open(os.path.join(root, "hnsw_rs_amd", "csrc", f), "rb").read())
try:
    commit = subprocess.check_output(["git", "-C", root, "rev-parse", "--short", "HEAD"], text=True,
                                     stderr=subprocess.DEVNULL).strip()
except Exception:  # the GPU box gets a snapshot without .git: the caller leaves the commit in a file
    cf = os.path.join(root, ".commit_for_profile")
    commit = open(cf).read().strip() if os.path.exists(cf) else "unknown"
json.dump({"entries": entries, "kernel_sources_sha16": hh.hexdigest()[:16], "commit": commit,
           "profile": os.path.basename(os.path.normpath(out)),
           "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes (scripts/profile.sh); "
                   "hbm_bytes_per_launch = (2 x FETCH_SIZE + WRITE_SIZE) KB: gfx950 tallies the 128-B requests "
                   "of 16-B/lane loads (LDS-DMA included) at 64 B (MI355X_MICROARCH.md, HBM)"},
          open(os.path.join(out, "traffic.json"), "w"), indent=1)
