"""Condense a scripts/profile_cfg.sh output directory (BASELINE configs[3] / configs[4] on one GPU): the bench
line, the PMC-derived HBM traffic of the timed search kernel per launch and -- configs[4] -- of all the launches
of hx_insert_kernel, and traffic_c<N>.json with the entries bench.py reads from profiles/traffic_latest.json.
All sizes in MB = 1e6 bytes."""
import csv
import glob
import json
import os
import sys

out, cfg = sys.argv[1], int(sys.argv[2])
full = json.loads(open(os.path.join(out, "bench.json")).read().strip().splitlines()[-1])
line = json.loads(open(os.path.join(out, "pmc_fetch.json")).read().strip().splitlines()[-1])
print("== bench line (python bench.py --config %d ...) ==" % cfg)
print(json.dumps(full))
print()
conf = line["config"]
B = conf["batch_per_gpu"]
srl = line["roofline"]  # (configs[4]: the search on the index the run built; the build has build_roofline)
print("bench under rocprofv3 --pmc: kernel_ms %.5f, efSearch %d" % (srl["kernel_ms"], conf["efSearch"]))


def rows_of(dname, cname):
    f = glob.glob(os.path.join(out, dname, "**", "*counter_collection.csv"), recursive=True)
    if not f:
        return []
    return [r for r in csv.DictReader(open(f[0])) if r.get("Counter_Name") == cname]


entries = []
# ---- the timed search kernel: the instantiation launched most often at the batch's grid size ----
res = {}
for cname, dname in (("FETCH_SIZE", "pmc_fetch"), ("WRITE_SIZE", "pmc_write")):
    rows = [r for r in rows_of(dname, cname) if "hx_insert_kernel" not in r["Kernel_Name"]
            and r.get("Grid_Size", r.get("Grid_Size_X")) == str(64 * B)]
    names = {}
    for r in rows:
        names[r["Kernel_Name"]] = names.get(r["Kernel_Name"], 0) + 1
    if not names:
        continue
    timed = max(names, key=names.get)
    vals = [float(r["Counter_Value"]) for r in rows if r["Kernel_Name"] == timed][-20:]
    res[cname] = sum(vals) / len(vals)
    print("%s per timed search launch (%s...): mean %.1f KB (raw counter) over %d launches" % (cname, timed[:60], res[cname], len(vals)))
if "FETCH_SIZE" in res:
    fk, wk = res["FETCH_SIZE"], res.get("WRITE_SIZE", 0.0)
    hbm = int((2 * fk + wk) * 1024)
    alg = srl["algorithmic_bytes_per_launch"]
    print("search kernel, HBM traffic per launch: read %.1f MB raw -> %.1f MB after the gfx950 x2 (16-B/lane loads; calibrated per "
          "access shape in profiles/r03_gather_shapes_fetch_size.txt), write %.2f MB; algorithmic %.1f MB; ratio %.2f" % (
              fk * 1024 / 1e6, 2 * fk * 1024 / 1e6, wk * 1024 / 1e6, alg / 1e6, hbm / alg))
    entries.append({"workload": conf["index_tag"], "ef": conf["efSearch"], "batch": B, "fetch_size_kb_raw": fk,
                    "write_size_kb": wk, "hbm_bytes_per_launch": hbm, "profile": os.path.basename(os.path.normpath(out))})
# ---- configs[4]: the build's kernel, every launch ----
if cfg == 4:
    tot = {}
    for cname, dname in (("FETCH_SIZE", "pmc_fetch"), ("WRITE_SIZE", "pmc_write")):
        vals = [float(r["Counter_Value"]) for r in rows_of(dname, cname) if "hx_insert_kernel" in r["Kernel_Name"]]
        tot[cname] = (sum(vals), len(vals))
        print("%s over all %d launches of hx_insert_kernel: %.1f MB (raw counter)" % (cname, len(vals), sum(vals) * 1024 / 1e6))
    br = line.get("build_roofline")
    if br and tot.get("FETCH_SIZE", (0, 0))[1]:
        hbm = int((2 * tot["FETCH_SIZE"][0] + tot.get("WRITE_SIZE", (0, 0))[0]) * 1024)
        print("insert kernel, HBM traffic of the whole build: %.1f MB (2 x FETCH_SIZE + WRITE_SIZE); algorithmic %.1f MB; ratio %.2f; "
              "kernel time under the counters %.2f s" % (hbm / 1e6, br["algorithmic_bytes"] / 1e6, hbm / br["algorithmic_bytes"], br["kernel_s"]))
        entries.append({"workload": conf["index_tag"], "kernel": "hx_insert_kernel", "hbm_bytes_all_launches": hbm,
                        "launches": tot["FETCH_SIZE"][1], "algorithmic_bytes_in_that_run": br["algorithmic_bytes"],
                        "profile": os.path.basename(os.path.normpath(out))})
json.dump(entries, open(os.path.join(out, "traffic_c%d.json" % cfg), "w"), indent=1)
