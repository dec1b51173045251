"""Copies what a `bash scripts/profile.sh TAG` + `bash scripts/profile_c2.sh TAG` pair left under gpurun_out/ into
profiles/ (the committed, judged place) and rebuilds profiles/traffic_latest.json from them.  usage: TAG (e.g. r03)"""
import glob, json, os, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
tag = sys.argv[1]
out, out2, prof = os.path.join(ROOT, "gpurun_out", tag), os.path.join(ROOT, "gpurun_out", tag + "_c2"), os.path.join(ROOT, "profiles")
shutil.copy(os.path.join(out, "summary.txt"), os.path.join(prof, tag + "_f32_and_quant8_summary.txt"))
shutil.copy(os.path.join(out, "bench.json"), os.path.join(prof, tag + "_bench_line.json"))
for k in ("f32", "quant8"):
    f = max(glob.glob(os.path.join(out, "trace_" + k, "**", "*kernel_stats.csv"), recursive=True), key=os.path.getmtime)
    shutil.copy(f, os.path.join(prof, "%s_%s_kernel_stats.csv" % (tag, k)))
t = json.load(open(os.path.join(out, "traffic.json")))
assert t["kernel_sources_sha16"] == bench.kernel_sources_sha16(), "kernel sources changed since the profile"
if os.path.exists(os.path.join(out2, "traffic_c2.json")):
    shutil.copy(os.path.join(out2, "summary.txt"), os.path.join(prof, tag + "_config2_summary.txt"))
    c2 = json.load(open(os.path.join(out2, "traffic_c2.json")))
    t["entries"] = [e for e in t["entries"] if e["workload"] != c2["workload"]] + [c2]
    t["note"] += "; the configs[2] entry comes from scripts/profile_c2.sh (counters restricted to hx_search_kernel)"
    line = json.loads(open(os.path.join(out2, "bench.json")).read().strip().splitlines()[-1])
    r = line["roofline"]
    if r["traffic"] is None:
        r["traffic"] = c2["hbm_bytes_per_launch"]
        r["traffic_measured_at"] = {"commit": t["commit"], "efSearch": c2["ef"], "profile": tag,
                                    "how": "scripts/profile_c2.sh: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes of the same command, (2 x FETCH_SIZE + WRITE_SIZE) KB, merged into this line afterwards"}
    open(os.path.join(prof, tag + "_bench_line_config2_10Mx768.json"), "w").write(json.dumps(line) + "\n")
    print("configs[2]: %.0f q/s, %.3f ms, frac %.4f, traffic %.2f GB" % (line["value"], line["ms_per_step"], r["frac"], r["traffic"] / 1e9))
# configs[3] / configs[4] (scripts/profile_cfg.sh): search-kernel traffic per launch, and the build's insert kernel
for cfg, name in ((3, "config3_100Mx128d_one_gpu"), (4, "config4_16Mx256d_one_gpu")):
    oc = os.path.join(ROOT, "gpurun_out", "%s_c%d" % (tag, cfg))
    tf = os.path.join(oc, "traffic_c%d.json" % cfg)
    if not os.path.exists(tf):
        continue
    ents = json.load(open(tf))
    keys = {(e["workload"], e.get("kernel")) for e in ents}
    t["entries"] = [e for e in t["entries"] if (e["workload"], e.get("kernel")) not in keys] + ents
    shutil.copy(os.path.join(oc, "summary.txt"), os.path.join(prof, "%s_%s_summary.txt" % (tag, name)))
    line = json.loads(open(os.path.join(oc, "bench.json")).read().strip().splitlines()[-1])
    for e in ents:
        how = {"commit": t["commit"], "profile": e.get("profile"), "how": "scripts/profile_cfg.sh: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in "
               "separate passes of the same command, (2 x FETCH_SIZE + WRITE_SIZE) KB, merged into this line afterwards"}
        if e.get("kernel") == "hx_insert_kernel" and line.get("build_roofline") and line["build_roofline"]["traffic"] is None:
            line["build_roofline"]["traffic"] = e["hbm_bytes_all_launches"]
            line["build_roofline"]["traffic_measured_at"] = how
        elif e.get("kernel") is None and line["roofline"]["traffic"] is None:
            line["roofline"]["traffic"] = e["hbm_bytes_per_launch"]
            line["roofline"]["traffic_measured_at"] = how
    open(os.path.join(prof, "%s_bench_line_%s.json" % (tag, name)), "w").write(json.dumps(line) + "\n")
    print("configs[%d]: value %.0f %s, search frac %.4f, traffic %s" % (cfg, line["value"], line["unit"], line["roofline"]["frac"], line["roofline"]["traffic"]))
# SQ counters of the timed kernels (scripts/pmc_sq.sh KIND EF -> gpurun_out/sq_KIND_EF/sq.json): ride on the entries
for e in t["entries"]:
    if e.get("kernel") is not None or "ef" not in e:
        continue
    kind = "quant8" if "_quant8_" in e["workload"] else "f32"
    sqf = os.path.join(ROOT, "gpurun_out", "sq_%s_%d" % (kind, e["ef"]), "sq.json")
    if os.path.exists(sqf) and e["workload"].startswith("n1000000_d100_"):
        e["sq"] = json.load(open(sqf))
    taf = os.path.join(ROOT, "gpurun_out", "ta_%s_%d" % (kind, e["ef"]), "ta.json")  # scripts/pmc_ta.sh KIND EF
    if os.path.exists(taf) and e["workload"].startswith("n1000000_d100_"):
        e["ta"] = json.load(open(taf))
lf = os.path.join(prof, "latency_floor_latest.json")
if os.path.exists(lf):
    fdoc = json.load(open(lf))
    print("latency floor file: sources %s (%s)" % (fdoc.get("kernel_sources_sha16"), "current" if fdoc.get("kernel_sources_sha16") == bench.kernel_sources_sha16() else "STALE"))
json.dump(t, open(os.path.join(prof, "traffic_latest.json"), "w"), indent=1)
b = json.load(open(os.path.join(prof, tag + "_bench_line.json")))
print("configs[1]: %.0f q/s, frac %.4f; traffic file for commit %s, sources %s" % (b["value"], b["roofline"]["frac"], t["commit"], t["kernel_sources_sha16"]))
