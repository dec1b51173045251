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
json.dump(t, open(os.path.join(prof, "traffic_latest.json"), "w"), indent=1)
b = json.load(open(os.path.join(prof, tag + "_bench_line.json")))
print("configs[1]: %.0f q/s, frac %.4f; traffic file for commit %s, sources %s" % (b["value"], b["roofline"]["frac"], t["commit"], t["kernel_sources_sha16"]))
