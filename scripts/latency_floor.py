"""The roof that binds a 1024-query batch of the headline configuration: one query's own dependent chain.
Stamps build (in-kernel cycle counters, `make -C hnsw_rs_amd/csrc stamps`):

    HNSW_MI355X_LIB=hnsw_rs_amd/libhnsw_mi355x_stamps.so python scripts/latency_floor.py [ef] [out.json]

Per query the lean f32 kernel walks the upper layers and then runs P layer-0 passes; a pass cannot be shorter than
its dependent steps: pick the candidate pair and read their adjacency rows (one round trip), gather the rows
(second round trip), run FullVec's left-to-right chain over them (full.rs:23-29: d dependent adds per row, rows side
by side in the lanes).  floor = staging + upper layers + P x (adjacency round trip + row round trip + chain), i.e.
the pass with ALL bookkeeping (visited set, merges, commits) free -- measured twice: in a 64-query launch (a wave alone
on its CU: the latencies of an idle machine) and in the 1024-query launch the metric times."""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
import hnsw_rs_amd as H

ef = int(sys.argv[1]) if len(sys.argv) > 1 else 68
outp = sys.argv[2] if len(sys.argv) > 2 else None
N, d, m, n = 1_000_000, 100, 16, 10
vs = H.synth_rows(0, 0x5EED0001, 0, N, d, 32)
idx = H.HNSW.new(m, 32, d, H.VEC_F32)
idx.set_device(0)
idx.insert_bulk_device(vs, 32, False)
idx.upload()
dev = torch.device("cuda:0")
qs_all = H.synth_rows(0, 0x5EED0002, 0, 1024, d, 8)
res = {}
for nq in (64, 1024):
    dQ = torch.from_numpy(qs_all[:nq]).to(dev)
    ids = torch.empty((nq, n), dtype=torch.int32, device=dev)
    dd = torch.empty((nq, n), dtype=torch.float32, device=dev)
    cnt = torch.empty(nq, dtype=torch.int32, device=dev)
    st = torch.empty((nq, 4), dtype=torch.int32, device=dev)
    dbg = torch.zeros((nq, 16), dtype=torch.int64, device=dev)
    os.environ["HX_DBG_PTR"] = str(dbg.data_ptr())
    for _ in range(3):
        idx.search_batch_device(dQ.data_ptr(), nq, n, ef, ids.data_ptr(), dd.data_ptr(), cnt.data_ptr(), st.data_ptr(), 0)
    torch.cuda.synchronize()
    dbg.zero_()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    idx.search_batch_device(dQ.data_ptr(), nq, n, ef, ids.data_ptr(), dd.data_ptr(), cnt.data_ptr(), st.data_ptr(), 0)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    D = dbg.cpu().numpy().astype(np.float64)
    tot, passes = D[:, 4], D[:, 6]
    ghz = tot.max() / (ms * 1e6)  # the slowest wave spans the kernel
    # stamp 1 spans look + row request + claims + what is left of the row round trip + chain; 2 and 3 are the last two alone
    per_pass = {"pick_and_adjacency_round_trip": D[:, 0].mean() / passes.mean(),
                "visited_look_row_request_claims (the rows are in flight underneath)": (D[:, 1] - D[:, 2] - D[:, 3]).mean() / passes.mean(),
                "row_round_trip_left_after_the_claims": D[:, 2].mean() / passes.mean(),
                "chain_and_sqrt": D[:, 3].mean() / passes.mean(),
                "merges_commits_overflow_rows": (D[:, 8] + D[:, 9] + D[:, 5] + D[:, 10]).mean() / passes.mean()}
    # per query: staging / upper layers + per pass the adjacency round trip, the row round trip measured alone
    # (scripts/micro/gather_latency.hip: a wave alone on its CU, 25 x 16 B per lane from 64 random rows) and the chain
    ROW_GATHER_ALONE = 1850.0
    floor_cyc = D[:, 7] + D[:, 0] + passes * ROW_GATHER_ALONE + D[:, 3]
    r = {"queries": nq, "kernel_ms_under_stamps": round(ms, 4), "implied_clock_ghz": round(ghz, 3),
         "passes_per_query": round(passes.mean(), 2), "cycles_per_query_mean": round(tot.mean()), "cycles_per_query_max": round(tot.max()),
         "staging_and_upper_layers_cycles": round(D[:, 7].mean()),
         "cycles_per_pass": {k: round(v) for k, v in per_pass.items()},
         "row_gather_alone_cycles (profiles/r02_gather_microbench_cache_resident.txt)": 1850,
         "floor_cycles_mean": round(floor_cyc.mean()), "floor_cycles_of_the_slowest_query": round(floor_cyc[np.argmax(tot)]),
         "floor_ms_slowest_query": round(floor_cyc[np.argmax(tot)] / ghz / 1e6, 4), "floor_ms_mean_query": round(floor_cyc.mean() / ghz / 1e6, 4),
         "floor_over_kernel": round(floor_cyc[np.argmax(tot)] / tot.max(), 3)}
    # the bookkeeping term in its parts (stamps 8, 9, 5, 10), per pass, and how many merges inserted 0 / 1-2 / more keys
    r["bookkeeping_cycles_per_pass"] = {"merge_of_c": round(D[:, 8].mean() / passes.mean()), "is_p_next_and_prefetch": round(D[:, 9].mean() / passes.mean()),
                                        "claims_of_p": round(D[:, 5].mean() / passes.mean()), "merge_of_p": round(D[:, 10].mean() / passes.mean())}
    r["per_query_counts"] = {"p_commits": round(D[:, 15].mean(), 2), "merges_inserting_0": round(D[:, 12].mean(), 2),
                             "merges_inserting_1_or_2": round(D[:, 13].mean(), 2), "merges_inserting_more": round(D[:, 14].mean(), 2),
                             "passes_with_a_second_claim_round": round(D[:, 11].mean(), 2)}
    res[str(nq)] = r
    print(json.dumps(r))
if outp:
    import hashlib
    hh = hashlib.sha256()
    for f in ("search_kernels.hip", "search_lean.hip", "coop_rows.inc", "search_common.h", "device_index.h"):
        hh.update(open(os.path.join("hnsw_rs_amd", "csrc", f), "rb").read())
    json.dump({"efSearch": ef, "vec_kind": "f32", "workload": "1M x 100d f32, M=16", "kernel_sources_sha16": hh.hexdigest()[:16],
               "note": "stamps build (in-kernel cycle counters; they lengthen the kernel by ~10 %): floor = staging + upper layers + "
                       "passes x (pick + adjacency round trip + the row gather measured alone [1850 cycles, a wave alone on its CU, "
                       "scripts/micro/gather_latency.hip] + FullVec's chain), every piece of bookkeeping free",
               "launches": res}, open(outp, "w"), indent=1)
