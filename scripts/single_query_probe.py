"""The reference's call pattern on the bench index: T host threads, one query per hnsw_search call each
(hnsw_bench_search_threads), for several T and coalescer settings; and the host-pointer batch entry.

    python scripts/single_query_probe.py [f32|quant8] [n_points] [ef]
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hnsw_rs_amd as H

kind_name = sys.argv[1] if len(sys.argv) > 1 else "f32"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
ef = int(sys.argv[3]) if len(sys.argv) > 3 else 68
d, m = 100, 16
kind = H.VEC_F32 if kind_name == "f32" else H.VEC_QUANT8
t0 = time.time()
vs = H.synth_rows(0, 0x5EED0001, 0, N, d, 16)
index = H.HNSW.new(m, 32, d, kind)
index.insert_bulk_device(vs, 16, False)
index.upload()
qs = H.synth_rows(0, 0x5EED0002, 0, 10240, d, 16)
print("index %d x %dd %s built in %.1fs" % (N, d, kind_name, time.time() - t0), flush=True)
ref_ids, _, ref_c, _ = index.search_batch(qs, 10, ef)

for window, depth in ((30, 3),):
    index.set_option("coalesce_us", window)
    index.set_option("coalesce_depth", depth)
    for T in (1, 16, 64, 256, 1024):
        if window < 0 and T > 64:
            continue
        keys = ("coalesced_batches", "coalesced_queries", "coalesce_ns_window", "coalesce_ns_turn", "coalesce_ns_gpu", "coalesce_ns_handout")
        s0 = [index.stat(k) for k in keys]
        ids, counts, calls, wall, lat = index.search_threads(qs, 10, ef, T, 1.0)
        dl = [index.stat(k) - x for k, x in zip(keys, s0)]
        nb, nq = dl[0], dl[1]
        same = bool(np.array_equal(ids, ref_ids))
        ph = " leader us/batch: window %.0f turn %.0f gpu %.0f handout %.0f" % tuple(x / 1e3 / nb for x in dl[2:]) if nb else ""
        ph += "  cpu %.1f user + %.1f sys cores" % (lat["cpu_user_s"] / wall, lat["cpu_sys_s"] / wall)
        print("coalesce_us %3d depth %d  T %4d: %9.0f q/s  p50 %6.0f us  p99 %6.0f us  mean batch %6.1f  identical %s%s" % (
            window, depth, T, calls / wall, lat["p50"], lat["p99"], (nq / nb) if nb else 1.0, same, ph), flush=True)

# host-pointer batch entry, one caller and two concurrent callers
import threading
B = 1024
qb = [np.ascontiguousarray(qs[i * B:(i + 1) * B]) for i in range(10)]
for i in range(3):
    index.search_batch(qb[i], 10, ef)
t1 = time.perf_counter()
for i in range(100):
    index.search_batch(qb[i % 10], 10, ef)
dt = time.perf_counter() - t1
print("hnsw_search_batch, one caller: %.1f us per 1024-query call, %.2f M q/s" % (dt / 100 * 1e6, 100 * B / dt / 1e6))
for C in (2, 3):
    def work(t):
        for i in range(100):
            index.search_batch(qb[(i + t) % 10], 10, ef)
    th = [threading.Thread(target=work, args=(t,)) for t in range(C)]
    t1 = time.perf_counter()
    [t.start() for t in th]
    [t.join() for t in th]
    dt = time.perf_counter() - t1
    print("hnsw_search_batch, %d concurrent callers: %.2f M q/s" % (C, C * 100 * B / dt / 1e6))
