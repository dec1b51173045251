"""single-query latency through the host-pointer entry points (what the Rust shim's ann_by_vector uses)"""
import sys, time; sys.path.insert(0, '.')
import numpy as np, hnsw_rs_amd as H
N, d, m = 1000000, 100, 16
vs = H.synth_rows(0, 0x5EED0001, 0, N, d, 32); qs = H.synth_rows(0, 0x5EED0002, 0, 2048, d, 8)
for kind in (H.VEC_QUANT8, H.VEC_F32):
    idx = H.HNSW.new(m, 32, d, kind); idx.insert_bulk_device(vs, 32, False); idx.upload()
    for q in qs[:50]: idx.ann_by_vector(q, 10, 64)
    t = time.perf_counter()
    for q in qs[:1000]: idx.ann_by_vector(q, 10, 64)
    dt = time.perf_counter() - t
    print('kind %d: ann_by_vector (1 query per call): %.1f us per call, %.0f calls/s' % (kind, dt * 1e3, 1000 / dt), flush=True)
    for nq in (16, 128, 1024):
        idx.search_batch(qs[:nq], 10, 64)
        t = time.perf_counter(); R = 20
        for r in range(R): idx.search_batch(qs[:nq], 10, 64)
        dt = (time.perf_counter() - t) / R
        print('kind %d: search_batch nq=%d host pointers: %.1f us per call, %.0f q/s' % (kind, nq, dt * 1e6, nq / dt), flush=True)
