"""Round 4, VERDICT item 3(c): the headline rows (1M x 100d f32) zero-padded to a 512-byte stride and searched by
hx_lean_f32_kernel<128, ...> (the cooperative whole-line gather of coop_rows.inc) against the 400-byte rows of
hx_lean_f32_kernel<100, ...>.  Zero padding is exact: (0 - 0)^2 = 0 and s + 0.0f == s, so ids and distance bits
must be the same; the padded rows cost 512 B each, and every one of them is exactly four whole 128-byte lines.

    python scripts/experiments/r04_padded_rows_probe.py [n_points]
"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import hnsw_rs_amd as H
from hnsw_rs_amd.distributed import make_device_search

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
m, n, B = 16, 10, 1024
dev = torch.device("cuda:0")
vs = H.synth_rows(0, 0x5EED0001, 0, N, 100, 16)
qs = H.synth_rows(0, 0x5EED0002, 0, 10 * B, 100, 16)
lv = H.draw_levels(m, N)
res = {}
for name, d in (("400-byte rows, d = 100", 100), ("zero-padded to 512 bytes, d = 128", 128)):
    rows = vs if d == 100 else np.concatenate([vs, np.zeros((N, 28), dtype=np.float32)], axis=1)
    q = qs if d == 100 else np.concatenate([qs, np.zeros((len(qs), 28), dtype=np.float32)], axis=1)
    idx = H.HNSW.new(m, 32, d, H.VEC_F32)
    t0 = time.time()
    idx.insert_bulk_device(np.ascontiguousarray(rows), 16, False, levels=lv)
    idx.upload()
    dQ = torch.from_numpy(np.ascontiguousarray(q)).to(dev).view(10, B, d)
    for ef in (64, 68):
        ls = make_device_search(idx, n, ef, B, dev)
        for i in range(10):
            ls(dQ[i % 10].contiguous())
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(100):
            ls(dQ[i % 10].contiguous())
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 100
        ids, dd = ls(dQ[0].contiguous())
        torch.cuda.synchronize()
        st = ls.stats[:B].cpu().numpy()
        res[(d, ef)] = (ids.cpu().numpy().copy(), dd.cpu().numpy().copy(), st[:, :3].copy())
        print("%-36s efSearch %d: %.4f ms per 1024-query batch (%.2f M q/s); build %.1f s; n_dist %.1f" % (
            name, ef, ms, B / ms / 1e3, time.time() - t0, st[:, 0].mean()), flush=True)
    del idx
# the two graphs are built by batch-parallel inserts of the same points in the same order with the same distances
# (padding is exact), so they should be the same graph and answer identically
for ef in (64, 68):
    a, b = res[(100, ef)], res[(128, ef)]
    print("efSearch %d: ids identical %s, distance bits identical %s, counters identical %s" % (
        ef, np.array_equal(a[0], b[0]), np.array_equal(a[1].view(np.uint32), b[1].view(np.uint32)), np.array_equal(a[2], b[2])))
