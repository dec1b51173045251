"""build N x 100d on the device, report build time, recall and batch-1024 search rate (gpurun)"""
import os, sys, time
sys.path.insert(0, '.')
import numpy as np, torch
import hnsw_rs_amd as H
N = int(sys.argv[1]); kind = int(sys.argv[2]) if len(sys.argv) > 2 else 0
d, m, B, n = 100, 16, 1024, 10
thr = min(32, os.cpu_count())
t = time.time(); vs = H.synth_rows(0, 0x5EED0001, 0, N, d, thr); qs = H.synth_rows(0, 0x5EED0002, 0, 8 * B, d, 8)
print('gen %.1fs' % (time.time() - t), flush=True)
idx = H.HNSW.new(m, 32, d, kind)
idx.set_option("inline_budget_mb", 8192)   # no 40-GB inline-rows copy for this probe
t = time.time(); idx.insert_bulk_device(vs, thr, True); print('device build %.1fs, %d layers' % (time.time() - t, idx.nb_layers()), flush=True)
del vs
t = time.time(); idx.upload(); print('upload %.1fs, %.0f MB in HBM' % (time.time() - t, idx.device_bytes() / 1e6), flush=True)
truth, _ = idx.brute_force(qs[:256], n)
dev = torch.device('cuda:0'); dQ = torch.from_numpy(qs).to(dev)
ids = torch.empty((8 * B, n), dtype=torch.int32, device=dev); dd = torch.empty((8 * B, n), dtype=torch.float32, device=dev)
cnt = torch.empty(8 * B, dtype=torch.int32, device=dev); st = torch.empty((8 * B, 4), dtype=torch.int32, device=dev)
for ef in (64, 96, 128):
    got, _, _, s = idx.search_batch(qs[:256], n, ef)
    rec = sum(len(set(a) & set(b)) for a, b in zip(got.tolist(), truth.tolist())) / 2560
    def run(b):
        o = b * B
        idx.search_batch_device(dQ[o:].data_ptr(), B, n, ef, ids[o:].data_ptr(), dd[o:].data_ptr(), cnt[o:].data_ptr(), st[o:].data_ptr(), 0)
    for b in range(8): run(b)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for r in range(4):
        for b in range(8): run(b)
    e1.record(); torch.cuda.synchronize(); ms = e0.elapsed_time(e1) / 32
    print('N=%d kind=%d ef=%d recall@10 %.4f  %.3f ms/batch  %.2f M q/s  n_dist %.0f n_exp %.1f' % (N, kind, ef, rec, ms, B / ms / 1e3, s[:, 0].mean(), s[:, 1].mean()), flush=True)
