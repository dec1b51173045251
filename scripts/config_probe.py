"""Probe of the other BASELINE configs' shapes on one GPU (gpurun): build N x d on the device, then
recall, batch-1024 search rate and algorithmic HBM rate.  usage: N kind d [normalise] [recipe]"""
import os, sys, time
sys.path.insert(0, '.')
import numpy as np, torch
import hnsw_rs_amd as H
N = int(sys.argv[1]); kind = int(sys.argv[2]); d = int(sys.argv[3])
norm = len(sys.argv) > 4 and sys.argv[4] == '1'
recipe = int(sys.argv[5]) if len(sys.argv) > 5 else 0
m, B, n = 16, 1024, 10
thr = min(32, os.cpu_count())
t = time.time(); vs = H.synth_rows(recipe, 0x5EED0001, 0, N, d, thr); qs = H.synth_rows(recipe, 0x5EED0002, 0, 8 * B, d, 8)
if norm:  # unit rows: cosine order == L2 order (SURVEY 8d, configs[2])
    vs /= np.linalg.norm(vs, axis=1, keepdims=True); qs /= np.linalg.norm(qs, axis=1, keepdims=True)
print('gen %.1fs' % (time.time() - t), flush=True)
idx = H.HNSW.new(m, 32, d, kind)
idx.set_option("inline_budget_mb", 8192)
if os.environ.get("BMAX"): idx.set_option("gpu_build_batch_max", int(os.environ["BMAX"]))
if os.environ.get("INLINE") is not None: idx.set_option("inline_rows", int(os.environ["INLINE"]))
t = time.time(); idx.insert_bulk_device(vs, thr, True); print('device build %.1fs, %d layers' % (time.time() - t, idx.nb_layers()), flush=True)
del vs
t = time.time(); idx.upload(); print('upload %.1fs, %.0f MB in HBM' % (time.time() - t, idx.device_bytes() / 1e6), flush=True)
truth, _ = idx.brute_force(qs[:256], n)
dev = torch.device('cuda:0'); dQ = torch.from_numpy(qs).to(dev)
ids = torch.empty((8 * B, n), dtype=torch.int32, device=dev); dd = torch.empty((8 * B, n), dtype=torch.float32, device=dev)
cnt = torch.empty(8 * B, dtype=torch.int32, device=dev); st = torch.empty((8 * B, 4), dtype=torch.int32, device=dev)
row_bytes = d + 8 if kind == 0 else 4 * d
for ef in (64, 128, 256):
    got, _, _, s = idx.search_batch(qs[:256], n, ef)
    rec = sum(len(set(a) & set(b)) for a, b in zip(got.tolist(), truth.tolist())) / 2560
    def run(b):
        o = b * B
        idx.search_batch_device(dQ[o:].data_ptr(), B, n, ef, ids[o:].data_ptr(), dd[o:].data_ptr(), cnt[o:].data_ptr(), st[o:].data_ptr(), 0)
    for b in range(8): run(b)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for r in range(2):
        for b in range(8): run(b)
    e1.record(); torch.cuda.synchronize(); ms = e0.elapsed_time(e1) / 16
    bq = s[:, 0].mean() * row_bytes + s[:, 1].mean() * 4 + s[:, 2].mean() * 4 + 4 * d + 8 * n
    print('N=%d d=%d kind=%d recipe=%d ef=%d recall@10 %.4f  %.3f ms/batch  %.2f M q/s  n_dist %.0f n_exp %.1f  alg %.0f GB/s' % (
        N, d, kind, recipe, ef, rec, ms, B / ms / 1e3, s[:, 0].mean(), s[:, 1].mean(), bq * B / ms / 1e6), flush=True)
    # one launch of all 8192 queries: the machine-filling regime
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    idx.search_batch_device(dQ.data_ptr(), 8 * B, n, ef, ids.data_ptr(), dd.data_ptr(), cnt.data_ptr(), st.data_ptr(), 0)
    torch.cuda.synchronize()
    e0.record()
    idx.search_batch_device(dQ.data_ptr(), 8 * B, n, ef, ids.data_ptr(), dd.data_ptr(), cnt.data_ptr(), st.data_ptr(), 0)
    e1.record(); torch.cuda.synchronize(); ms8 = e0.elapsed_time(e1)
    sa = st.cpu().numpy()
    bq8 = sa[:, 0].mean() * row_bytes + sa[:, 1].mean() * 4 + sa[:, 2].mean() * 4 + 4 * d + 8 * n
    print('    one launch of %d queries: %.3f ms  %.2f M q/s  alg %.0f GB/s; per query n_dist mean %.0f p99 %.0f max %d' % (
        8 * B, ms8, 8 * B / ms8 / 1e3, bq8 * 8 * B / ms8 / 1e6, sa[:, 0].mean(), np.percentile(sa[:, 0], 99), sa[:, 0].max()), flush=True)
