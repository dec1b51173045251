"""throw-away perf probe: build N x 100d, time batches of 1024 queries on one GPU"""
import sys, os, time
sys.path.insert(0, '.')
import numpy as np
import torch
import hnsw_rs_amd as H
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
KIND = int(sys.argv[3]) if len(sys.argv) > 3 else 0
ef = int(sys.argv[2]) if len(sys.argv) > 2 else 64
d, m, nq = 100, 16, 1024
thr = min(32, os.cpu_count())
t = time.time(); vs = H.synth_rows(0, 0x5EED0001, 0, N, d, thr); qs = H.synth_rows(0, 0x5EED0002, 0, nq * 8, d, thr)
print('gen %.1fs cpus=%d' % (time.time() - t, os.cpu_count()), flush=True)
t = time.time(); idx = H.HNSW.new(m, 32, d, KIND).insert_bulk(vs, thr, False); print('build %.1fs layers=%d' % (time.time() - t, idx.nb_layers()), flush=True)
t = time.time(); idx.upload(); print('upload %.1fs bytes=%.1fMB' % (time.time() - t, idx.device_bytes() / 1e6), flush=True)
dev = torch.device('cuda:0')
dQ = torch.from_numpy(qs).to(dev)
n = 10
ids = torch.empty((nq * 8, n), dtype=torch.int32, device=dev); dd = torch.empty((nq * 8, n), dtype=torch.float32, device=dev)
cnt = torch.empty(nq * 8, dtype=torch.int32, device=dev); st = torch.empty((nq * 8, 4), dtype=torch.int32, device=dev)
def run(b, stream):
    o = b * nq
    idx.search_batch_device(dQ[o:].data_ptr(), nq, n, ef, ids[o:].data_ptr(), dd[o:].data_ptr(), cnt[o:].data_ptr(), st[o:].data_ptr(), stream)
s0 = torch.cuda.current_stream().cuda_stream
for b in range(8): run(b, s0)
torch.cuda.synchronize()
stn = st.cpu().numpy()
print('status ok', (stn[:, 3] == 0).all(), 'n_dist %.1f n_exp %.1f sum_deg %.1f' % tuple(stn[:, :3].mean(0)), flush=True)
for rep in range(3):
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for b in range(8): run(b, s0)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 8
    print('1 stream: %.3f ms/batch -> %.0f q/s' % (ms, nq / ms * 1e3), flush=True)
for S in (2, 4, 8):
    streams = [torch.cuda.Stream() for _ in range(S)]
    torch.cuda.synchronize(); t = time.time()
    R = 5
    for r in range(R):
        for b in range(8): run(b, streams[b % S].cuda_stream)
    torch.cuda.synchronize(); ms = (time.time() - t) * 1e3 / (8 * R)
    print('%d streams: %.3f ms/batch -> %.0f q/s' % (S, ms, nq / ms * 1e3), flush=True)
# one big launch of 8192 queries
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
e0.record()
idx.search_batch_device(dQ.data_ptr(), nq * 8, n, ef, ids.data_ptr(), dd.data_ptr(), cnt.data_ptr(), st.data_ptr(), s0)
e1.record(); torch.cuda.synchronize(); ms = e0.elapsed_time(e1)
print('one launch of 8192: %.3f ms -> %.0f q/s' % (ms, 8192 / ms * 1e3), flush=True)
bq = stn[:, 0].mean() * 108 + stn[:, 1].mean() * 4 + stn[:, 2].mean() * 4 + 4 * d + 8 * n
print('alg bytes/query %.0f' % bq)
