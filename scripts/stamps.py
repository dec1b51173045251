"""Diagnostic: per-phase cycle shares of the search kernel's expansion loop (stamps build).
   HNSW_MI355X_LIB=hnsw_rs_amd/libhnsw_mi355x_stamps.so python scripts/stamps.py [N] [ef]"""
import os, sys, time
sys.path.insert(0, '.')
import numpy as np
import torch
import hnsw_rs_amd as H
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
ef = int(sys.argv[2]) if len(sys.argv) > 2 else 64
d, m, nq, n = 100, 16, 1024, 10
cache = '/tmp/hnsw_bench_cache/n%d_d100_m16_efc32_quant8_r0' % N
if os.path.isdir(cache):
    idx = H.HNSW.load(cache)
else:
    vs = H.synth_rows(0, 0x5EED0001, 0, N, d, 32)
    idx = H.HNSW.new(m, 32, d).insert_bulk(vs, 32, False)
    os.makedirs('/tmp/hnsw_bench_cache', exist_ok=True); idx.save(cache)
qs = H.synth_rows(0, 0x5EED0002, 0, nq, d, 8)
idx.upload()
dev = torch.device('cuda:0')
dQ = torch.from_numpy(qs).to(dev)
ids = torch.empty((nq, n), dtype=torch.int32, device=dev); dd = torch.empty((nq, n), dtype=torch.float32, device=dev)
cnt = torch.empty(nq, dtype=torch.int32, device=dev); st = torch.empty((nq, 4), dtype=torch.int32, device=dev)
dbg = torch.zeros((nq, 8), dtype=torch.int64, device=dev)
os.environ['HX_DBG_PTR'] = str(dbg.data_ptr())
for _ in range(3):
    idx.search_batch_device(dQ.data_ptr(), nq, n, ef, ids.data_ptr(), dd.data_ptr(), cnt.data_ptr(), st.data_ptr(), 0)
torch.cuda.synchronize()
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
e0.record(); idx.search_batch_device(dQ.data_ptr(), nq, n, ef, ids.data_ptr(), dd.data_ptr(), cnt.data_ptr(), st.data_ptr(), 0); e1.record()
torch.cuda.synchronize()
D = dbg.cpu().numpy().astype(np.float64); S = st.cpu().numpy()
print('kernel ms', e0.elapsed_time(e1), 'n_exp', S[:, 1].mean(), 'n_dist', S[:, 0].mean())
names = ['pick+block (hit: LDS image, miss: HBM)+predict+DMA issue', 'id extract + visited filter', 'distance', 'merge', 'TOTAL', '-', 'prediction hits (count)']
tot = D[:, 4].mean()
names[1]='id+visited on prediction HITS (total)'; names[5]='id+visited on prediction MISSES (total)'
for i in (0, 1, 5, 2, 3, 4, 6):
    print('%-60s %10.0f cycles/query  %5.1f%%   per expansion %7.0f' % (names[i], D[:, i].mean(), 100 * D[:, i].mean() / tot, D[:, i].mean() / S[:, 1].mean()))
hits=D[:,6].mean(); ne=S[:,1].mean(); print('per hit %.0f cycles, per miss %.0f cycles' % (D[:,1].mean()/hits, D[:,5].mean()/(ne-hits)))
print('unaccounted %.1f%%' % (100 * (tot - (D[:, :4].sum(1).mean()+D[:,5].mean())) / tot))
print('cycles total max %.0f min %.0f; 100MHz ticks? kernel_ms*1e5=%.0f' % (D[:, 4].max(), D[:, 4].min(), e0.elapsed_time(e1) * 1e5))
