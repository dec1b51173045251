import os, sys
sys.path.insert(0, '.')
import numpy as np, torch
import hnsw_rs_amd as H
N, ef = 1000000, int(sys.argv[1]) if len(sys.argv) > 1 else 64
d, m, nq, n = 100, 16, 1024, 10
cache = '/tmp/hnsw_bench_cache/n%d_d100_m16_efc32_quant8_r0' % N
if os.path.isdir(cache): idx = H.HNSW.load(cache)
else:
    idx = H.HNSW.new(m, 32, d).insert_bulk(H.synth_rows(0, 0x5EED0001, 0, N, d, 32), 32, False)
    os.makedirs('/tmp/hnsw_bench_cache', exist_ok=True); idx.save(cache)
qs = H.synth_rows(0, 0x5EED0002, 0, nq, d, 8); idx.upload()
dev = torch.device('cuda:0'); dQ = torch.from_numpy(qs).to(dev)
ids = torch.empty((nq, n), dtype=torch.int32, device=dev); dd = torch.empty((nq, n), dtype=torch.float32, device=dev)
cnt = torch.empty(nq, dtype=torch.int32, device=dev); st = torch.empty((nq, 4), dtype=torch.int32, device=dev)
dbg = torch.zeros((nq, 8), dtype=torch.int64, device=dev); os.environ['HX_DBG_PTR'] = str(dbg.data_ptr())
for _ in range(3): idx.search_batch_device(dQ.data_ptr(), nq, n, ef, ids.data_ptr(), dd.data_ptr(), cnt.data_ptr(), st.data_ptr(), 0)
torch.cuda.synchronize()
D = dbg.cpu().numpy().astype(np.float64); S = st.cpu().numpy(); ne = S[:, 1].mean()
names = {0: 'pick + DMA issue + wait + image read', 1: 'id + visited', 2: 'id + visited + distance', 3: 'exchange (write, barrier, read)', 5: 'merge', 4: 'TOTAL'}
for i in (0, 1, 2, 3, 5, 4): print('%-40s %9.0f cyc/query %5.1f%%  per expansion %6.0f' % (names[i], D[:, i].mean(), 100 * D[:, i].mean() / D[:, 4].mean(), D[:, i].mean() / ne))
print('hits', D[:, 6].mean(), 'n_exp', ne)
