"""Diagnostic: per-phase cycle shares of the f32 two-rows-per-pass loop (stamps build).
   HNSW_MI355X_LIB=hnsw_rs_amd/libhnsw_mi355x_stamps.so python scripts/stamps_f32.py [N] [ef ...]"""
import os, sys
sys.path.insert(0, '.')
import numpy as np
import torch
import hnsw_rs_amd as H
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
efs = [int(x) for x in sys.argv[2:]] or [68]
d, m, nq, n = 100, 16, 1024, 10
vs = H.synth_rows(0, 0x5EED0001, 0, N, d, 32)
idx = H.HNSW.new(m, 32, d, H.VEC_F32)
idx.set_device(0)
idx.insert_bulk_device(vs, 32, False)
qs = H.synth_rows(0, 0x5EED0002, 0, nq, d, 8)
idx.upload()
dev = torch.device('cuda:0')
dQ = torch.from_numpy(qs).to(dev)
ids = torch.empty((nq, n), dtype=torch.int32, device=dev); dd = torch.empty((nq, n), dtype=torch.float32, device=dev)
cnt = torch.empty(nq, dtype=torch.int32, device=dev); st = torch.empty((nq, 4), dtype=torch.int32, device=dev)
dbg = torch.zeros((nq, 8), dtype=torch.int64, device=dev)
os.environ['HX_DBG_PTR'] = str(dbg.data_ptr())
names = {0: 'pick c,p + adjacency row round trip', 1: 'visited: insert c, look up p', 2: 'row gather until all 25 pieces landed',
         3: 'chain + sqrt', 5: 'merge c', 6: 'pick + commit p (filter + merge)', 4: 'TOTAL (whole query)'}
for ef in efs:
    for _ in range(3):
        idx.search_batch_device(dQ.data_ptr(), nq, n, ef, ids.data_ptr(), dd.data_ptr(), cnt.data_ptr(), st.data_ptr(), 0)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record(); idx.search_batch_device(dQ.data_ptr(), nq, n, ef, ids.data_ptr(), dd.data_ptr(), cnt.data_ptr(), st.data_ptr(), 0); e1.record()
    torch.cuda.synchronize()
    D = dbg.cpu().numpy().astype(np.float64); S = st.cpu().numpy()
    ms = e0.elapsed_time(e1)
    tot = D[:, 4]
    passes = D[:, 7]
    print('== ef %d: kernel %.4f ms; n_exp %.1f n_dist %.1f passes %.1f (commits per pass %.2f)' % (
        ef, ms, S[:, 1].mean(), S[:, 0].mean(), passes.mean(), S[:, 1].mean() / passes.mean()))
    print('   total cycles per query: mean %.0f p50 %.0f p90 %.0f p99 %.0f max %.0f min %.0f ; kernel = %.0f cycles at 2.4 GHz -> implied clock if max wave spans the kernel: %.2f GHz' % (
        tot.mean(), np.percentile(tot, 50), np.percentile(tot, 90), np.percentile(tot, 99), tot.max(), tot.min(), ms * 2.4e6, tot.max() / (ms * 1e6)))
    for i in (0, 1, 2, 3, 5, 6, 4):
        print('   %-45s %9.0f cycles/query %5.1f%%   per pass %7.0f' % (names[i], D[:, i].mean(), 100 * D[:, i].mean() / tot.mean(), D[:, i].mean() / passes.mean()))
    acc = D[:, [0, 1, 2, 3, 5, 6]].sum(1).mean()
    print('   unaccounted (upper layers, staging, epilogue) %.1f%%' % (100 * (tot.mean() - acc) / tot.mean()))
    print('   n_exp: mean %.1f p90 %.0f max %.0f ; corr(total, n_exp) %.3f' % (S[:, 1].mean(), np.percentile(S[:, 1], 90), S[:, 1].max(), np.corrcoef(tot, S[:, 1])[0, 1]))
