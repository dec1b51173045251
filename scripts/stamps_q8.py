"""Diagnostic: per-phase cycle shares of the lean quant8 kernel (stamps build; every stamp drains the
memory counters, so the adjacency prefetch shows up where it is waited for).
   HNSW_MI355X_LIB=hnsw_rs_amd/libhnsw_mi355x_stamps.so python scripts/stamps_q8.py [N] [ef ...]"""
import os, sys
sys.path.insert(0, '.')
import numpy as np
import torch
import hnsw_rs_amd as H
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
efs = [int(x) for x in sys.argv[2:]] or [68]
d, m, nq, n = 100, 16, 1024, 10
vs = H.synth_rows(0, 0x5EED0001, 0, N, d, 32)
idx = H.HNSW.new(m, 32, d, H.VEC_QUANT8)
idx.set_device(0)
idx.insert_bulk_device(vs, 32, False)
qs = H.synth_rows(0, 0x5EED0002, 0, nq, d, 8)
idx.upload()
dev = torch.device('cuda:0')
dQ = torch.from_numpy(qs).to(dev)
ids = torch.empty((nq, n), dtype=torch.int32, device=dev); dd = torch.empty((nq, n), dtype=torch.float32, device=dev)
cnt = torch.empty(nq, dtype=torch.int32, device=dev); st = torch.empty((nq, 4), dtype=torch.int32, device=dev)
dbg = torch.zeros((nq, 16), dtype=torch.int64, device=dev)
os.environ['HX_DBG_PTR'] = str(dbg.data_ptr())
names = {7: 'staging + entry + upper layers', 0: 'pick + adjacency row (prefetch waited for here)', 1: 'visited look + claim + counts',
         3: 'rows + chain + key', 8: 'merge', 4: 'TOTAL (whole query)'}
for ef in efs:
    for _ in range(3):
        idx.search_batch_device(dQ.data_ptr(), nq, n, ef, ids.data_ptr(), dd.data_ptr(), cnt.data_ptr(), st.data_ptr(), 0)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record(); idx.search_batch_device(dQ.data_ptr(), nq, n, ef, ids.data_ptr(), dd.data_ptr(), cnt.data_ptr(), st.data_ptr(), 0); e1.record()
    torch.cuda.synchronize()
    D = dbg.cpu().numpy().astype(np.float64); S = st.cpu().numpy()
    tot = D[:, 4]; passes = D[:, 6]
    print('== quant8 ef %d: kernel %.4f ms; n_exp %.1f n_dist %.1f layer-0 passes %.1f' % (ef, e0.elapsed_time(e1), S[:, 1].mean(), S[:, 0].mean(), passes.mean()))
    print('   total cycles per query: mean %.0f p50 %.0f p99 %.0f max %.0f' % (tot.mean(), np.percentile(tot, 50), np.percentile(tot, 99), tot.max()))
    for i in (7, 0, 1, 3, 8, 4):
        print('   %-50s %9.0f cycles/query %5.1f%%   per pass %7.0f' % (names[i], D[:, i].mean(), 100 * D[:, i].mean() / tot.mean(), D[:, i].mean() / passes.mean()))
    acc = D[:, [0, 1, 3, 7, 8]].sum(1).mean()
    print('   unaccounted %.1f%%' % (100 * (tot.mean() - acc) / tot.mean()))
