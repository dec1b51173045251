"""Diagnostic: per-phase cycle shares of the lean f32 kernel (stamps build).
   HNSW_MI355X_LIB=hnsw_rs_amd/libhnsw_mi355x_stamps.so python scripts/stamps_lean.py [N] [ef ...]"""
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
dbg = torch.zeros((nq, 16), dtype=torch.int64, device=dev)
os.environ['HX_DBG_PTR'] = str(dbg.data_ptr())
names = {0: 'pick c,p + adjacency row round trip', 1: 'visited: insert c, look up p', 2: 'row gather until landed',
         3: 'chain + sqrt + key', 8: 'merge c', 9: 'overflow rows + pick + is-p-next', 5: 'commit p: mark + visited insert',
         10: 'commit p: merge + pick', 7: 'staging + entry + upper layers', 4: 'TOTAL (whole query)'}
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
    passes = D[:, 6]
    print('== ef %d: kernel %.4f ms; n_exp %.1f n_dist %.1f layer-0 passes %.1f' % (ef, ms, S[:, 1].mean(), S[:, 0].mean(), passes.mean()))
    print('   total cycles per query: mean %.0f p50 %.0f p90 %.0f p99 %.0f max %.0f min %.0f ; implied clock if the max wave spans the kernel: %.2f GHz' % (
        tot.mean(), np.percentile(tot, 50), np.percentile(tot, 90), np.percentile(tot, 99), tot.max(), tot.min(), tot.max() / (ms * 1e6)))
    for i in (7, 0, 1, 2, 3, 8, 9, 5, 10, 4):
        print('   %-45s %9.0f cycles/query %5.1f%%   per pass %7.0f' % (names[i], D[:, i].mean(), 100 * D[:, i].mean() / tot.mean(), D[:, i].mean() / passes.mean()))
    acc = D[:, [0, 1, 2, 3, 5, 7, 8, 9, 10]].sum(1).mean()
    print('   unaccounted %.1f%%' % (100 * (tot.mean() - acc) / tot.mean()))
    print('   per query: visited slow-loop rounds %.1f; merges with 0 / 1-2 / >=3 survivors: %.1f / %.1f / %.1f; p commits %.1f' % (
        D[:, 11].mean(), D[:, 12].mean(), D[:, 13].mean(), D[:, 14].mean(), D[:, 15].mean()))
    # what do the slowest queries look like?  (the kernel lasts as long as its slowest wave)
    order = np.argsort(-tot)[:8]
    print('   slowest queries: total cycles, n_exp, n_dist, passes, cycles/pass, slow-loop rounds, merges 0/1-2/>=3, p commits')
    for qi in order:
        print('     q%-5d %8.0f  n_exp %4d n_dist %5d passes %4d  %6.0f/pass  slow %3d  merges %3d/%3d/%3d  pcommits %3d' % (
            qi, tot[qi], S[qi, 1], S[qi, 0], passes[qi], tot[qi] / max(1, passes[qi]), D[qi, 11], D[qi, 12], D[qi, 13], D[qi, 14], D[qi, 15]))
    cc = np.corrcoef(tot, S[:, 1])[0, 1]
    fit = np.polyfit(S[:, 1].astype(np.float64), tot, 1)
    print('   corr(total cycles, n_exp) = %.3f; fit: cycles = %.0f * n_exp + %.0f; n_exp mean %.1f p99 %.0f max %d' % (
        cc, fit[0], fit[1], S[:, 1].mean(), np.percentile(S[:, 1], 99), S[:, 1].max()))
    res = tot - np.polyval(fit, S[:, 1].astype(np.float64))
    print('   residual of that fit: std %.0f, max %.0f, min %.0f' % (res.std(), res.max(), res.min()))
