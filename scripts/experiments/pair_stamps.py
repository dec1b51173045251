"""cycle accounting of the two-wave kernel (stamps build): HNSW_MI355X_LIB=.../libhnsw_mi355x_stamps.so HNSW_MI355X_PAIR=1"""
import os, sys
sys.path.insert(0, '.')
import numpy as np, torch
import hnsw_rs_amd as H
N, d, m, n, ef = 1_000_000, 100, 16, 10, 68
vs = H.synth_rows(0, 0x5EED0001, 0, N, d, 32)
idx = H.HNSW.new(m, 32, d, H.VEC_F32); idx.set_device(0); idx.insert_bulk_device(vs, 32, False); idx.upload()
dev = torch.device("cuda:0")
qs = H.synth_rows(0, 0x5EED0002, 0, 1024, d, 8)
for nq in (64, 1024):
    dQ = torch.from_numpy(qs[:nq]).to(dev)
    ids = torch.empty((nq, n), dtype=torch.int32, device=dev); dd = torch.empty((nq, n), dtype=torch.float32, device=dev)
    cnt = torch.empty(nq, dtype=torch.int32, device=dev); st = torch.empty((nq, 4), dtype=torch.int32, device=dev)
    dbg = torch.zeros((nq, 16), dtype=torch.int64, device=dev)
    os.environ["HX_DBG_PTR"] = str(dbg.data_ptr())
    for _ in range(3):
        idx.search_batch_device(dQ.data_ptr(), nq, n, ef, ids.data_ptr(), dd.data_ptr(), cnt.data_ptr(), st.data_ptr(), 0)
    torch.cuda.synchronize(); dbg.zero_()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    idx.search_batch_device(dQ.data_ptr(), nq, n, ef, ids.data_ptr(), dd.data_ptr(), cnt.data_ptr(), st.data_ptr(), 0)
    e1.record(); torch.cuda.synchronize()
    D = dbg.cpu().numpy().astype(np.float64); P = D[:, 6].mean()
    print('%d queries: %.4f ms; passes %.1f; walker per pass: gather %.0f, wait F %.0f, claims of p %.0f, send %.0f, check + next pair %.0f; layer 0 %.0f per pass; '
          'whole query %.0f cycles; keeper per pass: wait keys %.0f, merges %.0f, F %.0f; walker check alone %.0f' % (
              nq, e0.elapsed_time(e1), P, D[:, 0].mean() / P, D[:, 1].mean() / P, D[:, 2].mean() / P, D[:, 3].mean() / P, D[:, 4].mean() / P,
              D[:, 5].mean() / P, D[:, 7].mean(), D[:, 8].mean() / P, D[:, 9].mean() / P, D[:, 10].mean() / P, D[:, 11].mean() / P), flush=True)
