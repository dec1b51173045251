"""statuses of the two-wave kernel on the bench's 1M index (HNSW_MI355X_PAIR=1, raw device entry: no host re-run)"""
import os, sys
sys.path.insert(0, '.')
import numpy as np, torch
import hnsw_rs_amd as H
N, d, m, n = 1_000_000, 100, 16, 10
vs = H.synth_rows(0, 0x5EED0001, 0, N, d, 32)
idx = H.HNSW.new(m, 32, d, H.VEC_F32); idx.set_device(0); idx.insert_bulk_device(vs, 32, False); idx.upload()
dev = torch.device("cuda:0")
qs = H.synth_rows(0, 0x5EED0002, 0, 10240, d, 8)
for ef in (64, 68):
    nq = qs.shape[0]
    dQ = torch.from_numpy(qs).to(dev)
    ids = torch.empty((nq, n), dtype=torch.int32, device=dev); dd = torch.empty((nq, n), dtype=torch.float32, device=dev)
    cnt = torch.empty(nq, dtype=torch.int32, device=dev); st = torch.empty((nq, 4), dtype=torch.int32, device=dev)
    idx.search_batch_device(dQ.data_ptr(), nq, n, ef, ids.data_ptr(), dd.data_ptr(), cnt.data_ptr(), st.data_ptr(), 0)
    torch.cuda.synchronize()
    S = st.cpu().numpy()
    codes, c = np.unique(S[:, 3], return_counts=True)
    print('ef', ef, 'status codes', dict(zip(codes.tolist(), c.tolist())), 'bad queries', np.nonzero(S[:, 3])[0][:10].tolist(), flush=True)
    for qi in np.nonzero(S[:, 3])[0][:5]:
        print('  q', int(qi), 'n_dist', int(S[qi, 0]), 'n_exp', int(S[qi, 1]), 'sum_deg', int(S[qi, 2]), flush=True)
