"""kernel time of a 1024-query launch on the bench's 1M index (raw device entry), HNSW_MI355X_PAIR as set by the caller"""
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
    nq = 1024
    dQ = [torch.from_numpy(qs[b * nq:(b + 1) * nq]).to(dev) for b in range(10)]
    ids = torch.empty((nq, n), dtype=torch.int32, device=dev); dd = torch.empty((nq, n), dtype=torch.float32, device=dev)
    cnt = torch.empty(nq, dtype=torch.int32, device=dev); st = torch.empty((nq, 4), dtype=torch.int32, device=dev)
    def run(b):
        idx.search_batch_device(dQ[b % 10].data_ptr(), nq, n, ef, ids.data_ptr(), dd.data_ptr(), cnt.data_ptr(), st.data_ptr(), 0)
    for b in range(10): run(b)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for b in range(50): run(b)
    e1.record(); torch.cuda.synchronize()
    S = st.cpu().numpy()
    print('PAIR=%s ef %d: %.4f ms per 1024-query launch (%.2f M q/s); statuses of the last launch: %s' % (
        os.environ.get('HNSW_MI355X_PAIR', '0'), ef, e0.elapsed_time(e1) / 50, nq / (e0.elapsed_time(e1) / 50) / 1e3,
        dict(zip(*[x.tolist() for x in np.unique(S[:, 3], return_counts=True)]))), flush=True)
