"""batch-1024 time per efSearch through the GENERIC kernel (gpurun): 1M x d rows of a dimension / kind the lean kernels do
not serve.  usage: q8|f32 d ef..."""
import os, sys, time
sys.path.insert(0, ".")
import numpy as np, torch
import hnsw_rs_amd as H
kind = {"f32": H.VEC_F32, "q8": H.VEC_QUANT8}[sys.argv[1]]
d = int(sys.argv[2])
efs = [int(x) for x in sys.argv[3:]]
N, m, B, n = 1000000, 16, 1024, 10
vs = H.synth_rows(0, 0x5EED0001, 0, N, d, 32); qs = H.synth_rows(0, 0x5EED0002, 0, 8 * B, d, 8)
idx = H.HNSW.new(m, 32, d, kind); idx.insert_bulk_device(vs, 32, False); idx.set_option("inline_rows", 0); idx.upload()
dev = torch.device("cuda:0"); dQ = torch.from_numpy(qs).to(dev)
ids = torch.empty((8 * B, n), dtype=torch.int32, device=dev); dd = torch.empty((8 * B, n), dtype=torch.float32, device=dev)
cnt = torch.empty(8 * B, dtype=torch.int32, device=dev); st = torch.empty((8 * B, 4), dtype=torch.int32, device=dev)
for ef in efs:
    def run(b):
        o = b * B
        idx.search_batch_device(dQ[o:].data_ptr(), B, n, ef, ids[o:].data_ptr(), dd[o:].data_ptr(), cnt[o:].data_ptr(), st[o:].data_ptr(), 0)
    for b in range(8): run(b)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for b in range(8): run(b)
    e1.record(); torch.cuda.synchronize(); ms = e0.elapsed_time(e1) / 8
    s = st.cpu().numpy()
    print("1M x %dd %s ef=%d  %.4f ms/batch  %.2f M q/s  n_dist %.0f n_exp %.1f  us/exp %.2f  statuses %s" % (
        d, sys.argv[1], ef, ms, B / ms / 1e3, s[:, 0].mean(), s[:, 1].mean(), ms * 1e3 / s[:, 1].mean(), dict(zip(*np.unique(s[:, 3], return_counts=True)))), flush=True)
