"""f32 search rate against launch size on the 1M x 100d index (gpurun); HNSW_MI355X_HELPERS=0/1 and
HNSW_MI355X_LEAN=0 select the kernel variant for A/B runs"""
import os, sys
sys.path.insert(0, '.')
import numpy as np, torch
import hnsw_rs_amd as H
ef = int(sys.argv[1]) if len(sys.argv) > 1 else 68
sizes = [int(x) for x in sys.argv[2:]] or [64, 256, 512, 1024, 2048, 4096, 8192, 32768]
N, d, m, n = 1000000, 100, 16, 10
vs = H.synth_rows(0, 0x5EED0001, 0, N, d, 32); qs = H.synth_rows(0, 0x5EED0002, 0, 32768, d, 16)
dev = torch.device('cuda:0'); dQ = torch.from_numpy(qs).to(dev)
idx = H.HNSW.new(m, 32, d, H.VEC_F32)
idx.insert_bulk_device(vs, 32, False); idx.upload()
for B in sizes:
    ids = torch.empty((B, n), dtype=torch.int32, device=dev); dd = torch.empty((B, n), dtype=torch.float32, device=dev)
    cnt = torch.empty(B, dtype=torch.int32, device=dev); st = torch.empty((B, 4), dtype=torch.int32, device=dev)
    def run():
        idx.search_batch_device(dQ.data_ptr(), B, n, ef, ids.data_ptr(), dd.data_ptr(), cnt.data_ptr(), st.data_ptr(), 0)
    for _ in range(3): run()
    torch.cuda.synchronize()
    reps = max(4, 65536 // B // 4)
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): run()
    e1.record(); torch.cuda.synchronize(); ms = e0.elapsed_time(e1) / reps
    s = st.cpu().numpy().astype(np.int64)
    assert (s[:, 3] == 0).all()
    bq = (s[:, 0] * 400 + s[:, 1] * 4 + s[:, 2] * 4 + 4 * d + 8 * n).mean()
    print('f32 ef=%d launch of %5d queries: %8.4f ms  %6.2f M q/s  alg %5.0f GB/s (%.1f %% of 8 TB/s)' % (
        ef, B, ms, B / ms / 1e3, bq * B / ms / 1e6, bq * B / ms / 1e6 / 80), flush=True)
