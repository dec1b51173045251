"""search rate against launch size on the 1M x 100d index (gpurun): where latency ends and bandwidth begins"""
import os, sys, time
sys.path.insert(0, '.')
import numpy as np, torch
import hnsw_rs_amd as H
ef = int(sys.argv[1]) if len(sys.argv) > 1 else 68
N, d, m, n = 1000000, 100, 16, 10
vs = H.synth_rows(0, 0x5EED0001, 0, N, d, 32); qs = H.synth_rows(0, 0x5EED0002, 0, 32768, d, 16)
dev = torch.device('cuda:0'); dQ = torch.from_numpy(qs).to(dev)
for kind, name in ((H.VEC_F32, 'f32'), (H.VEC_QUANT8, 'quant8 inline rows'), (H.VEC_QUANT8, 'quant8 compact')):
    idx = H.HNSW.new(m, 32, d, kind)
    if name.endswith('compact'): idx.set_option('inline_rows', 0)
    idx.insert_bulk_device(vs, 32, False); idx.upload()
    row_bytes = d + 8 if kind == H.VEC_QUANT8 else 4 * d
    for B in (64, 256, 1024, 2048, 4096, 8192, 16384, 32768):
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
        bq = (s[:, 0] * row_bytes + s[:, 1] * 4 + s[:, 2] * 4 + 4 * d + 8 * n).mean()
        print('%-20s ef=%d launch of %5d queries: %8.3f ms  %6.2f M q/s  alg %5.0f GB/s (%.1f %% of 8 TB/s)' % (
            name, ef, B, ms, B / ms / 1e3, bq * B / ms / 1e6, bq * B / ms / 1e6 / 80), flush=True)
