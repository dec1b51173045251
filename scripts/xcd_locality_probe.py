"""Diagnostic: does it pay to put neighbouring queries on the same XCD (block b runs on XCD b % 8)?  The same
1024-query batches in three orders: as generated; grouped by nearest of 256 pivots and laid out so that a
group shares an XCD; the same grouping laid out so that a group is spread over all XCDs."""
import sys
sys.path.insert(0, '.')
import numpy as np, torch
import hnsw_rs_amd as H
kind = {'f32': H.VEC_F32, 'q8': H.VEC_QUANT8}[sys.argv[1]]; ef = int(sys.argv[2]) if len(sys.argv) > 2 else 68
N, d, m, B, n, NB = 1000000, 100, 16, 1024, 10, 8
vs = H.synth_rows(0, 0x5EED0001, 0, N, d, 32); qs = H.synth_rows(0, 0x5EED0002, 0, NB * B, d, 8)
idx = H.HNSW.new(m, 32, d, kind); idx.insert_bulk_device(vs, 32, False); idx.upload()
piv = vs[np.random.default_rng(1).choice(N, 256, replace=False)]
dev = torch.device('cuda:0')
def timed(Q):
    dQ = torch.from_numpy(np.ascontiguousarray(Q)).to(dev)
    ids = torch.empty((NB * B, n), dtype=torch.int32, device=dev); dd = torch.empty((NB * B, n), dtype=torch.float32, device=dev)
    cnt = torch.empty(NB * B, dtype=torch.int32, device=dev); st = torch.empty((NB * B, 4), dtype=torch.int32, device=dev)
    def run(b):
        o = b * B
        idx.search_batch_device(dQ[o:].data_ptr(), B, n, ef, ids[o:].data_ptr(), dd[o:].data_ptr(), cnt[o:].data_ptr(), st[o:].data_ptr(), 0)
    for b in range(NB): run(b)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for r in range(4):
        for b in range(NB): run(b)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (4 * NB)
orders = {'as generated': [], 'a group shares an XCD': [], 'a group spread over the XCDs': []}
for b in range(NB):
    q = qs[b * B:(b + 1) * B]
    key = ((q[:, None, :] - piv[None, :, :]) ** 2).sum(-1).argmin(1)
    srt = np.argsort(key, kind='stable')
    p = np.arange(B)
    orders['as generated'].append(q)
    orders['a group shares an XCD'].append(q[srt[(p % 8) * (B // 8) + p // 8]])   # block p -> XCD p % 8
    orders['a group spread over the XCDs'].append(q[srt])                           # neighbours in consecutive blocks
for name, lst in orders.items():
    for rep in range(2):
        print('%s ef %d, %-30s %.4f ms/batch' % (sys.argv[1], ef, name + ':', timed(np.concatenate(lst))), flush=True)
