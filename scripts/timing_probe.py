"""Diagnostic: does the way a launch is timed change what the kernel takes?  Batch-1024 f32, efSearch 68:
(a) 40 launches behind one another, one event pair around all; (b) an event pair per launch (bench.py's
kernel_ms); (c) a device synchronisation after every launch."""
import sys, time
sys.path.insert(0, '.')
import numpy as np, torch
import hnsw_rs_amd as H
kind = {'f32': H.VEC_F32, 'q8': H.VEC_QUANT8}[sys.argv[1]]  # by name: the enum is QUANT8 = 0, F32 = 1
ef = int(sys.argv[2]) if len(sys.argv) > 2 else 68
N, d, m, B, n, NB = 1000000, 100, 16, 1024, 10, 10
vs = H.synth_rows(0, 0x5EED0001, 0, N, d, 32); qs = H.synth_rows(0, 0x5EED0002, 0, NB * B, d, 8)
idx = H.HNSW.new(m, 32, d, kind); idx.insert_bulk_device(vs, 32, False); idx.upload()
dev = torch.device('cuda:0'); dQ = torch.from_numpy(qs).to(dev)
ids = torch.empty((NB * B, n), dtype=torch.int32, device=dev); dd = torch.empty((NB * B, n), dtype=torch.float32, device=dev)
cnt = torch.empty(NB * B, dtype=torch.int32, device=dev); st = torch.empty((NB * B, 4), dtype=torch.int32, device=dev)
def run(b):
    o = b * B
    idx.search_batch_device(dQ[o:].data_ptr(), B, n, ef, ids[o:].data_ptr(), dd[o:].data_ptr(), cnt[o:].data_ptr(), st[o:].data_ptr(), 0)
for b in range(NB): run(b)
torch.cuda.synchronize()
K = 40
for rep in range(3):
    if rep == 2:  # what bench.py does before it times: the host-pointer search of 10240 queries (recall), the MFMA scan
        qh = H.synth_rows(0, 0x5EED0002, 0, 10240, d, 8)
        idx.search_batch(qh, n, ef); idx.brute_force_fast(qh, n)
        print('after a 10240-query host-pointer search and the MFMA scan:', flush=True)
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(K): run(i % NB)
    e1.record(); torch.cuda.synchronize()
    print('(a) one event pair around %d launches: %.4f ms/launch' % (K, e0.elapsed_time(e1) / K), flush=True)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(K)]
    t0 = time.perf_counter()
    for i in range(K):
        ev[i][0].record(); run(i % NB); ev[i][1].record()
    torch.cuda.synchronize(); el = time.perf_counter() - t0
    per = [x.elapsed_time(y) for x, y in ev]
    print('(b) event pair per launch: mean %.4f ms (min %.4f max %.4f); wall %.4f ms/launch' % (np.mean(per), min(per), max(per), el * 1e3 / K), flush=True)
    per_b = {}
    for i, p in enumerate(per): per_b.setdefault(i % NB, []).append(p)
    print('    per query batch: ' + ' '.join('%.3f' % np.mean(per_b[b]) for b in range(NB)), flush=True)
    t0 = time.perf_counter()
    for i in range(K):
        run(i % NB); torch.cuda.synchronize()
    el = time.perf_counter() - t0
    print('(c) synchronise after every launch: wall %.4f ms/launch' % (el * 1e3 / K), flush=True)
