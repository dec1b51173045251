import sys, time; sys.path.insert(0, '.')
import numpy as np, hnsw_rs_amd as H
N, d, m = 1000000, 100, 16
vs = H.synth_rows(0, 0x5EED0001, 0, N, d, 32); qs = H.synth_rows(0, 0x5EED0002, 0, 10240, d, 8)
for kind, name in ((H.VEC_F32, 'f32'), (H.VEC_QUANT8, 'quant8')):
    for rep in range(2):
        t = time.time(); idx = H.HNSW.new(m, 32, d, kind).insert_bulk(vs, 32, False); tb = time.time() - t
        truth, _ = idx.brute_force(qs, 10)
        out = []
        for ef in (64, 68):
            ids, _, _, _ = idx.search_batch(qs, 10, ef)
            out.append('ef%d %.5f' % (ef, sum(len(set(a) & set(b)) for a, b in zip(ids.tolist(), truth.tolist())) / 102400))
        print(name, 'cpu build %.1fs' % tb, *out, flush=True)
