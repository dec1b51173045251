"""True recall@10 at efSearch 64 / 68 of the on-device build against its batch schedule (gpurun):
   usage: f32|q8 max:div [max:div ...]     e.g.  f32 8192:8 1024:32 256:64 64:256"""
import sys, time
sys.path.insert(0, '.')
import numpy as np
import hnsw_rs_amd as H
kind = {'f32': H.VEC_F32, 'q8': H.VEC_QUANT8}[sys.argv[1]]
N, d, m, n, NQ = 1000000, 100, 16, 10, 10240
vs = H.synth_rows(0, 0x5EED0001, 0, N, d, 32); qs = H.synth_rows(0, 0x5EED0002, 0, NQ, d, 8)
truth = None
for sched in sys.argv[2:]:
    bmax, bdiv = (int(x) for x in sched.split(':'))
    idx = H.HNSW.new(m, 32, d, kind)
    idx.set_device(0)
    idx.set_option("gpu_build_batch_max", bmax); idx.set_option("gpu_build_batch_div", bdiv)
    t = time.time(); idx.insert_bulk_device(vs, 16, False); tb = time.time() - t
    idx.upload()
    if truth is None:  # the same stored points whatever the schedule
        truth, _ = (idx.brute_force_fast(qs, n) if kind == H.VEC_F32 else idx.brute_force(qs, n))
    out = []
    for ef in (64, 68):
        got, _, _, st = idx.search_batch(qs, n, ef)
        hits = sum(len(set(a.tolist()) & set(b.tolist())) for a, b in zip(got, truth))
        out.append('ef %d: %.5f (n_dist %.0f)' % (ef, hits / float(NQ * n), st[:, 0].mean()))
    print('%s schedule max %5d, 1/%-4d build %5.1f s, %d layers: recall@10 %s' % (sys.argv[1], bmax, bdiv, tb, idx.nb_layers(), '; '.join(out)), flush=True)
    del idx
