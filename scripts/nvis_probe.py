import sys; sys.path.insert(0, '.')
import numpy as np, hnsw_rs_amd as H
N, d, m = 1000000, 100, 16
vs = H.synth_rows(0, 0x5EED0001, 0, N, d, 32); qs = H.synth_rows(0, 0x5EED0002, 0, 10240, d, 8)
idx = H.HNSW.new(m, 32, d, H.VEC_F32); idx.insert_bulk_device(vs, 32, False)
for ef in (64, 68, 96, 97, 128, 288):
    ids, _, _, st = idx.search_batch(qs, 10, ef)
    st = np.asarray(st)
    print('ef %d: n_dist mean %.0f p99 %.0f max %d; sum_deg max %d' % (ef, st[:, 0].mean(), np.percentile(st[:, 0], 99), st[:, 0].max(), st[:, 2].max()), flush=True)
