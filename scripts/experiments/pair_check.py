import os, sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
import hnsw_rs_amd as H
from util import oracle_from_product
N, d, m, nq = 200000, 100, 16, 2048
vs = H.synth_rows(0, 0x5EED0001, 0, N, d, 16); qs = H.synth_rows(0, 0x5EED0002, 0, nq, d, 8)
lv = H.draw_levels(m, N)
idx = H.HNSW.new(m, 32, d, H.VEC_F32).insert_bulk_device(vs, 16, False, levels=lv)
orc = oracle_from_product(idx, vs, lv)
bad = 0
for ef in (1, 10, 33, 64, 65, 68, 100, 128):
    t = time.time()
    g_ids, g_d, g_c, g_st = idx.search_batch(qs, 10, ef)
    o_ids, o_d, o_c, o_st = orc.search_batch(qs, 10, ef, nthreads=16)
    st = np.asarray(g_st)
    ok = (np.array_equal(g_ids, o_ids) and np.array_equal(g_d.view(np.uint32), o_d.view(np.uint32))
          and np.array_equal(g_c, o_c) and np.array_equal(st[:, :3], np.asarray(o_st)[:, :3]))
    bad += 0 if ok else 1
    print('ef=%d: %s status!=0: %d ids_equal_rows %d/%d (%.1fs)' % (ef, 'identical' if ok else 'MISMATCH', int((st[:, 3] != 0).sum()),
          int((g_ids == o_ids).all(axis=1).sum()), nq, time.time() - t), flush=True)
print('mismatching:', bad)
