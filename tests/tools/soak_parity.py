"""(test infrastructure: uses the CPU oracle; run from the repo root: python tests/tools/soak_parity.py)
parity soak on a mid-size index (gpurun): many queries x many efSearch values, both vector kinds,
ids / distances / counters against the CPU oracle holding the same graph"""
import sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
import hnsw_rs_amd as H
from oracle import oracle_py as O
from util import oracle_from_product
N, d, nq = 300000, 100, 4096
bad = 0
for kind in (H.VEC_F32, H.VEC_QUANT8):
    for m in (16, 6):
        vs = H.synth_rows(0, 0xABC0 + m, 0, N, d, 16); qs = H.synth_rows(0, 0xDEF0 + m, 0, nq, d, 8)
        lv = H.draw_levels(m, N)
        idx = H.HNSW.new(m, 32, d, kind).insert_bulk_device(vs, 16, False, levels=lv)
        orc = oracle_from_product(idx, vs, lv)
        for inline in ((1, 0) if kind == H.VEC_QUANT8 else (0,)):
            idx.set_option("inline_rows", inline)
            for ef in (1, 3, 10, 33, 64, 65, 100, 128, 129, 200, 300, 321, 400, 512):
                t = time.time()
                g_ids, g_d, g_c, g_st = idx.search_batch(qs, 10, ef)
                o_ids, o_d, o_c, o_st = orc.search_batch(qs, 10, ef, nthreads=16)
                ok = (np.array_equal(g_ids, o_ids) and np.array_equal(g_d.view(np.uint32), o_d.view(np.uint32))
                      and np.array_equal(g_c, o_c) and np.array_equal(np.asarray(g_st)[:, :3], np.asarray(o_st)[:, :3]))
                bad += 0 if ok else 1
                print('kind=%d m=%d inline=%d ef=%d: %s (%.1fs)' % (kind, m, inline, ef, 'identical' if ok else 'MISMATCH', time.time() - t), flush=True)
print('soak done, mismatching configurations:', bad)
sys.exit(1 if bad else 0)
