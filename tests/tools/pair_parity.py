"""(test infrastructure: uses the CPU oracle)  The two-wave search kernel (hnsw_rs_amd/csrc/pair_kernel.inc) against the
oracle: ids, distance bits, counts and traversal counters, on a built graph and on an imported graph whose layer-0 rows
overflow their slots (those queries leave the two-wave kernel and are run again by the one-wave kernel).  Run with
HNSW_MI355X_PAIR=1 in the environment (the switch is read once per process): tests/test_gpu_pair_kernel.py does."""
import os
import sys

sys.path.insert(0, '.')
sys.path.insert(0, 'tests')
import numpy as np

import hnsw_rs_amd as H
from oracle import oracle_py as O
from util import assert_search_equal, oracle_from_product, product_from_oracle, rand_vectors

assert os.environ.get("HNSW_MI355X_PAIR") == "1"
N, d, m, nq = 60000, 100, 16, 1024
vs = H.synth_rows(0, 0x5EED0001, 0, N, d, 8)
qs = H.synth_rows(0, 0x5EED0002, 0, nq, d, 8)
lv = H.draw_levels(m, N)
idx = H.HNSW.new(m, 32, d, H.VEC_F32).insert_bulk_device(vs, 8, False, levels=lv)
orc = oracle_from_product(idx, vs, lv)
for ef in (1, 7, 10, 40, 64, 65, 68, 100, 128):
    assert_search_equal(idx.search_batch(qs, 10, ef), orc.search_batch(qs, 10, ef, nthreads=8), "two-wave kernel, ef %d" % ef)
# one query per call (the reference's ann_by_vector) goes the same way
for i in range(8):
    want = orc.search_batch(qs[i:i + 1], 10, 68, nthreads=1)[0][0]
    assert idx.ann_by_vector(qs[i], 10, 68) == [int(x) for x in want[:10]]
# rows beyond their slots (degree > S0): extra symmetric edges grafted onto two hubs of layer 0, as
# tests/test_gpu_parity.py::test_rows_longer_than_the_adjacency_stride does -- the queries that meet such a row leave
# the two-wave kernel and are run again by the one-wave kernel
n2, m2 = 3000, 8
vs2, q2 = rand_vectors(n2, d, 3), rand_vectors(256, d, 4)
lv2 = O.draw_levels(n2, m2, 3)
o1 = O.OracleHNSW(m2, None, d, O.VEC_F32).insert_bulk(vs2, lv2)
ids, offs, nbrs = o1.layer_csr(0)
adj = {int(i): set(int(x) for x in nbrs[int(offs[k]):int(offs[k + 1])]) for k, i in enumerate(ids)}
for hub, cnt in ((5, 70), (9, 40)):
    for t in range(100, 100 + cnt):
        adj[hub].add(t)
        adj[t].add(hub)
rows = [sorted(adj[int(i)]) for i in ids]
offs2 = np.cumsum([0] + [len(r) for r in rows]).astype(np.uint64)
nbrs2 = np.concatenate([np.array(r, dtype=np.uint32) for r in rows])
o2 = O.OracleHNSW(m2, None, d, O.VEC_F32)
o2.import_points(vs2, lv2)
o2.import_layer(0, ids, offs2, nbrs2)
for l in range(1, o1.nb_layers):
    o2.import_layer(l, *o1.layer_csr(l))
o2.set_ep(o1.ep)
p2 = product_from_oracle(o2, vs2, lv2)
assert p2.get_layer(0).degree(5) > 64
for ef in (16, 64, 100):
    assert_search_equal(p2.search_batch(q2, 10, ef), o2.search_batch(q2, 10, ef), "two-wave kernel, overflowing rows, ef %d" % ef)
print("PAIR PARITY OK")
