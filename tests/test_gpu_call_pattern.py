"""The reference's own call pattern through the drop-in entry points: one query per call from many threads
(HNSW::ann_by_vector(&self, ...), hnsw/src/template.rs:306-335) and insert_vec followed at once by a search
(eval_glove/src/main.rs:37-41; template.rs:165-173).  Concurrent hnsw_search calls are gathered into one launch
and an insert_vec patches the live HBM snapshot; both must return exactly what the oracle returns."""
import threading

import numpy as np
import pytest

import hnsw_rs_amd as H
from oracle import oracle_py as O
from tests.util import assert_search_equal, oracle_from_product, rand_vectors, same_graph

pytestmark = pytest.mark.gpu


def build(n, d, m, kind, ef_cons=32, seed=1):
    vs = H.synth_rows(0, 0x5EED0001 + seed, 0, n, d)
    lv = O.draw_levels(n, m, seed)
    index = H.HNSW.new(m, ef_cons, d, kind).insert_bulk(vs, 8, False, levels=lv)
    return index, oracle_from_product(index, vs, lv), vs, lv


@pytest.mark.parametrize("kind", [H.VEC_QUANT8, H.VEC_F32])
def test_sixteen_threads_of_single_queries_are_coalesced_and_exact(kind):
    """16 threads x one query per call == the oracle, query by query; the calls were served by fewer launches"""
    index, orc, _, _ = build(20000, 100, 16, kind)
    qs = H.synth_rows(0, 0x5EED0002, 0, 512, 100)
    want_ids, _, want_c, _ = orc.search_batch(qs, 10, 64)
    index.upload()
    b0, q0 = index.stat("coalesced_batches"), index.stat("coalesced_queries")
    ids, counts, calls, wall, lat = index.search_threads(qs, 10, 64, threads=16, seconds=0.5)
    assert calls >= len(qs)
    assert np.array_equal(counts, want_c)
    assert np.array_equal(ids, want_ids), "coalesced single-query answers differ from the oracle"
    nb, nqd = index.stat("coalesced_batches") - b0, index.stat("coalesced_queries") - q0
    assert nqd == calls
    assert nb < calls, "16 concurrent callers were never gathered (%d launches for %d calls)" % (nb, calls)
    assert 1 < index.stat("coalesced_max_batch") <= 16
    # the same through Python threads and the mirror's ann_by_vector (the binding releases the GIL in the call)
    got = [None] * 64
    def work(t):
        for i in range(t, 64, 16):
            got[i] = index.ann_by_vector(qs[i], 10, 64)
    th = [threading.Thread(target=work, args=(t,)) for t in range(16)]
    [t.start() for t in th]
    [t.join() for t in th]
    for i in range(64):
        assert got[i] == [int(x) for x in want_ids[i][:want_c[i]]]


def test_coalescing_off_and_mixed_parameters():
    """coalesce_us < 0: every call launches by itself; callers with different (n, ef) never share a batch"""
    index, orc, _, _ = build(5000, 36, 8, H.VEC_QUANT8, ef_cons=16, seed=3)
    qs = H.synth_rows(0, 0x5EED0002, 0, 96, 36)
    index.upload()
    index.set_option("coalesce_us", -1)
    b0 = index.stat("coalesced_batches")
    ids, counts, calls, _, _ = index.search_threads(qs, 5, 32, threads=8, seconds=0.05)
    assert index.stat("coalesced_batches") == b0
    w_ids, _, w_c, _ = orc.search_batch(qs, 5, 32)
    assert np.array_equal(ids, w_ids) and np.array_equal(counts, w_c)
    index.set_option("coalesce_us", 30)
    params = [(5, 32), (10, 100), (3, 7), (1, 1)]
    want = {p: orc.search_batch(qs, p[0], p[1]) for p in params}
    errs = []
    def work(t):
        n, ef = params[t % len(params)]
        for rep in range(3):
            for i in range(len(qs)):
                got = index.ann_by_vector(qs[i], n, ef)
                w = [int(x) for x in want[(n, ef)][0][i][:want[(n, ef)][2][i]]]
                if got != w:
                    errs.append((t, i, n, ef))
    th = [threading.Thread(target=work, args=(t,)) for t in range(12)]
    [t.start() for t in th]
    [t.join() for t in th]
    assert not errs, errs[:5]


def test_a_failing_query_fails_alone_inside_a_batch():
    """a NaN query shares a launch with good ones: its caller gets HNSW_ERR_NAN_INPUT, the others their answers"""
    index, orc, _, _ = build(3000, 24, 8, H.VEC_F32, ef_cons=16, seed=5)
    qs = rand_vectors(64, 24, 9)
    bad = qs[0].copy()
    bad[3] = np.nan
    index.upload()
    index.set_option("coalesce_us", 2000)  # wide window: the callers below do meet
    want = orc.search_batch(qs, 4, 20)
    out, codes = {}, {}
    barrier = threading.Barrier(8)
    def work(t):
        for rep in range(20):
            barrier.wait()
            try:
                r = index.ann_by_vector(bad if t == 0 else qs[t], 4, 20)
                out[(t, rep)] = r
            except H.HnswError as e:
                codes[(t, rep)] = e.code
    th = [threading.Thread(target=work, args=(t,)) for t in range(8)]
    [t.start() for t in th]
    [t.join() for t in th]
    assert all(codes.get((0, rep)) == H._lib.ERR_NAN_INPUT for rep in range(20)), codes
    assert len(codes) == 20
    for t in range(1, 8):
        for rep in range(20):
            assert out[(t, rep)] == [int(x) for x in want[0][t][:want[2][t]]]
    assert index.stat("coalesced_max_batch") > 1


@pytest.mark.parametrize("kind,d,n0,steps", [(H.VEC_QUANT8, 100, 200000, 1000), (H.VEC_F32, 100, 30000, 300),
                                              (H.VEC_QUANT8, 36, 20000, 300), (H.VEC_F32, 128, 20000, 200)])
def test_insert_vec_then_search_patches_the_live_snapshot(kind, d, n0, steps):
    """steps x (insert_vec, ann_by_vector): no whole-snapshot upload after the first, every answer equal to the
    oracle's on the same graph (the oracle inserts the same vectors in lockstep: the host insertion is the literal
    algorithm, tests/test_host_build.py), the final HBM snapshot equal to a fresh upload of the final graph"""
    m = 16
    vs = H.synth_rows(0, 0x5EED0001, 0, n0, d)
    lv = O.draw_levels(n0, m, 7)
    index = H.HNSW.new(m, 32, d, kind)
    index.set_option("gpu_build", 2)  # the starting graph is built on the device (seconds, not minutes)
    index.insert_bulk(vs, 8, False, levels=lv)
    orc = oracle_from_product(index, vs, lv)
    new = H.synth_rows(0, 0x5EED0009, 0, steps, d)
    qs = H.synth_rows(0, 0x5EED0002, 0, steps, d)
    nl = O.draw_levels(steps, m, 11)
    nl[steps // 2] = index.nb_layers() + 1  # one insertion opens two new top layers and moves the entry point
    index.upload()
    up0, bytes0 = index.stat("uploads"), index.device_bytes()
    for i in range(steps):
        node = index.insert_vec(new[i], level=int(nl[i]))
        assert node == orc.insert_vec(new[i], int(nl[i])) == n0 + i
        got = index.ann_by_vector(qs[i], 10, 48)
        assert got == [int(x) for x in orc.ann_by_vector(qs[i], 10, 48)], "step %d" % i
        if i % 97 == 0:  # ... and the new point is found from its own vector
            assert index.ann_by_vector(new[i], 1, 48) == [node]
    assert index.stat("uploads") == up0, "an insert_vec threw the snapshot away"
    assert index.stat("point_patches") == steps and index.stat("patch_fallbacks") == 0
    assert index.device_bytes() < bytes0 * 1.3 + (1 << 20)
    assert same_graph(index, orc)
    # the patched snapshot against a fresh upload of the same host index: batch answers identical, and equal to the oracle
    allq = np.concatenate([qs[:128], new[:64]])
    patched = index.search_batch(allq, 10, 64)
    fresh = index.clone()
    assert_search_equal(fresh.search_batch(allq, 10, 64), patched, "fresh upload vs patched snapshot")
    assert_search_equal(patched, orc.search_batch(allq, 10, 64), "patched snapshot vs oracle")


def test_insert_vec_with_overflowing_rows_and_inline_rows():
    """rows above the adjacency stride get overflow lists through the patch; the inline-rows copy of
    layer 0 (8-bit rows, asked for explicitly) is rebuilt for the touched nodes"""
    d, m, n0 = 20, 4, 400
    vs = rand_vectors(n0, d, 21)
    lv = O.draw_levels(n0, m, 4)
    index = H.HNSW.new(m, 8, d, H.VEC_QUANT8).insert_bulk(vs, 1, False, levels=lv)
    index.set_option("inline_rows", 1)
    # every layer-0 row gets 40 more neighbours than the build gave it (stride 32 -> every row has an overflow
    # list): an insertion prunes the rows of the points it selects (template.rs:209-238) and each of the ~40
    # neighbours they drop is a touched row that STILL has more ids than slots, so the patch files overflow lists
    ids, offs, nbrs = index.get_layer(0).csr()
    rows = [set(nbrs[offs[i]:offs[i + 1]].tolist()) for i in range(len(ids))]
    for i in range(n0):
        for k in range(20):
            j = (i + 7 * k + 1) % n0
            rows[i].add(j)
            rows[j].add(i)
    flat = np.concatenate([np.array(sorted(r), dtype=np.uint32) for r in rows])
    o2 = np.zeros(len(ids) + 1, dtype=np.uint64)
    o2[1:] = np.cumsum([len(r) for r in rows])
    index.import_layer(0, ids, o2, flat)
    orc = oracle_from_product(index, vs, lv)
    index.upload()
    up0 = index.stat("uploads")
    new = rand_vectors(60, d, 22)
    qs = rand_vectors(60, d, 23)
    for i in range(60):
        assert index.insert_vec(new[i], level=0) == orc.insert_vec(new[i], 0)
        assert index.ann_by_vector(qs[i], 5, 30) == [int(x) for x in orc.ann_by_vector(qs[i], 5, 30)], "step %d" % i
    assert index.stat("uploads") == up0 and index.stat("patch_fallbacks") == 0
    assert same_graph(index, orc)
    assert max(index.get_layer(0).degree(i) for i in range(n0)) > 32
    assert_search_equal(index.search_batch(qs, 5, 40), orc.search_batch(qs, 5, 40), "overflow lists + inline rows, patched")
    index.set_option("inline_rows", 0)  # (drops the snapshot) the compact layout, uploaded afresh, and patched again
    for i in range(20):
        assert index.insert_vec(qs[i], level=0) == orc.insert_vec(qs[i], 0)
        assert index.ann_by_vector(new[i], 5, 30) == [int(x) for x in orc.ann_by_vector(new[i], 5, 30)], "compact, step %d" % i
    assert_search_equal(index.search_batch(qs, 5, 40), orc.search_batch(qs, 5, 40), "overflow lists, compact, patched")


def test_searching_threads_between_inserts():
    """phases of concurrent one-query callers alternating with insert_vec calls (never at the same time: the contract of
    include/hnsw_mi355x.h): the coalescer's batches outlive the snapshot's patches -- new points, a new top layer, a moved
    entry point -- and every answer is the oracle's on the graph as it stands in that phase"""
    d, m, n0 = 48, 8, 6000
    vs = H.synth_rows(0, 0x5EED0001, 0, n0, d)
    lv = O.draw_levels(n0, m, 13)
    index = H.HNSW.new(m, 16, d, H.VEC_QUANT8).insert_bulk(vs, 4, False, levels=lv)
    orc = oracle_from_product(index, vs, lv)
    qs = H.synth_rows(0, 0x5EED0002, 0, 160, d)
    new = H.synth_rows(0, 0x5EED0009, 0, 40, d)
    index.upload()
    up0 = index.stat("uploads")
    for phase in range(8):
        want = orc.search_batch(qs, 7, 40)
        ids, counts, calls, _, _ = index.search_threads(qs, 7, 40, threads=12, seconds=0.05)
        assert np.array_equal(ids, want[0]) and np.array_equal(counts, want[2]), "phase %d" % phase
        for i in range(5):
            v = new[phase * 5 + i]
            level = index.nb_layers() if (phase == 3 and i == 0) else 0  # phase 3 opens a new top layer
            assert index.insert_vec(v, level=level) == orc.insert_vec(v, level)
    assert index.stat("uploads") == up0 and index.stat("patch_fallbacks") == 0
    assert index.stat("coalesced_max_batch") > 1
