"""Id-level golden vectors of the search path (tests/golden/search_*.npz, written by
tests/golden/make_search_goldens.py) against (a) the C++ oracle, (b) the independent numpy restatement
oracle/restate_np.py and (c), on the GPU box, the HIP kernels through the C ABI.

Reference: hnsw/src/template/searcher.rs:23-103, hnsw/src/template/results.rs:59-61,96-116,148-180,
hnsw/src/template.rs:306-335."""
import hashlib
import os

import numpy as np
import pytest

import hnsw_rs_amd as H
from oracle import oracle_py as O
from oracle import restate_np as R
from tests.util import rand_vectors

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
KINDS = {"quant8": O.VEC_QUANT8, "f32": O.VEC_F32}
F = np.float32


def load_set(name):
    z = np.load(os.path.join(GOLDEN, name))
    if name == "search_testdata.npz":
        store = np.load(os.path.join(GOLDEN, "testdata_store.npy"))
        queries = np.load(os.path.join(GOLDEN, "testdata_queries.npy"))
    else:
        store = H.synth_rows(0, 0x5EED0001, 0, 10000, 100, 4)
        queries = H.synth_rows(0, 0x5EED0002, 0, 64, 100, 1)
    # the inputs the vectors were made from (the synthetic generator is part of what is frozen)
    assert hashlib.sha256(np.ascontiguousarray(store).tobytes()).hexdigest() == str(z["store_sha256"])
    assert hashlib.sha256(np.ascontiguousarray(queries).tobytes()).hexdigest() == str(z["queries_sha256"])
    return z, store, queries


def csr_of(z, kind_name):
    return [(z["%s_l%d_ids" % (kind_name, l)], z["%s_l%d_offs" % (kind_name, l)], z["%s_l%d_nbrs" % (kind_name, l)])
            for l in range(int(z["%s_nb_layers" % kind_name]))]


def check_against(z, kind_name, ef, ids, dists, counts, stats, who):
    assert np.array_equal(counts, z["%s_ef%d_counts" % (kind_name, ef)]), "%s: counts (ef %d)" % (who, ef)
    assert np.array_equal(ids, z["%s_ef%d_ids" % (kind_name, ef)]), "%s: ids (ef %d)" % (who, ef)
    want_bits = z["%s_ef%d_dist_bits" % (kind_name, ef)]
    mask = ids != O.UINT32_MAX
    assert np.array_equal(np.ascontiguousarray(dists).view(np.uint32)[mask], want_bits[mask]), \
        "%s: distance bits (ef %d)" % (who, ef)
    assert np.array_equal(np.asarray(stats)[:, :3].astype(np.int64), z["%s_ef%d_stats" % (kind_name, ef)].astype(np.int64)), \
        "%s: counters (ef %d)" % (who, ef)


# ---- the restatement is pinned by the reference's own KATs too --------------------------------------
def test_numpy_restatement_passes_the_reference_kats():
    sq2 = np.sqrt(F(2.0))
    kats = [([0.5], [0.25], F(0.25)), ([0.75], [0.25], F(0.5)), ([0.0, 0.0], [0.0, 1.0], F(1.0)),
            ([1.0, 0.0], [0.0, 1.0], sq2), ([-1.0, 0.0], [0.0, 1.0], sq2), ([1.0, 0.0], [0.0, -1.0], sq2)]
    for a, b, want in kats:  # vectors/src/quant.rs:154-194, vectors/src/full.rs:99-139
        a, b = np.array(a, dtype=F), np.array(b, dtype=F)
        assert R.dist_full(a, b)[0] == want and R.dist_full(b, a)[0] == want
        qa, qb = R.dequant(*R.quantize(a)), R.dequant(*R.quantize(b))
        assert R.dist_unrolled(qa, qb)[0] == want and R.dist_unrolled(qb, qa)[0] == want
    mn, dl, codes = R.quantize([0.5, 0.5, 0.5])
    assert mn == F(0.5) and dl == F(0.0) and not codes.any()


@pytest.mark.parametrize("name", ["search_testdata.npz", "search_synth10k.npz"])
@pytest.mark.parametrize("kind_name", ["quant8", "f32"])
def test_oracle_reproduces_the_goldens(name, kind_name):
    z, store, queries = load_set(name)
    kind = KINDS[kind_name]
    orc = O.OracleHNSW(int(z["m"]), int(z["ef_cons"]) or None, store.shape[1], kind)
    orc.import_points(store, z["levels"])
    for l, (ids, offs, nbrs) in enumerate(csr_of(z, kind_name)):
        orc.import_layer(l, ids, offs, nbrs)
    orc.set_ep(int(z["%s_ep" % kind_name]))
    for ef in z["efs"]:
        ids, dists, counts, stats = orc.search_batch(queries, int(z["topn"]), int(ef))
        check_against(z, kind_name, int(ef), ids, dists, counts, stats, "oracle")


@pytest.mark.parametrize("kind_name", ["quant8", "f32"])
def test_oracle_build_reproduces_the_golden_graph(kind_name):
    """the sequential build is deterministic given the level draws: the frozen graph is what it builds"""
    z, store, _ = load_set("search_testdata.npz")
    orc = O.OracleHNSW(int(z["m"]), None, store.shape[1], KINDS[kind_name]).insert_bulk(store, z["levels"])
    assert orc.nb_layers == int(z["%s_nb_layers" % kind_name]) and orc.ep == int(z["%s_ep" % kind_name])
    for l, want in enumerate(csr_of(z, kind_name)):
        got = orc.layer_csr(l)
        assert all(np.array_equal(a, b) for a, b in zip(got, want)), "layer %d" % l


@pytest.mark.parametrize("kind_name", ["quant8", "f32"])
def test_numpy_restatement_reproduces_the_goldens(kind_name):
    z, store, queries = load_set("search_testdata.npz")
    idx = R.Index.from_csr(store, KINDS[kind_name], csr_of(z, kind_name), int(z["%s_ep" % kind_name]))
    n = int(z["topn"])
    for ef in (1, 64):
        for qi in range(0, 100, 3):
            ids, dists, cn = R.ann_by_vector(idx, queries[qi], n, ef)
            k = len(ids)
            assert k == z["%s_ef%d_counts" % (kind_name, ef)][qi]
            assert np.array_equal(ids, z["%s_ef%d_ids" % (kind_name, ef)][qi, :k])
            assert np.array_equal(dists.view(np.uint32), z["%s_ef%d_dist_bits" % (kind_name, ef)][qi, :k])
            assert cn == tuple(int(x) for x in z["%s_ef%d_stats" % (kind_name, ef)][qi])


@pytest.mark.parametrize("kind", [O.VEC_QUANT8, O.VEC_F32])
def test_two_restatements_agree_on_fresh_inputs(kind):
    """oracle.cpp vs restate_np.py on data neither fixture holds: odd dimension, ties, ef < n, ef > N"""
    vs = rand_vectors(400, 37, 91)
    vs[50] = vs[10]
    vs[51] = vs[10]  # equal distances: the id breaks the tie
    lv = O.draw_levels(400, 8, 5)
    orc = O.OracleHNSW(8, 20, 37, kind).insert_bulk(vs, lv)
    idx = R.Index.from_csr(vs, kind, [orc.layer_csr(l) for l in range(orc.nb_layers)], orc.ep)
    qs = np.concatenate([rand_vectors(20, 37, 92), vs[10:11]])
    for n, ef in ((10, 3), (10, 40), (5, 1), (10, 500)):
        ids, dists, counts, stats = orc.search_batch(qs, n, ef)
        for qi in range(qs.shape[0]):
            i2, d2, cn = R.ann_by_vector(idx, qs[qi], n, ef)
            k = len(i2)
            assert counts[qi] == k and np.array_equal(ids[qi, :k], i2)
            assert np.array_equal(dists[qi, :k].view(np.uint32), d2.view(np.uint32))
            assert tuple(int(x) for x in stats[qi]) == cn
        bf_ids, bf_d = orc.brute_force(qs[:4], 7)
        for qi in range(4):
            i2, d2 = R.brute_force(idx, qs[qi], 7)
            assert np.array_equal(bf_ids[qi], i2) and np.array_equal(bf_d[qi].view(np.uint32), d2.view(np.uint32))


# ---- the HIP path against the frozen vectors ----------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("name", ["search_testdata.npz", "search_synth10k.npz"])
@pytest.mark.parametrize("kind_name", ["quant8", "f32"])
def test_hip_search_reproduces_the_goldens(name, kind_name):
    z, store, queries = load_set(name)
    kind = H.VEC_QUANT8 if kind_name == "quant8" else H.VEC_F32
    idx = H.HNSW.new(int(z["m"]), int(z["ef_cons"]) or None, store.shape[1], kind)
    idx.import_points(store, z["levels"])
    for l, (ids, offs, nbrs) in enumerate(csr_of(z, kind_name)):
        idx.import_layer(l, ids, offs, nbrs)
    idx.set_ep(int(z["%s_ep" % kind_name]))
    for ef in z["efs"]:
        ids, dists, counts, stats = idx.search_batch(queries, int(z["topn"]), int(ef))
        assert (stats[:, 3] == 0).all()
        check_against(z, kind_name, int(ef), ids, dists, counts, stats, "HIP")
        one = idx.ann_by_vector(queries[0], int(z["topn"]), int(ef))  # the shim's nq = 1 path
        assert one == [int(x) for x in z["%s_ef%d_ids" % (kind_name, int(ef))][0][:len(one)]]
