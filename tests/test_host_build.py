"""Host logic of the product (no GPU): the C ABI loads and exports every declared symbol, the
build path reproduces the oracle's literal restatement edge for edge, persistence round-trips in
the reference's byte format, and the reference's own unit tests for the index hold
(hnsw/src/template.rs:465-516,574-611)."""
import ctypes
import os
import re

import numpy as np
import pytest

import hnsw_rs_amd as H
from hnsw_rs_amd import _lib
from oracle import oracle_py as O
from tests.conftest import ROOT
from tests.util import rand_vectors, same_graph

DIM, N, M = 10, 100, 12  # template.rs:461-463


def test_abi_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "hnsw_mi355x.h")).read()
    declared = set(re.findall(r"\b(hnsw_[a-z_0-9]+)\s*\(", header))
    assert declared, "no declarations found"
    L = ctypes.CDLL(_lib.LIB_PATH)
    for name in sorted(declared):
        assert hasattr(L, name), "library does not export " + name
    assert declared == set(_lib.SYMBOLS), declared ^ set(_lib.SYMBOLS)


def test_hnsw_init():  # template.rs:465-468
    H.HNSW.new(12, None, 128)


def test_params_defaults():  # params.rs:20-42
    p = H.HNSW.new(12, None, 128).params
    assert (p.m, p.mmax, p.mmax0, p.ef_cons, p.dim) == (12, 12, 24, 24, 128)
    assert np.float32(p.ml) == O.default_ml(12)
    assert H.HNSW.new(16, 40, 8).params.ef_cons == 40


def test_hnsw_build_and_inserts():  # template.rs:470-504
    index = H.HNSW.new(M, None, DIM).insert_bulk(rand_vectors(N, DIM, 1), 1, False)
    assert index.len() == N
    index.insert_vec(rand_vectors(1, DIM, 2)[0])
    assert index.len() == N + 1
    index = index.insert_bulk(rand_vectors(N, DIM, 3), 1, False)
    assert index.len() == 2 * N + 1
    assert index.assert_param_compliance()


def test_can_not_add_different_dim():  # template.rs:506-516 #[should_panic]
    index = H.HNSW.new(12, None, 128).insert_bulk(rand_vectors(10, 128, 1), 1, False)
    with pytest.raises(H.HnswError) as e:
        index.insert_bulk(rand_vectors(10, 512, 2), 1, False)
    assert e.value.code == _lib.ERR_BAD_DIM
    with pytest.raises(H.HnswError):
        index.insert_bulk([[0.0] * 128, [0.0] * 127], 1, False)


def test_nan_rows_are_rejected():
    v = rand_vectors(10, 16, 1)
    v[3, 5] = np.nan
    for kind in (H.VEC_QUANT8, H.VEC_F32):
        with pytest.raises(H.HnswError) as e:
            H.HNSW.new(12, None, 16, kind).insert_bulk(v, 1, False)
        assert e.value.code == _lib.ERR_NAN_INPUT


def test_threaded_quantisation_reports_the_first_bad_row_and_leaves_the_index_untouched():
    """rows are quantised on nb_threads threads; a bad row anywhere rejects the whole call"""
    v = rand_vectors(30000, 8, 4)
    v[21000, 2] = np.nan
    v[9000, 1] = np.nan
    index = H.HNSW.new(8, None, 8)
    with pytest.raises(H.HnswError) as e:
        index.insert_bulk(v, 8, False)
    assert e.value.code == _lib.ERR_NAN_INPUT and "row 9000" in str(e.value)
    assert index.len() == 0
    # and the threaded path stores what the single-threaded one stores
    good = rand_vectors(20000, 8, 5)
    a = H.HNSW.new(8, None, 8)
    a.import_points(good, np.zeros(len(good), np.uint8))
    b = H.HNSW.new(8, None, 8).insert_bulk(good[:64], 8, False)  # small: below the threading threshold
    for i in (0, 17, 63):
        assert np.array_equal(a.get_point(i).get_vals(), b.get_point(i).get_vals())


@pytest.mark.parametrize("kind", [H.VEC_QUANT8, H.VEC_F32])
@pytest.mark.parametrize("n,d,m,seed", [(100, 10, 12, 1), (700, 16, 4, 2), (1500, 33, 8, 3)])
def test_single_thread_build_equals_oracle(kind, n, d, m, seed):
    vs = rand_vectors(n, d, seed)
    lv = O.draw_levels(n, m, seed)
    index = H.HNSW.new(m, None, d, kind).insert_bulk(vs, 1, False, levels=lv)
    orc = O.OracleHNSW(m, None, d, kind).insert_bulk(vs, lv)
    assert same_graph(index, orc)


def test_build_on_reference_test_data_equals_oracle(testdata):
    store, _ = testdata
    lv = O.draw_levels(1000, 12, 1)
    index = H.HNSW.new(12, None, 50).insert_bulk(store, 1, False, levels=lv)
    orc = O.OracleHNSW(12, None, 50).insert_bulk(store, lv)
    assert same_graph(index, orc)
    # H6: the degree cap can be exceeded; the layout must cope
    degs = [index.get_layer(0).degree(i) for i in range(1000)]
    assert max(degs) >= 24 and min(degs) > 0


def test_incremental_inserts_equal_oracle():
    vs, more = rand_vectors(300, 12, 5), rand_vectors(40, 12, 6)
    lv, lv2 = O.draw_levels(300, 6, 5), O.draw_levels(40, 6, 6)
    index = H.HNSW.new(6, 20, 12).insert_bulk(vs, 1, False, levels=lv)
    orc = O.OracleHNSW(6, 20, 12).insert_bulk(vs, lv)
    for v, l in zip(more[:10], lv2[:10]):
        assert index.insert_vec(v, level=int(l)) == orc.insert_vec(v, int(l))
    index.insert_bulk(more[10:], 1, False, levels=lv2[10:])
    orc.insert_bulk(more[10:], lv2[10:])
    assert same_graph(index, orc)


def test_multithreaded_build_is_a_valid_graph(testdata):
    store, _ = testdata
    index = H.HNSW.new(12, None, 50).insert_bulk(store, 4, False)
    assert index.len() == 1000 and index.assert_param_compliance()
    g = index.get_layer(0)
    for n in range(0, 1000, 37):
        for nb in g.neighbors(n):
            assert n in g.neighbors(nb) and nb != n


def test_default_levels_follow_the_geometric_law():
    lv = H.draw_levels(16, 200000)
    counts = np.bincount(lv)
    assert abs(counts[1] / counts[0] - 1 / 16) < 0.01 and counts.size <= 8
    assert np.array_equal(H.draw_levels(16, 50), lv[:50])  # reseeded with 0 at every call (Q9)


@pytest.mark.parametrize("kind", [H.VEC_QUANT8, H.VEC_F32])
def test_hnsw_serialize(tmp_path, kind):  # template.rs:574-611
    vs = rand_vectors(N, DIM, 9)
    index = H.HNSW.new(12, None, DIM, kind).insert_bulk(vs, 1, False)
    path = tmp_path / "serialization_test"
    index.save(path)
    loaded = H.HNSW.load(path)
    assert loaded.len() == N and loaded.vec_kind == kind
    for i in range(N):
        assert index.get_layer(0).neighbors(i) == loaded.get_layer(0).neighbors(i)
        assert np.array_equal(index.get_point(i).get_vals(), loaded.get_point(i).get_vals())
    p, q = index.params, loaded.params
    assert (p.ep, p.m, p.mmax, p.mmax0, p.ef_cons, p.dim) == (q.ep, q.m, q.mmax, q.mmax0, q.ef_cons, q.dim)
    # byte layout of the reference (params.rs:78-88: 52 bytes; points.rs:124-132)
    assert os.path.getsize(path / "params") == 52
    psize = 1 + (8 + DIM if kind == H.VEC_QUANT8 else 4 * DIM)
    assert os.path.getsize(path / "points") == 16 + N * psize
    raw = open(path / "params", "rb").read()
    assert int.from_bytes(raw[0:8], "big") == 12 and int.from_bytes(raw[36:44], "big") == DIM
    with pytest.raises(H.HnswError):
        H.HNSW.load(tmp_path / "missing")


def test_search_without_a_gpu_fails_loudly(gpu_available):
    if gpu_available:
        pytest.skip("a GPU is present")
    index = H.HNSW.new(12, None, DIM).insert_bulk(rand_vectors(N, DIM, 1), 1, False)
    with pytest.raises(H.HnswError) as e:
        index.ann_by_vector(rand_vectors(1, DIM, 2)[0], 10, 100)
    assert e.value.code in (_lib.ERR_NO_DEVICE, _lib.ERR_HIP)


def test_glove_text_loader_matches_the_binary_fixture(testdata, tmp_path):
    """load_glove_array (hnsw/src/helpers/glove.rs:14-71) on a text rendering of the fixture rows;
    '%.9g' round-trips every f32, and strtof is the correctly rounded parse Rust performs"""
    from hnsw_rs_amd.eval import load_glove_array
    store, _ = testdata
    path = tmp_path / "store.txt"
    with open(path, "w") as f:
        for i, row in enumerate(store[:50]):
            f.write("w%d " % i + " ".join("%.9g" % x for x in row) + "\n")
    words, arr = load_glove_array(0, path)
    assert words[:3] == ["w0", "w1", "w2"] and np.array_equal(arr, store[:50])
    words, arr = load_glove_array(7, path)
    assert arr.shape == (7, 50)
    with open(path, "a") as f:
        f.write("odd 1.0 2.0\n")
    with pytest.raises(ValueError):
        load_glove_array(0, path)


def test_snapshot_replication_argument_checks_without_a_device():
    """hnsw_snapshot_describe / _adopt / _commit refuse what they cannot serve before
    they touch a device"""
    import ctypes as C
    from hnsw_rs_amd import _lib
    L = _lib.lib()
    idx = H.HNSW.new(8, None, 10)
    desc = _lib.SnapshotDesc()
    assert L.hnsw_snapshot_describe(idx._h, C.byref(desc)) == _lib.ERR_EMPTY   # nothing to replicate yet
    assert L.hnsw_snapshot_commit(idx._h) == _lib.ERR_ARG                       # nothing adopted
    assert L.hnsw_snapshot_adopt(idx._h, C.byref(desc)) == _lib.ERR_ARG         # all-zero header: not a snapshot
    assert b"not a snapshot header" in L.hnsw_last_error()
    full = H.HNSW.new(8, None, 10).insert_bulk(rand_vectors(50, 10, 3), 1, False)
    desc.header[0] = 0x48584E53
    desc.header[1] = 1
    assert L.hnsw_snapshot_adopt(full._h, C.byref(desc)) == _lib.ERR_ARG        # receiver must be empty
    assert b"must be empty" in L.hnsw_last_error()


def _unit_rows(a):
    """the library's normalisation restated in numpy: one left-to-right f32 sum of squares (cumsum accumulates
    sequentially), f32 sqrt, f32 division"""
    a = np.ascontiguousarray(a, dtype=np.float32)
    ss = np.cumsum(a * a, axis=1, dtype=np.float32)[:, -1]
    return a / np.sqrt(ss, dtype=np.float32)[:, None]


def test_cosine_option_normalises_rows_on_the_way_in():
    """"metric_cosine" (an extension: the reference is Euclidean only, vectors/src/lib.rs:10-27): stored rows are
    the unit vectors, bit for bit what the numpy restatement gives; a zero vector is refused"""
    vs = rand_vectors(300, 24, 5) * np.float32(3.0) - np.float32(1.0)
    idx = H.HNSW.new(8, None, 24, H.VEC_F32)
    idx.set_option("metric_cosine", 1)
    idx.insert_bulk(vs, 2, False)
    want = _unit_rows(vs)
    got = np.stack([idx.get_point(i).get_vals() for i in range(300)])
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    node = idx.insert_vec(vs[7] * np.float32(10.0))
    assert np.array_equal(idx.get_point(node).get_vals().view(np.uint32), _unit_rows(vs[7:8] * np.float32(10.0))[0].view(np.uint32))
    with pytest.raises(H.HnswError):
        idx.insert_vec(np.zeros(24, dtype=np.float32))  # 0 / 0: no direction
    plain = H.HNSW.new(8, None, 24, H.VEC_F32).insert_bulk(vs, 2, False)
    assert np.array_equal(plain.get_point(3).get_vals(), vs[3])  # off by default


def test_handle_options_and_counters_without_a_device(gpu_available):
    """hnsw_set_option / hnsw_get_stat (round 4): the coalescer's knobs are range-checked, the counters exist and
    start at zero, unknown names are refused; the thread harness checks its arguments and -- like every search
    entry point -- fails loudly without a GPU instead of answering from the host"""
    idx = H.HNSW.new(8, None, 12).insert_bulk(rand_vectors(200, 12, 3), 1, False)
    for key in ("uploads", "point_patches", "patch_fallbacks", "coalesced_batches", "coalesced_queries", "coalesced_max_batch",
                "coalesce_ns_window", "coalesce_ns_turn", "coalesce_ns_gpu", "coalesce_ns_handout", "build_points",
                "build_batches", "build_rows_read", "build_adj_rows", "build_adj_ids", "build_records", "build_removals",
                "build_insert_kernel_us", "build_insert_phase_us", "build_connect_us"):
        assert idx.stat(key) == 0, key
    for bad in ("nope", "build_nope", "coalesce_ns"):
        with pytest.raises(H.HnswError):
            idx.stat(bad)
    idx.set_option("coalesce_us", 0)
    idx.set_option("coalesce_us", -1)
    idx.set_option("coalesce_depth", 4)
    idx.set_option("coalesce_max", 256)
    for key in ("coalesce_depth", "coalesce_max"):
        with pytest.raises(H.HnswError):
            idx.set_option(key, 0)
    # insert_vec on an index whose snapshot was never uploaded: nothing to patch, nothing counted
    node = idx.insert_vec(rand_vectors(1, 12, 4)[0])
    assert node == 200 and idx.stat("point_patches") == 0 and idx.stat("patch_fallbacks") == 0
    qs = rand_vectors(8, 12, 5)
    with pytest.raises(H.HnswError):
        idx.search_threads(qs, 3, 10, threads=0, seconds=0.01)
    if not gpu_available:
        with pytest.raises(H.HnswError) as e:
            idx.search_threads(qs, 3, 10, threads=2, seconds=0.01)
        assert e.value.code in (H._lib.ERR_NO_DEVICE, H._lib.ERR_HIP)
        idx.set_option("coalesce_us", 30)
        with pytest.raises(H.HnswError) as e:
            idx.ann_by_vector(qs[0], 3, 10)  # the coalesced path
        assert e.value.code in (H._lib.ERR_NO_DEVICE, H._lib.ERR_HIP)
