"""Pins the CPU oracle (and the product's host arithmetic) to every known-answer the reference's
own tests hold for the search path.  No GPU needed."""
import numpy as np
import pytest

import hnsw_rs_amd as H
from oracle import oracle_py as O
from tests.util import rand_vectors

F = np.float32
SQRT2 = np.sqrt(F(2.0))

# vectors/src/quant.rs:154-194 and vectors/src/full.rs:99-139 (same constants)
KATS = [([0.5], [0.25], F(0.25)),
        ([0.75], [0.25], F(0.5)),
        ([0.0, 0.0], [0.0, 1.0], F(1.0)),
        ([1.0, 0.0], [0.0, 1.0], SQRT2),
        ([-1.0, 0.0], [0.0, 1.0], SQRT2),
        ([1.0, 0.0], [0.0, -1.0], SQRT2)]


@pytest.mark.parametrize("a,b,want", KATS)
def test_distance_kat_quant(a, b, want):
    assert O.dist_quant(a, b) == want  # assert_eq! in the reference: exact
    assert O.dist_quant(b, a) == want


@pytest.mark.parametrize("a,b,want", KATS)
def test_distance_kat_full(a, b, want):
    assert O.dist_full(a, b) == want
    assert O.dist_full(b, a) == want


def test_distance_self_is_zero_and_nonnegative():
    # quant.rs:145-152,196-201 / full.rs:90-97,141-146
    vs = rand_vectors(101, 128, 7)
    assert O.dist_quant(vs[0], vs[0]) == F(0.0)
    assert O.dist_full(vs[0], vs[0]) == F(0.0)
    for v in vs[1:]:
        assert O.dist_quant(vs[0], v) >= 0 and O.dist_full(vs[0], v) >= 0


def test_quant_error_below_one_percent():
    # vectors/tests/full_lvq_tests.rs:4-27: through the generic `distance`, 1000 pairs, 128-d U[0,1)
    a, b = rand_vectors(1000, 128, 11), rand_vectors(1000, 128, 12)
    worst = 0.0
    for x, y in zip(a, b):
        full = O.dist_full(x, y)
        qf = O.dist_generic_qf(x, y)
        qq = O.dist_generic_qq(x, y)
        worst = max(worst, abs(qf - full) / full, abs(qq - full) / full)
    assert worst < 0.01


def test_constant_vector_quantises_through_nan():
    # delta = 0 -> 0/0 = NaN -> `as u8` = 0; dequantises to min exactly
    mn, dl, codes = O.quantize([0.5, 0.5, 0.5])
    assert mn == F(0.5) and dl == F(0.0) and not codes.any()


def test_quantiser_matches_numpy_restatement():
    """independent second restatement (numpy float32 scalar ops are single IEEE ops)"""
    for v in rand_vectors(50, 37, 3) * F(7.0) - F(3.0):
        mn, dl, codes = O.quantize(v)
        lb, ub = v.min(), v.max()
        delta = F(F(ub - lb) / F(255.0))
        want = np.floor(F(0.5) + (v - lb) / delta).astype(np.uint8)
        assert mn == lb and dl == delta and np.array_equal(codes, want)


def test_distance_unrolled_matches_numpy_restatement():
    rng = np.random.Generator(np.random.PCG64(5))
    for d in (1, 7, 8, 9, 50, 100, 128):
        x, y = rng.random(d, dtype=np.float32), rng.random(d, dtype=np.float32) * F(3)
        mx, dx, cx = O.quantize(x)
        my, dy, cy = O.quantize(y)
        xf = cx.astype(np.float32) * dx + mx
        yf = cy.astype(np.float32) * dy + my
        t2 = (xf - yf) * (xf - yf)
        acc = np.zeros(8, dtype=np.float32)
        full = d - d % 8
        for c in range(0, full, 8):
            acc += t2[c:c + 8]
        for i in range(full, d):
            acc[0] = acc[0] + t2[i]
        s = F(0.0)
        for j in range(8):
            s = F(s + acc[j])
        assert O.dist_quant(x, y) == np.sqrt(s)
        # FullVec: one left-to-right sum
        s = F(0.0)
        for i in range(d):
            t = F(x[i] - y[i])
            s = F(s + F(t * t))
        assert O.dist_full(x, y) == np.sqrt(s)


def test_dist_total_order():
    # graph/src/dist.rs:30-38; hnsw/src/template/results.rs:223-231: equal distances, different ids
    assert O.dist_cmp(0, 0.5, 1, 0.5) == -1
    assert O.dist_cmp(1, 0.5, 0, 0.5) == 1
    assert O.dist_cmp(4, 0.0, 2, 0.5) == -1
    assert O.dist_cmp(3, 0.5, 3, 0.5) == 0
    assert O.dist_cmp(3, float("nan"), 3, 0.5) == -2  # Rust panics


def test_host_arithmetic_equals_oracle():
    """the product's host-side distance / quantiser (used by the build path and accessors)"""
    vs = rand_vectors(64, 50, 21) - F(0.3)
    lv = np.zeros(64, dtype=np.uint8)
    for kind in (H.VEC_QUANT8, H.VEC_F32):
        idx = H.HNSW.new(12, None, 50, kind)
        idx.import_points(vs, lv)
        orc = O.OracleHNSW(12, None, 50, kind)
        orc.import_points(vs, lv)
        for a in range(0, 64, 5):
            for b in range(0, 64, 7):
                assert idx.distance(a, b) == orc.distance(a, b)
            assert np.array_equal(idx.get_point(a).get_vals(), orc.get_vals(a))
            if kind == H.VEC_QUANT8:
                mn, dl, codes = idx.get_point(a).quant()
                omn, odl, ocodes, _ = orc.get_quant(a)
                assert mn == omn and dl == odl and np.array_equal(codes, ocodes)
        assert idx.distance(0, 64) is None  # Option<f32>::None


def test_recall_on_reference_test_data(testdata):
    """hnsw_glove_build_eval (hnsw/src/template.rs:518-572) on the oracle: M = 12, 1 thread,
    ef = 100, n = 10, brute force over the same quantised distances; recall > 0.99 and every
    layer's min degree > 0."""
    store, queries = testdata
    lv = O.draw_levels(1000, 12, 1)
    orc = O.OracleHNSW(12, None, 50).insert_bulk(store, lv)
    ids, _, counts, _ = orc.search_batch(queries, 10, 100)
    bf, _ = orc.brute_force(queries, 10)
    hits = sum(len(set(a) & set(b)) for a, b in zip(ids, bf))
    assert hits / (len(queries) * 10) > 0.99
    for l in range(orc.nb_layers):
        nodes = orc.layer_nodes(l)
        if len(nodes) > 1:
            assert min(len(orc.neighbors(l, n)) for n in nodes) > 0


def test_graph_invariants_after_build():
    # graph/src/graph.rs:305-432: symmetry, no self loops
    vs = rand_vectors(300, 10, 31)
    lv = O.draw_levels(300, 12, 2)
    orc = O.OracleHNSW(12, None, 10).insert_bulk(vs, lv)
    for l in range(orc.nb_layers):
        for n in orc.layer_nodes(l):
            for nb in orc.neighbors(l, n):
                assert nb != n
                assert n in orc.neighbors(l, nb)
