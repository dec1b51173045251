"""Shared helpers of the parity tests."""
import numpy as np

import hnsw_rs_amd as H
from oracle import oracle_py as O


def oracle_from_product(index, vectors, levels):
    """An oracle index holding exactly the product index's points and graph (the oracle quantises
    the same float rows itself, so identical codes are part of what is being checked)."""
    orc = O.OracleHNSW(int(index.params.m), int(index.params.ef_cons), index.dim, index.vec_kind)
    orc.import_points(vectors, levels)
    for l in range(index.nb_layers()):
        ids, offs, nbrs = index.get_layer(l).csr()
        orc.import_layer(l, ids, offs, nbrs)
    orc.set_ep(int(index.params.ep))
    return orc


def product_from_oracle(orc, vectors, levels):
    idx = H.HNSW.new(orc.m, None, orc.dim, orc.vec_kind)
    idx.import_points(vectors, levels)
    for l in range(orc.nb_layers):
        ids, offs, nbrs = orc.layer_csr(l)
        idx.import_layer(l, ids, offs, nbrs)
    idx.set_ep(orc.ep)
    return idx


def same_graph(index, orc):
    if index.nb_layers() != orc.nb_layers or int(index.params.ep) != orc.ep:
        return False
    for l in range(orc.nb_layers):
        a, b = index.get_layer(l).csr(), orc.layer_csr(l)
        if not all(np.array_equal(x, y) for x, y in zip(a, b)):
            return False
    return True


def rand_vectors(n, d, seed):
    """make_rand_vectors (hnsw/src/template.rs:630-638): U[0,1)"""
    return np.random.Generator(np.random.PCG64(seed)).random((n, d), dtype=np.float32)


def assert_search_equal(got, want, what=""):
    """ids bit-exact, distances bit-exact (0 ulp; the stated tolerance is 1e-4), counters equal"""
    g_ids, g_d, g_c, g_s = got
    w_ids, w_d, w_c, w_s = want
    assert np.array_equal(g_c, w_c), what + " counts differ"
    bad = np.nonzero((g_ids != w_ids).any(axis=1))[0]
    assert bad.size == 0, "%s ids differ for queries %s: got %s want %s" % (
        what, bad[:5], g_ids[bad[:2]], w_ids[bad[:2]])
    mask = w_ids != O.UINT32_MAX
    assert np.array_equal(g_d[mask].view(np.uint32), w_d[mask].view(np.uint32)), what + " distances differ"
    assert np.allclose(g_d[mask], w_d[mask], rtol=0, atol=1e-4)
    assert np.array_equal(np.asarray(g_s)[:, :3], np.asarray(w_s)[:, :3].astype(np.int64)), \
        what + " counters (n_dist, n_exp, sum_deg) differ"
