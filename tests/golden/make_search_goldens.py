"""Generates the id-level golden fixtures of the search path ("Fixtures the build
should commit"):

    python tests/golden/make_search_goldens.py

  search_testdata.npz   the reference's own test-data (1000 x 50 store, 100 queries; M = 12 as in
                        hnsw/src/template.rs:518-572), both vector kinds
  search_synth10k.npz   10 000 x 100d synthetic rows (recipe A), 64 queries, M = 16, ef_construction = 32

Every file holds, per vector kind: the explicit level draws, the graph (CSR per layer) and entry point
the searches ran on, and for ef in {1, 10, 64, 100} the top-10 ids, the distance bit patterns, the result
counts and the traversal counters (n_dist, n_exp, sum_deg) of every query.

The reference is Rust and cannot run in this image (no rustc / cargo), and it
stores no expected id lists, so the vectors come from this repository's two independent restatements:
the C++ oracle (oracle/oracle.cpp) builds the graph and answers the queries, and the numpy restatement
(oracle/restate_np.py, two sorted containers + a set, written from searcher.rs / results.rs separately)
must reproduce every id, distance bit and counter before anything is written.  The fixtures freeze that
agreed behaviour: a later drift of the oracle AND the kernels together no longer goes unnoticed.
"""
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import hnsw_rs_amd as H  # noqa: E402  (synthetic rows only: hnsw_synth_rows is host code)
from oracle import oracle_py as O  # noqa: E402
from oracle import restate_np as R  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
EFS = (1, 10, 64, 100)
TOPN = 10
KINDS = {"quant8": O.VEC_QUANT8, "f32": O.VEC_F32}


def one_set(store, queries, m, ef_cons, level_seed):
    out = {"store_sha256": hashlib.sha256(np.ascontiguousarray(store).tobytes()).hexdigest(),
           "queries_sha256": hashlib.sha256(np.ascontiguousarray(queries).tobytes()).hexdigest(),
           "m": m, "ef_cons": ef_cons or 0, "level_seed": level_seed, "efs": np.array(EFS), "topn": TOPN}
    levels = O.draw_levels(store.shape[0], m, level_seed)
    out["levels"] = levels
    for name, kind in KINDS.items():
        orc = O.OracleHNSW(m, ef_cons, store.shape[1], kind).insert_bulk(store, levels)
        csr = [orc.layer_csr(l) for l in range(orc.nb_layers)]
        out["%s_ep" % name] = orc.ep
        out["%s_nb_layers" % name] = orc.nb_layers
        for l, (ids, offs, nbrs) in enumerate(csr):
            out["%s_l%d_ids" % (name, l)] = ids
            out["%s_l%d_offs" % (name, l)] = offs
            out["%s_l%d_nbrs" % (name, l)] = nbrs
        idx = R.Index.from_csr(store, kind, csr, orc.ep)
        for ef in EFS:
            ids, dists, counts, stats = orc.search_batch(queries, TOPN, ef)
            for qi in range(queries.shape[0]):
                i2, d2, cn = R.ann_by_vector(idx, queries[qi], TOPN, ef)
                k = len(i2)
                assert counts[qi] == k and np.array_equal(ids[qi, :k], i2), (name, ef, qi)
                assert np.array_equal(dists[qi, :k].view(np.uint32), d2.view(np.uint32)), (name, ef, qi)
                assert tuple(int(x) for x in stats[qi]) == cn, (name, ef, qi, stats[qi], cn)
            out["%s_ef%d_ids" % (name, ef)] = ids
            out["%s_ef%d_dist_bits" % (name, ef)] = dists.view(np.uint32)
            out["%s_ef%d_counts" % (name, ef)] = counts
            out["%s_ef%d_stats" % (name, ef)] = stats.astype(np.uint32)
        print("%s: %d layers, ep %d, both restatements agree on %d queries x %d ef values" % (
            name, orc.nb_layers, orc.ep, queries.shape[0], len(EFS)))
    return out


def main():
    store = np.load(os.path.join(HERE, "testdata_store.npy"))
    queries = np.load(os.path.join(HERE, "testdata_queries.npy"))
    np.savez_compressed(os.path.join(HERE, "search_testdata.npz"), **one_set(store, queries, 12, None, 1))
    store = H.synth_rows(0, 0x5EED0001, 0, 10000, 100, 4)
    queries = H.synth_rows(0, 0x5EED0002, 0, 64, 100, 1)
    np.savez_compressed(os.path.join(HERE, "search_synth10k.npz"), **one_set(store, queries, 16, 32, 7))


if __name__ == "__main__":
    main()
