"""Evaluation harness: the reference's GloVe workflow on the GPU engine.

Mirrors, on top of the C ABI:
  - `load_glove_array(lim, file, verbose)`      hnsw/src/helpers/glove.rs:14-71
  - the recall procedure of `hnsw_glove_build_eval`   hnsw/src/template.rs:518-572
  - `eval_glove`'s main                          eval_glove/src/main.rs:17-42

    python -m hnsw_rs_amd.eval STORE.txt QUERIES.txt [--m 12] [--ef 100] [--n 10] [--threads 1]

Exact nearest neighbours come from the engine's own exhaustive GPU scan under the index's metric
(quantised-vs-quantised for the shipped QuantVec kind, exactly like template.rs:531-541).
"""
import argparse
import ctypes
import re
import sys
import time

import numpy as np

from . import VEC_F32, VEC_QUANT8
from .hnsw import HNSW


# the grammar of Rust's `impl FromStr for f32`: sign, then inf / infinity / nan (any case) or a decimal
# number with an optional exponent; no surrounding whitespace, no underscores, no hex (Python's float()
# would take "1_0" and " 1")
_RUST_F32 = re.compile(r"[+-]?(?:inf|infinity|nan|(?:[0-9]+\.?[0-9]*|\.[0-9]+)(?:[eE][+-]?[0-9]+)?)\Z", re.IGNORECASE)


def _parse_f32(tok):
    """Rust's str::parse::<f32>: correctly rounded decimal -> f32 (glibc strtof), None if not a number"""
    if not _RUST_F32.match(tok):
        return None
    libc = _parse_f32.libc
    if libc is None:
        libc = _parse_f32.libc = ctypes.CDLL(None)
        libc.strtof.restype = ctypes.c_float
        libc.strtof.argtypes = [ctypes.c_char_p, ctypes.c_void_p]
    return np.float32(libc.strtof(tok.encode(), None))


_parse_f32.libc = None


def load_glove_array(lim, path, verbose=False):
    """-> (words, embeddings [n, d] float32).  lim = 0 reads every line.  A token that does not parse
    as f32 is appended to the word (glove.rs:44-54); a row whose length differs from the first row's is an
    error, checked like the reference from the third row on (glove.rs:57 tests `embeddings.len() > 1`; a
    ragged second row surfaces when the array is formed)."""
    words, rows = [], []
    with open(path) as f:
        for idx, line in enumerate(f):
            if lim > 0 and idx >= lim:
                break
            parts = line.rstrip("\n").split(" ")
            word, vals = parts[0], []
            for tok in parts[1:]:
                v = _parse_f32(tok)
                if v is None:
                    word += tok
                else:
                    vals.append(v)
            if len(rows) > 1 and len(vals) != len(rows[0]):
                raise ValueError("Line %d: vector is not the same size as others. Len: %d, Word %s" % (
                    idx + 1, len(vals), word))
            rows.append(vals)
            words.append(word)
            if verbose and idx % 10000 == 0:
                print("\rLoading Embeddings %d" % idx, end="", file=sys.stderr)
    if len(rows) > 1 and len(rows[1]) != len(rows[0]):
        raise ValueError("Line 2: vector is not the same size as others. Len: %d, Word %s" % (len(rows[1]), words[1]))
    return words, np.array(rows, dtype=np.float32)


def recall_at(index, queries, n, ef):
    """hits / (nq * n) of ann_by_vector(q, n, ef) against the exhaustive top-n (template.rs:543-552)"""
    truth, _ = index.brute_force(queries, n)
    ids, _, _, _ = index.search_batch(queries, n, ef)
    hits = sum(len(set(a.tolist()) & set(b.tolist())) for a, b in zip(ids, truth))
    return hits / float(len(queries) * n)


def main():
    p = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    p.add_argument("store")
    p.add_argument("queries")
    p.add_argument("--lim", type=int, default=0)
    p.add_argument("--m", type=int, default=12)
    p.add_argument("--ef", type=int, default=100)
    p.add_argument("--n", type=int, default=10)
    p.add_argument("--threads", type=int, default=1)
    p.add_argument("--kind", choices=["quant8", "f32"], default="quant8")
    p.add_argument("--device-build", action="store_true",
                   help="build the index on the GPU (hnsw_insert_bulk_device) instead of the reference's CPU algorithm")
    a = p.parse_args()
    _, store = load_glove_array(a.lim, a.store, True)
    _, queries = load_glove_array(0, a.queries, False)
    kind = VEC_QUANT8 if a.kind == "quant8" else VEC_F32
    t = time.time()
    index = HNSW.new(a.m, None, store.shape[1], kind)
    if a.device_build:
        index.insert_bulk_device(store, max(1, a.threads), True)
    else:
        index.insert_bulk(store, a.threads, True)
    print("built %d points in %.2fs" % (index.len(), time.time() - t))
    print("Final accuracy was %.4f" % recall_at(index, queries, a.n, a.ef))
    for layer in index.iter_layers():
        degs = [layer.degree(x) for x in layer.iter_nodes()]
        print("Layer %d: limit is %d\nMin degree %d\nMax degree %d" % (layer.level, layer.m, min(degs), max(degs)))
    # eval_glove main: insert row 0 again and query it
    index.insert_vec(store[0])
    print("nearest =", index.ann_by_vector(store[0], a.n, a.ef))


if __name__ == "__main__":
    main()
