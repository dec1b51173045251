"""An INDEPENDENT second restatement of the reference's search path, in numpy / plain Python.

TEST INFRASTRUCTURE ONLY (like everything under oracle/): imported by tests/ and by the fixture
generator tests/golden/make_search_goldens.py, never by the product package.

It shares no code with oracle.cpp: the two ordered sets of `Results` are two real Python sorted
containers (`selected`, `candidates`) plus a `set` (`visited`), exactly the three structures of
hnsw/src/template/results.rs:26-33, and `search_layer` / `ann_by_vector` are written straight from
hnsw/src/template/searcher.rs:23-103 and hnsw/src/template.rs:306-335.  Arithmetic: every numpy float32
elementwise operation is ONE correctly rounded IEEE operation, so the vectorised loops below reproduce
vectors/src/quant.rs:14-66 and vectors/src/full.rs:23-29 bit for bit (the
reference's distance KATs are asserted against these functions in tests/test_golden_search.py).

Besides pinning the oracle it can record, per query, the trace of expansions that the design
simulations under scripts/ replay.
"""
import numpy as np
from sortedcontainers import SortedList

F = np.float32
VEC_QUANT8, VEC_F32 = 0, 1


# ---- vectors/src/quant.rs:41-66 ------------------------------------------------------------------
def quantize(v):
    """QuantVec::new -> (min, delta, codes).  (x - lb) / delta, + 0.5, floor, `as u8` (saturating,
    NaN -> 0: a constant vector has delta = 0 and every code 0)."""
    v = np.asarray(v, dtype=F)
    ub, lb = v.max(), v.min()
    delta = F(F(ub - lb) / F(255.0))
    with np.errstate(divide="ignore", invalid="ignore"):
        b = (v - lb) / delta
        b = b + F(0.5)
        b = np.floor(b)
    b = np.where(np.isnan(b), F(0.0), b)
    codes = np.clip(b, F(0.0), F(255.0)).astype(np.uint8)
    return lb, delta, codes


def dequant(mn, delta, codes):
    """(code as f32) * delta + min: two roundings (quant.rs:26-27)"""
    return codes.astype(F) * F(delta) + F(mn)


def dist_unrolled(xf, yf):
    """QuantVec::distance_unrolled on already dequantised rows xf [k, d] against yf [d] (quant.rs:14-37):
    8 running sums over whole chunks, the d % 8 tail into sum 0, then the left fold of the 8 sums."""
    xf = np.atleast_2d(xf)
    d = xf.shape[1]
    t = xf - yf[None, :]
    t2 = t * t
    acc = np.zeros((xf.shape[0], 8), dtype=F)
    full = d - d % 8
    for c in range(0, full, 8):
        acc = acc + t2[:, c:c + 8]
    for i in range(full, d):
        acc[:, 0] = acc[:, 0] + t2[:, i]
    s = np.zeros(xf.shape[0], dtype=F)
    for j in range(8):
        s = s + acc[:, j]
    return np.sqrt(s)


def dist_full(x, y):
    """FullVec::distance on rows x [k, d] against y [d] (full.rs:23-29): one left-to-right sum"""
    x = np.atleast_2d(x)
    t = x - y[None, :]
    t2 = t * t
    s = np.zeros(x.shape[0], dtype=F)
    for i in range(x.shape[1]):
        s = s + t2[:, i]
    return np.sqrt(s)


class Index:
    """points + layered adjacency + entry point, nothing else"""

    def __init__(self, vectors, kind, layers, ep):
        """vectors: [N, d] float32 as given to the reference; kind: VEC_QUANT8 (every stored vector is
        quantised by QuantVec::new, as the shipped `VecType`) or VEC_F32 (FullVec);
        layers: list over layer number of dict node -> sequence of neighbour ids; ep: entry point."""
        v = np.ascontiguousarray(vectors, dtype=F)
        self.kind, self.ep, self.layers = kind, int(ep), layers
        if kind == VEC_QUANT8:
            rows = np.empty_like(v)
            for i in range(v.shape[0]):
                mn, dl, codes = quantize(v[i])
                rows[i] = dequant(mn, dl, codes)
            self.rows = rows  # dequantised once: code * delta + min is what every distance recomputes
        else:
            self.rows = v

    @staticmethod
    def from_csr(vectors, kind, csr_layers, ep):
        layers = []
        for ids, offs, nbrs in csr_layers:
            layers.append({int(n): nbrs[int(offs[i]):int(offs[i + 1])] for i, n in enumerate(ids)})
        return Index(vectors, kind, layers, ep)

    def point(self, vector):
        """Point::new(vector) (template.rs:313): the query goes through the same VecType"""
        q = np.asarray(vector, dtype=F)
        if self.kind == VEC_QUANT8:
            return dequant(*quantize(q))
        return q

    def dists(self, ids, point):
        ids = np.asarray(ids, dtype=np.int64)
        if self.kind == VEC_QUANT8:
            return dist_unrolled(self.rows[ids], point)
        return dist_full(self.rows[ids], point)


class Results:
    """hnsw/src/template/results.rs:26-33: two ordered sets of Dist = (dist, id) and the visited set"""

    def __init__(self):
        self.selected = SortedList()
        self.candidates = SortedList()
        self.visited = set()


def _key(dist, node):
    if dist != dist:
        raise ValueError("NaN distance: Dist::cmp panics (graph/src/dist.rs:32)")
    return (float(dist), int(node))  # f32 -> f64 is exact, so tuple order == Dist::cmp


def search_layer(index, results, layer, point, ef, counters=None, trace=None):
    """Searcher::search_layer (searcher.rs:23-103)"""
    for e in results.selected:                       # extend_candidates_with_selected
        if e not in results.candidates:
            results.candidates.add(e)
    for e in results.selected:                       # extend_visited_with_selected
        results.visited.add(e[1])
    while len(results.candidates) > 0:
        cand = results.candidates.pop(0)             # pop_first
        furthest = results.selected[-1]              # last
        if cand > furthest:
            break
        if cand[1] not in layer:
            raise KeyError("Error in search_layer: %d not in Graph" % cand[1])
        fresh = []
        nbrs = layer[cand[1]]
        for n in nbrs:                               # .filter(|node| results.insert_visited(**node))
            n = int(n)
            if n not in results.visited:
                results.visited.add(n)
                fresh.append(n)
        if counters is not None:
            counters[0] += len(fresh)
            counters[1] += 1
            counters[2] += len(nbrs)
        if trace is not None:
            trace.append((cand[1], len(fresh)))
        if not fresh:
            continue
        ds = index.dists(fresh, point)
        for n, dd in zip(fresh, ds):
            k = _key(dd, n)
            f2q = results.selected[-1]
            if len(results.selected) < ef:
                results.selected.add(k)
                results.candidates.add(k)
                continue
            if k < f2q:
                results.selected.add(k)
                results.candidates.add(k)
                if len(results.selected) > ef:
                    results.selected.pop(-1)         # pop_last
    results.candidates.clear()
    results.visited.clear()


def ann_by_vector(index, vector, n, ef, with_trace=False):
    """HNSW::ann_by_vector (template.rs:306-335) -> (ids, dists, (n_dist, n_exp, sum_deg)[, trace])
    n_dist counts the entry point's distance too, like the oracle's and the kernels' counters."""
    point = index.point(vector)
    r = Results()
    r.selected.add(_key(index.dists([index.ep], point)[0], index.ep))
    counters = [1, 0, 0]
    trace = [] if with_trace else None
    for layer_nb in range(len(index.layers) - 1, 0, -1):
        search_layer(index, r, index.layers[layer_nb], point, 1, counters, None)
    search_layer(index, r, index.layers[0], point, ef, counters, trace)
    top = list(r.selected[:n])                       # get_top_selected(n)
    ids = np.array([k[1] for k in top], dtype=np.uint32)
    ds = np.array([k[0] for k in top], dtype=F)
    if with_trace:
        return ids, ds, tuple(counters), trace
    return ids, ds, tuple(counters)


def brute_force(index, vector, k):
    """helpers/glove.rs:94-109: full sort of Dist over every stored point"""
    point = index.point(vector)
    ds = index.dists(np.arange(index.rows.shape[0]), point)
    order = sorted((_key(dd, i) for i, dd in enumerate(ds)))[:k]
    return np.array([o[1] for o in order], dtype=np.uint32), np.array([o[0] for o in order], dtype=F)
