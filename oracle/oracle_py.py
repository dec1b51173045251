"""ctypes binding of the CPU oracle (oracle/liboracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg -- never by the product package hnsw_rs_amd/.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")

VEC_QUANT8 = 0
VEC_F32 = 1
UINT32_MAX = 0xFFFFFFFF

ERRORS = {0: "ok", -1: "bad dim", -2: "NaN", -3: "node not in graph", -4: "empty", -5: "bad argument"}


class OracleError(RuntimeError):
    def __init__(self, code):
        super().__init__("oracle error %d (%s)" % (code, ERRORS.get(code, "?")))
        self.code = code


def build(force=False):
    if force or not os.path.exists(_LIB_PATH):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _LIB_PATH


_lib = None

u8p = C.POINTER(C.c_uint8)
u32p = C.POINTER(C.c_uint32)
u64p = C.POINTER(C.c_uint64)
f32p = C.POINTER(C.c_float)


def lib():
    global _lib
    if _lib is not None:
        return _lib
    build()
    L = C.CDLL(_LIB_PATH)
    L.orc_quantize.argtypes = [f32p, C.c_uint32, f32p, f32p, u8p]
    L.orc_quantize.restype = C.c_int
    L.orc_dist_quant.argtypes = [C.c_uint32, u8p, C.c_float, C.c_float, u8p, C.c_float, C.c_float]
    L.orc_dist_quant.restype = C.c_float
    L.orc_dist_full.argtypes = [C.c_uint32, f32p, f32p]
    L.orc_dist_full.restype = C.c_float
    L.orc_dist_generic_qq.argtypes = L.orc_dist_quant.argtypes
    L.orc_dist_generic_qq.restype = C.c_float
    L.orc_dist_generic_qf.argtypes = [C.c_uint32, u8p, C.c_float, C.c_float, f32p]
    L.orc_dist_generic_qf.restype = C.c_float
    L.orc_dist_cmp.argtypes = [C.c_uint32, C.c_float, C.c_uint32, C.c_float]
    L.orc_dist_cmp.restype = C.c_int
    L.orc_level_from_uniform.argtypes = [C.c_float, C.c_float]
    L.orc_level_from_uniform.restype = C.c_uint8
    L.orc_default_ml.argtypes = [C.c_uint32]
    L.orc_default_ml.restype = C.c_float
    L.orc_new.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.c_int]
    L.orc_new.restype = C.c_void_p
    L.orc_free.argtypes = [C.c_void_p]
    L.orc_free.restype = None
    L.orc_clone.argtypes = [C.c_void_p]
    L.orc_clone.restype = C.c_void_p
    L.orc_insert_bulk.argtypes = [C.c_void_p, f32p, C.c_uint64, u8p]
    L.orc_insert_bulk.restype = C.c_int
    L.orc_insert_vec.argtypes = [C.c_void_p, f32p, C.c_uint8, u32p]
    L.orc_insert_vec.restype = C.c_int
    L.orc_import_points.argtypes = [C.c_void_p, f32p, C.c_uint64, u8p]
    L.orc_import_points.restype = C.c_int
    L.orc_import_points_quant.argtypes = [C.c_void_p, u8p, f32p, f32p, C.c_uint64, u8p]
    L.orc_import_points_quant.restype = C.c_int
    L.orc_import_layer.argtypes = [C.c_void_p, C.c_uint32, C.c_uint64, u32p, u64p, u32p]
    L.orc_import_layer.restype = C.c_int
    L.orc_set_ep.argtypes = [C.c_void_p, C.c_uint32]
    L.orc_set_ep.restype = None
    L.orc_len.argtypes = [C.c_void_p]
    L.orc_len.restype = C.c_uint64
    L.orc_ep.argtypes = [C.c_void_p]
    L.orc_ep.restype = C.c_uint32
    L.orc_nb_layers.argtypes = [C.c_void_p]
    L.orc_nb_layers.restype = C.c_uint32
    L.orc_layer_nb_nodes.argtypes = [C.c_void_p, C.c_uint32]
    L.orc_layer_nb_nodes.restype = C.c_uint64
    L.orc_layer_m.argtypes = [C.c_void_p, C.c_uint32]
    L.orc_layer_m.restype = C.c_uint32
    L.orc_layer_nodes.argtypes = [C.c_void_p, C.c_uint32, u32p, C.c_uint64]
    L.orc_layer_nodes.restype = C.c_uint64
    L.orc_neighbors.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, u32p, C.c_uint64]
    L.orc_neighbors.restype = C.c_int64
    L.orc_distance.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, f32p]
    L.orc_distance.restype = C.c_int
    L.orc_get_vals.argtypes = [C.c_void_p, C.c_uint32, f32p]
    L.orc_get_vals.restype = C.c_int
    L.orc_get_quant.argtypes = [C.c_void_p, C.c_uint32, u8p, f32p, f32p, u8p]
    L.orc_get_quant.restype = C.c_int
    L.orc_ann_by_vector.argtypes = [C.c_void_p, f32p, C.c_uint32, C.c_uint32, u32p, f32p, u32p, u64p]
    L.orc_ann_by_vector.restype = C.c_int
    L.orc_search_batch.argtypes = [C.c_void_p, f32p, C.c_uint64, C.c_uint32, C.c_uint32, u32p, f32p,
                                   u32p, u64p, C.c_int]
    L.orc_search_batch.restype = C.c_int
    L.orc_search_layer.argtypes = [C.c_void_p, C.c_uint32, f32p, u32p, C.c_uint32, C.c_uint32, u32p,
                                   f32p, u32p, u64p]
    L.orc_search_layer.restype = C.c_int
    L.orc_distance_batch.argtypes = [C.c_void_p, f32p, u32p, C.c_uint64, f32p]
    L.orc_distance_batch.restype = C.c_int
    L.orc_brute_force.argtypes = [C.c_void_p, f32p, C.c_uint64, C.c_uint32, u32p, f32p, C.c_int]
    L.orc_brute_force.restype = C.c_int
    _lib = L
    return L


def _p(a, t):
    return a.ctypes.data_as(t) if a is not None else None


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _check(rc):
    if rc != 0:
        raise OracleError(rc)


# ---- arithmetic ------------------------------------------------------------------------------
def quantize(v):
    """QuantVec::new -> (min, delta, codes)"""
    v = _f32(v)
    codes = np.zeros(v.shape[0], dtype=np.uint8)
    mn, dl = C.c_float(), C.c_float()
    _check(lib().orc_quantize(_p(v, f32p), v.shape[0], C.byref(mn), C.byref(dl), _p(codes, u8p)))
    return np.float32(mn.value), np.float32(dl.value), codes


def dist_quant(a, b):
    """QuantVec::new(a).dist2other(QuantVec::new(b))"""
    ma, da, ca = quantize(a)
    mb, db, cb = quantize(b)
    return np.float32(lib().orc_dist_quant(ca.shape[0], _p(ca, u8p), da, ma, _p(cb, u8p), db, mb))


def dist_quant_codes(ca, da, ma, cb, db, mb):
    return np.float32(lib().orc_dist_quant(ca.shape[0], _p(ca, u8p), da, ma, _p(cb, u8p), db, mb))


def dist_full(a, b):
    a, b = _f32(a), _f32(b)
    return np.float32(lib().orc_dist_full(a.shape[0], _p(a, f32p), _p(b, f32p)))


def dist_generic_qq(a, b):
    ma, da, ca = quantize(a)
    mb, db, cb = quantize(b)
    return np.float32(lib().orc_dist_generic_qq(ca.shape[0], _p(ca, u8p), da, ma, _p(cb, u8p), db, mb))


def dist_generic_qf(a, b):
    ma, da, ca = quantize(a)
    b = _f32(b)
    return np.float32(lib().orc_dist_generic_qf(ca.shape[0], _p(ca, u8p), da, ma, _p(b, f32p)))


def dist_cmp(id_a, d_a, id_b, d_b):
    return lib().orc_dist_cmp(id_a, d_a, id_b, d_b)


def default_ml(m):
    return np.float32(lib().orc_default_ml(m))


def level_from_uniform(r, ml):
    return int(lib().orc_level_from_uniform(float(r), float(ml)))


def draw_levels(n, m, seed):
    """Explicit level draws for tests/benches (NOT rand's StdRng -- see oracle.h): numpy PCG64
    uniforms pushed through the reference's own formula floor(-ln(r) * ml)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    r = rng.random(n, dtype=np.float32)
    r[r == 0.0] = np.float32(0.5)
    ml = float(default_ml(m))
    return np.array([level_from_uniform(x, ml) for x in r], dtype=np.uint8)


# ---- index -----------------------------------------------------------------------------------
class OracleHNSW:
    """Mirror of hnsw::template::HNSW on the oracle."""

    def __init__(self, m, ef_cons=None, dim=0, vec_kind=VEC_QUANT8, _handle=None):
        self.L = lib()
        self.m, self.dim, self.vec_kind = m, dim, vec_kind
        self.h = _handle if _handle is not None else self.L.orc_new(m, ef_cons or 0, dim, vec_kind)

    def __del__(self):
        if getattr(self, "h", None):
            self.L.orc_free(self.h)
            self.h = None

    def clone(self):
        return OracleHNSW(self.m, None, self.dim, self.vec_kind, _handle=self.L.orc_clone(self.h))

    def insert_bulk(self, vectors, levels):
        v = _f32(vectors)
        assert v.ndim == 2 and v.shape[1] == self.dim
        lv = np.ascontiguousarray(levels, dtype=np.uint8)
        assert lv.shape[0] == v.shape[0]
        _check(self.L.orc_insert_bulk(self.h, _p(v, f32p), v.shape[0], _p(lv, u8p)))
        return self

    def insert_vec(self, vector, level):
        v = _f32(vector)
        out = C.c_uint32()
        _check(self.L.orc_insert_vec(self.h, _p(v, f32p), int(level), C.byref(out)))
        return out.value

    def import_points(self, vectors, levels=None):
        v = _f32(vectors)
        lv = None if levels is None else np.ascontiguousarray(levels, dtype=np.uint8)
        _check(self.L.orc_import_points(self.h, _p(v, f32p), v.shape[0], _p(lv, u8p)))

    def import_points_quant(self, codes, mins, deltas, levels=None):
        codes = np.ascontiguousarray(codes, dtype=np.uint8)
        mins, deltas = _f32(mins), _f32(deltas)
        lv = None if levels is None else np.ascontiguousarray(levels, dtype=np.uint8)
        _check(self.L.orc_import_points_quant(self.h, _p(codes, u8p), _p(mins, f32p), _p(deltas, f32p),
                                              codes.shape[0], _p(lv, u8p)))

    def import_layer(self, layer, node_ids, offsets, nbrs):
        node_ids = np.ascontiguousarray(node_ids, dtype=np.uint32)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        nbrs = np.ascontiguousarray(nbrs, dtype=np.uint32)
        _check(self.L.orc_import_layer(self.h, layer, node_ids.shape[0], _p(node_ids, u32p),
                                       _p(offsets, u64p), _p(nbrs, u32p)))

    def set_ep(self, ep):
        self.L.orc_set_ep(self.h, int(ep))

    def __len__(self):
        return int(self.L.orc_len(self.h))

    @property
    def ep(self):
        return int(self.L.orc_ep(self.h))

    @property
    def nb_layers(self):
        return int(self.L.orc_nb_layers(self.h))

    def layer_nb_nodes(self, layer):
        return int(self.L.orc_layer_nb_nodes(self.h, layer))

    def layer_m(self, layer):
        return int(self.L.orc_layer_m(self.h, layer))

    def layer_nodes(self, layer):
        n = self.layer_nb_nodes(layer)
        out = np.zeros(n, dtype=np.uint32)
        self.L.orc_layer_nodes(self.h, layer, _p(out, u32p), n)
        return out

    def neighbors(self, layer, node):
        buf = np.zeros(4096, dtype=np.uint32)
        deg = self.L.orc_neighbors(self.h, layer, int(node), _p(buf, u32p), buf.shape[0])
        if deg < 0:
            raise KeyError(node)
        if deg > buf.shape[0]:  # a hub: ask again with room for the whole row
            buf = np.zeros(int(deg), dtype=np.uint32)
            deg = self.L.orc_neighbors(self.h, layer, int(node), _p(buf, u32p), buf.shape[0])
        return buf[:deg].copy()

    def layer_csr(self, layer):
        """(node_ids ascending, offsets u64, nbrs u32 ascending per row)"""
        nodes = self.layer_nodes(layer)
        offs = [0]
        rows = []
        for nid in nodes:
            r = self.neighbors(layer, nid)
            rows.append(r)
            offs.append(offs[-1] + len(r))
        nb = np.concatenate(rows) if rows else np.zeros(0, np.uint32)
        return nodes, np.array(offs, dtype=np.uint64), nb.astype(np.uint32)

    def distance(self, a, b):
        out = C.c_float()
        rc = self.L.orc_distance(self.h, int(a), int(b), C.byref(out))
        return None if rc != 0 else np.float32(out.value)

    def get_vals(self, node):
        out = np.zeros(self.dim, dtype=np.float32)
        _check(self.L.orc_get_vals(self.h, int(node), _p(out, f32p)))
        return out

    def get_quant(self, node):
        codes = np.zeros(self.dim, dtype=np.uint8)
        mn, dl, lv = C.c_float(), C.c_float(), C.c_uint8()
        _check(self.L.orc_get_quant(self.h, int(node), _p(codes, u8p), C.byref(mn), C.byref(dl),
                                    C.byref(lv)))
        return np.float32(mn.value), np.float32(dl.value), codes, lv.value

    def ann_by_vector(self, vector, n, ef, with_dists=False, with_stats=False):
        v = _f32(vector)
        ids = np.zeros(n, dtype=np.uint32)
        dists = np.zeros(n, dtype=np.float32)
        cnt = C.c_uint32()
        stats = np.zeros(3, dtype=np.uint64)
        _check(self.L.orc_ann_by_vector(self.h, _p(v, f32p), n, ef, _p(ids, u32p), _p(dists, f32p),
                                        C.byref(cnt), _p(stats, u64p)))
        out = [ids[: cnt.value].copy()]
        if with_dists:
            out.append(dists[: cnt.value].copy())
        if with_stats:
            out.append(stats)
        return out[0] if len(out) == 1 else tuple(out)

    def search_batch(self, Q, n, ef, nthreads=1):
        """-> ids[nq,n] (pad UINT32_MAX), dists[nq,n], counts[nq], stats[nq,3]"""
        Q = _f32(Q)
        nq = Q.shape[0]
        ids = np.zeros((nq, n), dtype=np.uint32)
        dists = np.zeros((nq, n), dtype=np.float32)
        counts = np.zeros(nq, dtype=np.uint32)
        stats = np.zeros((nq, 3), dtype=np.uint64)
        _check(self.L.orc_search_batch(self.h, _p(Q, f32p), nq, n, ef, _p(ids, u32p), _p(dists, f32p),
                                       _p(counts, u32p), _p(stats, u64p), nthreads))
        return ids, dists, counts, stats

    def search_layer(self, layer, q, entry_ids, ef):
        q = _f32(q)
        e = np.ascontiguousarray(entry_ids, dtype=np.uint32)
        cap = max(ef, e.shape[0]) + 1
        ids = np.zeros(cap, dtype=np.uint32)
        dists = np.zeros(cap, dtype=np.float32)
        cnt = C.c_uint32()
        stats = np.zeros(3, dtype=np.uint64)
        _check(self.L.orc_search_layer(self.h, layer, _p(q, f32p), _p(e, u32p), e.shape[0], ef,
                                       _p(ids, u32p), _p(dists, f32p), C.byref(cnt), _p(stats, u64p)))
        return ids[: cnt.value].copy(), dists[: cnt.value].copy(), stats

    def distance_batch(self, q, ids):
        q = _f32(q)
        ids = np.ascontiguousarray(ids, dtype=np.uint32)
        out = np.zeros(ids.shape[0], dtype=np.float32)
        _check(self.L.orc_distance_batch(self.h, _p(q, f32p), _p(ids, u32p), ids.shape[0], _p(out, f32p)))
        return out

    def brute_force(self, Q, k, nthreads=1):
        Q = _f32(Q)
        nq = Q.shape[0]
        ids = np.zeros((nq, k), dtype=np.uint32)
        dists = np.zeros((nq, k), dtype=np.float32)
        _check(self.L.orc_brute_force(self.h, _p(Q, f32p), nq, k, _p(ids, u32p), _p(dists, f32p), nthreads))
        return ids, dists
