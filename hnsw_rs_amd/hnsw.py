"""Host-side mirror of the reference's `hnsw::template::HNSW` (hnsw/src/template.rs) on top of the
C ABI of libhnsw_mi355x.so.  Same method names, argument meaning and error behaviour as the Rust
API, so that the parity tests read like the reference's own tests; the search itself runs in the
HIP kernels behind `hnsw_search_batch*` (there is no Python / CPU search path).

    index = HNSW.new(12, None, dim)                    # template.rs:133
    index = index.insert_bulk(vectors, 1, False)       # template.rs:388 (returns the index)
    node = index.insert_vec(vector)                    # template.rs:165
    ids = index.ann_by_vector(query, 10, 100)          # template.rs:306
    index.save(path); index = HNSW.load(path)          # template.rs:43,75
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import VEC_F32, VEC_QUANT8, HnswError, Params, QueryStats, check

_f32p, _u8p, _u32p, _u64p = _lib.f32p, _lib.u8p, _lib.u32p, _lib.u64p


def _p(a, t):
    return a.ctypes.data_as(t) if a is not None else None


class Point:
    """What get_point(id) exposes in the reference (points/src/point.rs:6-10)."""

    def __init__(self, index, node):
        self._index, self.id = index, node
        lv = C.c_uint32()
        check(index._L.hnsw_get_level(index._h, node, C.byref(lv)))
        self.level = lv.value

    def get_vals(self):  # VecBase::get_vals, vectors/src/lib.rs:24-26
        out = np.zeros(self._index.dim, dtype=np.float32)
        check(self._index._L.hnsw_get_vector(self._index._h, self.id, _p(out, _f32p)))
        return out

    def quant(self):
        """(min, delta, codes) of the stored QuantVec (vectors/src/quant.rs:6-11)"""
        codes = np.zeros(self._index.dim, dtype=np.uint8)
        mn, dl = C.c_float(), C.c_float()
        check(self._index._L.hnsw_get_quant(self._index._h, self.id, _p(codes, _u8p), C.byref(mn),
                                            C.byref(dl)))
        return np.float32(mn.value), np.float32(dl.value), codes


class Graph:
    """Read-only view of one layer (graph/src/graph.rs:9-16)."""

    def __init__(self, index, level):
        self._index, self.level = index, level
        self.m = int(index._L.hnsw_layer_m(index._h, level))

    def nb_nodes(self):
        return int(self._index._L.hnsw_layer_nb_nodes(self._index._h, self.level))

    def iter_nodes(self):
        n = self.nb_nodes()
        out = np.zeros(max(n, 1), dtype=np.uint32)
        cnt = C.c_uint64()
        check(self._index._L.hnsw_layer_nodes(self._index._h, self.level, _p(out, _u32p), n, C.byref(cnt)))
        return out[:n]

    def neighbors(self, node):  # Err(NodeNotInGraph) -> HnswError
        buf = np.zeros(1024, dtype=np.uint32)
        deg = C.c_uint32()
        check(self._index._L.hnsw_neighbors(self._index._h, self.level, int(node), _p(buf, _u32p), 1024,
                                            C.byref(deg)))
        return set(int(x) for x in buf[: deg.value])

    def neighbors_vec(self, node):
        return sorted(self.neighbors(node))

    def degree(self, node):
        deg = C.c_uint32()
        check(self._index._L.hnsw_neighbors(self._index._h, self.level, int(node), None, 0, C.byref(deg)))
        return deg.value

    def contains(self, node):
        try:
            self.degree(node)
            return True
        except HnswError:
            return False

    def csr(self):
        """(node ids ascending, offsets u64, neighbour ids ascending per row)"""
        nn, nnz = C.c_uint64(), C.c_uint64()
        L, h = self._index._L, self._index._h
        check(L.hnsw_export_layer(h, self.level, None, None, None, C.byref(nn), C.byref(nnz)))
        ids = np.zeros(nn.value, dtype=np.uint32)
        offs = np.zeros(nn.value + 1, dtype=np.uint64)
        nbrs = np.zeros(max(nnz.value, 1), dtype=np.uint32)
        check(L.hnsw_export_layer(h, self.level, _p(ids, _u32p), _p(offs, _u64p), _p(nbrs, _u32p),
                                  C.byref(nn), C.byref(nnz)))
        return ids, offs, nbrs[: nnz.value]


class HNSW:
    def __init__(self, handle, dim, vec_kind):
        self._L = _lib.lib()
        self._h = handle
        self.dim = dim
        self.vec_kind = vec_kind

    def __del__(self):
        if getattr(self, "_h", None):
            self._L.hnsw_free(self._h)
            self._h = None

    # ---- construction -----------------------------------------------------------------------
    @staticmethod
    def new(m, ef_cons, dim, vec_kind=VEC_QUANT8):
        """HNSW::new(m, ef_cons: Option<usize>, dim), template.rs:133-144"""
        L = _lib.lib()
        h = C.c_void_p()
        check(L.hnsw_create(m, ef_cons or 0, dim, vec_kind, C.byref(h)))
        return HNSW(h, dim, vec_kind)

    def clone(self):
        h = C.c_void_p()
        check(self._L.hnsw_clone(self._h, C.byref(h)))
        return HNSW(h, self.dim, self.vec_kind)

    @property
    def params(self):
        p = Params()
        check(self._L.hnsw_get_params(self._h, C.byref(p)))
        return p

    def set_ep(self, ep):
        check(self._L.hnsw_set_ep(self._h, int(ep)))

    def _rows(self, vectors):
        """Vec<Vec<f32>> -> n x dim float32; a row of another length is the reference's
        dimension-mismatch panic (template.rs:253-262)."""
        if isinstance(vectors, np.ndarray):
            a = np.ascontiguousarray(vectors, dtype=np.float32)
            if a.ndim != 2 or a.shape[1] != self.dim:
                raise HnswError(_lib.ERR_BAD_DIM,
                                "The current index dimension is %d, but tried inserting points of "
                                "dimension %s" % (self.dim, a.shape[1:] and a.shape[1]))
            return a
        for v in vectors:
            if len(v) != self.dim:
                raise HnswError(_lib.ERR_BAD_DIM,
                                "The current index dimension is %d, but tried inserting points of "
                                "dimension %d" % (self.dim, len(v)))
        return np.ascontiguousarray(np.array(vectors, dtype=np.float32).reshape(-1, self.dim))

    # ---- build --------------------------------------------------------------------------------
    def insert_bulk(self, vectors, nb_threads, verbose, levels=None):
        """HNSW::insert_bulk(self, vectors, nb_threads, verbose) -> Result<HNSW, String>"""
        rows = self._rows(vectors)
        lv = None if levels is None else np.ascontiguousarray(levels, dtype=np.uint8)
        if lv is not None and lv.shape[0] != rows.shape[0]:
            raise HnswError(_lib.ERR_ARG, "levels and vectors differ in length")
        check(self._L.hnsw_insert_bulk_levels(self._h, _p(rows, _f32p), rows.shape[0], nb_threads,
                                              1 if verbose else 0, _p(lv, _u8p)))
        return self

    def insert_bulk_device(self, vectors, nb_threads, verbose, levels=None):
        """insert_bulk with the insertion searches + heuristic on the GPU (on-device build)"""
        rows = self._rows(vectors)
        lv = None if levels is None else np.ascontiguousarray(levels, dtype=np.uint8)
        check(self._L.hnsw_insert_bulk_device(self._h, _p(rows, _f32p), rows.shape[0], nb_threads,
                                              1 if verbose else 0, _p(lv, _u8p)))
        return self

    def insert_bulk_sharded(self, vectors, nb_threads, verbose, levels=None, group=None, device=None):
        """The on-device build sharded over the ranks of a torch.distributed group (BASELINE configs[4]):
        every rank passes the same vectors / levels and ends with an identical replica; the insertion
        searches of each batch are split over the ranks by position, the connect / prune / drop phases by row
        ownership; edge records, removals and changed rows travel through all-gathers of the size the batch needs
        (RCCL when the group's backend is nccl; staged through the host for gloo)."""
        import torch
        import torch.distributed as dist
        from ._lib import ALLGATHER_FN
        world, rank = dist.get_world_size(group), dist.get_rank(group)
        dev = torch.device(device if device is not None else "cuda:%d" % torch.cuda.current_device())
        slot = int(self._L.hnsw_sharded_slot_bytes(self._h, world))
        send = torch.zeros(slot, dtype=torch.uint8, device=dev)
        recv = torch.zeros(world * slot, dtype=torch.uint8, device=dev)
        on_device = dist.get_backend(group) == "nccl"
        failure = []

        def allgather(_ctx, nbytes):
            try:
                if nbytes > slot or nbytes <= 0:
                    return 1
                # the first nbytes of every rank's send buffer, rank r's at r * nbytes of the receive buffer
                if on_device:
                    dist.all_gather_into_tensor(recv[: world * nbytes], send[:nbytes], group=group)
                    torch.cuda.synchronize(dev)
                else:  # gloo: CPU tensors
                    parts = [torch.empty(nbytes, dtype=torch.uint8) for _ in range(world)]
                    dist.all_gather(parts, send[:nbytes].cpu(), group=group)
                    recv[: world * nbytes].copy_(torch.cat(parts))
                    torch.cuda.synchronize(dev)
                return 0
            except Exception as e:  # never unwind through the C frames
                failure.append(e)
                return 2

        cb = ALLGATHER_FN(allgather)
        rows = self._rows(vectors)
        lv = None if levels is None else np.ascontiguousarray(levels, dtype=np.uint8)
        rc = self._L.hnsw_insert_bulk_sharded(self._h, _p(rows, _f32p), rows.shape[0], nb_threads,
                                              1 if verbose else 0, _p(lv, _u8p), rank, world,
                                              C.c_void_p(send.data_ptr()), C.c_void_p(recv.data_ptr()), slot, cb, None)
        if failure:
            raise failure[0]
        check(rc)
        return self

    def insert_vec(self, vector, level=None):
        """HNSW::insert_vec(&mut self, &Vec<f32>) -> Result<NodeID, String>"""
        v = self._rows([vector] if not isinstance(vector, np.ndarray) else vector.reshape(1, -1))
        out = C.c_uint32()
        check(self._L.hnsw_insert_vec_level(self._h, _p(v, _f32p), -1 if level is None else int(level),
                                            C.byref(out)))
        return out.value

    def import_points(self, vectors, levels):
        rows = self._rows(vectors)
        lv = np.ascontiguousarray(levels, dtype=np.uint8)
        check(self._L.hnsw_import_points(self._h, _p(rows, _f32p), rows.shape[0], _p(lv, _u8p)))

    def import_layer(self, layer, node_ids, offsets, nbrs):
        node_ids = np.ascontiguousarray(node_ids, dtype=np.uint32)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        nbrs = np.ascontiguousarray(nbrs, dtype=np.uint32)
        check(self._L.hnsw_import_layer(self._h, layer, node_ids.shape[0], _p(node_ids, _u32p),
                                        _p(offsets, _u64p), _p(nbrs, _u32p)))

    # ---- query (GPU) ----------------------------------------------------------------------------
    def ann_by_vector(self, vector, n, ef):
        """HNSW::ann_by_vector(&self, &Vec<f32>, n, ef) -> Result<Vec<NodeID>, String>"""
        q = np.ascontiguousarray(vector, dtype=np.float32).reshape(-1)
        if q.shape[0] != self.dim:
            raise HnswError(_lib.ERR_BAD_DIM, "query has dimension %d, index %d" % (q.shape[0], self.dim))
        ids = np.zeros(max(n, 1), dtype=np.uint32)
        cnt = C.c_uint32()
        check(self._L.hnsw_search(self._h, _p(q, _f32p), n, ef, _p(ids, _u32p), C.byref(cnt)))
        return [int(x) for x in ids[: cnt.value]]

    def search_batch(self, Q, n, ef):
        """-> ids [nq, n] (pad UINT32_MAX), dists [nq, n], counts [nq], stats [nq, 4]
        (n_dist, n_exp, sum_deg, status)"""
        Q = np.ascontiguousarray(Q, dtype=np.float32)
        if Q.ndim != 2 or Q.shape[1] != self.dim:
            raise HnswError(_lib.ERR_BAD_DIM, "queries must be nq x %d" % self.dim)
        nq = Q.shape[0]
        ids = np.full((nq, max(n, 1)), _lib.UINT32_MAX, dtype=np.uint32)
        dists = np.full((nq, max(n, 1)), np.inf, dtype=np.float32)
        counts = np.zeros(nq, dtype=np.uint32)
        stats = np.zeros((nq, 4), dtype=np.int32)
        check(self._L.hnsw_search_batch(self._h, _p(Q, _f32p), nq, n, ef, _p(ids, _u32p), _p(dists, _f32p),
                                        _p(counts, _u32p),
                                        C.cast(stats.ctypes.data, C.POINTER(QueryStats))))
        return ids[:, :n], dists[:, :n], counts, stats.view(np.uint32).astype(np.int64)

    def search_batch_device(self, d_Q, nq, n, ef, d_ids, d_dists, d_counts, d_stats, stream=0):
        """All arguments are raw device pointers (ints); enqueues on `stream`, no sync."""
        check(self._L.hnsw_search_batch_device(self._h, d_Q, nq, n, ef, d_ids, d_dists, d_counts, d_stats,
                                               stream))

    def search_batch_device_finish(self, d_Q, nq, n, ef, d_ids, d_dists, d_counts, d_stats, stream=0):
        """Completes search_batch_device: synchronises, re-runs queries whose visited table overflowed,
        raises the first per-query error."""
        check(self._L.hnsw_search_batch_device_finish(self._h, d_Q, nq, n, ef, d_ids, d_dists, d_counts, d_stats,
                                                      stream))

    def distance_batch(self, q, ids):
        """VecBase::dist2many seam on the device"""
        q = np.ascontiguousarray(q, dtype=np.float32).reshape(-1)
        ids = np.ascontiguousarray(ids, dtype=np.uint32)
        out = np.zeros(ids.shape[0], dtype=np.float32)
        check(self._L.hnsw_distance_batch(self._h, _p(q, _f32p), _p(ids, _u32p), ids.shape[0],
                                          _p(out, _f32p)))
        return out

    def search_layer(self, layer, q, entry_ids, ef):
        """Searcher::search_layer seam on the device -> (ids, dists, stats)"""
        q = np.ascontiguousarray(q, dtype=np.float32).reshape(-1)
        e = np.ascontiguousarray(entry_ids, dtype=np.uint32)
        ids = np.zeros(max(ef, 1), dtype=np.uint32)
        dists = np.zeros(max(ef, 1), dtype=np.float32)
        cnt = C.c_uint32()
        st = QueryStats()
        check(self._L.hnsw_search_layer(self._h, layer, _p(q, _f32p), _p(e, _u32p), e.shape[0], ef,
                                        _p(ids, _u32p), _p(dists, _f32p), C.byref(cnt), C.byref(st)))
        return ids[: cnt.value].copy(), dists[: cnt.value].copy(), (st.n_dist, st.n_exp, st.sum_deg)

    def brute_force(self, Q, k):
        """exact top-k under the index's own metric, on the device"""
        Q = np.ascontiguousarray(Q, dtype=np.float32)
        nq = Q.shape[0]
        ids = np.zeros((nq, k), dtype=np.uint32)
        dists = np.zeros((nq, k), dtype=np.float32)
        check(self._L.hnsw_brute_force(self._h, _p(Q, _f32p), nq, k, _p(ids, _u32p), _p(dists, _f32p)))
        return ids, dists

    def brute_force_fast(self, Q, k):
        """ground truth on the matrix cores (MFMA screen + exact re-rank of the k + 8 best): f32 rows only;
        not bit-exact by construction, see include/hnsw_mi355x.h"""
        Q = np.ascontiguousarray(Q, dtype=np.float32)
        nq = Q.shape[0]
        ids = np.zeros((nq, k), dtype=np.uint32)
        dists = np.zeros((nq, k), dtype=np.float32)
        check(self._L.hnsw_brute_force_fast(self._h, _p(Q, _f32p), nq, k, _p(ids, _u32p), _p(dists, _f32p)))
        return ids, dists

    # ---- accessors ------------------------------------------------------------------------------
    def len(self):
        return int(self._L.hnsw_len(self._h))

    def __len__(self):
        return self.len()

    def distance(self, a, b):
        """HNSW::distance(a, b) -> Option<f32>"""
        out = C.c_float()
        rc = self._L.hnsw_distance(self._h, int(a), int(b), C.byref(out))
        return None if rc != _lib.OK else np.float32(out.value)

    def get_point(self, node):
        """HNSW::get_point(id) -> Option<&Point>"""
        if node < 0 or node >= self.len():
            return None
        return Point(self, int(node))

    def nb_layers(self):
        return int(self._L.hnsw_layer_count(self._h))

    def get_layer(self, layer_nb):
        if layer_nb >= self.nb_layers():
            raise HnswError(_lib.ERR_ARG, "Layer %d not found in the structure." % layer_nb)
        return Graph(self, layer_nb)

    def iter_layers(self):
        return [Graph(self, l) for l in range(self.nb_layers())]

    def assert_param_compliance(self):
        ok = C.c_int()
        check(self._L.hnsw_check_param_compliance(self._h, C.byref(ok)))
        return bool(ok.value)

    # ---- persistence ------------------------------------------------------------------------------
    def save(self, path):
        check(self._L.hnsw_save(self._h, str(path).encode()))

    @staticmethod
    def load(path):
        L = _lib.lib()
        h = C.c_void_p()
        check(L.hnsw_load(str(path).encode(), C.byref(h)))
        p = Params()
        check(L.hnsw_get_params(h, C.byref(p)))
        return HNSW(h, int(p.dim), int(p.vec_kind))

    # ---- device -------------------------------------------------------------------------------------
    def set_device(self, device):
        check(self._L.hnsw_set_device(self._h, int(device)))

    def set_option(self, key, value):
        check(self._L.hnsw_set_option(self._h, key.encode(), int(value)))

    def upload(self):
        check(self._L.hnsw_upload(self._h))

    def device_bytes(self):
        b = C.c_uint64()
        check(self._L.hnsw_device_bytes(self._h, C.byref(b)))
        return b.value

    def stat(self, key):
        """hnsw_get_stat: "uploads", "point_patches", "patch_fallbacks", "coalesced_batches", "coalesced_queries",
        "coalesced_max_batch" """
        b = C.c_uint64()
        check(self._L.hnsw_get_stat(self._h, key.encode(), C.byref(b)))
        return b.value

    def batch_threads(self, Q, nq, n, ef, callers, calls):
        """hnsw_bench_batch_threads: `callers` host threads x `calls` hnsw_search_batch calls of nq queries each (host
        pointers in and out); -> queries per second over all callers"""
        Q = np.ascontiguousarray(Q, dtype=np.float32)
        if Q.ndim != 2 or Q.shape[1] != self.dim:
            raise HnswError(_lib.ERR_BAD_DIM, "queries must be total x %d" % self.dim)
        wall = C.c_double()
        check(self._L.hnsw_bench_batch_threads(self._h, _p(Q, _f32p), Q.shape[0], int(nq), n, ef, int(callers), int(calls),
                                               C.byref(wall)))
        return callers * calls * nq / wall.value

    def search_threads(self, Q, n, ef, threads, seconds):
        """The reference's call pattern as a load (hnsw_bench_search_threads): `threads` host threads, each blocked in
        its own one-query hnsw_search call.  -> (ids [nq, n], counts [nq], calls, wall seconds,
        {p50, p90, p99, max, mean} latency in us + cpu_user_s / cpu_sys_s of the process over the run)"""
        Q = np.ascontiguousarray(Q, dtype=np.float32)
        if Q.ndim != 2 or Q.shape[1] != self.dim:
            raise HnswError(_lib.ERR_BAD_DIM, "queries must be nq x %d" % self.dim)
        nq = Q.shape[0]
        ids = np.full((nq, max(n, 1)), _lib.UINT32_MAX, dtype=np.uint32)
        counts = np.zeros(nq, dtype=np.uint32)
        calls, wall = C.c_uint64(), C.c_double()
        lat = (C.c_double * 7)()
        check(self._L.hnsw_bench_search_threads(self._h, _p(Q, _f32p), nq, n, ef, int(threads), float(seconds),
                                                _p(ids, _u32p), _p(counts, _u32p), C.byref(calls), C.byref(wall), lat))
        return ids[:, :n], counts, calls.value, wall.value, dict(zip(("p50", "p90", "p99", "max", "mean", "cpu_user_s", "cpu_sys_s"), list(lat)))

    # ---- replication over the GPUs of a node ------------------------------------
    @staticmethod
    def replicate(index, m, ef_cons, dim, vec_kind, group=None, src=0, device=None, chunk_bytes=1 << 30):
        """Every rank of the torch.distributed group calls this; `index` is the built index on rank `src`
        (ignored elsewhere).  The snapshot's flat HBM arrays are broadcast from `src` -- RCCL over xGMI when
        the group's backend is nccl, staged through the host for gloo (the CPU tests) -- and every other
        rank gets a device-only replica that answers searches (hnsw_snapshot_describe / _adopt / _commit).
        Returns the source's own index on `src`, the replica elsewhere.  Arrays travel in pieces of at most
        chunk_bytes: a 51-GB row table is not one collective."""
        import torch
        import torch.distributed as dist
        from ._lib import SnapshotDesc
        rank = dist.get_rank(group)
        on_device = dist.get_backend(group) == "nccl"
        dev = torch.device(device if device is not None else "cuda:%d" % torch.cuda.current_device())
        L = _lib.lib()
        desc = SnapshotDesc()
        if rank == src:
            index.set_device(dev.index if dev.index is not None else 0)
            check(L.hnsw_snapshot_describe(index._h, C.byref(desc)))
            out = index
        else:
            out = HNSW.new(m, ef_cons, dim, vec_kind)
            out.set_device(dev.index if dev.index is not None else 0)
        # sizes + header: 7 + 32 words, one small broadcast
        meta = torch.zeros(7 + 32, dtype=torch.int64)
        if rank == src:
            meta[:7] = torch.tensor([int(desc.bytes[i]) for i in range(7)], dtype=torch.int64)
            meta[7:] = torch.tensor([int(desc.header[i]) for i in range(32)], dtype=torch.int64)
        meta = meta.to(dev) if on_device else meta
        dist.broadcast(meta, src=src, group=group)
        meta = meta.cpu()
        if rank != src:
            for i in range(7):
                desc.bytes[i] = int(meta[i])
            for i in range(32):
                desc.header[i] = int(meta[7 + i])
            check(L.hnsw_snapshot_adopt(out._h, C.byref(desc)))

        class _Mem:  # a view of library-owned device memory for torch (CUDA array interface)
            def __init__(self, ptr, nbytes):
                self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 2}

        for i in range(7):
            nbytes = int(desc.bytes[i])
            if nbytes == 0:
                continue
            whole = torch.as_tensor(_Mem(int(desc.ptr[i]), nbytes), device=dev)
            for lo in range(0, nbytes, chunk_bytes):
                piece = whole[lo:min(nbytes, lo + chunk_bytes)]
                if on_device:
                    dist.broadcast(piece, src=src, group=group)
                else:
                    host = piece.cpu() if rank == src else torch.empty(piece.shape[0], dtype=torch.uint8)
                    dist.broadcast(host, src=src, group=group)
                    if rank != src:
                        piece.copy_(host)
        torch.cuda.synchronize(dev)
        if rank != src:
            check(L.hnsw_snapshot_commit(out._h))
        return out


def synth_rows(recipe, seed, first_row, n, d, nb_threads=8):
    out = np.zeros((n, d), dtype=np.float32)
    check(_lib.lib().hnsw_synth_rows(recipe, seed, first_row, n, d, _p(out, _f32p), nb_threads))
    return out


def draw_levels(m, n):
    out = np.zeros(n, dtype=np.uint8)
    check(_lib.lib().hnsw_draw_levels(m, n, _p(out, _u8p)))
    return out


def device_count():
    c = C.c_int()
    check(_lib.lib().hnsw_device_count(C.byref(c)))
    return c.value
