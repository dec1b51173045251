"""Query sharding over the GPUs of one node.

The search path shards embarrassingly over queries: every rank holds a full replica of the index
in its own HBM and answers a contiguous slice of the batch.  The only exchange is the one the
deployment has anyway -- queries arrive at one place and results return there -- so the data path
uses exactly two small collectives per batch over RCCL/xGMI (torch.distributed backend "nccl"):
    scatter  root -> rank r : Q[r*s : (r+1)*s]          (s x dim f32, ~400 KB at s = 1024)
    gather   rank r -> root : ids / dists of its slice   (s x n x 8 B, ~80 KB)
No collective touches the index after it has been replicated.  Messages are KB-scale, i.e. latency
bound; `ShardedSearcher.search` therefore lets the caller overlap them with the previous batch's
search by running on its own stream.

`local_search` is a callable so that the plumbing can be exercised with gloo on CPU (tests) while
production passes the HIP search (`make_device_search`).
"""
import torch
import torch.distributed as dist


def shard_size(nq, world):
    """queries per rank: ceil(nq / world), the last ranks may get padding"""
    return (nq + world - 1) // world


def shard_bounds(nq, world, rank):
    s = shard_size(nq, world)
    lo = min(nq, rank * s)
    return lo, min(nq, lo + s)


class ShardedSearcher:
    def __init__(self, local_search, dim, n, device, group=None, root=0, force_collectives=False):
        """local_search(Q [s, dim] f32 tensor on `device`) -> (ids [s, n] int32, dists [s, n] f32)"""
        self.local_search = local_search
        self.dim, self.n, self.device, self.group, self.root = dim, n, device, group, root
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.force_collectives = force_collectives and dist.is_initialized()

    def search(self, Q_root, nq):
        """Q_root: [nq, dim] on the root (ignored elsewhere).  Returns (ids, dists) of all nq
        queries on the root, (None, None) on the other ranks."""
        W, s = self.world, shard_size(nq, self.world)
        if W == 1 and not self.force_collectives:
            ids, dists = self.local_search(Q_root[:nq])
            return ids, dists
        q_local = torch.empty((s, self.dim), dtype=torch.float32, device=self.device)
        if self.rank == self.root:
            pad = W * s - nq
            Qp = Q_root[:nq]
            if pad:
                # padding rows repeat the last query: every slice stays a valid search input
                Qp = torch.cat([Qp, Qp[-1:].expand(pad, self.dim)], dim=0)
            chunks = [c.contiguous() for c in Qp.view(W, s, self.dim).unbind(0)]
            dist.scatter(q_local, chunks, src=self.root, group=self.group)
        else:
            dist.scatter(q_local, None, src=self.root, group=self.group)
        ids, dists = self.local_search(q_local)
        if self.rank == self.root:
            g_ids = [torch.empty_like(ids) for _ in range(W)]
            g_d = [torch.empty_like(dists) for _ in range(W)]
            dist.gather(ids, g_ids, dst=self.root, group=self.group)
            dist.gather(dists, g_d, dst=self.root, group=self.group)
            return torch.cat(g_ids, 0)[:nq], torch.cat(g_d, 0)[:nq]
        dist.gather(ids, None, dst=self.root, group=self.group)
        dist.gather(dists, None, dst=self.root, group=self.group)
        return None, None


def make_device_search(index, n, ef, max_queries, device):
    """local_search closure over the HIP path: device pointers in, device tensors out, enqueued on
    torch's current stream (no host synchronisation).  `local_search.check()` completes the last call:
    it synchronises, re-runs queries whose visited table overflowed and raises the first per-query error
    (hnsw_search_batch_device_finish)."""
    ids = torch.empty((max_queries, n), dtype=torch.int32, device=device)
    dists = torch.empty((max_queries, n), dtype=torch.float32, device=device)
    counts = torch.empty(max_queries, dtype=torch.int32, device=device)
    stats = torch.empty((max_queries, 4), dtype=torch.int32, device=device)
    last = {}

    def local_search(Q):
        nq = Q.shape[0]
        assert nq <= max_queries and Q.is_contiguous() and Q.dtype == torch.float32
        stream = torch.cuda.current_stream().cuda_stream
        index.search_batch_device(Q.data_ptr(), nq, n, ef, ids.data_ptr(), dists.data_ptr(),
                                  counts.data_ptr(), stats.data_ptr(), stream)
        last["call"] = (Q, nq, stream)
        return ids[:nq], dists[:nq]

    def check():
        Q, nq, stream = last["call"]
        index.search_batch_device_finish(Q.data_ptr(), nq, n, ef, ids.data_ptr(), dists.data_ptr(),
                                         counts.data_ptr(), stats.data_ptr(), stream)
        return ids[:nq], dists[:nq]

    local_search.check = check
    local_search.stats = stats
    local_search.counts = counts
    return local_search


class CudaLanes:
    """The two lanes of the pipeline as HIP streams (torch's stream objects on ROCm)."""

    def __init__(self, device):
        self.device = device
        self.compute = torch.cuda.current_stream(device)
        self.comm = torch.cuda.Stream(device=device)

    def event(self):
        return torch.cuda.Event()

    def on(self, lane):
        return torch.cuda.stream(lane)

    def wait(self, lane, ev):
        lane.wait_event(ev)

    def record(self, lane, ev):
        ev.record(lane)

    def drain(self):
        self.comm.synchronize()
        self.compute.synchronize()


class InlineLanes:
    """Everything runs at once on the caller's thread (CPU tensors, gloo): the tests' stand-in for the
    stream pair, which keeps the grouping / bucketing / status logic identical."""

    class _Nothing:
        def __enter__(self):
            return self

        def __exit__(self, *a):
            return False

    def __init__(self, device=None):
        self.device = device
        self.compute = self.comm = None

    def event(self):
        return None

    def on(self, lane):
        return InlineLanes._Nothing()

    def wait(self, lane, ev):
        pass

    def record(self, lane, ev):
        pass

    def drain(self):
        pass


class PipelinedShardedSearch:
    """The same exchange, bucketed and software-pipelined for throughput.  Collectives cost tens of
    microseconds of host and link latency each, a 1024-query search step ~200 us, so the exchange is
    done once per GROUP of up to `group` steps (fewer, larger messages: 3.2 MB of queries out, 0.6 MB
    of results back per rank at group = 8) and the collectives of one group run on a communication
    stream while the searches of the neighbouring group run on the compute stream:

        comm   : scatter(g0) scatter(g1) gather(g0) scatter(g2) gather(g1) ...
        compute:             search x8 (g0)          search x8 (g1)        ...

    ids and dists travel in ONE gather (a [G, 2, s, n] int32 buffer: ids, then the distance bits).
    Every step has its own statistics buffer; the number of queries of a group that did not finish with
    status 0 (visited table overflow, NaN) travels back with the results, one word per rank and group, and
    `results` reads a group's word before it hands out any of its rows: the rows of a group with failed queries are
    REFUSED (`FailedGroup`, carrying the per-rank counts; `with_status=True` returns rows and counts instead) -- a
    failed query never comes back unnoticed as a row of padding ids.  Every failing group is recorded (`failures`:
    group -> per-rank counts; `failure` = the first), on the root.  Failure is COLLECTIVE: only the root sees the
    words, so the library itself never raises in the middle of the agreed sequence of collectives (the other ranks
    would enter the next scatter / gather and wait for the RCCL timeout): every rank keeps issuing the same
    collectives, and `finish()` -- which every rank calls, also a root that caught a FailedGroup -- ends with a
    one-word all-reduce (MIN of every rank's own first failing group, enqueued right behind the last gather) after which
    EVERY rank raises.
    `submit(Q_root, g)` enqueues a group of g steps; `results(k, j)` (root only) returns step j of group k,
    valid from the submit after the group's own until `depth` more groups have been submitted; `finish()` drains
    everything and raises on every rank if any query of any group failed.  depth >= 2: a group's failure word is read one submit
    after its gather was enqueued, from a buffer the next gather of the same slot overwrites.

    search_step(q [s, dim], ids_out [s, n] int32, dist_bits_out [s, n] int32, stats_out [s, 4] int32, lane)
    enqueues one search on `lane`; `from_index` wraps the HIP path."""

    def __init__(self, search_step, dim, n, shard, device, group_steps=8, depth=2, group=None, root=0, lanes=None):
        assert depth >= 2, "depth 1 would let a gather overwrite the failure word of the group before it"
        self.search_step, self.dim, self.n, self.s = search_step, dim, n, shard
        self.device, self.group, self.root, self.depth, self.G = device, group, root, depth, group_steps
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.lanes = lanes if lanes is not None else CudaLanes(device)
        W, s, G = self.world, shard, group_steps
        self.q = [torch.empty((G, s, dim), dtype=torch.float32, device=device) for _ in range(depth)]
        # [G + 1, 2, s, n]: G steps of (ids, distance bits) and one trailing record whose first word is the
        # number of failed queries of the group
        self.res = [torch.zeros((G + 1, 2, s, n), dtype=torch.int32, device=device) for _ in range(depth)]
        self.stats = [torch.zeros((G, s, 4), dtype=torch.int32, device=device) for _ in range(depth)]
        self.out = ([torch.zeros((W, G + 1, 2, s, n), dtype=torch.int32, device=device) for _ in range(depth)]
                    if self.rank == root else None)
        self.ev_q = [self.lanes.event() for _ in range(depth)]
        self.ev_r = [self.lanes.event() for _ in range(depth)]
        self.ev_g = [self.lanes.event() for _ in range(depth)]
        # the root reads the failed-query words through a pinned host buffer filled on the communication lane:
        # a plain .cpu() would wait for the COMPUTE lane, i.e. stall the host behind the searches of the
        # group it has just enqueued (measured: 16 us per step at one rank)
        self.ev_c = [self.lanes.event() for _ in range(depth)]
        self.bad_host = None
        if self.rank == root:
            self.bad_host = [torch.zeros(W, dtype=torch.int32) for _ in range(depth)]
            if torch.device(device).type == "cuda":
                self.bad_host = [t.pin_memory() for t in self.bad_host]
        self.pending = None  # (group index, steps) whose gather has not been enqueued yet
        self.gathered = []   # groups gathered but not yet checked: (group index, steps)
        self.n_groups = 0
        self.failure = None  # root: (group index, failed queries per rank) of the first failing group
        self.failures = {}   # root: every failing group -> failed queries per rank
        # every rank: index of its own first group with a failed query (INT32_MAX: none), kept on the device; finish()
        # reduces it (MIN) over the ranks, so the verdict needs no word from the root and no host round trip of its own
        self.first_bad = torch.full((1,), 2**31 - 1, dtype=torch.int32, device=device)
        self.words = {}      # root: the failure words read so far (the last few groups)

    @staticmethod
    def from_index(index, dim, n, ef, shard, device, **kw):
        def step(q, ids_out, bits_out, stats_out, lane):
            index.search_batch_device(q.data_ptr(), q.shape[0], n, ef, ids_out.data_ptr(), bits_out.data_ptr(),
                                      0, stats_out.data_ptr(), lane.cuda_stream)
        return PipelinedShardedSearch(step, dim, n, shard, device, **kw)

    def _scatter(self, k, Q_root, g):
        b = k % self.depth
        L = self.lanes
        with L.on(L.comm):
            L.wait(L.comm, self.ev_r[b])  # the searches that last read q[b] are done
            dst = self.q[b][:g]
            if self.rank == self.root:
                # Q_root: [g, world * s, dim] -> per rank [g, s, dim]
                chunks = [c.contiguous() for c in Q_root.view(g, self.world, self.s, self.dim).unbind(1)]
                dist.scatter(dst, chunks, src=self.root, group=self.group)
            else:
                dist.scatter(dst, None, src=self.root, group=self.group)
            L.record(L.comm, self.ev_q[b])

    def _search(self, k, g):
        b = k % self.depth
        L = self.lanes
        L.wait(L.compute, self.ev_q[b])
        L.wait(L.compute, self.ev_g[b])  # the gather that last read res[b] is done
        with L.on(L.compute):
            for j in range(g):
                r = self.res[b][j]
                self.search_step(self.q[b][j], r[0], r[1], self.stats[b][j], L.compute)
        L.record(L.compute, self.ev_r[b])

    def _gather(self, k, g):
        b = k % self.depth
        L = self.lanes
        with L.on(L.comm):
            L.wait(L.comm, self.ev_r[b])
            # failed queries of the group, counted on the device next to the results -- on this lane: three
            # small kernels behind the last search of every group kept the compute lane idle for ~40 us
            bad = torch.count_nonzero(self.stats[b][:g, :, 3]).to(torch.int32)
            self.res[b][self.G].view(-1)[0:1].copy_(bad.view(1))
            self.first_bad.copy_(torch.minimum(self.first_bad, torch.where(bad.view(1) > 0, k, 2**31 - 1).to(torch.int32)))
            src = self.res[b]
            if self.rank == self.root:
                dist.gather(src, list(self.out[b].unbind(0)), dst=self.root, group=self.group)
                self.bad_host[b].copy_(self.out[b][:, self.G].reshape(self.world, -1)[:, 0], non_blocking=True)
                L.record(L.comm, self.ev_c[b])
            else:
                dist.gather(src, None, dst=self.root, group=self.group)
            L.record(L.comm, self.ev_g[b])
        self.gathered.append((k, g))

    def _word(self, k):
        """root: failed queries per rank of group k, read once through the pinned buffer of its slot (valid until
        `depth` more groups have been gathered, like the group's rows)"""
        if k in self.words:
            return self.words[k]
        slot = k % self.depth
        if self.ev_c[slot] is not None:
            self.ev_c[slot].synchronize()  # the word has landed; nothing else is waited for
        bad = [int(x) for x in self.bad_host[slot].tolist()]
        self.words[k] = bad
        self.words.pop(k - 4 * self.depth, None)
        if any(bad):
            self.failures[k] = bad
            if self.failure is None:
                self.failure = (k, bad)
        return bad

    def _check(self, keep_last=0):
        """root: read the word of every gathered group before its slot is reused.  Nothing is raised here: see
        the class comment."""
        while len(self.gathered) > keep_last:
            k, g = self.gathered.pop(0)
            if self.rank == self.root:
                self._word(k)

    def submit(self, Q_root, g):
        """Q_root: [g, world * shard, dim] on the root (None elsewhere), 1 <= g <= group_steps."""
        assert 1 <= g <= self.G
        k = self.n_groups
        self.n_groups += 1
        self._check(keep_last=1)  # the buffers of group k - depth are about to be reused
        self._scatter(k, Q_root, g)
        if self.pending is not None:
            self._gather(*self.pending)
        self._search(k, g)
        self.pending = (k, g)
        return k

    def finish(self):
        if self.pending is not None:
            self._gather(*self.pending)
            self.pending = None
        # The verdict reaches every rank: the first failing group of ANY rank, one all-reduce (MIN) enqueued on the
        # communication lane right behind the last gather -- no host round trip between the two (until round 4 the
        # root first drained both lanes, read the words and then broadcast a verdict: two synchronisations and a
        # collective one after the other, ~ 0.15 ms of a 20-step run's 4 ms)
        L = self.lanes
        with L.on(L.comm):
            verdict = self.first_bad.clone()
            dist.all_reduce(verdict, op=dist.ReduceOp.MIN, group=self.group)
        L.drain()
        self._check()
        first = int(verdict.cpu()[0])
        if first != 2**31 - 1:
            detail = (" %s queries per rank" % (self.failures[first],)) if first in self.failures else ""
            raise RuntimeError("group %d:%s did not finish with status 0 (visited table overflow or NaN); "
                               "re-run them through hnsw_search_batch" % (first, detail))

    def results(self, k, j, with_status=False):
        """(ids [W*s, n] int32, dists [W*s, n] f32) of step j of group k, on the root.  The group's failure word is
        read first: a group with failed queries raises FailedGroup instead of handing out rows that may be padding
        (the caller still calls finish(), which is collective); with_status=True returns (ids, dists, failed
        queries per rank) and never raises."""
        if self.pending is not None and self.pending[0] == k:
            raise ValueError("group %d has not been gathered yet: submit the next group or call finish() first" % k)
        if not (self.n_groups - self.depth <= k < self.n_groups) and k not in self.words:
            raise ValueError("group %d is no longer held (%d groups submitted, depth %d)" % (k, self.n_groups, self.depth))
        bad = self._word(k)
        o = self.out[k % self.depth][:, j]
        rows = (o[:, 0].reshape(-1, self.n), o[:, 1].reshape(-1, self.n).view(torch.float32))
        if with_status:
            return rows + (bad,)
        if any(bad):
            raise FailedGroup(k, bad)
        return rows


class FailedGroup(RuntimeError):
    """results() of a group in which queries did not finish with status 0; `.bad` = failed queries per rank"""

    def __init__(self, k, bad):
        super().__init__("group %d: %s queries per rank did not finish with status 0 (visited table overflow or NaN); "
                         "its rows are not handed out -- re-run them through hnsw_search_batch" % (k, bad))
        self.group, self.bad = k, bad
