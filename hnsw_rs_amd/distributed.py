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
    def __init__(self, local_search, dim, n, device, group=None, root=0):
        """local_search(Q [s, dim] f32 tensor on `device`) -> (ids [s, n] int32, dists [s, n] f32)"""
        self.local_search = local_search
        self.dim, self.n, self.device, self.group, self.root = dim, n, device, group, root
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0

    def search(self, Q_root, nq):
        """Q_root: [nq, dim] on the root (ignored elsewhere).  Returns (ids, dists) of all nq
        queries on the root, (None, None) on the other ranks."""
        W, s = self.world, shard_size(nq, self.world)
        if W == 1:
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
    torch's current stream (no host synchronisation)."""
    ids = torch.empty((max_queries, n), dtype=torch.int32, device=device)
    dists = torch.empty((max_queries, n), dtype=torch.float32, device=device)
    counts = torch.empty(max_queries, dtype=torch.int32, device=device)
    stats = torch.empty((max_queries, 4), dtype=torch.int32, device=device)

    def local_search(Q):
        nq = Q.shape[0]
        assert nq <= max_queries and Q.is_contiguous() and Q.dtype == torch.float32
        index.search_batch_device(Q.data_ptr(), nq, n, ef, ids.data_ptr(), dists.data_ptr(),
                                  counts.data_ptr(), stats.data_ptr(),
                                  torch.cuda.current_stream().cuda_stream)
        return ids[:nq], dists[:nq]

    local_search.stats = stats
    local_search.counts = counts
    return local_search
