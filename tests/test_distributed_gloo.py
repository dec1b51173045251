"""N > 1 path on CPU: world_size-2 gloo run of the query sharding plumbing
(hnsw_rs_amd/distributed.py).  The local search is the CPU oracle here (tests may use it); on the
GPU box the same ShardedSearcher wraps the HIP search and runs over RCCL."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tests.conftest import ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, nq, outfile):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from hnsw_rs_amd.distributed import ShardedSearcher, shard_bounds
    from oracle import oracle_py as O
    from tests.util import rand_vectors
    n, d, m, k, ef = 800, 12, 8, 5, 20
    vs = rand_vectors(n, d, 1)
    orc = O.OracleHNSW(m, None, d).insert_bulk(vs, O.draw_levels(n, m, 1))  # every rank: same replica

    def local_search(Q):
        ids, dists, _, _ = orc.search_batch(Q.numpy(), k, ef)
        return torch.from_numpy(ids.astype(np.int64)).to(torch.int32), torch.from_numpy(dists)

    s = ShardedSearcher(local_search, d, k, torch.device("cpu"))
    Q = torch.from_numpy(rand_vectors(nq, d, 2)) if rank == 0 else None
    ids, dists = s.search(Q, nq)
    lo, hi = shard_bounds(nq, world, rank)
    assert 0 <= lo <= hi <= nq
    if rank == 0:
        w_ids, w_d, _, _ = orc.search_batch(Q.numpy(), k, ef)
        ok = np.array_equal(ids.numpy().astype(np.uint32), w_ids) and np.array_equal(dists.numpy(), w_d)
        open(outfile, "w").write("ok" if ok else "mismatch")
    else:
        assert ids is None and dists is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("nq", [64, 37])  # 37: uneven split, the last shard is padded
def test_query_sharding_world_size_2_gloo(tmp_path, nq):
    out = str(tmp_path / "result.txt")
    mp.spawn(_worker, args=(2, _free_port(), nq, out), nprocs=2, join=True)
    assert open(out).read() == "ok"


def test_shard_bounds_cover_the_batch():
    from hnsw_rs_amd.distributed import shard_bounds, shard_size
    for nq in (1, 7, 1024, 1025):
        for w in (1, 2, 3, 8):
            cover = []
            for r in range(w):
                lo, hi = shard_bounds(nq, w, r)
                cover += list(range(lo, hi))
                assert hi - lo <= shard_size(nq, w)
            assert cover == list(range(nq))


def _pipe_worker(rank, world, port, outfile, inject_failure):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from hnsw_rs_amd.distributed import InlineLanes, PipelinedShardedSearch
    from oracle import oracle_py as O
    from tests.util import rand_vectors
    n, d, m, k, ef, shard, G = 600, 10, 8, 5, 16, 7, 2
    vs = rand_vectors(n, d, 1)
    orc = O.OracleHNSW(m, None, d).insert_bulk(vs, O.draw_levels(n, m, 1))

    calls = {"n": 0}

    def search_step(q, ids_out, bits_out, stats_out, lane):
        ids, dists, _, st = orc.search_batch(q.numpy(), k, ef)
        ids_out.copy_(torch.from_numpy(ids.astype(np.int64)).to(torch.int32))
        bits_out.copy_(torch.from_numpy(dists.view(np.int32).copy()))
        stats_out.zero_()
        stats_out[:, :3] = torch.from_numpy(st.astype(np.int64)).to(torch.int32)
        calls["n"] += 1
        if inject_failure and rank == 1 and calls["n"] == 1:
            # one query of the FIRST group on the non-root rank reports an overflow: the root learns it with
            # three more groups of collectives still to come, and must not leave the sequence
            stats_out[2, 3] = -7

    pipe = PipelinedShardedSearch(search_step, d, k, shard, torch.device("cpu"), group_steps=G, depth=2,
                                  lanes=InlineLanes())
    steps = 7  # groups of 2, 2, 2 and 1 steps: the last group is a partial one
    Q = torch.from_numpy(rand_vectors(steps * world * shard, d, 2)).view(steps, world * shard, d)
    from hnsw_rs_amd.distributed import FailedGroup
    got, failed, refused = {}, "", []
    try:
        i = 0
        while i < steps:
            g = min(G, steps - i)
            kk = pipe.submit(Q[i:i + g].contiguous() if rank == 0 else None, g)
            if rank == 0 and kk >= 1:
                # a root that consumes results BETWEEN submits: the group before is gathered by now.  The rows of a
                # failed group are refused (never padding ids without a signal), the other groups' rows are handed
                # out, and the root stays in the sequence of collectives either way
                try:
                    ids, dists = pipe.results(kk - 1, 0)
                    w_ids, w_d, _, _ = orc.search_batch(Q[i - G].numpy(), k, ef)
                    assert np.array_equal(ids.numpy().astype(np.uint32), w_ids) and np.array_equal(dists.numpy(), w_d)
                except FailedGroup as e:
                    refused.append((e.group, e.bad))
                    assert pipe.results(kk - 1, 0, with_status=True)[2] == e.bad
            i += g
        pipe.finish()
    except RuntimeError as e:
        failed = str(e)
    if rank == 0:
        ok = True
        if not inject_failure:
            # the last `depth` groups are still held: check them against a direct search
            for kk, first, g in ((2, 4, 2), (3, 6, 1)):
                for j in range(g):
                    ids, dists = pipe.results(kk, j)
                    w_ids, w_d, _, _ = orc.search_batch(Q[first + j].numpy(), k, ef)
                    ok &= np.array_equal(ids.numpy().astype(np.uint32), w_ids) and np.array_equal(dists.numpy(), w_d)
            ok &= failed == "" and refused == [] and pipe.failures == {}
        else:
            ok = "group 0:" in failed and "did not finish with status 0" in failed and pipe.failure == (0, [0, 1])
            ok &= refused == [(0, [0, 1])] and pipe.failures == {0: [0, 1]}
        open(outfile, "w").write("ok" if ok else "mismatch: " + failed)
    else:
        # the failure is collective: the other rank issued every collective of the sequence (or this
        # process would hang in the gloo timeout) and raised too
        open(outfile + ".rank%d" % rank, "w").write(failed)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("inject_failure", [False, True])
def test_pipelined_bucketed_exchange_world_size_2_gloo(tmp_path, inject_failure):
    """the N > 1 bench path (groups of steps, one scatter and one gather per group, a partial last group)
    with the stream pair replaced by an inline executor; a failed query on a non-root rank must surface"""
    out = str(tmp_path / "result.txt")
    mp.spawn(_pipe_worker, args=(2, _free_port(), out, inject_failure), nprocs=2, join=True)
    assert open(out).read() == "ok"
    other = open(out + ".rank1").read()
    assert ("group 0:" in other and "did not finish with status 0" in other) if inject_failure else other == ""
