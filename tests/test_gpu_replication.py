"""Replication of the HBM snapshot ("index replicated per GPU, ncclBroadcast of the
flat buffers") and the RCCL legs of the multi-GPU paths, exercised on the ONE GPU of the test box:
  - in one process: describe -> adopt -> device copies -> commit, the replica answers like its source;
  - two ranks sharing the GPU (gloo rendezvous, host-staged broadcast): HNSW.replicate end to end;
  - one rank over the nccl backend (= RCCL): the broadcast of library-owned device memory, the sharded
    build's device-buffer all-gather callback and the pipelined sharded search run through RCCL itself
    (a one-rank communicator, but the same calls on the same device buffers as an 8-GPU run).
The CPU side (argument checks, no device) is in test_host_build.py."""
import ctypes as C
import os
import socket
import sys

import numpy as np
import pytest

import hnsw_rs_amd as H
from hnsw_rs_amd import _lib
from tests.util import assert_search_equal

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


@pytest.mark.parametrize("kind", [H.VEC_QUANT8, H.VEC_F32])
def test_replica_made_by_device_copies_answers_like_its_source(kind):
    import torch
    n, d, m = 20000, 100, 16
    vs = H.synth_rows(0, 0x5EED0001, 0, n, d)
    qs = H.synth_rows(0, 0x5EED0002, 0, 256, d)
    src = H.HNSW.new(m, 32, d, kind).insert_bulk_device(vs, 4, False, levels=H.draw_levels(m, n))
    L = _lib.lib()
    desc = _lib.SnapshotDesc()
    _lib.check(L.hnsw_snapshot_describe(src._h, C.byref(desc)))
    assert desc.bytes[0] >= n * d and desc.bytes[1] == n * 32 * 4 and desc.ptr[0]
    rep = H.HNSW.new(m, 32, d, kind)
    there = _lib.SnapshotDesc()
    for i in range(7):
        there.bytes[i] = desc.bytes[i]
    for i in range(32):
        there.header[i] = desc.header[i]
    # before the arrays are there the replica refuses to search
    _lib.check(L.hnsw_snapshot_adopt(rep._h, C.byref(there)))
    with pytest.raises(H.HnswError):
        rep.search_batch(qs, 10, 64)

    class Mem:
        def __init__(self, ptr, nbytes):
            self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 2}

    for i in range(7):
        if desc.bytes[i]:
            a = torch.as_tensor(Mem(int(desc.ptr[i]), int(desc.bytes[i])), device="cuda:0")
            b = torch.as_tensor(Mem(int(there.ptr[i]), int(there.bytes[i])), device="cuda:0")
            b.copy_(a)
    torch.cuda.synchronize()
    _lib.check(L.hnsw_snapshot_commit(rep._h))
    assert rep.len() == n and rep.nb_layers() == src.nb_layers() and int(rep.params.ep) == int(src.params.ep)
    for ef in (10, 64, 100):
        assert_search_equal(rep.search_batch(qs, 10, ef), src.search_batch(qs, 10, ef), "replica ef=%d" % ef)
    bf_r, bf_s = rep.brute_force(qs[:16], 10), src.brute_force(qs[:16], 10)
    assert np.array_equal(bf_r[0], bf_s[0])
    # a device-only replica holds no host copy: mutation, persistence and per-point accessors refuse
    for call in (lambda: rep.insert_vec(vs[0]), lambda: rep.insert_bulk(vs[:4], 1, False),
                 lambda: rep.save("/tmp/never_written"), lambda: rep.get_point(0).get_vals(), lambda: rep.clone()):
        with pytest.raises(H.HnswError):
            call()
    # wrong shape of receiver / sizes that do not match the header are refused before anything is allocated
    other = H.HNSW.new(m, 32, d + 4, kind)
    with pytest.raises(H.HnswError):
        _lib.check(L.hnsw_snapshot_adopt(other._h, C.byref(there)))
    bad = _lib.SnapshotDesc()
    for i in range(7):
        bad.bytes[i] = desc.bytes[i]
    for i in range(32):
        bad.header[i] = desc.header[i]
    bad.bytes[1] -= 4
    fresh = H.HNSW.new(m, 32, d, kind)  # (kept in a name: a temporary would be freed under the call)
    with pytest.raises(H.HnswError):
        _lib.check(L.hnsw_snapshot_adopt(fresh._h, C.byref(bad)))


def _replicate_worker(rank, world, port, outdir, backend):
    import torch
    import torch.distributed as dist
    from tests.conftest import ROOT
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group(backend, rank=rank, world_size=world)
    import hnsw_rs_amd as HH
    n, d, m = 12000, 128, 16
    qs = HH.synth_rows(0, 0x5EED0002, 0, 128, d)
    idx = None
    if rank == 0:
        vs = HH.synth_rows(0, 0x5EED0001, 0, n, d)
        idx = HH.HNSW.new(m, 32, d, HH.VEC_F32).insert_bulk_device(vs, 4, False, levels=HH.draw_levels(m, n))
    mine = HH.HNSW.replicate(idx, m, 32, d, HH.VEC_F32, src=0, device="cuda:0", chunk_bytes=1 << 20)
    ids, dists, counts, stats = mine.search_batch(qs, 10, 64)
    np.savez(os.path.join(outdir, "rank%d.npz" % rank), ids=ids, dists=dists, counts=counts, stats=np.asarray(stats),
             n=np.array([mine.len(), mine.nb_layers(), mine.device_bytes()]))
    dist.barrier()
    dist.destroy_process_group()


def test_replicate_two_ranks_sharing_the_gpu(tmp_path):
    """rank 0 builds, rank 1 receives the flat arrays (gloo: staged through the host, in 1-MiB pieces)
    and answers identically"""
    import torch.multiprocessing as mp
    mp.spawn(_replicate_worker, args=(2, _free_port(), str(tmp_path), "gloo"), nprocs=2, join=True)
    r0, r1 = np.load(tmp_path / "rank0.npz"), np.load(tmp_path / "rank1.npz")
    for k in r0.files:
        assert np.array_equal(r0[k], r1[k]), k
    assert (r0["stats"][:, 3] == 0).all()


def _rccl_worker(rank, world, port, outdir):
    import torch
    import torch.distributed as dist
    from tests.conftest import ROOT
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=rank, world_size=world)
    import hnsw_rs_amd as HH
    from hnsw_rs_amd.distributed import PipelinedShardedSearch
    n, d, m, B, k, ef = 20000, 100, 16, 256, 10, 64
    vs = HH.synth_rows(0, 0x5EED0001, 0, n, d)
    lv = HH.draw_levels(m, n)
    dev = torch.device("cuda:0")
    # 1. the sharded build: its per-batch exchange is dist.all_gather_into_tensor on device buffers
    idx = HH.HNSW.new(m, 32, d).insert_bulk_sharded(vs, 4, False, levels=lv, device="cuda:0")
    one = HH.HNSW.new(m, 32, d).insert_bulk_device(vs, 4, False, levels=lv)
    same = all(all(np.array_equal(np.sort(x), np.sort(y)) if i == 2 else np.array_equal(x, y)
                   for i, (x, y) in enumerate(zip(a.csr(), b.csr()))) for a, b in zip(idx.iter_layers(), one.iter_layers()))
    # 2. replication: broadcast of library-owned device memory through RCCL
    rep = HH.HNSW.replicate(idx, m, 32, d, HH.VEC_QUANT8, src=0, device="cuda:0")
    # 3. the bench's N > 1 path: scatter / gather of device buffers on a communication stream
    steps, G = 5, 2
    Q = torch.from_numpy(HH.synth_rows(0, 0x5EED0002, 0, steps * B, d)).view(steps, B, d).to(dev)
    pipe = PipelinedShardedSearch.from_index(idx, d, k, ef, B, dev, group_steps=G, depth=2)
    got = {}
    i = 0
    while i < steps:
        g = min(G, steps - i)
        kk = pipe.submit(Q[i:i + g].contiguous(), g)
        if kk >= 1:
            first = (kk - 1) * G
            pipe.lanes.drain()
            for j in range(min(G, steps - first)):
                a, b = pipe.results(kk - 1, j)
                got[first + j] = (a.cpu().numpy().copy(), b.cpu().numpy().copy())
        i += g
    pipe.finish()
    last = (steps - 1) // G
    for j in range(steps - last * G):
        a, b = pipe.results(last, j)
        got[last * G + j] = (a.cpu().numpy().copy(), b.cpu().numpy().copy())
    ok = True
    for s in range(steps):
        w_ids, w_d, _, _ = idx.search_batch(Q[s].cpu().numpy(), k, ef)
        ok &= np.array_equal(got[s][0].astype(np.uint32), w_ids) and np.array_equal(got[s][1], w_d)
    np.savez(os.path.join(outdir, "rccl.npz"), same=np.array([int(same)]), pipe_ok=np.array([int(ok)]),
             rep_is_src=np.array([int(rep is idx)]))
    dist.barrier()
    dist.destroy_process_group()


def test_rccl_legs_on_a_one_rank_communicator(tmp_path):
    """backend nccl IS RCCL on ROCm: the all-gather callback of the sharded build, the snapshot broadcast and
    the pipelined scatter / gather run through it on device buffers (a communicator of one rank: the same
    calls, no peer)"""
    import torch.multiprocessing as mp
    mp.spawn(_rccl_worker, args=(1, _free_port(), str(tmp_path)), nprocs=1, join=True)
    r = np.load(tmp_path / "rccl.npz")
    assert r["same"][0] == 1, "sharded build over nccl differs from the single-GPU device build"
    assert r["pipe_ok"][0] == 1, "pipelined sharded search over nccl differs from a direct search"
    assert r["rep_is_src"][0] == 1
