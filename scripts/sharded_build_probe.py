"""two ranks on ONE GPU (gloo rendezvous, host-staged exchange): timing sanity of the sharded build.
run: python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 scripts/sharded_build_probe.py [N]"""
import os, sys, time
sys.path.insert(0, '.')
import numpy as np, torch, torch.distributed as dist
import hnsw_rs_amd as H
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
d, m = 100, 16
vs = H.synth_rows(0, 0x5EED0001, 0, N, d, 16)
idx = H.HNSW.new(m, 32, d)
dist.barrier(); t = time.time()
idx.insert_bulk_sharded(vs, 16, rank == 0)
dist.barrier(); dt = time.time() - t
if rank == 0:
    qs = H.synth_rows(0, 0x5EED0002, 0, 1024, d, 8)
    truth, _ = idx.brute_force(qs, 10)
    ids, _, _, _ = idx.search_batch(qs, 10, 64)
    print('sharded build, %d ranks on one GPU: %.2f s wall, recall@10 ef=64 %.4f' % (world, dt, sum(len(set(a) & set(b)) for a, b in zip(ids.tolist(), truth.tolist())) / 10240), flush=True)
dist.destroy_process_group()
