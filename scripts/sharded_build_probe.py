"""What the sharded build's ranks do, measured with W ranks that SHARE the one GPU of a gpurun box (gloo rendezvous,
the exchange staged through the host): per rank the insert kernel's and the connect + drop kernels' time (HIP events),
the rows it changed as an owner, the rows it received, the bytes of the variable-size exchanges -- once with phases
2 / 3 split by row ownership (the default) and once with every rank running them in full (HNSW_MI355X_SHARD_CONNECT=0).
The ranks run in lockstep on one device, so kernel times include each other's contention and the wall clock means
nothing; the COUNTS (rows, records, bytes) are what an 8-GPU run would see per rank.

    python scripts/sharded_build_probe.py [--n 1000000] [--dim 128] [--world 2]
"""
import argparse
import json
import os
import socket
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def worker(rank, world, port, n, d, m, outdir):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import hnsw_rs_amd as H
    vs = H.synth_rows(0, 0x5EED0001, 0, n, d)
    idx = H.HNSW.new(m, 32, d)
    idx.set_option("gpu_build_batch_max", 32768)
    idx.set_option("gpu_build_batch_div", 8)
    dist.barrier()
    t0 = time.time()
    idx.insert_bulk_sharded(vs, 4, False, levels=H.draw_levels(m, n))
    wall = time.time() - t0
    keys = ["points", "batches", "records", "removals", "rows_owned", "rows_received", "exchange_bytes", "exchange_us",
            "insert_kernel_us", "connect_kernel_us", "insert_phase_us", "connect_us"]
    st = {k: idx.stat("build_" + k) for k in keys}
    st["wall_s"] = round(wall, 2)
    edges = 0
    for layer in idx.iter_layers():
        _, offs, _ = layer.csr()
        edges += int(offs[-1])
    st["edges"] = edges
    json.dump(st, open(os.path.join(outdir, "rank%d.json" % rank), "w"))
    dist.barrier()
    dist.destroy_process_group()


def main():
    import tempfile
    import torch.multiprocessing as mp
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=1000000)
    ap.add_argument("--dim", type=int, default=128)
    ap.add_argument("--m", type=int, default=16)
    ap.add_argument("--world", type=int, default=2)
    a = ap.parse_args()
    for mode in ("1", "0"):
        os.environ["HNSW_MI355X_SHARD_CONNECT"] = mode
        out = tempfile.mkdtemp()
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        mp.spawn(worker, args=(a.world, port, a.n, a.dim, a.m, out), nprocs=a.world, join=True)
        print("== %d x %dd, m %d, %d ranks on one GPU, phases 2 / 3 %s ==" % (
            a.n, a.dim, a.m, a.world, "by row ownership" if mode == "1" else "in full on every rank"), flush=True)
        for r in range(a.world):
            st = json.load(open(os.path.join(out, "rank%d.json" % r)))
            print("rank %d: insert kernel %.2f s, connect + drop kernels %.3f s, rows owned %d, received %d, exchanges %.1f MB "
                  "in %.2f s (host-staged gloo), records %d, removals %d, edges %d, wall %.1f s" % (
                      r, st["insert_kernel_us"] / 1e6, st["connect_kernel_us"] / 1e6, st["rows_owned"], st["rows_received"],
                      st["exchange_bytes"] / 1e6, st["exchange_us"] / 1e6, st["records"], st["removals"], st["edges"], st["wall_s"]),
                  flush=True)


if __name__ == "__main__":
    main()
