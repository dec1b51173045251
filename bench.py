#!/usr/bin/env python3
"""bench.py -- queries/sec of the HNSW search hot path on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workload (BASELINE.json configs[1]: "1M x 100d fp32 L2, M=16 efSearch=64, batch=1024 queries"):
N = 1M x 100d synthetic GloVe-shaped vectors (recipe A of SURVEY.md section 8d), M = 16,
ef_construction = 32, n = 10; one step = one batch of 1024 queries per GPU answered by
`hnsw_search_batch_device` with queries and result buffers already resident in HBM.  efSearch is
the configured 64 when that reaches the metric's recall@10 >= 0.99 on this data, otherwise the
first larger value that does (the configured-64 rate is reported next to it).

Vector kind: the config says fp32, so the timed index stores f32 rows (HNSW_VEC_F32 = the
reference's `VecType = FullVec`, vectors/src/full.rs).  The reference SHIPS `VecType = QuantVec`
(8-bit codes dequantised to f32 on the fly, points/src/point.rs:4); that variant is measured in
the same run at N = 1 and reported under "quant8_reference_default".  All arithmetic is f32.

Index build is outside the timed region.  N > 1: one process per GPU, index replicated in every
GPU's HBM, each group of steps' 1024 x N queries scattered from rank 0 and the results gathered
back over RCCL (weak scaling), exchange and search pipelined on two streams.

One JSON line on stdout (rank 0); progress on stderr.
"""
import argparse
import json
import os
import shutil
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured streaming)
EF_LADDER = (64, 68, 72, 76, 80, 88, 96, 112, 128, 160, 192, 256)


def log(*a):
    print("[bench]", *a, file=sys.stderr, flush=True)


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=200)
    p.add_argument("--warmup", type=int, default=20)
    p.add_argument("--n-points", type=int, default=1_000_000)
    p.add_argument("--dim", type=int, default=100)
    p.add_argument("--m", type=int, default=16)
    p.add_argument("--ef-cons", type=int, default=32)
    p.add_argument("--ef", default="auto",
                   help="efSearch; 'auto' = the configured 64 if it reaches the metric's recall@10 >= 0.99 "
                        "on this data, else the first of %s that does" % (EF_LADDER,))
    p.add_argument("--min-recall", type=float, default=0.99)
    p.add_argument("--topn", type=int, default=10)
    p.add_argument("--batch", type=int, default=1024)
    p.add_argument("--kind", choices=["f32", "quant8"], default="f32")
    p.add_argument("--no-secondary", action="store_true", help="skip the other vector kind")
    p.add_argument("--no-concurrent", action="store_true",
                   help="skip the secondary several-batches-in-flight figure (profiling passes: keeps the kernel trace to the timed launches)")
    p.add_argument("--recipe", type=int, default=0, help="0 = A (low intrinsic dim), 1 = B (isotropic)")
    p.add_argument("--query-batches", type=int, default=10, help="distinct batches cycled through")
    p.add_argument("--build-threads", type=int, default=0)
    p.add_argument("--cpu-build", action="store_true",
                   help="build the index with the host threads only (default: on-device build, "
                        "hnsw_insert_bulk_device)")
    p.add_argument("--cpu-threads", type=int, default=0)
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--recall-queries", type=int, default=10240,
                   help="queries the true recall is measured on (all 10 bench batches: standard error 0.0003)")
    p.add_argument("--index-cache", default=os.environ.get("HNSW_BENCH_CACHE", "/tmp/hnsw_bench_cache"))
    return p.parse_args()


class Ctx:
    pass


def get_index(c, kind_name):
    """rank 0 builds (or loads a cached build) and saves; the other ranks load the same files."""
    import hnsw_rs_amd as H
    a = c.args
    kind = H.VEC_QUANT8 if kind_name == "quant8" else H.VEC_F32
    tag = "n%d_d%d_m%d_efc%d_%s_r%d" % (a.n_points, a.dim, a.m, a.ef_cons, kind_name, a.recipe)
    cache_dir = os.path.join(a.index_cache, tag)
    index = None
    t0 = time.time()
    if c.rank == 0:
        if os.path.isdir(cache_dir):
            try:
                index = H.HNSW.load(cache_dir)
                log("loaded cached %s index (%.1fs)" % (kind_name, time.time() - t0))
            except H.HnswError as e:
                log("cache unusable (%s); rebuilding" % e)
                shutil.rmtree(cache_dir, ignore_errors=True)
        if index is None:
            store = H.synth_rows(a.recipe, 0x5EED0001, 0, a.n_points, a.dim, min(32, c.ncpu))
            t1 = time.time()
            index = H.HNSW.new(a.m, a.ef_cons, a.dim, kind)
            if a.cpu_build:
                index.insert_bulk(store, c.build_threads, False)
            else:
                index.set_device(c.local_rank)
                index.insert_bulk_device(store, c.build_threads, False)
            log("built the %s index (%s, %d host threads) in %.1fs, %d layers" % (
                kind_name, "host build" if a.cpu_build else "on-device build", c.build_threads,
                time.time() - t1, index.nb_layers()))
            del store
            try:
                os.makedirs(a.index_cache, exist_ok=True)
                shutil.rmtree(cache_dir, ignore_errors=True)
                index.save(cache_dir)
            except (H.HnswError, OSError) as e:
                log("could not cache the index: %s" % e)
    if c.world > 1:
        c.dist.barrier()
        if c.rank != 0:
            index = H.HNSW.load(cache_dir)  # replicate: every rank holds the same index
    index.set_device(c.local_rank)
    index.upload()
    log("rank %d: %s index resident in HBM, %.1f MB" % (c.rank, kind_name, index.device_bytes() / 1e6))
    return index, tag


def choose_ef(c, index, kind_name):
    """true recall@n against exhaustive search under the index's own metric (the reference's own
    ground truth, hnsw/src/template.rs:531-541), on the first --recall-queries bench queries"""
    a, n = c.args, c.args.topn
    recall_by_ef = {}
    nr = min(a.recall_queries, c.queries.shape[0])
    qh = c.queries[:nr]
    bf, _ = index.brute_force(qh, n)

    def recall_at(e):
        got, _, _, _ = index.search_batch(qh, n, e)
        hits = sum(len(set(x.tolist()) & set(y.tolist())) for x, y in zip(got, bf))
        return hits / float(nr * n)

    if a.ef == "auto":
        ef = None
        for e in EF_LADDER:
            recall_by_ef[e] = round(recall_at(e), 5)
            log("%s efSearch %d: true recall@%d = %.4f (%d queries)" % (kind_name, e, n, recall_by_ef[e], nr))
            if recall_by_ef[e] >= a.min_recall:
                ef = e
                break
        if ef is None:
            ef = EF_LADDER[-1]
    else:
        ef = int(a.ef)
        recall_by_ef[ef] = round(recall_at(ef), 5)
    if 64 not in recall_by_ef:
        recall_by_ef[64] = round(recall_at(64), 5)
    return ef, recall_by_ef


def time_local(c, local_search, steps, warmup):
    """K back-to-back launches on the current stream, one HIP event pair per launch"""
    torch = c.torch
    B, nqb = c.args.batch, c.args.query_batches
    qs = [c.dQ[b][:B].contiguous() for b in range(nqb)]
    for i in range(warmup):
        local_search(qs[i % nqb])
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    t0 = time.perf_counter()
    for i in range(steps):
        ev[i][0].record()
        local_search(qs[(warmup + i) % nqb])
        ev[i][1].record()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    return elapsed, float(np.mean([x.elapsed_time(y) for x, y in ev]))


def analyse(c, index, kind_name, tag, ef, recall_by_ef, local_search, kern_ms):
    """roofline, parity and CPU baseline for one vector kind (rank 0)"""
    import hnsw_rs_amd as H
    a, torch = c.args, c.torch
    d, n, B, nqb, N, m = a.dim, a.topn, a.batch, a.query_batches, a.n_points, a.m
    kind = H.VEC_QUANT8 if kind_name == "quant8" else H.VEC_F32
    out = {}
    stats_all, ids_all = [], []
    for b in range(nqb):
        ids_b, _ = local_search(c.dQ[b][:B].contiguous())
        torch.cuda.synchronize()
        stats_all.append(local_search.stats[:B].cpu().numpy().copy())
        ids_all.append(ids_b.cpu().numpy().copy().view(np.uint32))
    st = np.concatenate(stats_all).astype(np.int64)
    ids_gpu = np.concatenate(ids_all)
    if (st[:, 3] != 0).any():
        sys.exit("search reported per-query errors: %s" % np.unique(st[:, 3]))
    row_bytes = (d + 8) if kind == H.VEC_QUANT8 else 4 * d
    # B_q = n_dist*row_bytes + n_exp*(4 + 4*deg) + 4*d + 8*n
    bq = st[:, 0] * row_bytes + st[:, 1] * 4 + st[:, 2] * 4 + 4 * d + 8 * n
    bytes_per_launch = float(bq.mean() * B)
    achieved = bytes_per_launch / (kern_ms * 1e-3) / 1e9
    traffic = None
    tfile = os.path.join(ROOT, "profiles", "traffic_latest.json")
    if os.path.exists(tfile):
        try:
            for tj in json.load(open(tfile)).get("entries", []):
                if tj.get("workload") == tag and tj.get("ef") == ef and tj.get("batch") == B:
                    traffic = tj.get("hbm_bytes_per_launch")
        except (OSError, ValueError, AttributeError):
            pass
    out["roofline"] = {
        "bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
        "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
        "kernel": "hx_search_kernel", "kernel_ms": round(kern_ms, 5),
        "algorithmic_bytes_per_launch": round(bytes_per_launch),
        "per_query": {"n_dist": round(float(st[:, 0].mean()), 2), "n_exp": round(float(st[:, 1].mean()), 2),
                      "sum_deg": round(float(st[:, 2].mean()), 2), "bytes": round(float(bq.mean()), 1),
                      "row_bytes": row_bytes},
    }
    out["recall_at_%d" % n] = recall_by_ef[ef]
    out["recall_by_ef"] = {str(k): v for k, v in sorted(recall_by_ef.items())}
    if ef != 64:  # the configured efSearch = 64 timed too when the metric's recall needed a larger ef
        from hnsw_rs_amd.distributed import make_device_search
        ls64 = make_device_search(index, n, 64, B, c.dev)
        _, ms64 = time_local(c, ls64, 50, 5)
        out["at_configured_efSearch_64"] = {"queries_per_s_per_gpu": round(B / ms64 * 1e3, 1),
                                            "ms_per_step": round(ms64, 5),
                                            "recall_at_%d" % n: recall_by_ef[64]}
    # Not the metric: the same 1024-query launches with several batches in flight (one stream each).
    # A 1024-query launch puts one wave on every SIMD; independent batches share the SIMDs and hide each
    # other's memory waits, which is what a server with concurrent requests sees.
    try:
        if a.no_concurrent:
            raise RuntimeError("--no-concurrent")
        from hnsw_rs_amd.distributed import make_device_search
        S_CONC = 4
        streams = [torch.cuda.Stream(device=c.dev) for _ in range(S_CONC)]
        searchers = [make_device_search(index, n, ef, B, c.dev) for _ in range(S_CONC)]
        qs = [c.dQ[b][:B].contiguous() for b in range(nqb)]
        steps = 80

        def burst(k):
            for i in range(k):
                with torch.cuda.stream(streams[i % S_CONC]):
                    searchers[i % S_CONC](qs[i % nqb])
        burst(8)
        torch.cuda.synchronize()
        tc = time.perf_counter()
        burst(steps)
        torch.cuda.synchronize()
        wall = time.perf_counter() - tc
        out["concurrent_batches"] = {
            "streams": S_CONC, "batch": B, "efSearch": ef, "queries_per_s_per_gpu": round(B * steps / wall, 1),
            "hbm_frac_algorithmic": round(bytes_per_launch * steps / wall / 1e9 / HBM_PEAK_GBS, 5),
            "note": "secondary; the metric's value times one batch at a time"}
    except Exception as e:  # a secondary figure must never cost the bench line
        log("concurrent-batches measurement skipped: %s" % e)
    if not a.no_cpu_baseline:
        from oracle import oracle_py as O
        t2 = time.time()
        orc = O.OracleHNSW(m, a.ef_cons, d, kind)
        store = H.synth_rows(a.recipe, 0x5EED0001, 0, N, d, min(32, c.ncpu))
        lv = np.zeros(N, dtype=np.uint8)
        for l in range(1, index.nb_layers()):
            lv[index.get_layer(l).iter_nodes()] = l
        orc.import_points(store, lv)
        del store
        for l in range(index.nb_layers()):
            orc.import_layer(l, *index.get_layer(l).csr())
        orc.set_ep(int(index.params.ep))
        log("oracle holds the same %s index (%.1fs)" % (kind_name, time.time() - t2))
        T = a.cpu_threads or max(1, min(16, c.ncpu))
        qcpu = c.queries.reshape(nqb, B * c.world, d)[:, :B].reshape(-1, d)
        t3 = time.time()
        reps = 0
        while True:  # about 10-30 s of CPU work in total
            o_ids, _, _, o_st = orc.search_batch(qcpu, n, ef, nthreads=T)
            reps += 1
            if time.time() - t3 > 1.2 or reps >= 8:
                break
        cpu_s = time.time() - t3
        t4 = time.time()
        orc.search_batch(qcpu[:2048], n, ef, nthreads=1)
        one_s = time.time() - t4
        same = float((o_ids == ids_gpu).all(axis=1).mean())
        out["cpu_baseline"] = {
            "value": round(reps * qcpu.shape[0] / cpu_s, 1), "unit": "queries/s", "cores": T, "kind": "port",
            "sample": "%d queries (the %d bench batches) x %d passes on %d threads = %.1f s wall; "
                      "1 thread: %.0f queries/s on 2048 queries" % (qcpu.shape[0], nqb, reps, T, cpu_s, 2048 / one_s),
            "single_thread_value": round(2048 / one_s, 1),
            "host_cpu": next((l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo")
                              if l.startswith("model name")), "unknown"),
        }
        out["parity"] = {"queries": int(qcpu.shape[0]), "ids_identical_fraction": same,
                         "counters_identical": bool(np.array_equal(o_st.astype(np.int64), st[:, :3]))}
        log("%s: cpu oracle %.0f q/s on %d threads; GPU ids identical for %.4f of queries" % (
            kind_name, reps * qcpu.shape[0] / cpu_s, T, same))
        del orc
    return out


def main():
    args = parse()
    # Exactly ONE line on stdout: libraries (RCCL prints a version banner) write to fd 1 too, so the
    # real stdout is set aside for the JSON line and fd 1 points to stderr for everything else.
    sys.stdout.flush()
    json_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    import torch
    import torch.distributed as dist

    import hnsw_rs_amd as H
    from hnsw_rs_amd.distributed import PipelinedShardedSearch, make_device_search

    c = Ctx()
    c.args, c.torch, c.dist = args, torch, dist
    c.world = int(os.environ.get("WORLD_SIZE", "1"))
    c.rank = int(os.environ.get("RANK", "0"))
    c.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world, rank = c.world, c.rank
    if world != args.gpus:
        log("warning: --gpus %d but WORLD_SIZE %d; using WORLD_SIZE" % (args.gpus, world))
    if not torch.cuda.is_available():
        sys.exit("bench.py needs an MI355X: torch.cuda.is_available() is False")
    torch.cuda.set_device(c.local_rank)
    c.dev = dev = torch.device("cuda", c.local_rank)
    # HNSW_BENCH_FORCE_DIST=1 runs the scatter / gather path even with one rank (a smoke test of the
    # multi-GPU code on a single-GPU box)
    force_dist = os.environ.get("HNSW_BENCH_FORCE_DIST") == "1"
    use_dist = world > 1 or force_dist
    if use_dist:
        if "MASTER_ADDR" not in os.environ:
            os.environ["MASTER_ADDR"] = "127.0.0.1"
            os.environ["MASTER_PORT"] = os.environ.get("MASTER_PORT", "29533")
        dist.init_process_group("nccl", device_id=dev, rank=rank, world_size=world)

    N, d, m, n, B = args.n_points, args.dim, args.m, args.topn, args.batch
    c.ncpu = os.cpu_count() or 8
    c.build_threads = args.build_threads or max(1, min(32, c.ncpu // max(1, world)))
    nqb = args.query_batches
    c.queries = H.synth_rows(args.recipe, 0x5EED0002, 0, nqb * B * world, d, min(16, c.ncpu))
    c.dQ = torch.from_numpy(c.queries).to(dev).view(nqb, B * world, d) if rank == 0 else None

    # ---- the timed kind ---------------------------------------------------------------------------
    index, tag = get_index(c, args.kind)
    if rank == 0:
        ef, recall_by_ef = choose_ef(c, index, args.kind)
    else:
        ef, recall_by_ef = 0, {}
    if use_dist:
        t = torch.tensor([ef], dtype=torch.int64, device=dev)
        dist.broadcast(t, src=0)
        ef = int(t.item())
    local_search = make_device_search(index, n, ef, B, dev)
    G = 8  # steps per exchange group (bucketed collectives)
    pipe = PipelinedShardedSearch.from_index(index, d, n, ef, B, dev, group_steps=G) if use_dist else None
    K, W = args.steps, args.warmup

    def run_dist(first, count):
        i = first
        while i < first + count:
            g = min(G, first + count - i)
            qg = torch.stack([c.dQ[(i + j) % nqb] for j in range(g)], 0) if rank == 0 else None
            pipe.submit(qg, g)  # scatter / searches / gather of neighbouring groups overlap
            i += g

    if pipe is None:
        elapsed, kern_ms = time_local(c, local_search, K, W)
    else:
        run_dist(0, W)
        pipe.finish()
        if rank == 0 and W > 0:  # the exchange returns what a local search returns
            last = W - 1
            chk_ids, chk_d = pipe.results(pipe.n_groups - 1, last % G)
            ref_ids, ref_d = local_search(c.dQ[last % nqb][:B].contiguous())
            torch.cuda.synchronize()
            assert torch.equal(chk_ids[:B], ref_ids) and torch.equal(chk_d[:B], ref_d), "gathered != local"
        torch.cuda.synchronize()
        dist.barrier()
        t_start = time.perf_counter()
        run_dist(W, K)  # exactly K steps
        pipe.finish()
        torch.cuda.synchronize()
        dist.barrier()
        elapsed = time.perf_counter() - t_start
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        kern_ms = None
        if rank == 0:
            _, kern_ms = time_local(c, local_search, 20, 2)  # the search launch alone, this rank's slice
    qps = K * B * world / elapsed

    if rank == 0:
        result = analyse(c, index, args.kind, tag, ef, recall_by_ef, local_search, kern_ms)
        kind_note = {"f32": "f32 rows (reference VecType = FullVec)",
                     "quant8": "8-bit codes dequantised to f32 (reference VecType = QuantVec, as shipped)"}
        out = {
            "metric": "queries/sec at recall@10>=0.99, 1M x 100d L2",
            "value": round(qps, 1), "unit": "queries/s", "n_gpus": world, "steps": K, "warmup": W,
            "ms_per_step": round(elapsed / K * 1e3, 5), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "configs[1]: %d x %dd fp32 L2, M=%d efSearch=%d, batch=%d queries per GPU; "
                                   "index rows: %s; synthetic GloVe-shaped recipe %s" % (
                                       N, d, m, ef, B, kind_note[args.kind], "AB"[args.recipe]),
                       "n_points": N, "dim": d, "M": m, "ef_construction": args.ef_cons, "efSearch": ef,
                       "n": n, "batch_per_gpu": B, "vec_kind": args.kind,
                       "parallelism": "replicated index, query batch sharded over %d GPU(s)" % world},
        }
        out.update(result)
        # ---- the other vector kind, same run (single GPU only) ---------------------------------------
        if world == 1 and not args.no_secondary:
            other = "quant8" if args.kind == "f32" else "f32"
            del index
            index2, tag2 = get_index(c, other)
            ef2, rec2 = choose_ef(c, index2, other)
            ls2 = make_device_search(index2, n, ef2, B, dev)
            el2, ms2 = time_local(c, ls2, K, W)
            sec = {"value": round(K * B / el2, 1), "unit": "queries/s", "ms_per_step": round(el2 / K * 1e3, 5),
                   "efSearch": ef2, "vec_kind": other, "index_rows": kind_note[other]}
            sec.update(analyse(c, index2, other, tag2, ef2, rec2, ls2, ms2))
            out["quant8_reference_default" if other == "quant8" else "f32_variant"] = sec
        print(json.dumps(out), file=json_out, flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
