#!/usr/bin/env python3
"""bench.py -- queries/sec of the HNSW search hot path on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config 1|2|3|4] [--n-points N]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

`python bench.py --gpus N` with N > 1 and no launcher environment starts the N ranks itself (one process
per GPU, before anything touches a GPU) and fails if the node has fewer GPUs; under a launcher it fails
when WORLD_SIZE is not N.  `n_gpus` in the JSON line is the size of the RCCL communicator.

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

Index build is outside the timed region.  N > 1: one process per GPU; rank 0 builds and the snapshot's
flat HBM arrays are broadcast to the other ranks over RCCL (HNSW.replicate), each group of steps'
1024 x N queries scattered from rank 0 and the results gathered back over RCCL (weak scaling),
exchange and search pipelined on two streams.

--config 3 (BASELINE configs[3]: 100M x 128d, replicated, query-sharded) is the same run at that size,
with efSearch walked up a coarser ladder.  --config 4 (configs[4]: 50M x 256d index build, points sharded,
RCCL neighbour all-gather) times the BUILD: every rank holds a replica and calls insert_bulk_sharded, the
insertion searches of each batch are split over the ranks and their edge records all-gathered; its line's
value is points inserted per second, with recall and the search rate of the built index beside it.
`--n-points` scales either down to what one GPU box builds in its time limit (say so in the line).

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


def kernel_sources_sha16():
    import hashlib
    h = hashlib.sha256()
    for f in ("search_kernels.hip", "search_lean.hip", "coop_rows.inc", "search_common.h", "device_index.h"):
        h.update(open(os.path.join(ROOT, "hnsw_rs_amd", "csrc", f), "rb").read())
    return h.hexdigest()[:16]
EF_LADDER = (64, 68, 72, 76, 80, 88, 96, 112, 128, 160, 192, 256)
EF_LADDER_LARGE = (64, 96, 128, 192, 256)  # configs[3] / [4]: 50-100M points need well beyond 64 for recall 0.99


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
    p.add_argument("--build-batch", default=None,
                   help="insert batches of the on-device build as MAX:DIV (min(MAX, connected / DIV) points at a "
                        "time).  Default 8192:8, the library default; 32768:8 for --config 4, whose metric is the build "
                        "(16M x 256d on one GPU: 17.1 -> 15.4 s, recall@10 unchanged; with the insertion searches "
                        "sharded over 8 ranks a batch of 8192 is 1024 waves per GPU -- less than half a machine).  Smaller batches stand closer to the reference's one-at-a-time "
                        "insertion: 256:64 lifts recall@10 by 0.0006 at efSearch 64 and 68 for 0.3 s more per 1M "
                        "points (scripts/build_schedule_recall.py) -- which puts f32 at 0.99008 at efSearch 64, i.e. "
                        "ON the metric's line (standard error 0.0003); the default keeps the headline clear of it")
    p.add_argument("--cpu-build", action="store_true",
                   help="build the index with the host threads only (default: on-device build, "
                        "hnsw_insert_bulk_device)")
    p.add_argument("--cpu-threads", type=int, default=0)
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--recall-queries", type=int, default=10240,
                   help="queries the true recall is measured on (all 10 bench batches: standard error 0.0003)")
    p.add_argument("--index-cache", default=os.environ.get("HNSW_BENCH_CACHE", "/tmp/hnsw_bench_cache"))
    p.add_argument("--config", type=int, default=1, choices=[1, 2, 3, 4],
                   help="BASELINE.json configs[i]: 1 = 1M x 100d fp32 L2, efSearch 64 (the metric); 2 = 10M x 768d fp32, "
                        "unit-normalised rows (cosine order = L2 order), efSearch 128; 3 = 100M x 128d L2, replicated "
                        "per GPU, query batch sharded; 4 = 50M x 256d index BUILD sharded over the GPUs")
    p.add_argument("--no-extras", action="store_true",
                   help="skip the PCIe-inclusive rate and the recipe-B efSearch sweep (profiling passes)")
    a = p.parse_args()
    if a.config == 2:  # configs[2]; explicit flags still win
        argv = " ".join(sys.argv[1:])
        if "--n-points" not in argv:
            a.n_points = 10_000_000
        if "--dim" not in argv:
            a.dim = 768
        if "--ef" not in argv:
            a.ef = "128"
        if "--recall-queries" not in argv:
            a.recall_queries = 2048
        if "--steps" not in argv:
            a.steps = 50
        a.no_secondary = True
        a.unit_rows = True
    else:
        a.unit_rows = False
    if a.config in (3, 4):  # configs[3] / configs[4]; explicit flags still win
        argv = " ".join(sys.argv[1:])
        if "--n-points" not in argv:
            a.n_points = 100_000_000 if a.config == 3 else 50_000_000
        if "--dim" not in argv:
            a.dim = 128 if a.config == 3 else 256
        if "--recall-queries" not in argv:
            a.recall_queries = 1024
        if "--steps" not in argv:
            a.steps = 50
        a.no_secondary = True
    return a


def cpu_share():
    """(threads to use, hardware concurrency, cgroup quota or None): the host cores this process may really
    use -- min(hardware concurrency, CPU affinity, cgroup quota)"""
    hw = os.cpu_count() or 1
    try:
        aff = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        aff = hw
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            quota = max(1, int(int(q) / int(per)))
    except (OSError, ValueError):
        pass
    return max(1, min(hw, aff, quota or hw)), hw, quota


def self_launch(args):
    """--gpus N > 1 without a launcher: start the N ranks here, before this process touches a GPU (a
    process that initialised HIP must not exec or fork workers), and hand on the child's line and code."""
    import subprocess
    import torch
    have = torch.cuda.device_count()  # counts devices without initialising them
    if have < args.gpus:
        sys.exit("bench.py --gpus %d: this node shows %d GPU(s)" % (args.gpus, have))
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    log("starting %d ranks: %s" % (args.gpus, " ".join(cmd)))
    sys.exit(subprocess.call(cmd))


class Ctx:
    pass


def make_rows(a, seed, first, n, threads, recipe=None):
    """synthetic GloVe-shaped rows; configs[2]: unit-normalised, so that the
    cosine order the config names is the L2 order the reference's metric gives"""
    import hnsw_rs_amd as H
    rows = H.synth_rows(a.recipe if recipe is None else recipe, seed, first, n, a.dim, threads)
    if a.unit_rows:
        rows /= np.linalg.norm(rows, axis=1, keepdims=True).astype(np.float32)
    return rows


def get_index(c, kind_name):
    """configs[1..3]: rank 0 builds (or loads a cached build) and the snapshot's flat HBM arrays are broadcast
    to the other ranks over RCCL (HNSW.replicate: SURVEY.md section 8e).  configs[4]: every rank holds a replica
    and the build itself is sharded (insert_bulk_sharded), timed into c.build_info."""
    import hnsw_rs_amd as H
    a = c.args
    kind = H.VEC_QUANT8 if kind_name == "quant8" else H.VEC_F32
    bb = a.build_batch or ("32768:8" if a.config == 4 else "8192:8")
    bmax, bdiv = (int(x) for x in bb.split(":"))
    c.build_batch = bb
    tag = "n%d_d%d_m%d_efc%d_%s_r%d%s%s" % (a.n_points, a.dim, a.m, a.ef_cons, kind_name, a.recipe,
                                           "_unit" if a.unit_rows else "", "" if bb == "8192:8" else "_b%d-%d" % (bmax, bdiv))
    cache_dir = os.path.join(a.index_cache, tag)
    index = None
    t0 = time.time()
    sharded_build = a.config == 4 and c.use_dist and not a.cpu_build
    builder = c.rank == 0 or sharded_build

    def slabs():
        # rows are generated and handed over in slabs (HNSW::insert_bulk may be called repeatedly,
        # template.rs:493-504): the host never holds more than one slab beside the index itself
        slab = a.n_points
        try:
            avail = int(next(l for l in open("/proc/meminfo") if l.startswith("MemAvailable")).split()[1]) * 1024
            if avail < 3 * 4 * a.dim * a.n_points * (c.world if sharded_build else 1):
                slab = max(1, min(a.n_points, (4 << 30) // (4 * a.dim)))
        except (OSError, StopIteration, ValueError):
            pass
        for first in range(0, a.n_points, slab):
            yield make_rows(a, 0x5EED0001, first, min(slab, a.n_points - first), min(32, c.ncpu))

    if builder:
        if a.config != 4 and os.path.isdir(cache_dir):
            try:
                index = H.HNSW.load(cache_dir)
                log("loaded cached %s index (%.1fs)" % (kind_name, time.time() - t0))
            except H.HnswError as e:
                log("cache unusable (%s); rebuilding" % e)
                shutil.rmtree(cache_dir, ignore_errors=True)
        if index is None:
            index = H.HNSW.new(a.m, a.ef_cons, a.dim, kind)
            index.set_device(c.local_rank)
            index.set_option("gpu_build_batch_max", bmax)
            index.set_option("gpu_build_batch_div", bdiv)
            build_s, gen_s = 0.0, 0.0
            tg = time.time()
            for store in slabs():
                gen_s += time.time() - tg
                if sharded_build:
                    c.dist.barrier()
                t1 = time.time()
                if a.cpu_build:
                    index.insert_bulk(store, c.build_threads, False)
                elif sharded_build:
                    index.insert_bulk_sharded(store, c.build_threads, False, device="cuda:%d" % c.local_rank)
                else:
                    index.insert_bulk_device(store, c.build_threads, False)
                if sharded_build:
                    c.torch.cuda.synchronize()
                    c.dist.barrier()
                build_s += time.time() - t1
                del store
                tg = time.time()
            if sharded_build:  # the build is as long as its slowest rank
                t = c.torch.tensor([build_s], dtype=c.torch.float64, device=c.dev)
                c.dist.all_reduce(t, op=c.dist.ReduceOp.MAX)
                build_s = float(t.item())
            c.build_info = {"seconds": round(build_s, 3), "points_per_s": round(a.n_points / build_s, 1),
                            "row_generation_seconds": round(gen_s, 1), "layers": index.nb_layers(),
                            "form": ("host build" if a.cpu_build else
                                     "on-device build, insertion searches sharded over %d rank(s) by position, connect / prune / drop by row "
                                     "ownership; edge records, removals and changed rows all-gathered over RCCL per batch "
                                     "(insert_bulk_sharded)" % c.world if sharded_build else
                                     "on-device build on one GPU (insert_bulk_device)"),
                            "host_threads": c.build_threads}
            if not a.cpu_build:
                c.build_info["device_counters"] = build_counters(index, a.dim, kind_name)
            log("built the %s index (%s, %d host threads) in %.1fs, %d layers" % (
                kind_name, c.build_info["form"], c.build_threads, build_s, index.nb_layers()))
            if a.config != 4 and c.rank == 0 and a.n_points <= 20_000_000:
                try:
                    os.makedirs(a.index_cache, exist_ok=True)
                    shutil.rmtree(cache_dir, ignore_errors=True)
                    index.save(cache_dir)
                except (H.HnswError, OSError) as e:
                    log("could not cache the index: %s" % e)
    if c.use_dist and not sharded_build:
        t1 = time.time()
        index = H.HNSW.replicate(index, a.m, a.ef_cons, a.dim, kind, src=0, device="cuda:%d" % c.local_rank)
        c.torch.cuda.synchronize()
        c.dist.barrier()
        c.replication = {"seconds": round(time.time() - t1, 3), "how": "RCCL broadcast of the snapshot's flat HBM arrays "
                         "from rank 0 (hnsw_snapshot_describe / _adopt / _commit)"}
        log("rank %d: snapshot replicated over RCCL in %.2fs" % (c.rank, time.time() - t1))
    else:
        index.set_device(c.local_rank)
        index.upload()
    log("rank %d: %s index resident in HBM, %.1f MB" % (c.rank, kind_name, index.device_bytes() / 1e6))
    return index, tag


def build_counters(index, d, kind_name):
    """What the on-device build of `index` read and how long its kernels ran (hnsw_get_stat "build_*"; this rank's
    share of the insertion searches when the build is sharded).  Algorithmic bytes of the build, the figure
    `build_roofline.achieved` uses: every vector row the insertion searches and the heuristic read (distance
    evaluations + staged rows, inserter.rs:40-126 / searcher.rs:109-153) x the un-padded row bytes, + 4 B per
    adjacency row read and per id in it, + 12 B per edge record filed."""
    g = lambda k: index.stat("build_" + k)
    row_bytes = (d + 8) if kind_name == "quant8" else 4 * d
    rows, adj, ids, rec = g("rows_read"), g("adj_rows"), g("adj_ids"), g("records")
    pts = max(1, g("points"))
    return {"points": pts, "batches": g("batches"), "rows_read": rows, "adj_rows_read": adj, "adj_ids_read": ids,
            "edge_records": rec, "removals": g("removals"), "row_bytes": row_bytes,
            "algorithmic_bytes": rows * row_bytes + 4 * adj + 4 * ids + 12 * rec,
            "per_point": {"rows_read": round(rows / pts, 1), "adj_rows_read": round(adj / pts, 1),
                          "bytes": round((rows * row_bytes + 4 * adj + 4 * ids + 12 * rec) / pts, 1)},
            "insert_kernel_s": g("insert_kernel_us") / 1e6, "insert_phase_s": g("insert_phase_us") / 1e6,
            "connect_phases_s": g("connect_us") / 1e6}


def choose_ef(c, index, kind_name):
    """true recall@n against exhaustive search under the index's own metric (the reference's own
    ground truth, hnsw/src/template.rs:531-541), on the first --recall-queries bench queries"""
    a, n = c.args, c.args.topn
    recall_by_ef = {}
    nr = min(a.recall_queries, c.queries.shape[0])
    qh = c.queries[:nr]
    # ground truth: the exact scan; for f32 rows the MFMA scan (screen + exact re-rank) after it has been
    # checked against the exact scan on the first 256 queries of this very index
    c.ground_truth = "exact scan (hx_brute_kernel)"
    if kind_name == "f32" and a.dim % 4 == 0 and n <= 12:
        bf, _ = index.brute_force_fast(qh, n)
        chk, _ = index.brute_force(qh[:256], n)
        if np.array_equal(chk, bf[:256]):
            c.ground_truth = "MFMA scan + exact re-rank (hx_brute_mfma_kernel), equal to the exact scan on 256 queries"
        else:
            log("MFMA scan differs from the exact scan on this index; using the exact scan")
            bf, _ = index.brute_force(qh, n)
    else:
        bf, _ = index.brute_force(qh, n)

    def recall_at(e):
        got, _, _, _ = index.search_batch(qh, n, e)
        hits = sum(len(set(x.tolist()) & set(y.tolist())) for x, y in zip(got, bf))
        return hits / float(nr * n)

    if a.ef == "auto":
        ef = None
        for e in (EF_LADDER if a.config == 1 else EF_LADDER_LARGE):
            recall_by_ef[e] = round(recall_at(e), 5)
            log("%s efSearch %d: true recall@%d = %.4f (%d queries)" % (kind_name, e, n, recall_by_ef[e], nr))
            if recall_by_ef[e] >= a.min_recall:
                ef = e
                break
        if ef is None:
            ef = EF_LADDER[-1] if a.config == 1 else EF_LADDER_LARGE[-1]
    else:
        ef = int(a.ef)
        recall_by_ef[ef] = round(recall_at(ef), 5)
    if 64 not in recall_by_ef:
        recall_by_ef[64] = round(recall_at(64), 5)
    return ef, recall_by_ef


def time_local(c, local_search, steps, warmup):
    """The timed region: K back-to-back launches on the current stream between two synchronisations, with one
    HIP event pair around all of them.  Returns (wall seconds, GPU ms per launch = event time / K).
    No events between the launches: a pair per launch costs ~7 us per step and reads 2 us long
    (scripts/timing_probe.py; rocprofv3's kernel durations agree with the figure returned here)."""
    torch = c.torch
    B, nqb = c.args.batch, c.args.query_batches
    qs = [c.dQ[b][:B].contiguous() for b in range(nqb)]
    for i in range(warmup):
        local_search(qs[i % nqb])
    torch.cuda.synchronize()
    r0, r1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    r0.record()
    for i in range(steps):
        local_search(qs[(warmup + i) % nqb])
    r1.record()
    while not r1.query():  # poll for the end (an interrupt-driven wait wakes up tens of microseconds late) ...
        pass
    torch.cuda.synchronize()  # ... and close the bracket as the contract asks
    elapsed = time.perf_counter() - t0
    return elapsed, r0.elapsed_time(r1) / steps


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
    # HBM traffic comes from separate rocprofv3 PMC passes (scripts/profile.sh) and is only quoted while
    # the kernels it was measured on are the ones running now (hash of the kernel sources)
    traffic, traffic_from, sq, ta = None, None, None, None
    tfile = os.path.join(ROOT, "profiles", "traffic_latest.json")
    if os.path.exists(tfile):
        try:
            tdoc = json.load(open(tfile))
            for tj in tdoc.get("entries", []):
                if tj.get("workload") == tag and tj.get("ef") == ef and tj.get("batch") == B:
                    if tdoc.get("kernel_sources_sha16") == kernel_sources_sha16():
                        traffic = tj.get("hbm_bytes_per_launch")
                        traffic_from = {"commit": tdoc.get("commit"), "efSearch": ef, "profile": tj.get("profile", tdoc.get("profile"))}
                        sq = tj.get("sq")
                        ta = tj.get("ta")
                    else:
                        traffic_from = {"stale": "kernels changed since commit %s; rerun scripts/profile.sh" % tdoc.get("commit")}
        except (OSError, ValueError, AttributeError):
            pass
    out["roofline"] = {
        "bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
        "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic, "traffic_measured_at": traffic_from,
        "kernel": (("hx_lean_f32_kernel" if kind == H.VEC_F32 else "hx_lean_q8_kernel")
                   if (d == 100 or (d == 128 and kind == H.VEC_F32 and os.environ.get("HNSW_MI355X_LEAN_128") != "0"))
                   and ef <= (128 if os.environ.get("HNSW_MI355X_LEAN_WIDE") == "0" else 256)
                   and os.environ.get("HNSW_MI355X_LEAN") != "0" and
                   (kind == H.VEC_F32 or os.environ.get("HNSW_MI355X_LEAN_Q8") != "0") else "hx_search_kernel"),
        "kernel_ms": round(kern_ms, 5),
        "kernel_ms_note": "average launch duration: one HIP event pair around the K launches of the timed region, / K",
        "algorithmic_bytes_per_launch": round(bytes_per_launch),
        "per_query": {"n_dist": round(float(st[:, 0].mean()), 2), "n_exp": round(float(st[:, 1].mean()), 2),
                      "sum_deg": round(float(st[:, 2].mean()), 2), "bytes": round(float(bq.mean()), 1),
                      "row_bytes": row_bytes},
    }
    if sq and sq.get("SQ_WAVE_CYCLES"):
        # The other roofs of the same kernel, from the SQ counters of a separate PMC pass (scripts/pmc_sq.sh): a wave
        # alone on its SIMD (a 1024-query launch) is bound by its own instruction stream and its waits, not by bytes.
        # SQ_WAVE_CYCLES and the SQ_WAIT_* / SQ_ACTIVE_* counters tick once per four clocks; a vector instruction holds
        # its SIMD's issue for four clocks, so instructions / wave-cycles IS the fraction of the VALU issue roof.
        wc = float(sq["SQ_WAVE_CYCLES"])
        out["roofline"]["issue"] = {
            "valu_issue_frac": round(sq.get("SQ_INSTS_VALU", 0.0) / wc, 4),
            "salu_issue_frac": round(sq.get("SQ_INSTS_SALU", 0.0) / wc, 4),
            "waiting_frac": round(sq.get("SQ_WAIT_ANY", 0.0) / wc, 4),
            "instruction_fetch_wait_frac": round(sq.get("SQ_WAIT_INST_ANY", 0.0) / wc, 4),
            "per_query": {"valu": round(sq.get("SQ_INSTS_VALU", 0.0) / B, 1), "salu": round(sq.get("SQ_INSTS_SALU", 0.0) / B, 1),
                          "lds": round(sq.get("SQ_INSTS_LDS", 0.0) / B, 1), "vmem_rd": round(sq.get("SQ_INSTS_VMEM_RD", 0.0) / B, 1),
                          "wave_cycles_x4": round(wc / B, 1)},
            "note": "valu_issue_frac = SQ_INSTS_VALU x 4 clocks / (SIMDs x the kernel's clocks) with one wave per SIMD; "
                    "counters from the profile named in traffic_measured_at"}
    if ta and ta.get("GRBM_GUI_ACTIVE") and ta.get("TA_TA_BUSY_sum"):
        # ... and the CU's L1 path (scripts/pmc_ta.sh): a lane reads its row 16 bytes at a time, one L1 tag access per
        # piece; neither the texture addresser nor the tag rate is a roof here -- the path is a third used
        cyc = float(ta["GRBM_GUI_ACTIVE"]) / 8.0  # (summed over the 8 XCDs)
        per_cu = lambda k: float(ta.get(k, 0.0)) / 256.0 / cyc
        out["roofline"]["l1_path"] = {
            "ta_busy_frac": round(per_cu("TA_TA_BUSY_sum"), 4),
            "tag_access_frac_of_one_per_clock": round(per_cu("TCP_TOTAL_CACHE_ACCESSES_sum"), 4),
            "waiting_for_l2_data_frac": round(per_cu("TCP_PENDING_STALL_CYCLES_sum"), 4),
            "ta_stalled_by_l1_frac": round(per_cu("TA_ADDR_STALLED_BY_TC_CYCLES_sum"), 4),
            "per_query": {"l1_tag_accesses": round(float(ta.get("TCP_TOTAL_CACHE_ACCESSES_sum", 0.0)) / B, 1),
                          "l2_read_requests": round(float(ta.get("TCP_TCC_READ_REQ_sum", 0.0)) / B, 1)},
            "note": "TA / TCP counters summed over 256 CUs / (256 x the kernel's clocks); counters from the profile named in "
                    "traffic_measured_at (profiles/r04_ta_tcp_counters.txt)"}
    out["recall_at_%d" % n] = recall_by_ef[ef]
    out["ground_truth"] = getattr(c, "ground_truth", None)
    out["recall_by_ef"] = {str(k): v for k, v in sorted(recall_by_ef.items())}
    if ef != 64:  # the configured efSearch = 64 timed too when the metric's recall needed a larger ef
        from hnsw_rs_amd.distributed import make_device_search
        ls64 = make_device_search(index, n, 64, B, c.dev)
        el64, ms64 = time_local(c, ls64, 50, 5)
        st64 = ls64.stats[:B].cpu().numpy().astype(np.float64)  # the last launch's traversal counters
        bq64 = st64[:, 0] * row_bytes + st64[:, 1] * 4 + st64[:, 2] * 4 + 4 * d + 8 * n
        out["at_configured_efSearch_64"] = {"queries_per_s_per_gpu": round(50 * B / el64, 1),
                                            "ms_per_step": round(el64 / 50 * 1e3, 5), "kernel_ms": round(ms64, 5),
                                            "hbm_frac_algorithmic": round(float(bq64.mean() * B) / (ms64 * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                                            "recall_at_%d" % n: recall_by_ef[64]}
    # ---- the roof that binds a 1024-query batch: one query's own dependent chain.  Measured live: the same kernel
    # over 64 queries (a wave alone on its CU -- nothing to share the address path, the LDS or the issue slots
    # with); a batch of 1024 is four such waves per CU, and how close its time stays to this is how little they
    # cost each other.  The cycle-level decomposition (stamps build, scripts/latency_floor.py) rides along while
    # the kernels it was measured on are the ones running now. ----
    if a.config == 1 and not a.no_extras:  # (profiling passes keep the kernel trace to full-size launches)
        try:
            from hnsw_rs_amd.distributed import make_device_search
            ls_few = make_device_search(index, n, ef, 64, c.dev)
            qf = [c.dQ[b][:64].contiguous() for b in range(nqb)]
            for i in range(5):
                ls_few(qf[i % nqb])
            torch.cuda.synchronize()
            g0, g1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            g0.record()
            for i in range(50):
                ls_few(qf[i % nqb])
            g1.record()
            torch.cuda.synchronize()
            ms_few = g0.elapsed_time(g1) / 50
            lf = {"kernel_ms_64_queries": round(ms_few, 5), "kernel_ms_1024_queries": round(kern_ms, 5),
                  "floor_over_kernel": round(ms_few / kern_ms, 4),
                  "note": "a query's ~%d dependent expansions cannot run faster than this whatever the batch size: against "
                          "THIS roof the 1024-query launch is at %.0f %%; the HBM roofline above is the roof of the "
                          "machine-filling regime (launches of >= 8192 queries: DESIGN.md section 10)" % (
                              round(float(st[:, 1].mean())), 100.0 * ms_few / kern_ms)}
            ffile = os.path.join(ROOT, "profiles", "latency_floor_latest.json")
            if os.path.exists(ffile):
                fdoc = json.load(open(ffile))
                if fdoc.get("kernel_sources_sha16") == kernel_sources_sha16() and fdoc.get("vec_kind") == kind_name and fdoc.get("efSearch") == ef:
                    lf["dependent_steps"] = fdoc.get("launches")
                    lf["dependent_steps_note"] = fdoc.get("note")
            out["latency_floor"] = lf
        except Exception as e:
            log("latency floor skipped: %s" % e)
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
    need_host = 2.3 * 4.0 * d * N + 400.0 * N  # rows twice (generated + the oracle's copy) + both graphs
    try:
        avail_host = int(next(l for l in open("/proc/meminfo") if l.startswith("MemAvailable")).split()[1]) * 1024.0
    except (OSError, StopIteration, ValueError):
        avail_host = float("inf")
    if not a.no_cpu_baseline and avail_host < need_host:
        out["cpu_baseline"] = {"value": None, "unit": "queries/s", "cores": 0, "kind": "port",
                               "sample": "skipped: the oracle would need %.0f GB of host memory beside the index, %.0f GB "
                                         "are available" % (need_host / 1e9, avail_host / 1e9)}
        log("cpu baseline skipped: not enough host memory for a second copy of the index")
    elif not a.no_cpu_baseline:
        from oracle import oracle_py as O
        t2 = time.time()
        orc = O.OracleHNSW(m, a.ef_cons, d, kind)
        store = make_rows(a, 0x5EED0001, 0, N, min(32, c.ncpu))
        lv = np.zeros(N, dtype=np.uint8)
        for l in range(1, index.nb_layers()):
            lv[index.get_layer(l).iter_nodes()] = l
        orc.import_points(store, lv)
        del store
        for l in range(index.nb_layers()):
            orc.import_layer(l, *index.get_layer(l).csr())
        orc.set_ep(int(index.params.ep))
        log("oracle holds the same %s index (%.1fs)" % (kind_name, time.time() - t2))
        # T = std::thread::hardware_concurrency() as far as this process may use it: the cgroup quota and
        # the CPU affinity bound it (a GPU box hands 16 of its 256 hardware threads to one GPU's job)
        T_share, hw, quota = cpu_share()
        T = a.cpu_threads or T_share
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
            "hardware_concurrency": hw, "cpu_quota": quota,
            "host_cpu": next((l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo")
                              if l.startswith("model name")), "unknown"),
        }
        out["parity"] = {"queries": int(qcpu.shape[0]), "ids_identical_fraction": same,
                         "counters_identical": bool(np.array_equal(o_st.astype(np.int64), st[:, :3]))}
        log("%s: cpu oracle %.0f q/s on %d threads; GPU ids identical for %.4f of queries" % (
            kind_name, reps * qcpu.shape[0] / cpu_s, T, same))
        del orc
    return out


def extras(c, index, kind_name, ef, local_search):
    """SURVEY 8(d) figures beside the metric (rank 0, one GPU): the PCIe-inclusive rate of the host-pointer
    entry, and the efSearch sweep on the hard isotropic recipe B"""
    import hnsw_rs_amd as H
    a, torch = c.args, c.torch
    n, B, nqb = a.topn, a.batch, a.query_batches
    out = {}
    # ---- hnsw_search_batch: pageable host buffers in, host buffers out, one H2D + one D2H per call ----
    try:
        qh = [np.ascontiguousarray(c.queries.reshape(nqb, -1, a.dim)[b][:B]) for b in range(nqb)]
        for b in range(min(3, nqb)):
            index.search_batch(qh[b], n, ef)
        steps = 40 if a.config == 1 else 10
        t0 = time.perf_counter()
        for i in range(steps):
            index.search_batch(qh[i % nqb], n, ef)
        dt = time.perf_counter() - t0
        out["pcie_inclusive"] = {"queries_per_s": round(steps * B / dt, 1), "ms_per_step": round(dt / steps * 1e3, 5),
                                 "efSearch": ef, "entry": "hnsw_search_batch (host pointers; H2D of the queries, D2H of ids / "
                                 "distances / counts / statistics, status check and overflow retry included)",
                                 "note": "never reported as `value`"}
        # the same entry from C threads (no interpreter in the loop: what a Rust caller sees), one and several
        # callers at once -- the entry is re-entrant, each call leases its own stream and staging, and the copies of
        # one call run under the kernel of another
        qall = np.ascontiguousarray(c.queries.reshape(nqb, -1, a.dim)[:, :B].reshape(-1, a.dim))
        conc = {}
        for C in (1, 2, 4):
            conc[str(C)] = round(index.batch_threads(qall, B, n, ef, C, 2 * steps), 1)
        out["pcie_inclusive"]["queries_per_s_c_callers"] = conc
    except Exception as e:
        log("pcie-inclusive measurement skipped: %s" % e)
    # ---- the reference's own call pattern: ONE query per call, T host threads each blocked in its call
    # (ann_by_vector(&self, ...), template.rs:306-335) through hnsw_search, the entry the Rust shim binds ----
    if a.config == 1:
        try:
            qs = np.ascontiguousarray(c.queries.reshape(nqb, -1, a.dim)[:, :B].reshape(-1, a.dim))
            want, _, want_c, _ = index.search_batch(qs, n, ef)  # (the batch path is held to the oracle in `parity`)
            legs = []
            lat1 = None
            for T, window in ((1, 30), (16, 30), (64, 30), (256, 30), (1024, 30), (16, -1)):
                index.set_option("coalesce_us", window)
                keys = ("coalesced_batches", "coalesced_queries")
                s0 = [index.stat(k) for k in keys]
                ids, counts, calls, wall, lat = index.search_threads(qs, n, ef, T, 1.0)
                nb, nqd = (index.stat(k) - x for k, x in zip(keys, s0))
                if T == 1:
                    lat1 = lat["p50"]
                legs.append({"threads": T, "coalescing": "off (every call launches by itself)" if window < 0 else "on",
                             "queries_per_s": round(calls / wall, 1), "p50_us": round(lat["p50"], 1),
                             "p99_us": round(lat["p99"], 1), "mean_batch": round(nqd / nb, 2) if nb else 1.0,
                             "host_cpu_cores_used": round((lat["cpu_user_s"] + lat["cpu_sys_s"]) / wall, 2),
                             "ceiling_threads_over_one_call_latency": round(T / (lat1 * 1e-6), 1) if lat1 else None,
                             "answers_identical_to_batch_path": bool(np.array_equal(ids, want) and np.array_equal(counts, want_c))})
            index.set_option("coalesce_us", 30)
            out["single_query_api"] = {
                "entry": "hnsw_search (one query per call, the shim's ann_by_vector); T threads of hnsw_bench_search_threads, "
                         "1 s each over the %d bench queries" % len(qs),
                "efSearch": ef, "legs": legs,
                "note": "a call cannot return before its own query's dependent chain of expansions has run (p50 at one "
                        "thread); T blocked callers therefore answer at most T / that latency per second whatever the "
                        "GPU could do with more queries in flight (Little's law) -- the ceiling column.  The CPU oracle's "
                        "threads are in cpu_baseline."}
        except Exception as e:
            log("single-query measurement skipped: %s" % e)
    # ---- recipe B (isotropic clusters): qps at the first efSearch that reaches the recall ----
    if a.config == 1 and a.recipe == 0:
        try:
            from hnsw_rs_amd.distributed import make_device_search
            t0 = time.time()
            kind = H.VEC_QUANT8 if kind_name == "quant8" else H.VEC_F32
            idxb = H.HNSW.new(a.m, a.ef_cons, a.dim, kind)
            idxb.set_device(c.local_rank)
            idxb.insert_bulk_device(make_rows(a, 0x5EED0001, 0, a.n_points, min(32, c.ncpu), recipe=1), c.build_threads, False)
            idxb.upload()
            qb = make_rows(a, 0x5EED0002, 0, 2 * B, min(16, c.ncpu), recipe=1)
            dqb = torch.from_numpy(qb).to(c.dev)
            truth, _ = idxb.brute_force(qb[:B], n)
            sweep = []
            for e in (64, 128, 256, 512):
                ls = make_device_search(idxb, n, e, B, c.dev)
                got, _ = ls(dqb[:B].contiguous())
                ls.check()
                got = got.cpu().numpy().view(np.uint32)
                rec = sum(len(set(x.tolist()) & set(y.tolist())) for x, y in zip(got, truth)) / float(B * n)
                for _ in range(3):
                    ls(dqb[B:].contiguous())
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for i in range(10):
                    ls(dqb[(i % 2) * B:(i % 2 + 1) * B].contiguous())
                e1.record()
                torch.cuda.synchronize()
                ms = e0.elapsed_time(e1) / 10
                sweep.append({"efSearch": e, "recall_at_%d" % n: round(rec, 4), "queries_per_s": round(B / ms * 1e3, 1),
                              "ms_per_step": round(ms, 5)})
            out["recipe_B_ef_sweep"] = {"data": "recipe B: 4096 isotropic clusters, the hard case of SURVEY 8(d); M=%d "
                                        "ef_construction=%d" % (a.m, a.ef_cons), "vec_kind": kind_name,
                                        "recall_queries": B, "sweep": sweep,
                                        "note": "this graph does not reach recall 0.99 at M=16 / ef_construction=32 on "
                                                "this data whoever builds it (DESIGN.md section 10)"}
            log("recipe B sweep done in %.1fs" % (time.time() - t0))
            del idxb
        except Exception as e:
            log("recipe-B sweep skipped: %s" % e)
    return out


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args)  # does not return
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
        sys.exit("bench.py --gpus %d was started with WORLD_SIZE=%d: the two must agree" % (args.gpus, world))
    if not torch.cuda.is_available():
        sys.exit("bench.py needs an MI355X: torch.cuda.is_available() is False")
    torch.cuda.set_device(c.local_rank)
    c.dev = dev = torch.device("cuda", c.local_rank)
    # HNSW_BENCH_FORCE_DIST=1 runs the scatter / gather path even with one rank (a smoke test of the
    # multi-GPU code on a single-GPU box)
    force_dist = os.environ.get("HNSW_BENCH_FORCE_DIST") == "1"
    use_dist = world > 1 or force_dist
    c.use_dist = use_dist
    if use_dist:
        if "MASTER_ADDR" not in os.environ:
            os.environ["MASTER_ADDR"] = "127.0.0.1"
            os.environ["MASTER_PORT"] = os.environ.get("MASTER_PORT", "29533")
        dist.init_process_group("nccl", device_id=dev, rank=rank, world_size=world)
        if dist.get_world_size() != args.gpus and not force_dist:
            sys.exit("RCCL communicator has %d ranks, --gpus says %d" % (dist.get_world_size(), args.gpus))
    n_gpus = dist.get_world_size() if use_dist else 1

    N, d, m, n, B = args.n_points, args.dim, args.m, args.topn, args.batch
    c.ncpu = os.cpu_count() or 8
    c.build_threads = args.build_threads or max(1, min(32, c.ncpu // max(1, world)))
    nqb = args.query_batches
    c.queries = make_rows(args, 0x5EED0002, 0, nqb * B * world, min(16, c.ncpu))
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
    G = int(os.environ.get("HNSW_BENCH_GROUP", "8"))  # steps per exchange group (bucketed collectives)
    pipe = PipelinedShardedSearch.from_index(index, d, n, ef, B, dev, group_steps=G) if use_dist else None
    K, W = args.steps, args.warmup

    def run_dist(first, count):
        # (starting with smaller groups so that the first, un-overlapped scatter is short was measured: no gain --
        # a 20-step run loses its ~0.5 ms to the host enqueueing the first group and to the closing barrier)
        i = first
        while i < first + count:
            g = min(G, first + count - i)
            qg = None
            if rank == 0:  # the group's queries are put together on the lane that scatters them
                with pipe.lanes.on(pipe.lanes.comm):
                    qg = torch.stack([c.dQ[(i + j) % nqb] for j in range(g)], 0)
            pipe.submit(qg, g)  # scatter / searches / gather of neighbouring groups overlap
            i += g
        return g  # steps in the last group

    if pipe is None:
        elapsed, kern_ms = time_local(c, local_search, K, W)
    else:
        g_last = run_dist(0, W) if W > 0 else 0
        pipe.finish()
        if rank == 0 and W > 0:  # the exchange returns what a local search returns
            last = W - 1
            chk_ids, chk_d = pipe.results(pipe.n_groups - 1, g_last - 1)
            ref_ids, ref_d = local_search(c.dQ[last % nqb][:B].contiguous())
            torch.cuda.synchronize()
            assert torch.equal(chk_ids[:B], ref_ids) and torch.equal(chk_d[:B], ref_d), "gathered != local"
        torch.cuda.synchronize()
        dist.barrier()
        t_start = time.perf_counter()
        run_dist(W, K)  # exactly K steps
        t_a = time.perf_counter()
        pipe.finish()
        t_b = time.perf_counter()
        torch.cuda.synchronize()
        t_c = time.perf_counter()
        dist.barrier()
        elapsed = time.perf_counter() - t_start
        if os.environ.get("HNSW_BENCH_TRACE") == "1":
            log("rank %d timed region: enqueue %.0f us, finish (last gather + drain + check) %.0f us, synchronize %.0f us, "
                "barrier %.0f us, total %.0f us for %d steps" % (rank, (t_a - t_start) * 1e6, (t_b - t_a) * 1e6,
                                                              (t_c - t_b) * 1e6, (t_start + elapsed - t_c) * 1e6,
                                                              elapsed * 1e6, K))
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
            "value": round(qps, 1), "unit": "queries/s", "n_gpus": n_gpus, "steps": K, "warmup": W,
            "ms_per_step": round(elapsed / K * 1e3, 5), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "configs[%d]: %d x %dd fp32 L2%s, M=%d efSearch=%d, batch=%d queries per GPU; "
                                   "index rows: %s; synthetic GloVe-shaped recipe %s" % (
                                       args.config, N, d,
                                       " on unit-normalised rows (cosine order = L2 order; the reference has no cosine metric)"
                                       if args.unit_rows else "", m, ef, B, kind_note[args.kind], "AB"[args.recipe]),
                       "n_points": N, "dim": d, "M": m, "ef_construction": args.ef_cons, "efSearch": ef,
                       "n": n, "batch_per_gpu": B, "vec_kind": args.kind, "index_tag": tag,
                       "build_batch": getattr(c, "build_batch", None),
                       "parallelism": "replicated index, query batch sharded over %d GPU(s)" % world},
        }
        out.update(result)
        if args.config == 2:
            out["metric"] = "queries/sec, %d x %dd fp32, unit rows, efSearch %d (BASELINE configs[2]; recall@10 reported)" % (N, d, ef)
        if args.config == 3:
            out["metric"] = ("queries/sec, %d x %dd fp32 L2, index replicated per GPU, query batch sharded, efSearch %d "
                             "(BASELINE configs[3]; recall@10 reported)" % (N, d, ef))
        if getattr(c, "replication", None):
            out["replication"] = c.replication
        if getattr(c, "build_info", None):
            out["index_build"] = c.build_info
        if args.config == 4:
            # configs[4] times the BUILD: one "step" is the whole insert_bulk of N points
            bi = c.build_info
            out["search_after_build"] = {"queries_per_s": out["value"], "ms_per_step": out["ms_per_step"], "steps": K,
                                         "warmup": W, "efSearch": ef, "note": "search rate of the index this run built"}
            sharded = "insert_bulk_sharded" in bi["form"]
            out["metric"] = ("points/sec inserted, index build %d x %dd fp32 (BASELINE configs[4]: on-device insert / "
                             "search_layer; %s; recall@10 of the built index reported)" % (
                                 N, d, ("insertion searches sharded over %d ranks, connect / prune / drop by row ownership, edge records / "
                                        "removals / changed rows all-gathered over RCCL per batch" % world) if sharded else
                                 "ONE GPU, insert_bulk_device: the sharded path (insert_bulk_sharded, RCCL all-gather) was NOT "
                                 "exercised by this run"))
            out["value"], out["unit"] = bi["points_per_s"], "points/s"
            out["steps"], out["warmup"], out["ms_per_step"] = 1, 0, round(bi["seconds"] * 1e3, 3)
            out["scaling"] = "strong"
            out["config"]["parallelism"] = (("every rank holds a replica; insertion searches of each batch split over %d GPUs, "
                                             "connect / prune / drop by row ownership, records / removals / changed rows all-gathered over RCCL" % world) if sharded else
                                            "one GPU, no ranks, no collective (the build ran as %s)" % bi["form"])
            # the roofline of the headline value's dominant kernel (hx_insert_kernel) and the CPU build beside it;
            # `roofline` / `cpu_baseline` above describe the SEARCH on the index this run built
            out["search_roofline_note"] = "`roofline`, `cpu_baseline`, `parity`, `recall_*` describe the search on the built index; the build's own are `build_roofline` and `cpu_build_baseline`"
            dc = bi.get("device_counters")
            if dc and dc["insert_kernel_s"] > 0:
                ach = dc["algorithmic_bytes"] / dc["insert_kernel_s"] / 1e9
                btraffic, bfrom = None, None
                tfile = os.path.join(ROOT, "profiles", "traffic_latest.json")
                try:
                    tdoc = json.load(open(tfile))
                    for tj in tdoc.get("entries", []):
                        if tj.get("workload") == tag and tj.get("kernel") == "hx_insert_kernel":
                            if tdoc.get("kernel_sources_sha16") == kernel_sources_sha16():
                                btraffic = tj.get("hbm_bytes_all_launches")
                                bfrom = {"commit": tdoc.get("commit"), "profile": tj.get("profile", tdoc.get("profile"))}
                            else:
                                bfrom = {"stale": "kernels changed since commit %s" % tdoc.get("commit")}
                except (OSError, ValueError, AttributeError):
                    pass
                out["build_roofline"] = {
                    "bound": "hbm", "kernel": "hx_insert_kernel", "achieved": round(ach, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(ach / HBM_PEAK_GBS, 5), "traffic": btraffic, "traffic_measured_at": bfrom,
                    "kernel_s": round(dc["insert_kernel_s"], 3),
                    "kernel_s_note": "sum of the insert kernel's launches, one HIP event pair around each, on its stream%s" % (
                        " (rank 0's share of the insertion searches)" if sharded else ""),
                    "algorithmic_bytes": dc["algorithmic_bytes"], "launches": dc["batches"],
                    "algorithmic_bytes_note": "rows read by the insertion searches and the heuristic (distance evaluations + "
                                              "staged rows) x %d B + 4 B per adjacency row and per id read + 12 B per edge "
                                              "record filed, counted by the kernel itself" % dc["row_bytes"],
                    "per_point": dc["per_point"],
                    "share_of_build": round(dc["insert_kernel_s"] / bi["seconds"], 3),
                    "other_phases_s": {"sort_connect_remove": round(dc["connect_phases_s"], 3),
                                       "host_and_transfers": round(bi["seconds"] - dc["insert_kernel_s"] - dc["connect_phases_s"], 3)}}
            if world == 1 and not args.no_cpu_baseline:
                try:
                    T_share, hw, quota = cpu_share()
                    T = args.cpu_threads or T_share
                    n_cpu = min(N, int(os.environ.get("HNSW_BENCH_CPU_BUILD_POINTS", "1000000")))
                    rows_cpu = make_rows(args, 0x5EED0001, 0, n_cpu, min(32, c.ncpu))
                    cpu_idx = H.HNSW.new(m, args.ef_cons, d, H.VEC_QUANT8 if args.kind == "quant8" else H.VEC_F32)
                    t0 = time.time()
                    cpu_idx.insert_bulk(rows_cpu, T, False)
                    cpu_s = time.time() - t0
                    del cpu_idx, rows_cpu
                    out["cpu_build_baseline"] = {
                        "value": round(n_cpu / cpu_s, 1), "unit": "points/s", "cores": T, "kind": "port",
                        "sample": "the first %d of the %d rows inserted by the library's host insert_bulk (the reference's "
                                  "algorithm and threading, template.rs:388-444: one thread pool per layer, per-row locks) on "
                                  "%d threads = %.1f s; a smaller index is cheaper per point (the searches are shorter), so "
                                  "this flatters the CPU" % (n_cpu, N, T, cpu_s),
                        "hardware_concurrency": hw, "cpu_quota": quota}
                    log("cpu build baseline: %d points in %.1fs on %d threads" % (n_cpu, cpu_s, T))
                except Exception as e:
                    log("cpu build baseline skipped: %s" % e)
        if world == 1 and not args.no_extras:
            out.update(extras(c, index, args.kind, ef, local_search))
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
