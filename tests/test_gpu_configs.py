"""Parity at the shapes BASELINE.json's configs name and on every kernel instantiation `launch_search` can
pick for them: the timed f32 d = 100 kernel at efSearch 64 .. 256, d = 128 and
d = 768 at efSearch 64 / 128, all on indexes of >= 20 000 points built on the device, plus the device and the
two-rank sharded build at d = 256.  Bar as everywhere: ids, distance bits and traversal counters identical
to the CPU oracle (hnsw/src/template/searcher.rs:23-103, hnsw/src/template.rs:306-335)."""
import os
import subprocess
import sys

import numpy as np
import pytest

import hnsw_rs_amd as H
from oracle import oracle_py as O
from tests.util import assert_search_equal, oracle_from_product

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def device_built(n, d, m, kind, recipe=0, unit=False):
    vs = H.synth_rows(recipe, 0x5EED0001, 0, n, d)
    qs = H.synth_rows(recipe, 0x5EED0002, 0, 128, d)
    if unit:  # configs[2]: cosine order = L2 order on unit-normalised rows
        vs /= np.linalg.norm(vs, axis=1, keepdims=True).astype(np.float32)
        qs /= np.linalg.norm(qs, axis=1, keepdims=True).astype(np.float32)
    lv = O.draw_levels(n, m, 0x5EED0003)
    idx = H.HNSW.new(m, 32, d, kind)
    idx.insert_bulk_device(vs, 8, False, levels=lv)
    return idx, oracle_from_product(idx, vs, lv), qs


@pytest.fixture(scope="module")
def f32_100d():
    return device_built(24000, 100, 16, H.VEC_F32)


@pytest.mark.parametrize("ef", [64, 68, 96, 128, 129, 192, 256, 257, 320, 321, 384, 385, 400, 448, 449, 512, 513])
def test_the_timed_f32_kernel(f32_100d, ef):
    """configs[1]: 100d f32, M = 16 -- the lean kernel with one list register (ef <= 64), head + tail (<= 128), four
    and eight interleaved registers (<= 256, <= 512), the generic kernel beyond"""
    idx, orc, qs = f32_100d
    assert_search_equal(idx.search_batch(qs, 10, ef), orc.search_batch(qs, 10, ef, nthreads=8), "f32 100d ef=%d" % ef)


def test_the_generic_f32_kernel_still_agrees(f32_100d):
    """HNSW_MI355X_LEAN=0 routes d = 100 f32 through hx_search_kernel's two-rows-per-pass loop (the round-1
    kernel, still what serves every other dimension); the choice is read once per process"""
    code = (
        "import sys, numpy as np; sys.path.insert(0, %r)\n"
        "import hnsw_rs_amd as H\n"
        "from oracle import oracle_py as O\n"
        "from tests.util import oracle_from_product, assert_search_equal\n"
        "n, d, m = 20000, 100, 16\n"
        "vs = H.synth_rows(0, 0x5EED0001, 0, n, d); qs = H.synth_rows(0, 0x5EED0002, 0, 128, d)\n"
        "lv = O.draw_levels(n, m, 7)\n"
        "idx = H.HNSW.new(m, 32, d, H.VEC_F32); idx.insert_bulk_device(vs, 8, False, levels=lv)\n"
        "orc = oracle_from_product(idx, vs, lv)\n"
        "for ef in (64, 68, 128, 192, 256):\n"
        "    assert_search_equal(idx.search_batch(qs, 10, ef), orc.search_batch(qs, 10, ef, nthreads=8), 'generic ef=%%d' %% ef)\n"
        "print('generic ok')\n" % ROOT)
    out = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, HNSW_MI355X_LEAN="0"),
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "generic ok" in out.stdout, out.stdout + out.stderr


def test_upload_without_pinned_staging_memory():
    """the snapshot upload streams through pinned buffers; where the host refuses to pin memory it falls back to a
    plain buffer and blocking copies (HNSW_MI355X_NO_PINNED=1 forces that) -- same snapshot, same answers"""
    code = (
        "import sys, numpy as np; sys.path.insert(0, %r)\n"
        "import hnsw_rs_amd as H\n"
        "from oracle import oracle_py as O\n"
        "from tests.util import oracle_from_product, assert_search_equal\n"
        "n, d, m = 30000, 100, 16\n"
        "vs = H.synth_rows(0, 0x5EED0001, 0, n, d); qs = H.synth_rows(0, 0x5EED0002, 0, 64, d)\n"
        "lv = O.draw_levels(n, m, 11)\n"
        "for kind in (H.VEC_F32, H.VEC_QUANT8):\n"
        "    idx = H.HNSW.new(m, 32, d, kind); idx.insert_bulk(vs, 8, False, levels=lv)\n"
        "    orc = oracle_from_product(idx, vs, lv)\n"
        "    assert_search_equal(idx.search_batch(qs, 10, 64), orc.search_batch(qs, 10, 64, nthreads=8), 'kind %%d' %% kind)\n"
        "print('plain staging ok')\n" % ROOT)
    out = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, HNSW_MI355X_NO_PINNED="1"),
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "plain staging ok" in out.stdout, out.stdout + out.stderr


@pytest.mark.parametrize("kind", [H.VEC_F32, H.VEC_QUANT8])
@pytest.mark.parametrize("limit", ["", "300", "2000"])
def test_two_level_visited_set_of_the_eight_register_kernel(limit, kind):
    """320 < ef <= 512 at d = 100, f32 and quant8 rows (round 4): the LDS visited table stays at 32 KiB and the ids beyond its limit
    go to a second level in HBM (Visited::look2, search_lean.hip).  HNSW_MI355X_VISITED_2L_LIMIT closes the LDS
    level after 300 / 2000 ids, so that every query crosses into the second level early / half way; results and
    counters must be the one-level table's, i.e. the oracle's (IntSet, results.rs:101-103).  Also the retry with a
    larger table when the second level fills up, and rows longer than the stride across the switch."""
    code = (
        "import sys, numpy as np; sys.path.insert(0, %r)\n"
        "import hnsw_rs_amd as H\n"
        "from oracle import oracle_py as O\n"
        "from tests.util import oracle_from_product, assert_search_equal\n"
        "n, d, m = 60000, 100, 16\n"
        "vs = H.synth_rows(0, 0x5EED0001, 0, n, d); qs = H.synth_rows(0, 0x5EED0002, 0, 192, d)\n"
        "lv = O.draw_levels(n, m, 7)\n"
        "idx = H.HNSW.new(m, 32, d, %d); idx.insert_bulk_device(vs, 8, False, levels=lv)\n"
        "orc = oracle_from_product(idx, vs, lv)\n"
        "for ef in (321, 384, 448, 512):\n"
        "    got = idx.search_batch(qs, 10, ef)\n"
        "    assert_search_equal(got, orc.search_batch(qs, 10, ef, nthreads=8), 'two-level ef=%%d' %% ef)\n"
        "    assert int(np.asarray(got[3])[:, 0].max()) > 2500  # (the searches do reach past the early limits)\n"
        "assert_search_equal(idx.search_batch(qs[:64], 400, 512), orc.search_batch(qs[:64], 400, 512, nthreads=8), 'n=400')\n"
        "# a hub with 3000 neighbours: its overflow rows are expanded in passes that straddle the switch\n"
        "ids, offs, nbrs = idx.get_layer(0).csr()\n"
        "rows = [set(nbrs[offs[i]:offs[i + 1]].tolist()) for i in range(len(ids))]\n"
        "hub = int(orc.search_batch(qs[:1], 1, 64)[0][0, 0])\n"
        "for x in range(0, 9000, 3):\n"
        "    if x != hub: rows[hub].add(x); rows[x].add(hub)\n"
        "flat = np.concatenate([np.array(sorted(r), dtype=np.uint32) for r in rows])\n"
        "o2 = np.zeros(len(ids) + 1, dtype=np.uint64); o2[1:] = np.cumsum([len(r) for r in rows])\n"
        "idx.import_layer(0, ids, o2, flat)\n"
        "orc2 = oracle_from_product(idx, vs, lv)\n"
        "for ef in (400, 512):\n"
        "    assert_search_equal(idx.search_batch(qs, 10, ef), orc2.search_batch(qs, 10, ef, nthreads=8), 'hub ef=%%d' %% ef)\n"
        "print('two-level ok')\n" % (ROOT, kind))
    env = dict(os.environ)
    if limit:
        env["HNSW_MI355X_VISITED_2L_LIMIT"] = limit
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "two-level ok" in out.stdout, out.stdout + out.stderr


@pytest.mark.parametrize("limit", ["", "400"])
def test_two_level_visited_set_of_the_generic_kernel(limit):
    """hx_search_kernel with eight / sixteen list registers (320 < ef <= 1024, every dimension and kind the lean kernels
    do not serve): 32 KiB of LDS table + the second level in HBM (round 4); with HNSW_MI355X_VISITED_2L_LIMIT the LDS level
    closes after 400 ids, so every query crosses over early.  Results and counters are the oracle's."""
    code = (
        "import sys, numpy as np; sys.path.insert(0, %r)\n"
        "import hnsw_rs_amd as H\n"
        "from oracle import oracle_py as O\n"
        "from tests.util import oracle_from_product, assert_search_equal\n"
        "for kind, d, n in ((H.VEC_QUANT8, 36, 40000), (H.VEC_F32, 256, 30000), (H.VEC_F32, 52, 30000), (H.VEC_QUANT8, 128, 30000)):\n"
        "    m = 16\n"
        "    vs = H.synth_rows(0, 0x5EED0001, 0, n, d); qs = H.synth_rows(0, 0x5EED0002, 0, 96, d)\n"
        "    lv = O.draw_levels(n, m, 9)\n"
        "    idx = H.HNSW.new(m, 32, d, kind); idx.insert_bulk_device(vs, 8, False, levels=lv)\n"
        "    idx.set_option('inline_rows', 0)\n"
        "    orc = oracle_from_product(idx, vs, lv)\n"
        "    for ef in (321, 400, 512, 700, 1024):\n"
        "        assert_search_equal(idx.search_batch(qs, 10, ef), orc.search_batch(qs, 10, ef, nthreads=8), 'generic two-level kind %%d d %%d ef %%d' %% (kind, d, ef))\n"
        "    g, w = idx.search_layer(0, qs[0], np.arange(5, dtype=np.uint32), 600), orc.search_layer(0, qs[0], np.arange(5, dtype=np.uint32), 600)\n"
        "    assert np.array_equal(g[0], w[0]) and np.array_equal(g[1].view(np.uint32), w[1].view(np.uint32)) and tuple(int(x) for x in g[2]) == tuple(int(x) for x in w[2]), 'search_layer seam'\n"
        "print('generic two-level ok')\n" % ROOT)
    env = dict(os.environ)
    if limit:
        env["HNSW_MI355X_VISITED_2L_LIMIT"] = limit
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and "generic two-level ok" in out.stdout, out.stdout + out.stderr


@pytest.mark.parametrize("kind", [H.VEC_QUANT8, H.VEC_F32])
def test_lean_kernels_with_the_four_register_list(kind):
    """129 <= ef <= 512 at d = 100: the lean kernels with four and eight interleaved list registers (round 3; the
    generic kernel served these before), top-n reaching into every register"""
    idx, orc, qs = device_built(20000, 100, 16, kind)
    for ef in (129, 160, 255, 256, 257, 300, 512):
        assert_search_equal(idx.search_batch(qs, 10, ef), orc.search_batch(qs, 10, ef, nthreads=8), "ef=%d" % ef)
    assert_search_equal(idx.search_batch(qs[:64], 250, 256), orc.search_batch(qs[:64], 250, 256, nthreads=8), "n=250")
    assert_search_equal(idx.search_batch(qs[:64], 300, 200), orc.search_batch(qs[:64], 300, 200, nthreads=8), "n > ef")
    assert_search_equal(idx.search_batch(qs[:64], 500, 512), orc.search_batch(qs[:64], 500, 512, nthreads=8), "n=500")


@pytest.mark.parametrize("kind", [H.VEC_QUANT8, H.VEC_F32])
def test_128d(kind):
    """configs[3] dimension"""
    idx, orc, qs = device_built(20000, 128, 16, kind)
    # f32: the lean kernel with the cooperative row gather (one list register up to ef 64, head + tail up to 128,
    # four interleaved registers up to 256, six / eight up to 384 / 512 with the two-level visited set: round 4),
    # the generic kernel beyond; n = 100 reaches into the tail register, n = 200 into the third and fourth of the
    # interleaved ones
    for ef in (1, 10, 64, 65, 100, 128, 129, 200, 256, 257, 300, 384, 385, 512, 513):
        assert_search_equal(idx.search_batch(qs, 10, ef), orc.search_batch(qs, 10, ef, nthreads=8), "128d ef=%d" % ef)
    assert_search_equal(idx.search_batch(qs[:64], 100, 120), orc.search_batch(qs[:64], 100, 120, nthreads=8), "128d n=100")
    assert_search_equal(idx.search_batch(qs[:64], 200, 230), orc.search_batch(qs[:64], 200, 230, nthreads=8), "128d n=200")


@pytest.mark.parametrize("kind", [H.VEC_QUANT8, H.VEC_F32])
def test_768d_unit_rows(kind):
    """configs[2]: 768d, unit-normalised rows, efSearch = 128"""
    idx, orc, qs = device_built(20000, 768, 16, kind, unit=True)
    for ef in (64, 128):
        assert_search_equal(idx.search_batch(qs[:64], 10, ef), orc.search_batch(qs[:64], 10, ef, nthreads=8),
                            "768d ef=%d" % ef)


@pytest.mark.parametrize("kind", [H.VEC_QUANT8, H.VEC_F32])
def test_device_build_at_256d(kind):
    """configs[4] dimension: the on-device build, judged by recall and graph invariants; the search on the
    graph it built is still the reference's search"""
    idx, orc, qs = device_built(20000, 256, 16, kind)
    assert idx.assert_param_compliance()
    truth, _ = idx.brute_force(qs, 10)
    got = idx.search_batch(qs, 10, 96)
    rec = sum(len(set(a) & set(b)) for a, b in zip(got[0].tolist(), truth.tolist())) / (len(qs) * 10)
    assert rec > 0.97, rec
    assert_search_equal(got, orc.search_batch(qs, 10, 96, nthreads=8), "256d")


def _sharded_worker(rank, world, port, outdir):
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import hnsw_rs_amd as HH
    n, d, m = 12000, 256, 16
    vs = HH.synth_rows(0, 0x5EED0001, 0, n, d)
    idx = HH.HNSW.new(m, 32, d)
    idx.insert_bulk_sharded(vs, 4, False, levels=HH.draw_levels(m, n))
    out = {}
    for layer in idx.iter_layers():
        ids, offs, nbrs = layer.csr()
        out["offs%d" % layer.level] = offs
        out["nbrs%d" % layer.level] = np.concatenate([np.sort(nbrs[int(offs[k]):int(offs[k + 1])]) for k in range(len(ids))])
    qs = HH.synth_rows(0, 0x5EED0002, 0, 64, d)
    truth, _ = idx.brute_force(qs, 10)
    got, _, _, _ = idx.search_batch(qs, 10, 96)
    out["recall"] = np.array([sum(len(set(a) & set(b)) for a, b in zip(got.tolist(), truth.tolist())) / 640.0])
    np.savez(os.path.join(outdir, "rank%d.npz" % rank), **out)
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_device_build_at_256d(tmp_path, monkeypatch):
    """configs[4]: points sharded over two ranks, edge records all-gathered per batch; both replicas end
    identical (two ranks on this one GPU, gloo rendezvous: only the exchange differs from an 8-GPU run).
    Phases 2 / 3 split by row ownership (the default) and run in full on every rank (HNSW_MI355X_SHARD_CONNECT=0)
    leave the same graph."""
    import socket
    import torch.multiprocessing as mp
    graphs = []
    for mode in ("1", "0"):
        monkeypatch.setenv("HNSW_MI355X_SHARD_CONNECT", mode)  # read once per process: the workers are new ones
        out = tmp_path / ("mode" + mode)
        out.mkdir()
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        mp.spawn(_sharded_worker, args=(2, port, str(out)), nprocs=2, join=True)
        r0, r1 = np.load(out / "rank0.npz"), np.load(out / "rank1.npz")
        for k in r0.files:
            assert np.array_equal(r0[k], r1[k]), (mode, k)
        assert r0["recall"][0] > 0.97
        graphs.append(r0)
    for k in graphs[0].files:
        assert np.array_equal(graphs[0][k], graphs[1][k]), k


@pytest.mark.parametrize("n,d", [(20000, 100), (9000, 128), (6000, 768), (5000, 36)])
def test_mfma_scan_finds_the_exact_neighbours(n, d):
    """hnsw_brute_force_fast (MFMA screen + exact re-rank) against the exact scan hnsw_brute_force: same ids,
    same distance bits on these sets (the fast scan is not exact by construction; this is the check the
    header promises)"""
    vs = H.synth_rows(0, 0x5EED0001, 0, n, d)
    qs = H.synth_rows(0, 0x5EED0002, 0, 100, d)
    idx = H.HNSW.new(8, 16, d, H.VEC_F32)
    idx.import_points(vs, np.zeros(n, dtype=np.uint8))
    idx.import_layer(0, np.arange(n, dtype=np.uint32), np.zeros(n + 1, dtype=np.uint64), np.zeros(0, dtype=np.uint32))
    idx.set_ep(0)
    want_ids, want_d = idx.brute_force(qs, 10)
    got_ids, got_d = idx.brute_force_fast(qs, 10)
    assert np.array_equal(got_ids, want_ids)
    assert np.array_equal(got_d.view(np.uint32), want_d.view(np.uint32))


def test_mfma_scan_rejects_what_it_does_not_serve():
    vs = H.synth_rows(0, 1, 0, 500, 50)
    idx = H.HNSW.new(8, 16, 50, H.VEC_F32)  # 50 is not a multiple of 4
    idx.import_points(vs, np.zeros(500, dtype=np.uint8))
    idx.import_layer(0, np.arange(500, dtype=np.uint32), np.zeros(501, dtype=np.uint64), np.zeros(0, dtype=np.uint32))
    idx.set_ep(0)
    with pytest.raises(H.HnswError):
        idx.brute_force_fast(vs[:4], 10)
    q8 = H.HNSW.new(8, 16, 100).insert_bulk(H.synth_rows(0, 1, 0, 300, 100), 1, False)
    with pytest.raises(H.HnswError):
        q8.brute_force_fast(H.synth_rows(0, 2, 0, 4, 100), 10)


@pytest.mark.parametrize("kind", [H.VEC_F32, H.VEC_QUANT8])
def test_lean_kernels_with_rows_longer_than_the_stride(kind):
    """d = 100 runs the lean kernels (search_lean.hip).  Nodes whose degree exceeds the 32-slot stride keep
    the rest of their row in the overflow CSR: the f32 kernel sends those rows through its pass body again
    (no runner-up in such a pass), the quant8 kernel walks them in place; a hub may also be the runner-up,
    which is then not speculated on (searcher.rs:45-71 sees one neighbour list either way)."""
    from tests.util import product_from_oracle, rand_vectors
    n, d, m = 3000, 100, 4  # layer-0 cap 8, stride 32
    vs, qs = rand_vectors(n, d, 31), rand_vectors(96, d, 32)
    lv = O.draw_levels(n, m, 5)
    orc = O.OracleHNSW(m, None, d, kind).insert_bulk(vs, lv)
    ids, offs, nbrs = orc.layer_csr(0)
    adj = {int(i): set(int(x) for x in nbrs[int(offs[k]):int(offs[k + 1])]) for k, i in enumerate(ids)}
    hubs = ((5, 70), (9, 40), (11, 33), (700, 150))
    for hub, cnt in hubs:
        for t in range(100, 100 + cnt):
            adj[hub].add(t)
            adj[t].add(hub)
    rows = [sorted(adj[int(i)]) for i in ids]
    orc2 = O.OracleHNSW(m, None, d, kind)
    orc2.import_points(vs, lv)
    orc2.import_layer(0, ids, np.cumsum([0] + [len(r) for r in rows]).astype(np.uint64),
                      np.concatenate([np.array(r, dtype=np.uint32) for r in rows]))
    for l in range(1, orc.nb_layers):
        orc2.import_layer(l, *orc.layer_csr(l))
    orc2.set_ep(orc.ep)
    index = product_from_oracle(orc2, vs, lv)
    assert index.get_layer(0).degree(700) > 128
    for i, (hub, _) in enumerate(hubs):  # queries that certainly expand the hubs
        qs[i] = vs[hub]
    for ef in (1, 10, 64, 68, 128):
        assert_search_equal(index.search_batch(qs, 10, ef), orc2.search_batch(qs, 10, ef),
                            "lean overflow kind=%d ef=%d" % (kind, ef))


@pytest.mark.parametrize("kind", [H.VEC_F32, H.VEC_QUANT8])
def test_one_large_launch_answers_like_many_small_ones(kind):
    """4096 queries in ONE launch (four waves per SIMD resident together, two per SIMD at a time for the f32
    kernel) against the oracle: occupancy must not change a result (each query owns its wave and its LDS)"""
    n, d, m = 24000, 100, 16
    vs = H.synth_rows(0, 0x5EED0001, 0, n, d)
    qs = H.synth_rows(0, 0x5EED0002, 0, 4096, d)
    lv = O.draw_levels(n, m, 0x5EED0003)
    idx = H.HNSW.new(m, 32, d, kind)
    idx.insert_bulk_device(vs, 8, False, levels=lv)
    orc = oracle_from_product(idx, vs, lv)
    for ef in (64, 68):
        assert_search_equal(idx.search_batch(qs, 10, ef), orc.search_batch(qs, 10, ef, nthreads=8),
                            "one launch of 4096, kind=%d ef=%d" % (kind, ef))


def test_device_build_batch_schedule_options():
    """hnsw_set_option "gpu_build_batch_max" / "gpu_build_batch_div": smaller insert batches (closer to the
    reference's one-at-a-time insert_bulk, template.rs:493-504) give a different, equally valid graph: every
    search on it is still the oracle's on that graph, and recall does not drop"""
    n, d, m = 20000, 100, 16
    vs = H.synth_rows(0, 0x5EED0001, 0, n, d)
    qs = H.synth_rows(0, 0x5EED0002, 0, 256, d)
    lv = O.draw_levels(n, m, 0x5EED0003)
    recalls = []
    for bmax, bdiv in ((8192, 8), (64, 64)):
        idx = H.HNSW.new(m, 32, d, H.VEC_F32)
        idx.set_option("gpu_build_batch_max", bmax)
        idx.set_option("gpu_build_batch_div", bdiv)
        idx.insert_bulk_device(vs, 8, False, levels=lv)
        orc = oracle_from_product(idx, vs, lv)
        got = idx.search_batch(qs, 10, 64)
        assert_search_equal(got, orc.search_batch(qs, 10, 64, nthreads=8), "schedule %d:%d" % (bmax, bdiv))
        truth, _ = idx.brute_force(qs, 10)
        recalls.append(sum(len(set(a.tolist()) & set(b.tolist())) for a, b in zip(got[0], truth)) / 2560.0)
    assert min(recalls) > 0.98, recalls
    with pytest.raises(H.HnswError):
        idx.set_option("gpu_build_batch_max", 0)


def test_device_build_with_batches_of_32768_points():
    """the largest batch the build takes ("gpu_build_batch_max" 32768, what bench.py --config 4 asks for): with
    div 2 a 100k-point build reaches it; the graph is valid, recall is that of the default schedule, the search is
    the oracle's, and the small-visited-table-first launches repeat their overflowing points instead of sending
    them to the CPU path (the build is still complete)"""
    n, d, m = 100000, 32, 16
    vs = H.synth_rows(0, 0x5EED0001, 0, n, d, 8)
    qs = H.synth_rows(0, 0x5EED0002, 0, 256, d)
    lv = O.draw_levels(n, m, 0x5EED0003)
    recalls = []
    for bmax, bdiv in ((8192, 8), (32768, 2)):
        idx = H.HNSW.new(m, 32, d, H.VEC_F32)
        idx.set_option("gpu_build_batch_max", bmax)
        idx.set_option("gpu_build_batch_div", bdiv)
        idx.insert_bulk_device(vs, 8, False, levels=lv)
        assert idx.len() == n and idx.assert_param_compliance()
        orc = oracle_from_product(idx, vs, lv)
        got = idx.search_batch(qs, 10, 64)
        assert_search_equal(got, orc.search_batch(qs, 10, 64, nthreads=8), "schedule %d:%d" % (bmax, bdiv))
        truth, _ = idx.brute_force(qs, 10)
        recalls.append(sum(len(set(a.tolist()) & set(b.tolist())) for a, b in zip(got[0], truth)) / 2560.0)
    assert recalls[1] > recalls[0] - 0.02 and min(recalls) > 0.9, recalls


@pytest.mark.parametrize("kind", [H.VEC_F32, H.VEC_QUANT8])
def test_full_size_properties_1m_x_100d(kind):
    """configs[1] at its full size (1M x 100d, M = 16, efSearch 64 / 68, batches of 1024): the oracle is checked
    at this size by bench.py's parity leg; here the properties that need no oracle.  (1) a result row is ascending in
    (distance, id) and holds n distinct ids; (2) every reported distance is bit-for-bit the distance kernel's for that
    (query, id) -- VecBase::dist2many, vectors/src/lib.rs:17-22; (3) the same batch twice gives the same bits;
    (4) a query answers the same alone as inside a batch of 1024 (one wave, one LDS block per query); (5) the exact
    nearest neighbour (exhaustive scan, template.rs:531-541) is in the top 10 for >= 98 % of the queries; (6) the
    snapshot the on-device build leaves in HBM (adjacency as built, the kept-last-edge rows patched with their
    overflow lists) answers bit for bit like a fresh upload of the host graph."""
    n_pts, d, m, n = 1_000_000, 100, 16, 10
    vs = H.synth_rows(0, 0x5EED0001, 0, n_pts, d, 16)
    qs = H.synth_rows(0, 0x5EED0002, 0, 1024, d, 8)
    idx = H.HNSW.new(m, 32, d, kind)
    idx.insert_bulk_device(vs, 16, False)
    assert idx.len() == n_pts
    for ef in (64, 68):
        ids, dists, counts, stats = idx.search_batch(qs, n, ef)
        assert (counts == n).all() and (np.asarray(stats)[:, 3] == 0).all()
        key = (dists.view(np.uint32).astype(np.uint64) << np.uint64(32)) | ids.astype(np.uint64)
        assert (np.diff(key.astype(np.int64), axis=1) > 0).all(), "rows ascending in (distance, id), ids distinct"
        for qi in (0, 17, 511, 1023):
            again = idx.distance_batch(qs[qi], ids[qi])
            assert np.array_equal(again.view(np.uint32), dists[qi].view(np.uint32)), "distance bits, query %d" % qi
        ids2, dists2, _, stats2 = idx.search_batch(qs, n, ef)
        assert np.array_equal(ids, ids2) and np.array_equal(dists.view(np.uint32), dists2.view(np.uint32))
        assert np.array_equal(np.asarray(stats), np.asarray(stats2))
        for qi in (3, 700):
            a_ids, a_d, _, a_st = idx.search_batch(qs[qi:qi + 1], n, ef)
            assert np.array_equal(a_ids[0], ids[qi]) and np.array_equal(a_d[0].view(np.uint32), dists[qi].view(np.uint32))
            assert np.array_equal(np.asarray(a_st)[0, :3], np.asarray(stats)[qi, :3])
    truth, _ = idx.brute_force(qs[:128], 1)
    got, _, _, _ = idx.search_batch(qs[:128], n, 68)
    hit = sum(int(truth[i, 0] in got[i]) for i in range(128))
    assert hit >= 126, hit
    kept = idx.search_batch(qs, n, 68)
    idx.set_option("inline_budget_mb", 65536)  # any layout option drops the snapshot: the next search uploads
    fresh = idx.search_batch(qs, n, 68)
    assert np.array_equal(kept[0], fresh[0]) and np.array_equal(kept[1].view(np.uint32), fresh[1].view(np.uint32))
    assert np.array_equal(np.asarray(kept[3]), np.asarray(fresh[3]))


@pytest.mark.parametrize("kind", [H.VEC_QUANT8, H.VEC_F32])
def test_cosine_option_is_l2_on_unit_vectors(kind):
    """"metric_cosine" = 1 on raw rows and raw queries answers exactly like a plain index over rows and queries
    normalised beforehand by the same arithmetic (the option is an extension: the reference has no cosine,
    vectors/src/lib.rs:10-27); the neighbours are the ones of largest cosine similarity"""
    from tests.test_host_build import _unit_rows
    n, d, m = 6000, 100, 16
    vs = H.synth_rows(0, 0x5EED0001, 0, n, d) * np.float32(2.5)
    qs = H.synth_rows(0, 0x5EED0002, 0, 128, d) * np.float32(0.3)
    lv = H.draw_levels(m, n)
    cos = H.HNSW.new(m, 32, d, kind)
    cos.set_option("metric_cosine", 1)
    cos.insert_bulk(vs, 1, False, levels=lv)  # one build thread: two builds of the same rows give the same graph
    ref = H.HNSW.new(m, 32, d, kind).insert_bulk(_unit_rows(vs), 1, False, levels=lv)
    uq = _unit_rows(qs)
    # the pin: the ORACLE (the reference's L2 arithmetic) over rows and queries normalised by the restated arithmetic
    # (numpy float32, one left-to-right sum of squares: tests/test_host_build.py::_unit_rows), holding the same graph
    orc = oracle_from_product(cos, _unit_rows(vs), lv)
    for ef in (16, 64):
        got = cos.search_batch(qs, 10, ef)
        assert_search_equal(got, orc.search_batch(uq, 10, ef), "cosine vs oracle on unit rows, ef=%d" % ef)
        assert_search_equal(got, ref.search_batch(uq, 10, ef), "cosine ef=%d" % ef)
    for i in range(4):  # the one-query entry (coalescer path) normalises too
        assert cos.ann_by_vector(qs[i], 10, 64) == [int(x) for x in orc.ann_by_vector(uq[i], 10, 64)]
    o_bf = orc.brute_force(uq[:8], 10)
    c_bf = cos.brute_force(qs[:8], 10)
    assert np.array_equal(c_bf[0], o_bf[0]) and np.array_equal(c_bf[1].view(np.uint32), o_bf[1].view(np.uint32))
    assert np.array_equal(c_bf[0], ref.brute_force(uq[:8], 10)[0])
    # a clone keeps the metric: it normalises its queries like the original
    assert_search_equal(cos.clone().search_batch(qs, 10, 64), orc.search_batch(uq, 10, 64), "clone of a cosine index")
    if kind == H.VEC_F32:
        # device-resident queries go through the same normalisation
        import torch
        dev = torch.device("cuda:0")
        dQ = torch.from_numpy(qs).to(dev)
        ids = torch.empty((128, 10), dtype=torch.int32, device=dev)
        dd = torch.empty((128, 10), dtype=torch.float32, device=dev)
        cnt = torch.empty(128, dtype=torch.int32, device=dev)
        st = torch.empty((128, 4), dtype=torch.int32, device=dev)
        cos.search_batch_device(dQ.data_ptr(), 128, 10, 64, ids.data_ptr(), dd.data_ptr(), cnt.data_ptr(), st.data_ptr(), 0)
        cos.search_batch_device_finish(dQ.data_ptr(), 128, 10, 64, ids.data_ptr(), dd.data_ptr(), cnt.data_ptr(), st.data_ptr(), 0)
        w = orc.search_batch(uq, 10, 64)  # the device-pointer entry against the oracle
        assert np.array_equal(ids.cpu().numpy().view(np.uint32), w[0])
        assert np.array_equal(dd.cpu().numpy().view(np.uint32), w[1].view(np.uint32))
        assert np.array_equal(st.cpu().numpy()[:, :3].astype(np.int64), np.asarray(w[3])[:, :3].astype(np.int64))
        assert torch.equal(dQ.cpu(), torch.from_numpy(qs))  # the caller's queries are left as they were
        # the exact top-10 by cosine similarity (float64) is what the exhaustive scan returns
        sims = (_unit_rows(vs).astype(np.float64) @ uq[:8].astype(np.float64).T).T
        best = np.argsort(-sims, axis=1)[:, :10]
        bf = cos.brute_force(qs[:8], 10)[0]
        assert np.mean([len(set(a.tolist()) & set(b.tolist())) for a, b in zip(bf, best)]) >= 9.9
