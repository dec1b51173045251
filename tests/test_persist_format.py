"""f-2: HNSW::save / HNSW::load in the reference's byte format (hnsw/src/template.rs:43-131)
against an index written by an INDEPENDENT writer (tests/ref_format.py, straight from hnsw/src/params.rs:78-88,
points/src/points.rs:124-132, graph/src/graph.rs:213-222), including a row longer than the layer's cap."""
import filecmp
import os

import numpy as np
import pytest

import hnsw_rs_amd as H
from oracle import oracle_py as O
from tests import ref_format as RF
from tests.util import assert_search_equal, rand_vectors

FIXTURE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_format_index")


def csr_rows(layer):
    ids, offs, nbrs = layer.csr()
    return {int(i): sorted(int(x) for x in nbrs[int(offs[k]):int(offs[k + 1])]) for k, i in enumerate(ids)}


def test_load_reads_an_independently_written_index():
    want = RF.read_index(FIXTURE)
    idx = H.HNSW.load(FIXTURE)
    m, mmax, mmax0, ml, ef_cons, dim, ep = want["params"]
    p = idx.params
    assert (int(p.m), int(p.mmax), int(p.mmax0), int(p.ef_cons), int(p.dim), int(p.ep)) == (m, mmax, mmax0, ef_cons, dim, ep)
    assert np.float32(p.ml) == np.float32(ml) and idx.vec_kind == H.VEC_QUANT8
    _, mins, deltas, codes = want["vectors"]
    assert idx.len() == len(want["levels"])
    for i in range(0, idx.len(), 7):
        pt = idx.get_point(i)
        mn, dl, cd = pt.quant()
        assert pt.level == want["levels"][i] and mn == mins[i] and dl == deltas[i] and np.array_equal(cd, codes[i])
    assert idx.nb_layers() == len(want["layers"])
    for l, rows in enumerate(want["layers"]):
        assert csr_rows(idx.get_layer(l)) == {k: sorted(v) for k, v in rows.items()}
    assert idx.get_layer(0).degree(7) > 2 * m  # the row above the cap survived the round trip


def test_save_writes_what_the_independent_writer_writes(tmp_path):
    out = str(tmp_path / "saved")
    H.HNSW.load(FIXTURE).save(out)
    for name in ("params", "points"):
        assert filecmp.cmp(os.path.join(FIXTURE, name), os.path.join(out, name), shallow=False), name
    a, b = RF.read_index(FIXTURE), RF.read_index(out)
    assert a["layer_m"] == b["layer_m"]
    for ra, rb in zip(a["layers"], b["layers"]):
        assert {k: sorted(v) for k, v in ra.items()} == {k: sorted(v) for k, v in rb.items()}
    for l in range(len(a["layers"])):
        assert filecmp.cmp(os.path.join(FIXTURE, "layers", str(l)), os.path.join(out, "layers", str(l)), shallow=False)


def test_f32_kind_round_trip_through_the_independent_reader(tmp_path):
    n, d, m = 120, 9, 4
    vs = rand_vectors(n, d, 5)
    idx = H.HNSW.new(m, None, d, H.VEC_F32).insert_bulk(vs, 1, False, levels=O.draw_levels(n, m, 2))
    out = str(tmp_path / "f32")
    idx.save(out)
    got = RF.read_index(out)
    assert got["vectors"][0] == "f32" and np.array_equal(got["vectors"][1], vs)
    assert got["params"][5] == d and got["params"][6] == int(idx.params.ep)
    for l, rows in enumerate(got["layers"]):
        assert csr_rows(idx.get_layer(l)) == {k: sorted(v) for k, v in rows.items()}
    back = H.HNSW.load(out)
    assert back.vec_kind == H.VEC_F32 and back.len() == n


def test_load_rejects_damaged_files(tmp_path):
    import shutil
    bad = str(tmp_path / "bad")
    shutil.copytree(FIXTURE, bad)
    with open(os.path.join(bad, "points"), "r+b") as f:
        f.truncate(1000)
    with pytest.raises(H.HnswError):
        H.HNSW.load(bad)
    bad2 = str(tmp_path / "bad2")
    shutil.copytree(FIXTURE, bad2)
    with open(os.path.join(bad2, "layers", "0"), "r+b") as f:  # a neighbour id beyond the points
        f.seek(7 + 4)
        f.write((0x00FFFFFF).to_bytes(4, "big"))
    with pytest.raises(H.HnswError):
        H.HNSW.load(bad2)
    bad3 = str(tmp_path / "bad3")
    shutil.copytree(FIXTURE, bad3)
    with open(os.path.join(bad3, "params"), "r+b") as f:  # entry point beyond the points
        f.seek(44)
        f.write((10 ** 6).to_bytes(8, "big"))
    with pytest.raises(H.HnswError):
        H.HNSW.load(bad3)


def oracle_of(spec):
    m, _, _, _, ef_cons, dim, ep = spec["params"]
    _, mins, deltas, codes = spec["vectors"]
    orc = O.OracleHNSW(m, ef_cons, dim, O.VEC_QUANT8)
    orc.import_points_quant(codes, mins, deltas, spec["levels"])
    for l, rows in enumerate(spec["layers"]):
        ids = np.array(sorted(rows), dtype=np.uint32)
        lists = [sorted(rows[int(i)]) for i in ids]
        offs = np.cumsum([0] + [len(r) for r in lists]).astype(np.uint64)
        orc.import_layer(l, ids, offs, np.array([x for r in lists for x in r], dtype=np.uint32))
    orc.set_ep(ep)
    return orc


@pytest.mark.gpu
def test_gpu_search_on_a_loaded_reference_format_index():
    """the only route to a cross-implementation check: an index in the reference's format, searched on the GPU"""
    spec = RF.read_index(FIXTURE)
    idx = H.HNSW.load(FIXTURE)
    orc = oracle_of(spec)
    qs = rand_vectors(40, spec["params"][5], 99) * np.float32(3.0) - np.float32(1.0)
    for ef in (1, 10, 64):
        assert_search_equal(idx.search_batch(qs, 10, ef), orc.search_batch(qs, 10, ef), "loaded ef=%d" % ef)


@pytest.mark.gpu
@pytest.mark.parametrize("kind", [H.VEC_QUANT8, H.VEC_F32])
def test_gpu_build_save_load_search(tmp_path, kind):
    """build -> hnsw_save -> hnsw_load -> GPU search equals the oracle on the same index"""
    n, d, m = 6000, 100, 16
    vs = H.synth_rows(0, 0x5EED0001, 0, n, d)
    qs = H.synth_rows(0, 0x5EED0002, 0, 96, d)
    lv = O.draw_levels(n, m, 21)
    built = H.HNSW.new(m, 32, d, kind).insert_bulk(vs, 8, False, levels=lv)
    path = str(tmp_path / "idx")
    built.save(path)
    loaded = H.HNSW.load(path)
    orc = O.OracleHNSW(m, 32, d, kind)
    orc.import_points(vs, lv)
    for l in range(built.nb_layers()):
        orc.import_layer(l, *built.get_layer(l).csr())
    orc.set_ep(int(built.params.ep))
    for ef in (10, 68):
        want = orc.search_batch(qs, 10, ef, nthreads=8)
        assert_search_equal(loaded.search_batch(qs, 10, ef), want, "loaded kind=%d ef=%d" % (kind, ef))
        assert_search_equal(built.search_batch(qs, 10, ef), want, "built kind=%d ef=%d" % (kind, ef))
