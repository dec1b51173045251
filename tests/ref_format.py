"""An independent writer / reader of the reference's on-disk index format, straight from the format
spec in the reference sources (test infrastructure; shares no code with csrc/persist.cpp):

  <dir>/params     52 bytes, big-endian: m u64, mmax u64, mmax0 u64, ml f32, ef_cons u64, dim u64, ep u64
                   (hnsw/src/params.rs:78-88; `size()` says 58, 52 are written)
  <dir>/points     len u64, point_size u64, then per point: level u8 + vector
                   (points/src/points.rs:124-132, points/src/point.rs:57-75);
                   QuantVec = min f32, delta f32, dim codes (vectors/src/quant.rs:102-110),
                   FullVec = dim x f32 (vectors/src/full.rs:54-69); ids are implicit by position
  <dir>/layers/<n> level u8, nb_nodes u32, m u16, then per node: id u32 + m x u32 neighbour slots padded
                   with 0xFFFFFFFF (graph/src/graph.rs:168-222)
"""
import os
import struct

import numpy as np

SENTINEL = 0xFFFFFFFF


def write_index(path, m, mmax, mmax0, ml, ef_cons, dim, ep, levels, vectors, layers, layer_m):
    """vectors: ("quant8", mins[N], deltas[N], codes[N, dim]) or ("f32", rows[N, dim]);
    layers: list over layer number of dict node -> iterable of neighbour ids;
    layer_m: the m field of every layer file (must be >= the largest degree of that layer)."""
    os.makedirs(os.path.join(path, "layers"), exist_ok=True)
    with open(os.path.join(path, "params"), "wb") as f:
        f.write(struct.pack(">QQQfQQQ", m, mmax, mmax0, ml, ef_cons, dim, ep))
    n = len(levels)
    with open(os.path.join(path, "points"), "wb") as f:
        if vectors[0] == "quant8":
            _, mins, deltas, codes = vectors
            f.write(struct.pack(">QQ", n, 1 + 8 + dim))
            for i in range(n):
                f.write(struct.pack(">Bff", int(levels[i]), float(mins[i]), float(deltas[i])))
                f.write(bytes(bytearray(int(c) for c in codes[i])))
        else:
            _, rows = vectors
            f.write(struct.pack(">QQ", n, 1 + 4 * dim))
            for i in range(n):
                f.write(struct.pack(">B", int(levels[i])))
                f.write(struct.pack(">%df" % dim, *[float(x) for x in rows[i]]))
    for l, rows in enumerate(layers):
        with open(os.path.join(path, "layers", str(l)), "wb") as f:
            f.write(struct.pack(">BIH", l, len(rows), layer_m[l]))
            for node in sorted(rows):
                nb = sorted(int(x) for x in rows[node])
                assert len(nb) <= layer_m[l]
                f.write(struct.pack(">I", node))
                f.write(struct.pack(">%dI" % layer_m[l], *(nb + [SENTINEL] * (layer_m[l] - len(nb)))))


def read_index(path):
    """-> dict(params, levels, vectors, layers, layer_m) in the shapes write_index takes"""
    with open(os.path.join(path, "params"), "rb") as f:
        m, mmax, mmax0, ml, ef_cons, dim, ep = struct.unpack(">QQQfQQQ", f.read())
    with open(os.path.join(path, "points"), "rb") as f:
        data = f.read()
    n, psize = struct.unpack(">QQ", data[:16])
    levels = np.zeros(n, dtype=np.uint8)
    if psize == 9 + dim:
        mins, deltas, codes = np.zeros(n, np.float32), np.zeros(n, np.float32), np.zeros((n, dim), np.uint8)
        for i in range(n):
            o = 16 + i * psize
            levels[i], mins[i], deltas[i] = struct.unpack(">Bff", data[o:o + 9])
            codes[i] = np.frombuffer(data[o + 9:o + psize], dtype=np.uint8)
        vectors = ("quant8", mins, deltas, codes)
    else:
        assert psize == 1 + 4 * dim
        rows = np.zeros((n, dim), np.float32)
        for i in range(n):
            o = 16 + i * psize
            levels[i] = data[o]
            rows[i] = struct.unpack(">%df" % dim, data[o + 1:o + psize])
        vectors = ("f32", rows)
    layers, layer_m = [], []
    names = sorted(os.listdir(os.path.join(path, "layers")), key=int)
    for name in names:
        with open(os.path.join(path, "layers", name), "rb") as f:
            d = f.read()
        level, nb_nodes, lm = struct.unpack(">BIH", d[:7])
        assert level == len(layers)
        rows, o = {}, 7
        for _ in range(nb_nodes):
            node = struct.unpack(">I", d[o:o + 4])[0]
            slots = struct.unpack(">%dI" % lm, d[o + 4:o + 4 + 4 * lm])
            nb = []
            for s in slots:  # the reader stops at the first sentinel (graph.rs:197-211)
                if s == SENTINEL:
                    break
                nb.append(s)
            rows[node] = nb
            o += 4 + 4 * lm
        assert o == len(d)
        layers.append(rows)
        layer_m.append(lm)
    return dict(params=(m, mmax, mmax0, ml, ef_cons, dim, ep), levels=levels, vectors=vectors, layers=layers,
                layer_m=layer_m)
