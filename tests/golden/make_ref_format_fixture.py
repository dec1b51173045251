"""Writes tests/golden/ref_format_index/: a small index in the reference's on-disk format produced by the
independent Python writer tests/ref_format.py (not by the library's own writer), with one layer-0 row
longer than the layer's cap (the reference's writer corrupts such a file, graph.rs:172-178; a well-formed
file stores the widened row length in the m field).

    python tests/golden/make_ref_format_fixture.py
"""
import os
import shutil
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle import oracle_py as O  # noqa: E402
from tests import ref_format as RF  # noqa: E402
from tests.util import rand_vectors  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def build():
    n, d, m = 300, 20, 6
    vs = rand_vectors(n, d, 4242) * np.float32(3.0) - np.float32(1.0)
    lv = O.draw_levels(n, m, 11)
    orc = O.OracleHNSW(m, None, d, O.VEC_QUANT8).insert_bulk(vs, lv)
    mins, deltas, codes = np.zeros(n, np.float32), np.zeros(n, np.float32), np.zeros((n, d), np.uint8)
    for i in range(n):
        mins[i], deltas[i], codes[i], _ = orc.get_quant(i)
    layers = []
    for l in range(orc.nb_layers):
        ids, offs, nbrs = orc.layer_csr(l)
        layers.append({int(i): [int(x) for x in nbrs[int(offs[k]):int(offs[k + 1])]] for k, i in enumerate(ids)})
    for t in range(100, 120):  # node 7 gets 20 more symmetric edges on layer 0: degree far above 2 m
        if t not in layers[0][7]:
            layers[0][7].append(t)
            layers[0][t].append(7)
    layer_m = [max(2 * m if l == 0 else m, max(len(r) for r in rows.values())) for l, rows in enumerate(layers)]
    ml = float(O.default_ml(m))
    return dict(m=m, mmax=m, mmax0=2 * m, ml=ml, ef_cons=2 * m, dim=d, ep=orc.ep, levels=lv,
                vectors=("quant8", mins, deltas, codes), layers=layers, layer_m=layer_m), vs


if __name__ == "__main__":
    out = os.path.join(HERE, "ref_format_index")
    shutil.rmtree(out, ignore_errors=True)
    spec, vs = build()
    RF.write_index(out, **spec)
    np.save(os.path.join(HERE, "ref_format_vectors.npy"), vs)
    print("wrote", out, "layers", len(spec["layers"]), "layer m fields", spec["layer_m"])
