import sys; sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np, hnsw_rs_amd as H
from oracle import oracle_py as O
n, d, m = 12000, 100, 5
vs = H.synth_rows(0, 0xC0FFEE + m, 0, n, d); lv = O.draw_levels(n, m, 9)
for mode in (0, 1, 2):
    idx = H.HNSW.new(m, 48, d, 0)
    if mode == 0:
        idx.insert_bulk(vs, 8, False, levels=lv)
    else:
        idx.set_option("gpu_build", mode); idx.insert_bulk_device(vs, 8, True, levels=lv)
    bad = []
    for layer in idx.iter_layers():
        ids, offs, nbrs = layer.csr()
        deg = np.diff(offs)
        lim = int(np.ceil((2 * m if layer.level == 0 else m) * np.float32(1.1)))
        bad.append((layer.level, len(ids), int((deg == 0).sum()), int((deg > lim).sum()), int(deg.max())))
    print(mode, idx.assert_param_compliance(), bad, flush=True)
