import sys; sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np, hnsw_rs_amd as H
from oracle import oracle_py as O
n, d, m = 30000, 100, 16
vs = H.synth_rows(0, 0x5EED0001, 0, n, d); lv = O.draw_levels(n, m, 0x5EED0003)
for mode in (1, 2, 1, 2, 1, 2):
    idx = H.HNSW.new(16, 32, 100).insert_bulk(vs[:5000], 8, False, levels=lv[:5000])
    c0 = idx.assert_param_compliance()
    idx.set_option("gpu_build", mode)
    lv2 = np.minimum(lv[5000:12000], lv[:5000].max())
    idx.insert_bulk(vs[5000:12000], 8, False, levels=lv2)
    bad = []
    for layer in idx.iter_layers():
        ids, offs, nbrs = layer.csr()
        deg = np.diff(offs)
        lim = int(np.ceil((32 if layer.level == 0 else 16) * np.float32(1.1)))
        bad.append((layer.level, len(ids), int((deg == 0).sum()), int((deg > lim).sum()), int(deg.max())))
    print(mode, c0, idx.assert_param_compliance(), bad, flush=True)
