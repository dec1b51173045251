"""How fast does a snapshot reach HBM?  16M x 256d f32 rows (16.4 GB) through DeviceIndex::upload's two pinned pieces.
   HNSW_MI355X_UPLOAD_PIECE_MB=<n> python scripts/upload_probe.py"""
import os, sys, time
sys.path.insert(0, ".")
import numpy as np
import hnsw_rs_amd as H
N, d = int(os.environ.get("N", "8000000")), 256
vs = H.synth_rows(0, 0x5EED0001, 0, N, d, 32)
idx = H.HNSW.new(16, 32, d, H.VEC_F32)
idx.import_points(vs, np.zeros(N, dtype=np.uint8))
ids = np.arange(N, dtype=np.uint32); offs = np.zeros(N + 1, dtype=np.uint64); nb = np.zeros(0, dtype=np.uint32)
idx.import_layer(0, ids, offs, nb)
for rep in range(3):
    idx.set_option("inline_rows", 0)  # drops the snapshot
    t = time.time(); idx.upload(); dt = time.time() - t
    print("piece %s MB: upload of %.1f GB in %.3f s = %.1f GB/s" % (os.environ.get("HNSW_MI355X_UPLOAD_PIECE_MB", "64"), idx.device_bytes() / 1e9, dt, idx.device_bytes() / 1e9 / dt), flush=True)
