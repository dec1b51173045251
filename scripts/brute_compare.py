"""exact VALU scan against the MFMA scan (gpurun): time and agreement.  usage: N d nq"""
import sys, time
sys.path.insert(0, '.')
import numpy as np
import hnsw_rs_amd as H
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
d = int(sys.argv[2]) if len(sys.argv) > 2 else 100
nq = int(sys.argv[3]) if len(sys.argv) > 3 else 2048
vs = H.synth_rows(0, 0x5EED0001, 0, N, d, 16); qs = H.synth_rows(0, 0x5EED0002, 0, nq, d, 8)
idx = H.HNSW.new(16, 32, d, H.VEC_F32)
idx.import_points(vs, np.zeros(N, dtype=np.uint8))
idx.import_layer(0, np.arange(N, dtype=np.uint32), np.zeros(N + 1, dtype=np.uint64), np.zeros(0, dtype=np.uint32))
idx.set_ep(0); idx.upload()
idx.brute_force_fast(qs[:64], 10)
t = time.time(); fi, fd = idx.brute_force_fast(qs, 10); tf = time.time() - t
flop = 2.0 * N * d * nq
print('MFMA scan: %d x %dd, %d queries: %.3f s wall (%.1f TFLOP/s of dot products, host merge included)' % (N, d, nq, tf, flop / tf / 1e12), flush=True)
t = time.time(); ei, ed = idx.brute_force(qs, 10); te = time.time() - t
print('exact scan: %.3f s wall (%.1f TFLOP/s)' % (te, flop / te / 1e12))
print('ids identical for %.4f of the queries; distance bits identical: %s' % ((fi == ei).all(axis=1).mean(), np.array_equal(fd.view(np.uint32), ed.view(np.uint32))))
