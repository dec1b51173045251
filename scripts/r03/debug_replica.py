import ctypes as C, sys
sys.path.insert(0, '.')
import numpy as np, torch
import hnsw_rs_amd as H
from hnsw_rs_amd import _lib
kind = int(sys.argv[1]) if len(sys.argv) > 1 else 0
n, d, m = 20000, 100, 16
vs = H.synth_rows(0, 0x5EED0001, 0, n, d); qs = H.synth_rows(0, 0x5EED0002, 0, 256, d)
src = H.HNSW.new(m, 32, d, kind).insert_bulk_device(vs, 4, False, levels=H.draw_levels(m, n))
L = _lib.lib(); desc = _lib.SnapshotDesc(); _lib.check(L.hnsw_snapshot_describe(src._h, C.byref(desc)))
print('bytes', list(desc.bytes), 'header', list(desc.header)[:18])
rep = H.HNSW.new(m, 32, d, kind); there = _lib.SnapshotDesc()
for i in range(7): there.bytes[i] = desc.bytes[i]
for i in range(32): there.header[i] = desc.header[i]
_lib.check(L.hnsw_snapshot_adopt(rep._h, C.byref(there)))
class Mem:
    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 2}
for i in range(7):
    if desc.bytes[i]:
        a = torch.as_tensor(Mem(int(desc.ptr[i]), int(desc.bytes[i])), device="cuda:0")
        b = torch.as_tensor(Mem(int(there.ptr[i]), int(there.bytes[i])), device="cuda:0")
        print(i, hex(a.data_ptr()), hex(int(desc.ptr[i])), hex(b.data_ptr()), hex(int(there.ptr[i])))
        b.copy_(a)
        torch.cuda.synchronize()
        b2 = torch.as_tensor(Mem(int(there.ptr[i]), int(there.bytes[i])), device="cuda:0")
        print('   equal after copy:', bool(torch.equal(a, b2)))
_lib.check(L.hnsw_snapshot_commit(rep._h))
for ef in (10, 64):
    g = rep.search_batch(qs, 10, ef); w = src.search_batch(qs, 10, ef)
    print('ef', ef, 'counts', g[2][:8], w[2][:8], 'ids equal', np.array_equal(g[0], w[0]), 'status', np.unique(np.asarray(g[3])[:, 3]))
    print(g[0][0], w[0][0])
