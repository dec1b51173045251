import sys, os
sys.path.insert(0, '.')
order = sys.argv[1]
import numpy as np
def maps():
    libs = set()
    for line in open('/proc/self/maps'):
        for k in ('libamdhip64', 'libhsa-runtime64', 'libhnsw_mi355x'):
            if k in line:
                libs.add(line.split()[-1])
    return sorted(libs)
if order == 'torch_first':
    import torch
    print('torch cuda', torch.cuda.is_available(), torch.cuda.device_count())
    x = torch.zeros(4, device='cuda:0'); print(x.sum().item())
import hnsw_rs_amd as H
print('hx devices', H.device_count())
vs = H.synth_rows(2, 1, 0, 500, 16)
idx = H.HNSW.new(8, None, 16).insert_bulk(vs, 1, False)
print(idx.ann_by_vector(vs[3], 3, 10))
if order != 'torch_first':
    import torch
    print('torch cuda', torch.cuda.is_available(), torch.cuda.device_count())
    x = torch.zeros(4, device='cuda:0'); print(x.sum().item())
print('\n'.join(maps()))
