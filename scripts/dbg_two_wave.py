import sys; sys.path.insert(0,'.')
import numpy as np
import hnsw_rs_amd as H
from oracle import oracle_py as O
from tests.util import rand_vectors, oracle_from_product
d=int(sys.argv[1]) if len(sys.argv)>1 else 33
n,m=600,8
vs=rand_vectors(n,d,100+d)*np.float32(2)-np.float32(.5); qs=rand_vectors(40,d,200+d)*np.float32(2)-np.float32(.5)
lv=O.draw_levels(n,m,d)
idx=H.HNSW.new(m,None,d).insert_bulk(vs,1,False,levels=lv); orc=oracle_from_product(idx,vs,lv)
for ef in (1,17):
    g=idx.search_batch(qs,10,ef); w=orc.search_batch(qs,10,ef)
    print('ef',ef,'ids eq',np.array_equal(g[0],w[0]))
    gs=np.asarray(g[3])[:,:3]; ws=np.asarray(w[3]).astype(np.int64)
    bad=np.nonzero((gs!=ws).any(1))[0]
    print('bad queries',bad[:10]); 
    for b in bad[:5]: print(b,'gpu',gs[b],'oracle',ws[b])
