"""(test infrastructure: uses the oracle's restatement; run from the repo root)
Design simulation: passes per query against the number of candidates speculated on per pass (replays
the reference traversal with oracle/restate_np.py on a 200k-point index)."""
import sys, time, pickle, os
sys.path.insert(0,'.')
import numpy as np
import hnsw_rs_amd as H
from oracle import restate_np as R
from sortedcontainers import SortedList
N=200000; d=100
vs=H.synth_rows(0,0x5EED0001,0,N,d,8); qs=H.synth_rows(0,0x5EED0002,0,60,d,1)
t=time.time(); idx=H.HNSW.new(16,32,d,H.VEC_F32).insert_bulk(vs,8,False); print('build',time.time()-t)
csr=[idx.get_layer(l).csr() for l in range(idx.nb_layers())]
ri=R.Index.from_csr(vs,1,csr,int(idx.params.ep))
def run(q,ef,K):
    point=ri.point(q)
    r=R.Results(); r.selected.add(R._key(ri.dists([ri.ep],point)[0],ri.ep))
    for l in range(len(ri.layers)-1,0,-1): R.search_layer(ri,r,ri.layers[l],point,1)
    layer=ri.layers[0]
    sel=r.selected; expanded=set(); visited=set(e[1] for e in sel)
    def unexp(k):
        out=[]
        for e in sel:
            if e[1] not in expanded:
                out.append(e)
                if len(out)==k: break
        return out
    def commit(c):
        expanded.add(c[1])
        for n in layer[c[1]]:
            n=int(n)
            if n in visited: continue
            visited.add(n)
            k=R._key(ri.dists([n],point)[0],n)
            if len(sel)<ef: sel.add(k)
            elif k<sel[-1]:
                sel.add(k); sel.pop(-1)
    passes=0; commits=0; hist=[0]*(K+1)
    while True:
        spec=unexp(K)
        if not spec: break
        passes+=1
        commit(spec[0]); commits+=1; done=1
        for i in range(1,len(spec)):
            nxt=unexp(1)
            if nxt and nxt[0]==spec[i]:
                commit(spec[i]); commits+=1; done+=1
            else: break
        hist[done]+=1
    return passes,commits,hist
for K in (1,2,3,4,6,8):
    P=C=0; Hh=np.zeros(K+1)
    for q in qs[:40]:
        p,c,h=run(q,68,K); P+=p; C+=c; Hh+=np.array(h)
    print('K=%d passes/query %.1f commits/query %.1f commits/pass %.2f  dist of commits per pass %s'%(K,P/40,C/40,C/P,np.round(Hh/Hh.sum(),3)[1:]))
