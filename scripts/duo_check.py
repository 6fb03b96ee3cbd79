import sys, numpy as np
sys.path.insert(0,'/root/repo')
from oracle import oracle
from redis_hnsw_amd import Index
oracle.build()
def bits(a): return np.ascontiguousarray(a,dtype=np.float32).view(np.uint32)
for (n,dim,m,ef,k,nq) in [(2000,128,16,200,10,128),(300,128,16,200,10,16),(3000,128,5,200,10,64),(1500,128,16,40,10,64),(1200,128,24,300,20,32),(1,128,16,200,10,4),(2,128,16,200,10,4)]:
    V=np.random.default_rng(1).random((n,dim),dtype=np.float32); Q=np.random.default_rng(2).random((nq,dim),dtype=np.float32)
    lv=oracle.draw_levels(n,m,7); o=oracle.OracleIndex(dim,m,ef); o.add_batch(V,lv)
    ix=Index("t",dim,m,ef); ix.import_graph(o.export())
    want=o.search_batch(Q,k)
    for duo in (1,0):
        ix.set_tuning("duo",duo); ix.reset_counters()
        ids,sims,n_out=ix.search_batch(Q,k)
        sc,_=ix.counters()
        ok=np.array_equal(ids[n_out[:,None]>np.arange(k)[None,:]],want[0][want[2][:,None]>np.arange(k)[None,:]]) and np.array_equal(n_out,want[2]) and np.array_equal(bits(sims)[n_out[:,None]>np.arange(k)[None,:]],bits(want[1])[want[2][:,None]>np.arange(k)[None,:]])
        print((n,dim,m,ef,k,nq),'duo',duo,'used',ix.last_search_was_duo(),'ok',ok,[sc.n_dist,sc.n_ids,sc.n_expand],[want[3].n_dist,want[3].n_ids,want[3].n_expand], flush=True)
    # single query calls
    r=[ [x.id for x in ix.search_knn(q,k)]==o.search(q,k)[0].tolist() for q in Q[:8]]
    print('  single calls identical', all(r))
    ix.close()
