#!/usr/bin/env python3
"""How much would a cache-friendlier node order buy?  Builds the C2 index, renumbers its nodes in
breadth-first order from the enterpoint (layer-0 graph), imports the renumbered graph into a second
index and times the same queries on both.  Renumbering changes nothing but tie-breaks and where a
node's vector and row live in HBM.  Usage: locality_probe.py [N]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bench import draw_levels
from redis_hnsw_amd import Index

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
dim, M, ef, k, B = 128, 16, 200, 10, 1024
V = np.random.default_rng(1).random((N, dim), dtype=np.float32)
Q = np.random.default_rng(2).random((4 * B, dim), dtype=np.float32)
a = Index("a", dim, M, ef)
a.add_batch(V, levels=draw_levels(N, M), mode="fast")
g = a.export_graph(with_vectors=False)

# breadth-first order over layer 0
rp, col = g["row_ptr"][0].astype(np.int64), g["col"][0].astype(np.int64)
order = np.full(N, -1, dtype=np.int64)      # order[new] = old
seen = np.zeros(N, dtype=bool)
frontier = np.array([int(g["enterpoint"])]); seen[frontier] = True
pos = 0
while pos < N:
    if len(frontier) == 0:
        rest = np.flatnonzero(~seen)
        if len(rest) == 0: break
        frontier = rest[:1]; seen[frontier] = True
    order[pos:pos + len(frontier)] = frontier; pos += len(frontier)
    nb = np.concatenate([col[rp[f]:rp[f + 1]] for f in frontier]) if len(frontier) < 2000 else \
        col[np.concatenate([np.arange(rp[f], rp[f + 1]) for f in frontier])]
    nb = np.unique(nb[~seen[nb]])
    seen[nb] = True
    frontier = nb
new_of_old = np.empty(N, dtype=np.int64); new_of_old[order] = np.arange(N)

def permute(g):
    out = dict(levels=np.asarray(g["levels"])[order], enterpoint=int(new_of_old[int(g["enterpoint"])]),
               max_layer=int(g["max_layer"]), row_ptr=[], col=[])
    for l in range(len(g["row_ptr"])):
        rp_l, col_l = g["row_ptr"][l].astype(np.int64), g["col"][l].astype(np.int64)
        deg = (rp_l[1:] - rp_l[:-1])[order]
        nrp = np.concatenate([[0], np.cumsum(deg)])
        idx = np.concatenate([np.arange(rp_l[o], rp_l[o + 1]) for o in order]) if False else None
        # gather rows in the new order (vectorised)
        starts = rp_l[:-1][order]
        take = np.repeat(starts - nrp[:-1], deg) + np.arange(nrp[-1])
        out["row_ptr"].append(nrp.astype(np.uint64)); out["col"].append(new_of_old[col_l[take]].astype(np.uint32))
    out["vectors"] = V[order]
    return out

b = Index("b", dim, M, ef)
b.import_graph(permute(g))
dev = torch.device("cuda", 0)
dQ = torch.from_numpy(Q).to(dev)
ids = torch.empty((B, k), dtype=torch.int32, device=dev); sims = torch.empty((B, k), dtype=torch.float32, device=dev)
nn = torch.empty(B, dtype=torch.int32, device=dev)
st = torch.cuda.current_stream()
def timeit(ix, name):
    def run(i):
        q = dQ[(i % 4) * B:(i % 4 + 1) * B]
        ix.search_batch_device(q.data_ptr(), B, k, ids.data_ptr(), sims.data_ptr(), nn.data_ptr(), st.cuda_stream)
    for i in range(3): run(i)
    torch.cuda.synchronize(); ix.reset_counters()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for i in range(20): run(i)
    e1.record(st); torch.cuda.synchronize()
    sc, _ = ix.counters()
    print("%-28s %.3f ms/batch  n_dist/q %.0f" % (name, e0.elapsed_time(e1) / 20, sc.n_dist / (20 * B)))
    return ids.cpu().numpy().copy()
ra = timeit(a, "insertion order")
rb = timeit(b, "breadth-first order")
same = np.mean([len(set(order[x].tolist()) & set(y.tolist())) / k for x, y in zip(rb, ra)])
print("result overlap after mapping ids back: %.4f" % same)
