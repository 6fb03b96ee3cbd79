import sys, numpy as np
sys.path.insert(0, ".")
from oracle import oracle
from redis_hnsw_amd import Index
from tests.util import graphs_equal, make_data
oracle.build()
for (n, dim, m, ef) in [(1200, 768, 16, 400), (1200, 128, 32, 400), (1200, 256, 32, 400), (1200, 128, 32, 100), (1200, 768, 32, 100), (1200, 128, 24, 200)]:
    V = make_data(n, dim, seed=81)
    lv = oracle.draw_levels(n, m, 6)
    o = oracle.OracleIndex(dim, m, ef); o.add_batch(V, lv)
    gi = Index("r", dim, m, ef)
    gi.add_batch(V, levels=lv, mode="exact")
    res = []
    for v in (7, 100, 555, 3):
        d0 = len(o.neighbors(v, 0))
        o.delete(v); gi.delete_node("node%d" % v)
        ok, why = graphs_equal(o.export(), gi.export_graph())
        res.append((v, d0, ok))
    print((n, dim, m, ef), "build ok; deletes (node, deg0, equal):", res, flush=True)
