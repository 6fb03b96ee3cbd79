"""Deletes on random small indexes, one-wave against four-wave commit kernels: the graph after every delete must be the
oracle's, and both kernels must give the same verdict on the speculative records (n_spec / n_fallback equal).  With the
-DHNSW_OCC_DEBUG library (HNSW_MI355X_LIB=.../libhnsw_mi355x_dbg.so, build.build_debug_library()) every record is also
recomputed and the validation's misses are counted: there must be none.
    python scripts/del_campaign.py [configs=16] [seed=1]"""
import sys, ctypes as C, numpy as np
sys.path.insert(0, ".")
from oracle import oracle
from redis_hnsw_amd import Index, _capi
from tests.util import graphs_equal, make_data
oracle.build()
lib = _capi.load()
lib.hnsw_debug_occ_ctl.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
ncfg = int(sys.argv[1]) if len(sys.argv) > 1 else 16
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
bad = 0
for c in range(ncfg):
    dim = int(rng.choice([128, 128, 256, 768, 768, 96]))
    m = int(rng.choice([5, 8, 16, 24, 32]))
    ef = int(rng.choice([40, 100, 200, 400]))
    n = int(rng.integers(600, 1500))
    V = make_data(n, dim, seed=100 + c)
    lv = oracle.draw_levels(n, m, 11 + c)
    victims = [int(x) for x in rng.choice(n, size=8, replace=False)]
    verdicts = {}
    for team in (1, 0):
        o = oracle.OracleIndex(dim, m, ef); o.add_batch(V, lv)
        gi = Index("c", dim, m, ef)
        gi.set_tuning("commit_team", team)
        gi.add_batch(V, levels=lv, mode="exact")
        ok, why = graphs_equal(o.export(), gi.export_graph())
        out = (C.c_uint64 * 18)()
        vs, misses = [], 0
        for v in victims:
            o.delete(v); gi.delete_node("node%d" % v)
            lib.hnsw_debug_occ_ctl(gi._h, out)
            vs.append((int(out[16]), int(out[17])))
            misses += int(out[9])
            ok2, why2 = graphs_equal(o.export(), gi.export_graph())
            if not ok2 and ok:
                ok, why = False, "after deleting %d: %s" % (v, why2)
        verdicts[team] = vs
        gi.close()
        if not ok or misses:
            bad += 1
        print((n, dim, m, ef), "team" if team else "solo", "graph equal:", ok, why, "validation misses:", misses, flush=True)
    same = verdicts[1] == verdicts[0]
    if not same:
        bad += 1
    print("   verdicts equal:", same, verdicts[1] if same else (verdicts[1], verdicts[0]), flush=True)
print("FAILURES:", bad)
