"""The four-wave commit on the dim-768 variant (T = 24), which hnsw_tu_occteam.hip keeps off: deletes with it forced on
("commit_team" = 2) beside the oracle's.  With a library built with -DHNSW_OCC_DEBUG (HNSW_MI355X_LIB=...) the delete
kernels print what they decide per neighbour (DELTRACE / DELROW / TEAMDIFF lines): run once with 0 and once with 2 and diff."""
import sys, numpy as np
sys.path.insert(0, ".")
from oracle import oracle
from redis_hnsw_amd import Index
from tests.util import graphs_equal, make_data
oracle.build()
import ctypes as C
from redis_hnsw_amd import _capi
lib = _capi.load()
has_ctl = hasattr(lib, "hnsw_debug_occ_ctl")
if has_ctl:
    lib.hnsw_debug_occ_ctl.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
force = int(sys.argv[1]) if len(sys.argv) > 1 else 2
cases = [(1200, 768, 16, 400), (1200, 768, 32, 100), (1200, 128, 32, 400), (1200, 768, 16, 100), (1200, 512, 16, 400), (1200, 1024, 16, 400), (600, 768, 16, 400)]
if len(sys.argv) > 2:
    cases = cases[:int(sys.argv[2])]
for (n, dim, m, ef) in cases:
    V = make_data(n, dim, seed=81)
    lv = oracle.draw_levels(n, m, 6)
    o = oracle.OracleIndex(dim, m, ef); o.add_batch(V, lv)
    gi = Index("r", dim, m, ef)
    gi.set_tuning("commit_team", force)
    gi.add_batch(V, levels=lv, mode="exact")
    ok0, why0 = graphs_equal(o.export(), gi.export_graph())
    res = []
    shown = False
    for v in (7, 100, 555, 3):
        d0 = len(o.neighbors(v, 0))
        o.delete(v); gi.delete_node("node%d" % v)
        if has_ctl:
            out = (C.c_uint64 * 18)()
            lib.hnsw_debug_occ_ctl(gi._h, out)
            print("  del %d: hash %x records %x stale %x spec %d recomputed %d why %s MISSES %d needless %d; ctl n_spec %d n_fallback %d" %
                  (v, out[0], out[1], out[2], out[3], out[4], "%x" % out[8], out[9], out[10], out[16], out[17]), flush=True)
        ok, why = graphs_equal(o.export(), gi.export_graph())
        res.append((v, d0, ok, "" if ok else why))
        if not ok and "node" in why and not shown:
            shown = True
            bad = int(why.split("node")[1].split()[0]); l = int(why.split("layer")[1].split(":")[0])
            ga, gb = o.export(), gi.export_graph()
            ra = ga["col"][l][int(ga["row_ptr"][l][bad]):int(ga["row_ptr"][l][bad + 1])].tolist()
            rb = gb["col"][l][int(gb["row_ptr"][l][bad]):int(gb["row_ptr"][l][bad + 1])].tolist()
            print("  first mismatch after deleting %d: node %d layer %d\n   oracle %s\n   engine %s\n   only oracle %s only engine %s" %
                  (v, bad, l, ra, rb, sorted(set(ra) - set(rb)), sorted(set(rb) - set(ra))), flush=True)
    print((n, dim, m, ef), "build equal:", ok0, "; deletes (node, deg0, equal):", res, flush=True)
