"""Development aid: where the two waves of the two-wave search kernel (hnsw_search_duo.hpp) spend an expansion -- shader
clocks per phase (needs the -DHNSW_PHASE_TIMERS build: HNSW_MI355X_LIB=redis_hnsw_amd/lib/libhnsw_mi355x_prof.so).
C1 (10 k x 128, M = 5, one query per call) and a lone 1024-query launch on the 1 M reference graph."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import FIXTURES, draw_levels, load_graph_fixture  # noqa: E402
from oracle import oracle  # noqa: E402
from redis_hnsw_amd import Index, _capi  # noqa: E402

NAMES = ["walker: wait for the row", "walker: chunk set-up + vector requests", "walker: visited filter",
         "walker: wait for vectors + distances", "walker: WAIT FOR THE KEEPER", "walker: accept + choice + message + row request",
         "keeper: mark + ranks + scatter + first unexpanded", "keeper: waiting for a message"]
lib = _capi.load()


def report(ix, label, n_queries):
    out = (C.c_uint64 * 8)()
    lib.hnsw_debug_phase_cycles(ix._h, out)
    sc, _ = ix.counters()
    tot = sum(out[:6])
    print("%s: %.1f expansions/query, walker %.0f clocks/expansion, two-wave form %s" % (label, sc.n_expand / n_queries, tot / max(sc.n_expand, 1), ix.last_search_was_duo()))
    for i, nm in enumerate(NAMES):
        print("   %-52s %7.0f clocks/expansion" % (nm, out[i] / max(sc.n_expand, 1)))


dim, ef, k = 128, 200, 10
V = np.random.default_rng(1).random((1_000_000, dim), dtype=np.float32)
Q = np.random.default_rng(2).random((2048, dim), dtype=np.float32)
o1 = oracle.OracleIndex(dim, 5, ef)
o1.add_batch(V[:10000], draw_levels(10000, 5, 7))
g1 = Index("c1", dim, 5, ef)
g1.import_graph(o1.export())
for q in Q[:20]:
    g1.search_knn(q, k)
lib.hnsw_reset_counters(g1._h)
for q in Q[:200]:
    g1.search_knn(q, k)
report(g1, "C1, one query per call", 200)
if os.path.exists(FIXTURES[(1_000_000, dim, 16, ef)]):
    g, _ = load_graph_fixture(FIXTURES[(1_000_000, dim, 16, ef)], V)
    ix = Index("c2", dim, 16, ef)
    ix.import_graph(g)
    ix.search_batch(Q[:1024], k)
    lib.hnsw_reset_counters(ix._h)
    ix.search_batch(Q[:1024], k)
    report(ix, "C2, one lone 1024-query launch", 1024)
