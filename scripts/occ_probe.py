#!/usr/bin/env python3
"""Exact-order parallel insert (hnsw_occ.hpp) against the oracle: identical graphs, rate, window statistics.
   python scripts/occ_probe.py N dim M ef [window] [check]"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from redis_hnsw_amd import Index, _capi
N, dim, M, ef = [int(x) for x in sys.argv[1:5]]
W = int(sys.argv[5]) if len(sys.argv) > 5 else 32
check = int(sys.argv[6]) if len(sys.argv) > 6 else 1
from oracle import oracle
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from tests.util import graphs_equal
V = np.random.default_rng(1).random((N, dim), dtype=np.float32)
lv = oracle.draw_levels(N, M, 7)
gi = Index("occ", dim, M, ef)
gi.set_tuning("occ_window", W)
if os.environ.get("PLAN_LEAN"): gi.set_tuning("plan_lean", int(os.environ["PLAN_LEAN"]))
if os.environ.get("COMMIT_PAR"): gi.set_tuning("commit_par", int(os.environ["COMMIT_PAR"]))
if os.environ.get("OCC_AHEAD"): gi.set_tuning("occ_ahead_x10", int(os.environ["OCC_AHEAD"]))
for kv in os.environ.get("TUNING", "").split(","):          # e.g. TUNING=occ_depth_x10=30,occ_front_max=12
    if kv:
        gi.set_tuning(kv.split("=")[0], int(kv.split("=")[1]))
t = time.time(); gi.add_batch(V, levels=lv, mode="exact"); dt = time.time() - t
lib = _capi.load()
out = (C.c_uint64 * 16)()
lib.hnsw_debug_occ.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
lib.hnsw_debug_occ(gi._h, out)
nc = max(out[0], 1)
print("N=%d dim=%d M=%d ef=%d W=%d: %.2f s = %.0f inserts/s; commits %d, spec shrinks %d, recomputed %d (%.1f%%), stale plans %d, deltas/commit %.1f, rounds %d (%.2f commits/round)" % (
    N, dim, M, ef, W, dt, N / dt, out[0], out[1], out[2], 100.0 * out[2] / max(out[1] + out[2], 1), out[3], out[4] / nc, out[5], out[0] / max(out[5], 1)), flush=True)
names = ["hash", "first check", "connect", "shrink checks", "apply", "recompute", "finish/other", "total"]
print("commit kernel, clocks per commit: " + ", ".join("%s %.0f" % (nm, out[6 + i] / nc) for i, nm in enumerate(names)), flush=True)
print("recomputed shrinks without a speculative record: %d; row-changed flags (validate+commit): %d" % (out[14], out[15]))
try:
    cz = (C.c_uint64 * 8)()
    lib.hnsw_debug_occ_causes.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
    lib.hnsw_debug_occ_causes(gi._h, cz)
    print("recomputed shrinks by first cause: no record %d, own row changed %d, pool not full %d, a relevant removal %d, a relevant addition %d" % tuple(cz[:5]))
except AttributeError:
    pass
try:
    pz = (C.c_uint64 * 21)()
    lib.hnsw_debug_occ_par.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
    lib.hnsw_debug_occ_par(gi._h, pz)
    if pz[0]:
        it = max(pz[5 + 6], 1)
        print("group commit kernel, us per iteration (workgroup 0): dry run %.1f, wait %.1f, validate %.1f, wait %.1f, apply %.1f, wait %.1f; %d iterations in %d launches (%.2f per launch)" % (
            tuple(pz[5 + i] / it / 100.0 for i in range(6)) + (it, pz[5 + 7], it / max(pz[5 + 7], 1))))
        nd = max(pz[1], 1)
        print("a dry run, us (mean of %d): hash + journal check %.1f, connect %.1f, record checks %.1f, row + speculative result %.1f, recompute %.1f, update_connections %.1f, finish %.1f; slowest of an iteration %.1f" % (
            (nd,) + tuple(pz[13 + i] / nd / 100.0 for i in range(7)) + (pz[20] / it / 100.0,)))
        print("parallel group commits: %d groups (%.2f per round, %.2f nodes per group), %d dry runs (%.2f per commit); groups closed by a stale link plan %d, "
              "a record used %d, a row rewritten %d" % (pz[0], pz[0] / max(out[5], 1), out[0] / pz[0], pz[1], pz[1] / nc, pz[2], pz[3], pz[4]))
except AttributeError:
    pass
if check:
    t = time.time(); o = oracle.OracleIndex(dim, M, ef); o.add_batch(V, lv); to = time.time() - t
    ok, why = graphs_equal(o.export(), gi.export_graph())
    print("oracle build %.1f s; graphs identical: %s %s" % (to, ok, why))
    sys.exit(0 if ok else 1)
