#!/usr/bin/env python3
"""Development aid: the reference-order windowed build AT THE 1 M END STATE -- NS new vectors inserted (hnsw_add_batch
mode 0) into the imported 1 M x 128 reference-order fixture, with the window / group-commit statistics of the run.
   TUNING=key=val,... python scripts/occ_at_scale.py [NS]"""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
from bench import FIXTURES, draw_levels, load_graph_fixture  # noqa: E402
from redis_hnsw_amd import Index, _capi  # noqa: E402

N, dim, M, ef = int(os.environ.get("BASE_N", "1000000")), 128, 16, 200
NS = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
V = np.random.default_rng(1).random((1_000_000, dim), dtype=np.float32)[:N]
g, _ = load_graph_fixture(FIXTURES[(N, dim, M, ef)] if N == 1_000_000 else os.path.join(ROOT, "data", "c2_ref_graph_%dk.npz" % (N // 1000)), V)
newV = np.random.default_rng(21).random((NS, dim), dtype=np.float32)
newL = draw_levels(NS, M, 23)
lib = _capi.load()
lib.hnsw_debug_occ.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
lib.hnsw_debug_occ_par.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
for tun in os.environ.get("TUNINGS", os.environ.get("TUNING", "")).split(";"):
    ix = Index("at-scale", dim, M, ef)
    ix.import_graph(g)
    for kv in tun.split(","):
        if kv:
            ix.set_tuning(kv.split("=")[0], int(kv.split("=")[1]))
    ix.add_batch(newV[:256], levels=newL[:256], mode="exact")      # warm-up (allocations, first launches)
    t = time.time()
    ix.add_batch(newV[256:], levels=newL[256:], mode="exact")
    dt = time.time() - t
    out, pz = (C.c_uint64 * 16)(), (C.c_uint64 * 21)()
    lib.hnsw_debug_occ(ix._h, out)
    lib.hnsw_debug_occ_par(ix._h, pz)
    n = NS - 256
    print(("[%s] %d inserts at " + str(N) + " nodes: %.3f s = %.0f inserts/s; rounds %d (%.2f commits/round, %.1f us/round), stale plans %d, recomputed shrinks %.1f%%") % (
        tun, n, dt, n / dt, out[5], out[0] / max(out[5], 1), 1e6 * dt / max(out[5], 1), out[3], 100.0 * out[2] / max(out[1] + out[2], 1)))
    if pz[0]:
        it = max(pz[5 + 6], 1)
        nd = max(pz[1], 1)
        print("   groups %d (%.2f per round, %.2f nodes each), dry runs %.2f per commit, %.2f iterations per launch; per iteration us: dry %.1f wait %.1f validate %.1f wait %.1f apply %.1f wait %.1f; "
              "mean dry run %.1f us (checks %.1f recompute %.1f update %.1f), slowest of an iteration %.1f; closed by link %d rec %d row %d" % (
                  pz[0], pz[0] / max(out[5], 1), out[0] / pz[0], pz[1] / max(out[0], 1), it / max(pz[5 + 7], 1),
                  pz[5] / it / 100.0, pz[6] / it / 100.0, pz[7] / it / 100.0, pz[8] / it / 100.0, pz[9] / it / 100.0, pz[10] / it / 100.0,
                  sum(pz[13 + i] for i in range(7)) / nd / 100.0, pz[15] / nd / 100.0, pz[17] / nd / 100.0, pz[18] / nd / 100.0, pz[20] / it / 100.0,
                  pz[2], pz[3], pz[4]), flush=True)
    try:
        lib.hnsw_debug_occ_par2.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
        p2 = (C.c_uint64 * 4)()
        lib.hnsw_debug_occ_par2(ix._h, p2)
        print("   rounds ended right after a group: %d" % p2[0])
    except AttributeError:
        pass
    ix.close()
