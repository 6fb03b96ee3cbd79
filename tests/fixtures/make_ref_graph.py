"""Build the reference-order graph of a bench configuration with the CPU oracle (TEST INFRASTRUCTURE)
and save it as a fixture: levels, enterpoint, per-layer CSR.  The vectors are not stored (bench.py
regenerates them from the seed).  One thread, the reference's serial insert order (core.rs:489-599).

    python tests/fixtures/make_ref_graph.py --nodes 1000000 --dim 128 --m 16 --ef 200 --out data/c2_ref_graph_1m.npz
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nodes", type=int, default=1_000_000)
    ap.add_argument("--dim", type=int, default=128)
    ap.add_argument("--m", type=int, default=16)
    ap.add_argument("--ef", type=int, default=200)
    ap.add_argument("--out", required=True)
    ap.add_argument("--checkpoint-every", type=int, default=0)
    a = ap.parse_args()
    from oracle import oracle
    oracle.build()
    V = np.random.default_rng(1).random((a.nodes, a.dim), dtype=np.float32)   # bench.py's base vectors
    lv = oracle.draw_levels(a.nodes, a.m, 7)                                   # bench.py's levels
    o = oracle.OracleIndex(a.dim, a.m, a.ef)
    t0 = time.time()
    for i in range(a.nodes):
        o.add(V[i], int(lv[i]))
        if i % 50000 == 0:
            print("%d nodes, %.0f s" % (i, time.time() - t0), flush=True)
    secs = time.time() - t0
    g = o.export()
    os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
    arrs = dict(levels=g["levels"].astype(np.uint8), enterpoint=np.int64(g["enterpoint"]), max_layer=np.int64(g["max_layer"]),
                nodes=np.int64(a.nodes), dim=np.int64(a.dim), m=np.int64(a.m), ef=np.int64(a.ef), build_seconds=np.float64(secs))
    for l, (rp, cl) in enumerate(zip(g["row_ptr"], g["col"])):
        arrs["deg%d" % l] = np.diff(rp.astype(np.int64)).astype(np.uint16)
        arrs["col%d" % l] = cl.astype(np.uint32)
    np.savez_compressed(a.out, **arrs)
    print("saved %s: %d nodes in %.0f s (%.0f inserts/s, one thread)" % (a.out, a.nodes, secs, a.nodes / secs))


if __name__ == "__main__":
    main()
