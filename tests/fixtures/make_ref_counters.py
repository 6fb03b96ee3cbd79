"""The reference's work counters of the C2 build (TEST INFRASTRUCTURE): the CPU oracle's serial insert of bench.py's
seeded vectors and levels (the build tests/fixtures/make_ref_graph.py saves the graph of), with the cumulative number
of metric evaluations (core.rs:550, 621, 652, 711), neighbour ids scanned and expansions recorded at the prefix sizes
bench.py builds.  SURVEY 8d's algorithmic bytes per insert are n_dist x 4 dim + n_ids x 4 with THESE counts; the
engine's own counters are smaller (it skips select_neighbors' extension where it provably adds nothing).

    python tests/fixtures/make_ref_counters.py --out data/c2_ref_insert_counters.json --checkpoint /tmp/refctr.npz
(~70 min of one core; with --checkpoint the oracle's graph and the cumulative counters are saved every 50 k inserts and
an interrupted run continues from there: the oracle re-imports the rows in stored order, so the build goes on identically
-- the prefixes recorded before and after a resume agree with an uninterrupted run's)
"""
import argparse
import json
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
    ap.add_argument("--marks", default="20000,50000,100000,200000,500000,1000000")
    ap.add_argument("--out", required=True)
    ap.add_argument("--checkpoint", default=None, help="resume file (the oracle's graph + cumulative counters every 50 k inserts): "
                                                        "an interrupted run continues from it instead of starting over")
    a = ap.parse_args()
    from oracle import oracle
    oracle.build()
    V = np.random.default_rng(1).random((a.nodes, a.dim), dtype=np.float32)   # bench.py's base vectors
    lv = oracle.draw_levels(a.nodes, a.m, 7)                                   # bench.py's levels
    marks = sorted(int(x) for x in a.marks.split(",") if int(x) <= a.nodes)
    o = oracle.OracleIndex(a.dim, a.m, a.ef)
    start, base = 0, np.zeros(3, dtype=np.int64)
    if a.checkpoint and os.path.exists(a.checkpoint):
        z = np.load(a.checkpoint)
        start = int(z["n"])
        L = int(z["max_layer"]) + 1
        g = dict(vectors=V[:start], levels=z["levels"], enterpoint=int(z["enterpoint"]), max_layer=int(z["max_layer"]),
                 row_ptr=[z["rp%d" % l] for l in range(L)], col=[z["col%d" % l] for l in range(L)])
        o = oracle.OracleIndex.from_graph(a.dim, a.m, a.ef, g)      # rows in stored order: the build continues identically
        base = z["counters"].astype(np.int64)
        print("resumed at %d nodes" % start, flush=True)
    out = dict(config=dict(nodes=a.nodes, dim=a.dim, M=a.m, ef=a.ef, vectors="default_rng(1).random", levels="draw_levels(seed 7)"),
               what="cumulative counters of the oracle's serial build after the first n inserts", prefixes={})
    if os.path.exists(a.out):
        try:
            out["prefixes"].update(json.load(open(a.out))["prefixes"])
        except (ValueError, KeyError):
            pass
    t0 = time.time()
    for i in range(start, a.nodes):
        o.add(V[i], int(lv[i]))
        if a.checkpoint and (i + 1) % 50000 == 0:
            c = o.insert_counters()
            g = o.export()
            arrs = dict(n=np.int64(i + 1), levels=g["levels"], enterpoint=np.int64(g["enterpoint"]), max_layer=np.int64(g["max_layer"]),
                        counters=base + np.array([c.n_dist, c.n_ids, c.n_expand], dtype=np.int64))
            for l, (rp, cl) in enumerate(zip(g["row_ptr"], g["col"])):
                arrs["rp%d" % l] = rp
                arrs["col%d" % l] = cl
            np.savez(a.checkpoint + ".tmp.npz", **arrs)
            os.replace(a.checkpoint + ".tmp.npz", a.checkpoint)
        if i + 1 in marks:
            c = o.insert_counters()
            out["prefixes"][str(i + 1)] = dict(n_dist=int(base[0] + c.n_dist), n_ids=int(base[1] + c.n_ids), n_expand=int(base[2] + c.n_expand),
                                               seconds=round(time.time() - t0, 1))
            print(i + 1, out["prefixes"][str(i + 1)], flush=True)
            json.dump(out, open(a.out, "w"), indent=1)
    print("done in %.0f s" % (time.time() - t0))


if __name__ == "__main__":
    main()
