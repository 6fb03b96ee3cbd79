"""The reference's work counters of the C2 build (TEST INFRASTRUCTURE): the CPU oracle's serial insert of bench.py's
seeded vectors and levels (the build tests/fixtures/make_ref_graph.py saves the graph of), with the cumulative number
of metric evaluations (core.rs:550, 621, 652, 711), neighbour ids scanned and expansions recorded at the prefix sizes
bench.py builds.  SURVEY 8d's algorithmic bytes per insert are n_dist x 4 dim + n_ids x 4 with THESE counts; the
engine's own counters are smaller (it skips select_neighbors' extension where it provably adds nothing).

    python tests/fixtures/make_ref_counters.py --out data/c2_ref_insert_counters.json      (~70 min, one core)
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
    a = ap.parse_args()
    from oracle import oracle
    oracle.build()
    V = np.random.default_rng(1).random((a.nodes, a.dim), dtype=np.float32)   # bench.py's base vectors
    lv = oracle.draw_levels(a.nodes, a.m, 7)                                   # bench.py's levels
    marks = sorted(int(x) for x in a.marks.split(",") if int(x) <= a.nodes)
    o = oracle.OracleIndex(a.dim, a.m, a.ef)
    out = dict(config=dict(nodes=a.nodes, dim=a.dim, M=a.m, ef=a.ef, vectors="default_rng(1).random", levels="draw_levels(seed 7)"),
               what="cumulative counters of the oracle's serial build after the first n inserts", prefixes={})
    t0 = time.time()
    for i in range(a.nodes):
        o.add(V[i], int(lv[i]))
        if i + 1 in marks:
            c = o.insert_counters()
            out["prefixes"][str(i + 1)] = dict(n_dist=int(c.n_dist), n_ids=int(c.n_ids), n_expand=int(c.n_expand),
                                               seconds=round(time.time() - t0, 1))
            print(i + 1, out["prefixes"][str(i + 1)], flush=True)
            json.dump(out, open(a.out, "w"), indent=1)
    print("done in %.0f s" % (time.time() - t0))


if __name__ == "__main__":
    main()
