#!/usr/bin/env python3
"""Writes tests/golden/tiecase_rust_lattice_del.npz: the transcription in its "rust" tie mode (std::collections::BinaryHeap
restated, SimPair ordered by sim only) builds an index on TIE-HEAVY data -- distinct points of {0,1,2}^8 -- DELETES a fifth
of its nodes (delete_node, core.rs:414-475, 824-863: every neighbour re-selects with the node ignored), adds a few more, and
the graph is stored.  tests/test_golden_cpu.py requires the C oracle's hnsw_oracle_add_std_heap / hnsw_oracle_delete_std_heap
(their own restatement of the same heap) to reproduce it row for row in stored order; the GPU's tie mode is tested against
the oracle.  BUILD CONTAINER ONLY (a few seconds).

    python tests/transcription/make_rust_tie_del_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from hnsw_transcription import Index  # noqa: E402
from make_rust_tie_golden import lattice, levels  # noqa: E402

GOLDEN = os.path.join(os.path.dirname(HERE), "golden")
N0, N1, DIM, M, EF = 600, 80, 8, 6, 24


def main():
    V, lv = lattice(N0 + N1, 11), levels(N0 + N1, M, seed=13)
    idx = Index(DIM, M, EF, V, lv, ties="rust")
    for i in range(N0):
        idx.add_node("node%d" % i, i)
    rng = np.random.default_rng(5)
    victims = rng.choice(N0, size=N0 // 5, replace=False)
    ep0 = idx.enterpoint.idx
    if ep0 not in victims:
        victims[0] = ep0                                 # the enterpoint's re-election (smallest id, see the transcription) too
    for v in victims:
        idx.delete_node("node%d" % int(v))
    for i in range(N0, N0 + N1):                         # inserts into the graph the deletes left
        idx.add_node("node%d" % i, i)
    n = N0 + N1
    dead = np.zeros(n, dtype=np.uint8)
    dead[victims] = 1
    out = dict(params=np.array([N0, N1, DIM, M, EF], dtype=np.int64), V=V, levels=lv, victims=victims.astype(np.uint32), dead=dead,
               enterpoint=np.int64(idx.enterpoint.idx), max_layer=np.int64(idx.max_layer))
    live = {int(name[4:]): node for name, node in idx.nodes.items()}
    for l in range(idx.max_layer + 1):
        rp, col = np.zeros(n + 1, dtype=np.uint64), []
        for i in range(n):
            x = live.get(i)
            if x is not None and l < len(x.neighbors):
                col.extend(y.idx for y in x.neighbors[l])
            rp[i + 1] = len(col)
        out["row_ptr_%d" % l], out["col_%d" % l] = rp, np.asarray(col, dtype=np.uint32)
    np.savez_compressed(os.path.join(GOLDEN, "tiecase_rust_lattice_del.npz"), **out)
    print("wrote tiecase_rust_lattice_del.npz: %d + %d nodes, %d deleted, %d layers, enterpoint %d" % (
        N0, N1, len(victims), idx.max_layer + 1, idx.enterpoint.idx))


if __name__ == "__main__":
    main()
