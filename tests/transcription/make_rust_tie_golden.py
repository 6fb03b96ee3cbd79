#!/usr/bin/env python3
"""Writes tests/golden/tiecase_rust_lattice.npz: HNSW.SEARCH answers of the transcription in its "rust" tie mode
(std::collections::BinaryHeap restated, SimPair ordered by sim only) on TIE-HEAVY data -- distinct points of the
lattice {0,1,2}^8, so that equal similarities meet at the stop test, the accept test and inside the answers.
tests/test_golden_cpu.py requires the C oracle's hnsw_oracle_search_std_heap (its own restatement of the same heap)
to reproduce them bit for bit on the same graph.  BUILD CONTAINER ONLY (a few seconds).

    python tests/transcription/make_rust_tie_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from hnsw_transcription import Index  # noqa: E402

GOLDEN = os.path.join(os.path.dirname(HERE), "golden")
N, DIM, M, EF, K, NQ = 700, 8, 6, 24, 6, 48


def lattice(n, seed):
    """n DISTINCT points of {0,1,2}^DIM (no duplicate vectors: SURVEY 8c)"""
    codes = np.random.default_rng(seed).choice(3 ** DIM, size=n, replace=False)
    return np.stack([(codes // 3 ** j) % 3 for j in range(DIM)], axis=1).astype(np.float32)


def levels(n, m, seed=7):
    u = np.maximum(np.random.default_rng(seed).random(n), np.finfo(np.float64).tiny)
    lv = np.floor(-np.log(u) / np.log(float(m))).astype(np.int32)
    lv[0] = 0
    return lv


def main():
    V, Q, lv = lattice(N, 1), lattice(NQ, 2), levels(N, M)
    idx = Index(DIM, M, EF, V, lv, ties="rust")
    for i in range(N):
        idx.add_node("node%d" % i, i)
    nodes = [idx.nodes["node%d" % i] for i in range(N)]
    out = dict(params=np.array([N, DIM, M, EF, K, NQ], dtype=np.int64), enterpoint=np.int64(idx.enterpoint.idx),
               max_layer=np.int64(idx.max_layer), levels=lv)
    for l in range(idx.max_layer + 1):
        rp, col = np.zeros(N + 1, dtype=np.uint64), []
        for i, x in enumerate(nodes):
            col.extend(y.idx for y in (x.neighbors[l] if l < len(x.neighbors) else []))
            rp[i + 1] = len(col)
        out["row_ptr_%d" % l], out["col_%d" % l] = rp, np.asarray(col, dtype=np.uint32)
    ids = np.full((NQ, K), 0xFFFFFFFF, dtype=np.uint32)
    sims = np.zeros((NQ, K), dtype=np.float32)
    n_out = np.zeros(NQ, dtype=np.uint32)
    idx.ties = {k_: 0 for k_ in idx.ties}
    for qi in range(NQ):
        res = idx.search_knn(Q[qi], K)
        n_out[qi] = len(res)
        for j, (sim, node) in enumerate(res):
            ids[qi, j], sims[qi, j] = node.idx, sim
    out.update(ids=ids, sim_bits=sims.view(np.uint32), n_out=n_out,
               accept_ties=np.int64(idx.ties.get("accept_657", 0)))
    np.savez_compressed(os.path.join(GOLDEN, "tiecase_rust_lattice.npz"), **out)
    print("wrote tiecase_rust_lattice.npz: %d nodes, %d layers, accept-test ties met by the %d queries: %d" % (
        N, idx.max_layer + 1, NQ, idx.ties.get("accept_657", 0)))


if __name__ == "__main__":
    main()
