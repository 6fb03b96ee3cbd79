#!/usr/bin/env python3
"""Writes tests/golden/transcribed_*.npz with tests/transcription/hnsw_transcription.py -- the reference's core.rs
transcribed into Python, which shares no code with the C oracle.  BUILD CONTAINER ONLY (pure Python: the 20 k case takes
about an hour); neither the test tiers nor the GPU box run it, they read the files it wrote.

    python tests/transcription/make_transcribed_golden.py [case ...]

Inputs are SURVEY.md section 8d's: vectors U[0,1)^dim from numpy default_rng(1), queries from default_rng(2), levels
floor(-ln U / ln M) from default_rng(7) with node 0 at level 0 (core.rs:393-405) -- regenerated from the seeds by the
tests, not stored.  Stored: every adjacency row of every layer in stored order, 64 queries' ids and similarity bits,
the work counters of the build and of the queries, the tie census (see the transcription's header), and the summary
figures SURVEY.md / BASELINE.md section 4 quote for the model shape (N = 20 k x 128, M = 16, ef = 200).
"""
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from hnsw_transcription import Index  # noqa: E402

GOLDEN = os.path.join(os.path.dirname(HERE), "golden")
CASES = {
    # name: (n, dim, m, ef, k, nq)
    "transcribed_3k_dim32": (3000, 32, 16, 200, 10, 64),
    "transcribed_20k_dim128": (20000, 128, 16, 200, 10, 64),
}


def draw_levels(n, m, seed=7):
    u = np.random.default_rng(seed).random(n)
    u = np.maximum(u, np.finfo(np.float64).tiny)
    lv = np.floor(-np.log(u) * (1.0 / np.log(float(m)))).astype(np.int64)
    lv[0] = 0
    return np.minimum(lv, 31).astype(np.int32)


def inputs(n, dim, m, nq):
    V = np.random.default_rng(1).random((n, dim), dtype=np.float32)
    Q = np.random.default_rng(2).random((nq, dim), dtype=np.float32)
    return V, Q, draw_levels(n, m, 7)


def run(name, metric="reference", ties="fifo"):
    n, dim, m, ef, k, nq = CASES[name]
    V, Q, lv = inputs(n, dim, m, nq)
    idx = Index(dim, m, ef, V, lv, metric=metric, ties=ties)
    t0 = time.time()
    for i in range(n):
        idx.add_node("node%d" % i, i)
        if i % 1000 == 999:
            print("  %s[%s,%s]: %d inserted, %.0f s" % (name, metric, ties, i + 1, time.time() - t0), flush=True)
    build_ties = dict(idx.ties)
    nodes = [idx.nodes["node%d" % i] for i in range(n)]
    levels = np.zeros(n, dtype=np.int32)
    for l, layer in enumerate(idx.layers):
        for x in layer:
            levels[x.idx] = l
    assert np.array_equal(levels, lv)
    out = dict(params=np.array([n, dim, m, ef, k, nq], dtype=np.int64), enterpoint=np.int64(idx.enterpoint.idx),
               max_layer=np.int64(idx.max_layer),
               insert_counters=np.array([idx.n_dist_insert, idx.n_ids_insert, idx.n_expand_insert], dtype=np.int64))
    for l in range(idx.max_layer + 1):
        rp = np.zeros(n + 1, dtype=np.uint64)
        col = []
        for i, x in enumerate(nodes):
            row = x.neighbors[l] if l < len(x.neighbors) else []
            col.extend(y.idx for y in row)
            rp[i + 1] = len(col)
        out["row_ptr_%d" % l] = rp
        out["col_%d" % l] = np.asarray(col, dtype=np.uint32)
    ids = np.full((nq, k), 0xFFFFFFFF, dtype=np.uint32)
    sims = np.full((nq, k), -np.inf, dtype=np.float32)
    n_out = np.zeros(nq, dtype=np.uint32)
    sc = np.zeros(3, dtype=np.int64)
    for qi in range(nq):
        res = idx.search_knn(Q[qi], k)
        n_out[qi] = len(res)
        for j, (sim, node) in enumerate(res):
            ids[qi, j] = node.idx
            sims[qi, j] = sim
        sc += (idx.n_dist, idx.n_ids, idx.n_expand)
    # brute-force recall@k (f64 distances: ground truth, not the reference's arithmetic)
    hit = 0
    for qi in range(nq):
        d = ((V.astype(np.float64) - Q[qi].astype(np.float64)) ** 2).sum(axis=1)
        hit += len(set(np.argsort(d, kind="stable")[:k].tolist()) & set(ids[qi, :n_out[qi]].tolist()))
    deg0 = np.diff(out["row_ptr_0"].astype(np.int64))
    degU = np.concatenate([np.diff(out["row_ptr_%d" % l].astype(np.int64)) for l in range(1, idx.max_layer + 1)] or [np.zeros(1, np.int64)])
    stats = dict(case=name, metric=metric, ties=ties, dist_per_insert=idx.n_dist_insert / (n - 1), dist_per_query=sc[0] / nq,
                 expansions_per_query=sc[2] / nq, recall_at_k=hit / (nq * k),
                 level_histogram=np.bincount(lv).tolist(), max_degree_layer0=int(deg0.max()),
                 nodes_over_m_max_0=int((deg0 > 2 * m).sum()), mean_degree_layer0=float(deg0.mean()),
                 max_degree_upper=int(degU.max()), upper_rows_over_m_max=int((degU > m).sum()),
                 ties_build=build_ties, ties_total=dict(idx.ties), seconds=round(time.time() - t0))
    out.update(ids=ids, sims_bits=sims.view(np.uint32), n_out=n_out, search_counters=sc,
               stats=np.frombuffer(json.dumps(stats).encode(), dtype=np.uint8))
    return out, stats


def main():
    """case[:ties][:f64] ...   ties = rust (default: std's BinaryHeap itself), total (the oracle's (sim, id) order), fifo"""
    for arg in (sys.argv[1:] or list(CASES)):
        parts = arg.split(":")
        name, metric, ties = parts[0], "reference", "rust"
        for p in parts[1:]:
            if p == "f64":
                metric = "f64"
            else:
                ties = p
        out, stats = run(name, metric, ties)
        print(json.dumps(stats), flush=True)
        if metric == "reference":
            fn = name + ("" if ties == "rust" else "_" + ties)
            np.savez_compressed(os.path.join(GOLDEN, fn + ".npz"), **out)
            print("wrote", fn, os.path.getsize(os.path.join(GOLDEN, fn + ".npz")), "bytes")
        else:   # the float64 model: summary figures only (profiles/), never a golden
            with open(os.path.join(os.path.dirname(os.path.dirname(HERE)), "profiles", "r4_%s_f64_model.json" % name), "w") as f:
                json.dump(stats, f, indent=1)


if __name__ == "__main__":
    main()
