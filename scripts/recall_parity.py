#!/usr/bin/env python3
"""Recall of the fast (batched) GPU build against the exact (reference-order)
GPU build on the same data and levels.  Usage: recall_parity.py N [dim M ef]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from bench import draw_levels
from tests.util import brute_force_topk, recall_at_k
from redis_hnsw_amd import Index

N = int(sys.argv[1]); dim = int(sys.argv[2]) if len(sys.argv) > 2 else 128
M = int(sys.argv[3]) if len(sys.argv) > 3 else 16; ef = int(sys.argv[4]) if len(sys.argv) > 4 else 200
modes = sys.argv[5].split(",") if len(sys.argv) > 5 else ["fast", "exact"]
k = 10
if os.environ.get("HNSW_DATA") == "clustered":
    from bench import clustered
    centers = np.random.default_rng(3).random((64, dim), dtype=np.float32)
    V = clustered(N, dim, 3, centers)
    Q = clustered(512, dim, 4, centers)
else:
    V = np.random.default_rng(1).random((N, dim), dtype=np.float32)
    Q = np.random.default_rng(2).random((512, dim), dtype=np.float32)
lv = draw_levels(N, M, 7)
gt = brute_force_topk(V, Q, k)
for mode in modes:
    gi = Index("x", dim, M, ef)
    for kv in os.environ.get("HNSW_TUNE", "").split(","):
        if "=" in kv:
            a, b = kv.split("="); gi.set_tuning(a, int(b))
    t = time.time(); gi.add_batch(V, levels=lv, mode=mode); dt = time.time() - t
    ids, sims, n = gi.search_batch(Q, k)
    sc, ic = gi.counters()
    inf = gi.info()
    print("%s build: N=%d %.2fs (%.0f ins/s) recall@10=%.4f n_dist/q=%.0f maxdeg0=%d maxdegU=%d stride0=%d ins_ndist/insert=%.0f" % (
        mode, N, dt, N / dt, recall_at_k(ids, gt), sc.n_dist / len(Q), inf.max_degree0, inf.max_degree_upper, inf.stride0, ic.n_dist / N), flush=True)
    gi.close()
