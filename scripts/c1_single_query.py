#!/usr/bin/env python3
"""BASELINE.json config 1: 10k nodes, dim 128, M=5, efCon=200, k=10, ONE query at a time
(the shape of a HNSW.SEARCH command): latency of hnsw_search (host buffers in and out) vs the CPU oracle."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from bench import draw_levels
from oracle import oracle
from redis_hnsw_amd import Index
N, dim, M, ef, k = 10000, 128, 5, 200, 10
V = np.random.default_rng(1).random((N, dim), dtype=np.float32)
Q = np.random.default_rng(2).random((200, dim), dtype=np.float32)
lv = draw_levels(N, M, 7)
o = oracle.OracleIndex(dim, M, ef); t = time.time(); o.add_batch(V, lv); t_cpu_build = time.time() - t
gi = Index("c1", dim, M, ef); t = time.time(); gi.add_batch(V, levels=lv, mode="exact"); t_gpu_build = time.time() - t
ok = True
for q in Q[:50]:
    a = gi.search_knn(q, k); ids, sims = o.search(q, k)
    ok &= [r.id for r in a] == ids.tolist() and np.array_equal(np.float32([r.sim for r in a]).view(np.uint32), sims.view(np.uint32))
for q in Q[:20]: gi.search_knn(q, k)
t = time.perf_counter()
for q in Q: gi.search_knn(q, k)
t_gpu = (time.perf_counter() - t) / len(Q)
t = time.perf_counter()
for q in Q: o.search(q, k)
t_cpu = (time.perf_counter() - t) / len(Q)
print("C1: exact build cpu %.1fs gpu %.1fs (identical graphs by the parity suite); single-query latency gpu %.0f us, cpu oracle %.0f us; results identical: %s" % (
    t_cpu_build, t_gpu_build, 1e6 * t_gpu, 1e6 * t_cpu, ok))
