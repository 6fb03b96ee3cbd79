"""Development aid: where the time of hnsw_search_batch (host buffers) goes on the reference-order 1 M graph,
for several chunkings.  HNSW_PIPE_TRACE=1 prints the per-chunk timeline of one call."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import FIXTURES, load_graph_fixture  # noqa: E402
from redis_hnsw_amd import Index  # noqa: E402

N, dim, M, ef, k = 1_000_000, 128, 16, 200, 10
V = np.random.default_rng(1).random((N, dim), dtype=np.float32)
Q = np.random.default_rng(2).random((16384, dim), dtype=np.float32)
g, _ = load_graph_fixture(FIXTURES[(N, dim, M, ef)], V)
ix = Index("probe", dim, M, ef)
ix.import_graph(g)
print(ix.pipeline_info())
for B in (1024, 2048, 4096, 8192, 16384):
    for chunk in (1024, 2048, 1 << 20):
        ix.set_tuning("pipe_chunk", chunk)
        ix.search_batch(Q[:B], k)
        t0 = time.perf_counter()
        reps = 4
        for _ in range(reps):
            ix.search_batch(Q[:B], k)
        dt = (time.perf_counter() - t0) / reps
        print("B=%5d chunk=%7d  %.3f ms  %.2f M QPS" % (B, chunk, 1e3 * dt, B / dt / 1e6), flush=True)
if os.environ.get("HNSW_PIPE_TRACE"):
    ix.set_tuning("pipe_chunk", 1024)
    ix.search_batch(Q[:8192], k)
