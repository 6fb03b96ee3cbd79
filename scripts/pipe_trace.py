import os, sys, time
sys.path.insert(0, "/root/repo" if os.path.exists("/root/repo/bench.py") else ".")
import numpy as np
from bench import FIXTURES, load_graph_fixture
from redis_hnsw_amd import Index
N, dim, M, ef, k = 1_000_000, 128, 16, 200, 10
V = np.random.default_rng(1).random((N, dim), dtype=np.float32)
Q = np.random.default_rng(2).random((8192, dim), dtype=np.float32)
g, _ = load_graph_fixture(FIXTURES[(N, dim, M, ef)], V)
ix = Index("t", dim, M, ef); ix.import_graph(g)
for _ in range(3): ix.search_batch(Q, k)
ts = []
for _ in range(8):
    t = time.perf_counter(); ix.search_batch(Q, k); ts.append(time.perf_counter() - t)
print("8192 per call ms:", " ".join("%.3f" % (1e3 * x) for x in ts))
os.environ["HNSW_PIPE_TRACE"] = "1"
ix.search_batch(Q, k)
