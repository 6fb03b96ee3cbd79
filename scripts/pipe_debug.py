import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import FIXTURES, load_graph_fixture
from redis_hnsw_amd import Index
N, dim, M, ef, k = 1_000_000, 128, 16, 200, 10
V = np.random.default_rng(1).random((N, dim), dtype=np.float32)
Q = np.random.default_rng(2).random((8192, dim), dtype=np.float32)
g, _ = load_graph_fixture(FIXTURES[(N, dim, M, ef)], V)
ix = Index("dbg", dim, M, ef)
ix.import_graph(g)
ix.set_tuning("pipe_chunk", 1 << 20)
ref = ix.search_batch(Q, k)
print("single launch n_out ok:", np.all(ref[2] == k))
ix.set_tuning("pipe_chunk", 1024)
for t in range(4):
    got = ix.search_batch(Q, k)
    bad = np.nonzero(got[2] != k)[0]
    badid = np.nonzero((got[0] != ref[0]).any(1))[0]
    print("try", t, "bad n_out rows:", bad[:10], len(bad), "values", got[2][bad[:10]], "id mismatches", len(badid), badid[:10])
