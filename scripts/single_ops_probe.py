"""HNSW.NODE.ADD / HNSW.NODE.DEL one call at a time on the 50 k reference-order prefix (the bench's single_add /
single_delete shape), for `rocprofv3 --kernel-trace --stats -- python scripts/single_ops_probe.py`."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from bench import FIXTURE_50K, draw_levels, load_graph_fixture
from redis_hnsw_amd import Index
NE, dim, M, ef = 50_000, 128, 16, 200
V = np.random.default_rng(1).random((1_000_000, dim), dtype=np.float32)[:NE]
g, _ = load_graph_fixture(FIXTURE_50K, V)
ix = Index("p", dim, M, ef)
ix.import_graph(g)
ev = np.random.default_rng(11).random((400, dim), dtype=np.float32)
el = draw_levels(400, M, 13)
for i in range(100):
    ix.add_node("w%d" % i, ev[i], level=int(el[i]))
t = time.time()
for i in range(100, 400):
    ix.add_node("s%d" % i, ev[i], level=int(el[i]))
print("single hnsw_add: %.3f ms per call" % (1e3 * (time.time() - t) / 300))
vic = [int(v) for v in np.random.default_rng(17).choice(NE, 300, replace=False)]
t = time.time()
for v in vic:
    ix.delete_node("node%d" % v)
print("single hnsw_delete: %.3f ms per call" % (1e3 * (time.time() - t) / 300))
