"""HNSW.NODE.ADD one node per call: the serial kernels (hnsw_add) against a one-node window (hnsw_add_batch with
occ_min_batch = 1: plan, speculative shrinks in parallel, validated commit).  python scripts/single_add_probe.py [N]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from bench import draw_levels
from redis_hnsw_amd import Index
N = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000
dim, M, ef = 128, 16, 200
V = np.random.default_rng(1).random((N + 1000, dim), dtype=np.float32)
lv = draw_levels(N + 1000, M)
ix = Index("p", dim, M, ef)
ix.add_batch(V[:N], levels=lv[:N], mode="fast")
t = time.time()
for i in range(N, N + 500):
    ix.add_node("n%d" % i, V[i], level=int(lv[i]))
dt = (time.time() - t) / 500
print("hnsw_add (serial kernels) at %d nodes: %.3f ms per insert" % (N, 1e3 * dt))
ix.set_tuning("occ_min_batch", 1)
t = time.time()
for i in range(N + 500, N + 1000):
    ix.add_batch(V[i:i + 1], levels=lv[i:i + 1], mode="exact")
dt = (time.time() - t) / 500
print("hnsw_add_batch of one node through the window: %.3f ms per insert" % (1e3 * dt))
# HNSW.NODE.DEL one node per call (k_delete_exact: the neighbours' re-selections one after the other on one wave)
rng = np.random.default_rng(5)
victims = rng.choice(N, 300, replace=False)
t = time.time()
for v in victims:
    ix.delete_node("node%d" % int(v))
dt = (time.time() - t) / len(victims)
print("hnsw_delete at %d nodes (speculative re-selections, fast-built graph: full rows + inbound sweep): %.3f ms per delete" % (N, 1e3 * dt))
ix.set_tuning("single_window", 0)
victims = [int(v) for v in rng.choice(N, 400, replace=False) if ix._ids.get("node%d" % int(v)) is not None][:300]
t = time.time()
n_del = 0
for v in victims:
    try:
        ix.delete_node("node%d" % v); n_del += 1
    except Exception:
        pass
dt = (time.time() - t) / max(n_del, 1)
print("hnsw_delete, one-wave kernel: %.3f ms per delete" % (1e3 * dt))
