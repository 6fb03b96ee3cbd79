import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from bench import draw_levels
from redis_hnsw_amd import Index
N, dim, M, ef = 200_000, 128, 16, 200
V = np.random.default_rng(1).random((N + 2000, dim), dtype=np.float32)
lv = draw_levels(N + 2000, M)
ix = Index("p", dim, M, ef)
if os.environ.get("PLAN_LEAN"): ix.set_tuning("plan_lean", int(os.environ["PLAN_LEAN"]))
ix.add_batch(V[:N], levels=lv[:N], mode="fast")
t = time.time()
for i in range(N, N + 1000):
    ix.add_node("n%d" % i, V[i], level=int(lv[i]))
dt = time.time() - t
print("exact hnsw_add at %d nodes: %.3f ms per insert (%.0f inserts/s)" % (N, dt, 1000 / dt))
t = time.time()
ix.add_batch(V[N + 1000:N + 2000], levels=lv[N + 1000:N + 2000], mode="exact")
dt = time.time() - t
print("exact hnsw_add_batch(mode 0): %.3f ms per insert (%.0f inserts/s)" % (dt, 1000 / dt))
