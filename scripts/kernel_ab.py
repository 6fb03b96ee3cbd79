"""Development aid: the search kernel's three regimes on the reference-order 1 M graph (C2) and on C1, with a
parity spot check against the oracle -- one line per figure, for A/B runs of kernel changes.
  lone     one 1024-query launch at a time            (one wave per SIMD: instruction-issue bound)
  steady   three 1024-query launches in flight        (the bench's headline shape)
  big      one 4096-query launch at a time
  c1       10 k x 128, M=5, one query per hnsw_search call (host buffers)"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch  # noqa: E402
from bench import FIXTURES, draw_levels, load_graph_fixture  # noqa: E402
from oracle import oracle  # noqa: E402
from redis_hnsw_amd import Index  # noqa: E402

N, dim, M, ef, k, B = 1_000_000, 128, 16, 200, 10, 1024
V = np.random.default_rng(1).random((N, dim), dtype=np.float32)
Q = np.random.default_rng(2).random((8 * B, dim), dtype=np.float32)
g, _ = load_graph_fixture(FIXTURES[(N, dim, M, ef)], V)
ix = Index("ab", dim, M, ef)
ix.import_graph(g)
for kv in os.environ.get("HNSW_TUNING", "").split(","):
    if kv:
        ix.set_tuning(kv.split("=")[0], int(kv.split("=")[1]))
dev = torch.device("cuda", 0)
dQ = torch.from_numpy(Q).to(dev)
S = int(os.environ.get("HNSW_STREAMS", "3"))
streams = [torch.cuda.Stream() for _ in range(S)]
outs = [(torch.empty((4 * B, k), dtype=torch.int32, device=dev), torch.empty((4 * B, k), dtype=torch.float32, device=dev),
         torch.empty((4 * B,), dtype=torch.int32, device=dev)) for _ in range(S)]


def launch(i, s_, nb=B):
    q = dQ[(i % 8) * B:(i % 8) * B + nb] if nb <= B else dQ[:nb]
    o_ = outs[s_]
    ix.search_batch_device(q.data_ptr(), nb, k, o_[0].data_ptr(), o_[1].data_ptr(), o_[2].data_ptr(), streams[s_].cuda_stream)


def timed(fn, reps):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(reps):
        fn(i)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


for i in range(60):
    launch(i, i % S)
torch.cuda.synchronize()
lone = min(timed(lambda i: launch(i, 0), 40) for _ in range(3))
steady = min(timed(lambda i: launch(i, i % S), 120) for _ in range(3))
big = min(timed(lambda i: launch(i, 0, 4 * B), 12) for _ in range(3))
print("lone   %.4f ms / launch  (%.2f M QPS)" % (1e3 * lone, B / lone / 1e6))
print("steady %.4f ms / step    (%.2f M QPS)" % (1e3 * steady, B / steady / 1e6))
print("big    %.4f ms / 4096    (%.2f M QPS)" % (1e3 * big, 4 * B / big / 1e6))
# parity spot check (bounded table in the steady shape)
o = oracle.OracleIndex.from_graph(dim, M, ef, g)
for i in range(3):
    launch(i, i % S)
torch.cuda.synchronize()
ok = True
for i in range(3):
    want = o.search_batch(Q[i * B + 7:i * B + 39], k, threads=8)
    got_i = outs[i][0][7:39].cpu().numpy().view(np.uint32)
    got_s = outs[i][1][7:39].cpu().numpy().view(np.uint32)
    ok = ok and np.array_equal(got_i, want[0]) and np.array_equal(got_s, want[1].view(np.uint32))
print("parity (96 queries, ids + sim bits):", ok)
# the lone launch's shape (two waves per query when "duo" is on): all 1024 answers and the work counters
ix.reset_counters()
launch(0, 0)
torch.cuda.synchronize()
was_duo = ix.last_search_was_duo()
sc, _ = ix.counters()
want = o.search_batch(Q[:B], k, threads=8)
ok1 = (np.array_equal(outs[0][0][:B].cpu().numpy().view(np.uint32), want[0])
       and np.array_equal(outs[0][1][:B].cpu().numpy().view(np.uint32), want[1].view(np.uint32))
       and np.array_equal(outs[0][2][:B].cpu().numpy().view(np.uint32), want[2]))
print("lone launch: two-wave form %s, 1024 answers identical: %s, counters (dist, ids, expand) %s oracle %s" % (
    was_duo, ok1, [sc.n_dist, sc.n_ids, sc.n_expand], [want[3].n_dist, want[3].n_ids, want[3].n_expand]))
# C1
n1, m1 = 10_000, 5
lv1 = draw_levels(n1, m1, 7)
o1 = oracle.OracleIndex(dim, m1, ef)
o1.add_batch(V[:n1], lv1)
g1 = Index("c1", dim, m1, ef)
g1.import_graph(o1.export())
for kv in os.environ.get("HNSW_TUNING", "").split(","):
    if kv:
        g1.set_tuning(kv.split("=")[0], int(kv.split("=")[1]))
same = all([r.id for r in g1.search_knn(q, k)] == o1.search(q, k)[0].tolist() for q in Q[:40])
for q in Q[:50]:
    g1.search_knn(q, k)
t0 = time.perf_counter()
for q in Q[:400]:
    g1.search_knn(q, k)
c1 = (time.perf_counter() - t0) / 400
print("c1     %.1f us / query, identical: %s, two-wave form: %s" % (1e6 * c1, same, g1.last_search_was_duo()))
g1.set_tuning("time_launches", 1)
ks = []
for q in Q[:100]:
    g1.search_knn(q, k)
    ks.append(g1.last_search_kernel_ms())
print("c1     kernel alone %.1f us (median of 100, HIP events around the launch)" % (1e3 * float(np.median(ks))))
