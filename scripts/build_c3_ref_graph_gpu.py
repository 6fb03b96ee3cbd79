#!/usr/bin/env python3
"""BASELINE config 3's shape (dim 768, M = 32, ef_construction = 400) as a REFERENCE-ORDER graph built ON THE GPU
(hnsw_add_batch mode 0: the windowed build, row for row the serial insert order of core.rs:489-599): the first 100 k nodes
are checked against the CPU oracle's serial build of the same vectors (data/c3_ref_graph_100k.npz, tests/fixtures/
make_ref_graph.py), then the build continues to N and the graph is saved in the fixtures' format.
    python scripts/build_c3_ref_graph_gpu.py [N=300000] [out=gpurun_out/c3_ref_graph_300k.npz]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
from bench import draw_levels, load_graph_fixture  # noqa: E402
from redis_hnsw_amd import Index  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 300_000
out = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "gpurun_out", "c3_ref_graph_%dk.npz" % (N // 1000))
dim, M, ef, P = 768, 32, 400, 100_000
V = np.random.default_rng(1).random((N, dim), dtype=np.float32)
lv = draw_levels(N, M, 7)
ix = Index("c3-build", dim, M, ef)
t0 = time.time()
ix.add_batch(V[:P], levels=lv[:P], mode="exact")
t1 = time.time() - t0
print("first %d nodes: %.1f s = %.0f inserts/s" % (P, t1, P / t1), flush=True)
fx = os.path.join(ROOT, "data", "c3_ref_graph_100k.npz")
same = None
if os.path.exists(fx) and N >= P:
    want, _ = load_graph_fixture(fx, V)
    got = ix.export_graph()
    same = (want["enterpoint"] == got["enterpoint"] and want["max_layer"] == got["max_layer"] and np.array_equal(want["levels"], got["levels"])
            and all(np.array_equal(a, b) for a, b in zip(want["row_ptr"], got["row_ptr"])) and all(np.array_equal(a, b) for a, b in zip(want["col"], got["col"])))
    print("the 100 k prefix equals the oracle's serial build row for row: %s" % same, flush=True)
    if not same:
        sys.exit(1)
step = 50_000
for a in range(P, N, step):
    b = min(N, a + step)
    ta = time.time()
    ix.add_batch(V[a:b], levels=lv[a:b], mode="exact")
    print("%d nodes, %.0f s (%.0f inserts/s over the last %d)" % (b, time.time() - t0, (b - a) / (time.time() - ta), b - a), flush=True)
secs = time.time() - t0
g = ix.export_graph()
tz = ix.tie_counters()
arrs = dict(levels=g["levels"].astype(np.uint8), enterpoint=np.int64(g["enterpoint"]), max_layer=np.int64(g["max_layer"]), nodes=np.int64(N),
            dim=np.int64(dim), m=np.int64(M), ef=np.int64(ef), build_seconds=np.float64(secs), built_by=np.bytes_(b"gpu windowed reference-order build"),
            prefix_100k_equals_oracle=np.int64(-1 if same is None else int(same)))
for l, (rp, cl) in enumerate(zip(g["row_ptr"], g["col"])):
    arrs["deg%d" % l] = np.diff(rp.astype(np.int64)).astype(np.uint16)
    arrs["col%d" % l] = cl.astype(np.uint32)
os.makedirs(os.path.dirname(os.path.abspath(out)), exist_ok=True)
np.savez_compressed(out, **arrs)
print("saved %s: %d nodes in %.0f s (%.0f inserts/s); max layer-0 degree %d; tie census of the build: %s" % (
    out, N, secs, N / secs, int(np.diff(g["row_ptr"][0].astype(np.int64)).max()), tz), flush=True)
