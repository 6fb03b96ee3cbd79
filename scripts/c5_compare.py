#!/usr/bin/env python3
"""BASELINE.json config 5 at full size: recall@10 of the GPU-built index (fast build) against the
index built in the reference's serial order (CPU oracle, one thread) on the same 1M x 128 data, and
the GPU search on both graphs.  Slow (the CPU build is ~30-40 minutes).  Usage: c5_compare.py [N]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from bench import draw_levels
from tests.util import brute_force_topk, recall_at_k
from oracle import oracle
from redis_hnsw_amd import Index

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
dim, M, ef, k, B = 128, 16, 200, 10, 1024
V = np.random.default_rng(1).random((N, dim), dtype=np.float32)
Q = np.random.default_rng(2).random((B, dim), dtype=np.float32)
lv = draw_levels(N, M, 7)
out = dict(N=N, dim=dim, M=M, ef=ef, k=k, queries=B)
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "c5_compare_%d.json" % N)
os.makedirs(os.path.dirname(OUT), exist_ok=True)


def save():
    json.dump(out, open(OUT, "w"), indent=1)
    print(json.dumps(out), flush=True)


gt = brute_force_topk(V, Q, k)

t = time.time(); gf = Index("fast", dim, M, ef); gf.add_batch(V, levels=lv, mode="fast"); out["gpu_fast_build_s"] = round(time.time() - t, 2)
ids, sims, _ = gf.search_batch(Q, k)
out["recall_gpu_fast_built"] = round(recall_at_k(ids, gt), 4)
save()

t = time.time(); o = oracle.OracleIndex(dim, M, ef); o.add_batch(V, lv); out["cpu_reference_order_build_s"] = round(time.time() - t, 1)
oids, osims, _, _ = o.search_batch(Q, k, threads=os.cpu_count())
out["recall_cpu_reference_order_built"] = round(recall_at_k(oids, gt), 4)
save()

g = o.export()
gc = Index("cpu-built", dim, M, ef); gc.import_graph(g)
ids2, sims2, _ = gc.search_batch(Q, k)
out["gpu_search_on_cpu_built_graph_identical"] = bool(np.array_equal(ids2, oids) and np.array_equal(sims2.view(np.uint32), osims.view(np.uint32)))
for name, gi in (("gpu_fast_built", gf), ("cpu_built", gc)):
    for _ in range(3): gi.search_batch(Q, k)
    ms = []
    for _ in range(10):
        gi.set_tuning("time_launches", 1); gi.search_batch(Q, k); ms.append(gi.last_search_kernel_ms())
    out["gpu_kernel_ms_on_%s_graph" % name] = round(float(np.mean(ms)), 4)
    out["gpu_qps_on_%s_graph" % name] = round(B / (float(np.mean(ms)) * 1e-3), 1)
    save()
deg = np.diff(g["row_ptr"][0].astype(np.int64)); out["cpu_built_layer0_degree_mean_max"] = [round(float(deg.mean()), 2), int(deg.max())]
save()
