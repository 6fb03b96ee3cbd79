#!/usr/bin/env python3
"""Search-kernel sweep on one GPU: one index, several (tuning, batch) variants, ms per launch from
HIP events on the launch stream plus the engine's work counters.

  python scripts/search_sweep.py --graph fast --nodes 1000000 --variants "B=1024;B=2048,waves_per_cu=8,tag_bb=10"
  python scripts/search_sweep.py --graph data/ref_graph_200k.npz ...

--graph fast builds with the batched GPU build; a .npz path imports a reference-order graph written
by tests/fixtures/make_ref_graph.py (vectors are regenerated from the bench seed)."""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def load_graph_npz(path, V):
    z = np.load(path)
    n = int(z["nodes"])
    L = int(z["max_layer"]) + 1
    row_ptr, col = [], []
    for l in range(L):
        deg = z["deg%d" % l].astype(np.uint64)
        rp = np.zeros(n + 1, dtype=np.uint64)
        np.cumsum(deg, out=rp[1:])
        row_ptr.append(rp)
        col.append(z["col%d" % l].astype(np.uint32))
    return dict(vectors=V[:n], levels=z["levels"].astype(np.uint32), enterpoint=int(z["enterpoint"]),
                max_layer=int(z["max_layer"]), row_ptr=row_ptr, col=col), n


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--graph", default="fast")
    ap.add_argument("--nodes", type=int, default=1_000_000)
    ap.add_argument("--dim", type=int, default=128)
    ap.add_argument("--m", type=int, default=16)
    ap.add_argument("--ef", type=int, default=200)
    ap.add_argument("--k", type=int, default=10)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--variants", default="B=1024")
    a = ap.parse_args()
    import torch
    from bench import draw_levels
    from redis_hnsw_amd import Index
    N, dim, M, ef, k = a.nodes, a.dim, a.m, a.ef, a.k
    V = np.random.default_rng(1).random((N, dim), dtype=np.float32)
    gi = Index("sweep", dim, M, ef)
    t = time.time()
    if a.graph == "fast":
        gi.add_batch(V, levels=draw_levels(N, M), mode="fast")
    else:
        g, N = load_graph_npz(a.graph, V)
        gi.import_graph(g)
    print("graph %s: %d nodes ready in %.1f s; max deg0 %d" % (a.graph, N, time.time() - t, gi.info().max_degree0), flush=True)
    dev = torch.device("cuda", 0)
    maxB = 8192
    Q = torch.from_numpy(np.random.default_rng(2).random((2 * maxB, dim), dtype=np.float32)).to(dev)
    st = torch.cuda.current_stream()
    base = None
    for var in a.variants.split(";"):
        kv = dict(x.split("=") for x in var.split(",") if "=" in x)
        B = int(kv.pop("B", 1024))
        nstreams = int(kv.pop("streams", 1))
        if int(kv.pop("bf16", 0)):
            gi.set_tuning("compress_bf16", 1)        # one way: later variants run on the bf16 copy too
            base = None
        if int(kv.pop("fp8", 0)):
            gi.set_tuning("compress_fp8", 1)         # likewise
            base = None
        # defaults first, then the variant's knobs
        for key, val in dict(waves_per_cu=8, tag_bb=-1, tag_table=1, grid=-1, visited_bounded=1, launch_concurrency=1, lean=1, query_in_lds=0).items():
            gi.set_tuning(key, val)
        for key, val in kv.items():
            gi.set_tuning(key, int(val))
        ids = torch.empty((B, k), dtype=torch.int32, device=dev)
        sims = torch.empty((B, k), dtype=torch.float32, device=dev)
        nn = torch.empty(B, dtype=torch.int32, device=dev)

        streams = [st] + [torch.cuda.Stream() for _ in range(nstreams - 1)]
        outs = [(ids, sims, nn)] + [(torch.empty_like(ids), torch.empty_like(sims), torch.empty_like(nn)) for _ in range(nstreams - 1)]

        def run(i):
            o = (i % 2) * B
            s_ = streams[i % nstreams]
            a_, b_, c_ = outs[i % nstreams]
            gi.search_batch_device(Q[o:o + B].data_ptr(), B, k, a_.data_ptr(), b_.data_ptr(), c_.data_ptr(), s_.cuda_stream)
        for i in range(3):
            run(i)
        torch.cuda.synchronize()
        gi.reset_counters()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record(st)
        for i in range(a.reps):
            run(i)
        e1.record(st)
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / a.reps if nstreams == 1 else (time.perf_counter() - t0) * 1e3 / a.reps
        sc, _ = gi.counters()
        nq = a.reps * B
        esz = 1 if 'fp8' in var else (2 if 'bf16' in var else 4)
        byt = (sc.n_dist * esz * dim + sc.n_ids * 4) / a.reps + B * (4 * dim + 8 * k)
        run(0)
        torch.cuda.synchronize()
        res = (ids.cpu().numpy().copy(), sims.cpu().numpy().copy())
        same = ""
        if base is not None and base[0].shape[0] >= 256 and B >= 256:
            same = " same-as-first=%s" % (np.array_equal(base[0][:256], res[0][:256]) and np.array_equal(base[1][:256].view(np.uint32), res[1][:256].view(np.uint32)))
        if base is None:
            base = res
        print("%-44s %.4f ms/launch %9.0f QPS  %.0f GB/s (%.3f of 8 TB/s)  n_dist/q %.0f n_ids/q %.0f n_exp/q %.1f spills %d%s" % (
            var, ms, B / ms * 1e3, byt / (ms * 1e-3) / 1e9, byt / (ms * 1e-3) / 8e12, sc.n_dist / nq, sc.n_ids / nq,
            sc.n_expand / nq, sc.n_spill, same), flush=True)


if __name__ == "__main__":
    main()
