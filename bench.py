#!/usr/bin/env python3
"""bench.py -- HNSW.SEARCH throughput of the MI355X engine on BASELINE.json's
headline configuration (C2: 1M x 128 f32, M=16, ef=200, k=10, 1024-query batches).

A step = one pass of the hot path (hnsw_search_batch_device) over one batch of
queries already resident in HBM.  One process per GPU; the index is replicated,
every rank serves its own batch (weak scaling) and the [B,k] results are
all-gathered over RCCL.  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec


def draw_levels(n, m, seed=7):
    """floor(-ln U / ln M), numpy default_rng(seed); node 0 at level 0 (core.rs:393-405, 601-605)."""
    u = np.maximum(np.random.default_rng(seed).random(n), np.finfo(np.float64).tiny)
    lv = np.floor(-np.log(u) / np.log(float(m))).astype(np.int64)
    lv[0] = 0
    return np.minimum(lv, 31).astype(np.int32)


def clustered(n, dim, seed, centers):
    """64-cluster Gaussian mixture, sigma 0.1: lower intrinsic dimension than uniform"""
    rng = np.random.default_rng(seed)
    a = rng.integers(0, centers.shape[0], size=n)
    return (centers[a] + 0.1 * rng.standard_normal((n, dim), dtype=np.float32)).astype(np.float32)


def brute_force_gt(torch, V_dev, Q_dev, k):
    """exact top-k by squared L2 on the GPU (fp32 matmul, chunked)"""
    vn = (V_dev * V_dev).sum(1)
    out = []
    for i in range(0, Q_dev.shape[0], 256):
        q = Q_dev[i:i + 256]
        d = vn[None, :] - 2.0 * (q @ V_dev.T)
        out.append(d.topk(k, dim=1, largest=False).indices)
    return torch.cat(out).cpu().numpy()


def spawn_ranks(args):
    """`python bench.py --gpus N` without a launcher: re-exec under torch.distributed.run, one rank per GPU
    (what the driver's own command line does).  With fewer visible devices than ranks the ranks share
    cuda:0 and the collectives run over gloo on host copies -- a functional check of the N>1 path, not a
    scaling measurement (config.parallelism says so)."""
    import socket
    import subprocess
    import torch
    ndev = torch.cuda.device_count()
    if ndev == 0:
        raise SystemExit("bench.py needs a GPU: the engine has no CPU path")
    env = os.environ.copy()
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if ndev < args.gpus:
        env["HNSW_BENCH_ONE_DEVICE"] = "1"
        env["HNSW_BENCH_BACKEND"] = "gloo"
        print("[bench] %d ranks on %d visible device(s): sharing cuda:0, gloo collectives" % (args.gpus, ndev),
              file=sys.stderr, flush=True)
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    raise SystemExit(subprocess.call(cmd, env=env))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--nodes", type=int, default=1_000_000)
    ap.add_argument("--dim", type=int, default=128)
    ap.add_argument("--m", type=int, default=16)
    ap.add_argument("--ef", type=int, default=200)
    ap.add_argument("--k", type=int, default=10)
    ap.add_argument("--batch", type=int, default=1024)
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of the CPU baseline leg")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-clustered", action="store_true")
    ap.add_argument("--build", default="fast", choices=["fast", "exact"])
    ap.add_argument("--verify-gather", action="store_true",
                    help="N>1: rank 0 re-runs every rank's first batch on its own replica and compares with the gathered result")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return spawn_ranks(args)

    import torch
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the engine has no CPU path")
    one_device = bool(os.environ.get("HNSW_BENCH_ONE_DEVICE"))
    if one_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dist = None
    # HNSW_BENCH_BACKEND=gloo + HNSW_BENCH_ONE_DEVICE=1 let the N>1 control flow be exercised on a
    # single-GPU box (every rank on cuda:0, collectives on host copies); the driver's runs use RCCL.
    backend = os.environ.get("HNSW_BENCH_BACKEND", "nccl")
    coll_dev = torch.device("cuda", local_rank) if backend == "nccl" else torch.device("cpu")
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    def log(msg):
        if rank == 0:
            print("[bench %.1fs] %s" % (time.time() - t00, msg), file=sys.stderr, flush=True)

    t00 = time.time()
    from redis_hnsw_amd import Index
    N, dim, M, ef, k, B = args.nodes, args.dim, args.m, args.ef, args.k, args.batch
    cfg_is_c2 = (N, dim, M, ef, k, B) == (1_000_000, 128, 16, 200, 10, 1024)
    t0 = time.time()
    V = np.random.default_rng(1).random((N, dim), dtype=np.float32)
    n_qbatches = 8
    Qall = np.random.default_rng(2).random((n_qbatches * B * world, dim), dtype=np.float32)
    levels = draw_levels(N, M, 7)

    # ---- build (HNSW.NODE.ADD on the GPU), replicate ---------------------------------
    index = Index("bench", dim, M, ef, device=local_rank)
    graph = None
    t_build = None
    if rank == 0:
        tb = time.time()
        index.add_batch(V, levels=levels, mode=args.build)
        torch.cuda.synchronize()
        t_build = time.time() - tb
        log("built %d nodes in %.2f s (%s)" % (N, t_build, args.build))
        if world > 1 or not args.no_cpu_baseline:
            graph = index.export_graph(with_vectors=False)
    if world > 1:
        # one-time index distribution: rank 0's graph to every replica over RCCL
        from redis_hnsw_amd import shard
        g = shard.broadcast_graph(dist, graph, N, src=0, device=coll_dev)
        if rank != 0:
            g["vectors"] = V
            index.import_graph(g)
        dist.barrier()

    # ---- device-resident inputs/outputs ---------------------------------------------
    dev = torch.device("cuda", local_rank)
    myQ = torch.from_numpy(Qall[rank * n_qbatches * B:(rank + 1) * n_qbatches * B]).to(dev)
    # ids and similarities share one buffer so that a single all-gather moves both
    d_out = torch.empty((2, B, k), dtype=torch.int32, device=dev)
    d_ids = d_out[0]
    d_sims = d_out[1].view(torch.float32)
    d_n = torch.empty((B,), dtype=torch.int32, device=dev)
    g_out = torch.empty((world * 2, B, k), dtype=torch.int32, device=coll_dev) if world > 1 else None
    stream = torch.cuda.current_stream()

    # N > 1: the gather of step i runs on its own stream while step i+1 searches (two result buffers)
    overlap = world > 1
    if overlap:
        comm_stream = torch.cuda.Stream()
        bufs = [d_out, torch.empty_like(d_out)]
        gouts = [g_out, torch.empty_like(g_out)]
        ready = [torch.cuda.Event(), torch.cuda.Event()]
        gathered = [torch.cuda.Event(), torch.cuda.Event()]

    def step(i):
        q = myQ[(i % n_qbatches) * B:(i % n_qbatches + 1) * B]
        if not overlap:
            index.search_batch_device(q.data_ptr(), B, k, d_ids.data_ptr(), d_sims.data_ptr(), d_n.data_ptr(),
                                      stream.cuda_stream)
            return
        s_ = i % 2
        buf = bufs[s_]
        if i >= 2:
            stream.wait_event(gathered[s_])          # the gather that last read this buffer is done
        index.search_batch_device(q.data_ptr(), B, k, buf[0].data_ptr(), buf[1].data_ptr(), d_n.data_ptr(),
                                  stream.cuda_stream)
        ready[s_].record(stream)
        with torch.cuda.stream(comm_stream):
            comm_stream.wait_event(ready[s_])
            # the path's one real exchange: gather every shard's top-k (host copies in the gloo test mode)
            shard.gather_packed(dist, buf if backend == "nccl" else buf.cpu(), world, gouts[s_])
            gathered[s_].record(comm_stream)

    log("inputs resident; warm-up")
    for i in range(args.warmup):
        step(i)
    torch.cuda.synchronize()
    index.reset_counters()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t_start = time.perf_counter()
    ev0.record(stream)
    for i in range(args.steps):
        step(args.warmup + i)
    ev1.record(stream)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t_wall = time.perf_counter() - t_start
    if world > 1:
        tt = torch.tensor([t_wall], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        t_wall = float(tt.item())
    stream_ms = ev0.elapsed_time(ev1)
    gather_ok = None
    if world > 1 and args.verify_gather:
        # every rank searches its first batch, one gather; rank 0 repeats all of them on its own replica
        index.search_batch_device(myQ[:B].data_ptr(), B, k, d_ids.data_ptr(), d_sims.data_ptr(), d_n.data_ptr(),
                                  stream.cuda_stream)
        torch.cuda.synchronize()
        got = shard.gather_packed(dist, d_out if backend == "nccl" else d_out.cpu(), world).cpu().numpy()
        if rank == 0:
            gather_ok = True
            for r in range(world):
                q = torch.from_numpy(Qall[r * n_qbatches * B:r * n_qbatches * B + B]).to(dev)
                index.search_batch_device(q.data_ptr(), B, k, d_ids.data_ptr(), d_sims.data_ptr(), d_n.data_ptr(),
                                          stream.cuda_stream)
                torch.cuda.synchronize()
                gather_ok = gather_ok and bool(np.array_equal(got[r], d_out.cpu().numpy()))
            log("gathered == unsharded: %s" % gather_ok)
    log("timed region done: %.3f ms/step" % (1e3 * t_wall / args.steps))
    sc, _ = index.counters()

    # ---- recall@k against brute force (rank 0's batches) ------------------------------
    recall = None
    if rank == 0:
        V_dev = torch.from_numpy(V).to(dev)
        nb = min(2, n_qbatches)
        hits = tot = 0
        for b in range(nb):
            q = myQ[b * B:(b + 1) * B]
            index.search_batch_device(q.data_ptr(), B, k, d_ids.data_ptr(), d_sims.data_ptr(), d_n.data_ptr(),
                                      stream.cuda_stream)
            torch.cuda.synchronize()
            got = d_ids.cpu().numpy().astype(np.int64)
            gt = brute_force_gt(torch, V_dev, q, k)
            for a, bb in zip(got, gt):
                hits += len(set(a.tolist()) & set(bb.tolist()))
                tot += k
        recall = hits / tot
        log("recall@%d = %.4f" % (k, recall))
        del V_dev

    if rank != 0:
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return

    # ---- informational: the same configuration on clustered data, where the reference algorithm's
    # recall is high enough for recall parity to mean something (uniform 128-d: 0.27 at 1 M)
    clus = None
    if cfg_is_c2 and not args.no_clustered and world == 1:
        centers = np.random.default_rng(3).random((64, dim), dtype=np.float32)
        Vc = clustered(N, dim, 3, centers)
        Qc = torch.from_numpy(clustered(B, dim, 4, centers)).to(dev)
        ic = Index("bench-clustered", dim, M, ef, device=local_rank)
        ic.add_batch(Vc, levels=levels, mode=args.build)
        for _ in range(2):
            ic.search_batch_device(Qc.data_ptr(), B, k, d_ids.data_ptr(), d_sims.data_ptr(), d_n.data_ptr(), stream.cuda_stream)
        torch.cuda.synchronize()
        tc0 = time.perf_counter()
        for _ in range(5):
            ic.search_batch_device(Qc.data_ptr(), B, k, d_ids.data_ptr(), d_sims.data_ptr(), d_n.data_ptr(), stream.cuda_stream)
        torch.cuda.synchronize()
        tcl = (time.perf_counter() - tc0) / 5
        Vc_dev = torch.from_numpy(Vc).to(dev)
        gtc = brute_force_gt(torch, Vc_dev, Qc, k)
        gotc = d_ids.cpu().numpy().astype(np.int64)
        hit = sum(len(set(a.tolist()) & set(bb.tolist())) for a, bb in zip(gotc, gtc))
        clus = dict(data="64-cluster Gaussian mixture, sigma 0.1", recall_at_10=round(hit / (B * k), 4),
                    value=round(B / tcl, 1), unit="queries/s")
        log("clustered data: recall@%d = %.4f, %.3f ms/step" % (k, hit / (B * k), 1e3 * tcl))
        del Vc_dev, Vc
        ic.close()

    # ---- the same batch through the host-buffer entry point (PCIe in and out); informational
    Qh = Qall[:B]
    index.search_batch(Qh, k)
    th = time.perf_counter()
    for _ in range(5):
        index.search_batch(Qh, k)
    host_qps = 5 * B / (time.perf_counter() - th)

    # ---- roofline of the dominant kernel (k_search) -------------------------------------
    launches = args.steps
    bytes_per_launch = (sc.n_dist * 4 * dim + sc.n_ids * 4) / launches + B * (4 * dim + 8 * k)
    kernel_ms = stream_ms / launches     # HIP events on the launch stream around the timed region
    achieved = bytes_per_launch / (kernel_ms * 1e-3) / 1e9
    # HBM traffic per launch from the committed rocprofv3 PMC passes of this same command
    # (FETCH_SIZE x2 gfx950 correction + WRITE_SIZE, see profiles/); null for other workloads
    traffic = None
    try:
        tj = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
        if all(tj["config"].get(kk) == vv for kk, vv in dict(nodes=N, dim=dim, M=M, ef=ef, k=k, batch=B).items()):
            traffic = tj["k_search_hbm_bytes_per_launch"]
    except (OSError, ValueError, KeyError):
        pass
    roofline = dict(bound="hbm", achieved=round(achieved, 1), peak=HBM_PEAK_GBS, unit="GB/s",
                    frac=round(achieved / HBM_PEAK_GBS, 4), traffic=traffic,
                    kernel="k_search", kernel_ms=round(kernel_ms, 4),
                    algorithmic_bytes_per_launch=int(bytes_per_launch),
                    n_dist_per_query=round(sc.n_dist / (launches * B), 1),
                    n_ids_per_query=round(sc.n_ids / (launches * B), 1))

    # ---- CPU baseline: the oracle (C restatement of the Rust path), bounded sample -------
    cpu = None
    if not args.no_cpu_baseline and world == 1:   # timed at N=1 only: the other ranks would idle in the barrier
        from oracle import oracle
        graph["vectors"] = V
        o = oracle.OracleIndex.from_graph(dim, M, ef, graph)
        Qs = Qall[:B]
        o.search_batch(Qs[:64], k, threads=1)              # warm-up
        done = 0
        tc = time.perf_counter()
        chunk = 128
        while True:
            lo = done % B
            o.search_batch(Qs[lo:lo + chunk], k, threads=1)
            done += chunk
            if time.perf_counter() - tc > args.cpu_seconds:
                break
        t1 = time.perf_counter() - tc
        cores = os.cpu_count() or 1
        tc = time.perf_counter()
        reps = 0
        while True:
            o.search_batch(Qs, k, threads=cores)
            reps += 1
            if time.perf_counter() - tc > args.cpu_seconds / 2:
                break
        tN = time.perf_counter() - tc
        cpu = dict(value=round(done / t1, 1), unit="queries/s", cores=1, kind="port",
                   sample="%d queries of the same 1024-query batch on the same graph, 1 thread, %.1f s" % (done, t1),
                   all_cores=dict(value=round(reps * B / tN, 1), cores=cores))

    known = {(1_000_000, 128, 16, 200, 10, 1024): "C2", (1_000_000, 768, 32, 400, 100, 4096): "C3",
             (10_000_000, 128, 16, 200, 10, 1024): "C4"}
    cfg_name = known.get((N, dim, M, ef, k, B), "custom")
    qps = world * B * args.steps / t_wall
    out = {
        "metric": "HNSW.SEARCH QPS + recall@10, 1M x 128 f32, ef=200" if cfg_name == "C2" else
                  "HNSW.SEARCH QPS + recall@%d (%s)" % (k, cfg_name),
        "value": round(qps, 1), "unit": "queries/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(1e3 * t_wall / args.steps, 4), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "%s: %d nodes x dim %d, M=%d, ef=%d, k=%d, batch=%d queries/GPU, uniform[0,1) f32, replicated index"
                               % (cfg_name, N, dim, M, ef, k, B),
                   "nodes": N, "dim": dim, "M": M, "ef": ef, "k": k, "batch": B, "build": args.build,
                   "parallelism": "replica x%d, query batch sharded%s" % (
                       world, " (ranks share one device, gloo: functional check only)" if one_device and world > 1 else "")},
        "gather_verified": gather_ok,
        "recall_at_%d" % k: None if recall is None else round(recall, 4),
        "build_seconds": None if t_build is None else round(t_build, 2),
        "host_buffers_qps": round(host_qps, 1),
        "clustered": clus,
        "setup_seconds": round(time.time() - t0, 1),
        "roofline": roofline,
        "cpu_baseline": cpu,
    }
    print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
