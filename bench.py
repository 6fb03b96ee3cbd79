#!/usr/bin/env python3
"""bench.py -- HNSW.SEARCH throughput of the MI355X engine on BASELINE.json's headline configuration
(C2: 1M x 128 f32, M=16, ef=200, k=10, 1024-query batches).

A step = one pass of the hot path (hnsw_search_batch_device = one k_search launch) over one batch of
1024 queries already resident in HBM.  Consecutive steps are issued round-robin on `--streams` HIP streams
(default 3), so three batches are in flight at a time: the chip holds 2048 queries (two wavefronts per SIMD --
a lone 1024-query launch puts one on each, and a SIMD needs two to keep issuing), and
the third batch is what the dispatcher backfills from while the first two drain their long queries.  Nothing is
tuned for that: the engine sizes every launch from the launches it sees in flight, and the library asks the HIP
runtime for 8 hardware queues when it is loaded (streams that share a queue serialise; what the engine's own
lanes got is measured and reported as config.pipeline).  The same pipelining lives INSIDE the library for
callers that hand over one large batch: `host_buffers` (hnsw_search_batch, 8192 queries from host memory, PCIe
in and out) and `device_call` (one hnsw_search_batch_device call of 4096 / 8192 / 16384 queries) report it.
One process per GPU; the index is replicated, every rank serves its own batches (weak scaling) and the [B,k]
results are all-gathered over RCCL.  Prints ONE JSON line on rank 0.  (`single_process_group`, informational: the
one-process form a Redis module would use, hnsw_group_* -- only when this process sees more than one device.)

The graph the headline runs on (--graph):
  reference  the REFERENCE-ORDER graph (core.rs:489-599, one insert after the other): read from the fixture
             data/c2_ref_graph_1m.npz that tests/fixtures/make_ref_graph.py writes (the CPU oracle's serial
             build of the same seeded vectors and levels, ~35 min on one core -- too long to repeat inside
             a bench run), imported through hnsw_import.  Default when the fixture is present.
  exact      built here on the GPU in the reference's order (hnsw_add_batch mode 0).
  fast       the batched GPU build (hnsw_add_batch mode 1): not the reference's graph, recall parity only.
"""
import argparse
import json
import os
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")   # read by the HIP runtime when it starts (before torch is imported)

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: 8.0 TB/s spec -- the roofline's denominator
HBM_MEASURED_GBS = 6300.0  # same guide: what a streaming copy sustains from DRAM on this part
GATHER_PROFILE = os.path.join(ROOT, "profiles", "r5_gather_bw.txt")   # scripts/microbench/gather_bw.hip on the GPU box
FIXTURES = {(1_000_000, 128, 16, 200): os.path.join(ROOT, "data", "c2_ref_graph_1m.npz")}
FIXTURE_50K = os.path.join(ROOT, "data", "c2_ref_graph_50k.npz")
# the reference's own work counters of the C2 build at several prefix sizes (tests/fixtures/make_ref_counters.py)
REF_INSERT_COUNTERS = os.path.join(ROOT, "data", "c2_ref_insert_counters.json")
# BASELINE.json's configurations by name (bench.py --workload c3); c1 is timed inside every default run (c1_single_query)
WORKLOADS = {
    "c1": dict(nodes=10_000, dim=128, m=5, ef=200, k=10, batch=1),
    "c2": dict(nodes=1_000_000, dim=128, m=16, ef=200, k=10, batch=1024),
    "c3": dict(nodes=1_000_000, dim=768, m=32, ef=400, k=100, batch=4096),
    "c4": dict(nodes=10_000_000, dim=128, m=16, ef=200, k=10, batch=1024),
    "c5": dict(nodes=1_000_000, dim=128, m=16, ef=200, k=10, batch=1024, graph="exact"),
}


def insert_roofline(n_dist, n_ids, n_inserts, seconds, dim, source):
    """SURVEY 8d, per insert: bytes = n_dist_ins x 4 dim + n_ids_ins x 4 with the REFERENCE's counts (metric calls at
    core.rs:550, 621, 652, 711; ids scanned), over the time the engine took"""
    by = float(n_dist) * 4 * dim + float(n_ids) * 4
    gbs = by / max(seconds, 1e-12) / 1e9
    return dict(bound="hbm", achieved=round(gbs, 2), peak=HBM_PEAK_GBS, unit="GB/s", frac=round(gbs / HBM_PEAK_GBS, 5),
                algorithmic_bytes_per_insert=int(by / max(n_inserts, 1)), n_dist_per_insert=round(n_dist / max(n_inserts, 1), 1),
                n_ids_per_insert=round(n_ids / max(n_inserts, 1), 1), counts=source,
                note="an insert is a chain of dependent expansions on a few wavefronts: latency-bound by construction, the "
                     "fraction says how far from a bandwidth-bound stream it is")


def measured_gather(row_bytes, matrix_mb):
    """profiles/r5_gather_bw.txt: what the part delivers for read-only random gathers of whole rows of `row_bytes`
    (the search kernel's access pattern with nothing else to do) -- over 8 GB, where the 256 MB Infinity Cache cannot
    help, and over an array the size of this workload's vector matrix.  GB/s, best residency of each; None if absent."""
    far, this = None, None
    try:
        for line in open(GATHER_PROFILE):
            if not line.startswith("gather "):
                continue
            kv = dict(t.split("=") for t in line.split("#")[0].split()[1:])
            if int(kv["row_bytes"]) != row_bytes:
                continue
            gbs = 1e3 * float(kv["tbs"])
            mb = float(kv["array_mb"])
            if mb >= 8000:
                far = max(far or 0.0, gbs)
            if abs(mb - matrix_mb) <= 0.02 * matrix_mb:
                this = max(this or 0.0, gbs)
    except (OSError, ValueError, KeyError):
        return None
    if far is None:
        return None
    return dict(over_8_gb=far, over_an_array_of_this_matrix_size=this, source="profiles/r5_gather_bw.txt")


def ref_insert_counters(prefix):
    try:
        ent = json.load(open(REF_INSERT_COUNTERS))["prefixes"][str(prefix)]
        return ent["n_dist"], ent["n_ids"]
    except (OSError, ValueError, KeyError):
        return None


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


def load_graph_fixture(path, V):
    """levels / enterpoint / per-layer CSR written by tests/fixtures/make_ref_graph.py (+ this run's vectors)"""
    z = np.load(path)
    n = int(z["nodes"])
    row_ptr, col = [], []
    for l in range(int(z["max_layer"]) + 1):
        rp = np.zeros(n + 1, dtype=np.uint64)
        np.cumsum(z["deg%d" % l].astype(np.uint64), out=rp[1:])
        row_ptr.append(rp)
        col.append(z["col%d" % l].astype(np.uint32))
    return dict(vectors=V[:n], levels=z["levels"].astype(np.uint32), enterpoint=int(z["enterpoint"]),
                max_layer=int(z["max_layer"]), row_ptr=row_ptr, col=col), float(z["build_seconds"])


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def usable_cores():
    """host threads this process may really use: the affinity mask, capped by the cgroup CPU quota
    (a container that sees 256 CPUs but has a 16-CPU quota is throttled beyond 16 busy threads)"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if q > 0:
                    n = min(n, max(1, q // per))
            break
        except (OSError, ValueError, IndexError):
            continue
    return n


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


def timed_launch_rows(rows, batch):
    """Of the (kernel, lds, grid, dispatches, avg) groups of a --pmc pass, the ones that are the TIMED step's launch
    shape: one workgroup of 64 threads per query (the specialised kernel), or the general kernel's grid-stride form
    (at most 2048 workgroups) -- never the two-wave kernel of a lone launch (128 threads per query) and never the
    1024-query chunks hnsw_search_batch cuts a host batch into.  Most dispatches first."""
    want = {64 * batch, 64 * min(batch, 2048)}
    hit = [r for r in rows if int(r[2]) in want and "duo" not in r[0]]
    return sorted(hit, key=lambda r: -r[3])


def measure_traffic(argv, log, batch, algorithmic):
    """HBM bytes per k_search launch of THIS command line, measured now: two short re-runs under
    `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` (separate passes: the two do not fit one), the dispatch group
    whose grid is the timed step's (`timed_launch_rows`), FETCH x 2 (gfx950 correction) + WRITE.  The figure must lie
    within 0.9 .. 1.3 x the algorithmic bytes of a launch, else the wrong group was read (or the kernel re-reads) and
    the leg FAILS: (None, reason).  (None, None) if rocprofv3 is missing or a pass fails -- the caller then falls back
    to the committed figure and says so."""
    import shutil
    import sqlite3
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3")
    if not exe:
        return None, None
    keep, skip = [], False
    for a in argv:                                   # same workload; fewer steps, no extras, no nested profiling
        if skip:
            skip = False
            continue
        if a in ("--steps", "--warmup", "--cpu-seconds"):
            skip = True
            continue
        if a.startswith(("--steps=", "--warmup=", "--cpu-seconds=")):
            continue
        keep.append(a)
    sub = keep + ["--steps", "12", "--warmup", "2", "--only-timed"]
    tmp = tempfile.mkdtemp(prefix="hnsw_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    vals = {}
    try:
        for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
            out = os.path.join(tmp, ctr)
            cmd = [exe, "--pmc", ctr, "-d", out, "-o", "p", "--", sys.executable, os.path.abspath(__file__)] + sub
            p = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=600)
            db = None
            for dp, _, files in os.walk(out):
                for f in files:
                    if f.endswith("_results.db"):
                        db = os.path.join(dp, f)
            if p.returncode != 0 or db is None:
                log("rocprofv3 --pmc %s failed (rc %s): roofline.traffic falls back to profiles/traffic.json" % (ctr, p.returncode))
                return None, None
            cur = sqlite3.connect(db).cursor()
            rows = list(cur.execute(
                "select kernel_name, lds_block_size, grid_size, count(*), avg(value) from counters_collection "
                "where kernel_name like '%k_search%' and counter_name = ? group by kernel_name, lds_block_size, grid_size",
                (ctr,)))
            rows = timed_launch_rows(rows, batch)
            if not rows:
                return None, "no k_search dispatch group with the timed step's grid (64 x %d) under --pmc %s" % (batch, ctr)
            vals[ctr] = rows[0]
        kib_r, kib_w = vals["FETCH_SIZE"][4], vals["WRITE_SIZE"][4]
        traffic = int(kib_r * 1024 * 2.0 + kib_w * 1024)
        src = ("measured in this run: rocprofv3 --pmc FETCH_SIZE, then --pmc WRITE_SIZE, of `bench.py %s`; "
               "%d launches of %s with grid %d (the timed step's: one workgroup per query); FETCH_SIZE %.0f KiB x 2 (gfx950) + "
               "WRITE_SIZE %.0f KiB per launch; launches are serialised under --pmc" % (
                   " ".join(sub), vals["FETCH_SIZE"][3], vals["FETCH_SIZE"][0].split("(")[0].replace("void ", ""),
                   vals["FETCH_SIZE"][2], kib_r, kib_w))
        ratio = traffic / max(algorithmic, 1.0)
        log("traffic per launch %.3f GB (in-run PMC) = %.3f x the algorithmic bytes" % (traffic / 1e9, ratio))
        if not 0.9 <= ratio <= 1.3:
            return None, "REJECTED: %.3f GB per launch is %.2f x the algorithmic bytes (outside 0.9 .. 1.3); %s" % (traffic / 1e9, ratio, src)
        return traffic, src
    except (subprocess.TimeoutExpired, OSError, sqlite3.Error) as e:
        log("in-run traffic measurement failed: %r" % (e,))
        return None, None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--workload", default=None, choices=sorted(WORKLOADS),
                    help="a BASELINE.json configuration by name: sets --nodes/--dim/--m/--ef/--k/--batch (and, for c5, "
                         "--graph exact: the index is BUILT on the GPU in the reference's insert order before it is queried)")
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--nodes", type=int, default=1_000_000)
    ap.add_argument("--dim", type=int, default=128)
    ap.add_argument("--m", type=int, default=16)
    ap.add_argument("--ef", type=int, default=200)
    ap.add_argument("--k", type=int, default=10)
    ap.add_argument("--batch", type=int, default=1024)
    ap.add_argument("--streams", type=int, default=3, help="steps in flight (HIP streams used round-robin)")
    ap.add_argument("--waves-per-cu", type=int, default=0, help="engine tuning waves_per_cu (0 = default 8)")
    ap.add_argument("--launch-concurrency", type=int, default=0,
                    help="engine tuning launch_concurrency (0 = default: the engine observes the launches in flight)")
    ap.add_argument("--no-traffic", action="store_true", help="do not re-run under rocprofv3 --pmc for roofline.traffic")
    ap.add_argument("--only-timed", action="store_true",
                    help="stop after the timed region (what the --pmc passes of roofline.traffic run); prints no JSON line")
    ap.add_argument("--tuning", action="append", default=[], metavar="KEY=VALUE",
                    help="extra engine tuning for experiments (hnsw_set_tuning), repeatable")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of the CPU baseline leg")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-clustered", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the informational legs (large batch, fast build, C1)")
    ap.add_argument("--graph", "--build", dest="graph", default="auto", choices=["auto", "reference", "exact", "fast"])
    ap.add_argument("--dry", action="store_true",
                    help="stop after the index is in place on every rank (replication over RCCL for N > 1) and the 64-query "
                         "identity check: a two-minute proof of the multi-GPU path before a long run; prints one JSON line")
    ap.add_argument("--verify-gather", action="store_true",
                    help="N>1: rank 0 re-runs every rank's first batch on its own replica and compares with the gathered result")
    args = ap.parse_args()
    if args.workload:
        for key, val in WORKLOADS[args.workload].items():
            if key == "graph":
                if args.graph == "auto":
                    args.graph = val
            else:
                setattr(args, key, val)
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
    from redis_hnsw_amd import Index, shard
    N, dim, M, ef, k, B = args.nodes, args.dim, args.m, args.ef, args.k, args.batch
    S = max(1, args.streams)
    cfg_is_c2 = (N, dim, M, ef, k, B) == (1_000_000, 128, 16, 200, 10, 1024)
    t0 = time.time()
    V = np.random.default_rng(1).random((N, dim), dtype=np.float32)
    n_qbatches = 8
    Qall = np.random.default_rng(2).random((n_qbatches * B * world, dim), dtype=np.float32)
    levels = draw_levels(N, M, 7)

    # ---- the graph -------------------------------------------------------------------
    fixture = FIXTURES.get((N, dim, M, ef))
    mode = args.graph
    if mode == "auto":
        mode = "reference" if fixture and os.path.exists(fixture) else "fast"
    if mode == "reference" and not (fixture and os.path.exists(fixture)):
        raise SystemExit("--graph reference needs %s (python tests/fixtures/make_ref_graph.py --out ...)" % fixture)
    index = Index("bench", dim, M, ef, device=local_rank)
    if args.launch_concurrency:
        index.set_tuning("launch_concurrency", args.launch_concurrency)
    if args.waves_per_cu:
        index.set_tuning("waves_per_cu", args.waves_per_cu)
    for kv in args.tuning:
        key, val = kv.split("=")
        index.set_tuning(key, int(val))
    graph = None
    t_build = None
    replication = None
    graph_desc = {"reference": "reference-order (serial core.rs:489-599 order; fixture built by the CPU oracle, imported with hnsw_import)",
                  "exact": "reference-order, built on the GPU (hnsw_add_batch mode 0)",
                  "fast": "batched GPU build (hnsw_add_batch mode 1; NOT the reference's graph)"}[mode]
    replicate = world > 1 and os.environ.get("HNSW_BENCH_REPLICATE", "1") != "0"
    if mode == "reference":
        # rank 0 imports the fixture; the replicas receive the index over RCCL (HNSW_BENCH_REPLICATE=0: every rank
        # reads the fixture itself instead)
        if rank == 0 or not replicate:
            graph, oracle_build_s = load_graph_fixture(fixture, V)
            tb = time.time()
            index.import_graph(graph)
            torch.cuda.synchronize()
            log("imported the reference-order graph (%d nodes) in %.2f s" % (N, time.time() - tb))
    else:
        if rank == 0:
            tb = time.time()
            index.add_batch(V, levels=levels, mode=mode)
            torch.cuda.synchronize()
            t_build = time.time() - tb
            log("built %d nodes in %.2f s (%s)" % (N, t_build, mode))
            if world == 1 and not args.no_cpu_baseline:
                graph = index.export_graph(with_vectors=False)
    if replicate or (world > 1 and mode != "reference"):
        # one-time index distribution: rank 0's tables straight out of its HBM into every replica's
        # -- vectors included -- as device-pointer broadcasts over RCCL (host-staged when the ranks share one
        # device in the gloo functional mode)
        tb = time.time()
        via = "device" if backend == "nccl" else "host"
        if via == "device":
            # the collective has never carried raw engine pointers on this node before: prove it on 4 KB first
            probe = torch.arange(1024, dtype=torch.int32, device=torch.device("cuda", local_rank)) * (1 if rank == 0 else 0)
            try:
                dist.broadcast(shard.device_bytes(probe.data_ptr(), 4096, torch.device("cuda", local_rank)), src=0)
                torch.cuda.synchronize()
                ok_probe = bool((probe == torch.arange(1024, dtype=torch.int32, device=probe.device)).all().item())
            except RuntimeError as e:                    # pragma: no cover (needs a multi-GPU node)
                log("RCCL broadcast on a raw device pointer failed (%s): host-staged replication instead" % (e,))
                ok_probe = False
            flag = torch.tensor([1 if ok_probe else 0], dtype=torch.int32, device=torch.device("cuda", local_rank))
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if int(flag.item()) == 0:
                via = "host"
        moved = shard.replicate_index(dist, index, src=0, device=torch.device("cuda", local_rank), via=via)
        dist.barrier()
        t_repl = time.time() - tb
        log("replicated the index to %d ranks: %.2f GB per replica in %.2f s (%s)" % (
            world, moved / 1e9, t_repl, "RCCL, HBM to HBM" if via == "device" else "host-staged, %s" % backend))
        replication = dict(bytes_per_replica=int(moved), seconds=round(t_repl, 3),
                           transport="rccl broadcast of device pointers" if via == "device" else "%s, host-staged" % backend)

    # ---- device-resident inputs/outputs ---------------------------------------------
    dev = torch.device("cuda", local_rank)
    myQ = torch.from_numpy(Qall[rank * n_qbatches * B:(rank + 1) * n_qbatches * B]).to(dev)
    streams = [torch.cuda.Stream() for _ in range(S)]
    extra_stream = torch.cuda.Stream()               # a fourth one for the bf16 leg (created now: streams made later, once more
    #                                                  than GPU_MAX_HW_QUEUES exist, share a hardware queue and serialise)
    # ids and similarities share one buffer so that a single all-gather moves both
    bufs = [torch.empty((2, B, k), dtype=torch.int32, device=dev) for _ in range(S)]
    d_ns = [torch.empty((B,), dtype=torch.int32, device=dev) for _ in range(S)]
    d_out = bufs[0]
    d_ids, d_sims, d_n = d_out[0], d_out[1].view(torch.float32), d_ns[0]
    cur = torch.cuda.current_stream()
    if replication is not None:
        # every replica must answer like rank 0's index: the same 64 queries everywhere, ids + similarity bits gathered
        chk_q = torch.from_numpy(Qall[:64]).to(dev)
        chk = torch.empty((2, 64, k), dtype=torch.int32, device=dev)
        chk_n = torch.empty((64,), dtype=torch.int32, device=dev)
        index.search_batch_device(chk_q.data_ptr(), 64, k, chk[0].data_ptr(), chk[1].data_ptr(), chk_n.data_ptr(), cur.cuda_stream)
        torch.cuda.synchronize()
        allc = shard.gather_packed(dist, chk if backend == "nccl" else chk.cpu(), world).cpu().numpy()
        same = all(np.array_equal(allc[0], allc[r_]) for r_ in range(1, world))
        replication["replicas_answer_identically"] = bool(same)
        if not same:
            raise SystemExit("index replication: a replica answers differently from rank 0")

    def peer_report():
        """what this rank's device can reach directly (xGMI peer access), for the N > 1 line"""
        nd = torch.cuda.device_count()
        if one_device or nd < 2:
            return None
        try:
            return [int(d) for d in range(nd) if d != local_rank and torch.cuda.can_device_access_peer(local_rank, d)]
        except (RuntimeError, AssertionError):
            return None

    if args.dry:
        info = index.info()
        mine = dict(rank=rank, device=local_rank, nodes=int(info.node_count), hbm_bytes=int(info.hbm_bytes), peers=peer_report())
        objs = [mine]
        if world > 1:
            objs = [None] * world
            dist.all_gather_object(objs, mine)
        if rank == 0:
            print(json.dumps({"dry": True, "n_gpus": world, "graph": mode, "collective_backend": backend if world > 1 else None,
                              "rccl_world": world if (world > 1 and backend == "nccl") else None,
                              "index_replication": replication, "ranks": objs,
                              "setup_seconds": round(time.time() - t0, 1)}))
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return

    # N > 1: the gather of step i runs on its own stream while later steps search
    if world > 1:
        comm_stream = torch.cuda.Stream()
        gouts = [torch.empty((world * 2, B, k), dtype=torch.int32, device=coll_dev) for _ in range(S)]
        ready = [torch.cuda.Event() for _ in range(S)]
        gathered = [torch.cuda.Event() for _ in range(S)]

    def step(i):
        q = myQ[(i % n_qbatches) * B:(i % n_qbatches + 1) * B]
        s_ = i % S
        st, buf = streams[s_], bufs[s_]
        if world > 1 and i >= S:
            st.wait_event(gathered[s_])              # the gather that last read this buffer is done
        index.search_batch_device(q.data_ptr(), B, k, buf[0].data_ptr(), buf[1].data_ptr(), d_ns[s_].data_ptr(),
                                  st.cuda_stream)
        if world > 1:
            ready[s_].record(st)
            with torch.cuda.stream(comm_stream):
                comm_stream.wait_event(ready[s_])
                # the path's one real exchange: gather every shard's top-k (host copies in the gloo test mode)
                shard.gather_packed(dist, buf if backend == "nccl" else buf.cpu(), world, gouts[s_])
                gathered[s_].record(comm_stream)

    log("inputs resident; warm-up")
    # the clocks of an idle GPU need more than a few half-millisecond launches to settle: a fixed number of
    # untimed launches first (reported as config.prewarm_launches), then the W warm-up steps asked for
    PREWARM = 48
    for i in range(PREWARM):
        step(i)
    torch.cuda.synchronize()
    for i in range(max(args.warmup, 0)):
        step(PREWARM + i)
    torch.cuda.synchronize()
    index.reset_counters()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    ev0 = [torch.cuda.Event(enable_timing=True) for _ in range(S)]
    ev1 = [torch.cuda.Event(enable_timing=True) for _ in range(S)]
    per_stream = [len(range(s_, args.steps, S)) for s_ in range(S)]
    marks = [[] for _ in range(S)]                   # one event after every launch, on its stream
    t_start = time.perf_counter()
    for s_ in range(S):
        ev0[s_].record(streams[s_])
    for i in range(args.steps):
        step(PREWARM + args.warmup + i)
        if world == 1:
            e_ = torch.cuda.Event(enable_timing=True)
            e_.record(streams[i % S])
            marks[i % S].append(e_)
    for s_ in range(S):
        ev1[s_].record(streams[s_])
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t_wall = time.perf_counter() - t_start
    t_local = t_wall                              # this rank's own time (the roofline is rank 0's kernel)
    if world > 1:
        tt = torch.tensor([t_wall], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        t_wall = float(tt.item())
    per_rank = None
    gather_cmp = None
    if world > 1:
        # what every rank saw (the driver computes efficiency from `value`; these show the spread behind it)
        objs = [None] * world
        dist.all_gather_object(objs, dict(rank=rank, seconds=t_local, qps=B * args.steps / t_local, peers=peer_report()))
        per_rank = dict(qps=[round(o_["qps"], 1) for o_ in objs], seconds=[round(o_["seconds"], 5) for o_ in objs],
                        peer_access=[o_["peers"] for o_ in objs],
                        ms_per_step_min=round(1e3 * min(o_["seconds"] for o_ in objs) / args.steps, 4),
                        ms_per_step_max=round(1e3 * max(o_["seconds"] for o_ in objs) / args.steps, 4))
        # the per-step exchange as an RCCL all-gather of the packed [2,B,k] block vs the alternative
        # without a collective -- every rank copies its own block to pinned host memory (hipMemcpyAsync D2H)
        reps_g = 50
        hostbuf = torch.empty((2, B, k), dtype=torch.int32).pin_memory()
        gsrc = bufs[0] if backend == "nccl" else bufs[0].cpu()
        for _ in range(5):
            shard.gather_packed(dist, gsrc, world, gouts[0])
            hostbuf.copy_(bufs[0], non_blocking=True)
        torch.cuda.synchronize()
        dist.barrier()
        tg = time.perf_counter()
        for _ in range(reps_g):
            shard.gather_packed(dist, gsrc, world, gouts[0])
        torch.cuda.synchronize()
        t_ag = (time.perf_counter() - tg) / reps_g
        td = time.perf_counter()
        for _ in range(reps_g):
            hostbuf.copy_(bufs[0], non_blocking=True)
        torch.cuda.synchronize()
        t_d2h = (time.perf_counter() - td) / reps_g
        tt2 = torch.tensor([t_ag, t_d2h], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(tt2, op=dist.ReduceOp.MAX)
        gather_cmp = dict(bytes_per_rank=int(2 * B * k * 4), allgather_us=round(1e6 * float(tt2[0]), 1),
                          d2h_to_pinned_us=round(1e6 * float(tt2[1]), 1), transport=backend,
                          note="back to back, nothing overlapped; in the timed loop the gather of step i runs on its own "
                               "stream under the search of step i+1")
    # average duration of ONE k_search launch: launches on a stream run back to back, so the stream's
    # elapsed time / its launches (what rocprofv3 --kernel-trace reports as the kernel's average)
    kernel_ms = float(np.mean([ev0[s_].elapsed_time(ev1[s_]) / per_stream[s_] for s_ in range(S) if per_stream[s_]]))
    # per-launch durations (launches on a stream run back to back): robust against a short timed region
    per_launch = []
    for s_ in range(S):
        prev = ev0[s_]
        for e_ in marks[s_]:
            per_launch.append(prev.elapsed_time(e_))
            prev = e_
    kernel_ms_median = float(np.median(per_launch)) if per_launch else None
    kernel_ms_p90 = float(np.percentile(per_launch, 90)) if per_launch else None
    log("timed region done: %.3f ms/step, %.3f ms per launch with %d in flight" % (1e3 * t_wall / args.steps, kernel_ms, S))
    if args.only_timed:
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return
    sc, _ = index.counters()

    gather_ok = None
    if world > 1 and args.verify_gather:
        # every rank searches its first batch, one gather; rank 0 repeats all of them on its own replica
        index.search_batch_device(myQ[:B].data_ptr(), B, k, d_ids.data_ptr(), d_sims.data_ptr(), d_n.data_ptr(),
                                  cur.cuda_stream)
        torch.cuda.synchronize()
        got = shard.gather_packed(dist, d_out if backend == "nccl" else d_out.cpu(), world).cpu().numpy()
        if rank == 0:
            gather_ok = True
            for r in range(world):
                q = torch.from_numpy(Qall[r * n_qbatches * B:r * n_qbatches * B + B]).to(dev)
                index.search_batch_device(q.data_ptr(), B, k, d_ids.data_ptr(), d_sims.data_ptr(), d_n.data_ptr(),
                                          cur.cuda_stream)
                torch.cuda.synchronize()
                gather_ok = gather_ok and bool(np.array_equal(got[r], d_out.cpu().numpy()))
            log("gathered == unsharded: %s" % gather_ok)

    def search_now(q_dev, nq):
        index.search_batch_device(q_dev.data_ptr(), nq, k, d_ids.data_ptr(), d_sims.data_ptr(), d_n.data_ptr(), cur.cuda_stream)
        torch.cuda.synchronize()

    # ---- exact work counters of the timed batches (one launch at a time, nothing forgotten) -------
    # With several launches sharing the CUs the LDS visited table is the bounded one: a
    # re-met node may be evaluated twice, so the timed region's n_dist can exceed the reference's.  The
    # algorithmic bytes are the reference's: count them in a pass where the table holds everything.
    index.set_tuning("launch_concurrency", 1)
    index.set_tuning("waves_per_cu", 4)
    index.set_tuning("visited_bounded", 0)           # the exact set: LDS table, HBM table beyond it
    index.reset_counters()
    nb_exact = min(n_qbatches, args.steps, max(1, 8192 // B))
    for b in range(nb_exact):
        search_now(myQ[b * B:(b + 1) * B], B)
    sx, _ = index.counters()
    index.set_tuning("visited_bounded", 1)
    index.set_tuning("waves_per_cu", args.waves_per_cu or 8)
    index.set_tuning("launch_concurrency", args.launch_concurrency)
    n_dist_q, n_ids_q, n_exp_q = sx.n_dist / (nb_exact * B), sx.n_ids / (nb_exact * B), sx.n_expand / (nb_exact * B)
    redo = sc.n_dist / (args.steps * B) / n_dist_q - 1.0 if args.steps else 0.0

    # ---- recall@k against brute force (rank 0's batches) ------------------------------
    recall = None
    if rank == 0:
        V_dev = torch.from_numpy(V).to(dev)
        nb = min(2, n_qbatches)
        hits = tot = 0
        for b in range(nb):
            q = myQ[b * B:(b + 1) * B]
            search_now(q, B)
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

    extras = world == 1 and not args.no_extras
    # ---- the literal BASELINE shape: ONE 1024-query launch at a time (one wave per SIMD, nothing to backfill from)
    lone = None
    if world == 1:
        for _ in range(3):
            search_now(myQ[:B], B)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 12
        e0.record(cur)
        for r_ in range(reps):
            q = myQ[(r_ % n_qbatches) * B:(r_ % n_qbatches + 1) * B]
            index.search_batch_device(q.data_ptr(), B, k, d_ids.data_ptr(), d_sims.data_ptr(), d_n.data_ptr(), cur.cuda_stream)
        e1.record(cur)
        torch.cuda.synchronize()
        msl = e0.elapsed_time(e1) / reps
        byl = B * (n_dist_q * 4 * dim + n_ids_q * 4 + 4 * dim + 8 * k)
        lone = dict(batch=B, launches_in_flight=1, two_wave_kernel=bool(index.last_search_was_duo()),
                    kernel_ms=round(msl, 4), value=round(B / msl * 1e3, 1), unit="queries/s",
                    achieved=round(byl / (msl * 1e-3) / 1e9, 1), frac=round(byl / (msl * 1e-3) / 1e9 / HBM_PEAK_GBS, 4))
        log("lone %d-query launches: %.3f ms each, %.0f GB/s" % (B, msl, lone["achieved"]))

    # ---- host memory in, host memory out (PCIe both ways; never `value`): hnsw_search_batch pipelines a large
    # batch itself -- pinned staging, H2D / kernel / D2H of different chunks overlapped on the engine's lanes
    Qh = Qall[:8 * B]
    index.search_batch(Qh, k)
    per_call8 = []
    for _ in range(6):
        tc0 = time.perf_counter()
        hb_ids, _, _ = index.search_batch(Qh, k)
        per_call8.append(time.perf_counter() - tc0)
    host_qps = Qh.shape[0] / float(np.median(per_call8))
    index.search_batch(Qall[:B], k)
    per_call = []
    for _ in range(6):
        tc0 = time.perf_counter()
        index.search_batch(Qall[:B], k)
        per_call.append(time.perf_counter() - tc0)
    host_qps_1024 = B / float(np.median(per_call))
    host = dict(batch=int(Qh.shape[0]), value=round(host_qps, 1), unit="queries/s", one_batch_of_1024=round(host_qps_1024, 1),
                per_call_ms=[round(1e3 * x, 3) for x in per_call8], per_call_ms_1024=[round(1e3 * x, 3) for x in per_call],
                note="hnsw_search_batch from pageable host memory, results back in host memory, one call at a time; the rates are "
                     "batch / MEDIAN call time of six calls (every call is listed: one call in a run can stall for several ms on "
                     "the host side)")
    log("host buffers: %.0f QPS at B=%d (%s ms), %.0f at B=%d (%s ms)" % (
        host_qps, Qh.shape[0], " ".join("%.2f" % (1e3 * x) for x in per_call8), host_qps_1024, B,
        " ".join("%.2f" % (1e3 * x) for x in per_call)))
    pipe = index.pipeline_info()

    # ---- one hnsw_search_batch_device CALL per size, calls back to back on one stream: the engine splits a call
    # into 1024-query chunks over its own lanes and joins them back, so every call pays its own drain
    big = None
    dev_calls = []
    if extras:
        for Bb in (4096, 8192, 16384):
            Qb = torch.from_numpy(np.random.default_rng(5).random((Bb, dim), dtype=np.float32)).to(dev)
            bi = torch.empty((Bb, k), dtype=torch.int32, device=dev)
            bs = torch.empty((Bb, k), dtype=torch.float32, device=dev)
            bn = torch.empty((Bb,), dtype=torch.int32, device=dev)
            for _ in range(2):
                index.search_batch_device(Qb.data_ptr(), Bb, k, bi.data_ptr(), bs.data_ptr(), bn.data_ptr(), cur.cuda_stream)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            reps = 8
            e0.record(cur)
            for _ in range(reps):
                index.search_batch_device(Qb.data_ptr(), Bb, k, bi.data_ptr(), bs.data_ptr(), bn.data_ptr(), cur.cuda_stream)
            e1.record(cur)
            torch.cuda.synchronize()
            msb = e0.elapsed_time(e1) / reps
            byb = Bb * (n_dist_q * 4 * dim + n_ids_q * 4 + 4 * dim + 8 * k)
            ent = dict(batch=Bb, calls_in_flight=1, ms_per_call=round(msb, 4), value=round(Bb / msb * 1e3, 1), unit="queries/s",
                       achieved=round(byb / (msb * 1e-3) / 1e9, 1), frac=round(byb / (msb * 1e-3) / 1e9 / HBM_PEAK_GBS, 4))
            dev_calls.append(ent)
            log("one %d-query device call: %.3f ms, %.0f GB/s" % (Bb, msb, ent["achieved"]))
            del Qb, bi, bs, bn
        big = dev_calls[0]

    # ---- informational: the same configuration on clustered data, where the reference algorithm's
    # recall is high enough for recall parity to mean something (uniform 128-d: 0.22-0.27 at 1 M)
    clus = None
    fast_build = None
    if cfg_is_c2 and extras:
        tb = time.time()
        ifast = Index("bench-fast", dim, M, ef, device=local_rank)
        ifast.add_batch(V, levels=levels, mode="fast")
        torch.cuda.synchronize()
        tfb = time.time() - tb
        ifast.search_batch_device(myQ[:B].data_ptr(), B, k, d_ids.data_ptr(), d_sims.data_ptr(), d_n.data_ptr(), cur.cuda_stream)
        torch.cuda.synchronize()
        V_dev = torch.from_numpy(V).to(dev)
        gotf = d_ids.cpu().numpy().astype(np.int64)
        gtf = brute_force_gt(torch, V_dev, myQ[:B], k)
        del V_dev
        rc = ref_insert_counters(N)
        fast_build = dict(build_seconds=round(tfb, 2), inserts_per_s=round(N / tfb, 1),
                          roofline=None if rc is None else insert_roofline(
                              rc[0], rc[1], N, tfb, dim, "the oracle's serial build of the same %d vectors (data/c2_ref_insert_counters.json); "
                              "the batched build itself evaluates about half of them" % N),
                          recall_at_10=round(sum(len(set(a.tolist()) & set(b.tolist())) for a, b in zip(gotf, gtf)) / (B * k), 4),
                          note="hnsw_add_batch mode 1 (batched GPU build, BASELINE config 5): not the reference's insert order")
        ifast.close()
        log("fast GPU build: %.2f s, recall@10 %.4f" % (tfb, fast_build["recall_at_10"]))
    exact_build = None
    single_add = None
    single_delete = None
    if cfg_is_c2 and extras and world == 1:
        # the reference-order build (hnsw_add_batch mode 0: plans in parallel, validated in-order commits) on a
        # bounded prefix, CHECKED row for row against the oracle's serial build of the same prefix (committed
        # fixture, tests/fixtures/make_ref_graph.py --nodes 50000); the full 1 M build and its identity check are
        # scripts/exact_build_check.py
        NE = 50_000
        ie = Index("bench-exact", dim, M, ef, device=local_rank)
        te = time.time()
        ie.add_batch(V[:NE], levels=levels[:NE], mode="exact")
        te = time.time() - te
        identical, why = None, "data/c2_ref_graph_50k.npz is missing"
        if os.path.exists(FIXTURE_50K):
            want, _ = load_graph_fixture(FIXTURE_50K, V)
            got = ie.export_graph()
            identical = (want["enterpoint"] == got["enterpoint"] and want["max_layer"] == got["max_layer"]
                         and np.array_equal(want["levels"], got["levels"])
                         and all(np.array_equal(a_, b_) for a_, b_ in zip(want["row_ptr"], got["row_ptr"]))
                         and all(np.array_equal(a_, b_) for a_, b_ in zip(want["col"], got["col"])))
            why = "levels, enterpoint and every adjacency row of every layer in stored order == the CPU oracle's serial build"
            if not identical:
                raise SystemExit("gpu_exact_build: the GPU's reference-order graph differs from the oracle's fixture")
        rc = ref_insert_counters(NE)
        exact_build = dict(nodes=NE, build_seconds=round(te, 2), inserts_per_s=round(NE / te, 1), identical=identical,
                           checked_against=why,
                           roofline=None if rc is None else insert_roofline(
                               rc[0], rc[1], NE, te, dim, "the oracle's serial build of the same %d-node prefix (data/c2_ref_insert_counters.json)" % NE),
                           note="hnsw_add_batch mode 0 on the first 50 k nodes (the rate grows with the index; the whole 1 M "
                                "build: profiles/r3_c5_exact_build_1m.json)")
        # HNSW.NODE.ADD as the Redis command issues it (src/lib.rs:356: one add_node per call): single hnsw_add
        # calls on that index, timed, then CHECKED against the oracle making the same inserts on the same graph
        if identical:
            NA = 200
            extra_v = np.random.default_rng(11).random((NA, dim), dtype=np.float32)
            extra_l = draw_levels(NA, M, 13)
            ta = time.time()
            for i in range(NA):
                ie.add_node("single%d" % i, extra_v[i], level=int(extra_l[i]))
            ta = (time.time() - ta) / NA
            from oracle import oracle as _orc                # checker only, after the timed region
            want["vectors"] = V[:NE]
            oa = _orc.OracleIndex.from_graph(dim, M, ef, want)
            c0_ = oa.insert_counters()
            c0_ = (c0_.n_dist, c0_.n_ids)
            tc = time.time()
            for i in range(NA):
                oa.add(extra_v[i], int(extra_l[i]))
            tc = (time.time() - tc) / NA
            c1_ = oa.insert_counters()
            ga, gb = oa.export(), ie.export_graph()
            same = (ga["enterpoint"] == gb["enterpoint"] and np.array_equal(ga["levels"], gb["levels"])
                    and all(np.array_equal(a_, b_) for a_, b_ in zip(ga["row_ptr"], gb["row_ptr"]))
                    and all(np.array_equal(a_, b_) for a_, b_ in zip(ga["col"], gb["col"])))
            if not same:
                raise SystemExit("single_add: the graph after %d hnsw_add calls differs from the oracle's" % NA)
            single_add = dict(workload="HNSW.NODE.ADD: %d single hnsw_add calls (host vectors) on the %d-node reference-order index" % (NA, NE),
                              gpu_ms=round(1e3 * ta, 3), cpu_oracle_ms=round(1e3 * tc, 3), identical=True,
                              roofline=insert_roofline(c1_.n_dist - c0_[0], c1_.n_ids - c0_[1], NA, ta * NA, dim,
                                                       "the oracle making the same %d inserts on the same graph" % NA))
            # HNSW.NODE.DEL the same way (src/lib.rs:397): single hnsw_delete calls, timed, then checked
            ND = 100
            victims = [int(v) for v in np.random.default_rng(17).choice(NE, ND, replace=False)]
            td = time.time()
            for v in victims:
                ie.delete_node("node%d" % v)
            td = (time.time() - td) / ND
            tcd = time.time()
            for v in victims:
                oa.delete(v)
            tcd = (time.time() - tcd) / ND
            ga, gb = oa.export(), ie.export_graph()
            same = (ga["enterpoint"] == gb["enterpoint"]
                    and all(np.array_equal(a_, b_) for a_, b_ in zip(ga["row_ptr"], gb["row_ptr"]))
                    and all(np.array_equal(a_, b_) for a_, b_ in zip(ga["col"], gb["col"])))
            if not same:
                raise SystemExit("single_delete: the graph after %d hnsw_delete calls differs from the oracle's" % ND)
            single_delete = dict(workload="HNSW.NODE.DEL: %d single hnsw_delete calls on the same index" % ND,
                                 gpu_ms=round(1e3 * td, 3), cpu_oracle_ms=round(1e3 * tcd, 3), identical=True)
            log("single hnsw_delete: %.3f ms per call (CPU oracle %.3f ms), graphs identical" % (1e3 * td, 1e3 * tcd))
            oa.close()
            log("single hnsw_add: %.3f ms per call (CPU oracle %.3f ms), graphs identical" % (1e3 * ta, 1e3 * tc))
        ie.close()
        log("exact GPU build of %d nodes: %.1f s, identical to the oracle's: %s" % (NE, te, identical))
    if cfg_is_c2 and extras and not args.no_clustered:
        centers = np.random.default_rng(3).random((64, dim), dtype=np.float32)
        Vc = clustered(N, dim, 3, centers)
        Qc = torch.from_numpy(clustered(B, dim, 4, centers)).to(dev)
        ic = Index("bench-clustered", dim, M, ef, device=local_rank)
        ic.add_batch(Vc, levels=levels, mode="fast")
        for _ in range(2):
            ic.search_batch_device(Qc.data_ptr(), B, k, d_ids.data_ptr(), d_sims.data_ptr(), d_n.data_ptr(), cur.cuda_stream)
        torch.cuda.synchronize()
        tc0 = time.perf_counter()
        for _ in range(5):
            ic.search_batch_device(Qc.data_ptr(), B, k, d_ids.data_ptr(), d_sims.data_ptr(), d_n.data_ptr(), cur.cuda_stream)
        torch.cuda.synchronize()
        tcl = (time.perf_counter() - tc0) / 5
        Vc_dev = torch.from_numpy(Vc).to(dev)
        gtc = brute_force_gt(torch, Vc_dev, Qc, k)
        gotc = d_ids.cpu().numpy().astype(np.int64)
        hit = sum(len(set(a.tolist()) & set(bb.tolist())) for a, bb in zip(gotc, gtc))
        clus = dict(data="64-cluster Gaussian mixture, sigma 0.1 (fast GPU build)", recall_at_10=round(hit / (B * k), 4),
                    value=round(B / tcl, 1), unit="queries/s")
        log("clustered data: recall@%d = %.4f, %.3f ms/step" % (k, hit / (B * k), 1e3 * tcl))
        del Vc_dev, Vc
        ic.close()

    # ---- informational: the compressed serving copies of the same graph (NOT the reference's
    # arithmetic inputs: vectors rounded to bf16 / fp8 e4m3, then the reference's f32 kernel on the stored values) --
    # same kind of timed loop, 1024-query calls round-robin on `cs` streams (the bf16 / fp8 forms of the dim-128 kernel
    # hold three waves per SIMD: four calls in flight fill it)
    bf16 = None
    fp8 = None
    if extras and graph is not None and dim % 32 == 0:
        V_dev = torch.from_numpy(V).to(dev)
        gt_c = brute_force_gt(torch, V_dev, myQ[:B], k)
        del V_dev
        search_now(myQ[:B], B)
        got32 = d_ids.cpu().numpy().astype(np.int64)
        for fmt_name, esz, cs in (("bf16", 2, 4), ("fp8", 1, 4)):
            ib = Index("bench-" + fmt_name, dim, M, ef, device=local_rank)
            gb = dict(graph)
            gb["vectors"] = V
            ib.import_graph(gb)
            ib.set_tuning("compress_" + fmt_name, 1)
            cstreams = (streams + [extra_stream])[:cs]
            cbufs = [(torch.empty((B, k), dtype=torch.int32, device=dev), torch.empty((B, k), dtype=torch.float32, device=dev),
                      torch.empty((B,), dtype=torch.int32, device=dev)) for _ in range(cs)]

            def cstep(i):
                q = myQ[(i % n_qbatches) * B:(i % n_qbatches + 1) * B]
                o_ = cbufs[i % cs]
                ib.search_batch_device(q.data_ptr(), B, k, o_[0].data_ptr(), o_[1].data_ptr(), o_[2].data_ptr(), cstreams[i % cs].cuda_stream)
            for i in range(3 * cs):
                cstep(i)
            torch.cuda.synchronize()
            nbc = 72
            tb0 = time.perf_counter()
            for i in range(nbc):
                cstep(i)
            torch.cuda.synchronize()
            tbc = (time.perf_counter() - tb0) / nbc
            # the same work handed over as ONE call of 16 B queries (the engine's own lanes keep the chunks in flight):
            # independent of how the runtime maps the caller's streams onto hardware queues, which the figure above is not
            # (two of the four streams on one queue halve it)
            Bc = 16 * B
            Qc = torch.from_numpy(np.random.default_rng(5).random((Bc, dim), dtype=np.float32)).to(dev)
            cbig = (torch.empty((Bc, k), dtype=torch.int32, device=dev), torch.empty((Bc, k), dtype=torch.float32, device=dev),
                    torch.empty((Bc,), dtype=torch.int32, device=dev))
            for _ in range(2):
                ib.search_batch_device(Qc.data_ptr(), Bc, k, cbig[0].data_ptr(), cbig[1].data_ptr(), cbig[2].data_ptr(), cur.cuda_stream)
            torch.cuda.synchronize()
            tb1 = time.perf_counter()
            for _ in range(6):
                ib.search_batch_device(Qc.data_ptr(), Bc, k, cbig[0].data_ptr(), cbig[1].data_ptr(), cbig[2].data_ptr(), cur.cuda_stream)
            torch.cuda.synchronize()
            tbig = (time.perf_counter() - tb1) / 6 / 16            # per B queries
            shapes = {"%d calls of %d in flight" % (cs, B): round(B / tbc, 1), "one call of %d at a time" % Bc: round(B / tbig, 1)}
            del Qc, cbig
            tbc = min(tbc, tbig)
            ib.search_batch_device(myQ[:B].data_ptr(), B, k, d_ids.data_ptr(), d_sims.data_ptr(), d_n.data_ptr(), cur.cuda_stream)
            torch.cuda.synchronize()
            gotc = d_ids.cpu().numpy().astype(np.int64)
            byc = B * (n_dist_q * esz * dim + n_ids_q * 4 + 4 * dim + 8 * k)
            ent = dict(value=round(B / tbc, 1), unit="queries/s", ms_per_step=round(1e3 * tbc, 4), calls_in_flight=cs, launch_shapes=shapes,
                       kernel=("specialised dim-128 kernel, %s rows" if ib.last_search_was_lean() else "general kernel, %s rows") % fmt_name,
                       recall_at_10=round(sum(len(set(a.tolist()) & set(b.tolist())) for a, b in zip(gotc, gt_c)) / (B * k), 4),
                       top10_overlap_with_f32=round(sum(len(set(a.tolist()) & set(b.tolist())) for a, b in zip(gotc, got32)) / (B * k), 4),
                       achieved=round(byc / tbc / 1e9, 1), frac=round(byc / tbc / 1e9 / HBM_PEAK_GBS, 4),
                       note="separate mode: vectors stored as %s (%d B/component in the gather), widened exactly, f32 accumulation in the "
                            "reference's order; bit-identical to the reference on the stored (rounded) vectors, not to the f32 headline; "
                            "achieved / frac are on this mode's own algorithmic bytes" % (fmt_name, esz))
            ib.close()
            log("%s copy: %.3f ms/step (%.2f M QPS), recall@10 %.4f" % (fmt_name, 1e3 * tbc, B / tbc / 1e6, ent["recall_at_10"]))
            if fmt_name == "bf16":
                bf16 = ent
            else:
                fp8 = ent

    # ---- one process, every visible GPU (SURVEY 8e "one process, 8 devices"; what a Redis module would use): the
    # C ABI's hnsw_group_* layer -- replicas by peer copies, the host batch split over the members, one host thread
    # per replica.  Informational: runs when this process sees more than one device (HNSW_BENCH_GROUP=1 forces the
    # functional form with a second member on the same device); never `value`.
    group_leg = None
    ndev_vis = torch.cuda.device_count()
    if extras and (ndev_vis > 1 or os.environ.get("HNSW_BENCH_GROUP") == "1"):
        try:
            from redis_hnsw_amd.group import Group
            devs = [d for d in range(ndev_vis) if d != local_rank] if ndev_vis > 1 else [local_rank]
            tg0 = time.perf_counter()
            grp = Group(index, devs)
            t_make = time.perf_counter() - tg0
            Gn = len(grp)
            Qg = np.ascontiguousarray(np.tile(Qall[:8 * B], (Gn, 1))[:8 * B * Gn])
            g_ids, _, _ = grp.search_batch(Qg, k)
            tg = time.perf_counter()
            for _ in range(3):
                g_ids, _, _ = grp.search_batch(Qg, k)
            g_qps = 3 * Qg.shape[0] / (time.perf_counter() - tg)
            same = bool(np.array_equal(g_ids[:Qh.shape[0]], hb_ids))
            group_leg = dict(members=Gn, devices=[local_rank] + devs, create_seconds=round(t_make, 3),
                             batch=int(Qg.shape[0]), value=round(g_qps, 1), unit="queries/s",
                             vs_one_member=round(g_qps / host_qps, 3), answers_equal_single_index=same,
                             note="hnsw_group_search_batch from pageable host memory (8192 queries per member per call); "
                                  "weak scaling against host_buffers")
            grp.close()
            log("single-process group of %d: %.0f QPS (%.2fx one member), identical answers: %s" % (Gn, g_qps, g_qps / host_qps, same))
            if not same:
                raise SystemExit("hnsw_group_search_batch answers differ from the single index")
        except Exception as e:                       # informational leg: report, never fail the bench line
            group_leg = dict(error=repr(e)[:300])
            log("single-process group leg failed: %r" % (e,))

    # ---- roofline of the dominant kernel (k_search) -------------------------------------
    # algorithmic bytes per launch = B x (n_dist*4*dim + n_ids*4 + 4*dim + 8*k)  (SURVEY 8d), the reference's counts
    bytes_per_launch = B * (n_dist_q * 4 * dim + n_ids_q * 4 + 4 * dim + 8 * k)
    in_flight = min(S, args.steps) if args.steps else 1
    per_launch = bytes_per_launch / (kernel_ms * 1e-3) / 1e9
    # `in_flight` launches run at a time; what the chip delivered is their bytes over the timed region's wall
    # time (this rank's), which also pays for the gaps between launches: never above in_flight x per_launch
    overlapped = in_flight * per_launch
    achieved = min(overlapped, bytes_per_launch * args.steps / t_local / 1e9) if args.steps else overlapped
    # HBM traffic per launch: measured IN THIS RUN when rocprofv3 is on PATH (two short re-runs of this command
    # under --pmc FETCH_SIZE / WRITE_SIZE, separate passes; FETCH_SIZE x2 is the gfx950 correction of
    # MI355X_MICROARCH.md, calibrated on this access pattern in profiles/*_pmc_hbm.txt), else the committed figure
    traffic, traffic_source = None, None
    if not args.no_traffic and world == 1 and args.steps:
        traffic, traffic_source = measure_traffic(sys.argv[1:], log, B, bytes_per_launch)
    if traffic is None and not (traffic_source or "").startswith("REJECTED"):
        try:
            tj = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
            want = dict(nodes=N, dim=dim, M=M, ef=ef, k=k, batch=B, graph=mode, streams=S)
            for ent in tj.get("entries", []):
                cfg_e = ent.get("config") or {}
                if cfg_e and all(cfg_e.get(kk) == vv for kk, vv in want.items()) and "k_search_hbm_bytes_per_launch" in ent:
                    traffic = ent["k_search_hbm_bytes_per_launch"]
                    traffic_source = "profiles/traffic.json@%s (not measured in this run)" % ent.get("commit", "?")
                    break
        except (OSError, ValueError, KeyError, TypeError):
            pass
    if per_rank is not None and args.steps:
        # every rank's own roofline fraction: its steps' algorithmic bytes (rank 0's per-query figure; the ranks' queries
        # are draws of the same distribution) over its own wall time of the timed region
        per_rank["roofline_frac"] = [round(bytes_per_launch * args.steps / sec_ / 1e9 / HBM_PEAK_GBS, 4) for sec_ in per_rank["seconds"]]
    roofline = dict(bound="hbm", achieved=round(achieved, 1), peak=HBM_PEAK_GBS, unit="GB/s",
                    frac=round(achieved / HBM_PEAK_GBS, 4), traffic=traffic, traffic_source=traffic_source,
                    peak_hbm_spec=HBM_PEAK_GBS, hbm_measured_copy=HBM_MEASURED_GBS,
                    hbm_measured_gather=measured_gather(4 * dim, N * dim * 4 / 1e6),
                    frac_of_measured_copy=round(achieved / HBM_MEASURED_GBS, 4),
                    what="ALGORITHMIC bytes (the reference's n_dist x 4 dim + n_ids x 4 + query + results per query) over time, "
                         "against the 8 TB/s HBM3E spec.  It is not a DRAM-level rate: the counters behind `traffic` sit at the "
                         "L2's memory side and include Infinity-Cache hits (256 MB of cache in front of this workload's %.2f GB vector "
                         "matrix), which is how the figure can exceed what a streaming copy sustains from DRAM "
                         "(hbm_measured_copy; hbm_measured_gather is the same access pattern without the walk: random %d-byte "
                         "rows, profiles/r5_gather_bw.txt)" % (N * dim * 4 / 1e9, 4 * dim),
                    kernel="k_search", kernel_ms=round(kernel_ms, 4),
                    kernel_ms_median=None if kernel_ms_median is None else round(kernel_ms_median, 4),
                    kernel_ms_p90=None if kernel_ms_p90 is None else round(kernel_ms_p90, 4),
                    launches_in_flight=in_flight,
                    per_launch_gbs=round(per_launch, 1), in_flight_x_per_launch_gbs=round(overlapped, 1),
                    note="achieved = algorithmic bytes of the timed steps / wall time of the timed region (launches overlap: "
                         "launches_in_flight x algorithmic_bytes_per_launch / kernel_ms, the average duration of one launch "
                         "by HIP events per stream, is the upper figure in_flight_x_per_launch_gbs)",
                    algorithmic_bytes_per_launch=int(bytes_per_launch),
                    n_dist_per_query=round(n_dist_q, 1), n_ids_per_query=round(n_ids_q, 1),
                    n_expand_per_query=round(n_exp_q, 1),
                    re_evaluated_fraction=round(max(redo, 0.0), 5),
                    lone_launch_1024=lone,
                    single_launch_4096=big)

    # ---- CPU baseline: the oracle (C restatement of the Rust path), bounded sample -------
    cpu = None
    c1 = None
    c2s = None
    cfg_name_early = {(1_000_000, 128, 16, 200): "C2", (1_000_000, 768, 32, 400): "C3", (10_000_000, 128, 16, 200): "C4"}.get((N, dim, M, ef), "this")
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
        cores = usable_cores()
        o.search_batch(Qs, k, threads=cores)               # starts the persistent workers
        tc = time.perf_counter()
        reps = 0
        while True:
            o.search_batch(Qs, k, threads=cores)
            reps += 1
            if time.perf_counter() - tc > args.cpu_seconds / 2:
                break
        tN = time.perf_counter() - tc
        # tie census of the bench's own queries (oracle/hnsw_oracle.c: hnsw_oracle_tie_census): the reference orders
        # SimPair by sim alone and leaves equal sims to std's BinaryHeap; engine and oracle break them by id.  Only a
        # query where a DECISION met equal sims (core.rs:635, :657) or the k + 1 nearest hold equal sims can be answered
        # differently by the Rust binary; those few are re-run in the Rust binary's own order (std's heap restated,
        # pinned against the transcription) and compared with what the ENGINE answered
        tq0 = time.perf_counter()
        n_cen = 0
        cen = dict(queries=0, stop_test_ties=0, accept_test_ties=0, queries_with_decision_tie=0, queries_with_answer_tie=0,
                   queries_with_any_tie=0)
        tied_q = []
        Qcen = Qall[:n_qbatches * B]
        while n_cen < Qcen.shape[0] and time.perf_counter() - tq0 < max(args.cpu_seconds, 6.0):
            for qi in range(n_cen, min(n_cen + 256, Qcen.shape[0])):
                c_ = o.tie_census(Qcen[qi:qi + 1], k)
                for key_ in cen:
                    cen[key_] += c_[key_]
                if c_["queries_with_any_tie"]:
                    tied_q.append(qi)
            n_cen = min(n_cen + 256, Qcen.shape[0])
        differ = 0
        for qi in tied_q:
            rid, rsim = o.search_std_heap(Qcen[qi], k)
            gid, gsim, gn = index.search_batch(Qcen[qi:qi + 1], k)
            gid, gsim = gid[0, :int(gn[0])], gsim[0, :int(gn[0])]
            differ += not (np.array_equal(gid, rid) and np.array_equal(gsim.view(np.uint32), rsim.view(np.uint32)))
        cen.update(tied_queries_answered_differently_in_the_rust_heap_order=differ,
                   note="of the first %d queries of this bench: decisions that met EQUAL similarities of two different nodes "
                        "(the only place the reference's sim-only order and the (sim, id) order can part); every tied query was "
                        "re-run in std::collections::BinaryHeap's own order (restated, tests/golden/tiecase_rust_lattice.npz) "
                        "and compared with the engine's answer, ids and similarity bits" % n_cen)
        log("tie census of %d queries: %d with a decision tie, %d with an answer tie; %d answered differently in the Rust heap order" % (
            n_cen, cen["queries_with_decision_tie"], cen["queries_with_answer_tie"], differ))
        cpu = dict(value=round(done / t1, 1), unit="queries/s", cores=1, kind="port", tie_census=cen,
                   sample="%d queries of the same 1024-query batch on the same graph, 1 thread, %.1f s" % (done, t1),
                   cpu_model=cpu_model(),
                   all_cores=dict(value=round(reps * B / tN, 1), cores=cores, visible_cpus=os.cpu_count(),
                                  sample="%d passes over the batch, persistent workers with per-thread scratch; "
                                         "cores = affinity mask capped by the cgroup CPU quota" % reps))
        # ---- one HNSW.SEARCH per call on THIS index (the command's shape, src/lib.rs:484), beside the oracle's
        c2s = None
        if extras:
            Q1 = Qall[:200]
            for q in Q1[:20]:
                index.search_knn(q, k)
            same = all([r.id for r in index.search_knn(q, k)] == o.search(q, k)[0].tolist() for q in Q1[:20])
            tq = time.perf_counter()
            for q in Q1:
                index.search_knn(q, k)
            t_g = (time.perf_counter() - tq) / len(Q1)
            tq = time.perf_counter()
            for q in Q1:
                o.search(q, k)
            t_c = (time.perf_counter() - tq) / len(Q1)
            c2s = dict(workload="%s index, one query per hnsw_search call (host buffers)" % cfg_name_early,
                       gpu_us=round(1e6 * t_g, 1), cpu_oracle_us=round(1e6 * t_c, 1), identical=bool(same),
                       two_wave_kernel=bool(index.last_search_was_duo()))
            log("one query per call on this index: %.1f us (CPU oracle %.1f us), identical: %s" % (1e6 * t_g, 1e6 * t_c, same))
        o.close()
        # ---- C1 (BASELINE config 1): 10k x 128, M=5, ef=200, k=10, ONE query per call (the shape of a
        # HNSW.SEARCH command): hnsw_search latency, host buffers in and out, beside the oracle's
        if extras:
            n1, m1 = 10_000, 5
            lv1 = draw_levels(n1, m1, 7)
            o1 = oracle.OracleIndex(dim, m1, ef)
            o1.add_batch(V[:n1], lv1)
            g1 = Index("c1", dim, m1, ef, device=local_rank)
            g1.import_graph(o1.export())
            Q1 = Qall[:300]
            same = True
            for q in Q1[:40]:
                a = g1.search_knn(q, k)
                ids1, sims1 = o1.search(q, k)
                same = same and [r.id for r in a] == ids1.tolist()
            tq = time.perf_counter()
            for q in Q1:
                g1.search_knn(q, k)
            t_g = (time.perf_counter() - tq) / len(Q1)
            tq = time.perf_counter()
            for q in Q1:
                o1.search(q, k)
            t_c = (time.perf_counter() - tq) / len(Q1)
            c1 = dict(workload="C1: 10k x 128, M=5, ef=200, k=10, one query per hnsw_search call (host buffers)",
                      gpu_us=round(1e6 * t_g, 1), cpu_oracle_us=round(1e6 * t_c, 1), identical=bool(same),
                      two_wave_kernel=bool(g1.last_search_was_duo()))
            g1.close(); o1.close()

    known = {(1_000_000, 128, 16, 200, 10, 1024): "C2", (1_000_000, 768, 32, 400, 100, 4096): "C3",
             (10_000_000, 128, 16, 200, 10, 1024): "C4", (10_000, 128, 5, 200, 10, 1): "C1"}
    cfg_name = known.get((N, dim, M, ef, k, B), "custom")
    if cfg_name == "C2" and args.graph == "exact":
        cfg_name = "C5"                                  # the same index, BUILT on the GPU in the reference's order first
    qps = world * B * args.steps / t_wall
    out = {
        "metric": "HNSW.SEARCH QPS + recall@10, 1M x 128 f32, ef=200" if cfg_name == "C2" else
                  "HNSW.SEARCH QPS + recall@%d (%s)" % (k, cfg_name),
        "value": round(qps, 1), "unit": "queries/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(1e3 * t_wall / args.steps, 4), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "%s: %d nodes x dim %d, M=%d, ef=%d, k=%d, batch=%d queries/GPU, uniform[0,1) f32, replicated index"
                               % (cfg_name, N, dim, M, ef, k, B),
                   "nodes": N, "dim": dim, "M": M, "ef": ef, "k": k, "batch": B, "graph": mode, "graph_desc": graph_desc,
                   "steps_in_flight": S, "hw_queues": int(os.environ.get("GPU_MAX_HW_QUEUES", "4")),
                   "launch_concurrency": args.launch_concurrency or "observed by the engine", "pipeline": pipe,
                   "prewarm_launches": PREWARM,
                   "parallelism": "replica x%d, query batch sharded%s" % (
                       world, " (ranks share one device, gloo: functional check only)" if one_device and world > 1 else "")},
        "gather_verified": gather_ok,
        "rccl_world": world if (world > 1 and backend == "nccl") else None,
        "collective_backend": backend if world > 1 else None,
        "per_rank": per_rank, "index_replication": replication, "topk_exchange": gather_cmp,
        "recall_at_%d" % k: None if recall is None else round(recall, 4),
        "build_seconds": None if t_build is None else round(t_build, 2),
        "host_buffers_qps": round(host_qps, 1), "host_buffers": host, "single_process_group": group_leg, "device_call": dev_calls,
        "gpu_fast_build": fast_build, "gpu_exact_build": exact_build, "single_add": single_add, "single_delete": single_delete,
        "clustered": clus,
        "bf16_storage_mode": bf16, "fp8_storage_mode": fp8,
        "c1_single_query": c1, "single_query_on_this_index": c2s,
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
