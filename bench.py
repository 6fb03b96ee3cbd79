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
FIXTURES = {(1_000_000, 128, 16, 200): os.path.join(ROOT, "data", "c2_ref_graph_1m.npz"),
            # BASELINE config 3's shape as reference-order graphs: 100 k built by the CPU oracle; 300 k built on the GPU by the windowed
            # reference-order build (scripts/build_c3_ref_graph_gpu.py: 604 s), its 100 k prefix checked row for row against the former
            (100_000, 768, 32, 400): os.path.join(ROOT, "data", "c3_ref_graph_100k.npz"),
            (300_000, 768, 32, 400): os.path.join(ROOT, "data", "c3_ref_graph_300k.npz")}
FIXTURE_50K = os.path.join(ROOT, "data", "c2_ref_graph_50k.npz")
# the reference's own work counters of the C2 build at several prefix sizes (tests/fixtures/make_ref_counters.py)
REF_INSERT_COUNTERS = os.path.join(ROOT, "data", "c2_ref_insert_counters.json")
# BASELINE.json's configurations by name (bench.py --workload c3); c1 is timed inside every default run (c1_single_query)
WORKLOADS = {
    "c1": dict(nodes=10_000, dim=128, m=5, ef=200, k=10, batch=1),
    "c2": dict(nodes=1_000_000, dim=128, m=16, ef=200, k=10, batch=1024),
    "c3": dict(nodes=1_000_000, dim=768, m=32, ef=400, k=100, batch=4096),
    "c3ref": dict(nodes=300_000, dim=768, m=32, ef=400, k=100, batch=4096),     # C3's launch shape on a REFERENCE-ORDER graph (fixture)
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


class Bench:
    """One run of the bench.  What every leg reads -- the index, the resident queries, the streams, the figures earlier legs
    measured -- lives on the instance (`s`); each leg is a method that leaves its part of the JSON line there."""

    def __init__(self, args):
        s = self
        s.args = args
        import torch as _torch
        s.torch = _torch
        s.rank = int(os.environ.get("RANK", "0"))
        s.world = int(os.environ.get("WORLD_SIZE", "1"))
        s.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        if not s.torch.cuda.is_available():
            raise SystemExit("bench.py needs a GPU: the engine has no CPU path")
        s.one_device = bool(os.environ.get("HNSW_BENCH_ONE_DEVICE"))
        if s.one_device:
            s.local_rank = 0
        s.torch.cuda.set_device(s.local_rank)
        s.dist = None
        # HNSW_BENCH_BACKEND=gloo + HNSW_BENCH_ONE_DEVICE=1 let the N>1 control flow be exercised on a
        # single-GPU box (every rank on cuda:0, collectives on host copies); the driver's runs use RCCL.
        s.backend = os.environ.get("HNSW_BENCH_BACKEND", "nccl")
        s.coll_dev = s.torch.device("cuda", s.local_rank) if s.backend == "nccl" else s.torch.device("cpu")
        if s.world > 1:
            import torch.distributed as _dist
            s.dist = _dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            if s.backend == "nccl":
                s.dist.init_process_group("nccl", rank=s.rank, world_size=s.world, device_id=s.torch.device("cuda", s.local_rank))
            else:
                s.dist.init_process_group(s.backend, rank=s.rank, world_size=s.world)

        def log(msg):
            if s.rank == 0:
                print("[bench %.1fs] %s" % (time.time() - s.t00, msg), file=sys.stderr, flush=True)

        s.log = log
        s.t00 = time.time()
        from redis_hnsw_amd import Index as _Index
        from redis_hnsw_amd import shard as _shard
        s.Index = _Index
        s.shard = _shard
        s.N, s.dim, s.M, s.ef, s.k, s.B = s.args.nodes, s.args.dim, s.args.m, s.args.ef, s.args.k, s.args.batch
        s.S = max(1, s.args.streams)
        s.cfg_is_c2 = (s.N, s.dim, s.M, s.ef, s.k, s.B) == (1_000_000, 128, 16, 200, 10, 1024)
        s.t0 = time.time()
        s.V = np.random.default_rng(1).random((s.N, s.dim), dtype=np.float32)
        s.n_qbatches = 8
        s.Qall = np.random.default_rng(2).random((s.n_qbatches * s.B * s.world, s.dim), dtype=np.float32)
        s.levels = draw_levels(s.N, s.M, 7)
        s.first_contact()

    def first_contact(self):
        """N > 1: within the first seconds every rank says what it sees -- its device, the RCCL world, which peers it can
        reach directly, and how the two ways of moving one step's [2, B, k] result block compare (all-gather over RCCL vs a
        plain D2H copy) -- and the run FAILS if the world is not the --gpus it was asked for.  The driver's multi-GPU node is
        the first hardware this path meets: the report is what tells a broken fabric from a slow one."""
        s = self
        want = int(getattr(s.args, "gpus", 1) or 1)
        if want != s.world:
            raise SystemExit("bench.py --gpus %d but WORLD_SIZE is %d: launch with torch.distributed.run --nproc-per-node %d (rank %d)"
                             % (want, s.world, want, s.rank))
        s.first_contact_rows = None
        if s.world == 1:
            return
        t = s.torch
        if s.dist.get_world_size() != s.world or s.dist.get_rank() != s.rank:
            raise SystemExit("process group disagrees with the environment: world %d/%d rank %d/%d"
                             % (s.dist.get_world_size(), s.world, s.dist.get_rank(), s.rank))
        nd = t.cuda.device_count()
        if not s.one_device and nd < s.world:
            raise SystemExit("rank %d: %d ranks but only %d visible GPUs (one process per GPU; HNSW_BENCH_ONE_DEVICE=1 only for dry control-flow runs)"
                             % (s.rank, s.world, nd))
        dev = t.device("cuda", s.local_rank)
        peers = None
        if not s.one_device and nd > 1:
            try:
                peers = [int(t.cuda.can_device_access_peer(s.local_rank, d)) if d != s.local_rank else 1 for d in range(nd)]
            except (RuntimeError, AssertionError):
                peers = None
        blk = t.zeros((2, s.B, s.k), dtype=t.int32, device=dev if s.backend == "nccl" else "cpu")
        outb = t.empty((s.world * 2, s.B, s.k), dtype=t.int32, device=blk.device)
        times = {}
        for name, fn in (("all_gather_us", lambda: s.dist.all_gather_into_tensor(outb, blk)),
                         ("d2h_copy_us", lambda: blk.cpu() if blk.is_cuda else blk.clone())):
            fn()
            t.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(5):
                fn()
            t.cuda.synchronize()
            times[name] = round(1e6 * (time.perf_counter() - t0) / 5, 1)
        mine = dict(first_contact=True, rank=s.rank, local_rank=s.local_rank, device=t.cuda.get_device_name(s.local_rank),
                    visible_gpus=nd, backend=s.backend, rccl_world=s.dist.get_world_size() if s.backend == "nccl" else None,
                    peer_access_row=peers, result_block_bytes=int(blk.numel() * 4), **times)
        rows = [None] * s.world
        s.dist.all_gather_object(rows, mine)
        if s.rank == 0:
            for r_ in rows:
                print("[bench first-contact] " + json.dumps(r_), file=sys.stderr, flush=True)
            if s.backend == "nccl" and any(r_["rccl_world"] != want for r_ in rows):
                raise SystemExit("rccl_world != --gpus %d on some rank" % want)
        s.first_contact_rows = rows

    def build_index(self):
        """the graph the headline runs on (--graph): rank 0 imports / builds it, the replicas receive it over RCCL"""
        s = self
        # ---- the graph -------------------------------------------------------------------
        s.fixture = FIXTURES.get((s.N, s.dim, s.M, s.ef))
        s.mode = s.args.graph
        if s.mode == "auto":
            s.mode = "reference" if s.fixture and os.path.exists(s.fixture) else "fast"
        if s.mode == "reference" and not (s.fixture and os.path.exists(s.fixture)):
            raise SystemExit("--graph reference needs %s (python tests/fixtures/make_ref_graph.py --out ...)" % s.fixture)
        s.index = s.Index("bench", s.dim, s.M, s.ef, device=s.local_rank)
        if s.args.launch_concurrency:
            s.index.set_tuning("launch_concurrency", s.args.launch_concurrency)
        if s.args.waves_per_cu:
            s.index.set_tuning("waves_per_cu", s.args.waves_per_cu)
        for kv in s.args.tuning:
            key, val = kv.split("=")
            s.index.set_tuning(key, int(val))
        s.graph = None
        s.t_build = None
        s.replication = None
        s.graph_desc = {"reference": "reference-order (serial core.rs:489-599 order; fixture built by the CPU oracle, imported with hnsw_import)",
                      "exact": "reference-order, built on the GPU (hnsw_add_batch mode 0)",
                      "fast": "batched GPU build (hnsw_add_batch mode 1; NOT the reference's graph)"}[s.mode]
        s.replicate = s.world > 1 and os.environ.get("HNSW_BENCH_REPLICATE", "1") != "0"
        if s.mode == "reference":
            # rank 0 imports the fixture; the replicas receive the index over RCCL (HNSW_BENCH_REPLICATE=0: every rank
            # reads the fixture itself instead)
            if s.rank == 0 or not s.replicate:
                s.graph, s.oracle_build_s = load_graph_fixture(s.fixture, s.V)
                if "built_by" in np.load(s.fixture).files:
                    s.graph_desc = ("reference-order (serial core.rs:489-599 order; fixture built on the GPU by hnsw_add_batch mode 0, its first "
                                    "100 k nodes checked row for row against the CPU oracle's serial build; imported with hnsw_import)")
                tb = time.time()
                s.index.import_graph(s.graph)
                s.torch.cuda.synchronize()
                s.log("imported the reference-order graph (%d nodes) in %.2f s" % (s.N, time.time() - tb))
        else:
            if s.rank == 0:
                tb = time.time()
                s.index.add_batch(s.V, levels=s.levels, mode=s.mode)
                s.torch.cuda.synchronize()
                s.t_build = time.time() - tb
                s.log("built %d nodes in %.2f s (%s)" % (s.N, s.t_build, s.mode))
                if s.world == 1 and not s.args.no_cpu_baseline:
                    s.graph = s.index.export_graph(with_vectors=False)
        if s.replicate or (s.world > 1 and s.mode != "reference"):
            # one-time index distribution: rank 0's tables straight out of its HBM into every replica's
            # -- vectors included -- as device-pointer broadcasts over RCCL (host-staged when the ranks share one
            # device in the gloo functional mode)
            tb = time.time()
            via = "device" if s.backend == "nccl" else "host"
            if via == "device":
                # the collective has never carried raw engine pointers on this node before: prove it on 4 KB first
                probe = s.torch.arange(1024, dtype=s.torch.int32, device=s.torch.device("cuda", s.local_rank)) * (1 if s.rank == 0 else 0)
                try:
                    s.dist.broadcast(s.shard.device_bytes(probe.data_ptr(), 4096, s.torch.device("cuda", s.local_rank)), src=0)
                    s.torch.cuda.synchronize()
                    ok_probe = bool((probe == s.torch.arange(1024, dtype=s.torch.int32, device=probe.device)).all().item())
                except RuntimeError as e:                    # pragma: no cover (needs a multi-GPU node)
                    s.log("RCCL broadcast on a raw device pointer failed (%s): host-staged replication instead" % (e,))
                    ok_probe = False
                flag = s.torch.tensor([1 if ok_probe else 0], dtype=s.torch.int32, device=s.torch.device("cuda", s.local_rank))
                s.dist.all_reduce(flag, op=s.dist.ReduceOp.MIN)
                if int(flag.item()) == 0:
                    via = "host"
            moved = s.shard.replicate_index(s.dist, s.index, src=0, device=s.torch.device("cuda", s.local_rank), via=via)
            s.dist.barrier()
            t_repl = time.time() - tb
            s.log("replicated the index to %d ranks: %.2f GB per replica in %.2f s (%s)" % (
                s.world, moved / 1e9, t_repl, "RCCL, HBM to HBM" if via == "device" else "host-staged, %s" % s.backend))
            s.replication = dict(bytes_per_replica=int(moved), seconds=round(t_repl, 3),
                               transport="rccl broadcast of device pointers" if via == "device" else "%s, host-staged" % s.backend)


    def stage_inputs(self):
        """queries, result buffers and streams resident on the device; every replica must answer like rank 0's index"""
        s = self
        # ---- device-resident inputs/outputs ---------------------------------------------
        s.dev = s.torch.device("cuda", s.local_rank)
        s.myQ = s.torch.from_numpy(s.Qall[s.rank * s.n_qbatches * s.B:(s.rank + 1) * s.n_qbatches * s.B]).to(s.dev)
        s.streams = [s.torch.cuda.Stream() for _ in range(s.S)]
        s.extra_stream = s.torch.cuda.Stream()               # a fourth one for the bf16 leg (created now: streams made later, once more
        #                                                  than GPU_MAX_HW_QUEUES exist, share a hardware queue and serialise)
        # ids and similarities share one buffer so that a single all-gather moves both
        s.bufs = [s.torch.empty((2, s.B, s.k), dtype=s.torch.int32, device=s.dev) for _ in range(s.S)]
        s.d_ns = [s.torch.empty((s.B,), dtype=s.torch.int32, device=s.dev) for _ in range(s.S)]
        s.d_out = s.bufs[0]
        s.d_ids, s.d_sims, s.d_n = s.d_out[0], s.d_out[1].view(s.torch.float32), s.d_ns[0]
        s.cur = s.torch.cuda.current_stream()
        if s.replication is not None:
            # every replica must answer like rank 0's index: the same 64 queries everywhere, ids + similarity bits gathered
            chk_q = s.torch.from_numpy(s.Qall[:64]).to(s.dev)
            chk = s.torch.empty((2, 64, s.k), dtype=s.torch.int32, device=s.dev)
            chk_n = s.torch.empty((64,), dtype=s.torch.int32, device=s.dev)
            s.index.search_batch_device(chk_q.data_ptr(), 64, s.k, chk[0].data_ptr(), chk[1].data_ptr(), chk_n.data_ptr(), s.cur.cuda_stream)
            s.torch.cuda.synchronize()
            allc = s.shard.gather_packed(s.dist, chk if s.backend == "nccl" else chk.cpu(), s.world).cpu().numpy()
            same = all(np.array_equal(allc[0], allc[r_]) for r_ in range(1, s.world))
            s.replication["replicas_answer_identically"] = bool(same)
            if not same:
                raise SystemExit("index replication: a replica answers differently from rank 0")

        def peer_report():
            """what this rank's device can reach directly (xGMI peer access), for the N > 1 line"""
            nd = s.torch.cuda.device_count()
            if s.one_device or nd < 2:
                return None
            try:
                return [int(d) for d in range(nd) if d != s.local_rank and s.torch.cuda.can_device_access_peer(s.local_rank, d)]
            except (RuntimeError, AssertionError):
                return None
        s.peer_report = peer_report

    def dry_run(self):
        """--dry: stop once the index is in place on every rank; prints one JSON line"""
        s = self
        info = s.index.info()
        mine = dict(rank=s.rank, device=s.local_rank, nodes=int(info.node_count), hbm_bytes=int(info.hbm_bytes), peers=s.peer_report())
        objs = [mine]
        if s.world > 1:
            objs = [None] * s.world
            s.dist.all_gather_object(objs, mine)
        if s.rank == 0:
            print(json.dumps({"dry": True, "n_gpus": s.world, "graph": s.mode, "collective_backend": s.backend if s.world > 1 else None,
                              "rccl_world": s.world if (s.world > 1 and s.backend == "nccl") else None,
                              "index_replication": s.replication, "ranks": objs,
                              "setup_seconds": round(time.time() - s.t0, 1)}))
        if s.world > 1:
            s.dist.barrier()
            s.dist.destroy_process_group()
        return


    def timed_region(self):
        """warm-up, then exactly --steps steps between barriers (max over ranks): BASELINE's metric.  True = --only-timed, nothing else to do"""
        s = self
        if s.world > 1:
            s.comm_stream = s.torch.cuda.Stream()
            s.gouts = [s.torch.empty((s.world * 2, s.B, s.k), dtype=s.torch.int32, device=s.coll_dev) for _ in range(s.S)]
            s.ready = [s.torch.cuda.Event() for _ in range(s.S)]
            s.gathered = [s.torch.cuda.Event() for _ in range(s.S)]

        def step(i):
            q = s.myQ[(i % s.n_qbatches) * s.B:(i % s.n_qbatches + 1) * s.B]
            s_ = i % s.S
            st, buf = s.streams[s_], s.bufs[s_]
            if s.world > 1 and i >= s.S:
                st.wait_event(s.gathered[s_])              # the gather that last read this buffer is done
            s.index.search_batch_device(q.data_ptr(), s.B, s.k, buf[0].data_ptr(), buf[1].data_ptr(), s.d_ns[s_].data_ptr(),
                                      st.cuda_stream)
            if s.world > 1:
                s.ready[s_].record(st)
                with s.torch.cuda.stream(s.comm_stream):
                    s.comm_stream.wait_event(s.ready[s_])
                    # the path's one real exchange: gather every shard's top-k (host copies in the gloo test mode)
                    s.shard.gather_packed(s.dist, buf if s.backend == "nccl" else buf.cpu(), s.world, s.gouts[s_])
                    s.gathered[s_].record(s.comm_stream)

        s.step = step
        s.log("inputs resident; warm-up")
        # the clocks of an idle GPU need more than a few half-millisecond launches to settle: a fixed number of
        # untimed launches first (reported as config.prewarm_launches), then the W warm-up steps asked for
        s.PREWARM = 48
        for i in range(s.PREWARM):
            s.step(i)
        s.torch.cuda.synchronize()
        for i in range(max(s.args.warmup, 0)):
            s.step(s.PREWARM + i)
        s.torch.cuda.synchronize()
        s.index.reset_counters()
        if s.world > 1:
            s.dist.barrier()
        s.torch.cuda.synchronize()
        ev0 = [s.torch.cuda.Event(enable_timing=True) for _ in range(s.S)]
        ev1 = [s.torch.cuda.Event(enable_timing=True) for _ in range(s.S)]
        per_stream = [len(range(s_, s.args.steps, s.S)) for s_ in range(s.S)]
        marks = [[] for _ in range(s.S)]                   # one event after every launch, on its stream
        t_start = time.perf_counter()
        for s_ in range(s.S):
            ev0[s_].record(s.streams[s_])
        for i in range(s.args.steps):
            s.step(s.PREWARM + s.args.warmup + i)
            if s.world == 1:
                e_ = s.torch.cuda.Event(enable_timing=True)
                e_.record(s.streams[i % s.S])
                marks[i % s.S].append(e_)
        for s_ in range(s.S):
            ev1[s_].record(s.streams[s_])
        s.torch.cuda.synchronize()
        if s.world > 1:
            s.dist.barrier()
        s.t_wall = time.perf_counter() - t_start
        s.t_local = s.t_wall                              # this rank's own time (the roofline is rank 0's kernel)
        if s.world > 1:
            tt = s.torch.tensor([s.t_wall], dtype=s.torch.float64, device=s.coll_dev)
            s.dist.all_reduce(tt, op=s.dist.ReduceOp.MAX)
            s.t_wall = float(tt.item())
        s.per_rank = None
        s.gather_cmp = None
        if s.world > 1:
            # what every rank saw (the driver computes efficiency from `value`; these show the spread behind it)
            objs = [None] * s.world
            s.dist.all_gather_object(objs, dict(rank=s.rank, seconds=s.t_local, qps=s.B * s.args.steps / s.t_local, peers=s.peer_report()))
            s.per_rank = dict(qps=[round(o_["qps"], 1) for o_ in objs], seconds=[round(o_["seconds"], 5) for o_ in objs],
                            peer_access=[o_["peers"] for o_ in objs],
                            ms_per_step_min=round(1e3 * min(o_["seconds"] for o_ in objs) / s.args.steps, 4),
                            ms_per_step_max=round(1e3 * max(o_["seconds"] for o_ in objs) / s.args.steps, 4))
            # the per-step exchange as an RCCL all-gather of the packed [2,B,k] block vs the alternative
            # without a collective -- every rank copies its own block to pinned host memory (hipMemcpyAsync D2H)
            reps_g = 50
            hostbuf = s.torch.empty((2, s.B, s.k), dtype=s.torch.int32).pin_memory()
            gsrc = s.bufs[0] if s.backend == "nccl" else s.bufs[0].cpu()
            for _ in range(5):
                s.shard.gather_packed(s.dist, gsrc, s.world, s.gouts[0])
                hostbuf.copy_(s.bufs[0], non_blocking=True)
            s.torch.cuda.synchronize()
            s.dist.barrier()
            tg = time.perf_counter()
            for _ in range(reps_g):
                s.shard.gather_packed(s.dist, gsrc, s.world, s.gouts[0])
            s.torch.cuda.synchronize()
            t_ag = (time.perf_counter() - tg) / reps_g
            td = time.perf_counter()
            for _ in range(reps_g):
                hostbuf.copy_(s.bufs[0], non_blocking=True)
            s.torch.cuda.synchronize()
            t_d2h = (time.perf_counter() - td) / reps_g
            tt2 = s.torch.tensor([t_ag, t_d2h], dtype=s.torch.float64, device=s.coll_dev)
            s.dist.all_reduce(tt2, op=s.dist.ReduceOp.MAX)
            s.gather_cmp = dict(bytes_per_rank=int(2 * s.B * s.k * 4), allgather_us=round(1e6 * float(tt2[0]), 1),
                              d2h_to_pinned_us=round(1e6 * float(tt2[1]), 1), transport=s.backend,
                              note="back to back, nothing overlapped; in the timed loop the gather of step i runs on its own "
                                   "stream under the search of step i+1")
        # average duration of ONE k_search launch: launches on a stream run back to back, so the stream's
        # elapsed time / its launches (what rocprofv3 --kernel-trace reports as the kernel's average)
        s.kernel_ms = float(np.mean([ev0[s_].elapsed_time(ev1[s_]) / per_stream[s_] for s_ in range(s.S) if per_stream[s_]]))
        # per-launch durations (launches on a stream run back to back): robust against a short timed region
        per_launch = []
        for s_ in range(s.S):
            prev = ev0[s_]
            for e_ in marks[s_]:
                per_launch.append(prev.elapsed_time(e_))
                prev = e_
        s.kernel_ms_median = float(np.median(per_launch)) if per_launch else None
        s.kernel_ms_p90 = float(np.percentile(per_launch, 90)) if per_launch else None
        s.log("timed region done: %.3f ms/step, %.3f ms per launch with %d in flight" % (1e3 * s.t_wall / s.args.steps, s.kernel_ms, s.S))
        if s.args.only_timed:
            if s.world > 1:
                s.dist.barrier()
                s.dist.destroy_process_group()
            return True
        s.sc, _ = s.index.counters()

        s.gather_ok = None
        if s.world > 1 and s.args.verify_gather:
            # every rank searches its first batch, one gather; rank 0 repeats all of them on its own replica
            s.index.search_batch_device(s.myQ[:s.B].data_ptr(), s.B, s.k, s.d_ids.data_ptr(), s.d_sims.data_ptr(), s.d_n.data_ptr(),
                                      s.cur.cuda_stream)
            s.torch.cuda.synchronize()
            got = s.shard.gather_packed(s.dist, s.d_out if s.backend == "nccl" else s.d_out.cpu(), s.world).cpu().numpy()
            if s.rank == 0:
                s.gather_ok = True
                for r in range(s.world):
                    q = s.torch.from_numpy(s.Qall[r * s.n_qbatches * s.B:r * s.n_qbatches * s.B + s.B]).to(s.dev)
                    s.index.search_batch_device(q.data_ptr(), s.B, s.k, s.d_ids.data_ptr(), s.d_sims.data_ptr(), s.d_n.data_ptr(),
                                              s.cur.cuda_stream)
                    s.torch.cuda.synchronize()
                    s.gather_ok = s.gather_ok and bool(np.array_equal(got[r], s.d_out.cpu().numpy()))
                s.log("gathered == unsharded: %s" % s.gather_ok)

        return False

    def exact_counters_and_recall(self):
        """the reference's work counters of the timed batches (the algorithmic bytes of the roofline); recall@k against brute force"""
        s = self
        def search_now(q_dev, nq):
            s.index.search_batch_device(q_dev.data_ptr(), nq, s.k, s.d_ids.data_ptr(), s.d_sims.data_ptr(), s.d_n.data_ptr(), s.cur.cuda_stream)
            s.torch.cuda.synchronize()

        s.search_now = search_now
        # ---- exact work counters of the timed batches (one launch at a time, nothing forgotten) -------
        # With several launches sharing the CUs the LDS visited table is the bounded one: a
        # re-met node may be evaluated twice, so the timed region's n_dist can exceed the reference's.  The
        # algorithmic bytes are the reference's: count them in a pass where the table holds everything.
        s.index.set_tuning("launch_concurrency", 1)
        s.index.set_tuning("waves_per_cu", 4)
        s.index.set_tuning("visited_bounded", 0)           # the exact set: LDS table, HBM table beyond it
        s.index.reset_counters()
        nb_exact = min(s.n_qbatches, s.args.steps, max(1, 8192 // s.B))
        for b in range(nb_exact):
            s.search_now(s.myQ[b * s.B:(b + 1) * s.B], s.B)
        sx, _ = s.index.counters()
        s.index.set_tuning("visited_bounded", 1)
        s.index.set_tuning("waves_per_cu", s.args.waves_per_cu or 8)
        s.index.set_tuning("launch_concurrency", s.args.launch_concurrency)
        s.n_dist_q, s.n_ids_q, s.n_exp_q = sx.n_dist / (nb_exact * s.B), sx.n_ids / (nb_exact * s.B), sx.n_expand / (nb_exact * s.B)
        s.redo = s.sc.n_dist / (s.args.steps * s.B) / s.n_dist_q - 1.0 if s.args.steps else 0.0
        # ---- the ENGINE's own tie census of the same batches (hnsw_get_tie_counters): decisions that compared equal distances
        # of two different nodes, counted by the census form of the search kernel -- a superset of the reference's own
        # decisions (the oracle's census of the same queries is cpu_baseline.tie_census); untimed, separate pass
        s.engine_ties = None
        try:
            s.index.set_tuning("tie_census", 1)
            s.index.reset_counters()
            for b in range(nb_exact):
                s.search_now(s.myQ[b * s.B:(b + 1) * s.B], s.B)
            tz = s.index.tie_counters()
            s.engine_ties = dict(queries=nb_exact * s.B, queries_with_tie=tz["queries_with_tie"], events=tz["search_events"],
                                 note="k of N queries met a decision that compared equal distances of two different nodes (stop test "
                                      "core.rs:635, accept test :657, or equal distances among the k + 1 nearest): only those can be "
                                      "answered differently by the reference's binary; counted by the kernel itself, a superset of the "
                                      "oracle's per-decision census")
            # tuning tie_mode = 1: the flagged queries answered again in the reference binary's own heap order (one lane each,
            # hnsw_std_heap.hpp); what that costs a batch and how many answers it changes -- untimed elsewhere, off by default
            import time as _t
            s.search_now(s.myQ[:s.B], s.B)
            ids_total = s.d_ids.cpu().numpy().copy()
            s.index.set_tuning("tie_mode", 1)
            s.search_now(s.myQ[:s.B], s.B)                                 # (allocates the std-order scratch)
            t0 = _t.perf_counter()
            s.search_now(s.myQ[:s.B], s.B)
            t_mode = _t.perf_counter() - t0
            ids_std = s.d_ids.cpu().numpy().copy()
            s.index.set_tuning("tie_mode", 0)
            t0 = _t.perf_counter()
            s.search_now(s.myQ[:s.B], s.B)
            t_census = _t.perf_counter() - t0
            s.engine_ties["tie_mode"] = dict(batch=s.B, ms_per_batch=round(t_mode * 1e3, 2), ms_per_batch_census_only=round(t_census * 1e3, 2),
                                             answers_changed=int((ids_total != ids_std).any(axis=1).sum()),
                                             note="one lone batch with tuning tie_mode = 1: the flagged queries are answered again as the "
                                                  "Rust binary would (std BinaryHeap order, one wavefront per query)")
        except Exception as e:                                            # a shape without a census kernel
            s.engine_ties = dict(queries=0, note="not counted: %s" % e)
        finally:
            s.index.set_tuning("tie_mode", 0)
            s.index.set_tuning("tie_census", 0)
            s.index.reset_counters()

        # ---- recall@k against brute force (rank 0's batches) ------------------------------
        s.recall = None
        if s.rank == 0:
            V_dev = s.torch.from_numpy(s.V).to(s.dev)
            nb = min(2, s.n_qbatches)
            hits = tot = 0
            for b in range(nb):
                q = s.myQ[b * s.B:(b + 1) * s.B]
                s.search_now(q, s.B)
                got = s.d_ids.cpu().numpy().astype(np.int64)
                gt = brute_force_gt(s.torch, V_dev, q, s.k)
                for a, bb in zip(got, gt):
                    hits += len(set(a.tolist()) & set(bb.tolist()))
                    tot += s.k
            s.recall = hits / tot
            s.log("recall@%d = %.4f" % (s.k, s.recall))
            del V_dev


    def lone_launch_leg(self):
        """the literal BASELINE shape: one launch at a time"""
        s = self
        # ---- the literal BASELINE shape: ONE 1024-query launch at a time (one wave per SIMD, nothing to backfill from)
        s.lone = None
        if s.world == 1:
            for _ in range(3):
                s.search_now(s.myQ[:s.B], s.B)
            e0, e1 = s.torch.cuda.Event(enable_timing=True), s.torch.cuda.Event(enable_timing=True)
            reps = 12
            e0.record(s.cur)
            for r_ in range(reps):
                q = s.myQ[(r_ % s.n_qbatches) * s.B:(r_ % s.n_qbatches + 1) * s.B]
                s.index.search_batch_device(q.data_ptr(), s.B, s.k, s.d_ids.data_ptr(), s.d_sims.data_ptr(), s.d_n.data_ptr(), s.cur.cuda_stream)
            e1.record(s.cur)
            s.torch.cuda.synchronize()
            msl = e0.elapsed_time(e1) / reps
            byl = s.B * (s.n_dist_q * 4 * s.dim + s.n_ids_q * 4 + 4 * s.dim + 8 * s.k)
            s.lone = dict(batch=s.B, launches_in_flight=1, two_wave_kernel=bool(s.index.last_search_was_duo()),
                        kernel_ms=round(msl, 4), value=round(s.B / msl * 1e3, 1), unit="queries/s",
                        achieved=round(byl / (msl * 1e-3) / 1e9, 1), frac=round(byl / (msl * 1e-3) / 1e9 / HBM_PEAK_GBS, 4))
            s.log("lone %d-query launches: %.3f ms each, %.0f GB/s" % (s.B, msl, s.lone["achieved"]))


    def host_buffers_leg(self):
        """hnsw_search_batch: host memory in, host memory out"""
        s = self
        # ---- host memory in, host memory out (PCIe both ways; never `value`): hnsw_search_batch pipelines a large
        # batch itself -- pinned staging, H2D / kernel / D2H of different chunks overlapped on the engine's lanes
        s.Qh = s.Qall[:8 * s.B]
        s.index.search_batch(s.Qh, s.k)
        per_call8 = []
        for _ in range(6):
            tc0 = time.perf_counter()
            s.hb_ids, _, _ = s.index.search_batch(s.Qh, s.k)
            per_call8.append(time.perf_counter() - tc0)
        s.host_qps = s.Qh.shape[0] / float(np.median(per_call8))
        s.index.search_batch(s.Qall[:s.B], s.k)
        per_call = []
        for _ in range(6):
            tc0 = time.perf_counter()
            s.index.search_batch(s.Qall[:s.B], s.k)
            per_call.append(time.perf_counter() - tc0)
        host_qps_1024 = s.B / float(np.median(per_call))
        s.host = dict(batch=int(s.Qh.shape[0]), value=round(s.host_qps, 1), unit="queries/s", one_batch_of_1024=round(host_qps_1024, 1),
                    per_call_ms=[round(1e3 * x, 3) for x in per_call8], per_call_ms_1024=[round(1e3 * x, 3) for x in per_call],
                    note="hnsw_search_batch from pageable host memory, results back in host memory, one call at a time; the rates are "
                         "batch / MEDIAN call time of six calls (every call is listed: one call in a run can stall for several ms on "
                         "the host side)")
        s.log("host buffers: %.0f QPS at B=%d (%s ms), %.0f at B=%d (%s ms)" % (
            s.host_qps, s.Qh.shape[0], " ".join("%.2f" % (1e3 * x) for x in per_call8), host_qps_1024, s.B,
            " ".join("%.2f" % (1e3 * x) for x in per_call)))
        s.pipe = s.index.pipeline_info()


    def device_call_leg(self):
        """one hnsw_search_batch_device call per size"""
        s = self
        # ---- one hnsw_search_batch_device CALL per size, calls back to back on one stream: the engine splits a call
        # into 1024-query chunks over its own lanes and joins them back, so every call pays its own drain
        s.big = None
        s.dev_calls = []
        if s.extras:
            for Bb in (4096, 8192, 16384):
                Qb = s.torch.from_numpy(np.random.default_rng(5).random((Bb, s.dim), dtype=np.float32)).to(s.dev)
                bi = s.torch.empty((Bb, s.k), dtype=s.torch.int32, device=s.dev)
                bs = s.torch.empty((Bb, s.k), dtype=s.torch.float32, device=s.dev)
                bn = s.torch.empty((Bb,), dtype=s.torch.int32, device=s.dev)
                for _ in range(2):
                    s.index.search_batch_device(Qb.data_ptr(), Bb, s.k, bi.data_ptr(), bs.data_ptr(), bn.data_ptr(), s.cur.cuda_stream)
                s.torch.cuda.synchronize()
                e0, e1 = s.torch.cuda.Event(enable_timing=True), s.torch.cuda.Event(enable_timing=True)
                reps = 8
                e0.record(s.cur)
                for _ in range(reps):
                    s.index.search_batch_device(Qb.data_ptr(), Bb, s.k, bi.data_ptr(), bs.data_ptr(), bn.data_ptr(), s.cur.cuda_stream)
                e1.record(s.cur)
                s.torch.cuda.synchronize()
                msb = e0.elapsed_time(e1) / reps
                byb = Bb * (s.n_dist_q * 4 * s.dim + s.n_ids_q * 4 + 4 * s.dim + 8 * s.k)
                ent = dict(batch=Bb, calls_in_flight=1, ms_per_call=round(msb, 4), value=round(Bb / msb * 1e3, 1), unit="queries/s",
                           achieved=round(byb / (msb * 1e-3) / 1e9, 1), frac=round(byb / (msb * 1e-3) / 1e9 / HBM_PEAK_GBS, 4))
                s.dev_calls.append(ent)
                s.log("one %d-query device call: %.3f ms, %.0f GB/s" % (Bb, msb, ent["achieved"]))
                del Qb, bi, bs, bn
            s.big = s.dev_calls[0]


    def build_legs(self):
        """the GPU builds: batched (not the reference's order); reference-order on a prefix, then single adds / deletes; clustered data"""
        s = self
        # ---- informational: the same configuration on clustered data, where the reference algorithm's
        # recall is high enough for recall parity to mean something (uniform 128-d: 0.22-0.27 at 1 M)
        s.clus = None
        s.fast_build = None
        if s.cfg_is_c2 and s.extras:
            tb = time.time()
            ifast = s.Index("bench-fast", s.dim, s.M, s.ef, device=s.local_rank)
            ifast.add_batch(s.V, levels=s.levels, mode="fast")
            s.torch.cuda.synchronize()
            tfb = time.time() - tb
            ifast.search_batch_device(s.myQ[:s.B].data_ptr(), s.B, s.k, s.d_ids.data_ptr(), s.d_sims.data_ptr(), s.d_n.data_ptr(), s.cur.cuda_stream)
            s.torch.cuda.synchronize()
            V_dev = s.torch.from_numpy(s.V).to(s.dev)
            gotf = s.d_ids.cpu().numpy().astype(np.int64)
            gtf = brute_force_gt(s.torch, V_dev, s.myQ[:s.B], s.k)
            del V_dev
            rc = ref_insert_counters(s.N)
            s.fast_build = dict(build_seconds=round(tfb, 2), inserts_per_s=round(s.N / tfb, 1),
                              roofline=None if rc is None else insert_roofline(
                                  rc[0], rc[1], s.N, tfb, s.dim, "the oracle's serial build of the same %d vectors (data/c2_ref_insert_counters.json); "
                                  "the batched build itself evaluates about half of them" % s.N),
                              recall_at_10=round(sum(len(set(a.tolist()) & set(b.tolist())) for a, b in zip(gotf, gtf)) / (s.B * s.k), 4),
                              note="hnsw_add_batch mode 1 (batched GPU build, BASELINE config 5): not the reference's insert order")
            ifast.close()
            s.log("fast GPU build: %.2f s, recall@10 %.4f" % (tfb, s.fast_build["recall_at_10"]))
        s.exact_build = None
        s.single_add = None
        s.single_delete = None
        if s.cfg_is_c2 and s.extras and s.world == 1:
            # the reference-order build (hnsw_add_batch mode 0: plans in parallel, validated in-order commits) on a
            # bounded prefix, CHECKED row for row against the oracle's serial build of the same prefix (committed
            # fixture, tests/fixtures/make_ref_graph.py --nodes 50000); the full 1 M build and its identity check are
            # scripts/exact_build_check.py
            NE = 50_000
            ie = s.Index("bench-exact", s.dim, s.M, s.ef, device=s.local_rank)
            te = time.time()
            ie.add_batch(s.V[:NE], levels=s.levels[:NE], mode="exact")
            te = time.time() - te
            identical, why = None, "data/c2_ref_graph_50k.npz is missing"
            if os.path.exists(FIXTURE_50K):
                want, _ = load_graph_fixture(FIXTURE_50K, s.V)
                got = ie.export_graph()
                identical = (want["enterpoint"] == got["enterpoint"] and want["max_layer"] == got["max_layer"]
                             and np.array_equal(want["levels"], got["levels"])
                             and all(np.array_equal(a_, b_) for a_, b_ in zip(want["row_ptr"], got["row_ptr"]))
                             and all(np.array_equal(a_, b_) for a_, b_ in zip(want["col"], got["col"])))
                why = "levels, enterpoint and every adjacency row of every layer in stored order == the CPU oracle's serial build"
                if not identical:
                    raise SystemExit("gpu_exact_build: the GPU's reference-order graph differs from the oracle's fixture")
            rc = ref_insert_counters(NE)
            tze = ie.tie_counters()
            s.exact_build = dict(nodes=NE, build_seconds=round(te, 2), inserts_per_s=round(NE / te, 1), identical=identical,
                               ties=dict(plans_with_tie=tze["plans_with_tie"], events=tze["insert_events"], inserts=NE,
                                         note="plans (of %d inserts; a re-planned node counts again) whose search or select_neighbors compared "
                                              "equal distances of two different nodes, plus such cuts in the shrink loop: where the "
                                              "reference's binary may link differently (hnsw_get_tie_counters)" % NE),
                               checked_against=why,
                               roofline=None if rc is None else insert_roofline(
                                   rc[0], rc[1], NE, te, s.dim, "the oracle's serial build of the same %d-node prefix (data/c2_ref_insert_counters.json)" % NE),
                               note="hnsw_add_batch mode 0 on the first 50 k nodes (the rate grows with the index; the whole 1 M "
                                    "build: profiles/r6_c5_exact_build_1m.json)")
            # HNSW.NODE.ADD as the Redis command issues it (src/lib.rs:356: one add_node per call): single hnsw_add
            # calls on that index, timed, then CHECKED against the oracle making the same inserts on the same graph
            if identical:
                NA = 200
                extra_v = np.random.default_rng(11).random((NA, s.dim), dtype=np.float32)
                extra_l = draw_levels(NA, s.M, 13)
                ta = time.time()
                for i in range(NA):
                    ie.add_node("single%d" % i, extra_v[i], level=int(extra_l[i]))
                ta = (time.time() - ta) / NA
                from oracle import oracle as _orc                # checker only, after the timed region
                want["vectors"] = s.V[:NE]
                oa = _orc.OracleIndex.from_graph(s.dim, s.M, s.ef, want)
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
                s.single_add = dict(workload="HNSW.NODE.ADD: %d single hnsw_add calls (host vectors) on the %d-node reference-order index" % (NA, NE),
                                  gpu_ms=round(1e3 * ta, 3), cpu_oracle_ms=round(1e3 * tc, 3), identical=True,
                                  roofline=insert_roofline(c1_.n_dist - c0_[0], c1_.n_ids - c0_[1], NA, ta * NA, s.dim,
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
                s.single_delete = dict(workload="HNSW.NODE.DEL: %d single hnsw_delete calls on the same index" % ND,
                                     gpu_ms=round(1e3 * td, 3), cpu_oracle_ms=round(1e3 * tcd, 3), identical=True)
                s.log("single hnsw_delete: %.3f ms per call (CPU oracle %.3f ms), graphs identical" % (1e3 * td, 1e3 * tcd))
                oa.close()
                s.log("single hnsw_add: %.3f ms per call (CPU oracle %.3f ms), graphs identical" % (1e3 * ta, 1e3 * tc))
            ie.close()
            s.log("exact GPU build of %d nodes: %.1f s, identical to the oracle's: %s" % (NE, te, identical))
        if s.cfg_is_c2 and s.extras and not s.args.no_clustered:
            centers = np.random.default_rng(3).random((64, s.dim), dtype=np.float32)
            Vc = clustered(s.N, s.dim, 3, centers)
            Qc = s.torch.from_numpy(clustered(s.B, s.dim, 4, centers)).to(s.dev)
            ic = s.Index("bench-clustered", s.dim, s.M, s.ef, device=s.local_rank)
            ic.add_batch(Vc, levels=s.levels, mode="fast")
            for _ in range(2):
                ic.search_batch_device(Qc.data_ptr(), s.B, s.k, s.d_ids.data_ptr(), s.d_sims.data_ptr(), s.d_n.data_ptr(), s.cur.cuda_stream)
            s.torch.cuda.synchronize()
            tc0 = time.perf_counter()
            for _ in range(5):
                ic.search_batch_device(Qc.data_ptr(), s.B, s.k, s.d_ids.data_ptr(), s.d_sims.data_ptr(), s.d_n.data_ptr(), s.cur.cuda_stream)
            s.torch.cuda.synchronize()
            tcl = (time.perf_counter() - tc0) / 5
            Vc_dev = s.torch.from_numpy(Vc).to(s.dev)
            gtc = brute_force_gt(s.torch, Vc_dev, Qc, s.k)
            gotc = s.d_ids.cpu().numpy().astype(np.int64)
            hit = sum(len(set(a.tolist()) & set(bb.tolist())) for a, bb in zip(gotc, gtc))
            s.clus = dict(data="64-cluster Gaussian mixture, sigma 0.1 (fast GPU build)", recall_at_10=round(hit / (s.B * s.k), 4),
                        value=round(s.B / tcl, 1), unit="queries/s")
            s.log("clustered data: recall@%d = %.4f, %.3f ms/step" % (s.k, hit / (s.B * s.k), 1e3 * tcl))
            del Vc_dev, Vc
            ic.close()


    def compressed_legs(self):
        """bf16 / fp8 serving copies of the same graph"""
        s = self
        # ---- informational: the compressed serving copies of the same graph (NOT the reference's
        # arithmetic inputs: vectors rounded to bf16 / fp8 e4m3, then the reference's f32 kernel on the stored values) --
        # same kind of timed loop, 1024-query calls round-robin on `cs` streams (the bf16 / fp8 forms of the dim-128 kernel
        # hold three waves per SIMD: four calls in flight fill it)
        s.bf16 = None
        s.fp8 = None
        if s.extras and s.graph is not None and s.dim % 32 == 0:
            V_dev = s.torch.from_numpy(s.V).to(s.dev)
            gt_c = brute_force_gt(s.torch, V_dev, s.myQ[:s.B], s.k)
            del V_dev
            s.search_now(s.myQ[:s.B], s.B)
            got32 = s.d_ids.cpu().numpy().astype(np.int64)
            for fmt_name, esz, cs in (("bf16", 2, 4), ("fp8", 1, 4)):
                ib = s.Index("bench-" + fmt_name, s.dim, s.M, s.ef, device=s.local_rank)
                gb = dict(s.graph)
                gb["vectors"] = s.V
                ib.import_graph(gb)
                ib.set_tuning("compress_" + fmt_name, 1)
                cstreams = (s.streams + [s.extra_stream])[:cs]
                cbufs = [(s.torch.empty((s.B, s.k), dtype=s.torch.int32, device=s.dev), s.torch.empty((s.B, s.k), dtype=s.torch.float32, device=s.dev),
                          s.torch.empty((s.B,), dtype=s.torch.int32, device=s.dev)) for _ in range(cs)]

                def cstep(i):
                    q = s.myQ[(i % s.n_qbatches) * s.B:(i % s.n_qbatches + 1) * s.B]
                    o_ = cbufs[i % cs]
                    ib.search_batch_device(q.data_ptr(), s.B, s.k, o_[0].data_ptr(), o_[1].data_ptr(), o_[2].data_ptr(), cstreams[i % cs].cuda_stream)
                for i in range(3 * cs):
                    cstep(i)
                s.torch.cuda.synchronize()
                nbc = 72
                tb0 = time.perf_counter()
                for i in range(nbc):
                    cstep(i)
                s.torch.cuda.synchronize()
                tbc = (time.perf_counter() - tb0) / nbc
                # the same work handed over as ONE call of 16 B queries (the engine's own lanes keep the chunks in flight):
                # independent of how the runtime maps the caller's streams onto hardware queues, which the figure above is not
                # (two of the four streams on one queue halve it)
                Bc = 16 * s.B
                Qc = s.torch.from_numpy(np.random.default_rng(5).random((Bc, s.dim), dtype=np.float32)).to(s.dev)
                cbig = (s.torch.empty((Bc, s.k), dtype=s.torch.int32, device=s.dev), s.torch.empty((Bc, s.k), dtype=s.torch.float32, device=s.dev),
                        s.torch.empty((Bc,), dtype=s.torch.int32, device=s.dev))
                for _ in range(2):
                    ib.search_batch_device(Qc.data_ptr(), Bc, s.k, cbig[0].data_ptr(), cbig[1].data_ptr(), cbig[2].data_ptr(), s.cur.cuda_stream)
                s.torch.cuda.synchronize()
                tb1 = time.perf_counter()
                for _ in range(6):
                    ib.search_batch_device(Qc.data_ptr(), Bc, s.k, cbig[0].data_ptr(), cbig[1].data_ptr(), cbig[2].data_ptr(), s.cur.cuda_stream)
                s.torch.cuda.synchronize()
                tbig = (time.perf_counter() - tb1) / 6 / 16            # per B queries
                shapes = {"%d calls of %d in flight" % (cs, s.B): round(s.B / tbc, 1), "one call of %d at a time" % Bc: round(s.B / tbig, 1)}
                del Qc, cbig
                tbc = min(tbc, tbig)
                ib.search_batch_device(s.myQ[:s.B].data_ptr(), s.B, s.k, s.d_ids.data_ptr(), s.d_sims.data_ptr(), s.d_n.data_ptr(), s.cur.cuda_stream)
                s.torch.cuda.synchronize()
                gotc = s.d_ids.cpu().numpy().astype(np.int64)
                byc = s.B * (s.n_dist_q * esz * s.dim + s.n_ids_q * 4 + 4 * s.dim + 8 * s.k)
                ent = dict(value=round(s.B / tbc, 1), unit="queries/s", ms_per_step=round(1e3 * tbc, 4), calls_in_flight=cs, launch_shapes=shapes,
                           kernel=("specialised dim-128 kernel, %s rows" if ib.last_search_was_lean() else "general kernel, %s rows") % fmt_name,
                           recall_at_10=round(sum(len(set(a.tolist()) & set(b.tolist())) for a, b in zip(gotc, gt_c)) / (s.B * s.k), 4),
                           top10_overlap_with_f32=round(sum(len(set(a.tolist()) & set(b.tolist())) for a, b in zip(gotc, got32)) / (s.B * s.k), 4),
                           achieved=round(byc / tbc / 1e9, 1), frac=round(byc / tbc / 1e9 / HBM_PEAK_GBS, 4),
                           note="separate mode: vectors stored as %s (%d B/component in the gather), widened exactly, f32 accumulation in the "
                                "reference's order; bit-identical to the reference on the stored (rounded) vectors, not to the f32 headline; "
                                "achieved / frac are on this mode's own algorithmic bytes" % (fmt_name, esz))
                ib.close()
                s.log("%s copy: %.3f ms/step (%.2f M QPS), recall@10 %.4f" % (fmt_name, 1e3 * tbc, s.B / tbc / 1e6, ent["recall_at_10"]))
                if fmt_name == "bf16":
                    s.bf16 = ent
                else:
                    s.fp8 = ent


    def group_leg_(self):
        """one process, every visible GPU (hnsw_group_*)"""
        s = self
        # ---- one process, every visible GPU (SURVEY 8e "one process, 8 devices"; what a Redis module would use): the
        # C ABI's hnsw_group_* layer -- replicas by peer copies, the host batch split over the members, one host thread
        # per replica.  Informational: runs when this process sees more than one device (HNSW_BENCH_GROUP=1 forces the
        # functional form with a second member on the same device); never `value`.
        s.group_leg = None
        ndev_vis = s.torch.cuda.device_count()
        if s.extras and (ndev_vis > 1 or os.environ.get("HNSW_BENCH_GROUP") == "1"):
            try:
                from redis_hnsw_amd.group import Group
                devs = [d for d in range(ndev_vis) if d != s.local_rank] if ndev_vis > 1 else [s.local_rank]
                tg0 = time.perf_counter()
                grp = Group(s.index, devs)
                t_make = time.perf_counter() - tg0
                Gn = len(grp)
                Qg = np.ascontiguousarray(np.tile(s.Qall[:8 * s.B], (Gn, 1))[:8 * s.B * Gn])
                g_ids, _, _ = grp.search_batch(Qg, s.k)
                tg = time.perf_counter()
                for _ in range(3):
                    g_ids, _, _ = grp.search_batch(Qg, s.k)
                g_qps = 3 * Qg.shape[0] / (time.perf_counter() - tg)
                same = bool(np.array_equal(g_ids[:s.Qh.shape[0]], s.hb_ids))
                s.group_leg = dict(members=Gn, devices=[s.local_rank] + devs, create_seconds=round(t_make, 3),
                                 batch=int(Qg.shape[0]), value=round(g_qps, 1), unit="queries/s",
                                 vs_one_member=round(g_qps / s.host_qps, 3), answers_equal_single_index=same,
                                 note="hnsw_group_search_batch from pageable host memory (8192 queries per member per call); "
                                      "weak scaling against host_buffers")
                grp.close()
                s.log("single-process group of %d: %.0f QPS (%.2fx one member), identical answers: %s" % (Gn, g_qps, g_qps / s.host_qps, same))
                if not same:
                    raise SystemExit("hnsw_group_search_batch answers differ from the single index")
            except Exception as e:                       # informational leg: report, never fail the bench line
                s.group_leg = dict(error=repr(e)[:300])
                s.log("single-process group leg failed: %r" % (e,))


    def roofline_leg(self):
        """roofline of the dominant kernel"""
        s = self
        # ---- roofline of the dominant kernel (k_search) -------------------------------------
        # algorithmic bytes per launch = B x (n_dist*4*dim + n_ids*4 + 4*dim + 8*k)  (SURVEY 8d), the reference's counts
        bytes_per_launch = s.B * (s.n_dist_q * 4 * s.dim + s.n_ids_q * 4 + 4 * s.dim + 8 * s.k)
        in_flight = min(s.S, s.args.steps) if s.args.steps else 1
        per_launch = bytes_per_launch / (s.kernel_ms * 1e-3) / 1e9
        # `in_flight` launches run at a time; what the chip delivered is their bytes over the timed region's wall
        # time (this rank's), which also pays for the gaps between launches: never above in_flight x per_launch
        overlapped = in_flight * per_launch
        achieved = min(overlapped, bytes_per_launch * s.args.steps / s.t_local / 1e9) if s.args.steps else overlapped
        # HBM traffic per launch: measured IN THIS RUN when rocprofv3 is on PATH (two short re-runs of this command
        # under --pmc FETCH_SIZE / WRITE_SIZE, separate passes; FETCH_SIZE x2 is the gfx950 correction of
        # MI355X_MICROARCH.md, calibrated on this access pattern in profiles/*_pmc_hbm.txt), else the committed figure
        traffic, traffic_source = None, None
        if not s.args.no_traffic and s.world == 1 and s.args.steps:
            traffic, traffic_source = measure_traffic(sys.argv[1:], s.log, s.B, bytes_per_launch)
        if traffic is None and not (traffic_source or "").startswith("REJECTED"):
            try:
                tj = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
                want = dict(nodes=s.N, dim=s.dim, M=s.M, ef=s.ef, k=s.k, batch=s.B, graph=s.mode, streams=s.S)
                for ent in tj.get("entries", []):
                    cfg_e = ent.get("config") or {}
                    if cfg_e and all(cfg_e.get(kk) == vv for kk, vv in want.items()) and "k_search_hbm_bytes_per_launch" in ent:
                        traffic = ent["k_search_hbm_bytes_per_launch"]
                        traffic_source = "profiles/traffic.json@%s (not measured in this run)" % ent.get("commit", "?")
                        break
            except (OSError, ValueError, KeyError, TypeError):
                pass
        if s.per_rank is not None and s.args.steps:
            # every rank's own roofline fraction: its steps' algorithmic bytes (rank 0's per-query figure; the ranks' queries
            # are draws of the same distribution) over its own wall time of the timed region
            s.per_rank["roofline_frac"] = [round(bytes_per_launch * s.args.steps / sec_ / 1e9 / HBM_PEAK_GBS, 4) for sec_ in s.per_rank["seconds"]]
        s.roofline = dict(bound="hbm", achieved=round(achieved, 1), peak=HBM_PEAK_GBS, unit="GB/s",
                        frac=round(achieved / HBM_PEAK_GBS, 4), traffic=traffic, traffic_source=traffic_source,
                        peak_hbm_spec=HBM_PEAK_GBS, hbm_measured_copy=HBM_MEASURED_GBS,
                        hbm_measured_gather=measured_gather(4 * s.dim, s.N * s.dim * 4 / 1e6),
                        frac_of_measured_copy=round(achieved / HBM_MEASURED_GBS, 4),
                        what="ALGORITHMIC bytes (the reference's n_dist x 4 dim + n_ids x 4 + query + results per query) over time, "
                             "against the 8 TB/s HBM3E spec.  It is not a DRAM-level rate: the counters behind `traffic` sit at the "
                             "L2's memory side and include Infinity-Cache hits (256 MB of cache in front of this workload's %.2f GB vector "
                             "matrix), which is how the figure can exceed what a streaming copy sustains from DRAM "
                             "(hbm_measured_copy; hbm_measured_gather is the same access pattern without the walk: random %d-byte "
                             "rows, profiles/r5_gather_bw.txt)" % (s.N * s.dim * 4 / 1e9, 4 * s.dim),
                        kernel="k_search", kernel_ms=round(s.kernel_ms, 4),
                        kernel_ms_median=None if s.kernel_ms_median is None else round(s.kernel_ms_median, 4),
                        kernel_ms_p90=None if s.kernel_ms_p90 is None else round(s.kernel_ms_p90, 4),
                        launches_in_flight=in_flight,
                        per_launch_gbs=round(per_launch, 1), in_flight_x_per_launch_gbs=round(overlapped, 1),
                        note="achieved = algorithmic bytes of the timed steps / wall time of the timed region (launches overlap: "
                             "launches_in_flight x algorithmic_bytes_per_launch / kernel_ms, the average duration of one launch "
                             "by HIP events per stream, is the upper figure in_flight_x_per_launch_gbs)",
                        algorithmic_bytes_per_launch=int(bytes_per_launch),
                        n_dist_per_query=round(s.n_dist_q, 1), n_ids_per_query=round(s.n_ids_q, 1),
                        n_expand_per_query=round(s.n_exp_q, 1),
                        re_evaluated_fraction=round(max(s.redo, 0.0), 5),
                        lone_launch_1024=s.lone,
                        single_launch_4096=s.big)


    def cpu_legs(self):
        """the oracle on the host cores: the CPU baseline, the tie census, one query per call on this index, C1"""
        s = self
        # ---- CPU baseline: the oracle (C restatement of the Rust path), bounded sample -------
        s.cpu = None
        s.c1 = None
        s.c2s = None
        cfg_name_early = {(1_000_000, 128, 16, 200): "C2", (1_000_000, 768, 32, 400): "C3", (10_000_000, 128, 16, 200): "C4"}.get((s.N, s.dim, s.M, s.ef), "this")
        if not s.args.no_cpu_baseline and s.world == 1:   # timed at N=1 only: the other ranks would idle in the barrier
            from oracle import oracle
            s.graph["vectors"] = s.V
            o = oracle.OracleIndex.from_graph(s.dim, s.M, s.ef, s.graph)
            Qs = s.Qall[:s.B]
            o.search_batch(Qs[:64], s.k, threads=1)              # warm-up
            done = 0
            tc = time.perf_counter()
            chunk = min(128, s.B)                          # (C1's preset has one query per batch)
            while True:
                lo = done % s.B
                o.search_batch(Qs[lo:lo + chunk], s.k, threads=1)
                done += chunk
                if time.perf_counter() - tc > s.args.cpu_seconds:
                    break
            t1 = time.perf_counter() - tc
            cores = usable_cores()
            o.search_batch(Qs, s.k, threads=cores)               # starts the persistent workers
            tc = time.perf_counter()
            reps = 0
            while True:
                o.search_batch(Qs, s.k, threads=cores)
                reps += 1
                if time.perf_counter() - tc > s.args.cpu_seconds / 2:
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
            Qcen = s.Qall[:s.n_qbatches * s.B]
            while n_cen < Qcen.shape[0] and time.perf_counter() - tq0 < max(s.args.cpu_seconds, 6.0):
                for qi in range(n_cen, min(n_cen + 256, Qcen.shape[0])):
                    c_ = o.tie_census(Qcen[qi:qi + 1], s.k)
                    for key_ in cen:
                        cen[key_] += c_[key_]
                    if c_["queries_with_any_tie"]:
                        tied_q.append(qi)
                n_cen = min(n_cen + 256, Qcen.shape[0])
            differ = 0
            for qi in tied_q:
                rid, rsim = o.search_std_heap(Qcen[qi], s.k)
                gid, gsim, gn = s.index.search_batch(Qcen[qi:qi + 1], s.k)
                gid, gsim = gid[0, :int(gn[0])], gsim[0, :int(gn[0])]
                differ += not (np.array_equal(gid, rid) and np.array_equal(gsim.view(np.uint32), rsim.view(np.uint32)))
            cen.update(tied_queries_answered_differently_in_the_rust_heap_order=differ,
                       note="of the first %d queries of this bench: decisions that met EQUAL similarities of two different nodes "
                            "(the only place the reference's sim-only order and the (sim, id) order can part); every tied query was "
                            "re-run in std::collections::BinaryHeap's own order (restated, tests/golden/tiecase_rust_lattice.npz) "
                            "and compared with the engine's answer, ids and similarity bits" % n_cen)
            s.log("tie census of %d queries: %d with a decision tie, %d with an answer tie; %d answered differently in the Rust heap order" % (
                n_cen, cen["queries_with_decision_tie"], cen["queries_with_answer_tie"], differ))
            s.cpu = dict(value=round(done / t1, 1), unit="queries/s", cores=1, kind="port", tie_census=cen,
                       sample="%d queries of the same 1024-query batch on the same graph, 1 thread, %.1f s" % (done, t1),
                       cpu_model=cpu_model(),
                       all_cores=dict(value=round(reps * s.B / tN, 1), cores=cores, visible_cpus=os.cpu_count(),
                                      sample="%d passes over the batch, persistent workers with per-thread scratch; "
                                             "cores = affinity mask capped by the cgroup CPU quota" % reps))
            # ---- one HNSW.SEARCH per call on THIS index (the command's shape, src/lib.rs:484), beside the oracle's
            s.c2s = None
            if s.extras:
                Q1 = s.Qall[:200]
                for q in Q1[:20]:
                    s.index.search_knn(q, s.k)
                same = all([r.id for r in s.index.search_knn(q, s.k)] == o.search(q, s.k)[0].tolist() for q in Q1[:20])
                tq = time.perf_counter()
                for q in Q1:
                    s.index.search_knn(q, s.k)
                t_g = (time.perf_counter() - tq) / len(Q1)
                tq = time.perf_counter()
                for q in Q1:
                    o.search(q, s.k)
                t_c = (time.perf_counter() - tq) / len(Q1)
                s.c2s = dict(workload="%s index, one query per hnsw_search call (host buffers)" % cfg_name_early,
                           gpu_us=round(1e6 * t_g, 1), cpu_oracle_us=round(1e6 * t_c, 1), identical=bool(same),
                           two_wave_kernel=bool(s.index.last_search_was_duo()))
                s.log("one query per call on this index: %.1f us (CPU oracle %.1f us), identical: %s" % (1e6 * t_g, 1e6 * t_c, same))
            # ---- HNSW.NODE.ADD in the reference's order AT THIS INDEX'S SCALE (BASELINE config 5's end state): a batch of new
            # vectors inserted into a copy of the 1 M reference-order graph through hnsw_add_batch mode 0 (plans in parallel,
            # commits in validated parallel groups), then CHECKED row for row against the oracle making the same inserts
            s.exact_at_scale = None
            if s.extras and s.cfg_is_c2 and s.mode == "reference":
                NS = 4096
                newV = np.random.default_rng(21).random((NS, s.dim), dtype=np.float32)
                newL = draw_levels(NS, s.M, 23)
                ix = s.Index("bench-at-scale", s.dim, s.M, s.ef, device=s.local_rank)
                gb = dict(s.graph)
                gb["vectors"] = s.V
                ix.import_graph(gb)
                s.torch.cuda.synchronize()
                ix.reset_counters()
                tg = time.time()
                ix.add_batch(newV, levels=newL, mode="exact")
                s.torch.cuda.synchronize()
                tg = time.time() - tg
                tzs = ix.tie_counters()
                c0_ = o.insert_counters()
                c0_ = (c0_.n_dist, c0_.n_ids)
                tc_ = time.time()
                for i in range(NS):
                    o.add(newV[i], int(newL[i]))
                tc_ = time.time() - tc_
                c1_ = o.insert_counters()
                ga, gb2 = o.export(), ix.export_graph()
                same = (ga["enterpoint"] == gb2["enterpoint"] and np.array_equal(ga["levels"], gb2["levels"])
                        and all(np.array_equal(a_, b_) for a_, b_ in zip(ga["row_ptr"], gb2["row_ptr"]))
                        and all(np.array_equal(a_, b_) for a_, b_ in zip(ga["col"], gb2["col"])))
                if not same:
                    raise SystemExit("exact_build_at_scale: the graph after %d reference-order inserts at 1 M nodes differs from the oracle's" % NS)
                s.exact_at_scale = dict(workload="HNSW.NODE.ADD x %d in the reference's order on the %d-node reference-order index (hnsw_add_batch mode 0)" % (NS, s.N),
                                        build_seconds=round(tg, 3), inserts_per_s=round(NS / tg, 1), cpu_oracle_inserts_per_s=round(NS / tc_, 1),
                                        identical=True, checked_against="every adjacency row of every layer == the oracle after the same inserts",
                                        ties=dict(plans_with_tie=tzs["plans_with_tie"], events=tzs["insert_events"], inserts=NS),
                                        roofline=insert_roofline(c1_.n_dist - c0_[0], c1_.n_ids - c0_[1], NS, tg, s.dim,
                                                                 "the oracle making the same %d inserts on the same graph" % NS),
                                        note="the rate of the reference-order build where BASELINE config 5 ends; the whole build from an "
                                             "empty index: profiles/r6_c5_exact_build_1m.json (113.7 s = 8 796 inserts/s, identical)")
                ix.close()
                s.log("reference-order inserts at 1 M nodes: %.0f inserts/s (CPU oracle %.0f), graphs identical" % (NS / tg, NS / tc_))
            o.close()
            # ---- C1 (BASELINE config 1): 10k x 128, M=5, ef=200, k=10, ONE query per call (the shape of a
            # HNSW.SEARCH command): hnsw_search latency, host buffers in and out, beside the oracle's
            if s.extras:
                n1, m1 = 10_000, 5
                lv1 = draw_levels(n1, m1, 7)
                o1 = oracle.OracleIndex(s.dim, m1, s.ef)
                o1.add_batch(s.V[:n1], lv1)
                g1 = s.Index("c1", s.dim, m1, s.ef, device=s.local_rank)
                g1.import_graph(o1.export())
                Q1 = s.Qall[:300]
                same = True
                for q in Q1[:40]:
                    a = g1.search_knn(q, s.k)
                    ids1, sims1 = o1.search(q, s.k)
                    same = same and [r.id for r in a] == ids1.tolist()
                tq = time.perf_counter()
                for q in Q1:
                    g1.search_knn(q, s.k)
                t_g = (time.perf_counter() - tq) / len(Q1)
                tq = time.perf_counter()
                for q in Q1:
                    o1.search(q, s.k)
                t_c = (time.perf_counter() - tq) / len(Q1)
                s.c1 = dict(workload="C1: 10k x 128, M=5, ef=200, k=10, one query per hnsw_search call (host buffers)",
                          gpu_us=round(1e6 * t_g, 1), cpu_oracle_us=round(1e6 * t_c, 1), identical=bool(same),
                          two_wave_kernel=bool(g1.last_search_was_duo()))
                g1.close(); o1.close()


    def emit(self):
        """the one JSON line"""
        s = self
        known = {(1_000_000, 128, 16, 200, 10, 1024): "C2", (1_000_000, 768, 32, 400, 100, 4096): "C3",
                 (10_000_000, 128, 16, 200, 10, 1024): "C4", (10_000, 128, 5, 200, 10, 1): "C1",
                 (300_000, 768, 32, 400, 100, 4096): "C3's shape on a 300 k reference-order graph"}
        cfg_name = known.get((s.N, s.dim, s.M, s.ef, s.k, s.B), "custom")
        if cfg_name == "C2" and s.args.graph == "exact":
            cfg_name = "C5"                                  # the same index, BUILT on the GPU in the reference's order first
        qps = s.world * s.B * s.args.steps / s.t_wall
        out = {
            "metric": "HNSW.SEARCH QPS + recall@10, 1M x 128 f32, ef=200" if cfg_name == "C2" else
                      "HNSW.SEARCH QPS + recall@%d (%s)" % (s.k, cfg_name),
            "value": round(qps, 1), "unit": "queries/s", "n_gpus": s.world, "steps": s.args.steps, "warmup": s.args.warmup,
            "ms_per_step": round(1e3 * s.t_wall / s.args.steps, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%s: %d nodes x dim %d, M=%d, ef=%d, k=%d, batch=%d queries/GPU, uniform[0,1) f32, replicated index"
                                   % (cfg_name, s.N, s.dim, s.M, s.ef, s.k, s.B),
                       "nodes": s.N, "dim": s.dim, "M": s.M, "ef": s.ef, "k": s.k, "batch": s.B, "graph": s.mode, "graph_desc": s.graph_desc,
                       "steps_in_flight": s.S, "hw_queues": int(os.environ.get("GPU_MAX_HW_QUEUES", "4")),
                       "launch_concurrency": s.args.launch_concurrency or "observed by the engine", "pipeline": s.pipe,
                       "prewarm_launches": s.PREWARM,
                       "parallelism": "replica x%d, query batch sharded%s" % (
                           s.world, " (ranks share one device, gloo: functional check only)" if s.one_device and s.world > 1 else "")},
            "gather_verified": s.gather_ok,
            "rccl_world": s.world if (s.world > 1 and s.backend == "nccl") else None,
            "collective_backend": s.backend if s.world > 1 else None,
            "tie_census_engine": getattr(s, "engine_ties", None),
            "per_rank": s.per_rank, "index_replication": s.replication, "topk_exchange": s.gather_cmp, "first_contact": s.first_contact_rows,
            "recall_at_%d" % s.k: None if s.recall is None else round(s.recall, 4),
            "build_seconds": None if s.t_build is None else round(s.t_build, 2),
            "host_buffers_qps": round(s.host_qps, 1), "host_buffers": s.host, "single_process_group": s.group_leg, "device_call": s.dev_calls,
            "gpu_fast_build": s.fast_build, "gpu_exact_build": s.exact_build, "gpu_exact_build_at_1m": s.exact_at_scale, "single_add": s.single_add, "single_delete": s.single_delete,
            "clustered": s.clus,
            "bf16_storage_mode": s.bf16, "fp8_storage_mode": s.fp8,
            "c1_single_query": s.c1, "single_query_on_this_index": s.c2s,
            "setup_seconds": round(time.time() - s.t0, 1),
            "roofline": s.roofline,
            "cpu_baseline": s.cpu,
        }
        print(json.dumps(out))
        if s.world > 1:
            s.dist.barrier()
            s.dist.destroy_process_group()


    def run(self):
        s = self
        s.build_index()
        s.stage_inputs()
        if s.args.dry:
            return s.dry_run()
        if s.timed_region():
            return
        s.exact_counters_and_recall()
        if s.rank != 0:
            if s.world > 1:
                s.dist.barrier()
                s.dist.destroy_process_group()
            return
        s.extras = s.world == 1 and not s.args.no_extras
        s.lone_launch_leg()
        s.host_buffers_leg()
        s.device_call_leg()
        s.build_legs()
        s.compressed_legs()
        s.group_leg_()
        s.roofline_leg()
        s.exact_at_scale = None
        s.cpu_legs()
        s.emit()


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
    Bench(args).run()


if __name__ == "__main__":
    main()
