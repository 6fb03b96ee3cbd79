#!/usr/bin/env python3
"""Per-phase cycle breakdown of k_search (needs a -DHNSW_PHASE_TIMERS build:
HNSW_MI355X_LIB=redis_hnsw_amd/lib/libhnsw_mi355x_prof.so).  Usage: phase_profile.py [N]"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bench import draw_levels
from redis_hnsw_amd import Index, _capi

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
dim, M, ef, k, B = 128, 16, 200, 10, (int(sys.argv[2]) if len(sys.argv) > 2 else 1024)
V = np.random.default_rng(1).random((N, dim), dtype=np.float32)
Q = np.random.default_rng(2).random((4 * B, dim), dtype=np.float32)
gi = Index("p", dim, M, ef)
for kv in os.environ.get("HNSW_TUNE", "").split(","):
    if "=" in kv:
        a, b = kv.split("="); gi.set_tuning(a, int(b))
t = time.time(); gi.add_batch(V, levels=draw_levels(N, M), mode="fast"); print("build %.2fs" % (time.time() - t))
dev = torch.device("cuda", 0)
dQ = torch.from_numpy(Q).to(dev)
ids = torch.empty((B, k), dtype=torch.int32, device=dev); sims = torch.empty((B, k), dtype=torch.float32, device=dev)
nn = torch.empty(B, dtype=torch.int32, device=dev)
st = torch.cuda.current_stream()
def run(i):
    q = dQ[(i % 4) * B:(i % 4 + 1) * B]
    gi.search_batch_device(q.data_ptr(), B, k, ids.data_ptr(), sims.data_ptr(), nn.data_ptr(), st.cuda_stream)
for i in range(3): run(i)
torch.cuda.synchronize(); gi.reset_counters()
reps = 10
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(st)
for i in range(reps): run(i)
e1.record(st); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / reps
sc, _ = gi.counters()
print("kernel %.3f ms/batch  %.0f QPS  n_dist/q %.0f n_expand/q %.1f n_ids/q %.0f spills %d" % (
    ms, B / ms * 1e3, sc.n_dist / (reps * B), sc.n_expand / (reps * B), sc.n_ids / (reps * B), sc.n_spill))
bytes_ = sc.n_dist * 4 * dim + sc.n_ids * 4
print("algorithmic GB/s %.0f" % (bytes_ / reps / (ms * 1e-3) / 1e9))
lib = _capi.load()
if hasattr(lib, "hnsw_debug_phase_cycles"):
    out = (C.c_uint64 * 8)()
    lib.hnsw_debug_phase_cycles(gi._h, out)
    tot = sum(out[:6])
    if tot:
        names = ["pop+row fetch", "visited filter", "gather+dist", "merge W"]
        per_step = sc.n_expand
        for i, nm in enumerate(names):
            print("  %-16s %6.1f%%  %8.0f cycles/expansion" % (nm, 100.0 * out[i] / tot, out[i] / per_step))
        print("  choose next + row request   %8.0f cycles/expansion" % (out[4] / per_step))
        print("  merge ranks (under row fetch) %6.0f cycles/expansion" % (out[5] / per_step))
        print("  longest wave %.1f us" % (out[7] / 100.0))
        print("  total cycles/expansion %.0f ; cycles/query %.0f" % (tot / per_step, tot / (reps * B)))

if hasattr(lib, "hnsw_debug_phase_cycles") and sum(out[:4]):
    run(0); torch.cuda.synchronize()
    a = ids.cpu().numpy().astype(np.int64)
    kc, ne, nd = a[:, k - 1], a[:, k - 2], a[:, k - 3]
    print("per-query kcycles: mean %.0f p50 %.0f p90 %.0f p99 %.0f max %.0f" % (kc.mean(), np.percentile(kc, 50), np.percentile(kc, 90), np.percentile(kc, 99), kc.max()))
    print("per-query expansions: mean %.0f p99 %.0f max %d ; dist: mean %.0f p99 %.0f max %d" % (ne.mean(), np.percentile(ne, 99), ne.max(), nd.mean(), np.percentile(nd, 99), nd.max()))
    print("cycles per expansion by query: p50 %.0f p99 %.0f max %.0f" % tuple(np.percentile(kc * 1024.0 / ne, [50, 99, 100])))
    order = np.argsort(kc)[-8:]
    print("slowest queries (idx, kcycles, expansions, dist):", [(int(i), int(kc[i]), int(ne[i]), int(nd[i])) for i in order])
    print("corr(kcycles, dist) = %.3f" % np.corrcoef(kc, nd)[0, 1])
    blk = np.arange(B)
    print("mean kcycles by block%%8 (XCD):", [int(kc[blk % 8 == x].mean()) for x in range(8)])
