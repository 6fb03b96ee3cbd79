#!/usr/bin/env python3
"""BASELINE config 5 in the reference's order: build the C2 index on the GPU with hnsw_add_batch(mode 0) (the
windowed exact insert) and compare it row for row with the CPU oracle's serial build of the same data
(the fixture data/c2_ref_graph_1m.npz).   python scripts/exact_build_check.py [N]"""
import ctypes as C, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from bench import draw_levels, load_graph_fixture, FIXTURES
from redis_hnsw_amd import Index, _capi
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
dim, M, ef = 128, 16, 200
V = np.random.default_rng(1).random((1_000_000, dim), dtype=np.float32)[:N]
lv = draw_levels(1_000_000, M, 7)[:N]
gi = Index("c5", dim, M, ef)
for kv in os.environ.get("TUNING", "").split(","):          # e.g. TUNING=commit_par=0,occ_window=64
    if kv:
        gi.set_tuning(kv.split("=")[0], int(kv.split("=")[1]))
t = time.time(); gi.add_batch(V, levels=lv, mode="exact"); dt = time.time() - t
lib = _capi.load()
out = (C.c_uint64 * 16)()
lib.hnsw_debug_occ.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
lib.hnsw_debug_occ(gi._h, out)
res = dict(nodes=N, build_seconds=round(dt, 1), inserts_per_s=round(N / dt, 1), rounds=int(out[5]),
           commits_per_round=round(out[0] / max(out[5], 1), 2), speculative_shrinks=int(out[1]), recomputed_shrinks=int(out[2]),
           stale_plans=int(out[3]), journal_deltas_per_commit=round(out[4] / max(out[0], 1), 1))
try:
    pz = (C.c_uint64 * 21)()
    lib.hnsw_debug_occ_par.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
    lib.hnsw_debug_occ_par(gi._h, pz)
    if pz[0]:
        it = max(pz[11], 1)
        res["parallel_commit"] = dict(groups=int(pz[0]), groups_per_round=round(pz[0] / max(out[5], 1), 2), nodes_per_group=round(out[0] / pz[0], 2),
                                      dry_runs_per_commit=round(pz[1] / max(out[0], 1), 2),
                                      groups_closed_by=dict(stale_link_plan=int(pz[2]), record_used=int(pz[3]), row_rewritten=int(pz[4])),
                                      iterations=int(it), launches=int(pz[12]),
                                      us_per_iteration_workgroup0=dict(zip(("dry_run", "wait", "validate", "wait2", "apply", "wait3"),
                                                                           [round(pz[5 + i] / it / 100.0, 1) for i in range(6)])))
except AttributeError:
    pass
fx = FIXTURES[(1_000_000, dim, M, ef)]
if N == 1_000_000 and os.path.exists(fx):
    g, secs = load_graph_fixture(fx, V)
    e = gi.export_graph()
    same = e["enterpoint"] == g["enterpoint"] and e["max_layer"] == g["max_layer"] and np.array_equal(e["levels"], g["levels"])
    for l in range(g["max_layer"] + 1):
        same = same and np.array_equal(e["row_ptr"][l], g["row_ptr"][l]) and np.array_equal(e["col"][l], g["col"][l])
    res["identical_to_oracle_serial_build"] = bool(same)
    res["oracle_build_seconds_one_core"] = round(secs, 1)
print(json.dumps(res))
