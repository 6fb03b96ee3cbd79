"""Replay one case of scripts/fuzz_search.py (same seed -> same data, queries and tunings; the fast build itself is not
reproducible link for link, so the graph differs from run to run) and, on a mismatch, narrow it down on THAT graph:
which storage format, which tunings, which batch shape.  usage: python scripts/repro_search.py SEED [repeats] [--save]"""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from oracle import oracle as oracle_mod
from redis_hnsw_amd import index as eng

oracle_mod.build()
seed = int(sys.argv[1])
repeats = int(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2].isdigit() else 1


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def compare(tag, got, want, verbose=True):
    ids, sims, n_out = got
    oids, osims, on = want[0], want[1], want[2]
    bad = []
    for b in range(ids.shape[0]):
        c = int(on[b])
        if int(n_out[b]) != c or not np.array_equal(ids[b, :c], oids[b, :c]) or not np.array_equal(bits(sims[b, :c]), bits(osims[b, :c])):
            bad.append(b)
    print("%-46s %d of %d queries differ %s" % (tag, len(bad), ids.shape[0], bad[:8]), flush=True)
    if bad and verbose:
        b = bad[0]
        c = int(on[b])
        j = next((i for i in range(c) if ids[b, i] != oids[b, i] or bits(sims[b, i:i + 1])[0] != bits(osims[b, i:i + 1])[0]), -1)
        print("   query %d: n_out %d vs %d; first difference at rank %d" % (b, int(n_out[b]), c, j))
        print("   engine ids ", ids[b, max(j - 2, 0):j + 4].tolist(), " sims", sims[b, max(j - 2, 0):j + 4].tolist())
        print("   oracle ids ", oids[b, max(j - 2, 0):j + 4].tolist(), " sims", osims[b, max(j - 2, 0):j + 4].tolist())
    return bad


for rep in range(repeats):
    rng = np.random.default_rng(seed)
    n = int(rng.choice([300, 2000, 20000, 70000, 150000]))
    dim = int(rng.choice([32, 64, 96, 128, 128, 128, 256, 768, 100]))
    if dim >= 256:
        n = min(n, 20000)
    m = int(rng.choice([3, 5, 8, 16, 16, 24, 32, 40, 48]))
    ef = int(rng.choice([10, 50, 200, 200, 300, 400, 512, 700]))
    k = int(rng.choice([1, 10, 100]))
    B = int(rng.choice([1, 17, 256, 1500, 5000]))
    kind = str(rng.choice(["uniform", "clustered", "lattice"]))
    if kind == "uniform":
        V = rng.random((n, dim), dtype=np.float32)
    elif kind == "clustered":
        c = rng.random((32, dim), dtype=np.float32)
        V = (c[rng.integers(0, 32, n)] + 0.02 * rng.standard_normal((n, dim)).astype(np.float32)).astype(np.float32)
    else:
        V = rng.integers(0, 4, (n, dim)).astype(np.float32)
    Q = np.ascontiguousarray(V[rng.integers(0, n, B)] + rng.random((B, dim), dtype=np.float32) * np.float32(rng.choice([0, 0.05, 1.0])))
    tun = []
    for key, vals in (("visited_bounded", [0, 1]), ("lean", [0, 1]), ("waves_per_cu", [1, 2, 4, 8, 12]),
                      ("lds_hash_bits", [8, 10, 12]), ("tag_table", [0, 1]), ("idbits", [20, 24]),
                      ("launch_concurrency", [1, 2]), ("query_in_lds", [0, 1]), ("pipe_chunk", [64, 512, 1024]),
                      ("grid_stride", [0, 1]), ("force_restride", [16, 64])):
        if rng.random() < 0.35:
            tun.append((key, int(rng.choice(vals))))
    fmt = str(rng.choice(["f32", "f32", "bf16", "fp8"])) if dim % 32 == 0 else "f32"
    print("run %d: n=%d dim=%d m=%d ef=%d k=%d B=%d %s tun=%s fmt=%s" % (rep, n, dim, m, ef, k, B, kind, tun, fmt), flush=True)
    gi = eng.Index("rp", dim, m, ef)
    gi.add_batch(V, levels=oracle_mod.draw_levels(n, m, seed), mode="fast")
    n_del = int(rng.choice([0, 0, 5]))
    for i in rng.permutation(n)[:n_del]:
        gi.delete_node("node%d" % i)
    g = gi.export_graph()
    g["vectors"] = V
    o32 = oracle_mod.OracleIndex.from_graph(dim, m, ef, g)
    want32 = o32.search_batch(Q, k, threads=8)
    # ---- f32, default tunings, then the case's tunings one by one and together ----
    bad_any = compare("f32, default tunings", gi.search_batch(Q, k), want32)
    for key, val in tun:
        try:
            gi.set_tuning(key, val)
        except eng.HNSWError as e:
            print("   tuning %s=%d refused: %s" % (key, val, e.msg))
        bad_any += compare("f32, + %s=%d" % (key, val), gi.search_batch(Q, k), want32)
    if fmt != "f32":
        import torch
        gi.set_tuning("compress_" + fmt, 1)
        g2 = dict(g)
        g2["vectors"] = torch.from_numpy(V).to(torch.bfloat16 if fmt == "bf16" else torch.float8_e4m3fn).to(torch.float32).numpy()
        oc = oracle_mod.OracleIndex.from_graph(dim, m, ef, g2)
        wantc = oc.search_batch(Q, k, threads=8)
        bad = compare("%s, the case's tunings" % fmt, gi.search_batch(Q, k), wantc)
        bad_any += bad
        if bad:
            for key, val in (("visited_bounded", 1), ("visited_bounded", 0), ("grid_stride", 0), ("pipe_chunk", 1024), ("waves_per_cu", 4)):
                gi.set_tuning(key, val)
                compare("%s, then %s=%d" % (fmt, key, val), gi.search_batch(Q, k), wantc, verbose=False)
            b = bad[0]
            compare("%s, the first bad query alone (B=1)" % fmt, gi.search_batch(Q[b:b + 1], k), (wantc[0][b:b + 1], wantc[1][b:b + 1], wantc[2][b:b + 1]))
            compare("%s, 64 queries around it" % fmt, gi.search_batch(Q[b - 32:b + 32], k), (wantc[0][b - 32:b + 32], wantc[1][b - 32:b + 32], wantc[2][b - 32:b + 32]), verbose=False)
        oc.close()
    if bad_any and "--save" in sys.argv:
        np.savez_compressed("gpurun_out/repro_graph_%d_%d.npz" % (seed, rep), levels=g["levels"], enterpoint=g["enterpoint"],
                            max_layer=g["max_layer"], **{"rp%d" % l: r for l, r in enumerate(g["row_ptr"])},
                            **{"col%d" % l: c for l, c in enumerate(g["col"])})
        print("   graph saved", flush=True)
    gi.close()
    o32.close()
