"""Differential search campaign (GPU box): random index shapes built by the fast GPU build, exported, loaded into
the oracle (from_graph) and searched by both under random engine tunings.  ids, similarity bits, n_out and the
work counters must be equal.  usage: python scripts/fuzz_search.py [seconds] [first_seed]"""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from oracle import oracle as oracle_mod
from redis_hnsw_amd import index as eng

oracle_mod.build()
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 300.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
t0 = time.time()
done = bad = skipped = 0


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


while time.time() - t0 < budget:
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
    # compressed serving copies: any dim % 32 == 0, any M and ef (round 3)
    fmt = str(rng.choice(["f32", "f32", "bf16", "fp8"])) if dim % 32 == 0 else "f32"
    case = dict(seed=seed, n=n, dim=dim, m=m, ef=ef, k=k, B=B, kind=kind, tun=tun, fmt=fmt)
    try:
        gi = eng.Index("fs", dim, m, ef)
        gi.add_batch(V, levels=oracle_mod.draw_levels(n, m, seed), mode="fast")
        n_del = int(rng.choice([0, 0, 5]))
        for i in rng.permutation(n)[:n_del]:
            gi.delete_node("node%d" % i)
        g = gi.export_graph()
        g["vectors"] = V
        for key, val in tun:
            try:
                gi.set_tuning(key, val)
            except eng.HNSWError:
                pass
        if fmt != "f32":
            gi.set_tuning("compress_" + fmt, 1)
            import torch
            g["vectors"] = torch.from_numpy(g["vectors"]).to(torch.bfloat16 if fmt == "bf16" else torch.float8_e4m3fn).to(torch.float32).numpy()
            probe = rng.integers(0, n, 8)
            assert all(np.array_equal(bits(gi._vector(int(i))), bits(g["vectors"][int(i)])) for i in probe), "stored values"
        Qo = Q                                                # queries stay f32 in the bf16 storage mode
        o = oracle_mod.OracleIndex.from_graph(dim, m, ef, g)
        gi.reset_counters()
        ids, sims, n_out = gi.search_batch(Q, k)
        sc, _ = gi.counters()
        oids, osims, on, oct = o.search_batch(Qo, k, threads=8)
        assert np.array_equal(n_out, on), "n_out"
        for b in range(B):
            c = int(on[b])
            assert np.array_equal(ids[b, :c], oids[b, :c]), "query %d ids" % b
            assert np.array_equal(bits(sims[b, :c]), bits(osims[b, :c])), "query %d sims" % b
        lossy = not any(kv == ("visited_bounded", 0) for kv in tun)
        if not lossy:
            assert (sc.n_dist, sc.n_ids, sc.n_expand) == (oct.n_dist, oct.n_ids, oct.n_expand), "counters"
        gi.close()
        o.close()
    except eng.HNSWError as e:                                # documented refusals are not failures
        if "served by the specialised kernel" in str(e) or "row strides" in str(e):
            skipped += 1
        else:
            bad += 1
            print("FAIL", case, "HNSWError", str(e)[:200], flush=True)
    except Exception as e:
        bad += 1
        print("FAIL", case, type(e).__name__, str(e).split("\n")[0][:200], flush=True)
    done += 1
    seed += 1
print("cases %d, failures %d, refused %d, %.0f s" % (done, bad, skipped, time.time() - t0))
