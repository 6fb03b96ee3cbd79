"""Replay one case of tests/test_gpu_fuzz.py checking the graph after EVERY op; print the first divergence."""
import os
import sys
import numpy as np
sys.path.insert(0, ".")
from oracle import oracle as oracle_mod
from redis_hnsw_amd import index as eng
from tests.util import graphs_equal
from tests.test_gpu_fuzz import CASES, _data

oracle_mod.build()
if "," in sys.argv[1]:                              # kind,dim,m,ef,n_ops,seed of a campaign line
    f = sys.argv[1].split(",")
    case = (f[0], int(f[1]), int(f[2]), int(f[3]), int(f[4]), int(f[5]))
else:
    case = CASES[int(sys.argv[1])]
kind, dim, m, ef, n_ops, seed = case
rng = np.random.default_rng(seed)
pool = _data(kind, 6000, dim, rng)
used = 0
o = oracle_mod.OracleIndex(dim, m, ef)
gi = eng.Index("fz", dim, m, ef)
for kv in sys.argv[2:]:
    k, v = kv.split("=")
    gi.set_tuning(k, int(v))
live = []
for op_i in range(n_ops):
    r = rng.random()
    n_now = o.node_count
    what = ""
    if r < 0.30 or n_now < 4:
        V = pool[used:used + 1]; used += 1
        lv = int(oracle_mod.draw_levels(1, m, 100 + seed + op_i)[0])
        oi, ot = o.add(V[0], lv, want_touched=True)
        got = []
        gi.add_node("node%d" % oi, V[0], lambda s, nid: got.append(nid), level=lv)
        what = "add %d lv %d touched_equal=%s" % (oi, lv, sorted(got) == sorted(ot.tolist()))
        live.append(oi)
    elif r < 0.55:
        nb = int(rng.choice([2, 17, 63, 64, 65, 150, 400]))
        V = pool[used:used + nb]; used += nb
        lv = oracle_mod.draw_levels(V.shape[0], m, 100 + seed + op_i)
        base = o.node_count
        o.add_batch(V, lv)
        gi.add_batch(V, levels=lv, mode="exact")
        live.extend(range(base, base + V.shape[0]))
        what = "add_batch %d at %d" % (nb, base)
    elif r < 0.75 and len(live) > 8:
        i = o.enterpoint if rng.random() < 0.2 else int(live[rng.integers(0, len(live))])
        ot = o.delete(int(i), want_touched=True)
        got = []
        gi.delete_node("node%d" % i, lambda s, nid: got.append(nid))
        what = "delete %d (deg0 touched %d) touched_equal=%s" % (i, len(ot), sorted(got) == sorted(ot.tolist()))
        live.remove(i)
    elif r < 0.93:
        B = int(rng.choice([1, 3, 40, 130, 700, 1500] if os.environ.get("STRESS") else [1, 3, 40])); k = int(rng.choice([1, 5, ef, ef + 7]))
        Q = pool[rng.integers(0, pool.shape[0], B)] + (0 if rng.random() < 0.5 else rng.random((B, dim), dtype=np.float32) * np.float32(0.1))
        what = "search"
    else:
        blob = gi.serialize(); gi.close(); gi = eng.Index.deserialize(blob)
        what = "snapshot"
    ok, why = graphs_equal(o.export(), gi.export_graph())
    print(op_i, what, "n=%d" % o.node_count, "OK" if ok else "DIVERGED: " + why, flush=True)
    if not ok:
        ga, gb = o.export(), gi.export_graph()
        for l in range(len(ga["row_ptr"])):
            ra, rb = ga["row_ptr"][l], gb["row_ptr"][l]
            da, db = np.diff(ra.astype(np.int64)), np.diff(rb.astype(np.int64))
            bad = np.nonzero(da != db)[0]
            print(" layer", l, "rows with different degree:", bad[:10], da[bad[:10]], db[bad[:10]])
            for x in bad[:3]:
                print("   oracle", ga["col"][l][ra[x]:ra[x+1]]); print("   engine", gb["col"][l][rb[x]:rb[x+1]])
        break
