"""Differential build campaign (GPU box): larger indexes than tests/test_gpu_fuzz.py, built in the reference's
order through the windowed exact insert in bulk calls of random sizes, with deletes and single adds between
them; the graph must equal the oracle's serial build row for row after every bulk call.
usage: python scripts/fuzz_build.py [seconds] [first_seed]"""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from oracle import oracle as oracle_mod
from redis_hnsw_amd import index as eng
from tests.test_gpu_fuzz import _data
from tests.util import graphs_equal

oracle_mod.build()
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 300.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
t0 = time.time()
done = bad = 0
while time.time() - t0 < budget:
    rng = np.random.default_rng(seed)
    kind = str(rng.choice(["uniform", "clustered", "lattice", "dupes", "line"]))
    dim = int(rng.choice([4, 32, 64, 128, 128, 256]))
    if kind == "line":
        dim = 4
    m = int(rng.choice([2, 5, 8, 16, 16, 32, 40, 64]))
    ef = int(rng.choice([max(m, 8), 40, 100, 200, 400]))
    n = int(rng.choice([8000, 20000, 40000]))
    tun = []
    for key, vals in (("occ_window", [8, 32, 64]), ("occ_ahead_x10", [10, 15, 40]), ("select_shortcut", [0, 1]),
                      ("occ_log_cap", [500, 3072]), ("plan_lean", [0, 1]), ("commit_par", [0, 2, 2]), ("plan_split", [0, 1]),
                      ("occ_chain", [1, 8]), ("commit_team", [0, 1]), ("occ_stage_ahead", [0, 8, 32]), ("occ_depth_x10", [0, 30]),
                      ("occ_front_max", [3, 16]), ("par_max_resident", [3, 0])):
        if rng.random() < 0.3:
            tun.append((key, int(rng.choice(vals))))
    case = dict(seed=seed, kind=kind, dim=dim, m=m, ef=ef, n=n, tun=tun)
    try:
        V = _data(kind, n, dim, rng)
        lv = oracle_mod.draw_levels(n, m, seed)
        o = oracle_mod.OracleIndex(dim, m, ef)
        gi = eng.Index("fb", dim, m, ef)
        for key, val in tun:
            gi.set_tuning(key, val)
        pos = 0
        while pos < n:
            step = int(min(n - pos, rng.choice([100, 1000, 5000, 15000])))
            o.add_batch(V[pos:pos + step], lv[pos:pos + step])
            gi.add_batch(V[pos:pos + step], levels=lv[pos:pos + step], mode="exact")
            pos += step
            ok, why = graphs_equal(o.export(), gi.export_graph())
            assert ok, "after %d nodes: %s" % (pos, why)
            for _ in range(int(rng.integers(0, 4))):           # a few deletes between bulk calls
                i = int(rng.integers(0, pos))
                if o.is_live(i):
                    o.delete(i)
                    gi.delete_node("node%d" % i)
        ok, why = graphs_equal(o.export(), gi.export_graph())
        assert ok, "end: " + why
        dbg = gi.info()
        gi.close()
        o.close()
        print("ok", case, "%.0f s" % (time.time() - t0), flush=True)
    except Exception as e:
        bad += 1
        print("FAIL", case, type(e).__name__, str(e).split("\n")[0][:200], flush=True)
    done += 1
    seed += 1
print("cases %d, failures %d, %.0f s" % (done, bad, time.time() - t0))
