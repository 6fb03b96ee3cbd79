"""Differential op sequences under tuning tie_mode (GPU box): random configurations on tie-heavy and tie-free data; batch adds,
single adds, deletes and searches through the C ABI against the oracle's std-heap operations (hnsw_oracle_add_std_heap /
delete_std_heap / search_std_heap -- the Rust binary's own tie order).  Graphs row for row in stored order, answers bit for bit.
usage: python scripts/fuzz_ties.py [seconds] [first_seed]"""
import os
import sys
import time
import traceback

import numpy as np

sys.path.insert(0, ".")
from oracle import oracle as oracle_mod                                     # noqa: E402
from redis_hnsw_amd import index as eng                                     # noqa: E402
from tests.util import graphs_equal, make_data                              # noqa: E402

oracle_mod.build()


def data(kind, n, dim, rng):
    if kind == "uniform":
        return make_data(n, dim, seed=int(rng.integers(1, 1 << 30)))
    if kind == "quantised":
        V = np.round(make_data(n + n // 4, dim, seed=int(rng.integers(1, 1 << 30))) * 2.0) / 2.0
    elif kind == "binary":
        V = rng.integers(0, 2, size=(n + n // 4, dim)).astype(np.float32)
    else:                                                                   # lattice
        V = rng.integers(0, 3, size=(n + n // 4, dim)).astype(np.float32)
    V = np.unique(V, axis=0)
    rng.shuffle(V)
    return V[:n]


def run(seed):
    rng = np.random.default_rng(seed)
    kind = str(rng.choice(["uniform", "quantised", "binary", "lattice"]))
    dim = int(rng.choice([8, 32, 40, 128, 128, 128]))
    m = int(rng.choice([3, 5, 8, 12, 16, 24]))
    ef = int(rng.choice([max(m, 4), 24, 64, 200]))
    mode = int(rng.choice([1, 1, 1, 2]))
    if os.environ.get("FUZZ_TIE_MODE"):                                     # (reproduction aid: the same case under the other mode)
        mode = int(os.environ["FUZZ_TIE_MODE"])
    n0 = int(rng.choice([60, 300, 900]))
    V = data(kind, n0 + 200, dim, rng)
    n0 = min(n0, max(len(V) - 60, 8))
    lv = oracle_mod.draw_levels(len(V), m, int(rng.integers(1, 1000)))
    case = (kind, dim, m, ef, mode, n0, len(V), seed)
    o = oracle_mod.OracleIndex(dim, m, ef)
    gi = eng.Index("fz", dim, m, ef)
    gi.set_tuning("tie_mode", mode)
    if rng.random() < 0.4:
        gi.set_tuning("occ_window", int(rng.choice([4, 16, 64])))
    try:
        o.add_batch_std_heap(V[:n0], lv[:n0])
        gi.add_batch(V[:n0], levels=lv[:n0], mode="exact")
        ok, why = graphs_equal(o.export(), gi.export_graph())
        assert ok, "after the batch build: " + why
        alive, nxt = list(range(n0)), n0
        for op in range(60):
            r = rng.random()
            if r < 0.35 and nxt < len(V):
                o.add_batch_std_heap(V[nxt:nxt + 1], lv[nxt:nxt + 1])
                gi.add_node("s%d" % nxt, V[nxt], level=int(lv[nxt]))
                alive.append(nxt)
                nxt += 1
            elif r < 0.5 and nxt + 8 <= len(V):
                b = int(rng.integers(2, 9))
                o.add_batch_std_heap(V[nxt:nxt + b], lv[nxt:nxt + b])
                gi.add_batch(V[nxt:nxt + b], levels=lv[nxt:nxt + b], mode="exact")
                alive.extend(range(nxt, nxt + b))
                nxt += b
            elif r < 0.75 and len(alive) > 6:
                v = alive.pop(int(rng.integers(0, len(alive))))
                ot = o.delete_std_heap(v, want_touched=True)
                got = []
                gi.delete_node(gi._names[v], update_fn=lambda s, nid: got.append(nid))
                assert sorted(got) == sorted(ot.tolist()), "op %d: touched set of delete %d" % (op, v)
            else:
                k = int(rng.choice([1, 5, 10]))
                Q = data(kind, 24, dim, rng) if rng.random() < 0.7 else V[rng.integers(0, nxt, size=24)]
                ids, sims, n_out = gi.search_batch(Q, k)
                for i, q in enumerate(Q):
                    oids, osims = o.search_std_heap(q, k)
                    if not (n_out[i] == len(oids) and np.array_equal(ids[i, :len(oids)], oids)) and os.environ.get("FUZZ_TIE_DIAG"):
                        tids, tsims = o.search(q, k)
                        gi.reset_counters()
                        one = gi.search_batch(q[None, :], k)
                        print("DIAG k", k, "std", oids, osims, "total", tids, "engine", ids[i], "alone", one[0][0], "oracle census", o.tie_census(q[None, :], k),
                              "engine counters", gi.tie_counters(), flush=True)
                    assert n_out[i] == len(oids) and np.array_equal(ids[i, :len(oids)], oids), "op %d: query %d" % (op, i)
                    assert np.array_equal(sims[i, :len(oids)].view(np.uint32), np.asarray(osims, dtype=np.float32).view(np.uint32))
            if op % 15 == 14:
                ok, why = graphs_equal(o.export(), gi.export_graph())
                assert ok, "after op %d: %s" % (op, why)
        ok, why = graphs_equal(o.export(), gi.export_graph())
        assert ok, "at the end: " + why
    finally:
        gi.close()
        o.close()
    return case


if __name__ == "__main__":
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 240.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 7000
    if len(sys.argv) > 3:                                                   # explicit seeds: python scripts/fuzz_ties.py 0 0 9035 9069
        for sd in sys.argv[3:]:
            try:
                print("ok", run(int(sd)), flush=True)
            except Exception as e:
                print("FAIL seed", sd, type(e).__name__, str(e).split("\n")[0][:200], flush=True)
        sys.exit(0)
    t0 = time.time()
    done = bad = 0
    while time.time() - t0 < budget:
        try:
            run(seed)
        except Exception as e:
            bad += 1
            print("FAIL seed", seed, type(e).__name__, str(e).split("\n")[0][:200], flush=True)
            traceback.print_exc(limit=2)
        done += 1
        seed += 1
    print("tie-mode cases %d, failures %d, %.0f s" % (done, bad, time.time() - t0))
