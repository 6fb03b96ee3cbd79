"""Randomised differential test: one op sequence (NODE.ADD single / bulk, NODE.DEL, SEARCH, snapshot and
restore) driven through the C ABI and through the CPU oracle in lock step.

After every few ops the two graphs must be equal row for row in stored order, the touched sets
(core.rs:441-446, 580-584) equal as sets, and every search equal in ids, similarity bits and n_out.
The data sets are chosen to hit what uniform data does not: exact ties (lattice, duplicates), chain-like
neighbourhoods (the reference's own line fixture, core_tests.rs), tight clusters, dims on both metric
orders (metrics.rs:20-31 vs :33-46).
"""
import numpy as np
import pytest

from tests.util import graphs_equal

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    import redis_hnsw_amd
    from redis_hnsw_amd import index as idxmod
    return idxmod


def _bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def _data(kind, n, dim, rng):
    if kind == "uniform":
        return rng.random((n, dim), dtype=np.float32)
    if kind == "clustered":
        c = rng.random((8, dim), dtype=np.float32)
        return (c[rng.integers(0, 8, n)] + 0.01 * rng.standard_normal((n, dim)).astype(np.float32)).astype(np.float32)
    if kind == "line":                                   # core_tests.rs: node i = [i, i, i, i]
        return np.repeat(np.arange(n, dtype=np.float32)[:, None], dim, axis=1)[rng.permutation(n)]
    if kind == "lattice":                                # small integer coordinates: many exact ties
        return rng.integers(0, 3, (n, dim)).astype(np.float32)
    if kind == "dupes":                                  # every vector appears ~4 times
        base = rng.random((max(n // 4, 1), dim), dtype=np.float32)
        return base[rng.integers(0, base.shape[0], n)]
    raise AssertionError(kind)


CASES = [
    # kind, dim, m, ef, n_ops, seed
    ("uniform", 128, 16, 200, 60, 1),
    ("uniform", 33, 5, 16, 80, 2),          # scalar metric order
    ("clustered", 64, 8, 40, 70, 3),
    ("line", 4, 5, 16, 80, 4),
    ("lattice", 32, 4, 24, 70, 5),
    ("dupes", 32, 6, 30, 70, 6),
    ("uniform", 32, 2, 8, 90, 7),           # m = 2: rows of 2 / 4, shrinks on almost every link
    ("clustered", 256, 32, 100, 40, 8),     # wide rows
    ("lattice", 4, 12, 64, 60, 9),          # ties everywhere with wide rows
    ("uniform", 32, 32, 64, 60, 10),        # M = 32: m_max0 = 64 = one wave; deletes push rows past 64 ids
    ("dupes", 64, 32, 80, 50, 11),
    ("clustered", 128, 16, 200, 60, 12),
    ("uniform", 100, 3, 10, 90, 13),
]


@pytest.mark.parametrize("kind,dim,m,ef,n_ops,seed", CASES)
def test_random_op_sequences_match_the_oracle(eng, oracle_mod, kind, dim, m, ef, n_ops, seed, tunings=(), stress=False):
    """stress (scripts/fuzz_campaign.py): batches large enough for the library's own search pipeline (chunks on three
    lanes), and at the end the index is turned into a bf16 / fp8 serving copy and compared with the oracle on the
    stored values (tombstones, wide rows and whatever tunings the case carries included)."""
    rng = np.random.default_rng(seed)
    pool = _data(kind, 6000, dim, rng)
    used = 0
    o = oracle_mod.OracleIndex(dim, m, ef)
    gi = eng.Index("fz", dim, m, ef)
    for key, val in tunings:                             # scripts/fuzz_campaign.py: non-default engine paths
        gi.set_tuning(key, val)
    live = []
    level_seed = 100 + seed

    def take(n):
        nonlocal used
        n = min(n, pool.shape[0] - used)
        V = pool[used:used + n]
        used += n
        return V

    def check_graph(where):
        ok, why = graphs_equal(o.export(), gi.export_graph())
        assert ok, "%s: %s" % (where, why)
        assert gi.node_count == o.live_count

    for op_i in range(n_ops):
        r = rng.random()
        n_now = o.node_count
        if r < 0.30 or n_now < 4:                                            # single NODE.ADD with touched set
            V = take(1)
            if V.shape[0] == 0:
                continue
            lv = int(oracle_mod.draw_levels(1, m, level_seed + op_i)[0])
            oi, ot = o.add(V[0], lv, want_touched=True)
            got = []
            gid = gi.add_node("node%d" % oi, V[0], lambda s, nid: got.append(nid), level=lv)
            assert gid == oi
            assert sorted(got) == sorted(ot.tolist()), "op %d: touched set of add %d" % (op_i, oi)
            live.append(oi)
        elif r < 0.55:                                                       # bulk NODE.ADD, serial or windowed
            V = take(int(rng.choice([2, 17, 63, 64, 65, 150, 400])))
            if V.shape[0] == 0:
                continue
            lv = oracle_mod.draw_levels(V.shape[0], m, level_seed + op_i)
            base = o.node_count
            o.add_batch(V, lv)
            gi.add_batch(V, levels=lv, mode="exact")
            live.extend(range(base, base + V.shape[0]))
        elif r < 0.75 and len(live) > 8:                                     # NODE.DEL (sometimes the enterpoint)
            i = o.enterpoint if rng.random() < 0.2 else int(live[rng.integers(0, len(live))])
            ot = o.delete(int(i), want_touched=True)
            got = []
            gi.delete_node("node%d" % i, lambda s, nid: got.append(nid))
            assert sorted(got) == sorted(ot.tolist()), "op %d: touched set of delete %d" % (op_i, i)
            live.remove(i)
        elif r < 0.93:                                                       # SEARCH, batch and single
            B = int(rng.choice([1, 3, 40, 130, 700, 1500] if stress else [1, 3, 40]))
            k = int(rng.choice([1, 5, ef, ef + 7]))
            Q = pool[rng.integers(0, pool.shape[0], B)] + (0 if rng.random() < 0.5 else
                                                            rng.random((B, dim), dtype=np.float32) * np.float32(0.1))
            Q = np.ascontiguousarray(Q, dtype=np.float32)
            ids, sims, n_out = gi.search_batch(Q, k)
            oids, osims, on, _ = o.search_batch(Q, k)
            assert np.array_equal(n_out, on), "op %d" % op_i
            for b in range(B):
                c = int(on[b])
                assert np.array_equal(ids[b, :c], oids[b, :c]), "op %d query %d" % (op_i, b)
                assert np.array_equal(_bits(sims[b, :c]), _bits(osims[b, :c])), "op %d query %d" % (op_i, b)
            one = gi.search_knn(Q[0], k)
            assert [x.id for x in one] == ids[0, : int(n_out[0])].tolist()
        else:                                                                # snapshot -> fresh engine
            blob = gi.serialize()
            names = dict(gi._ids)
            gi.close()
            gi = eng.Index.deserialize(blob)
            assert dict(gi._ids) == names
            for key, val in tunings:
                gi.set_tuning(key, val)
        if op_i % 6 == 5:
            check_graph("after op %d" % op_i)
    check_graph("end")
    if stress and dim % 32 == 0 and o.live_count > 4:
        fmt = str(rng.choice(["bf16", "fp8"]))
        if kind in ("line",) and fmt == "fp8":
            fmt = "bf16"                                  # coordinates up to 6000: beyond e4m3's range (clamped at 448)
        gi.set_tuning("compress_" + fmt, 1)
        g = o.export()
        n_all = o.node_count
        Vs = np.zeros((n_all, dim), np.float32)
        for i in range(n_all):
            Vs[i] = gi._vector(i)
        g["vectors"] = Vs
        o2 = oracle_mod.OracleIndex.from_graph(dim, m, ef, g)
        for B in (1, 90, 1100):
            Q = np.ascontiguousarray(pool[rng.integers(0, pool.shape[0], B)] +
                                     rng.random((B, dim), dtype=np.float32) * np.float32(0.05), dtype=np.float32)
            k = int(rng.choice([1, 10, ef]))
            ids, sims, n_out = gi.search_batch(Q, k)
            oids, osims, on, _ = o2.search_batch(Q, k, threads=8)
            assert np.array_equal(n_out, on), "compressed %s B=%d" % (fmt, B)
            for b in range(B):
                c = int(on[b])
                assert np.array_equal(ids[b, :c], oids[b, :c]), "compressed %s query %d" % (fmt, b)
                assert np.array_equal(_bits(sims[b, :c]), _bits(osims[b, :c])), "compressed %s query %d" % (fmt, b)
        o2.close()
    gi.close()


@pytest.mark.parametrize("kind,dim,m,ef,n_ops,seed,tunings", [
    # found by scripts/fuzz_campaign.py: waves_per_cu = 1 gave the insert kernels the whole CU's LDS for the visited
    # table and left none for the validation scratch of the windowed insert (launch refused, "invalid argument")
    ("dupes", 100, 31, 200, 40, 70010, (("select_shortcut", 0), ("visited_bounded", 1), ("waves_per_cu", 1))),
    ("dupes", 96, 16, 200, 40, 70119, (("visited_bounded", 0), ("occ_min_batch", 2), ("occ_ahead_x10", 30), ("waves_per_cu", 1))),
    ("uniform", 128, 32, 700, 30, 5, (("waves_per_cu", 1),)),
    # round 5: the windowed inserts of these sequences commit in validated parallel groups whatever their yield
    # (commit_par = 2; tiny dense graphs, where nearly every node conflicts with its predecessor)
    ("uniform", 128, 16, 200, 80, 500001, (("commit_par", 2), ("occ_min_batch", 2))),
    ("lattice", 32, 4, 24, 120, 500002, (("commit_par", 2), ("occ_min_batch", 2), ("occ_window", 16))),
    ("dupes", 12, 2, 8, 120, 500003, (("commit_par", 2), ("occ_min_batch", 2), ("commit_team", 0))),
    ("clustered", 768, 32, 100, 40, 500004, (("commit_par", 2), ("occ_min_batch", 2))),
])
def test_random_op_sequences_under_non_default_tunings(eng, oracle_mod, kind, dim, m, ef, n_ops, seed, tunings):
    test_random_op_sequences_match_the_oracle(eng, oracle_mod, kind, dim, m, ef, n_ops, seed, tunings)


@pytest.mark.parametrize("kind,dim,m,ef,n_ops,seed,tunings", [
    # found by the round-3 campaign (scripts/fuzz_campaign.py with stress=True, 40 of 282 cases, all M > 32): the device
    # list behind update_fn is pushed without deduplication and was capped at 8192 entries -- a delete at M = 48 / 64
    # pushes 10-40 k -- so the reported touched set silently lost ids (the graph itself was right)
    ("lattice", 64, 64, 200, 40, 300003, ()),
    ("uniform", 4, 48, 100, 80, 300065, ()),
    ("clustered", 33, 33, 300, 120, 300067, (("occ_log_cap", 300), ("lean", 1), ("lds_hash_bits", 8), ("occ_min_batch", 2),
                                              ("occ_ahead_x10", 30), ("waves_per_cu", 4), ("tag_table", 1), ("pipe_chunk", 256),
                                              ("grid_stride", 1))),
    ("uniform", 256, 64, 40, 80, 300281, (("select_shortcut", 0), ("pipe_chunk", 1024), ("force_restride", 48))),
    # the stress form on the reference's usual shapes: large batches through the pipeline, compressed copy at the end
    ("uniform", 128, 16, 200, 60, 300500, ()),
    ("clustered", 64, 40, 100, 60, 300501, (("pipe_chunk", 64),)),
    # found by the campaign once HNSW.NODE.DEL speculated its re-selections (1 of 120 cases): on a dense graph (M = 48)
    # one validation pass met more (reader, id) pairs than its list holds (1024); the overflow marked the LINK PLAN
    # stale -- which a shrinks-only pass never looks at -- instead of the shrink, so a stale re-selection was applied
    ("clustered", 128, 48, 200, 40, 800104, ()),
    # ... and, on a 20-node m = 2 graph: a re-selection that does not fill its list must count every change as relevant
    ("uniform", 32, 2, 8, 90, 7, ()),
])
def test_stress_sequences_large_m_pipeline_and_compressed_copies(eng, oracle_mod, kind, dim, m, ef, n_ops, seed, tunings):
    test_random_op_sequences_match_the_oracle(eng, oracle_mod, kind, dim, m, ef, n_ops, seed, tunings, stress=True)
