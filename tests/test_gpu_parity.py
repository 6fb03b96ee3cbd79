"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle.

Bit-exact: ids, similarities (compared as raw f32 bits), work counters and --
for the exact insert -- every adjacency row in stored order.
"""
import numpy as np
import pytest

from tests.util import brute_force_topk, build_oracle, graphs_equal, make_data, recall_at_k

pytestmark = pytest.mark.gpu
EPS = float(np.finfo(np.float32).eps)


@pytest.fixture(scope="module")
def eng():
    import redis_hnsw_amd
    from redis_hnsw_amd import index as idxmod
    return idxmod


def _bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


# ---- metric: metrics_tests.rs on the device ----------------------------------
def test_metric_kats_on_device(eng):
    z, o, big = np.zeros((1, 512), np.float32), np.ones((1, 512), np.float32), np.full((1, 512), 512.0, np.float32)
    assert abs(eng.metric_pairs(o, o)[0] - 0.0) < EPS                 # diff_is_zero
    assert abs(eng.metric_pairs(z, o)[0] - -512.0) < EPS              # diff_is_512
    assert abs(eng.metric_pairs(z, big)[0] - -134217728.0) < EPS      # diff_is_512_2_x512
    a, b = np.zeros((1, 33), np.float32), np.ones((1, 33), np.float32)
    assert abs(eng.metric_pairs(a, b)[0] - -33.0) < EPS               # diff_non_x32 (scalar order)


@pytest.mark.parametrize("dim", [32, 64, 128, 256, 768, 4, 33, 100])
def test_metric_bit_exact(eng, oracle_mod, dim):
    rng = np.random.default_rng(dim)
    n = 300
    a = rng.random((n, dim), dtype=np.float32)
    b = rng.random((n, dim), dtype=np.float32)
    # wide dynamic range + exact zeros + denormal-sized differences
    a[:50] *= 1e3
    b[50:60] = a[50:60]
    a[60:70] = b[60:70] + np.float32(1e-22)
    got = eng.metric_pairs(a, b)
    want = np.array([oracle_mod.euclidean(a[i], b[i]) for i in range(n)], dtype=np.float32)
    assert np.array_equal(_bits(got), _bits(want))


# ---- search parity on an oracle-built graph -----------------------------------
CONFIGS = [
    # n, dim, m, ef, k, nq
    (1200, 32, 5, 16, 5, 64),
    (600, 4, 5, 16, 5, 64),       # scalar metric order (dim % 32 != 0)
    (2000, 128, 16, 200, 10, 128),
    (1500, 64, 8, 100, 10, 64),   # AVX order, generic dim
    (900, 768, 32, 400, 100, 32),
    (300, 128, 16, 200, 10, 16),  # index smaller than ef... reachable < ef
]


@pytest.fixture(scope="module")
def built(oracle_mod):
    cache = {}

    def get(n, dim, m, ef):
        key = (n, dim, m, ef)
        if key not in cache:
            V = make_data(n, dim, seed=1)
            o, lv = build_oracle(oracle_mod, V, m, ef)
            cache[key] = (V, o, lv)
        return cache[key]
    return get


@pytest.mark.parametrize("n,dim,m,ef,k,nq", CONFIGS)
def test_search_parity(eng, oracle_mod, built, n, dim, m, ef, k, nq):
    V, o, lv = built(n, dim, m, ef)
    g = o.export()
    gi = eng.Index("foo", dim, m, ef)
    gi.import_graph(g)
    assert gi.node_count == n and gi.max_layer == o.max_layer and gi.enterpoint_id == o.enterpoint
    Q = make_data(nq, dim, seed=2)
    gi.reset_counters()
    ids, sims, n_out = gi.search_batch(Q, k)
    oids, osims, on, oct = o.search_batch(Q, k)
    assert np.array_equal(n_out, on)
    for i in range(nq):
        c = int(on[i])
        assert np.array_equal(ids[i, :c], oids[i, :c]), "query %d ids" % i
        assert np.array_equal(_bits(sims[i, :c]), _bits(osims[i, :c])), "query %d sims" % i
        assert np.all(ids[i, c:] == 0xFFFFFFFF)
    sc, _ = gi.counters()
    assert (sc.n_dist, sc.n_ids, sc.n_expand) == (oct.n_dist, oct.n_ids, oct.n_expand)
    # identical recall@k by construction; check it anyway against brute force
    kk = min(k, 10)
    gt = brute_force_topk(V, Q, kk)
    assert recall_at_k(ids, gt) == recall_at_k(oids, gt)
    # single-query entry point agrees with the batch one
    r = gi.search_knn(Q[0], k)
    assert [x.id for x in r] == ids[0, : int(n_out[0])].tolist()
    gi.close()


def test_search_visited_spill_is_exact(eng, oracle_mod, built):
    """A tiny LDS visited table forces the HBM spill path; results must not change."""
    n, dim, m, ef, k, nq = 2000, 128, 16, 200, 10, 64
    V, o, lv = built(n, dim, m, ef)
    gi = eng.Index("foo", dim, m, ef)
    gi.import_graph(o.export())
    Q = make_data(nq, dim, seed=3)
    gi.set_tuning("visited_bounded", 0)          # the HBM spill path (the insert kernels' mode)
    gi.set_tuning("lds_hash_bits", 8)
    gi.reset_counters()
    ids, sims, n_out = gi.search_batch(Q, k)
    sc, _ = gi.counters()
    assert sc.n_spill > 0
    oids, osims, on, oct = o.search_batch(Q, k)
    assert np.array_equal(ids, oids) and np.array_equal(_bits(sims), _bits(osims))
    assert sc.n_dist == oct.n_dist
    # and a second launch on the same (now used) spill tables is still exact
    ids2, sims2, _ = gi.search_batch(Q, k)
    assert np.array_equal(ids2, oids)
    gi.close()


@pytest.mark.parametrize("bb", [3, 5, 7])
def test_tag_table_spill_and_decode_is_exact(eng, oracle_mod, built, bb):
    """A tiny 16-bit tag table: full-bucket chains and capacity spills decode every entry back to its
    id (inverse hash) and continue in HBM; ids, similarities and counters must not change."""
    n, dim, m, ef, k, nq = 2000, 128, 16, 200, 10, 64
    V, o, lv = built(n, dim, m, ef)
    gi = eng.Index("foo", dim, m, ef)
    gi.import_graph(o.export())
    Q = make_data(nq, dim, seed=3)
    gi.set_tuning("visited_bounded", 0)
    gi.set_tuning("tag_bb", bb)
    gi.reset_counters()
    ids, sims, n_out = gi.search_batch(Q, k)
    sc, _ = gi.counters()
    assert sc.n_spill > 0
    oids, osims, on, oct = o.search_batch(Q, k)
    assert np.array_equal(ids, oids) and np.array_equal(_bits(sims), _bits(osims))
    assert (sc.n_dist, sc.n_ids, sc.n_expand) == (oct.n_dist, oct.n_ids, oct.n_expand)
    gi.close()


@pytest.mark.parametrize("knob,val", [("tag_bb", 3), ("tag_bb", 6), ("tag_bb", 8), ("lds_hash_bits", 8)])
@pytest.mark.parametrize("cfg", [(2000, 128, 16, 200, 10), (900, 768, 32, 400, 100), (1200, 32, 5, 16, 5)])
def test_bounded_visited_table_is_exact(eng, oracle_mod, built, knob, val, cfg):
    """The search's default: a full LDS visited table stops recording instead of moving to HBM.  A node
    met again is evaluated again: it fails the accept test again (W's furthest only improves) unless it
    still is a member of W, and those are dropped by key equality -- so ids, similarities, n_out, the
    expansions and the ids scanned are the reference's; only the distance evaluations can exceed it."""
    n, dim, m, ef, k = cfg
    nq = 48
    V, o, lv = built(n, dim, m, ef)
    gi = eng.Index("foo", dim, m, ef)
    gi.import_graph(o.export())
    Q = make_data(nq, dim, seed=3)
    if knob == "lds_hash_bits":
        gi.set_tuning("tag_table", 0)
    gi.set_tuning(knob, val)
    gi.reset_counters()
    ids, sims, n_out = gi.search_batch(Q, k)
    sc, _ = gi.counters()
    oids, osims, on, oct = o.search_batch(Q, k)
    assert np.array_equal(n_out, on)
    assert np.array_equal(ids, oids) and np.array_equal(_bits(sims), _bits(osims))
    if ef >= 200 and not (knob == "tag_bb" and n <= (1 << val) * 6):
        assert sc.n_spill > 0, "the table was meant to fill up"
    assert (sc.n_ids, sc.n_expand) == (oct.n_ids, oct.n_expand)
    assert sc.n_dist >= oct.n_dist
    if sc.n_spill == 0:
        assert sc.n_dist == oct.n_dist
    gi.close()


@pytest.mark.parametrize("idbits,bounded", [(24, 0), (24, 1), (22, 0), (27, 1)])
def test_search_parity_with_the_id_range_of_a_10m_index(eng, oracle_mod, built, idbits, bounded):
    """C4 (10 M nodes) hashes ids over 2^24: 13 tag bits at 2048 buckets, the edge of the 16-bit tag table
    (27 bits no longer fit: the 32-bit-id table takes over).  The tag hash is a bijection of [0, 2^idbits) for
    any idbits >= log2(capacity), so the same code path runs on a small graph."""
    n, dim, m, ef, k, nq = 2000, 128, 16, 200, 10, 96
    V, o, lv = built(n, dim, m, ef)
    gi = eng.Index("foo", dim, m, ef)
    gi.import_graph(o.export())
    gi.set_tuning("idbits", idbits)
    gi.set_tuning("visited_bounded", bounded)
    Q = make_data(nq, dim, seed=6)
    gi.reset_counters()
    ids, sims, n_out = gi.search_batch(Q, k)
    oids, osims, on, oct = o.search_batch(Q, k)
    assert np.array_equal(n_out, on) and np.array_equal(ids, oids) and np.array_equal(_bits(sims), _bits(osims))
    sc, _ = gi.counters()
    assert (sc.n_ids, sc.n_expand) == (oct.n_ids, oct.n_expand)
    if sc.n_spill == 0 or not bounded:
        assert sc.n_dist == oct.n_dist
    # and with the small table of the 8-waves-per-CU launches (2^24 ids: 14 tag bits would be needed at
    # 1024 buckets, so those launches use the 32-bit table)
    gi.set_tuning("tag_bb", 10)
    ids2, sims2, _ = gi.search_batch(Q, k)
    assert np.array_equal(ids2, oids) and np.array_equal(_bits(sims2), _bits(osims))
    # the launch shape of eight waves per CU (here: 96 queries announced as one of 16 concurrent launches):
    # the specialised kernel's 1024-bucket table, whose entries keep 2 displacement bits at 2^24 ids
    gi.set_tuning("tag_bb", -1)
    gi.set_tuning("visited_bounded", 1)
    gi.set_tuning("launch_concurrency", 8)
    Qw = np.concatenate([Q, Q, Q])[:272]                       # 272 x 8 > 2048 waves: 8 per CU
    ids3, sims3, _ = gi.search_batch(Qw, k)
    assert np.array_equal(ids3[:nq], oids) and np.array_equal(_bits(sims3[:nq]), _bits(osims))
    gi.close()


def test_c1_single_query_calls(eng, oracle_mod):
    """BASELINE config 1: 10k x 128, M=5, ef=200, k=10, one query per hnsw_search call (the only shape the
    HNSW.SEARCH command can issue, src/lib.rs:462-496) on the reference-order graph."""
    n, dim, m, ef, k = 10_000, 128, 5, 200, 10
    V = make_data(n, dim, seed=1)
    lv = oracle_mod.draw_levels(n, m, 7)
    o = oracle_mod.OracleIndex(dim, m, ef)
    o.add_batch(V, lv)
    gi = eng.Index("c1", dim, m, ef)
    gi.import_graph(o.export())
    Q = make_data(200, dim, seed=2)
    short = 0
    for q in Q:
        r = gi.search_knn(q, k)                      # hnsw_search: host buffers, one launch per call
        oids, osims = o.search(q, k)
        # min(k, ef, reachable) results (core.rs:878-890): with M=5 the reference's shrinks can isolate a node
        # at layer 0 while it is still linked above, and a descent that lands on it returns that node alone
        # (query 89 of this set does)
        assert len(r) == len(oids)
        short += len(oids) < k
        assert [x.id for x in r] == oids.tolist()
        assert np.array_equal(_bits(np.float32([x.sim for x in r])), _bits(osims))
    assert short >= 1                                # the "reachable < k" case is covered
    gi.close()


def test_default_table_never_forgets_at_c2_scale_batch(eng, oracle_mod, built):
    """With the default table (sized for the batch) nothing is forgotten: all three counters are the oracle's."""
    n, dim, m, ef, k, nq = 2000, 128, 16, 200, 10, 128
    V, o, lv = built(n, dim, m, ef)
    gi = eng.Index("foo", dim, m, ef)
    gi.import_graph(o.export())
    Q = make_data(nq, dim, seed=5)
    gi.reset_counters()
    ids, sims, n_out = gi.search_batch(Q, k)
    sc, _ = gi.counters()
    oids, osims, on, oct = o.search_batch(Q, k)
    assert np.array_equal(ids, oids) and np.array_equal(_bits(sims), _bits(osims))
    assert sc.n_spill == 0 and (sc.n_dist, sc.n_ids, sc.n_expand) == (oct.n_dist, oct.n_ids, oct.n_expand)
    gi.close()


def test_32bit_visited_table_still_exact(eng, oracle_mod, built):
    """tag_table=0 keeps the 32-bit-id LDS table (used for indexes above 16 M ids)"""
    n, dim, m, ef, k, nq = 2000, 128, 16, 200, 10, 64
    V, o, lv = built(n, dim, m, ef)
    gi = eng.Index("foo", dim, m, ef)
    gi.import_graph(o.export())
    gi.set_tuning("tag_table", 0)
    Q = make_data(nq, dim, seed=3)
    ids, sims, _ = gi.search_batch(Q, k)
    oids, osims, _, _ = o.search_batch(Q, k)
    assert np.array_equal(ids, oids) and np.array_equal(_bits(sims), _bits(osims))
    gi.close()


def test_export_round_trip(eng, oracle_mod, built):
    V, o, lv = built(1200, 32, 5, 16)
    g = o.export()
    gi = eng.Index("foo", 32, 5, 16)
    gi.import_graph(g)
    g2 = gi.export_graph(with_vectors=False)
    ok, why = graphs_equal(g, g2)
    assert ok, why
    assert gi.neighbors(0, 0).tolist() == o.neighbors(0, 0).tolist()
    gi.close()


# ---- core_tests.rs hnsw_test through the GPU ------------------------------------
def test_hnsw_test_reference_kat(eng):
    """core_tests.rs:12-53 with the engine in place of Index<f32,f32>."""
    index = eng.Index("foo", 4, 5, 16, seed=3)
    assert index.name == "foo" and index.data_dim == 4 and index.m == 5 and index.ef_construction == 16
    assert index.node_count == 0 and index.max_layer == 0 and index.enterpoint is None
    assert index.search_knn(np.zeros(4, np.float32), 5) == []       # core.rs:481-483
    calls = []
    for i in range(100):
        index.add_node("node%d" % i, np.full(4, float(i), np.float32), lambda s, n: calls.append(s))
    assert index.node_count == 100 and index.enterpoint is not None
    res = index.search_knn(np.full(4, 10.0, np.float32), 5)
    assert len(res) == 5
    assert abs(res[0].sim - 0.0) < EPS and res[0].name == "node10"
    for r, want in zip(res[1:], [-4.0, -4.0, -16.0, -16.0]):
        assert abs(r.sim - want) < EPS
    assert len(calls) > 0
    index.close()


def test_reply_name_is_last_dot_segment_and_inputs_are_cast_to_f32(eng):
    """core.rs:885-887 (result name = last '.'-separated segment of the node key, the module stores
    "hnsw.{idx}.{node}") and src/lib.rs:345-346,469-470 (f64 arguments are cast with `as f32`)"""
    index = eng.Index("foo", 4, 5, 16)
    third = 1.0 / 3.0                                     # not representable in f32
    index.add_node("hnsw.foo.alpha", np.array([third] * 4, dtype=np.float64))
    index.add_node("hnsw.foo.beta", np.array([2.0] * 4, dtype=np.float64))
    res = index.search_knn(np.array([third] * 4, dtype=np.float64), 2)
    assert [r.name for r in res] == ["alpha", "beta"]
    assert res[0].sim == 0.0                              # both sides rounded the same way
    d = np.float32(2.0) - np.float32(third)
    assert res[1].sim == -float(np.float32(np.float32(np.float32(d * d) + np.float32(d * d)) + np.float32(d * d)) + np.float32(d * d))
    index.close()


def test_errors_match_reference(eng):
    from redis_hnsw_amd import HNSWError
    index = eng.Index("foo", 4, 5, 16)
    with pytest.raises(HNSWError) as e:
        index.add_node("a", np.zeros(3, np.float32))
    assert e.value.error_string() == 'String("data dimension: 3 does not match Index")'   # core.rs:390
    index.add_node("a", np.zeros(4, np.float32))
    index.add_node("b", np.ones(4, np.float32))
    with pytest.raises(HNSWError) as e:
        index.add_node("b", np.ones(4, np.float32))
    assert e.value.error_string() == 'String("Node: \\"b\\" already exists")'             # core.rs:408
    with pytest.raises(HNSWError) as e:
        index.search_knn(np.zeros(5, np.float32), 1)
    assert e.value.error_string() == 'String("data dimension: 5 does not match Index")'   # core.rs:479
    # k larger than the index: min(k, ef, reachable) results (core.rs:878-890)
    assert len(index.search_knn(np.zeros(4, np.float32), 10)) == 2
    # non-finite components are refused (documented limit: the reference's NaN ordering is not reproduced)
    with pytest.raises(HNSWError):
        index.add_node("c", np.array([0, np.nan, 0, 0], np.float32))
    with pytest.raises(HNSWError):
        index.search_knn(np.array([0, np.inf, 0, 0], np.float32), 1)
    assert index.node_count == 2
    index.close()


# ---- exact insert: link-for-link identical graphs ---------------------------------
@pytest.mark.parametrize("single_window", [1, 0])
@pytest.mark.parametrize("n,dim,m,ef", [(400, 32, 5, 16), (700, 128, 16, 200), (300, 4, 5, 16), (500, 64, 6, 40)])
def test_exact_insert_builds_identical_graph(eng, oracle_mod, n, dim, m, ef, single_window):
    """single hnsw_add calls: as a one-node window (the default: plan, speculative shrinks in parallel, validated
    commit) and through the serial kernels -- same graph, same update_fn sets; the serial kernels' work counters
    are the reference's evaluation counts"""
    V = make_data(n, dim, seed=11)
    lv = oracle_mod.draw_levels(n, m, 5)
    o = oracle_mod.OracleIndex(dim, m, ef)
    gi = eng.Index("foo", dim, m, ef)
    gi.set_tuning("select_shortcut", 0)     # the full select_neighbors extension: its evaluations are the reference's
    gi.set_tuning("single_window", single_window)
    for i in range(n):
        if i % 7 == 0:   # compare the touched sets on a sample
            oid, ot = o.add(V[i], int(lv[i]), want_touched=True)
            got = []
            gi.add_node("node%d" % i, V[i], lambda s, nid: got.append(nid), level=int(lv[i]))
            assert sorted(got) == sorted(ot.tolist()), "touched set of insert %d" % i
        else:
            o.add(V[i], int(lv[i]))
            gi.add_node("node%d" % i, V[i], level=int(lv[i]))
    ok, why = graphs_equal(o.export(), gi.export_graph())
    assert ok, why
    # the work accounting matches too: evaluations done + the econn evaluations skipped (the window counts the work
    # it did -- speculative shrinks see the rows of the snapshot, so their pools can differ by irrelevant ids)
    _, ic = gi.counters()
    oc = o.insert_counters()
    if not single_window:
        assert ic.n_dist + ic.n_spill == oc.n_dist
    else:
        assert 0.9 * oc.n_dist <= ic.n_dist + ic.n_spill <= 1.3 * oc.n_dist
    gi.close()


@pytest.mark.parametrize("n,dim,m,ef", [(700, 128, 16, 200), (500, 32, 8, 8), (400, 16, 5, 3), (300, 4, 5, 16)])
def test_select_is_the_head_of_W(eng, oracle_mod, n, dim, m, ef):
    """select_neighbors right after search_level(ef >= m) is the m nearest of W (hnsw_insert.hpp): the default path
    skips the extension's evaluations and must still build the reference's graph -- including ef == m, and
    ef < m where the shortcut must NOT apply -- with fewer distance evaluations than the reference makes."""
    V = make_data(n, dim, seed=12)
    lv = oracle_mod.draw_levels(n, m, 5)
    o = oracle_mod.OracleIndex(dim, m, ef)
    gi = eng.Index("foo", dim, m, ef)
    gi.set_tuning("occ_window", 0)
    for i in range(n):
        o.add(V[i], int(lv[i]))
        gi.add_node("node%d" % i, V[i], level=int(lv[i]))
    ok, why = graphs_equal(o.export(), gi.export_graph())
    assert ok, why
    _, ic = gi.counters()
    oc = o.insert_counters()
    if ef >= m:
        assert ic.n_dist + ic.n_spill < oc.n_dist
    else:
        assert ic.n_dist + ic.n_spill == oc.n_dist
    gi.close()


def test_exact_batch_insert_equals_oracle(eng, oracle_mod):
    n, dim, m, ef = 600, 128, 16, 200
    V = make_data(n, dim, seed=12)
    lv = oracle_mod.draw_levels(n, m, 9)
    o = oracle_mod.OracleIndex(dim, m, ef)
    o.add_batch(V, lv)
    gi = eng.Index("foo", dim, m, ef)
    gi.add_batch(V, levels=lv, mode="exact")
    ok, why = graphs_equal(o.export(), gi.export_graph())
    assert ok, why
    Q = make_data(32, dim, seed=2)
    ids, sims, _ = gi.search_batch(Q, 10)
    oids, osims, _, _ = o.search_batch(Q, 10)
    assert np.array_equal(ids, oids) and np.array_equal(_bits(sims), _bits(osims))
    gi.close()


def test_fast_build_with_spilling_visited_sets(eng, oracle_mod):
    """The fast build's plan kernel with an LDS table far too small: every
    search / select spills to HBM while the index (and the spill tables) grow."""
    n, dim, m, ef, k = 3000, 32, 8, 64, 10
    V = make_data(n, dim, seed=4)
    gi = eng.Index("foo", dim, m, ef)
    gi.set_tuning("lds_hash_bits", 7)
    gi.set_tuning("fast_seed", 64)
    gi.add_batch(V, mode="fast")
    assert gi.node_count == n
    Q = make_data(64, dim, seed=2)
    ids, sims, n_out = gi.search_batch(Q, k)
    g = gi.export_graph()
    g["vectors"] = V
    o2 = oracle_mod.OracleIndex.from_graph(dim, m, ef, g)
    oids2, osims2, _, _ = o2.search_batch(Q, k)
    assert np.array_equal(ids, oids2) and np.array_equal(_bits(sims), _bits(osims2))
    # the reference-order build of the same data reaches 0.884 recall@10 (oracle, seed 0 levels)
    assert recall_at_k(ids, brute_force_topk(V, Q, k)) > 0.85
    gi.close()


def test_cpp_host_mirror_runs_reference_tests():
    """tests/cpp/hnsw_test.cpp = core_tests.rs + metrics_tests.rs over the C++ host mirror"""
    import subprocess
    from redis_hnsw_amd import build
    exe = build.build_host_test()
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "hnsw_test ok" in r.stdout


# ---- committed golden vectors -------------------------------------------------------
from tests.golden_util import golden_cases, load_golden, load_transcribed, transcribed_cases  # noqa: E402


@pytest.mark.parametrize("name", golden_cases())
def test_engine_reproduces_golden(eng, name):
    """exact GPU insert -> the golden graph; GPU search -> golden ids / similarity bits / counters"""
    c = load_golden(name)
    gi = eng.Index("g", c["dim"], c["m"], c["ef"])
    nf = c["n_first"]
    gi.add_batch(c["V"][:nf], levels=c["levels"][:nf], mode="exact")
    for i in c["deleted"]:
        gi.delete_node("node%d" % int(i))
    if nf < c["n"]:
        gi.add_batch(c["V"][nf:], levels=c["levels"][nf:], mode="exact")
    ok, why = graphs_equal(c["graph"], gi.export_graph())
    assert ok, why
    gi.reset_counters()
    ids, sims, n_out = gi.search_batch(c["Q"], c["k"])
    assert np.array_equal(ids, c["ids"]) and np.array_equal(_bits(sims), c["sims_bits"])
    assert np.array_equal(n_out, c["n_out"])
    sc, _ = gi.counters()
    assert [sc.n_dist, sc.n_ids, sc.n_expand] == c["search_counters"].tolist()
    gi.close()


@pytest.mark.parametrize("name", transcribed_cases("exact"))
def test_engine_reproduces_the_transcription(eng, name):
    """The independent witness (tests/transcription/: core.rs transcribed into Python, nothing of the oracle in it):
    HNSW.NODE.ADD on the GPU builds the transcription's graph row for row -- through the windowed batch form AND, for
    a prefix, one hnsw_add per call -- and HNSW.SEARCH gives its ids, similarity bits and work counters."""
    c = load_transcribed(name)
    gi = eng.Index("t", c["dim"], c["m"], c["ef"])
    gi.set_tuning("select_shortcut", 0)              # the reference's full select_neighbors: its counters are the file's
    n_single = 300
    for i in range(n_single):                        # the command's shape (src/lib.rs:356)
        gi.add_node("node%d" % i, c["V"][i], level=int(c["levels"][i]))
    gi.add_batch(c["V"][n_single:], names=["node%d" % i for i in range(n_single, c["n"])],
                 levels=c["levels"][n_single:], mode="exact")
    ok, why = graphs_equal(c["graph"], gi.export_graph())
    assert ok, why
    gi.reset_counters()
    ids, sims, n_out = gi.search_batch(c["Q"], c["k"])
    assert np.array_equal(n_out, c["n_out"]) and np.array_equal(ids, c["ids"])
    assert np.array_equal(_bits(sims), c["sims_bits"])
    sc, _ = gi.counters()
    assert [sc.n_dist, sc.n_ids, sc.n_expand] == c["search_counters"].tolist()
    gi.close()


# ---- fast build: recall parity ------------------------------------------------------
def test_fast_build_recall_parity(eng, oracle_mod, built):
    n, dim, m, ef, k = 6000, 32, 16, 200, 10
    V = make_data(n, dim, seed=1)
    o, lv = build_oracle(oracle_mod, V, m, ef)
    Q = make_data(256, dim, seed=2)
    gt = brute_force_topk(V, Q, k)
    oids, _, _, _ = o.search_batch(Q, k)
    r_ref = recall_at_k(oids, gt)
    gi = eng.Index("foo", dim, m, ef)
    gi.add_batch(V, levels=lv, mode="fast")
    assert gi.node_count == n
    ids, sims, n_out = gi.search_batch(Q, k)
    r_fast = recall_at_k(ids, gt)
    assert r_fast >= r_ref - 0.02, (r_fast, r_ref)
    # the GPU search on the GPU-built graph is still exactly what the oracle's search does on that graph
    g = gi.export_graph(with_vectors=False)
    g["vectors"] = V
    o2 = oracle_mod.OracleIndex.from_graph(dim, m, ef, g)
    oids2, osims2, _, _ = o2.search_batch(Q, k)
    assert np.array_equal(ids, oids2) and np.array_equal(_bits(sims), _bits(osims2))
    gi.close()


# ---- edge cases --------------------------------------------------------------------------
def test_search_edge_cases(eng, oracle_mod, built):
    n, dim, m, ef = 1200, 32, 5, 16
    V, o, lv = built(n, dim, m, ef)
    gi = eng.Index("foo", dim, m, ef)
    gi.import_graph(o.export())
    # k larger than ef: min(k, ef, reachable) results (core.rs:878-890), padded rows
    Q = make_data(5, dim, seed=9)
    ids, sims, n_out = gi.search_batch(Q, 40)
    oids, osims, on, _ = o.search_batch(Q, 40)
    assert np.array_equal(n_out, on) and int(n_out.max()) <= ef
    for i in range(5):
        c = int(on[i])
        assert np.array_equal(ids[i, :c], oids[i, :c]) and np.all(ids[i, c:] == 0xFFFFFFFF)
        assert np.all(np.isneginf(sims[i, c:]))
    # a query equal to a stored vector: similarity is -0.0 exactly like -(+0.0) on the CPU (metrics.rs:75)
    ids, sims, n_out = gi.search_batch(V[17:18], 3)
    oids, osims, _, _ = o.search_batch(V[17:18], 3)
    assert ids[0, 0] == 17 and np.array_equal(_bits(sims), _bits(osims))
    assert _bits(sims)[0, 0] == 0x80000000
    # batch sizes that are not multiples of anything, and larger than the resident grid
    for B in (1, 3, 65, 2500):
        Qb = make_data(B, dim, seed=20 + B)
        ids, sims, n_out = gi.search_batch(Qb, 5)
        oids, osims, on, _ = o.search_batch(Qb, 5)
        assert np.array_equal(n_out, on)
        # some nodes of this M=5 graph end up with an empty layer-0 row (the reference's shrink can strip
        # a node bare), so a few queries reach fewer than k results: compare the valid prefix only
        valid = np.arange(5)[None, :] < on[:, None]
        assert np.array_equal(ids[valid], oids[valid]) and np.array_equal(_bits(sims)[valid], _bits(osims)[valid])
        assert np.all(ids[~valid] == 0xFFFFFFFF)
    gi.close()


def test_import_then_exact_insert_continues_identically(eng, oracle_mod):
    """make_index-style bulk load (src/lib.rs:252-315) followed by HNSW.NODE.ADD: same graph as the oracle doing both"""
    n0, n1, dim, m, ef = 500, 150, 32, 6, 24
    V = make_data(n0 + n1, dim, seed=31)
    lv = oracle_mod.draw_levels(n0 + n1, m, 8)
    o = oracle_mod.OracleIndex(dim, m, ef)
    o.add_batch(V[:n0], lv[:n0])
    gi = eng.Index("foo", dim, m, ef)
    gi.import_graph(o.export())
    for i in range(n0, n0 + n1):
        o.add(V[i], int(lv[i]))
        gi.add_node("node%d" % i, V[i], level=int(lv[i]))
    ok, why = graphs_equal(o.export(), gi.export_graph())
    assert ok, why
    gi.close()


def test_exact_insert_with_duplicate_vectors_and_ties(eng, oracle_mod):
    """every vector stored three times: all similarities tie in triples; the (sim, id) order decides"""
    base = make_data(120, 8, seed=41)
    V = np.concatenate([base, base, base])
    n, dim, m, ef = V.shape[0], 8, 4, 12
    lv = oracle_mod.draw_levels(n, m, 2)
    o = oracle_mod.OracleIndex(dim, m, ef)
    o.add_batch(V, lv)
    gi = eng.Index("foo", dim, m, ef)
    gi.add_batch(V, levels=lv, mode="exact")
    ok, why = graphs_equal(o.export(), gi.export_graph())
    assert ok, why
    ids, sims, n_out = gi.search_batch(base[:32], 6)
    oids, osims, on, _ = o.search_batch(base[:32], 6)
    assert np.array_equal(ids, oids) and np.array_equal(_bits(sims), _bits(osims))
    gi.close()


def test_exact_insert_after_fast_build(eng, oracle_mod):
    """HNSW.NODE.ADD on a fast-built (not necessarily symmetric) graph must work and stay searchable"""
    n, dim, m, ef = 3000, 32, 8, 64
    V = make_data(n + 50, dim, seed=51)
    gi = eng.Index("foo", dim, m, ef)
    gi.add_batch(V[:n], mode="fast")
    for i in range(n, n + 50):
        gi.add_node("node%d" % i, V[i])
    assert gi.node_count == n + 50
    res = gi.search_knn(V[n + 7], 1)
    assert res[0].id == n + 7 and res[0].sim == 0.0
    gi.close()


@pytest.mark.parametrize("n,dim,m,ef,k", [(400, 32, 32, 64, 20), (500, 64, 8, 600, 50), (300, 128, 12, 1024, 10)])
def test_wide_rows_and_large_ef(eng, oracle_mod, n, dim, m, ef, k):
    """M=32 (rows wider than one 64-lane chunk) and ef up to the 1024 limit (R = 16 list slices):
    exact insert and search against the oracle"""
    V = make_data(n, dim, seed=81)
    lv = oracle_mod.draw_levels(n, m, 6)
    o = oracle_mod.OracleIndex(dim, m, ef)
    o.add_batch(V, lv)
    gi = eng.Index("foo", dim, m, ef)
    gi.add_batch(V, levels=lv, mode="exact")
    ok, why = graphs_equal(o.export(), gi.export_graph())
    assert ok, why
    Q = make_data(24, dim, seed=2)
    gi.reset_counters()
    ids, sims, n_out = gi.search_batch(Q, k)
    oids, osims, on, oct = o.search_batch(Q, k)
    assert np.array_equal(n_out, on)
    valid = np.arange(k)[None, :] < on[:, None]
    assert np.array_equal(ids[valid], oids[valid]) and np.array_equal(_bits(sims)[valid], _bits(osims)[valid])
    sc, _ = gi.counters()
    assert (sc.n_dist, sc.n_ids, sc.n_expand) == (oct.n_dist, oct.n_ids, oct.n_expand)
    gi.close()


@pytest.mark.parametrize("n,dim,m,ef,k", [(2000, 32, 8, 2000, 50), (1500, 128, 16, 3000, 10), (1200, 12, 5, 1500, 20)])
def test_ef_construction_beyond_1024(eng, oracle_mod, n, dim, m, ef, k):
    """The reference takes any EFCON (core.rs:322-347, src/lib.rs:39-56).  Beyond 1024 the engine keeps W in LDS
    instead of registers (up to 4096: the slow, exact form): HNSW.NODE.ADD one call at a time and as a batch,
    HNSW.NODE.DEL and HNSW.SEARCH against the oracle -- graphs row for row, answers and counters bit for bit.
    ef larger than the index (every search returns the whole graph) and the scalar metric order included."""
    V = make_data(n, dim, seed=83)
    lv = oracle_mod.draw_levels(n, m, 6)
    o = oracle_mod.OracleIndex(dim, m, ef)
    o.add_batch(V, lv)
    gi = eng.Index("bigef", dim, m, ef)
    n1 = 40
    for i in range(n1):
        gi.add_node("node%d" % i, V[i], level=int(lv[i]))
    gi.add_batch(V[n1:], names=["node%d" % i for i in range(n1, n)], levels=lv[n1:], mode="exact")
    ok, why = graphs_equal(o.export(), gi.export_graph())
    assert ok, why
    for v in (3, 77, 500):
        o.delete(v)
        gi.delete_node("node%d" % v)
    ok, why = graphs_equal(o.export(), gi.export_graph())
    assert ok, why
    Q = make_data(24, dim, seed=2)
    gi.reset_counters()
    ids, sims, n_out = gi.search_batch(Q, k)
    oids, osims, on, oct = o.search_batch(Q, k)
    assert np.array_equal(n_out, on)
    valid = np.arange(k)[None, :] < on[:, None]
    assert np.array_equal(ids[valid], oids[valid]) and np.array_equal(_bits(sims)[valid], _bits(osims)[valid])
    sc, _ = gi.counters()
    assert (sc.n_dist, sc.n_ids, sc.n_expand) == (oct.n_dist, oct.n_ids, oct.n_expand)
    gi.close()


@pytest.mark.parametrize("n,dim,m,ef,tunings", [
    (900, 32, 32, 1500, {}),                       # nodes of level >= 1 hold more than 64 links over their layers: the speculative delete declines
    (900, 32, 48, 1200, {"single_window": 0}),     # the serial delete kernel for every node
    (700, 32, 16, 2000, {"occ_window": 0}),
    (600, 32, 70, 1100, {}),                       # M > 64: serial kernels only
])
def test_deletes_beyond_ef_1024_take_the_serial_kernel_too(eng, oracle_mod, n, dim, m, ef, tunings):
    """ef_construction > 1024 keeps W in LDS (R = 64).  HNSW.NODE.DEL falls back to the one-wave k_delete_exact whenever the
    speculative form declines -- a node with more than 64 neighbours over all its layers, single_window = 0, no window,
    M > 64 -- and that kernel has to exist for R = 64 as well (round 4's advisor finding: it did not)."""
    V = make_data(n, dim, seed=85)
    lv = oracle_mod.draw_levels(n, m, 9)
    o = oracle_mod.OracleIndex(dim, m, ef)
    o.add_batch(V, lv)
    gi = eng.Index("del-bigef", dim, m, ef)
    for key, val in tunings.items():
        gi.set_tuning(key, val)
    gi.add_batch(V, levels=lv, mode="exact")
    ok, why = graphs_equal(o.export(), gi.export_graph())
    assert ok, why
    upper = [int(i) for i in np.nonzero(lv >= 1)[0][:6]]
    for v in upper + [5, 123, 400]:
        if not o.is_live(v):
            continue
        o.delete(v)
        gi.delete_node("node%d" % v)
        ok, why = graphs_equal(o.export(), gi.export_graph())
        assert ok, "after deleting %d: %s" % (v, why)
    gi.close()
    o.close()


def test_restride_keeps_the_graph(eng, oracle_mod):
    """widening the adjacency tables (what the engine does when degrees approach the row capacity)
    in the middle of an exact build must not change anything"""
    n, dim, m, ef = 500, 32, 6, 32
    V = make_data(n, dim, seed=91)
    lv = oracle_mod.draw_levels(n, m, 3)
    o = oracle_mod.OracleIndex(dim, m, ef)
    o.add_batch(V, lv)
    gi = eng.Index("foo", dim, m, ef)
    s0 = gi.info().stride0
    for i in range(n):
        gi.add_node("node%d" % i, V[i], level=int(lv[i]))
        if i in (100, 300):
            gi.set_tuning("force_restride", 16)
    assert gi.info().stride0 == s0 + 32
    ok, why = graphs_equal(o.export(), gi.export_graph())
    assert ok, why
    Q = make_data(16, dim, seed=2)
    ids, sims, _ = gi.search_batch(Q, 5)
    oids, osims, _, _ = o.search_batch(Q, 5)
    assert np.array_equal(ids, oids) and np.array_equal(_bits(sims), _bits(osims))
    gi.close()


def test_many_layers_small_m(eng, oracle_mod):
    """M=2 gives level_mult = 1/ln 2: tall hierarchies (10+ layers) and tiny rows"""
    n, dim, m, ef = 600, 32, 2, 8
    V = make_data(n, dim, seed=61)
    lv = oracle_mod.draw_levels(n, m, 3)
    assert lv.max() >= 7
    o = oracle_mod.OracleIndex(dim, m, ef)
    o.add_batch(V, lv)
    gi = eng.Index("foo", dim, m, ef)
    gi.add_batch(V, levels=lv, mode="exact")
    ok, why = graphs_equal(o.export(), gi.export_graph())
    assert ok, why
    Q = make_data(40, dim, seed=2)
    ids, sims, _ = gi.search_batch(Q, 5)
    oids, osims, _, _ = o.search_batch(Q, 5)
    assert np.array_equal(ids, oids) and np.array_equal(_bits(sims), _bits(osims))
    gi.close()


# ---- HNSW.NODE.DEL (core.rs:414-475) ------------------------------------------------------
def test_hnsw_test_delete_reference_kat(eng):
    """core_tests.rs:55-80 through the engine"""
    from redis_hnsw_amd import HNSWError
    index = eng.Index("foo", 4, 5, 16, seed=5)
    n = 100
    for i in range(n):
        index.add_node("node%d" % i, np.full(4, float(i), np.float32))
    for i in range(n):
        index.delete_node("node%d" % i, lambda s, nid: None)
        assert index.node_count == n - i - 1
        g = index.export_graph()
        for col in g["col"]:
            assert i not in col.tolist()
    assert index.enterpoint is None and index.search_knn(np.zeros(4, np.float32), 3) == []
    with pytest.raises(HNSWError) as e:
        index.delete_node("node3")
    assert e.value.error_string() == 'String("Node: \\"node3\\" does not exist")'     # core.rs:421
    index.add_node("again", np.zeros(4, np.float32))                                # core.rs:393-405
    assert index.node_count == 1 and index.enterpoint == "again"
    index.close()


@pytest.mark.parametrize("single_window", [1, 0])
@pytest.mark.parametrize("n,dim,m,ef", [(300, 32, 5, 16), (400, 128, 16, 64), (250, 4, 5, 16), (600, 128, 24, 100)])
def test_delete_matches_oracle(eng, oracle_mod, n, dim, m, ef, single_window):
    """HNSW.NODE.DEL with the neighbours' re-selections computed speculatively in parallel and applied in order
    (the default) and through the one-wave kernel: graphs, update_fn sets, searches and enterpoint re-election
    equal the oracle's.  (M = 24: rows of up to 48 ids -- nodes with more than 32 neighbours fall back to the
    one-wave kernel inside the default path.)"""
    V = make_data(n + 60, dim, seed=71)
    lv = oracle_mod.draw_levels(n + 60, m, 4)
    o = oracle_mod.OracleIndex(dim, m, ef)
    gi = eng.Index("foo", dim, m, ef)
    gi.set_tuning("single_window", single_window)
    o.add_batch(V[:n], lv[:n])
    gi.add_batch(V[:n], levels=lv[:n], mode="exact")
    order = np.random.default_rng(5).permutation(n)[: n // 2]
    ep0 = o.enterpoint
    order = np.concatenate([[ep0], order[order != ep0]])          # delete the enterpoint first (re-election)
    for step, i in enumerate(order):
        ot = o.delete(int(i), want_touched=True)
        got = []
        gi.delete_node("node%d" % i, lambda s, nid: got.append(nid))
        assert sorted(got) == sorted(ot.tolist()), "touched set of delete %d" % i
        if step % 25 == 0 or step == len(order) - 1:
            ok, why = graphs_equal(o.export(), gi.export_graph())
            assert ok, "after deleting %d nodes: %s" % (step + 1, why)
    assert gi.node_count == o.live_count
    Q = make_data(48, dim, seed=2)
    ids, sims, n_out = gi.search_batch(Q, 5)
    oids, osims, on, _ = o.search_batch(Q, 5)
    assert np.array_equal(n_out, on) and np.array_equal(ids, oids) and np.array_equal(_bits(sims), _bits(osims))
    assert not (set(ids.ravel().tolist()) & set(int(x) for x in order))
    # inserts after deletes keep matching (ids are not reused)
    for i in range(n, n + 60):
        o.add(V[i], int(lv[i]))
        gi.add_node("node%d" % i, V[i], level=int(lv[i]))
    ok, why = graphs_equal(o.export(), gi.export_graph())
    assert ok, why
    gi.close()


# ---- BASELINE.json's full size (C2): properties + sampled oracle parity ----------------------
def test_full_size_c2_properties_and_sampled_parity(eng, oracle_mod):
    """1M x 128, M=16, ef=200, k=10, batch 1024 (fast GPU build): size-independent properties on the
    whole batch, and bit-exact parity with the oracle searching the same exported graph on a sample."""
    from bench import draw_levels
    N, dim, M, ef, k, B = 1_000_000, 128, 16, 200, 10, 1024
    V = np.random.default_rng(1).random((N, dim), dtype=np.float32)
    Q = np.random.default_rng(2).random((B, dim), dtype=np.float32)
    gi = eng.Index("c2", dim, M, ef)
    gi.add_batch(V, levels=draw_levels(N, M, 7), mode="fast")
    assert gi.node_count == N
    ids, sims, n_out = gi.search_batch(Q, k)
    assert np.all(n_out == k)                                              # min(k, ef, reachable) = k
    assert np.all(ids < N)
    assert np.all(sims[:, :-1] >= sims[:, 1:])                             # nearest first (core.rs:878-890)
    assert np.all(sims <= 0)                                               # sim = -(squared L2)
    assert all(len(set(r.tolist())) == k for r in ids)                     # no node twice
    # the similarity reported for an id is the metric of that pair, recomputed independently in f64
    d = ((Q[:64, None, :].astype(np.float64) - V[ids[:64].astype(np.int64)].astype(np.float64)) ** 2).sum(-1)
    assert np.allclose(-d, sims[:64], rtol=1e-5, atol=0)
    ids2, sims2, _ = gi.search_batch(Q, k)                                 # idempotent
    assert np.array_equal(ids, ids2) and np.array_equal(_bits(sims), _bits(sims2))
    half, _, _ = gi.search_batch(Q[:100], k)                               # batch composition does not matter
    assert np.array_equal(half, ids[:100])
    # a stored vector that is found reports similarity -0.0 (HNSW does not promise to find it: on this
    # data the reference algorithm's recall@10 is 0.27)
    self_ids, self_sims, _ = gi.search_batch(V[12345:12346], 1)
    assert (self_ids[0, 0] == 12345) == (_bits(self_sims)[0, 0] == 0x80000000)
    # sampled parity: the oracle searching the very same graph
    g = gi.export_graph()
    g["vectors"] = V
    o = oracle_mod.OracleIndex.from_graph(dim, M, ef, g)
    gi.reset_counters()
    sids, ssims, _ = gi.search_batch(Q[:48], k)
    oids, osims, on, oct = o.search_batch(Q[:48], k, threads=8)
    assert np.array_equal(sids, oids) and np.array_equal(_bits(ssims), _bits(osims))
    sc, _ = gi.counters()
    assert (sc.n_dist, sc.n_ids, sc.n_expand) == (oct.n_dist, oct.n_ids, oct.n_expand)
    gi.close()


# ---- snapshot / restore -------------------------------------------------------------------
def test_snapshot_round_trip_continues_identically(eng, oracle_mod):
    n, dim, m, ef = 400, 32, 6, 32
    V = make_data(n + 80, dim, seed=101)
    a = eng.Index("snap", dim, m, ef, seed=9)
    for i in range(n):
        a.add_node("hnsw.snap.n%d" % i, V[i])              # levels drawn by the engine's generator
    for i in (3, 77, a.enterpoint_id, 250):
        a.delete_node("hnsw.snap.n%d" % i)
    blob = a.serialize()
    b = eng.Index.deserialize(blob)
    assert (b.name, b.data_dim, b.m, b.ef_construction, b.node_count) == ("snap", dim, m, ef, a.node_count)
    assert b.enterpoint == a.enterpoint and b.max_layer == a.max_layer
    ok, why = graphs_equal(a.export_graph(), b.export_graph())
    assert ok, why
    Q = make_data(32, dim, seed=2)
    ia, sa, na = a.search_batch(Q, 5)
    ib, sb, nb = b.search_batch(Q, 5)
    assert np.array_equal(ia, ib) and np.array_equal(_bits(sa), _bits(sb)) and np.array_equal(na, nb)
    assert [r.name for r in a.search_knn(Q[0], 3)] == [r.name for r in b.search_knn(Q[0], 3)]
    # both continue with the same level draws and end with the same graph
    for i in range(n, n + 80):
        a.add_node("hnsw.snap.n%d" % i, V[i])
        b.add_node("hnsw.snap.n%d" % i, V[i])
    ok, why = graphs_equal(a.export_graph(), b.export_graph())
    assert ok, why
    a.close(); b.close()


# ---- BASELINE.json config 4 at full size on one GPU ------------------------------------------
def test_full_size_c4_10m_properties_and_sampled_parity(eng, oracle_mod):
    """10M x 128, M=16, ef=200, k=10 (one replica of C4; fast GPU build): ids above 2^23, the 24-bit tag
    hash, size-independent properties on a 1024-query batch, and bit-exact parity with the oracle searching
    the same exported graph on a sample."""
    from bench import draw_levels
    N, dim, M, ef, k, B = 10_000_000, 128, 16, 200, 10, 1024
    V = np.random.default_rng(1).random((N, dim), dtype=np.float32)
    Q = np.random.default_rng(2).random((B, dim), dtype=np.float32)
    gi = eng.Index("c4", dim, M, ef)
    gi.add_batch(V, levels=draw_levels(N, M, 7), mode="fast")
    assert gi.node_count == N
    ids, sims, n_out = gi.search_batch(Q, k)
    assert np.all(n_out == k) and np.all(ids < N)
    assert ids.max() >= (1 << 23)                                          # the upper half of the id range is reachable
    assert np.all(sims[:, :-1] >= sims[:, 1:]) and np.all(sims <= 0)
    assert all(len(set(r.tolist())) == k for r in ids)
    d = ((Q[:64, None, :].astype(np.float64) - V[ids[:64].astype(np.int64)].astype(np.float64)) ** 2).sum(-1)
    assert np.allclose(-d, sims[:64], rtol=1e-5, atol=0)
    ids2, sims2, _ = gi.search_batch(Q, k)                                 # idempotent
    assert np.array_equal(ids, ids2) and np.array_equal(_bits(sims), _bits(sims2))
    # the 8-waves-per-CU launch shape (4096 queries) agrees with the 1024-query one on the shared queries
    Q4 = np.concatenate([Q, np.random.default_rng(3).random((3 * B, dim), dtype=np.float32)])
    ids4, sims4, _ = gi.search_batch(Q4, k)
    assert np.array_equal(ids4[:B], ids) and np.array_equal(_bits(sims4[:B]), _bits(sims))
    g = gi.export_graph()
    g["vectors"] = V
    o = oracle_mod.OracleIndex.from_graph(dim, M, ef, g)
    gi.reset_counters()
    sids, ssims, _ = gi.search_batch(Q[:16], k)
    oids, osims, on, oct = o.search_batch(Q[:16], k, threads=8)
    assert np.array_equal(sids, oids) and np.array_equal(_bits(ssims), _bits(osims))
    sc, _ = gi.counters()
    assert (sc.n_dist, sc.n_ids, sc.n_expand) == (oct.n_dist, oct.n_ids, oct.n_expand)
    gi.close()


# ---- exact-order PARALLEL insert (hnsw_occ.hpp): the window must reproduce the serial graph ----------
@pytest.mark.parametrize("n,dim,m,ef,window", [
    (3000, 32, 8, 64, 32),       # generic AVX dim
    (2500, 128, 16, 200, 16),    # the C2 shape
    (2500, 128, 16, 200, 64),
    (1500, 12, 4, 24, 8),        # scalar metric order, m = 4: many layers, enterpoint changes inside windows
    (1200, 768, 32, 400, 16),    # wide rows, R = 8
    (3000, 8, 2, 8, 16),         # m = 2: rows outgrow the initial stride inside a window (restride between rounds)
    (2500, 16, 3, 12, 32),
    (20000, 128, 16, 200, 32),   # C2's shape at 20 k nodes
])
def test_windowed_exact_build_is_the_serial_graph(eng, oracle_mod, n, dim, m, ef, window):
    """hnsw_add_batch(mode 0) plans a window of inserts in parallel and commits them in id order after
    validating each plan against the journal of row changes: the graph, the enterpoint and
    every adjacency row in stored order must be the oracle's, and searches on it bit-identical."""
    V = make_data(n, dim, seed=81)
    lv = oracle_mod.draw_levels(n, m, 5)
    o = oracle_mod.OracleIndex(dim, m, ef)
    o.add_batch(V, lv)
    gi = eng.Index("w", dim, m, ef)
    gi.set_tuning("occ_window", window)
    half = n // 2
    gi.add_batch(V[:half], levels=lv[:half], mode="exact")       # two batches: the second starts on a live graph
    gi.add_batch(V[half:], levels=lv[half:], mode="exact")
    ok, why = graphs_equal(o.export(), gi.export_graph())
    assert ok, why
    Q = make_data(32, dim, seed=2)
    ids, sims, n_out = gi.search_batch(Q, 5)
    oids, osims, on, _ = o.search_batch(Q, 5)
    assert np.array_equal(n_out, on)
    for i in range(len(Q)):                      # min(k, ef, reachable) results: m = 2 graphs have tiny components
        c = int(on[i])
        assert np.array_equal(ids[i, :c], oids[i, :c]) and np.array_equal(_bits(sims[i, :c]), _bits(osims[i, :c]))
    # single exact inserts and deletes keep working on the result
    W2 = make_data(20, dim, seed=82)
    for i in range(20):
        o.add(W2[i], 0)
        gi.add_node("late%d" % i, W2[i], level=0)
    o.delete(7)
    gi.delete_node("node7")
    ok, why = graphs_equal(o.export(), gi.export_graph())
    assert ok, why
    gi.close()


@pytest.mark.parametrize("n,dim,m,ef,window", [
    (2500, 128, 16, 200, 32),    # the C2 shape
    (1500, 12, 4, 24, 16),       # scalar metric order, m = 4: enterpoint changes inside groups
    (1200, 768, 32, 400, 16),    # wide rows, R = 8
    (3000, 8, 2, 8, 16),         # m = 2: restrides between rounds, rows that do not fill
    (12000, 32, 8, 64, 48),
])
def test_parallel_validated_commits_build_the_serial_graph(eng, oracle_mod, n, dim, m, ef, window):
    """commit_par = 2: every round's commits go through k_occ_commit_par (hnsw_occ_par.hpp) -- one workgroup per window
    node runs its whole commit in a private overlay of the rows it rewrites, nodes whose reads and rows the earlier
    nodes' deltas do not touch are applied together as a group (the rules were proven on the CPU,
    tests/experiments/occ_model.c PAR=1).  The graph must be the oracle's serial graph row for row, the same as with
    the in-order commit wave (commit_par = 0), with groups of more than one node actually formed -- and the same again
    with and without the two-stage plans (plan_split: the dim-128 shapes, where the specialised plan kernel runs)."""
    import ctypes as C
    V = make_data(n, dim, seed=91)
    lv = oracle_mod.draw_levels(n, m, 5)
    o = oracle_mod.OracleIndex(dim, m, ef)
    o.add_batch(V, lv)
    want = o.export()
    lib = eng._capi.load()
    lib.hnsw_debug_occ_par.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
    lib.hnsw_debug_occ.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
    for par, split in ((2, 1), (2, 0), (0, 1)):          # split: a far node's upper layers are planned a round early (OccSlot::stage)
        gi = eng.Index("p%d%d" % (par, split), dim, m, ef)
        gi.set_tuning("occ_window", window)
        gi.set_tuning("commit_par", par)
        gi.set_tuning("plan_split", split)
        half = n // 2
        gi.add_batch(V[:half], levels=lv[:half], mode="exact")
        gi.add_batch(V[half:], levels=lv[half:], mode="exact")
        ok, why = graphs_equal(want, gi.export_graph())
        assert ok, "commit_par=%d plan_split=%d: %s" % (par, split, why)
        pz, oc = (C.c_uint64 * 21)(), (C.c_uint64 * 16)()
        assert lib.hnsw_debug_occ_par(gi._h, pz) == 0 and lib.hnsw_debug_occ(gi._h, oc) == 0
        if par:
            assert pz[0] > 0 and oc[0] >= pz[0]
            if dim <= 128 and n >= 2500:     # (on the small dense shapes nearly every node conflicts with its predecessor)
                assert oc[0] > pz[0], "no group of more than one node was formed: %d commits in %d groups" % (oc[0], pz[0])
        else:
            assert pz[0] == 0
        Q = make_data(16, dim, seed=3)
        ids, sims, n_out = gi.search_batch(Q, 5)
        oids, osims, on, _ = o.search_batch(Q, 5)
        assert np.array_equal(n_out, on)
        for i in range(len(Q)):
            c = int(on[i])
            assert np.array_equal(ids[i, :c], oids[i, :c]) and np.array_equal(_bits(sims[i, :c]), _bits(osims[i, :c]))
        gi.close()
    o.close()


def test_group_commit_declines_when_its_grid_could_not_be_resident(eng, oracle_mod):
    """k_occ_commit_par separates its iterations with a spin barrier over the whole grid: every workgroup must be resident
    (each asks for nearly a CU's whole LDS).  The launcher asks the runtime how many are (occupancy x CUs of the device or
    partition) and leaves a round with more window nodes than that to the in-order commit kernel -- here the limit is
    pretended (tuning par_max_resident = 2): no group commit runs, the graph is the oracle's all the same."""
    import ctypes as C
    n, dim, m, ef = 2500, 128, 16, 200
    V = make_data(n, dim, seed=92)
    lv = oracle_mod.draw_levels(n, m, 6)
    o = oracle_mod.OracleIndex(dim, m, ef)
    o.add_batch(V, lv)
    lib = eng._capi.load()
    lib.hnsw_debug_occ_par.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
    for limit, expect_groups in ((2, False), (0, True)):                   # 0 = what the device really holds
        gi = eng.Index("res%d" % limit, dim, m, ef)
        gi.set_tuning("commit_par", 2)
        gi.set_tuning("par_max_resident", limit)
        gi.add_batch(V, levels=lv, mode="exact")
        ok, why = graphs_equal(o.export(), gi.export_graph())
        assert ok, why
        pz = (C.c_uint64 * 21)()
        assert lib.hnsw_debug_occ_par(gi._h, pz) == 0
        assert (pz[0] > 0) == expect_groups, (limit, pz[0])
        gi.close()
    o.close()


def test_windowed_exact_build_on_clustered_and_duplicated_data(eng, oracle_mod):
    """Not uniform: a 16-cluster Gaussian mixture (dense neighbourhoods, many more relevant row changes per
    window) with every 10th vector repeated (tied similarities: the validation counts ties as relevant)."""
    from bench import clustered
    n, dim, m, ef = 4000, 32, 8, 64
    centers = np.random.default_rng(3).random((16, dim), dtype=np.float32)
    V = clustered(n, dim, 4, centers)
    V[10::10] = V[9::10][: len(V[10::10])]
    lv = oracle_mod.draw_levels(n, m, 6)
    o = oracle_mod.OracleIndex(dim, m, ef)
    o.add_batch(V, lv)
    gi = eng.Index("c", dim, m, ef)
    gi.add_batch(V, levels=lv, mode="exact")
    ok, why = graphs_equal(o.export(), gi.export_graph())
    assert ok, why
    gi.close()


def test_windowed_build_survives_restrides(eng, oracle_mod):
    """The committing wave stops the round when a row could run out of room; the host widens the adjacency
    tables and the window continues with the plans it has.  Forced here by demanding extra room per row."""
    n, dim, m, ef = 1500, 32, 8, 48
    V = make_data(n, dim, seed=84)
    lv = oracle_mod.draw_levels(n, m, 5)
    o = oracle_mod.OracleIndex(dim, m, ef)
    o.add_batch(V, lv)
    gi = eng.Index("r", dim, m, ef)
    gi.set_tuning("occ_slack_extra", 30)
    s0 = gi.info().stride0
    gi.add_batch(V, levels=lv, mode="exact")
    assert gi.info().stride0 > s0                       # at least one restride happened inside the build
    ok, why = graphs_equal(o.export(), gi.export_graph())
    assert ok, why
    gi.close()


def test_windowed_build_falls_back_to_the_serial_kernels(eng, oracle_mod):
    """A plan whose read log does not fit cannot be validated: the window hands that node to the serial
    plan + commit kernels and starts a new epoch.  Forced for every node by a tiny log."""
    n, dim, m, ef = 500, 32, 6, 32
    V = make_data(n, dim, seed=85)
    lv = oracle_mod.draw_levels(n, m, 5)
    o = oracle_mod.OracleIndex(dim, m, ef)
    o.add_batch(V, lv)
    gi = eng.Index("s", dim, m, ef)
    gi.set_tuning("occ_log_cap", 8)
    gi.add_batch(V, levels=lv, mode="exact")
    ok, why = graphs_equal(o.export(), gi.export_graph())
    assert ok, why
    gi.close()


def test_windowed_and_serial_exact_builds_agree(eng):
    n, dim, m, ef = 2000, 64, 6, 48
    V = make_data(n, dim, seed=83)
    a = eng.Index("a", dim, m, ef, seed=3)
    b = eng.Index("b", dim, m, ef, seed=3)
    a.set_tuning("occ_window", 0)                 # the serial kernels, one insert after the other
    a.add_batch(V, mode="exact")                  # levels drawn by the engine: the same draws in both
    b.add_batch(V, mode="exact")
    ok, why = graphs_equal(a.export_graph(), b.export_graph())
    assert ok, why
    a.close(); b.close()


# ---- the multi-wave forms (round 4): same answers, same graphs, and they are the ones that run -----------------
@pytest.mark.parametrize("n,dim,m,ef,k,nq,wide", [(2000, 128, 16, 200, 10, 96, 0), (1200, 128, 24, 300, 20, 40, 0),
                                                  (1500, 128, 16, 40, 10, 64, 0), (1500, 128, 16, 200, 10, 48, 40)])
def test_two_wave_search_is_the_one_wave_search(eng, oracle_mod, built, n, dim, m, ef, k, nq, wide):
    """hnsw_search_duo.hpp: a walker and a W-keeper wavefront per query for small batches and single calls.  Same ids,
    similarity bits and work counters as the one-wave kernel and as the oracle -- narrow and wide adjacency rows, ef
    below and above 64, a batch and one query per call -- and it IS the form that runs for these shapes."""
    V, o, lv = built(n, dim, m, ef)
    Q = make_data(nq, dim, seed=2)
    want = o.search_batch(Q, k)
    valid = np.arange(k)[None, :] < want[2][:, None]
    gi = eng.Index("duo", dim, m, ef)
    gi.import_graph(o.export())
    if wide:
        gi.set_tuning("force_restride", wide)            # rows of 64..127 words: the two-row-word form
    for duo in (1, 0):
        gi.set_tuning("duo", duo)
        gi.reset_counters()
        ids, sims, n_out = gi.search_batch(Q, k)
        assert gi.last_search_was_duo() == bool(duo)
        assert np.array_equal(n_out, want[2]) and np.array_equal(ids[valid], want[0][valid])
        assert np.array_equal(_bits(sims)[valid], _bits(want[1])[valid])
        sc, _ = gi.counters()
        assert (sc.n_dist, sc.n_ids, sc.n_expand) == (want[3].n_dist, want[3].n_ids, want[3].n_expand)
    gi.set_tuning("duo", 1)
    for q in Q[:10]:
        assert [r.id for r in gi.search_knn(q, k)] == o.search(q, k)[0].tolist()
        assert gi.last_search_was_duo()
    gi.close()


@pytest.mark.parametrize("plan_duo,commit_team", [(1, 1), (0, 1), (1, 0), (0, 0)])
def test_two_wave_plans_and_four_wave_commits_build_the_reference_graph(eng, oracle_mod, plan_duo, commit_team):
    """HNSW.NODE.ADD / HNSW.NODE.DEL with the insert plans in their two-wave form and the commit kernels as a team of four
    wavefronts (and each of them switched off): the same graph as the oracle's, row for row, through a windowed batch,
    single adds and deletes -- including deletes of nodes with more than 32 neighbours (64 speculative records)."""
    n, dim, m, ef = 3000, 128, 16, 100
    V = make_data(n + 60, dim, seed=85)
    lv = oracle_mod.draw_levels(n + 60, m, 4)
    o = oracle_mod.OracleIndex(dim, m, ef)
    o.add_batch(V, lv)
    gi = eng.Index("team", dim, m, ef)
    gi.set_tuning("plan_duo", plan_duo)
    gi.set_tuning("commit_team", commit_team)
    gi.add_batch(V[:n], levels=lv[:n], mode="exact")
    for i in range(n, n + 60):
        gi.add_node("node%d" % i, V[i], level=int(lv[i]))
    ok, why = graphs_equal(o.export(), gi.export_graph())
    assert ok, why
    deg = [len(o.neighbors(i, 0)) for i in range(n)]
    victims = [int(i) for i in np.argsort(deg)[-6:]] + [5, 17, 600]          # the widest rows first
    assert max(deg) > 32
    for v in victims:
        o.delete(v)
        gi.delete_node("node%d" % v)
    ok, why = graphs_equal(o.export(), gi.export_graph())
    assert ok, why
    gi.close()


@pytest.mark.parametrize("commit_team", [1, 0])
@pytest.mark.parametrize("n,dim,m,ef", [(1200, 768, 16, 400), (1200, 768, 32, 100)])
def test_dim768_deletes_equal_the_oracle_after_every_delete(eng, oracle_mod, n, dim, m, ef, commit_team):
    """The dim-768 variant's HNSW.NODE.DEL, graph compared after EVERY delete.  These are the cases on which the
    four-wave commit once accepted stale speculative re-selections: a delete writes more journal entries than its LDS
    mirror holds, the validation reads the rest back from HBM, and the wave-level synchronisation did not wait for the
    stores (hnsw_wave_sync.hpp).  Nodes with more neighbours than m_max0 among the victims."""
    V = make_data(n, dim, seed=81)
    lv = oracle_mod.draw_levels(n, m, 6)
    o = oracle_mod.OracleIndex(dim, m, ef)
    o.add_batch(V, lv)
    gi = eng.Index("d768", dim, m, ef)
    gi.set_tuning("commit_team", commit_team)
    gi.add_batch(V, levels=lv, mode="exact")
    ok, why = graphs_equal(o.export(), gi.export_graph())
    assert ok, why
    for v in (7, 100, 555, 3, 148, 484):
        o.delete(v)
        gi.delete_node("node%d" % v)
        ok, why = graphs_equal(o.export(), gi.export_graph())
        assert ok, "after deleting %d: %s" % (v, why)
    for i, v in enumerate((7, 100, 555)):                     # and back in: single adds on the thinned graph
        o.add(V[v], int(lv[v]))
        gi.add_node("again%d" % i, V[v], level=int(lv[v]))
    ok, why = graphs_equal(o.export(), gi.export_graph())
    assert ok, why
    gi.close()


def test_one_wave_and_four_wave_commits_give_the_same_verdicts_on_speculative_records(eng, oracle_mod):
    """Deletes on random small indexes (dims 96 / 128 / 256 / 768): after every delete the graph is the oracle's, and the
    one-wave and the four-wave commit kernel used the SAME speculative re-selections and recomputed the same others
    (the control block's n_spec / n_fallback).  Before hnsw_wave_sync.hpp they did not: different builds of the same
    validation flagged different records on dim 768, one of them too few."""
    import ctypes as C
    from redis_hnsw_amd import _capi
    lib = _capi.load()
    lib.hnsw_debug_occ_ctl.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
    rng = np.random.default_rng(5)
    for c, (dim, m, ef) in enumerate([(768, 16, 400), (128, 32, 100), (256, 8, 200), (96, 16, 100), (768, 24, 40)]):
        n = int(rng.integers(600, 1200))
        V = make_data(n, dim, seed=200 + c)
        lv = oracle_mod.draw_levels(n, m, 21 + c)
        victims = [int(x) for x in rng.choice(n, size=6, replace=False)]
        verdicts = {}
        for team in (1, 0):
            o = oracle_mod.OracleIndex(dim, m, ef)
            o.add_batch(V, lv)
            gi = eng.Index("v", dim, m, ef)
            gi.set_tuning("commit_team", team)
            gi.add_batch(V, levels=lv, mode="exact")
            out = (C.c_uint64 * 18)()
            vs = []
            for v in victims:
                o.delete(v)
                gi.delete_node("node%d" % v)
                assert lib.hnsw_debug_occ_ctl(gi._h, out) == 0
                vs.append((int(out[16]), int(out[17])))
                ok, why = graphs_equal(o.export(), gi.export_graph())
                assert ok, "%s team=%d after deleting %d: %s" % ((n, dim, m, ef), team, v, why)
            verdicts[team] = vs
            gi.close()
        assert verdicts[1] == verdicts[0], ((n, dim, m, ef), verdicts)


@pytest.mark.parametrize("n,dim,m,ef", [(1200, 32, 100, 200), (600, 128, 65, 100), (700, 64, 128, 200)])
def test_m_above_64_builds_searches_and_deletes_like_the_oracle(eng, oracle_mod, n, dim, m, ef):
    """The reference does not bound M (core.rs:322-347, src/lib.rs:39-56).  Above 64 a node's selected links no longer fit
    one per lane: the serial insert / delete kernels walk them 64 at a time, select_neighbors keeps up to 2M = 256 keys.
    Batch build (both modes: the fast one is the exact one here), single adds with the update list, searches, deletes:
    the oracle's graph row for row, its answers bit for bit."""
    V = make_data(n + 20, dim, seed=91)
    lv = oracle_mod.draw_levels(n + 20, m, 8)
    o = oracle_mod.OracleIndex(dim, m, ef)
    o.add_batch(V[:n], lv[:n])
    gi = eng.Index("widem", dim, m, ef)
    half = n // 2
    gi.add_batch(V[:half], levels=lv[:half], mode="exact")
    gi.add_batch(V[half:n], levels=lv[half:n], mode="fast")
    ok, why = graphs_equal(o.export(), gi.export_graph())
    assert ok, why
    assert max(len(o.neighbors(i, 0)) for i in range(n)) > 64
    for i in range(n, n + 20):
        oid, ot = o.add(V[i], int(lv[i]), want_touched=True)
        got = []
        gi.add_node("node%d" % i, V[i], lambda s_, nid: got.append(nid), level=int(lv[i]))
        assert sorted(got) == sorted(ot.tolist()), "touched set of insert %d" % i
    ok, why = graphs_equal(o.export(), gi.export_graph())
    assert ok, why
    Q = make_data(48, dim, seed=92)
    k = 10
    ids, sims, n_out = gi.search_batch(Q, k)
    oids, osims, on, _ = o.search_batch(Q, k)
    assert np.array_equal(n_out, on)
    for q in range(len(Q)):
        nv = int(on[q])
        assert np.array_equal(ids[q, :nv], oids[q, :nv]) and np.array_equal(_bits(sims[q, :nv]), _bits(osims[q, :nv]))
    deg = [len(o.neighbors(i, 0)) for i in range(n)]
    for v in [int(i) for i in np.argsort(deg)[-3:]] + [3, 77]:
        o.delete(v)
        gi.delete_node("node%d" % v)
        ok, why = graphs_equal(o.export(), gi.export_graph())
        assert ok, "after deleting %d: %s" % (v, why)
    for q in Q[:8]:
        assert [r.id for r in gi.search_knn(q, k)] == o.search(q, k)[0].tolist()
    gi.close()
