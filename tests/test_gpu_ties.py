"""The tie census (hnsw_get_tie_counters): the engine counts every decision that compared equal
distances of two different nodes -- the only places where its (distance, id) order and the reference's sim-only order
on std's BinaryHeap (core.rs:292-300, :635, :657, :733) can part.  The oracle counts the reference's own decisions one
neighbour at a time (hnsw_oracle_tie_census, hnsw_oracle_last_add_ties); the engine merges a whole adjacency row at once
and counts a SUPERSET: whenever the oracle sees a tie in a query / an insert, so must the engine, and on tie-free data
both see none."""
import numpy as np
import pytest

from tests.util import graphs_equal, make_data

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    from redis_hnsw_amd import index as idxmod
    return idxmod


def _binary(n, dim, seed):
    """vectors of {0, 1}^dim: squared distances are small integers, equal everywhere"""
    return np.random.default_rng(seed).integers(0, 2, size=(n, dim)).astype(np.float32)


@pytest.mark.parametrize("ef", [40, 200])
def test_search_census_sees_every_tie_the_oracle_sees(eng, oracle_mod, ef):
    n, dim, m, k = 3000, 128, 16, 10
    V = np.unique(_binary(n, dim, 5), axis=0)
    n = len(V)
    lv = oracle_mod.draw_levels(n, m, 9)
    o = oracle_mod.OracleIndex(dim, m, ef)
    o.add_batch(V, lv)
    g = o.export()
    gi = eng.Index("ties", dim, m, ef)
    gi.import_graph(g)
    gi.set_tuning("tie_census", 1)
    Q = _binary(96, dim, 6)
    seen = extra_flagged = 0
    for q in Q:
        cen = o.tie_census(q[None, :], k)
        gi.reset_counters()
        ids, sims, n_out = gi.search_batch(q[None, :], k)
        oids, osims = o.search(q, k)
        assert np.array_equal(ids[0, :len(oids)], oids)                    # same answers as ever (the census kernel is the kernel)
        t = gi.tie_counters()
        want = cen["queries_with_any_tie"]
        if want:
            seen += 1
            assert t["search_events"] >= 1 and t["queries_with_tie"] == 1, (cen, t)
        if not want and t["search_events"]:
            extra_flagged += 1                                              # the superset's price: counted here, no decision there
    assert seen >= 48                                                       # the data does tie
    # a batch: the per-query flags add up
    gi.reset_counters()
    gi.search_batch(Q, k)
    cen = o.tie_census(Q, k)
    t = gi.tie_counters()
    assert t["queries_with_tie"] >= cen["queries_with_any_tie"] and t["queries_with_tie"] == seen + extra_flagged
    gi.close()
    o.close()


def test_no_ties_on_uniform_data_and_no_count_without_the_tuning(eng, oracle_mod):
    n, dim, m, ef, k = 6000, 128, 16, 200, 10
    V = make_data(n, dim, seed=51)
    lv = oracle_mod.draw_levels(n, m, 3)
    o = oracle_mod.OracleIndex(dim, m, ef)
    cen_build = o.add_batch_census(V, lv)
    gi = eng.Index("noties", dim, m, ef)
    gi.add_batch(V, levels=lv, mode="exact")                               # the windowed reference-order build
    ok, why = graphs_equal(o.export(), gi.export_graph())
    assert ok, why
    t = gi.tie_counters()
    if cen_build[0] == 0:                                                   # (f32 similarities of 6 k uniform vectors: no decision tie)
        assert t["insert_events"] == 0 and t["plans_with_tie"] == 0, t
    else:
        assert t["insert_events"] >= 1
    Q = make_data(512, dim, seed=52)
    gi.reset_counters()
    gi.search_batch(Q, k)
    assert gi.tie_counters()["search_events"] == 0                          # not counted: the tuning is off
    gi.set_tuning("tie_census", 1)
    ids, sims, n_out = gi.search_batch(Q, k)
    oids, osims, on, _ = o.search_batch(Q, k)
    assert np.array_equal(ids, oids) and np.array_equal(sims.view(np.uint32), osims.view(np.uint32))
    cen = o.tie_census(Q, k)
    t = gi.tie_counters()
    # a superset, but a tight one: a whole row is merged at once, so an arrival next to an equal key can be counted where
    # the reference, walking the row id by id, never compared the two
    # (e.g. an equal pair across W's end is counted when it forms; the reference compares the two only if the outer one
    # is popped while the inner one is still W's last).  A few per cent of the queries at most on uniform data.
    assert cen["queries_with_any_tie"] <= t["queries_with_tie"] <= cen["queries_with_any_tie"] + len(Q) // 32, (cen, t)
    gi.close()
    o.close()


def test_insert_census_sees_every_tie_the_oracle_sees(eng, oracle_mod):
    """single hnsw_add calls (the form the Redis command issues) on tie-heavy data: whenever the oracle's insert met a
    decision tie -- stop test, accept test or a select_neighbors cut, in the plan or in the shrink loop -- the engine's
    counters moved too; the graphs stay identical (both use the (distance, id) order)."""
    n0, extra, dim, m, ef = 1200, 160, 128, 8, 48
    V = np.unique(_binary(n0 + extra + 64, dim, 15), axis=0)[:n0 + extra]
    lv = oracle_mod.draw_levels(len(V), m, 4)
    o = oracle_mod.OracleIndex(dim, m, ef)
    o.add_batch(V[:n0], lv[:n0])
    gi = eng.Index("insties", dim, m, ef)
    gi.add_batch(V[:n0], levels=lv[:n0], mode="exact")
    ok, why = graphs_equal(o.export(), gi.export_graph())
    assert ok, why
    assert gi.tie_counters()["insert_events"] > 0                           # the windowed build met ties as well
    import ctypes as C
    lib = oracle_mod.lib()
    seen = 0
    for i in range(n0, n0 + extra):
        o.add(V[i], int(lv[i]))
        one = np.zeros(4, dtype=np.uint64)
        lib.hnsw_oracle_last_add_ties(o._h, one.ctypes.data_as(C.POINTER(C.c_uint64)))
        gi.reset_counters()
        gi.add_node("n%d" % i, V[i], level=int(lv[i]))
        t = gi.tie_counters()
        if int(one.sum()):
            seen += 1
            assert t["insert_events"] >= 1, (i, one, t)
    assert seen >= extra // 4
    ok, why = graphs_equal(o.export(), gi.export_graph())
    assert ok, why
    gi.close()
    o.close()


def test_a_search_without_a_census_kernel_reads_as_unknown(eng, oracle_mod):
    """tie_census = 1 on a shape the census form of the kernel does not serve (dim 64: the general kernel): the searches are
    answered as ever, and the search half of the counters says UNKNOWN rather than zero; the insert half still counts."""
    n, dim, m, ef = 800, 64, 8, 40
    V = make_data(n, dim, seed=61)
    lv = oracle_mod.draw_levels(n, m, 2)
    o = oracle_mod.OracleIndex(dim, m, ef)
    o.add_batch(V, lv)
    gi = eng.Index("unk", dim, m, ef)
    gi.add_batch(V, levels=lv, mode="exact")
    gi.set_tuning("tie_census", 1)
    gi.reset_counters()
    Q = make_data(32, dim, seed=62)
    ids, sims, n_out = gi.search_batch(Q, 5)
    oids, osims, on, _ = o.search_batch(Q, 5)
    assert np.array_equal(ids, oids) and np.array_equal(sims.view(np.uint32), osims.view(np.uint32))
    t = gi.tie_counters()
    assert t["search_events"] is None and t["queries_with_tie"] is None and t["insert_events"] == 0
    gi.reset_counters()                                                     # a new period: nothing ran, nothing is unknown
    assert gi.tie_counters()["search_events"] == 0
    gi.close()
    o.close()


@pytest.mark.parametrize("kind,n,dim,m,ef", [("binary", 1200, 128, 8, 48), ("binary", 700, 32, 5, 16), ("lattice", 900, 20, 6, 24),
                                             ("uniform", 800, 128, 16, 64)])
def test_the_std_heap_kernel_builds_what_the_std_heap_oracle_builds(eng, oracle_mod, kind, n, dim, m, ef):
    """tie_mode = 2: EVERY insert runs on the one-wavefront kernel that restates the reference's insert() on std's BinaryHeap
    (hnsw_std_heap.hpp) -- against hnsw_oracle_add_std_heap, which is pinned to the transcription's "rust"-mode golden.  Tie-
    heavy data (binary / lattice vectors: equal similarities at every turn, both metric orders), so the heap's sift order
    decides nearly every insert; rows must match in stored order."""
    rng = np.random.default_rng(77)
    if kind == "binary":
        V = np.unique(rng.integers(0, 2, size=(n + 200, dim)).astype(np.float32), axis=0)[:n]
        rng.shuffle(V)
    elif kind == "lattice":
        V = rng.integers(0, 3, size=(n, dim)).astype(np.float32)            # duplicates included
    else:
        V = make_data(n, dim, seed=78)
    n = len(V)
    lv = oracle_mod.draw_levels(n, m, 8)
    o = oracle_mod.OracleIndex(dim, m, ef)
    o.add_batch_std_heap(V, lv)
    gi = eng.Index("std", dim, m, ef)
    gi.set_tuning("tie_mode", 2)
    gi.add_batch(V, levels=lv, mode="exact")
    ok, why = graphs_equal(o.export(), gi.export_graph())
    assert ok, why
    gi.close()
    o.close()


def test_tie_mode_builds_the_reference_binary_s_graph(eng, oracle_mod):
    """tie_mode = 1 on SURVEY's model shape (20 k x 128 uniform f32, M = 16, ef = 200): the windowed reference-order build,
    with every insert the census flags -- in its plan or in the dry run of its commit -- handed UNTOUCHED to the one-wavefront
    kernel that restates insert() on std's BinaryHeap.  The result must be the graph the transcription built in its
    "rust" tie mode (tests/golden/transcribed_20k_dim128.npz: 16 accept-test and 2 select-cut ties decided by the heap's
    sift order, after which the total order's graph differs in 3.7 % of the layer-0 rows): every row of every layer, in
    stored order.  Without tie_mode the engine builds the total order's graph (every other test)."""
    import ctypes as C
    from tests.golden_util import load_transcribed
    c = load_transcribed("transcribed_20k_dim128")
    assert c["stats"]["ties"] == "rust"
    gi = eng.Index("tiemode", c["dim"], c["m"], c["ef"])
    gi.set_tuning("tie_mode", 1)
    gi.add_batch(c["V"], levels=c["levels"], mode="exact")
    ok, why = graphs_equal(c["graph"], gi.export_graph())
    assert ok, why
    lib = eng._capi.load()
    lib.hnsw_debug_tie_redone.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
    red = C.c_uint64(0)
    assert lib.hnsw_debug_tie_redone(gi._h, C.byref(red)) == 0
    assert 18 <= red.value < c["n"] // 10, red.value                       # the flagged few, not the build
    gi.close()


@pytest.mark.parametrize("kind,n,dim,m,ef,mode", [("binary", 3000, 128, 16, 40, 1), ("binary", 3000, 128, 16, 200, 1), ("binary", 3000, 128, 16, 40, 2),
                                                  ("lattice", 1500, 64, 8, 32, 1), ("uniform", 6000, 128, 16, 200, 1)])
def test_tie_mode_answers_queries_as_the_reference_binary_does(eng, oracle_mod, kind, n, dim, m, ef, mode):
    """tie_mode on HNSW.SEARCH: the queries the census kernel flags (mode 1), or all of them (mode 2, and mode 1 on a shape
    without a census kernel -- dim 64 here), are answered again by the one-wavefront kernel that restates search_knn / search_level
    on std's BinaryHeap: ids in the reference's pop order and similarities bit for bit against hnsw_oracle_search_std_heap,
    on a graph built in std order too.  On binary / lattice data nearly every query ties somewhere."""
    rng = np.random.default_rng(91)
    if kind == "binary":
        V = np.unique(rng.integers(0, 2, size=(n + 200, dim)).astype(np.float32), axis=0)[:n]
        rng.shuffle(V)
        Q = rng.integers(0, 2, size=(160, dim)).astype(np.float32)
    elif kind == "lattice":
        V = rng.integers(0, 3, size=(n, dim)).astype(np.float32)
        Q = rng.integers(0, 3, size=(160, dim)).astype(np.float32)
    else:
        V = make_data(n, dim, seed=92)
        Q = make_data(512, dim, seed=93)
    n, k = len(V), 10
    lv = oracle_mod.draw_levels(n, m, 8)
    o = oracle_mod.OracleIndex(dim, m, ef)
    o.add_batch_std_heap(V, lv)
    gi = eng.Index("stdq", dim, m, ef)
    gi.import_graph(o.export())
    gi.set_tuning("tie_mode", mode)
    gi.reset_counters()
    ids, sims, n_out = gi.search_batch(Q, k)
    differs_from_total_order = 0
    for i, q in enumerate(Q):
        oids, osims = o.search_std_heap(q, k)
        assert n_out[i] == len(oids)
        assert np.array_equal(ids[i, :len(oids)], oids), (i, ids[i], oids)
        assert np.array_equal(sims[i, :len(oids)].view(np.uint32), np.asarray(osims, dtype=np.float32).view(np.uint32))
        tids, _ = o.search(q, k)
        differs_from_total_order += not np.array_equal(tids, oids)
    if kind != "uniform":
        assert differs_from_total_order > 0                                 # the mode had something to decide
    t = gi.tie_counters()
    if mode == 1 and dim == 128:
        assert t["queries_with_tie"] is not None
        if kind == "uniform":
            assert t["queries_with_tie"] <= len(Q) // 16                    # the flagged few, not the batch
    # a second batch through the asynchronous device entry and the pipelined host entry gives the same answers
    ids2, sims2, n2 = gi.search_batch(Q[:64], k)
    assert np.array_equal(ids2, ids[:64]) and np.array_equal(n2, n_out[:64])
    gi.close()
    o.close()


@pytest.mark.parametrize("kind,n", [("uniform", 2500), ("quantised", 900)])
def test_single_adds_in_tie_mode_link_what_the_reference_binary_links(eng, oracle_mod, kind, n):
    """tie_mode = 1, one hnsw_add per call (the HNSW.NODE.ADD command's shape): each insert runs as a one-node window whose
    plan and whose commit -- a dry run of the group commit kernel, which writes nothing before it knows -- are gated by the
    tie census; only a flagged insert is redone on the std-order kernel.  The graph must be the std-order oracle's row for
    row; on uniform data nearly every insert takes the gated window, on quantised data (distances on a coarse grid) most
    are redone.  The touched list (update_fn, core.rs:535-537, 787-816) covers every row the insert changed."""
    import ctypes as C
    dim, m, ef = 128, 16, 64
    V = make_data(n, dim, seed=33)
    if kind == "quantised":
        V = np.unique(np.round(V * 2.0) / 2.0, axis=0)
        np.random.default_rng(2).shuffle(V)
    n = len(V)
    lv = oracle_mod.draw_levels(n, m, 6)
    o = oracle_mod.OracleIndex(dim, m, ef)
    o.add_batch_std_heap(V, lv)
    want = o.export()
    gi = eng.Index("single-ties", dim, m, ef)
    gi.set_tuning("tie_mode", 1)
    before = None
    for i in range(n):
        sample = i >= 40 and i % 97 == 0
        if sample:
            before = gi.export_graph()
        got = []
        gi.add_node("n%d" % i, V[i], (lambda s, nid: got.append(nid)) if sample else None, level=int(lv[i]))
        if sample:
            after = gi.export_graph()
            changed = set()
            for lc in range(len(after["row_ptr"])):                        # rows are CSR over all ids, per layer
                ra, ca = after["row_ptr"][lc].astype(np.int64), after["col"][lc]
                if lc < len(before["row_ptr"]):
                    rb, cb = before["row_ptr"][lc].astype(np.int64), before["col"][lc]
                else:
                    rb, cb = np.zeros(i + 1, dtype=np.int64), np.zeros(0, dtype=np.uint32)
                for node in range(i):
                    if not np.array_equal(ca[ra[node]:ra[node + 1]], cb[rb[node]:rb[node + 1]]):
                        changed.add(node)
            assert changed <= set(got), (i, sorted(changed - set(got))[:8])
    ok, why = graphs_equal(want, gi.export_graph())
    assert ok, why
    lib = eng._capi.load()
    lib.hnsw_debug_tie_redone.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
    red = C.c_uint64(0)
    assert lib.hnsw_debug_tie_redone(gi._h, C.byref(red)) == 0
    if kind == "uniform":
        assert red.value < n // 8, red.value                                # the flagged few: the rest took the gated window
    else:
        assert red.value > n // 8, red.value
    gi.close()
    o.close()


def test_tie_mode_deletes_reproduce_the_transcription(eng):
    """HNSW.NODE.DEL under tie_mode: delete_node_from_neighbors (core.rs:824-863) on std's BinaryHeap restated
    (k_delete_std_heap).  The transcription's "rust"-mode run on lattice data -- 600 adds, 120 deletes (the enterpoint among
    them), 80 more adds -- replayed through the C ABI with tie_mode = 1 (dim 8: no tie-counting plan kernel, so every insert
    runs on the std-order kernel too) must leave the transcription's graph, row for row in stored order."""
    from tests.golden_util import load_tiecase_del
    c = load_tiecase_del()
    gi = eng.Index("del-ties", c["dim"], c["m"], c["ef"])
    gi.set_tuning("tie_mode", 1)
    V, lv, n0 = c["V"], c["levels"], c["n0"]
    gi.add_batch(V[:n0], levels=lv[:n0], mode="exact")
    for v in c["victims"]:
        gi.delete_node(gi._names[int(v)])
    for i in range(n0, n0 + c["n1"]):
        gi.add_node("late%d" % i, V[i], level=int(lv[i]))
    ok, why = graphs_equal(c["graph"], gi.export_graph())
    assert ok, why
    gi.close()


def test_tie_mode_adds_and_deletes_interleaved_equal_the_std_heap_oracle(eng, oracle_mod):
    """dim 128, quantised data (distances on a coarse grid: ties at every turn), tie_mode = 1: a windowed build, then single
    deletes and single adds interleaved; graph and touched sets against the std-heap oracle after every phase."""
    dim, m, ef, n0, extra = 128, 12, 48, 700, 120
    V = np.unique(np.round(make_data(n0 + extra + 64, dim, seed=44) * 2.0) / 2.0, axis=0)
    np.random.default_rng(3).shuffle(V)
    V = V[:n0 + extra]
    lv = oracle_mod.draw_levels(len(V), m, 12)
    o = oracle_mod.OracleIndex(dim, m, ef)
    o.add_batch_std_heap(V[:n0], lv[:n0])
    gi = eng.Index("mix-ties", dim, m, ef)
    gi.set_tuning("tie_mode", 1)
    gi.add_batch(V[:n0], levels=lv[:n0], mode="exact")
    ok, why = graphs_equal(o.export(), gi.export_graph())
    assert ok, why
    rng = np.random.default_rng(8)
    alive = list(range(n0))
    for j in range(extra):
        v = alive.pop(int(rng.integers(0, len(alive))))
        ot = o.delete_std_heap(v, want_touched=True)
        got = []
        gi.delete_node(gi._names[v], update_fn=lambda s, nid: got.append(nid))
        assert sorted(got) == sorted(ot.tolist()), "touched set of delete %d" % v
        i = n0 + j
        o.add_batch_std_heap(V[i:i + 1], lv[i:i + 1])
        gi.add_node("x%d" % i, V[i], level=int(lv[i]))
        alive.append(i)
        if j % 40 == 39:
            ok, why = graphs_equal(o.export(), gi.export_graph())
            assert ok, "after %d rounds: %s" % (j + 1, why)
    ok, why = graphs_equal(o.export(), gi.export_graph())
    assert ok, why
    gi.close()
    o.close()


def test_tie_mode_refuses_a_compressed_index(eng, oracle_mod):
    """bf16 / fp8 storage modes keep no f32 vectors: the std-order kernels have nothing to compute the reference's similarities
    from, and a search under tie_mode says so instead of answering from the wrong bytes"""
    n, dim, m, ef = 600, 128, 8, 40
    V = make_data(n, dim, seed=71)
    gi = eng.Index("fmt-ties", dim, m, ef)
    gi.add_batch(V, levels=oracle_mod.draw_levels(n, m, 3), mode="exact")
    gi.set_tuning("compress_bf16", 1)
    gi.search_batch(V[:8], 5)                                               # the compressed walk, as ever
    gi.set_tuning("tie_mode", 1)
    with pytest.raises(Exception) as e:
        gi.search_batch(V[:8], 5)
    assert "tie_mode" in str(e.value)
    gi.set_tuning("tie_mode", 0)
    gi.search_batch(V[:8], 5)
    gi.close()
