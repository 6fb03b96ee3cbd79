"""The engine's own search pipeline (hnsw_search_batch / hnsw_search_batch_device split large batches into
chunks on engine-owned streams) and the round-2 advisor findings around the specialised kernel: results must be
the oracle's whatever the chunking, the launch shape or the order of calls."""
import numpy as np
import pytest

from tests.util import build_oracle, graphs_equal, make_data

pytestmark = pytest.mark.gpu


def _bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


@pytest.fixture(scope="module")
def eng():
    from redis_hnsw_amd import index as idxmod
    return idxmod


@pytest.fixture(scope="module")
def small(oracle_mod):
    n, dim, m, ef = 6000, 128, 16, 200
    V = make_data(n, dim, seed=1)
    o, lv = build_oracle(oracle_mod, V, m, ef)
    return V, o, lv, (n, dim, m, ef)


@pytest.mark.parametrize("chunk,B", [(256, 3000), (1024, 2500), (100, 777), (64, 65)])
def test_pipelined_host_search_equals_the_oracle(eng, small, chunk, B):
    V, o, lv, (n, dim, m, ef) = small
    g = o.export()
    g["vectors"] = V
    gi = eng.Index("pipe", dim, m, ef)
    gi.import_graph(g)
    gi.set_tuning("pipe_chunk", chunk)
    Q = make_data(B, dim, seed=5)
    k = 10
    ids, sims, n_out = gi.search_batch(Q, k)                       # ceil(B / chunk) chunks over 3 lanes
    oids, osims, on, _ = o.search_batch(Q, k, threads=8)
    assert np.array_equal(n_out, on) and np.array_equal(ids, oids) and np.array_equal(_bits(sims), _bits(osims))
    ids2, sims2, _ = gi.search_batch(Q, k)                         # staging buffers are reused: same again
    assert np.array_equal(ids, ids2) and np.array_equal(_bits(sims), _bits(sims2))
    # k larger than before: the staging grows
    ids3, sims3, n3 = gi.search_batch(Q[:300], 40)
    o3 = o.search_batch(Q[:300], 40, threads=8)
    assert np.array_equal(ids3, o3[0]) and np.array_equal(_bits(sims3), _bits(o3[1])) and np.array_equal(n3, o3[2])
    gi.close()


def test_pipelined_search_refuses_a_non_finite_component_in_a_late_chunk(eng, small):
    V, o, lv, (n, dim, m, ef) = small
    g = o.export()
    g["vectors"] = V
    gi = eng.Index("pipe-nan", dim, m, ef)
    gi.import_graph(g)
    gi.set_tuning("pipe_chunk", 128)
    Q = make_data(1000, dim, seed=6)
    Q[900, 17] = np.inf
    with pytest.raises(eng.HNSWError):
        gi.search_batch(Q, 5)
    Q[900, 17] = 0.5
    ids, _, n_out = gi.search_batch(Q, 5)                          # and the handle keeps working
    assert np.all(n_out == 5) and np.array_equal(ids, o.search_batch(Q, 5, threads=8)[0])
    gi.close()


def test_pipelined_device_call_joins_back_into_the_callers_stream(eng, small):
    import torch
    V, o, lv, (n, dim, m, ef) = small
    g = o.export()
    g["vectors"] = V
    gi = eng.Index("pipe-dev", dim, m, ef)
    gi.import_graph(g)
    gi.set_tuning("pipe_device", 1)                                # off by default (one launch per call is faster)
    gi.set_tuning("pipe_chunk", 200)
    gi.set_tuning("pipe_min_batch", 400)
    dev = torch.device("cuda", 0)
    B, k = 1500, 10
    Q = make_data(B, dim, seed=7)
    st = torch.cuda.Stream()
    ids_t = torch.zeros((B, k), dtype=torch.int32, device=dev)
    sims_t = torch.zeros((B, k), dtype=torch.float32, device=dev)
    n_t = torch.zeros((B,), dtype=torch.int32, device=dev)
    with torch.cuda.stream(st):
        dQ = torch.from_numpy(Q).to(dev, non_blocking=True)        # the input is produced ON the caller's stream
        for _ in range(3):                                         # back-to-back calls on one stream: fork after join
            gi.search_batch_device(dQ.data_ptr(), B, k, ids_t.data_ptr(), sims_t.data_ptr(), n_t.data_ptr(), st.cuda_stream)
        host_ids = torch.empty_like(ids_t, device="cpu").pin_memory()
        host_ids.copy_(ids_t, non_blocking=True)                   # consumer ordered on the caller's stream only
    st.synchronize()
    oids, osims, on, _ = o.search_batch(Q, k, threads=8)
    assert np.array_equal(host_ids.numpy().view(np.uint32), oids)
    assert np.array_equal(_bits(sims_t.cpu().numpy()), _bits(osims))
    assert np.array_equal(n_t.cpu().numpy().view(np.uint32), on)
    # an exact insert right after waits for the lanes (searches in flight read the rows it rewrites)
    with torch.cuda.stream(st):
        gi.search_batch_device(dQ.data_ptr(), B, k, ids_t.data_ptr(), sims_t.data_ptr(), n_t.data_ptr(), st.cuda_stream)
    gi.add_node("late", make_data(1, dim, seed=8)[0], level=0)
    st.synchronize()
    assert np.array_equal(ids_t.cpu().numpy().view(np.uint32), oids)
    gi.close()


def test_pipeline_lanes_overlap_and_report_it(eng, small):
    V, o, lv, (n, dim, m, ef) = small
    gi = eng.Index("pipe-info", dim, m, ef)
    info = gi.pipeline_info()                                      # creates the lanes and measures them
    assert info["lanes"] == 3 and info["chunk"] == 1024
    assert info["overlap"] == 1, "the engine's streams share a hardware queue: %r" % (info,)
    assert 0.5 < info["probe_ratio"] < 1.6
    gi.close()


def test_callers_own_streams_need_no_tuning(eng, small):
    """three caller streams, nothing promised: each launch sizes its LDS share from the launches it sees in
    flight; results are the oracle's whichever table size a launch got"""
    import torch
    V, o, lv, (n, dim, m, ef) = small
    g = o.export()
    g["vectors"] = V
    gi = eng.Index("auto", dim, m, ef)
    gi.import_graph(g)
    dev = torch.device("cuda", 0)
    B, k, S = 1024, 10, 3
    Q = make_data(6 * B, dim, seed=9)
    dQ = torch.from_numpy(Q).to(dev)
    streams = [torch.cuda.Stream() for _ in range(S)]
    outs = [(torch.empty((B, k), dtype=torch.int32, device=dev), torch.empty((B, k), dtype=torch.float32, device=dev),
             torch.empty((B,), dtype=torch.int32, device=dev)) for _ in range(6)]
    torch.cuda.synchronize()
    for i in range(6):
        gi.search_batch_device(dQ[i * B:(i + 1) * B].data_ptr(), B, k, outs[i][0].data_ptr(), outs[i][1].data_ptr(),
                               outs[i][2].data_ptr(), streams[i % S].cuda_stream)
    torch.cuda.synchronize()
    oids, osims, on, _ = o.search_batch(Q, k, threads=8)
    for i in range(6):
        assert np.array_equal(outs[i][0].cpu().numpy().view(np.uint32), oids[i * B:(i + 1) * B])
        assert np.array_equal(_bits(outs[i][1].cpu().numpy()), _bits(osims[i * B:(i + 1) * B]))
    gi.close()


# ---- advisor findings, round 2 ---------------------------------------------------------------------
def test_delete_keeps_an_m16_index_on_the_specialised_kernel(eng, oracle_mod):
    """HNSW.NODE.DEL asks for row slack by the deleted node's own degree, not by the index's maximum, so deletes
    do not widen the rows of an M = 16 index; and rows of up to 127 ids (a hub of a small index, a restride) are
    still served by the dim-128 kernel (its two-row-word form)."""
    n, dim, m, ef = 3000, 128, 16, 200
    V = make_data(n, dim, seed=3)
    o, lv = build_oracle(oracle_mod, V, m, ef)
    gi = eng.Index("del16", dim, m, ef)
    gi.add_batch(V, levels=lv, mode="exact")
    Q = make_data(64, dim, seed=4)
    ids, sims, n_out = gi.search_batch(Q, 10)
    inf0 = gi.info()
    assert gi.last_search_was_lean(), (gi.lean_blocker(), inf0.stride0, inf0.stride_upper, inf0.max_degree0, inf0.max_degree_upper)
    oids, osims, on, _ = o.search_batch(Q, 10)
    assert np.array_equal(ids, oids) and np.array_equal(_bits(sims), _bits(osims)) and np.array_equal(n_out, on)
    assert inf0.max_degree0 >= 2 * m
    for i in (5, 700, 1999, 2500):
        gi.delete_node("node%d" % i)
        o.delete(i)
    inf1 = gi.info()
    assert (inf1.stride0, inf1.stride_upper) == (inf0.stride0, inf0.stride_upper)
    ids, sims, n_out = gi.search_batch(Q, 10)
    assert gi.last_search_was_lean()
    oids, osims, on, _ = o.search_batch(Q, 10)
    assert np.array_equal(ids, oids) and np.array_equal(_bits(sims), _bits(osims)) and np.array_equal(n_out, on)
    gi.close(); o.close()


@pytest.mark.parametrize("m,ef,k", [(16, 400, 10), (32, 512, 100), (5, 257, 300)])
def test_ef_up_to_512_stays_on_the_specialised_kernel(eng, oracle_mod, m, ef, k):
    """ef_construction 257..512 at dim 128 (W in eight register slices, R = 8): same answers and counters as the oracle
    through the specialised kernel, narrow and wide rows, k above and below ef"""
    n, dim = 2500, 128
    V = make_data(n, dim, seed=23)
    o, lv = build_oracle(oracle_mod, V, m, ef)
    g = o.export()
    g["vectors"] = V
    gi = eng.Index("r8", dim, m, ef)
    gi.import_graph(g)
    Q = make_data(200, dim, seed=24)
    gi.set_tuning("waves_per_cu", 4)                               # the 32 KB table: nothing is forgotten at this size
    gi.reset_counters()
    ids, sims, n_out = gi.search_batch(Q, k)
    assert gi.last_search_was_lean(), gi.lean_blocker()
    oids, osims, on, oct = o.search_batch(Q, k, threads=8)
    assert np.array_equal(n_out, on)
    for b in range(Q.shape[0]):
        c = int(on[b])
        assert np.array_equal(ids[b, :c], oids[b, :c]) and np.array_equal(_bits(sims[b, :c]), _bits(osims[b, :c]))
    sc, _ = gi.counters()
    assert (sc.n_dist, sc.n_ids, sc.n_expand) == (oct.n_dist, oct.n_ids, oct.n_expand)
    gi.set_tuning("waves_per_cu", 8)                               # bounded table: re-met nodes may be evaluated twice
    gi.reset_counters()
    ids2, sims2, n2 = gi.search_batch(Q, k)
    assert gi.last_search_was_lean()
    assert np.array_equal(n2, on) and np.array_equal(ids2, ids) and np.array_equal(_bits(sims2), _bits(sims))
    sc, _ = gi.counters()
    assert (sc.n_ids, sc.n_expand) == (oct.n_ids, oct.n_expand) and sc.n_dist >= oct.n_dist
    one = gi.search_knn(Q[0], k)
    assert [x.id for x in one] == oids[0, : int(on[0])].tolist()
    gi.close(); o.close()


@pytest.mark.parametrize("m,widen", [(16, 0), (16, 32), (32, 0), (24, 0)])
def test_wide_rows_stay_on_the_specialised_kernel(eng, oracle_mod, m, widen):
    """rows of 64..127 ids (M > 16, or an M = 16 index after a restride): same answers and counters as the oracle,
    through the two-row-word form of the dim-128 kernel, with the bounded table and with the exact one"""
    n, dim, ef, k = 2500, 128, 200, 10
    V = make_data(n, dim, seed=13)
    o, lv = build_oracle(oracle_mod, V, m, ef)
    g = o.export()
    g["vectors"] = V
    gi = eng.Index("wide", dim, m, ef)
    gi.import_graph(g)
    if widen:
        gi.set_tuning("force_restride", widen)
    inf = gi.info()
    assert (inf.stride0 > 64) == (m > 16 or widen > 0)
    Q = make_data(300, dim, seed=14)
    gi.reset_counters()
    ids, sims, n_out = gi.search_batch(Q, k)
    assert gi.last_search_was_lean(), gi.lean_blocker()
    oids, osims, on, oct = o.search_batch(Q, k, threads=8)
    assert np.array_equal(ids, oids) and np.array_equal(_bits(sims), _bits(osims)) and np.array_equal(n_out, on)
    sc, _ = gi.counters()
    assert (sc.n_ids, sc.n_expand) == (oct.n_ids, oct.n_expand) and sc.n_dist >= oct.n_dist
    gi.set_tuning("waves_per_cu", 4)                               # the 32 KB table: nothing is forgotten at this size
    gi.reset_counters()
    ids2, sims2, _ = gi.search_batch(Q, k)
    sc, _ = gi.counters()
    assert np.array_equal(ids2, oids) and (sc.n_dist, sc.n_ids, sc.n_expand) == (oct.n_dist, oct.n_ids, oct.n_expand)
    gi.close(); o.close()


def _stored(gi, n):
    return np.stack([gi._vector(i) for i in range(n)])


@pytest.mark.parametrize("fmt,dim,m,ef,k", [("bf16", 128, 16, 200, 10),    # the specialised kernel's bf16 form
                                            ("bf16", 128, 16, 400, 10),    # ef 257..512: the R = 8 form of the dim-128 kernel
                                            ("bf16", 64, 8, 100, 10), ("bf16", 768, 32, 400, 100), ("bf16", 128, 32, 64, 10),
                                            ("fp8", 128, 16, 200, 10), ("fp8", 256, 5, 16, 5), ("fp8", 768, 32, 400, 50)])
def test_compressed_storage_is_the_reference_on_the_stored_values(eng, oracle_mod, fmt, dim, m, ef, k):
    """bf16 / fp8 serving copies for ANY dim % 32 == 0, M and ef.  The stored values are widened back
    exactly and the arithmetic is the reference's f32 kernel, so the mode is checked EXACTLY: ids, similarity bits
    and work counters equal the oracle's on the same graph with the vectors the engine reports as stored; the
    stored values themselves are within the format's rounding of the originals."""
    n = 900
    V = make_data(n, dim, seed=51)
    o, lv = build_oracle(oracle_mod, V, m, ef)
    g = o.export()
    g["vectors"] = V
    gi = eng.Index("cmp", dim, m, ef)
    gi.import_graph(g)
    Q = make_data(64, dim, seed=52)
    ids32, _, _ = gi.search_batch(Q, k)
    bytes32 = gi.info().hbm_bytes
    gi.set_tuning("compress_" + fmt, 1)
    assert gi.info().hbm_bytes <= bytes32 - n * dim * (2 if fmt == "bf16" else 3)
    Vs = _stored(gi, n)
    rel = np.abs(Vs - V) / np.maximum(np.abs(V), 2.0 ** -6)
    assert rel.max() <= (2.0 ** -8 if fmt == "bf16" else 2.0 ** -4) * 1.0001          # half an ulp of 8 / 4 significant bits
    if fmt == "fp8":
        import torch
        want = torch.from_numpy(V).to(torch.float8_e4m3fn).to(torch.float32).numpy()   # OCP e4m3, round to nearest even
        assert np.array_equal(Vs.view(np.uint32), want.view(np.uint32))
    g2 = dict(g)
    g2["vectors"] = Vs
    o2 = oracle_mod.OracleIndex.from_graph(dim, m, ef, g2)
    gi.set_tuning("visited_bounded", 0)
    gi.reset_counters()
    ids, sims, n_out = gi.search_batch(Q, k)
    oids, osims, on, oct = o2.search_batch(Q, k, threads=8)
    assert np.array_equal(n_out, on) and np.array_equal(ids, oids) and np.array_equal(_bits(sims), _bits(osims))
    sc, _ = gi.counters()
    assert (sc.n_dist, sc.n_ids, sc.n_expand) == (oct.n_dist, oct.n_ids, oct.n_expand)
    gi.set_tuning("visited_bounded", 1)
    ids_b, sims_b, _ = gi.search_batch(Q, k)
    assert np.array_equal(ids_b, oids) and np.array_equal(_bits(sims_b), _bits(osims))
    r = gi.search_knn(Q[0], k)                                       # the single-query entry point
    assert [x.id for x in r] == oids[0][: len(r)].tolist()
    overlap = np.mean([len(set(a.tolist()) & set(b.tolist())) / k for a, b in zip(ids, ids32)])
    assert overlap > (0.9 if fmt == "bf16" else 0.5), overlap      # a different (rounded) dataset, close to the original
    # read-only, one way
    with pytest.raises(eng.HNSWError):
        gi.add_node("x", V[0])
    with pytest.raises(eng.HNSWError):
        gi.set_tuning("compress_fp8" if fmt == "bf16" else "compress_bf16", 1)
    gi.close(); o.close(); o2.close()


def test_compressed_storage_limits_and_knobs(eng, oracle_mod):
    V = make_data(600, 100, seed=53)
    a = eng.Index("cmp-scalar", 100, 8, 32)                          # dim % 32 != 0: the scalar metric order has no compressed form
    a.add_batch(V, mode="fast")
    before = a.search_batch(V[:20], 5)
    for key in ("compress_bf16", "compress_fp8"):
        with pytest.raises(eng.HNSWError) as e:
            a.set_tuning(key, 1)
        assert "dim % 32" in e.value.msg
    after = a.search_batch(V[:20], 5)
    assert np.array_equal(before[0], after[0]) and np.array_equal(_bits(before[1]), _bits(after[1]))
    a.add_node("still-writable", V[0] * 0.5)
    a.close()
    # a bf16 dim-128 index: every tuning that takes the specialised kernel away now falls back to the general
    # kernel's bf16 form -- same answers
    W = make_data(1500, 128, seed=3)
    Q = make_data(32, 128, seed=4)
    c = eng.Index("bf-ok", 128, 16, 200)
    c.add_batch(W, mode="fast")
    c.set_tuning("compress_bf16", 1)
    ok = c.search_batch(Q, 10)
    assert c.last_search_was_lean()
    for key, val in (("lean", 0), ("visited_bounded", 0), ("tag_table", 0), ("force_restride", 80)):
        c.set_tuning(key, val)
        again = c.search_batch(Q, 10)
        assert not c.last_search_was_lean()
        assert np.array_equal(ok[0], again[0]) and np.array_equal(_bits(ok[1]), _bits(again[1])), key
    c.close()


def test_delete_on_a_one_directional_graph_reports_every_row_it_edited(eng):
    """fast-built graphs have links without a reverse: HNSW.NODE.DEL sweeps them away (k_purge_inbound), and the
    owners of the swept rows are part of the touched set the caller writes through (the persisted hnswnodet
    values would keep the deleted key otherwise)."""
    n, dim, m, ef = 5000, 32, 8, 64
    V = make_data(n, dim, seed=31)
    gi = eng.Index("purge", dim, m, ef)
    gi.set_tuning("fast_seed", 128)
    gi.add_batch(V, mode="fast")
    Q = make_data(100, dim, seed=2)
    ids0, _, _ = gi.search_batch(Q, 10)
    uniq, cnt = np.unique(ids0.ravel(), return_counts=True)
    checked = 0
    for victim in [int(x) for x in uniq[np.argsort(-cnt)][:12]]:
        g = gi.export_graph()
        inbound = set()
        for rp, col in zip(g["row_ptr"], g["col"]):
            rows = np.searchsorted(rp, np.nonzero(col == victim)[0], side="right") - 1
            inbound |= set(int(r) for r in rows)
        touched = set()
        gi.delete_node("node%d" % victim, update_fn=lambda name, i: touched.add(int(i)))
        assert inbound - {victim} <= touched, sorted(inbound - touched)[:5]
        own = set(int(x) for l in range(len(g["col"])) for x in g["col"][l][int(g["row_ptr"][l][victim]):int(g["row_ptr"][l][victim + 1])])
        checked += len(inbound - own - {victim})                       # rows only the sweep could have found
        g2 = gi.export_graph()
        assert not any((c == victim).any() for c in g2["col"])
    assert checked > 0, "no one-directional inbound link was exercised"
    gi.close()


# ---- the reference's parameter range: any M (core.rs:322-347); the engine serves M up to 64 --------------------
@pytest.mark.parametrize("m,dim,ef,n", [(48, 32, 64, 1500), (64, 128, 100, 1200), (40, 4, 48, 900)])
def test_large_m_builds_the_reference_graph(eng, oracle_mod, m, dim, ef, n):
    """M > 32: the shrink's select_neighbors(m_max0 = 2M) selects up to 128 links (two register slices of S),
    rows are up to 2M + slack ids wide.  Exact inserts (windowed and serial), deletes and searches equal the oracle's."""
    from tests.util import graphs_equal
    V = make_data(n + 30, dim, seed=41)
    lv = oracle_mod.draw_levels(n + 30, m, 7)
    o = oracle_mod.OracleIndex(dim, m, ef)
    o.add_batch(V[:n], lv[:n])
    gi = eng.Index("bigm", dim, m, ef)
    gi.add_batch(V[:n], levels=lv[:n], mode="exact")                 # windowed exact build
    ok, why = graphs_equal(o.export(), gi.export_graph())
    assert ok, why
    assert gi.info().max_degree0 > 64                                  # rows really are wider than one wave load
    for i in range(n, n + 30):                                         # serial exact inserts (hnsw_add)
        gi.add_node("late%d" % i, V[i], level=int(lv[i]))
        o.add(V[i], int(lv[i]))
    for i in (7, 333, 800):
        gi.delete_node("node%d" % i)
        o.delete(i)
    ok, why = graphs_equal(o.export(), gi.export_graph())
    assert ok, why
    Q = make_data(100, dim, seed=42)
    k = 10
    gi.set_tuning("visited_bounded", 0)
    gi.reset_counters()
    ids, sims, n_out = gi.search_batch(Q, k)
    oids, osims, on, oct = o.search_batch(Q, k, threads=8)
    assert np.array_equal(n_out, on)
    for q in range(len(Q)):
        nv = int(on[q])
        assert np.array_equal(ids[q, :nv], oids[q, :nv]) and np.array_equal(_bits(sims[q, :nv]), _bits(osims[q, :nv]))
    sc, _ = gi.counters()
    assert (sc.n_dist, sc.n_ids, sc.n_expand) == (oct.n_dist, oct.n_ids, oct.n_expand)
    gi.close(); o.close()


def test_m_above_128_is_refused_with_the_limit_in_the_message(eng):
    with pytest.raises(eng.HNSWError) as e:
        eng.Index("toobig", 16, 129, 100)
    assert "M <= 128" in e.value.msg
    gi = eng.Index("fast64", 64, 64, 128)                              # and the fast build works at the limit
    V = make_data(3000, 64, seed=43)
    gi.add_batch(V, mode="fast")
    ids, _, n_out = gi.search_batch(V[:50], 5)
    assert np.all(n_out == 5) and np.mean(ids[:, 0] == np.arange(50)) > 0.9
    gi.close()


def test_pipelined_search_waits_for_the_lazily_filled_spill_tables(eng, oracle_mod):
    """The HBM visited tables are (re)allocated and filled on the engine's stream when a search finds them too small --
    the first search, or the first one after the index grew or restrode.  A host batch split over the engine's lanes
    took its dependency on that stream BEFORE the fill was enqueued, so the chunk on lane 1 ran under the fill and lost
    visited marks: duplicate ids in its answers (scripts/fuzz_search.py seed 410169, reproduced 5 of 5 with
    scripts/repro_search.py: 270 of the second chunk's 750 queries)."""
    n, dim, m, ef, k = 20000, 64, 16, 200, 10
    V = make_data(n, dim, seed=71)
    gi = eng.Index("spillrace", dim, m, ef)
    gi.add_batch(V, levels=oracle_mod.draw_levels(n, m, 7), mode="fast")
    g = gi.export_graph()
    g["vectors"] = V
    o = oracle_mod.OracleIndex.from_graph(dim, m, ef, g)
    Q = make_data(1500, dim, seed=72)
    want = o.search_batch(Q, k, threads=8)
    gi.set_tuning("visited_bounded", 0)                       # the exact set: every query moves to its HBM table ...
    gi.set_tuning("lds_hash_bits", 8)                         # ... after 224 ids
    for widen in (16, 16):                                    # each restride makes the next search re-allocate the tables
        gi.set_tuning("force_restride", widen)
        ids, sims, n_out = gi.search_batch(Q, k)              # two chunks of 750 on two lanes
        assert np.array_equal(n_out, want[2])
        bad = [b for b in range(Q.shape[0]) if not np.array_equal(ids[b], want[0][b]) or not np.array_equal(_bits(sims[b]), _bits(want[1][b]))]
        assert not bad, "queries %s ... differ (%d)" % (bad[:6], len(bad))
        assert all(len(set(r.tolist())) == k for r in ids)    # no id twice in an answer
    sc, _ = gi.counters()
    assert sc.n_spill > 0                                     # the HBM tables were really in use
    gi.close(); o.close()


@pytest.mark.parametrize("m,ef,widen,n", [(16, 200, 0, 1500),    # C2's shape: narrow rows, W in four register slices
                                          (16, 200, 32, 1200),   # the same index after a restride: two row words per lane
                                          (32, 300, 0, 1000),    # M = 32 rows are wide from the start; eight slices
                                          (5, 40, 0, 1500),      # C1's M, one slice, many layers
                                          (8, 512, 0, 700)])     # the largest ef the specialised routine takes
def test_insert_plans_with_the_specialised_search_build_the_reference_graph(eng, oracle_mod, m, ef, widen, n):
    """dim-128 insert plans -- single hnsw_add calls and the windowed exact build -- search with the specialised
    routine (hnsw_plan_lean.hpp).  The graph must be the oracle's row for row, and the plans' work counters the
    ones the general plan kernels (plan_lean = 0) report: same W, same evaluations, same expansions."""
    dim = 128
    V = make_data(n, dim, seed=21)
    lv = oracle_mod.draw_levels(n, m, 9)
    o = oracle_mod.OracleIndex(dim, m, ef)
    o.add_batch(V, lv)
    want = o.export()
    got_counters = []
    for lean in (1, 0):
        gi = eng.Index("pl%d" % lean, dim, m, ef)
        gi.set_tuning("plan_lean", lean)
        gi.set_tuning("plan_split", 0)      # (the two-stage plans exist in the specialised kernel only: a stage redone after it went stale is extra work)
        a, b = n // 3, 2 * n // 3
        gi.add_batch(V[:a], levels=lv[:a], mode="exact")           # the window (read logs included)
        if widen:
            gi.set_tuning("force_restride", widen)
        for i in range(a, a + 40):                                 # single adds: the serial plan kernel
            gi.add_node("s%d" % i, V[i], level=int(lv[i]))
        gi.add_batch(V[a + 40:b], levels=lv[a + 40:b], mode="exact")
        gi.set_tuning("occ_window", 0)                             # the rest one by one through the batch entry point
        gi.add_batch(V[b:], levels=lv[b:], mode="exact")
        ok, why = graphs_equal(want, gi.export_graph())
        assert ok, "plan_lean=%d: %s" % (lean, why)
        _, ic = gi.counters()
        got_counters.append((ic.n_dist + ic.n_spill, ic.n_ids, ic.n_expand))
        gi.close()
    assert got_counters[0] == got_counters[1]
    o.close()
