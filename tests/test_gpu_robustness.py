"""GPU tests of the behaviour around the hot path: deletes on fast-built graphs, per-call device status,
untrusted snapshots, drawn levels, snapshots continued against the oracle."""
import ctypes as C

import numpy as np
import pytest

from tests.util import graphs_equal, make_data

pytestmark = pytest.mark.gpu


def _bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


@pytest.fixture(scope="module")
def eng():
    from redis_hnsw_amd import index as idxmod
    return idxmod


def test_delete_after_fast_build_leaves_no_inbound_links(eng):
    """The fast build prunes one-directionally, so a deleted node can still be pointed at by rows that are
    not among its own neighbours: HNSW.NODE.DEL sweeps those links too and searches never return it."""
    n, dim, m, ef, k = 6000, 32, 8, 64, 10
    V = make_data(n, dim, seed=11)
    gi = eng.Index("foo", dim, m, ef)
    gi.set_tuning("fast_seed", 128)
    gi.add_batch(V, mode="fast")
    Q = make_data(200, dim, seed=2)
    ids0, _, _ = gi.search_batch(Q, k)
    # delete the most frequently returned nodes (the hubs: the ones most rows point at)
    uniq, cnt = np.unique(ids0.ravel(), return_counts=True)
    victims = [int(x) for x in uniq[np.argsort(-cnt)][:150]]
    for v in victims:
        gi.delete_node("node%d" % v)
    g = gi.export_graph()
    dead = set(victims)
    for col in g["col"]:
        assert not (set(col.tolist()) & dead), "a row still points at a deleted node"
    ids1, sims1, n1 = gi.search_batch(Q, k)
    assert not (set(ids1[ids1 != 0xFFFFFFFF].tolist()) & dead)
    assert np.all(n1 == k)
    # and the index keeps working in the reference's exact mode afterwards
    W = make_data(40, dim, seed=12)
    for i in range(40):
        gi.add_node("late%d" % i, W[i])
    ids2, _, _ = gi.search_batch(W, 1)
    assert gi.node_count == n - len(victims) + 40
    assert not (set(ids2.ravel().tolist()) & dead)
    gi.close()


def test_status_is_per_call_and_reimported_fast_graph_accepts_exact_ops(eng):
    """A fast-built graph that is exported and imported again (what a replica does) is recognised as having
    one-directional links; exact adds and deletes on it succeed and keep host names and engine ids in step."""
    n, dim, m, ef = 4000, 32, 8, 64
    V = make_data(n + 50, dim, seed=21)
    a = eng.Index("a", dim, m, ef)
    a.set_tuning("fast_seed", 128)
    a.add_batch(V[:n], mode="fast")
    g = a.export_graph()
    g["vectors"] = V[:n]
    b = eng.Index("b", dim, m, ef)
    b.import_graph(g)
    for i in range(n, n + 50):
        b.add_node("x%d" % i, V[i])                        # used to trip ST_ASYMMETRIC for good
    for i in range(0, 300, 7):
        b.delete_node("node%d" % i)
    for i in range(10):
        b.add_node("y%d" % i, V[i] + 0.5)
    assert b.node_count == n + 50 - len(range(0, 300, 7)) + 10
    ids, _, n_out = b.search_batch(V[n:n + 50], 1)
    assert np.array_equal(ids[:, 0], np.arange(n, n + 50))   # every late node finds itself
    r = b.search_knn(V[n + 3], 1)
    assert r[0].name == "x%d" % (n + 3)
    a.close(); b.close()


def test_corrupt_snapshots_are_rejected(eng):
    n, dim, m, ef = 300, 16, 5, 16
    V = make_data(n, dim, seed=31)
    a = eng.Index("s", dim, m, ef, seed=4)
    for i in range(n):
        a.add_node("n%d" % i, V[i])
    blob = bytearray(a.serialize())
    good = eng.Index.deserialize(bytes(blob))
    assert good.node_count == n
    good.close()
    nsnap = int.from_bytes(blob[:8], "little")
    hdr = 8                                   # [u64 length][SnapHeader ...]
    # SnapHeader: magic[8], version, dim, m, efc, n, n_dead, max_layer, n_layers, enterpoint(i64), rng[4]
    off = {"n": hdr + 8 + 16, "n_dead": hdr + 8 + 20, "max_layer": hdr + 8 + 24, "n_layers": hdr + 8 + 28,
           "enterpoint": hdr + 8 + 32}

    def mutated(field, value, width=4):
        b = bytearray(blob)
        b[off[field]:off[field] + width] = int(value).to_bytes(width, "little", signed=value < 0)
        return bytes(b)

    bad = [mutated("max_layer", 31), mutated("n_layers", 99), mutated("n_dead", 5), mutated("enterpoint", n + 7, 8),
           mutated("n", n + 1000), (nsnap // 2).to_bytes(8, "little") + bytes(blob[8:8 + nsnap // 2]) + bytes(blob[8 + nsnap:])]
    # a row_ptr that runs backwards (first layer's table starts after levels, tombstones, vectors)
    pad8 = lambda x: (x + 7) & ~7
    rp0 = hdr + 80 + pad8(n * 4) + pad8(n) + pad8(n * dim * 4) + 8
    b2 = bytearray(blob)
    b2[rp0 + 8 * 5:rp0 + 8 * 6] = (1 << 40).to_bytes(8, "little")
    bad.append(bytes(b2))
    for i, bb in enumerate(bad):
        with pytest.raises(eng.HNSWError):
            eng.Index.deserialize(bb)
    a.close()


def test_empty_index_device_search_pads_with_minus_inf(eng):
    import torch
    gi = eng.Index("e", 32, 5, 16)
    dev = torch.device("cuda", 0)
    q = torch.zeros((4, 32), dtype=torch.float32, device=dev)
    ids = torch.zeros((4, 3), dtype=torch.int32, device=dev)
    sims = torch.zeros((4, 3), dtype=torch.float32, device=dev)
    nn = torch.ones(4, dtype=torch.int32, device=dev)
    gi.search_batch_device(q.data_ptr(), 4, 3, ids.data_ptr(), sims.data_ptr(), nn.data_ptr(),
                           torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert nn.cpu().tolist() == [0, 0, 0, 0]                           # core.rs:481-483: not an error
    assert np.all(np.isneginf(sims.cpu().numpy()))
    assert np.all(ids.cpu().numpy().view(np.uint32) == 0xFFFFFFFF)
    gi.close()


@pytest.mark.parametrize("seed", [0, 7, 123456789])
def test_engine_and_oracle_draw_the_same_levels(eng, oracle_mod, seed):
    """level = floor(-ln U / ln m) (core.rs:601-605) from the same seeded generator on both sides: letting both
    DRAW (no explicit levels) must end in identical graphs."""
    n, dim, m, ef = 500, 32, 4, 24
    V = make_data(n, dim, seed=41)
    o = oracle_mod.OracleIndex(dim, m, ef, seed=seed)
    gi = eng.Index("d", dim, m, ef, seed=seed)
    for i in range(n):
        o.add(V[i])                                          # level = -1: draw
        gi.add_node("n%d" % i, V[i])
    go, gg = o.export(), gi.export_graph()
    assert go["max_layer"] >= 2                              # m = 4: several layers are in play
    ok, why = graphs_equal(go, gg)
    assert ok, why
    gi.close()


def test_snapshot_restored_engine_continues_like_the_oracle(eng, oracle_mod):
    """Restore from a snapshot (with tombstones, including a deleted enterpoint), then keep inserting and
    deleting: the restored engine must stay identical to the ORACLE that never stopped."""
    n, dim, m, ef = 500, 32, 5, 24
    V = make_data(n + 120, dim, seed=51)
    lv = oracle_mod.draw_levels(n + 120, m, 3)
    o = oracle_mod.OracleIndex(dim, m, ef)
    a = eng.Index("snap", dim, m, ef)
    o.add_batch(V[:n], lv[:n])
    a.add_batch(V[:n], levels=lv[:n], mode="exact")
    for i in (5, 99, o.enterpoint, 301):
        o.delete(int(i))
        a.delete_node("node%d" % i)
    b = eng.Index.deserialize(a.serialize())
    a.close()
    ok, why = graphs_equal(o.export(), b.export_graph())
    assert ok, why
    for i in range(n, n + 120):
        o.add(V[i], int(lv[i]))
        b.add_node("node%d" % i, V[i], level=int(lv[i]))
        if i % 10 == 0:
            o.delete(i - 7)
            b.delete_node("node%d" % (i - 7))
    ok, why = graphs_equal(o.export(), b.export_graph())
    assert ok, why
    Q = make_data(40, dim, seed=2)
    ids, sims, n_out = b.search_batch(Q, 5)
    oids, osims, on, _ = o.search_batch(Q, 5)
    assert np.array_equal(n_out, on) and np.array_equal(ids, oids) and np.array_equal(_bits(sims), _bits(osims))
    b.close()


def test_rust_shim_call_sequence_in_c(eng):
    """tests/cpp/shim_sequence.c: HNSW.NEW / NODE.ADD / NODE.DEL / SEARCH driven through the C ABI the way the
    Rust shim of INTEGRATION.md does it, with a stand-in keyspace for the hnswnodet write-through -- a
    REFERENCE-SHAPED one: a node promoted with l > l_max is saved with its pre-promotion rows only (core.rs:523),
    the hnswindex value keeps each node in the set of its top layer (core.rs:596) and is edited per command the way
    GpuIndex::sync_redis does it.  The index reloaded from that keyspace (levels from the layer sets, layer count
    from max_layer, src/lib.rs:287-299) then takes three more multi-level adds and a delete and must hold the
    oracle's rows, link for link, and give the oracle's answers."""
    import subprocess
    from redis_hnsw_amd import build
    exe = build.build_shim_test()
    import os
    gdir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    r = subprocess.run([exe, os.path.join(gdir, "shim_sequence.txt"), os.path.join(gdir, "shim_reload.txt")],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    # every HNSW.SEARCH answer through the shim's sequence equals the CPU oracle's (names + similarity bits)
    assert "shim_sequence ok" in r.stdout and "60 answers equal to the oracle's golden file" in r.stdout
    assert "every row and 20 answers equal to the oracle's after 3 more adds and a delete" in r.stdout


def test_reference_rdb_layout_round_trip_through_the_engine(eng, oracle_mod):
    """hnswindex + hnswnodet values (src/types.rs:243-284, 410-428) written from a live engine index (with
    tombstones), read back through make_index -> hnsw_import: same graph by NAME, same answers, and the
    restored index continues exactly like the oracle."""
    from redis_hnsw_amd import rdb
    n, dim, m, ef = 300, 16, 5, 24
    V = make_data(n + 40, dim, seed=61)
    lv = oracle_mod.draw_levels(n + 40, m, 8)
    o = oracle_mod.OracleIndex(dim, m, ef)
    a = eng.Index("hnsw.rdb", dim, m, ef)
    key = lambda i: "hnsw.rdb.n%d" % i
    for i in range(n):
        o.add(V[i], int(lv[i]))
        a.add_node(key(i), V[i], level=int(lv[i]))
    gone = [4, 77, int(o.enterpoint), 200]
    for i in gone:
        o.delete(i)
        a.delete_node(key(i))
    index_value, node_values = rdb.dump_index(a)
    assert len(node_values) == n - len(gone) and all(key(i) not in node_values for i in gone)
    ir = rdb.load_index(index_value)
    assert (ir.name, ir.node_count, ir.max_layer, ir.enterpoint) == ("hnsw.rdb", n - len(gone), o.max_layer, key(o.enterpoint))
    b = rdb.restore_index(index_value, node_values)
    assert b.node_count == n - len(gone) and b.enterpoint == key(o.enterpoint) and b.max_layer == o.max_layer
    # same graph by name (ids are compacted over the tombstones)
    live = [i for i in range(n) if i not in gone]
    for new, old in enumerate(live):
        for l in range(int(lv[old] if old else 0) + 1):
            if l > o.max_layer:
                break
            assert [key(int(j)) for j in o.neighbors(old, l)] == [b._names[int(j)] for j in b.neighbors(new, l)], (old, l)
    Q = make_data(30, dim, seed=2)
    for q in Q:
        ra, rb = a.search_knn(q, 5), b.search_knn(q, 5)
        assert [(r.name, r.sim) for r in ra] == [(r.name, r.sim) for r in rb]
    # both keep inserting: same names, same links
    for i in range(n, n + 40):
        a.add_node(key(i), V[i], level=int(lv[i]))
        b.add_node(key(i), V[i], level=int(lv[i]))
    ia, na = rdb.dump_index(a)
    ib, nb = rdb.dump_index(b)
    assert rdb.load_index(ia).enterpoint == rdb.load_index(ib).enterpoint
    assert set(na) == set(nb)
    assert all(rdb.load_node(na[k]) == rdb.load_node(nb[k]) for k in na)
    a.close(); b.close()
    # names that are not full keys are stored the way the reference stores them: hnsw.{idx} / hnsw.{idx}.{node}
    # (src/lib.rs:137, 342-343); replies keep the last '.' segment (core.rs:885-887)
    c = eng.Index("plain", dim, m, ef)
    for i in range(60):
        c.add_node("n%d" % i, V[i], level=int(lv[i]))
    iv, nv = rdb.dump_index(c)
    assert rdb.load_index(iv).name == "hnsw.plain" and set(nv) == {"hnsw.plain.n%d" % i for i in range(60)}
    d = rdb.restore_index(iv, nv)
    assert [r.name for r in d.search_knn(Q[0], 5)] == [r.name for r in c.search_knn(Q[0], 5)]
    assert all(r.name.startswith("n") and "." not in r.name for r in d.search_knn(Q[0], 5))
    c.close(); d.close()


def _bf16_round(V):
    """f32 -> bf16 (round to nearest even) -> f32: the values the compressed index stores"""
    u = np.ascontiguousarray(V, dtype=np.float32).view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
    return r.astype(np.uint32).view(np.float32).reshape(V.shape)


def test_bf16_storage_mode_is_the_reference_on_rounded_vectors(eng, oracle_mod):
    """SURVEY 8 f-4, first step: a read-only serving copy with bf16 vectors (half the gather bytes).  The
    arithmetic stays the reference's f32 AVX2 kernel on the stored values, so the mode is checked EXACTLY:
    ids and similarity bits equal the oracle's on the same graph with bf16-rounded vectors.  Against the f32
    index the similarities move by at most 2^-7 relative (each component carries <= 2^-9 relative rounding
    error; squared differences of values in [0,1) amplify it), and the top-10 sets overlap almost entirely."""
    n, dim, m, ef, k, nq = 3000, 128, 16, 200, 10, 96
    V = make_data(n, dim, seed=1)
    lv = oracle_mod.draw_levels(n, m, 7)
    o = oracle_mod.OracleIndex(dim, m, ef)
    o.add_batch(V, lv)
    g = o.export()
    gi = eng.Index("b", dim, m, ef)
    gi.import_graph(g)
    Q = make_data(nq, dim, seed=2)
    ids32, sims32, _ = gi.search_batch(Q, k)
    bytes32 = gi.info().hbm_bytes
    gi.set_tuning("compress_bf16", 1)
    assert gi.info().hbm_bytes <= bytes32 - n * dim * 2            # the vector matrix halved
    g16 = dict(g)
    g16["vectors"] = _bf16_round(V)
    o16 = oracle_mod.OracleIndex.from_graph(dim, m, ef, g16)
    oids, osims, on, oct = o16.search_batch(Q, k)
    gi.reset_counters()
    ids, sims, n_out = gi.search_batch(Q, k)
    assert np.array_equal(n_out, on) and np.array_equal(ids, oids) and np.array_equal(_bits(sims), _bits(osims))
    sc, _ = gi.counters()
    assert (sc.n_dist, sc.n_ids, sc.n_expand) == (oct.n_dist, oct.n_ids, oct.n_expand)
    # the 8-waves-per-CU launch shape and the single-query entry point
    gi.set_tuning("launch_concurrency", 8)
    Qw = np.concatenate([Q, Q, Q])[:272]
    idsw, simsw, _ = gi.search_batch(Qw, k)
    assert np.array_equal(idsw[:nq], oids) and np.array_equal(_bits(simsw[:nq]), _bits(osims))
    r = gi.search_knn(Q[0], k)
    assert [x.id for x in r] == oids[0].tolist()
    # stored value of a vector = its bf16 rounding
    assert np.array_equal(_bits(gi._vector(5)), _bits(_bf16_round(V[5:6])[0]))
    # tolerance against the f32 index, stated: similarities within 2^-7 relative on shared ids, overlap >= 0.97
    overlap = np.mean([len(set(a.tolist()) & set(b.tolist())) / k for a, b in zip(ids, ids32)])
    assert overlap >= 0.97
    for a, sa, b, sb in zip(ids, sims, ids32, sims32):
        common = {int(x): float(s) for x, s in zip(b, sb)}
        for x, s in zip(a, sa):
            if int(x) in common:
                assert abs(float(s) - common[int(x)]) <= 2.0 ** -7 * abs(common[int(x)])
    # read-only
    with pytest.raises(eng.HNSWError):
        gi.add_node("late", V[0])
    with pytest.raises(eng.HNSWError):
        gi.delete_node("node3")
    gi.close()


def test_inserts_wait_for_searches_in_flight_on_other_streams(eng, oracle_mod):
    """hnsw_search_batch_device only enqueues; an insert / delete issued right behind searches that are still
    running on OTHER streams must not change the graph under them.  Six streams (more than the runtime's
    default hardware queues), large batches so that the searches are still running when the inserts are
    issued: every search must equal the oracle's on the graph as it was when the search was enqueued."""
    import torch
    n, dim, m, ef, k, B = 6000, 128, 16, 200, 10, 3000
    V = make_data(n + 400, dim, seed=31)
    lv = oracle_mod.draw_levels(n + 400, m, 9)
    o = oracle_mod.OracleIndex(dim, m, ef)
    o.add_batch(V[:n], lv[:n])
    gi = eng.Index("s", dim, m, ef)
    gi.import_graph(o.export())
    streams = [torch.cuda.Stream() for _ in range(6)]
    gi.set_tuning("launch_concurrency", len(streams))
    Q = make_data(B, dim, seed=32)
    dQ = torch.from_numpy(Q).cuda()
    torch.cuda.synchronize()
    outs, wants = [], []
    added = n
    for rnd in range(4):
        for st in streams:
            ids = torch.empty((B, k), dtype=torch.int32, device="cuda")
            sims = torch.empty((B, k), dtype=torch.float32, device="cuda")
            nn = torch.empty(B, dtype=torch.int32, device="cuda")
            gi.search_batch_device(dQ.data_ptr(), B, k, ids.data_ptr(), sims.data_ptr(), nn.data_ptr(), st.cuda_stream)
            outs.append((ids, sims, nn, len(wants)))
        wants.append(o.search_batch(Q, k, threads=8))                    # the graph these searches must see
        # mutate right behind them: single adds, a windowed bulk add, a delete
        for i in range(added, added + 3):
            o.add(V[i], int(lv[i]))
            gi.add_node("node%d" % i, V[i], level=int(lv[i]))
        added += 3
        o.add_batch(V[added:added + 80], lv[added:added + 80])
        gi.add_batch(V[added:added + 80], levels=lv[added:added + 80], mode="exact")
        added += 80
        victim = 100 + rnd
        o.delete(victim)
        gi.delete_node("node%d" % victim)
    torch.cuda.synchronize()
    for ids, sims, nn, w in outs:
        oids, osims, on, _ = wants[w]
        assert np.array_equal(nn.cpu().numpy().astype(np.uint32), on)
        assert np.array_equal(ids.cpu().numpy().view(np.uint32), oids)
        assert np.array_equal(_bits(sims.cpu().numpy()), _bits(osims))
    ok, why = graphs_equal(o.export(), gi.export_graph())
    assert ok, why
    gi.close()


def _with_hub(g, hub, spokes):
    """layer-0 CSR of g with `hub` linked to every node of `spokes` it is not linked to yet (both directions,
    appended to the stored rows -- what core.rs:794 does to third parties, many times over)"""
    rp, col = g["row_ptr"][0].astype(np.int64), g["col"][0]
    rows = [list(col[rp[i]:rp[i + 1]]) for i in range(len(rp) - 1)]
    have = set(rows[hub])
    for s in spokes:
        if s != hub and s not in have:
            rows[hub].append(int(s))
            rows[s].append(int(hub))
            have.add(s)
    out = dict(g)
    out["row_ptr"] = [np.concatenate([[0], np.cumsum([len(r) for r in rows])]).astype(np.uint64)] + list(g["row_ptr"][1:])
    out["col"] = [np.array([x for r in rows for x in r], dtype=np.uint32)] + list(g["col"][1:])
    return out, len(rows[hub])


@pytest.mark.parametrize("dim,m,ef,spokes", [(32, 8, 40, 700), (128, 16, 200, 900), (12, 5, 24, 600)])
def test_a_hub_of_several_hundred_links_is_searched_shrunk_and_deleted_like_the_oracle(eng, oracle_mod, dim, m, ef, spokes):
    """The reference does not bound degrees (core.rs:790-796 appends to third parties without a shrink).
    Rows hold up to 1023 ids: an imported hub far beyond m_max0 is searched with the oracle's counters, shrunk back
    to m_max0 by the first insert that selects it (select_neighbors over all its links, every dropped link removed
    from the other side), and can be deleted -- graphs equal after every step."""
    n, k = 1500, 10
    V = make_data(n, dim, seed=31)
    lv = oracle_mod.draw_levels(n, m, 3)
    o0 = oracle_mod.OracleIndex(dim, m, ef)
    o0.add_batch(V, lv)
    g, deg = _with_hub(o0.export(), 5, range(100, 100 + spokes))
    o0.close()
    assert deg >= spokes
    o = oracle_mod.OracleIndex.from_graph(dim, m, ef, g)
    gi = eng.Index("hub", dim, m, ef)
    gi.import_graph(g)
    assert gi.info().max_degree0 == deg and gi.info().stride0 > deg
    ok, why = graphs_equal(o.export(), gi.export_graph())
    assert ok, why
    Q = np.concatenate([make_data(60, dim, seed=32), V[5:6] + 1e-3, V[100:140] + 1e-3]).astype(np.float32)
    gi.reset_counters()
    ids, sims, n_out = gi.search_batch(Q, k)
    oids, osims, on, oct = o.search_batch(Q, k)
    assert np.array_equal(ids, oids) and np.array_equal(_bits(sims), _bits(osims)) and np.array_equal(n_out, on)
    sc, _ = gi.counters()
    assert (sc.n_ids, sc.n_expand) == (oct.n_ids, oct.n_expand)
    # a spoke goes first (the hub re-selects nothing: it is only a neighbour of the deleted node), then inserts
    # next to the hub: the first one that links to it shrinks its row
    o.delete(150)
    gi.delete_node("node150")
    ok, why = graphs_equal(o.export(), gi.export_graph())
    assert ok, "after deleting a spoke: " + why
    rng = np.random.default_rng(33)
    near = (V[5][None, :] + 0.01 * rng.standard_normal((12, dim))).astype(np.float32)
    for i in range(6):
        o.add(near[i], 0)
        gi.add_node("near%d" % i, near[i], level=0)
        ok, why = graphs_equal(o.export(), gi.export_graph())
        assert ok, "after insert %d next to the hub: %s" % (i, why)
    go = o.export()
    assert int(go["row_ptr"][0][6] - go["row_ptr"][0][5]) <= 2 * m + 1      # the hub was shrunk (core.rs:560-569)
    lvb = np.zeros(6, dtype=np.int32)
    o.add_batch(near[6:], lvb)                                             # and through the window
    gi.add_batch(near[6:], levels=lvb, mode="exact")
    ok, why = graphs_equal(o.export(), gi.export_graph())
    assert ok, "after the windowed inserts: " + why
    o.delete(5)
    gi.delete_node("node5")
    ok, why = graphs_equal(o.export(), gi.export_graph())
    assert ok, "after deleting the hub: " + why
    ids, sims, n_out = gi.search_batch(Q, k)
    oids, osims, on, _ = o.search_batch(Q, k)
    assert np.array_equal(ids, oids) and np.array_equal(_bits(sims), _bits(osims)) and np.array_equal(n_out, on)
    gi.close(); o.close()
