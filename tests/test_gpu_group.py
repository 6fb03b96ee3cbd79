"""One process, several GPUs (SURVEY 8e "one process, 8 devices, one stream each"): the C ABI's hnsw_group_* layer.
The GPU box has one device, so the members share it (a device may be listed more than once) -- what is checked is the
path's logic: replicas are exact copies, a sharded batch equals the unsharded one and the oracle's, replayed writes
keep every member's graph equal to the oracle's row for row, and a group that fell behind says so."""
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
def grp():
    from redis_hnsw_amd import group
    return group


def _same_answers(a, b, n_out=None):
    ids, sims, n = a
    oids, osims, on = b[0], b[1], b[2]
    assert np.array_equal(n, on)
    for r in range(ids.shape[0]):
        c = int(on[r])
        assert np.array_equal(ids[r, :c], oids[r, :c]) and np.array_equal(_bits(sims[r, :c]), _bits(osims[r, :c])), r


def _members_equal_oracle(g, o):
    og = o.export()
    for i in range(len(g)):
        ok, why = graphs_equal(og, g.member_export(i))
        assert ok, "member %d: %s" % (i, why)
        assert g.member_info(i).node_count == o.live_count


def _member_devices():
    """[(devices of the members besides the primary on device 0, id)]: all on the one device (this tier's box), and --
    the moment more than one GPU is visible -- one member per OTHER device: peer copies, per-device streams and
    hipDeviceEnablePeerAccess then run for the first time"""
    import torch
    nd = torch.cuda.device_count() if torch.cuda.is_available() else 0
    cases = [pytest.param([0, 0], id="one-device")]
    cases.append(pytest.param(list(range(1, nd)), id="one-member-per-gpu",
                              marks=pytest.mark.skipif(nd < 2, reason="needs at least two visible GPUs")))
    return cases


@pytest.mark.parametrize("devices", _member_devices())
def test_sharded_search_and_replayed_writes_match_the_oracle(eng, grp, oracle_mod, devices):
    n, dim, m, ef, k = 2500, 128, 16, 200, 10
    V = make_data(n + 400, dim, seed=31)
    o, lv = build_oracle(oracle_mod, V[:n], m, ef)
    graph = o.export()
    graph["vectors"] = V[:n]
    p = eng.Index("grp", dim, m, ef)
    p.import_graph(graph)
    g = grp.Group(p, devices)                                       # the primary + one member per listed device
    assert len(g) == 1 + len(devices)
    _members_equal_oracle(g, o)
    Q = make_data(701, dim, seed=32)                                # 701 = 233 + 234 + 234: ragged shards
    want = o.search_batch(Q, k, threads=8)
    _same_answers(g.search_batch(Q, k), want)
    _same_answers(p.search_batch(Q, k), want)                       # unsharded == sharded
    _same_answers(g.search_batch(Q[:2], k), (want[0][:2], want[1][:2], want[2][:2]))   # fewer queries than members
    # HNSW.NODE.ADD / DEL through the group: every member replays the exact operation, touched sets are the oracle's
    for j in range(6):
        lvl = int(oracle_mod.draw_levels(1, m, 900 + j)[0])
        oi, ot = o.add(V[n + j], lvl, want_touched=True)
        got = []
        gi = g.add_node("new%d" % j, V[n + j], lambda s, nid: got.append(nid), level=lvl)
        assert gi == oi and sorted(got) == sorted(ot.tolist())
    for i in (7, 1200, n + 2, -1):
        i = o.enterpoint if i < 0 else i                            # the last one re-elects the enterpoint (core.rs:449-472)
        ot = o.delete(int(i), want_touched=True)
        got = []
        g.delete_node(p._name_of(int(i)), lambda s, nid: got.append(nid))
        assert sorted(got) == sorted(ot.tolist())
    _members_equal_oracle(g, o)
    # bulk exact insert (the windowed exact path) on every member at once
    lv2 = oracle_mod.draw_levels(300, m, 77)
    o.add_batch(V[n + 6:n + 306], lv2)
    g.add_batch(V[n + 6:n + 306], levels=lv2, mode="exact")
    _members_equal_oracle(g, o)
    _same_answers(g.search_batch(Q, k), o.search_batch(Q, k, threads=8))
    # a level drawn by the group (level < 0) is the same on every member
    g.add_node("drawn", V[n + 306])
    lvls = [g.member_export(i)["levels"] for i in range(len(g))]
    assert all(np.array_equal(lvls[0], x) for x in lvls[1:])
    g0 = g.member_export(0)
    for i in range(1, len(g)):
        ok, why = graphs_equal(g0, g.member_export(i))
        assert ok, why
    g.close(); p.close(); o.close()


def test_group_that_fell_behind_says_so_and_refresh_recopies(eng, grp, oracle_mod):
    n, dim, m, ef, k = 1500, 64, 8, 64, 5
    V = make_data(n + 700, dim, seed=41)
    p = eng.Index("grp2", dim, m, ef)
    p.add_batch(V[:n], mode="exact")
    g = grp.Group(p, [0])
    Q = make_data(300, dim, seed=42)
    a = g.search_batch(Q, k)
    p.add_node("behind-the-group's-back", V[n])                     # written to the primary directly
    with pytest.raises(eng.HNSWError) as e:
        g.search_batch(Q, k)
    assert "hnsw_group_refresh" in e.value.msg
    g.refresh()
    b = g.search_batch(Q, k)
    _same_answers(b, p.search_batch(Q, k))
    # the fast build is not reproducible link for link: it runs on the primary, the replicas are re-copied
    g.add_batch(V[n + 1:n + 601], mode="fast")
    g0 = g.member_export(0)
    ok, why = graphs_equal(g0, g.member_export(1))
    assert ok, why
    _same_answers(g.search_batch(Q, k), p.search_batch(Q, k))
    # deletes on a fast-built (one-directional) graph replay identically too
    g.delete_node("node100")
    g.delete_node("node%d" % (n + 50))
    ok, why = graphs_equal(g.member_export(0), g.member_export(1))
    assert ok, why
    g.close(); p.close()


def test_group_from_an_empty_index_and_compressed_replicas(eng, grp, oracle_mod):
    dim, m, ef, k = 128, 16, 200, 10
    V = make_data(900, dim, seed=51)
    lv = oracle_mod.draw_levels(900, m, 7)
    o = oracle_mod.OracleIndex(dim, m, ef)
    p = eng.Index("grp3", dim, m, ef)
    g = grp.Group(p, [0, 0])                                        # nothing to copy yet
    for j in range(3):                                              # the first node (core.rs:393-405) and two more, one by one
        o.add(V[j], int(lv[j]))
        g.add_node("n%d" % j, V[j], level=int(lv[j]))
    o.add_batch(V[3:], lv[3:])
    g.add_batch(V[3:], levels=lv[3:], mode="exact")
    _members_equal_oracle(g, o)
    Q = make_data(200, dim, seed=52)
    _same_answers(g.search_batch(Q, k), o.search_batch(Q, k, threads=8))
    g.close()
    # a compressed serving copy replicates in its own format
    p.set_tuning("compress_bf16", 1)
    g2 = grp.Group(p, [0])
    want = p.search_batch(Q, k)
    _same_answers(g2.search_batch(Q, k), want)
    with pytest.raises(eng.HNSWError):
        g2.add_node("read-only", V[0])                              # the primary refuses; no replica is touched
    _same_answers(g2.search_batch(Q, k), want)
    g2.close(); p.close(); o.close()


def test_group_with_no_replicas_is_the_primary(eng, grp):
    p = eng.Index("grp4", 32, 5, 16)
    V = make_data(300, 32, seed=61)
    g = grp.Group(p, [])
    g.add_batch(V, mode="exact")
    assert len(g) == 1
    a = g.search_batch(V[:50], 3)
    _same_answers(a, p.search_batch(V[:50], 3))
    with pytest.raises(eng.HNSWError):
        grp.Group(p, [97])                                          # no such device: a loud error, with a reason
    g.close(); p.close()
