"""N > 1 path on CPU: world_size-2 gloo processes exercise the query sharding,
the top-k all-gather and the one-time index broadcast that bench.py uses with
RCCL.  The per-rank "search" is the oracle here (no GPU in this tier), which
also proves sharded == unsharded results."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, tmpdir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import oracle
    from redis_hnsw_amd import shard
    from tests.util import graphs_equal, make_data
    n, dim, m, ef, k, B = 500, 32, 6, 32, 5, 64
    V = make_data(n, dim, seed=1)
    Q = make_data(B, dim, seed=2)
    # rank 0 builds, everyone receives the same graph
    graph = None
    if rank == 0:
        o = oracle.OracleIndex(dim, m, ef)
        o.add_batch(V, oracle.draw_levels(n, m, 7))
        graph = o.export()
    # vectors travel with the graph (rank 1 never sees V otherwise); arrays keep their native width
    g = shard.broadcast_graph(dist, graph, n, src=0, vectors=V if rank == 0 else None)
    if rank == 0:
        ok, why = graphs_equal({k_: graph[k_] for k_ in ("levels", "enterpoint", "max_layer", "row_ptr", "col")}, g)
        assert ok, why
    assert g["levels"].dtype == np.uint32 and g["row_ptr"][0].dtype == np.uint64 and g["col"][0].dtype == np.uint32
    assert np.array_equal(g["vectors"].view(np.uint32), V.view(np.uint32))
    replica = oracle.OracleIndex.from_graph(dim, m, ef, g)
    lo, hi = shard.shard_bounds(B, world, rank)
    ids, sims, n_out, _ = replica.search_batch(Q[lo:hi], k)
    all_ids, all_sims = shard.gather_topk(dist, torch.from_numpy(ids.astype(np.int64)), torch.from_numpy(sims), world)
    # unsharded reference on every rank
    rids, rsims, _, _ = replica.search_batch(Q, k)
    assert np.array_equal(all_ids.numpy(), rids.astype(np.int64))
    assert np.array_equal(all_sims.numpy().view(np.uint32), rsims.view(np.uint32))
    # the single-collective form bench.py uses: ids and similarity bits in one int32 buffer
    packed = torch.from_numpy(np.stack([ids.astype(np.int32), sims.view(np.int32)]))
    allp = shard.gather_packed(dist, packed, world).numpy()
    assert np.array_equal(allp[:, 0].reshape(-1, k), rids.astype(np.int32))
    assert np.array_equal(allp[:, 1].reshape(-1, k).view(np.uint32), rsims.view(np.uint32))
    dist.barrier()
    dist.destroy_process_group()
    open(os.path.join(tmpdir, "ok%d" % rank), "w").write("ok")


def test_shard_bounds_cover_batch():
    from redis_hnsw_amd import shard
    for B in (1, 7, 64, 1024, 1025):
        for world in (1, 2, 3, 8):
            spans = [shard.shard_bounds(B, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == B
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))


def test_two_rank_gloo_shard_gather_broadcast(tmp_path, oracle_mod):
    world = 2
    port = _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    assert all(os.path.exists(tmp_path / ("ok%d" % r)) for r in range(world))
