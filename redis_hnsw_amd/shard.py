"""Multi-GPU serving of the search path: a replicated index, the query batch
sharded across ranks, top-k gathered over the process group (RCCL on GPUs, gloo
in the CPU tests).  One process per GPU (torch.distributed); the search itself
needs no collective -- queries are independent (core.rs:477 takes &self) -- so
the only exchanges are the one-time index distribution and the [B,k] gather.
"""
import numpy as np


def shard_bounds(n_queries, world, rank):
    """Contiguous split: rank r serves queries [lo, hi)."""
    lo = (n_queries * rank) // world
    hi = (n_queries * (rank + 1)) // world
    return lo, hi


def gather_topk(dist, ids, sims, world, out_ids=None, out_sims=None):
    """All-gather equal-sized [b,k] shards into [world*b, k], rank-major (= query order
    for a contiguous split).  Works on CUDA tensors (RCCL) and CPU tensors (gloo)."""
    import torch
    if out_ids is None:
        out_ids = torch.empty((world * ids.shape[0],) + tuple(ids.shape[1:]), dtype=ids.dtype, device=ids.device)
    if out_sims is None:
        out_sims = torch.empty((world * sims.shape[0],) + tuple(sims.shape[1:]), dtype=sims.dtype, device=sims.device)
    dist.all_gather_into_tensor(out_ids, ids.contiguous())
    dist.all_gather_into_tensor(out_sims, sims.contiguous())
    return out_ids, out_sims


def gather_packed(dist, packed, world, out=None):
    """One collective instead of two: `packed` is an int32 [2, b, k] buffer whose plane 0 holds the ids
    and plane 1 the f32 similarities (bit pattern); returns [world, 2, b, k]."""
    import torch
    if out is None:   # concatenation layout (world * 2, b, k): accepted by both RCCL and gloo
        out = torch.empty((world * packed.shape[0],) + tuple(packed.shape[1:]), dtype=packed.dtype,
                          device=packed.device)
    dist.all_gather_into_tensor(out, packed.contiguous())
    return out.view((world,) + tuple(packed.shape))


class _DeviceBytes:
    """zero-copy view of raw device memory for torch.as_tensor (CUDA array interface, u8)"""

    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {"shape": (int(nbytes),), "typestr": "|u1", "data": (int(ptr), False), "version": 2}


def device_bytes(ptr, nbytes, device):
    import torch
    return torch.as_tensor(_DeviceBytes(ptr, nbytes), device=device)


def replicate_index(dist, index, src=0, device=None, via="device", chunk_bytes=1 << 30):
    """One-time index distribution: rank `src` holds the built `index`, every other rank an EMPTY
    Index with the same dim / M / ef_construction; on return all of them hold the same index (rows in the same
    stored order, enterpoint, tombstones, names).  The tables -- vectors, layer-0 rows, upper rows, upper slot
    bases, levels -- never leave HBM: hnsw_replica_view / _prepare hand out device pointers and the broadcasts
    run on zero-copy views of them (RCCL over xGMI), in native dtypes and the engine's own row layout.
    via="host" stages every piece through host memory instead (gloo, or ranks that share one device: a functional
    check).  Pieces larger than chunk_bytes are sent in slices.  Returns the bytes moved per replica."""
    import torch
    from . import _capi
    rank = dist.get_rank()
    meta = [None]
    if rank == src:
        r = index.replica_view()
        meta[0] = dict(scalars={k: int(getattr(r, k)) for k in _capi.Replica.SCALARS}, names=list(index._names),
                       dead=index.tombstones().tobytes() if int(r.n_dead) else None)
    dist.broadcast_object_list(meta, src=src)
    meta = meta[0]
    if rank != src:
        r = index.replica_prepare(meta["scalars"])
    n = int(r.n)
    pieces = [(r.vec, int(r.vec_bytes)), (r.adj0, int(r.adj0_bytes)), (r.adj_upper, int(r.adj_upper_bytes)),
              (r.upper_base, 4 * n), (r.levels, 4 * n)]
    dev = device if device is not None else torch.device("cuda", torch.cuda.current_device())
    moved = 0
    for ptr, nbytes in pieces:
        if not nbytes:
            continue
        whole = device_bytes(ptr, nbytes, dev)
        for lo in range(0, nbytes, chunk_bytes):
            t = whole[lo:lo + chunk_bytes]
            if via == "device":
                dist.broadcast(t, src=src)
            else:
                ht = t.cpu() if rank == src else torch.empty(t.shape, dtype=torch.uint8)
                dist.broadcast(ht, src=src)
                if rank != src:
                    t.copy_(ht)
        moved += nbytes
    torch.cuda.synchronize()
    if rank != src:
        dead = None if meta["dead"] is None else __import__("numpy").frombuffer(meta["dead"], dtype="uint8")
        index.replica_commit(r, dead, names=meta["names"])
    return moved


def broadcast_graph(dist, graph, n_nodes, src=0, device="cpu", vectors=None, dim=None):
    """Host-side form of the index distribution (per-layer CSR as produced by Index.export_graph()): what a CPU
    replica or a gloo process group uses; GPUs use replicate_index.  Arrays travel in their native width (uint32 /
    uint64, sent as the same-size signed type the backends support).  vectors: [n, dim] f32 on `src` to send the
    vector matrix too (dim must be given on the other ranks)."""
    import torch
    rank = dist.get_rank()
    meta = [None]
    if rank == src:
        meta[0] = dict(enterpoint=int(graph["enterpoint"]), max_layer=int(graph["max_layer"]),
                       nnz=[int(len(c)) for c in graph["col"]], dim=None if vectors is None else int(vectors.shape[1]))
    dist.broadcast_object_list(meta, src=src)
    meta = meta[0]
    L = meta["max_layer"] + 1

    def bc(arr, n, dtype):
        sdt = {np.dtype(np.uint32): np.int32, np.dtype(np.uint64): np.int64, np.dtype(np.float32): np.float32}[np.dtype(dtype)]
        if rank == src:
            t = torch.from_numpy(np.ascontiguousarray(arr, dtype=dtype).view(sdt).reshape(-1)).to(device)
        else:
            t = torch.empty(n, dtype=getattr(torch, np.dtype(sdt).name), device=device)
        if n:
            dist.broadcast(t, src=src)
        return t.cpu().numpy().view(dtype)

    levels = bc(graph["levels"] if rank == src else None, n_nodes, np.uint32)
    row_ptr, col = [], []
    for l in range(L):
        row_ptr.append(bc(graph["row_ptr"][l] if rank == src else None, n_nodes + 1, np.uint64))
        col.append(bc(graph["col"][l] if rank == src else None, meta["nnz"][l], np.uint32))
    out = dict(levels=levels, enterpoint=meta["enterpoint"], max_layer=meta["max_layer"], row_ptr=row_ptr, col=col)
    if meta["dim"]:
        out["vectors"] = bc(vectors if rank == src else None, n_nodes * meta["dim"], np.float32).reshape(n_nodes, meta["dim"])
    return out
