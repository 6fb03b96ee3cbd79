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


def broadcast_graph(dist, graph, n_nodes, src=0, device="cpu"):
    """One-time index distribution: rank `src` holds `graph` (levels, enterpoint,
    max_layer, per-layer CSR as produced by Index.export_graph()); every rank
    returns an identical copy.  Vectors are not sent here: the benchmark
    regenerates them from the seed, a deployment would broadcast them the same way."""
    import torch
    rank = dist.get_rank()
    meta = [None]
    if rank == src:
        meta[0] = dict(enterpoint=int(graph["enterpoint"]), max_layer=int(graph["max_layer"]),
                       nnz=[int(len(c)) for c in graph["col"]])
    dist.broadcast_object_list(meta, src=src)
    meta = meta[0]
    L = meta["max_layer"] + 1

    def bc(arr, n, dtype):
        t = torch.from_numpy(np.ascontiguousarray(arr).astype(dtype)).to(device) if rank == src \
            else torch.empty(n, dtype=getattr(torch, np.dtype(dtype).name), device=device)
        dist.broadcast(t, src=src)
        return t.cpu().numpy()

    levels = bc(graph["levels"] if rank == src else None, n_nodes, np.int64).astype(np.uint32)
    row_ptr, col = [], []
    for l in range(L):
        row_ptr.append(bc(graph["row_ptr"][l] if rank == src else None, n_nodes + 1, np.int64).astype(np.uint64))
        col.append(bc(graph["col"][l] if rank == src else None, meta["nnz"][l], np.int64).astype(np.uint32))
    return dict(levels=levels, enterpoint=meta["enterpoint"], max_layer=meta["max_layer"], row_ptr=row_ptr, col=col)
