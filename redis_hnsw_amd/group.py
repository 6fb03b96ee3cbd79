"""One process, several GPUs ("one process, 8 devices, one stream each") over the C ABI's hnsw_group_*
entry points -- what a Redis module, a single process, uses where bench.py runs one rank per GPU.

A Group wraps an Index (the primary, which keeps the name map) and one replica per further device.  search_batch
shards the batch contiguously over the members; add_node / delete_node / add_batch(mode="exact") are replayed on every
member (the reference's insert and delete are deterministic given the level, core.rs:489-599, :414-475), so the
members stay identical row for row; add_batch(mode="fast") builds on the primary and re-copies the replicas."""
import ctypes as C

import numpy as np

from . import _capi
from .index import HNSWError, Index, _f32, _fp, _u32p


class Group:
    def __init__(self, primary, devices, seed=0):
        """primary: an Index; devices: HIP ordinals for the replicas (repeats allowed: several members per GPU)"""
        if not isinstance(primary, Index):
            raise TypeError("Group(primary: Index, devices)")
        self._lib = _capi.load()
        self.primary = primary
        devs = (C.c_int * max(len(devices), 1))(*[int(d) for d in devices])
        g = _capi.H()
        st = self._lib.hnsw_group_create(primary._h, devs, len(devices), int(seed), C.byref(g))
        self._g = g
        if st != _capi.OK:
            msg = self._lib.hnsw_group_last_error(g).decode() if g else "hnsw_group_create failed"
            self.close()
            raise HNSWError(msg, st)

    def close(self):
        """destroys the replicas; the primary Index stays the caller's"""
        if getattr(self, "_g", None):
            self._lib.hnsw_group_destroy(self._g)
            self._g = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, st):
        if st != _capi.OK:
            raise HNSWError(self._lib.hnsw_group_last_error(self._g).decode(), st)

    def __len__(self):
        return int(self._lib.hnsw_group_size(self._g))

    def member_info(self, i):
        h = self._lib.hnsw_group_member(self._g, int(i))
        if not h:
            raise IndexError(i)
        info = _capi.Info()
        if self._lib.hnsw_get_info(h, C.byref(info)) != _capi.OK:
            raise HNSWError(self._lib.hnsw_last_error(h).decode())
        return info

    def member_export(self, i):
        """the graph of member i (per-layer CSR in stored order), through a borrowed Index view of its handle"""
        view = Index.__new__(Index)
        view.__dict__.update(self.primary.__dict__)
        view._h = self._lib.hnsw_group_member(self._g, int(i))
        try:
            return view.export_graph()
        finally:
            view._h = None                                   # borrowed: never destroyed through the view

    def refresh(self):
        self._check(self._lib.hnsw_group_refresh(self._g))

    # -- HNSW.SEARCH, sharded ---------------------------------------------------------------------------
    def search_batch(self, Q, k):
        Q = _f32(Q)
        dim = self.primary.data_dim
        if Q.ndim != 2 or Q.shape[1] != dim:
            raise HNSWError("data dimension: %d does not match Index" % (Q.shape[-1]), _capi.ERR_DIM_MISMATCH)
        B = Q.shape[0]
        ids = np.full((B, k), 0xFFFFFFFF, dtype=np.uint32)
        sims = np.full((B, k), -np.inf, dtype=np.float32)
        n_out = np.zeros(B, dtype=np.uint32)
        self._check(self._lib.hnsw_group_search_batch(self._g, _fp(Q), B, dim, k, _u32p(ids), _fp(sims), _u32p(n_out)))
        return ids, sims, n_out

    # -- writes, replayed on every member ---------------------------------------------------------------
    def add_node(self, name, data, update_fn=None, level=-1):
        p = self.primary
        data = _f32(data).ravel()
        if data.size != p.data_dim:                          # core.rs:389-391
            raise HNSWError("data dimension: %d does not match Index" % data.size, _capi.ERR_DIM_MISMATCH)
        if p.node_count != 0 and name in p._ids:             # core.rs:407-409
            raise HNSWError('Node: "%s" already exists' % name, _capi.ERR_DUPLICATE)
        out_id = C.c_uint32(0)
        cap = 65536
        touched = np.empty(cap, dtype=np.uint32)
        nt = C.c_uint32(0)
        self._check(self._lib.hnsw_group_add(self._g, _fp(data), data.size, int(level), C.byref(out_id), _u32p(touched), cap,
                                             C.byref(nt)))
        i = int(out_id.value)
        while len(p._names) <= i:
            p._names.append(None)
        p._names[i] = name
        p._ids[name] = i
        if update_fn is not None:                            # core.rs:580-584
            for t in p._touched(touched, nt, cap):
                update_fn(p._name_of(int(t)), int(t))
        return i

    def delete_node(self, name, update_fn=None):
        p = self.primary
        i = p._ids.get(name)
        if i is None:                                        # core.rs:419-422
            raise HNSWError('Node: "%s" does not exist' % name, _capi.ERR_NOT_FOUND)
        cap = 65536
        touched = np.empty(cap, dtype=np.uint32)
        nt = C.c_uint32(0)
        self._check(self._lib.hnsw_group_delete(self._g, i, _u32p(touched), cap, C.byref(nt)))
        del p._ids[name]
        p._names[i] = None
        if update_fn is not None:                            # core.rs:441-446
            for t in p._touched(touched, nt, cap):
                update_fn(p._name_of(int(t)), int(t))

    def add_batch(self, vectors, names=None, levels=None, mode="exact"):
        p = self.primary
        V = _f32(vectors)
        if V.ndim != 2 or V.shape[1] != p.data_dim:
            raise HNSWError("data dimension: %d does not match Index" % (V.shape[1] if V.ndim == 2 else V.size),
                            _capi.ERR_DIM_MISMATCH)
        n = V.shape[0]
        base = len(p._names)
        names = list(names) if names is not None else ["node%d" % (base + i) for i in range(n)]
        lv = np.ascontiguousarray(levels, dtype=np.int32) if levels is not None else None
        self._check(self._lib.hnsw_group_add_batch(self._g, _fp(V), n, p.data_dim,
                                                   lv.ctypes.data_as(_capi.i32p) if lv is not None else None,
                                                   0 if mode == "exact" else 1))
        for i, nm in enumerate(names):
            p._ids[nm] = base + i
        p._names.extend(names)
