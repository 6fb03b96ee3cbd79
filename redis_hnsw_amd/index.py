"""Host-side mirror of the reference's Index<f32,f32> (src/hnsw/core.rs:303-486)
over the C ABI.  Names live here (the reference keeps `nodes: HashMap<String,
Node>` in the Rust module, core.rs:316); the engine speaks dense u32 ids."""
import ctypes as C
from collections import namedtuple

import numpy as np

from . import _capi

SearchResult = namedtuple("SearchResult", ["sim", "name", "id"])  # core.rs:48-52 (data omitted: src/lib.rs never reads it)


class HNSWError(Exception):
    """core.rs:25-46.  error_string() is the Debug rendering the Redis client sees."""

    def __init__(self, msg, status=_capi.ERR_INVALID):
        super().__init__(msg)
        self.msg = msg
        self.status = status

    def error_string(self):
        return 'String("%s")' % self.msg.replace("\\", "\\\\").replace('"', '\\"')


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _fp(a):
    return a.ctypes.data_as(_capi.fp)


def _u32p(a):
    return a.ctypes.data_as(_capi.u32p)


class Index:
    """Index::new(name, euclidean, data_dim, m, ef_construction) (core.rs:322-347)."""

    def __init__(self, name, data_dim, m=5, ef_construction=200, seed=0, device=0):
        self._lib = _capi.load()
        self.name = name
        self.data_dim = int(data_dim)
        self.m = int(m)
        self.m_max = self.m
        self.m_max_0 = 2 * self.m
        self.ef_construction = int(ef_construction)
        self.level_mult = 1.0 / np.log(float(m)) if m > 1 else float("inf")
        self._names = []          # id -> node name
        self._ids = {}            # node name -> id
        h = _capi.H()
        st = self._lib.hnsw_create(self.data_dim, self.m, self.ef_construction, seed, device, C.byref(h))
        self._h = h
        if st != _capi.OK:
            msg = self._lib.hnsw_last_error(h).decode() if h else "hnsw_create failed"
            self._lib.hnsw_destroy(h)
            self._h = None
            raise HNSWError(msg, st)

    # -- lifetime ---------------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None):
            self._lib.hnsw_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, st):
        if st != _capi.OK:
            raise HNSWError(self._lib.hnsw_last_error(self._h).decode(), st)

    # -- state the reference exposes as pub fields (core.rs:313-317) -------------
    def info(self):
        i = _capi.Info()
        self._check(self._lib.hnsw_get_info(self._h, C.byref(i)))
        return i

    @property
    def node_count(self):
        return int(self.info().node_count)

    @property
    def max_layer(self):
        return int(self.info().max_layer)

    @property
    def enterpoint(self):
        """name of the enterpoint node, or None (core.rs:317)."""
        e = int(self.info().enterpoint)
        return None if e < 0 else self._name_of(e)

    @property
    def enterpoint_id(self):
        return int(self.info().enterpoint)

    def _name_of(self, i):
        return self._names[i] if i < len(self._names) and self._names[i] is not None else str(i)

    # -- HNSW.NODE.ADD (core.rs:383-412) ------------------------------------------
    def add_node(self, name, data, update_fn=None, level=-1):
        data = _f32(data).ravel()
        if data.size != self.data_dim:                       # core.rs:389-391
            raise HNSWError("data dimension: %d does not match Index" % data.size, _capi.ERR_DIM_MISMATCH)
        # core.rs:393-409: the duplicate check sits after the first-node branch
        if self.node_count != 0 and name in self._ids:
            raise HNSWError('Node: "%s" already exists' % name, _capi.ERR_DUPLICATE)
        out_id = C.c_uint32(0)
        if update_fn is not None:
            cap = 65536
            touched = np.empty(cap, dtype=np.uint32)
            nt = C.c_uint32(0)
            self._check(self._lib.hnsw_add(self._h, _fp(data), data.size, int(level), C.byref(out_id),
                                           _u32p(touched), cap, C.byref(nt)))
        else:
            self._check(self._lib.hnsw_add(self._h, _fp(data), data.size, int(level), C.byref(out_id),
                                           None, 0, None))
        i = int(out_id.value)
        assert i == len(self._names)
        self._names.append(name)
        self._ids[name] = i
        if update_fn is not None:                              # core.rs:580-584
            for t in self._touched(touched, nt, cap):
                update_fn(self._name_of(int(t)), int(t))
        return i

    @staticmethod
    def _touched(buf, nt, cap):
        """the ids update_fn is called with; the engine reports how many there are even when the buffer is smaller"""
        if nt.value > cap:
            raise HNSWError("update_fn list of %d ids does not fit the %d-entry buffer" % (nt.value, cap), _capi.ERR_CAPACITY)
        return buf[: nt.value]

    # -- HNSW.NODE.DEL (core.rs:414-475) --------------------------------------------
    def delete_node(self, name, update_fn=None):
        i = self._ids.get(name)
        if i is None:                                          # core.rs:419-422
            raise HNSWError('Node: "%s" does not exist' % name, _capi.ERR_NOT_FOUND)
        cap = 65536
        touched = np.empty(cap, dtype=np.uint32)
        nt = C.c_uint32(0)
        self._check(self._lib.hnsw_delete(self._h, i, _u32p(touched), cap, C.byref(nt)))
        del self._ids[name]
        self._names[i] = None
        if update_fn is not None:                              # core.rs:441-446
            for t in self._touched(touched, nt, cap):
                update_fn(self._name_of(int(t)), int(t))

    def add_batch(self, vectors, names=None, levels=None, mode="exact"):
        """Bulk NODE.ADD (hnsw_add_batch).  mode 'exact' replays the reference's
        serial algorithm; 'fast' is the batched GPU build (recall parity only)."""
        V = _f32(vectors)
        n = V.shape[0]
        if V.ndim != 2 or V.shape[1] != self.data_dim:
            raise HNSWError("data dimension: %d does not match Index" % (V.shape[1] if V.ndim == 2 else V.size),
                            _capi.ERR_DIM_MISMATCH)
        base = len(self._names)
        names = list(names) if names is not None else ["node%d" % (base + i) for i in range(n)]
        lv = None
        if levels is not None:
            lv = np.ascontiguousarray(levels, dtype=np.int32)
        self._check(self._lib.hnsw_add_batch(self._h, _fp(V), n, self.data_dim,
                                             lv.ctypes.data_as(_capi.i32p) if lv is not None else None,
                                             0 if mode == "exact" else 1))
        for i, nm in enumerate(names):
            self._ids[nm] = base + i
        self._names.extend(names)

    # -- HNSW.SEARCH (core.rs:477-486) ---------------------------------------------
    def search_knn(self, data, k):
        data = _f32(data).ravel()
        if data.size != self.data_dim:                       # core.rs:478-480
            raise HNSWError("data dimension: %d does not match Index" % data.size, _capi.ERR_DIM_MISMATCH)
        ids = np.empty(max(k, 1), dtype=np.uint32)
        sims = np.empty(max(k, 1), dtype=np.float32)
        n = C.c_uint32(0)
        self._check(self._lib.hnsw_search(self._h, _fp(data), data.size, k, _u32p(ids), _fp(sims), C.byref(n)))
        # core.rs:885-887: the reply name is the last '.'-separated segment of the node key
        return [SearchResult(float(sims[i]), self._name_of(int(ids[i])).split(".")[-1], int(ids[i]))
                for i in range(n.value)]

    def search_batch(self, Q, k):
        """B queries at once -> (ids [B][k] u32, sims [B][k] f32, n_out [B])."""
        Q = _f32(Q)
        if Q.ndim != 2 or Q.shape[1] != self.data_dim:
            raise HNSWError("data dimension: %d does not match Index" % (Q.shape[-1]), _capi.ERR_DIM_MISMATCH)
        B = Q.shape[0]
        ids = np.full((B, k), 0xFFFFFFFF, dtype=np.uint32)
        sims = np.full((B, k), -np.inf, dtype=np.float32)
        n_out = np.zeros(B, dtype=np.uint32)
        self._check(self._lib.hnsw_search_batch(self._h, _fp(Q), B, self.data_dim, k, _u32p(ids), _fp(sims),
                                                _u32p(n_out)))
        return ids, sims, n_out

    def search_batch_device(self, dQ_ptr, B, k, d_ids_ptr, d_sims_ptr, d_nout_ptr, stream=None):
        """Everything resident in HBM (raw device pointers, e.g. torch .data_ptr())."""
        self._check(self._lib.hnsw_search_batch_device(self._h, dQ_ptr, B, self.data_dim, k, d_ids_ptr, d_sims_ptr,
                                                       d_nout_ptr, stream))

    def pipeline_info(self):
        """lanes / measured overlap of the engine's search pipeline (hnsw_pipeline_info)"""
        p = _capi.Pipeline()
        self._check(self._lib.hnsw_pipeline_info(self._h, C.byref(p)))
        return dict(lanes=int(p.lanes), overlap=int(p.overlap), probe_ratio=round(float(p.probe_ratio), 3),
                    priorities=int(p.priorities), chunk=int(p.chunk), min_batch=int(p.min_batch),
                    hw_queues_env=int(p.hw_queues_env))

    def last_search_was_lean(self):
        """development aid: did the latest search launch use the specialised dim-128 kernel"""
        f = self._lib.hnsw_debug_last_search_path
        f.restype, f.argtypes = C.c_int, [_capi.H, _capi.u32p]
        v = C.c_uint32(0)
        self._check(f(self._h, C.byref(v)))
        return bool(v.value & 1)

    def last_search_was_duo(self):
        """development aid: did the latest search launch use the two-wave form (a walker and a W-keeper wave per query)"""
        f = self._lib.hnsw_debug_last_search_path
        f.restype, f.argtypes = C.c_int, [_capi.H, _capi.u32p]
        v = C.c_uint32(0)
        self._check(f(self._h, C.byref(v)))
        return bool(v.value & 2)

    def lean_blocker(self):
        """development aid: why the specialised dim-128 kernel cannot serve this index ('' = it can)"""
        f = self._lib.hnsw_debug_lean_blocker
        f.restype, f.argtypes = C.c_char_p, [_capi.H]
        return f(self._h).decode()

    def last_search_kernel_ms(self):
        ms = C.c_float(0)
        self._check(self._lib.hnsw_last_search_kernel_ms(self._h, C.byref(ms)))
        return float(ms.value)

    # -- bulk import / export (make_index, src/lib.rs:252-315) ----------------------
    def import_graph(self, g, names=None):
        V = _f32(g["vectors"])
        n = V.shape[0]
        levels = np.ascontiguousarray(g["levels"], dtype=np.uint32)
        L = len(g["row_ptr"])
        rps = [np.ascontiguousarray(r, dtype=np.uint64) for r in g["row_ptr"]]
        cols = [np.ascontiguousarray(c if len(c) else np.zeros(1), dtype=np.uint32) for c in g["col"]]
        rp_arr = (_capi.u64p * L)(*[r.ctypes.data_as(_capi.u64p) for r in rps])
        cl_arr = (_capi.u32p * L)(*[c.ctypes.data_as(_capi.u32p) for c in cols])
        self._check(self._lib.hnsw_import(self._h, n, _fp(V), _u32p(levels), int(g["enterpoint"]), L, rp_arr, cl_arr))
        self._names = list(names) if names is not None else ["node%d" % i for i in range(n)]
        self._ids = {nm: i for i, nm in enumerate(self._names)}

    def export_graph(self, with_vectors=False):
        inf = self.info()
        n = int(inf.allocated_ids)
        levels = np.zeros(max(n, 1), dtype=np.uint32)
        if n:
            self._check(self._lib.hnsw_get_levels(self._h, _u32p(levels)))
        row_ptr, col = [], []
        for l in range(int(inf.max_layer) + 1):
            nnz = C.c_uint64(0)
            self._check(self._lib.hnsw_layer_nnz(self._h, l, C.byref(nnz)))
            rp = np.zeros(n + 1, dtype=np.uint64)
            cl = np.zeros(max(int(nnz.value), 1), dtype=np.uint32)
            self._check(self._lib.hnsw_export_layer(self._h, l, rp.ctypes.data_as(_capi.u64p), _u32p(cl)))
            row_ptr.append(rp)
            col.append(cl[: int(nnz.value)])
        g = dict(levels=levels[:n], enterpoint=int(inf.enterpoint), max_layer=int(inf.max_layer),
                 row_ptr=row_ptr, col=col)
        if with_vectors:
            V = np.zeros((n, self.data_dim), dtype=np.float32)
            for i in range(n):
                self._check(self._lib.hnsw_get_vector(self._h, i, _fp(V[i])))
            g["vectors"] = V
        return g

    # -- snapshot (the module's RDB save/load, src/types.rs:176-284) --------------------------
    def serialize(self):
        """bytes: engine snapshot followed by the host-side name table (JSON)"""
        import json
        nbytes = C.c_uint64(0)
        self._check(self._lib.hnsw_serialize_size(self._h, C.byref(nbytes)))
        buf = (C.c_ubyte * max(int(nbytes.value), 1))()
        wrote = C.c_uint64(0)
        self._check(self._lib.hnsw_serialize(self._h, buf, nbytes.value, C.byref(wrote)))
        names = json.dumps(dict(name=self.name, names=self._names)).encode()
        return int(wrote.value).to_bytes(8, "little") + bytes(buf)[: wrote.value] + names

    @classmethod
    def deserialize(cls, blob, seed=0, device=0):
        import json
        n = int.from_bytes(blob[:8], "little")
        snap, meta = blob[8:8 + n], json.loads(blob[8 + n:].decode())
        lib = _capi.load()
        h = _capi.H()
        cbuf = (C.c_ubyte * max(n, 1)).from_buffer_copy(snap if n else b"\0")
        st = lib.hnsw_deserialize(cbuf, n, seed, device, C.byref(h))
        if st != _capi.OK:
            msg = lib.hnsw_last_error(h).decode() if h else "bad snapshot"
            if h:
                lib.hnsw_destroy(h)
            raise HNSWError(msg, st)
        self = cls.__new__(cls)
        self._lib, self._h, self.name = lib, h, meta["name"]
        i = _capi.Info()
        lib.hnsw_get_info(h, C.byref(i))
        self.data_dim, self.m, self.m_max, self.m_max_0 = int(i.dim), int(i.m), int(i.m_max), int(i.m_max0)
        self.ef_construction = int(i.ef_construction)
        self.level_mult = 1.0 / np.log(float(self.m))
        self._names = meta["names"]
        self._ids = {nm: j for j, nm in enumerate(self._names) if nm is not None}
        return self

    def level_of(self, i):
        """top layer of node id i (the layer set core.rs:596 puts it in); O(1)"""
        l = C.c_uint32(0)
        self._check(self._lib.hnsw_get_level(self._h, int(i), C.byref(l)))
        return int(l.value)

    def _vector(self, i):
        """the stored vector of node id i (hnsw_get_vector)"""
        out = np.zeros(self.data_dim, dtype=np.float32)
        self._check(self._lib.hnsw_get_vector(self._h, int(i), _fp(out)))
        return out

    def neighbors(self, i, layer):
        inf = self.info()
        cap = int(max(inf.stride0, inf.stride_upper))
        out = np.empty(cap, dtype=np.uint32)
        n = C.c_uint32(0)
        self._check(self._lib.hnsw_get_neighbors(self._h, int(i), int(layer), _u32p(out), cap, C.byref(n)))
        return out[: n.value].copy()

    # -- replicas (one-time index distribution) ------------------------------------------
    def replica_view(self):
        """sizes + DEVICE pointers of this index's tables (hnsw_replica_view)"""
        r = _capi.Replica()
        self._check(self._lib.hnsw_replica_view(self._h, C.byref(r)))
        return r

    def replica_prepare(self, scalars):
        """allocate for a source's tables (dict of _capi.Replica.SCALARS) on this empty index -> Replica with pointers"""
        r = _capi.Replica()
        for key in _capi.Replica.SCALARS:
            setattr(r, key, int(scalars[key]))
        self._check(self._lib.hnsw_replica_prepare(self._h, C.byref(r)))
        return r

    def replica_commit(self, r, dead=None, names=None):
        d = None
        if dead is not None:
            d = np.ascontiguousarray(dead, dtype=np.uint8)
        self._check(self._lib.hnsw_replica_commit(self._h, C.byref(r), d.ctypes.data_as(C.POINTER(C.c_uint8)) if d is not None else None))
        n = int(r.n)
        self._names = list(names) if names is not None else ["node%d" % i for i in range(n)]
        self._ids = {nm: i for i, nm in enumerate(self._names) if nm is not None}

    def tombstones(self):
        n = int(self.info().allocated_ids)
        d = np.zeros(max(n, 1), dtype=np.uint8)
        if n:
            self._check(self._lib.hnsw_get_tombstones(self._h, d.ctypes.data_as(C.POINTER(C.c_uint8))))
        return d[:n]

    # -- counters / knobs -----------------------------------------------------------
    def counters(self):
        s, i = _capi.Counters(), _capi.Counters()
        self._check(self._lib.hnsw_get_counters(self._h, C.byref(s), C.byref(i)))
        return s, i

    def reset_counters(self):
        self._check(self._lib.hnsw_reset_counters(self._h))

    def tie_counters(self):
        """decisions that compared equal distances of two different nodes since the last reset_counters() -- where the
        reference's heap order could have chosen differently (hnsw_get_tie_counters): dict(search_events, queries_with_tie,
        insert_events, plans_with_tie); search counts need set_tuning("tie_census", 1)"""
        out = (C.c_uint64 * 4)()
        self._check(self._lib.hnsw_get_tie_counters(self._h, out))
        unknown = int(out[0]) == 2 ** 64 - 1                  # a search ran on a kernel without a census form
        return dict(search_events=None if unknown else int(out[0]), queries_with_tie=None if unknown else int(out[1]),
                    insert_events=int(out[2]), plans_with_tie=int(out[3]))

    def set_tuning(self, key, value):
        self._check(self._lib.hnsw_set_tuning(self._h, key.encode(), int(value)))


def metric_pairs(a, b, device=0):
    """sims[i] = euclidean(a[i], b[i]) computed by the device metric (metrics.rs:14-23)."""
    lib = _capi.load()
    a, b = _f32(a), _f32(b)
    assert a.shape == b.shape and a.ndim == 2
    out = np.zeros(a.shape[0], dtype=np.float32)
    st = lib.hnsw_metric_pairs(device, _fp(a), _fp(b), a.shape[0], a.shape[1], _fp(out))
    if st != _capi.OK:
        raise HNSWError("hnsw_metric_pairs failed (status %d)" % st, st)
    return out
