"""ctypes binding of the CPU parity oracle (oracle/hnsw_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  Nothing under redis_hnsw_amd/ imports this.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libhnsw_oracle.so")
_lib = None


class Counters(C.Structure):
    _fields_ = [("n_dist", C.c_uint64), ("n_ids", C.c_uint64), ("n_expand", C.c_uint64)]


def build(force=False):
    """Compile oracle/libhnsw_oracle.so with gcc (make)."""
    src = os.path.join(_HERE, "hnsw_oracle.c")
    if (force or not os.path.exists(_LIB_PATH)
            or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src)):
        subprocess.check_call(["make", "-s", "-C", _HERE, "-B", "libhnsw_oracle.so"])
    return _LIB_PATH


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIB_PATH):
        build()
    L = C.CDLL(_LIB_PATH)
    fp, u32p, u64p = C.POINTER(C.c_float), C.POINTER(C.c_uint32), C.POINTER(C.c_uint64)
    for name in ("hnsw_oracle_sim_scalar", "hnsw_oracle_sim_avx",
                 "hnsw_oracle_sim_avx_emulated", "hnsw_oracle_euclidean"):
        f = getattr(L, name)
        f.restype = C.c_float
        f.argtypes = [fp, fp, C.c_size_t]
    L.hnsw_oracle_new.restype = C.c_void_p
    L.hnsw_oracle_new.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint64]
    L.hnsw_oracle_free.argtypes = [C.c_void_p]
    L.hnsw_oracle_add.restype = C.c_int64
    L.hnsw_oracle_add.argtypes = [C.c_void_p, fp, C.c_int32, u32p, C.c_uint32, u32p]
    L.hnsw_oracle_delete.restype = C.c_int
    L.hnsw_oracle_delete.argtypes = [C.c_void_p, C.c_uint32, u32p, C.c_uint32, u32p]
    L.hnsw_oracle_delete_std_heap.restype = C.c_int
    L.hnsw_oracle_delete_std_heap.argtypes = [C.c_void_p, C.c_uint32, u32p, C.c_uint32, u32p]
    L.hnsw_oracle_live_count.restype = C.c_uint32
    L.hnsw_oracle_live_count.argtypes = [C.c_void_p]
    L.hnsw_oracle_is_live.restype = C.c_int
    L.hnsw_oracle_is_live.argtypes = [C.c_void_p, C.c_uint32]
    L.hnsw_oracle_search.restype = C.c_uint32
    L.hnsw_oracle_search.argtypes = [C.c_void_p, fp, C.c_uint32, u32p, fp, C.POINTER(Counters)]
    L.hnsw_oracle_search_batch.restype = None
    L.hnsw_oracle_search_batch.argtypes = [C.c_void_p, fp, C.c_uint32, C.c_uint32, u32p, fp,
                                           u32p, C.c_uint32, C.POINTER(Counters)]
    for name in ("hnsw_oracle_node_count", "hnsw_oracle_max_layer"):
        getattr(L, name).restype = C.c_uint32
        getattr(L, name).argtypes = [C.c_void_p]
    L.hnsw_oracle_enterpoint.restype = C.c_int64
    L.hnsw_oracle_enterpoint.argtypes = [C.c_void_p]
    L.hnsw_oracle_level.restype = C.c_uint32
    L.hnsw_oracle_level.argtypes = [C.c_void_p, C.c_uint32]
    L.hnsw_oracle_degree.restype = C.c_uint32
    L.hnsw_oracle_degree.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32]
    L.hnsw_oracle_neighbors.restype = C.c_uint32
    L.hnsw_oracle_neighbors.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, u32p, C.c_uint32]
    L.hnsw_oracle_vector.restype = fp
    L.hnsw_oracle_vector.argtypes = [C.c_void_p, C.c_uint32]
    L.hnsw_oracle_insert_counters.argtypes = [C.c_void_p, C.POINTER(Counters)]
    L.hnsw_oracle_layer_nnz.restype = C.c_uint64
    L.hnsw_oracle_layer_nnz.argtypes = [C.c_void_p, C.c_uint32]
    L.hnsw_oracle_export_layer.argtypes = [C.c_void_p, C.c_uint32, u64p, u32p]
    L.hnsw_oracle_export_levels.argtypes = [C.c_void_p, u32p]
    L.hnsw_oracle_import.restype = C.c_void_p
    L.hnsw_oracle_import.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, fp, u32p,
                                     C.c_int64, C.c_uint32, C.POINTER(u64p), C.POINTER(u32p)]
    L.hnsw_oracle_set_strict_ties.argtypes = [C.c_int]
    L.hnsw_oracle_tie_census.argtypes = [C.c_void_p, fp, C.c_uint32, C.c_uint32, u64p]
    L.hnsw_oracle_tie_census.restype = None
    L.hnsw_oracle_search_std_heap.argtypes = [C.c_void_p, fp, C.c_uint32, u32p, fp]
    L.hnsw_oracle_search_std_heap.restype = C.c_uint32
    L.hnsw_oracle_add_std_heap.restype = C.c_int64
    L.hnsw_oracle_add_std_heap.argtypes = [C.c_void_p, fp, C.c_int32, u64p]
    L.hnsw_oracle_last_add_ties.restype = None
    L.hnsw_oracle_last_add_ties.argtypes = [C.c_void_p, u64p]
    _lib = L
    return L


def set_strict_ties(on):
    """test switch: sim-only comparisons at core.rs:635, :657, :733 (default: the (sim, id) total order)"""
    lib().hnsw_oracle_set_strict_ties(1 if on else 0)


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _u32p(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint32))


def _u64p(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint64))


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def sim_scalar(a, b):
    a, b = _f32(a), _f32(b)
    return float(lib().hnsw_oracle_sim_scalar(_fp(a), _fp(b), a.size))


def sim_avx(a, b):
    a, b = _f32(a), _f32(b)
    assert a.size % 32 == 0
    return float(lib().hnsw_oracle_sim_avx(_fp(a), _fp(b), a.size))


def sim_avx_emulated(a, b):
    a, b = _f32(a), _f32(b)
    assert a.size % 32 == 0
    return float(lib().hnsw_oracle_sim_avx_emulated(_fp(a), _fp(b), a.size))


def euclidean(a, b):
    a, b = _f32(a), _f32(b)
    return float(lib().hnsw_oracle_euclidean(_fp(a), _fp(b), a.size))


class OracleIndex:
    """Dense-id mirror of Index<f32,f32> (core.rs:303-347)."""

    def __init__(self, dim, m=5, ef_construction=200, seed=0, _handle=None):
        self.dim, self.m, self.ef_construction = dim, m, ef_construction
        self._h = _handle if _handle is not None else lib().hnsw_oracle_new(dim, m, ef_construction, seed)

    def close(self):
        if self._h:
            lib().hnsw_oracle_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- build ---------------------------------------------------------------
    def add(self, v, level=-1, want_touched=False):
        v = _f32(v)
        assert v.size == self.dim
        if want_touched:
            cap = 65536
            t = np.empty(cap, dtype=np.uint32)
            n = C.c_uint32(0)
            i = lib().hnsw_oracle_add(self._h, _fp(v), int(level), _u32p(t), cap, C.byref(n))
            assert n.value <= cap
            return int(i), t[: n.value].copy()
        return int(lib().hnsw_oracle_add(self._h, _fp(v), int(level), None, 0, None))

    def add_batch(self, V, levels=None):
        V = _f32(V)
        for i in range(V.shape[0]):
            self.add(V[i], -1 if levels is None else int(levels[i]))

    def add_batch_std_heap(self, V, levels=None):
        """HNSW.NODE.ADD x N in the Rust binary's own tie order (sim-only comparisons on std's BinaryHeap, restated):
        -> (stop-test ties, accept-test ties, select-cut ties, order ties) met by the build"""
        V = _f32(V)
        ties = np.zeros(4, dtype=np.uint64)
        for i in range(V.shape[0]):
            lib().hnsw_oracle_add_std_heap(self._h, _fp(V[i]), -1 if levels is None else int(levels[i]), _u64p(ties))
        return tuple(int(x) for x in ties)

    def add_batch_census(self, V, levels=None):
        """add_batch (the (sim, id) total order) with the tie census of every insert: -> (inserts that met a decision
        tie of any kind, stop-test ties, accept-test ties, select-cut ties, order ties)"""
        V = _f32(V)
        tot = np.zeros(4, dtype=np.uint64)
        one = np.zeros(4, dtype=np.uint64)
        n_ins = 0
        for i in range(V.shape[0]):
            first = self.live_count == 0
            self.add(V[i], -1 if levels is None else int(levels[i]))
            if first:
                continue
            lib().hnsw_oracle_last_add_ties(self._h, _u64p(one))
            tot += one
            n_ins += int(one.sum() != 0)
        return (n_ins,) + tuple(int(x) for x in tot)

    # -- delete (core.rs:414-475) ----------------------------------------------
    def delete(self, i, want_touched=False):
        cap = 65536
        t = np.empty(cap, dtype=np.uint32)
        n = C.c_uint32(0)
        rc = lib().hnsw_oracle_delete(self._h, int(i), _u32p(t), cap, C.byref(n))
        if rc != 0:
            raise KeyError("Node: %r does not exist" % (i,))
        assert n.value <= cap
        return t[: n.value].copy() if want_touched else None

    def delete_std_heap(self, i, want_touched=False):
        """HNSW.NODE.DEL in the Rust binary's own tie order (sim-only comparisons on std's BinaryHeap, restated)"""
        cap = 65536
        t = np.empty(cap, dtype=np.uint32)
        n = C.c_uint32(0)
        rc = lib().hnsw_oracle_delete_std_heap(self._h, int(i), _u32p(t), cap, C.byref(n))
        if rc != 0:
            raise KeyError("Node: %r does not exist" % (i,))
        return t[: n.value].copy() if want_touched else None

    @property
    def live_count(self):
        return int(lib().hnsw_oracle_live_count(self._h))

    def is_live(self, i):
        return bool(lib().hnsw_oracle_is_live(self._h, int(i)))

    # -- search --------------------------------------------------------------
    def search(self, q, k, counters=False):
        q = _f32(q)
        assert q.size == self.dim
        ids = np.empty(k, dtype=np.uint32)
        sims = np.empty(k, dtype=np.float32)
        ct = Counters()
        n = lib().hnsw_oracle_search(self._h, _fp(q), k, _u32p(ids), _fp(sims), C.byref(ct))
        if counters:
            return ids[:n].copy(), sims[:n].copy(), ct
        return ids[:n].copy(), sims[:n].copy()

    def search_batch(self, Q, k, threads=1):
        Q = _f32(Q)
        B = Q.shape[0]
        ids = np.zeros((B, k), dtype=np.uint32)
        sims = np.zeros((B, k), dtype=np.float32)
        n_out = np.zeros(B, dtype=np.uint32)
        ct = Counters()
        lib().hnsw_oracle_search_batch(self._h, _fp(Q), B, k, _u32p(ids), _fp(sims), _u32p(n_out),
                                       threads, C.byref(ct))
        return ids, sims, n_out, ct

    def search_std_heap(self, q, k):
        """HNSW.SEARCH in the Rust binary's own tie order (sim-only comparisons on std's BinaryHeap, restated)"""
        q = _f32(q)
        ids = np.empty(k, dtype=np.uint32)
        sims = np.empty(k, dtype=np.float32)
        n = lib().hnsw_oracle_search_std_heap(self._h, _fp(q), k, _u32p(ids), _fp(sims))
        return ids[:n].copy(), sims[:n].copy()

    def tie_census(self, Q, k):
        """decisions of B searches that met EQUAL similarities of two different nodes (core.rs:635, :657, and the
        k + 1 nearest of the answer): the only places where the reference's sim-only order and the (sim, id) order of
        oracle and engine can part.  dict of counts; one thread."""
        Q = _f32(Q)
        out = np.zeros(6, dtype=np.uint64)
        lib().hnsw_oracle_tie_census(self._h, _fp(Q), Q.shape[0], k, out.ctypes.data_as(C.POINTER(C.c_uint64)))
        return dict(queries=int(out[0]), stop_test_ties=int(out[1]), accept_test_ties=int(out[2]),
                    queries_with_decision_tie=int(out[3]), queries_with_answer_tie=int(out[4]), queries_with_any_tie=int(out[5]))

    # -- introspection ---------------------------------------------------------
    @property
    def node_count(self):
        return int(lib().hnsw_oracle_node_count(self._h))

    @property
    def max_layer(self):
        return int(lib().hnsw_oracle_max_layer(self._h))

    @property
    def enterpoint(self):
        return int(lib().hnsw_oracle_enterpoint(self._h))

    def level(self, i):
        return int(lib().hnsw_oracle_level(self._h, i))

    def neighbors(self, i, layer):
        d = lib().hnsw_oracle_degree(self._h, i, layer)
        out = np.empty(max(d, 1), dtype=np.uint32)
        lib().hnsw_oracle_neighbors(self._h, i, layer, _u32p(out), d)
        return out[:d].copy()

    def insert_counters(self):
        ct = Counters()
        lib().hnsw_oracle_insert_counters(self._h, C.byref(ct))
        return ct

    def export(self):
        """-> dict(vectors [n][dim], levels [n], enterpoint, max_layer, row_ptr [L], col [L])"""
        n = self.node_count
        levels = np.zeros(n, dtype=np.uint32)
        vectors = np.zeros((n, self.dim), dtype=np.float32)
        if n:
            lib().hnsw_oracle_export_levels(self._h, _u32p(levels))
            base = lib().hnsw_oracle_vector(self._h, 0)
            vectors[:] = np.ctypeslib.as_array(base, shape=(n, self.dim))
        row_ptr, col = [], []
        for l in range(self.max_layer + 1):
            nnz = int(lib().hnsw_oracle_layer_nnz(self._h, l))
            rp = np.zeros(n + 1, dtype=np.uint64)
            cl = np.zeros(max(nnz, 1), dtype=np.uint32)
            lib().hnsw_oracle_export_layer(self._h, l, _u64p(rp), _u32p(cl))
            row_ptr.append(rp)
            col.append(cl[:nnz])
        return dict(vectors=vectors, levels=levels, enterpoint=self.enterpoint,
                    max_layer=self.max_layer, row_ptr=row_ptr, col=col)

    @classmethod
    def from_graph(cls, dim, m, ef_construction, g):
        n = int(g["vectors"].shape[0])
        vectors = _f32(g["vectors"])
        levels = np.ascontiguousarray(g["levels"], dtype=np.uint32)
        L = len(g["row_ptr"])
        rps = [np.ascontiguousarray(r, dtype=np.uint64) for r in g["row_ptr"]]
        cls_ = [np.ascontiguousarray(c if len(c) else np.zeros(1), dtype=np.uint32) for c in g["col"]]
        rp_arr = (C.POINTER(C.c_uint64) * L)(*[_u64p(r) for r in rps])
        cl_arr = (C.POINTER(C.c_uint32) * L)(*[_u32p(c) for c in cls_])
        h = lib().hnsw_oracle_import(dim, m, ef_construction, n, _fp(vectors), _u32p(levels),
                                     int(g["enterpoint"]), L, rp_arr, cl_arr)
        return cls(dim, m, ef_construction, _handle=h)


def draw_levels(n, m, seed=7):
    """Levels as SURVEY.md section 8d prescribes: floor(-ln U / ln M) with
    numpy default_rng(seed); node 0 forced to level 0 (core.rs:393-405)."""
    u = np.random.default_rng(seed).random(n)
    u = np.maximum(u, np.finfo(np.float64).tiny)
    lv = np.floor(-np.log(u) * (1.0 / np.log(float(m)))).astype(np.int64)
    lv[0] = 0
    return np.minimum(lv, 31).astype(np.int32)
