"""An INDEPENDENT WITNESS for the CPU oracle: zhao-lang/redis_hnsw's src/hnsw/core.rs transcribed into Python
statement by statement (TEST INFRASTRUCTURE, never shipped, never imported by the product).

The C oracle (oracle/hnsw_oracle.c) restates the reference with dense ids, explicit heaps and a total key order; every
golden vector it writes it also checks.  This file shares nothing with it: objects instead of ids, `heapq` standing in
for `std::collections::BinaryHeap`, Python sets for the `HashSet`s, the reference's control flow line for line
(function and variable names are the reference's; each block cites the lines it follows).  It is run ONLY in the build
container by make_transcribed_golden.py, which writes tests/golden/transcribed_*.npz; the CPU tier then requires the C
oracle to reproduce those files exactly, the GPU tier requires the engine to.

Two places are not a literal statement-for-statement rendering, both without an effect on results:

* the metric (`metrics.rs:48-77`) is evaluated for a whole adjacency row at once (numpy, the same 4 x 8 accumulator
  order, fused multiply-add rounded ONCE) and the loop then takes `sims[j]` where the reference calls `(self.mfunc)` --
  `mfunc` is pure, so hoisting the call changes nothing; the work counters count the reference's call sites
  (`core.rs:550, 621, 652, 711`), not the hoisted evaluations;
* `BinaryHeap` leaves the order of EQUAL keys unspecified (and `core_tests.rs:50-53` does not pin it): `heapq` pops
  equal similarities first-in first-out here.  `ties` counts every pop / peek that had to choose between equal
  similarities of different nodes, every accept test (`core.rs:657`) that met equality with W's furthest, and every
  `select_neighbors` whose cut fell between equal similarities (`core.rs:733`), so a run can be certified tie-free (then no heap implementation could have answered differently).
"""
import heapq
import math

import numpy as np


# ---------------------------------------------------------------------------------------------- metrics.rs
def fma_f32(a, b, c):
    """_mm256_fmadd_ps on float32 arrays: a*b + c rounded once.  The product of two f32 is exact in f64; the f64 sum
    rounds to 53 bits, and rounding THAT to 24 bits is wrong only when the f64 sum sits exactly on an f32 midpoint
    while the exact sum does not: those are moved off the midpoint by the sign of the f64 addition's error term."""
    p = a.astype(np.float64) * b.astype(np.float64)
    c64 = c.astype(np.float64)
    s = p + c64
    bb = s - p
    err = (p - (s - bb)) + (c64 - bb)                      # two-sum: p + c64 == s + err exactly
    low = s.view(np.uint64) & np.uint64(0x1FFFFFFF)
    mid = (low == np.uint64(0x10000000)) & (err != 0.0)
    if mid.any():
        s = np.where(mid, np.nextafter(s, np.where(err > 0, np.inf, -np.inf)), s)
    return s.astype(np.float32)


def sim_func_avx_euc_rows(v1, rows):
    """metrics.rs:48-77 for one `v1` against many `v2` (rows [n][dim], dim % 32 == 0): four accumulators of eight
    lanes, element 32t + 8a + j goes to lane j of accumulator a with one FMA per t (:55-69); (e1 + e2) + (e3 + e4)
    (:71-74); hsum256 = low128 + high128, then movehdup + add, movehl + add_ss (:25-42); negated (:75)."""
    n, dim = rows.shape
    d = (v1[None, :] - rows).astype(np.float32).reshape(n, dim // 32, 4, 8)
    e = np.zeros((n, 4, 8), dtype=np.float32)
    for t in range(dim // 32):
        e = fma_f32(d[:, t], d[:, t], e)
    v = ((e[:, 0] + e[:, 1]).astype(np.float32) + (e[:, 2] + e[:, 3]).astype(np.float32)).astype(np.float32)
    s = (v[:, :4] + v[:, 4:]).astype(np.float32)           # _mm256_castps256_ps128 + _mm256_extractf128_ps
    x = (s[:, 0] + s[:, 1]).astype(np.float32)             # lane 0 of v + movehdup(v)
    y = (s[:, 2] + s[:, 3]).astype(np.float32)             # lane 2, brought down by movehl
    return -((x + y).astype(np.float32))


def sim_func_euc_rows(v1, rows):
    """metrics.rs:79-84: a sequential left fold of (x - y) * (x - y), no FMA"""
    acc = np.zeros(rows.shape[0], dtype=np.float32)
    for i in range(rows.shape[1]):
        d = (v1[i] - rows[:, i]).astype(np.float32)
        acc = (acc + (d * d).astype(np.float32)).astype(np.float32)
    return -acc


def sim_f64_rows(v1, rows):
    """NOT the reference's arithmetic: the same quantity in float64 (what a quick model of the algorithm would use)"""
    d = v1.astype(np.float64)[None, :] - rows.astype(np.float64)
    return -(d * d).sum(axis=1)


# ---------------------------------------------------------------------------------------------- core.rs
class Node:                                              # core.rs:96-100, 179
    __slots__ = ("name", "idx", "neighbors")

    def __init__(self, name, idx):
        self.name = name
        self.idx = idx                                   # row of the data matrix (the reference holds a Vec<T>)
        self.neighbors = []                              # Vec<Vec<NodeWeak>>

    def push_levels(self, level):                        # core.rs:127-135
        while len(self.neighbors) < level + 1:
            self.neighbors.append([])

    def add_neighbor(self, level, neighbor):             # core.rs:137-143
        self.push_levels(level)
        if neighbor not in self.neighbors[level]:
            self.neighbors[level].append(neighbor)

    def rm_neighbor(self, level, neighbor):              # core.rs:145-152 (unwrap: the link must exist)
        index = self.neighbors[level].index(neighbor)
        del self.neighbors[level][index]


class Heap:
    """BinaryHeap<SimPair> (reverse=False: pops the LARGEST sim) / BinaryHeap<Reverse<SimPair>> (reverse=True: the
    smallest).  SimPair orders by sim only (core.rs:292-300); equal similarities pop first-in first-out (heapq with an
    insertion counter) -- ONE of the orders Rust leaves open.  ties = "fifo"."""
    __slots__ = ("h", "reverse", "owner")

    def __init__(self, owner, reverse=False, items=None):
        self.owner, self.reverse = owner, reverse
        self.h = list(items) if items is not None else []

    def clone(self):
        return type(self)(self.owner, self.reverse, self.h)

    def _key(self, sim, node):
        self.owner.seq += 1
        return ((sim if self.reverse else -sim), self.owner.seq, sim, node)

    def push(self, sim, node):
        heapq.heappush(self.h, self._key(sim, node))

    def _tie_check(self):
        # the root's children are the only entries that can equal it without being below another equal entry
        h = self.h
        for c in (1, 2):
            if c < len(h) and h[c][0] == h[0][0] and h[c][3] is not h[0][3]:
                self.owner.ties["heap_order"] += 1
                return

    def pop(self):
        self._tie_check()
        e = heapq.heappop(self.h)
        return e[2], e[3]

    def peek(self):
        self._tie_check()
        return self.h[0][2], self.h[0][3]

    def __len__(self):
        return len(self.h)

    def is_empty(self):
        return not self.h

    def pairs(self):
        return [(e[2], e[3]) for e in self.h]


class TotalHeap(Heap):
    """ties = "total": the order the C oracle and the engine use wherever the reference compares similarities -- larger
    sim first, then the SMALLER id (tie order).  Not Rust's; a total order, so that the result does
    not depend on any heap's internals."""
    __slots__ = ()

    def _key(self, sim, node):
        return ((sim, -node.idx, sim, node) if self.reverse else (-sim, node.idx, sim, node))


class RustHeap:
    """ties = "rust": std::collections::BinaryHeap itself -- the array, sift_up (stops at a parent that is >= the
    element) and pop = swap the last element to the root, sift_down_to_bottom (always towards the greater child, the
    RIGHT one when the two are equal) and sift_up from there -- restated from the standard library's published source
    (library/alloc/src/collections/binary_heap.rs, unchanged in this respect since 2015).  Elements compare by sim only
    (core.rs:292-300), so this reproduces which of two equal similarities the reference's binary pops, evicts or
    iterates first.  `into_vec` / iteration (core.rs:670-673, :786) = the array as it is."""
    __slots__ = ("sims", "nodes", "reverse", "owner")

    def __init__(self, owner, reverse=False, items=None):
        self.owner, self.reverse = owner, reverse
        if items is None:
            self.sims, self.nodes = [], []
        else:
            self.sims, self.nodes = list(items[0]), list(items[1])

    def clone(self):
        return RustHeap(self.owner, self.reverse, (self.sims, self.nodes))

    def _sift_up(self, start, pos):
        sims, nodes, rev = self.sims, self.nodes, self.reverse
        es, en = sims[pos], nodes[pos]
        while pos > start:
            parent = (pos - 1) // 2
            ps = sims[parent]
            if (ps <= es) if rev else (es <= ps):            # hole.element() <= hole.get(parent)
                break
            sims[pos], nodes[pos] = ps, nodes[parent]
            pos = parent
        sims[pos], nodes[pos] = es, en
        return pos

    def push(self, sim, node):
        self.sims.append(sim)
        self.nodes.append(node)
        self._sift_up(0, len(self.sims) - 1)

    def pop(self):
        sims, nodes, rev = self.sims, self.nodes, self.reverse
        s, n = sims.pop(), nodes.pop()
        if sims:
            s, sims[0] = sims[0], s
            n, nodes[0] = nodes[0], n
            # sift_down_to_bottom(0)
            end = len(sims)
            pos = 0
            es, en = sims[0], nodes[0]
            child = 1
            while child <= end - 2 and end >= 2:
                a, b = sims[child], sims[child + 1]
                if (b <= a) if rev else (a <= b):            # hole.get(child) <= hole.get(child + 1)
                    child += 1
                sims[pos], nodes[pos] = sims[child], nodes[child]
                pos = child
                child = 2 * pos + 1
            if child == end - 1:
                sims[pos], nodes[pos] = sims[child], nodes[child]
                pos = child
            sims[pos], nodes[pos] = es, en
            self._sift_up(0, pos)
        return s, n

    def peek(self):
        return self.sims[0], self.nodes[0]

    def __len__(self):
        return len(self.sims)

    def is_empty(self):
        return not self.sims

    def pairs(self):
        return list(zip(self.sims, self.nodes))

    @property
    def h(self):                                         # (sim, node) view for the tie census of select_neighbors
        return [(None, None, s, n) for s, n in zip(self.sims, self.nodes)]


class Index:                                             # core.rs:302-347
    def __init__(self, data_dim, m, ef_construction, data, levels, metric="reference", ties="fifo"):
        self.data_dim = data_dim
        self.m = m
        self.m_max = m                                   # :335
        self.m_max_0 = m * 2                             # :336
        self.ef_construction = ef_construction           # :337
        self.level_mult = 1.0 / math.log(m)              # :338
        self.node_count = 0
        self.max_layer = 0
        self.layers = []
        self.nodes = {}
        self.enterpoint = None
        self.data = data                                 # [N][dim] f32: node i's vector is data[i]
        self.levels = levels                             # the level draws (core.rs:344 seeds from entropy: an input here)
        if metric == "f64":
            self.rows_metric = sim_f64_rows
        elif data_dim % 32 == 0:                         # metrics.rs:18 (an AVX2 host)
            self.rows_metric = sim_func_avx_euc_rows
        else:
            self.rows_metric = sim_func_euc_rows
        self.tie_mode = ties
        self.HeapT = {"fifo": Heap, "total": TotalHeap, "rust": RustHeap}[ties]
        self.seq = 0
        self.ties = {"heap_order": 0, "accept_657": 0, "select_733": 0}
        self.n_dist_insert = 0                           # mfunc calls at core.rs:550, 621, 652, 711 during inserts
        self.n_ids_insert = 0                            # neighbour ids scanned at core.rs:646, 699-700, 547
        self.n_expand_insert = 0
        self.n_dist = self.n_ids = self.n_expand = 0     # the counters of the call in progress

    # the reference compares similarities only (core.rs:635, 657, 733); ties = "total" applies the oracle's / the
    # engine's (sim, smaller id) order there as well
    def nearer(self, asim, anode, bsim, bnode):
        if self.tie_mode == "total":
            return asim > bsim or (asim == bsim and anode.idx < bnode.idx)
        return asim > bsim

    def farther(self, asim, anode, bsim, bnode):
        if self.tie_mode == "total":
            return asim < bsim or (asim == bsim and anode.idx > bnode.idx)
        return asim < bsim

    def mfunc_rows(self, v1, nodes):
        if not nodes:
            return []
        return self.rows_metric(v1, self.data[[x.idx for x in nodes]])

    # ---- core.rs:383-412
    def add_node(self, name, idx):
        if self.node_count == 0:
            node = Node(name, idx)
            self.enterpoint = node
            self.layers.append({node})
            self.nodes[name] = node
            self.node_count += 1
            return
        if name in self.nodes:
            raise KeyError("Node: %r already exists" % name)
        self.n_dist = self.n_ids = self.n_expand = 0
        self.insert(name, idx)
        self.n_dist_insert += self.n_dist
        self.n_ids_insert += self.n_ids
        self.n_expand_insert += self.n_expand

    # ---- core.rs:489-599
    def insert(self, name, idx):
        data = self.data[idx]
        l = int(self.levels[idx])                        # gen_random_level, :495
        l_max = self.max_layer
        self.nodes[name] = Node(name, idx)               # :498-505
        self.node_count += 1
        query = self.nodes[name]
        ep = self.enterpoint
        lc = l_max
        while lc > l:                                    # :511-520
            w = self.search_level(data, ep, 1, lc)
            ep = w.pop()[1]
            if lc == 0:
                break
            lc -= 1
        for lc in range(min(l_max, l), -1, -1):          # :523
            w = self.search_level(data, ep, self.ef_construction, lc)
            neighbors = self.select_neighbors(query, w, self.m, lc, None)        # :525-531
            self.connect_neighbors(query, neighbors, lc)                         # :532
            while not neighbors.is_empty():              # :540-574, nearest first
                _, e = neighbors.pop()
                eneighbors = e.neighbors[lc]
                econn = self.HeapT(self)
                sims = self.mfunc_rows(self.data[e.idx], eneighbors)             # :549-553
                self.n_dist += len(eneighbors)
                self.n_ids += len(eneighbors)
                for j, n in enumerate(eneighbors):
                    econn.push(sims[j], n)
                m_max = self.m_max_0 if lc == 0 else self.m_max                  # :560
                if len(econn) > m_max:
                    enewconn = self.select_neighbors(e, econn, m_max, lc, None)  # :568
                    self.update_node_connections(e, enewconn, econn, lc, None)   # :569
            ep = w.peek()[1]                             # :576
        if l > l_max:                                    # :587-593
            self.max_layer = l
            self.enterpoint = query
            while len(self.layers) < l + 1:
                self.layers.append(set())
        self.layers[l].add(query)                        # :596

    # ---- core.rs:607-675
    def search_level(self, query, ep, ef, level):
        v = {ep}
        qsim = self.mfunc_rows(query, [ep])[0]           # :621
        self.n_dist += 1
        c = self.HeapT(self)
        w = self.HeapT(self, reverse=True)
        c.push(qsim, ep)
        w.push(qsim, ep)
        while not c.is_empty():
            csim, cnode = c.pop()
            fsim, fnode = w.peek()
            if self.farther(csim, cnode, fsim, fnode):   # :635  cpair.sim < fpair.sim
                break
            cnode.push_levels(level)                     # :642
            neighbors = cnode.neighbors[level]
            self.n_expand += 1
            self.n_ids += len(neighbors)
            sims = None
            for j, neighbor in enumerate(neighbors):     # :646, stored order
                if neighbor not in v:
                    v.add(neighbor)
                    fsim, fnode = w.peek()
                    if sims is None:
                        sims = self.mfunc_rows(query, neighbors)
                    esim = sims[j]                       # :652
                    self.n_dist += 1
                    if esim == fsim and len(w) >= ef and fnode is not neighbor:
                        self.ties["accept_657"] += 1     # a total order on (sim, id) could answer differently here
                    if self.nearer(esim, neighbor, fsim, fnode) or len(w) < ef:       # :657  esim > fpair.sim
                        c.push(esim, neighbor)
                        w.push(esim, neighbor)
                        if len(w) > ef:
                            w.pop()
        res = self.HeapT(self)                           # :670-674
        for sim, node in w.pairs():
            res.push(sim, node)
        return res

    # ---- core.rs:677-757 (both flags are true at every call site, :528-529, 565-566, 850-851)
    def select_neighbors(self, query, c, m, lc, ignored_node):
        r = self.HeapT(self)
        w = c.clone()
        wd = self.HeapT(self)
        ccopy = c.clone()                                # :690-696
        v = set()
        while not ccopy.is_empty():
            v.add(ccopy.pop()[1])
        ccopy = c.clone()
        qdata = self.data[query.idx]
        while not ccopy.is_empty():                      # :698-721
            _, enode = ccopy.pop()
            row = enode.neighbors[lc]
            self.n_ids += len(row)
            sims = None
            for j, eneighbor in enumerate(row):
                if eneighbor is query or (ignored_node is not None and eneighbor is ignored_node):
                    continue
                if eneighbor not in v:
                    if sims is None:
                        sims = self.mfunc_rows(qdata, row)
                    self.n_dist += 1                     # :711
                    w.push(sims[j], eneighbor)
                    v.add(eneighbor)
        while not w.is_empty() and len(r) < m:           # :724-738
            esim, enode = w.pop()
            if enode is query or (ignored_node is not None and enode is ignored_node):
                continue
            if r.is_empty() or self.nearer(esim, enode, *r.peek()):   # :733 -- r.peek() is r's NEAREST: only the first passes
                r.push(esim, enode)
            else:
                wd.push(esim, enode)
        while not wd.is_empty() and len(r) < m:          # :741-754
            psim, pnode = wd.pop()
            if pnode is query or (ignored_node is not None and pnode is ignored_node):
                continue
            r.push(psim, pnode)
        # which of two equal similarities makes the cut is the heap's choice: count a tie across the boundary
        if len(r) == m:
            worst = min(e[2] for e in r.h)
            rest = [e for e in wd.h] + [e for e in w.h]
            if any(e[2] == worst and e[3] is not query and e[3] is not ignored_node for e in rest):
                self.ties["select_733"] += 1
        return r

    # ---- core.rs:759-774
    def connect_neighbors(self, query, neighbors, level):
        neighbors = neighbors.clone()
        while not neighbors.is_empty():
            _, n = neighbors.pop()
            query.add_neighbor(level, n)
            n.add_neighbor(level, query)

    # ---- core.rs:776-822
    def update_node_connections(self, node, new_neighbors, old_neighbors, level, ignored_node):
        newconn = new_neighbors.clone()
        rmconn = [n for _, n in old_neighbors.pairs()]   # into_vec(): the heap's array, any order
        updated = {node}
        while not newconn.is_empty():
            _, n = newconn.pop()
            node.add_neighbor(level, n)
            n.add_neighbor(level, node)
            updated.add(n)
            if n in rmconn:
                rmconn.remove(n)
        while rmconn:
            rm = rmconn.pop()
            node.rm_neighbor(level, rm)
            if ignored_node is not None and rm is ignored_node:
                continue
            rm.rm_neighbor(level, node)
            updated.add(rm)
        return updated

    # ---- core.rs:414-475, 824-863
    def delete_node(self, name):
        node = self.nodes.pop(name)
        self.node_count -= 1
        for lc in range(self.max_layer, -1, -1):
            if node in self.layers[lc]:
                self.layers[lc].remove(node)
                break
        for lc in range(len(node.neighbors)):
            self.delete_node_from_neighbors(node, lc)
        if self.enterpoint is node:
            new_ep = None
            for lc in range(self.max_layer, -1, -1):
                if self.layers[lc]:
                    # core.rs:453 takes HashSet::iter().next(): ANY node of the layer.  The oracle and the engine take
                    # the smallest id; so does this transcription, or the three could not be compared.
                    new_ep = min(self.layers[lc], key=lambda x: x.idx)
                    break
                self.layers.pop()
                if self.max_layer > 0:
                    self.max_layer -= 1
            self.enterpoint = new_ep

    def delete_node_from_neighbors(self, node, lc):
        for n in list(node.neighbors[lc]):
            nneighbors = n.neighbors[lc]
            nconn = self.HeapT(self)
            sims = self.mfunc_rows(self.data[n.idx], nneighbors)
            for j, nn in enumerate(nneighbors):
                nconn.push(sims[j], nn)
            m_max = self.m_max_0 if lc == 0 else self.m_max
            nnewconn = self.select_neighbors(n, nconn, m_max, lc, node)
            self.update_node_connections(n, nnewconn, nconn, lc, node)

    # ---- core.rs:477-486, 865-892
    def search_knn(self, query, k):
        if self.node_count == 0:
            return []
        self.n_dist = self.n_ids = self.n_expand = 0
        ep = self.enterpoint
        lc = self.max_layer
        while lc > 0:                                    # :869-874
            w = self.search_level(query, ep, 1, lc)
            ep = w.peek()[1]
            lc -= 1
        w = self.search_level(query, ep, self.ef_construction, 0)               # :876, ef = ef_construction (:485)
        res = []
        while len(res) < k and not w.is_empty():         # :878-890
            sim, node = w.pop()
            res.append((sim, node))
        return res
