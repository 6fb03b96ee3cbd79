// hnsw_std_heap.hpp -- HNSW.NODE.ADD, HNSW.NODE.DEL and HNSW.SEARCH in the REFERENCE BINARY's own tie order (tuning "tie_mode").
//
// The reference orders SimPair by similarity alone (core.rs:292-300) and keeps C, W and every selection in
// std::collections::BinaryHeap: which of two EQUAL similarities pops, is evicted or is linked first is decided by that
// heap's sift procedures.  The engine's kernels use the total order (distance, id) instead -- a whole adjacency row merged
// at once needs one -- and count every place where the two can part (hnsw_get_tie_counters; that the list is complete is
// under test on the CPU: tests/test_golden_cpu.py).  With tie_mode on, an operation the census flags is REDONE here as the
// reference executes it: insert() / search_level() / select_neighbors() / connect_neighbors() / update_node_connections() /
// delete_node_from_neighbors() of core.rs:489-863 statement by statement on a restatement of std's heap (push = sift_up;
// pop = swap the last element into the root, sift_down_to_bottom towards the greater child -- the right one when equal --
// then sift_up; clone / into_vec / into_iter = the array as it is).  One wavefront per operation, its lanes in lockstep with
// the same values; they share work only where the ORDER of the reference's decisions cannot notice: the metric, a row's
// similarities and visited stamps (fetched ahead of the decisions), the independent steps of a sift, and the sort that
// core.rs:724-754 amounts to when no similarity repeats among what it selects (std_select_distinct).  9 ms per insert at
// 20 k x 128, ef 200; ~3 % of the inserts on uniform f32 data are redone.  The result is what the Rust binary links, row for
// row (tests/test_gpu_ties.py against the transcription's "rust"-mode goldens; scripts/fuzz_ties.py against the oracle).
#pragma once
#include "hnsw_insert.hpp"

namespace hnsw {

struct StdPair { float sim; uint32_t id; };
// entries [0, lcap) -- the levels every sift passes through -- live in LDS (at g_std_lds + lo), the rest in HBM
struct StdHeap { StdPair *a; uint32_t lo, n, cap, lcap; int reverse; };   // reverse: BinaryHeap<Reverse<SimPair>> (pops the smallest sim)

// LDS entries of the ten heaps (C, W, res, w, wd, ccopy, nbrs, econn, enew, t): 13 056 x 8 B = 102 KB of dynamic LDS.  w --
// select_neighbors' working heap: the candidates and every fresh neighbour of theirs, ~3 000 entries after a layer-0 search
// with ef 200 -- is the one that must not spill: a push into its HBM part is a memory round trip, 15 of them per row
constexpr uint32_t kStdLdsW = 8192;
constexpr uint32_t kStdLdsTotal = 2048 + 512 + 512 + kStdLdsW + 512 + 512 + 64 + 128 + 64 + 512;
constexpr size_t kStdLdsBytes = (size_t)kStdLdsTotal * 8;
extern __shared__ StdPair g_std_lds[];
__device__ __forceinline__ StdPair std_get(const StdHeap &h, uint32_t i)
{
    if (i < h.lcap) return g_std_lds[h.lo + i];
    return h.a[i];
}
__device__ __forceinline__ void std_set(const StdHeap &h, uint32_t i, StdPair v)
{
    if (i < h.lcap) g_std_lds[h.lo + i] = v;
    else h.a[i] = v;
}

// scratch of one std-order operation (HBM): visited stamps + ten heaps of `hcap` entries
struct StdScratch {
    uint32_t *stamp;          // [node capacity]: epoch stamps (HashSet v, core.rs:614, 692)
    uint32_t *epoch;          // [1]
    StdPair *heaps;           // [10][hcap]
    uint32_t hcap;
    uint32_t *status;         // [1]: 1 = a heap overflowed (nothing can be trusted: the host reports it)
};

__device__ __forceinline__ bool std_le(const StdHeap &h, float x, float y) { return h.reverse ? y <= x : x <= y; }
// The sifts below are std's (library/alloc/src/collections/binary_heap: sift_up, sift_down_to_bottom) with their dependent
// steps taken side by side: WHICH elements are compared and in which order they move is std's, so the array after a push / pop
// is the array std leaves; only the waiting is different -- the ancestors of a hole (sift_up) and the greater-child choice
// of every inner node (sift_down_to_bottom) are independent of the element being sifted, so the lanes fetch them at once.
//
// sift_up(0, pos) of element e already stored at pos: e climbs while it is not <= its parent.  Lane k holds the k-th
// ancestor's parent; the first k whose parent stops e is where e lands; the parents below it move down one step each.
__device__ __forceinline__ void std_sift_up(StdHeap &h, uint32_t pos, StdPair e)
{
    if (pos == 0u) { std_set(h, 0u, e); return; }
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t depth = 31u - (uint32_t)__builtin_clz(pos + 1u);          // ancestors above pos: A[k] = ((pos + 1) >> k) - 1, k = 1 .. depth
    const uint32_t sh = lane < 30u ? lane : 30u;                               // (depth <= 21: the lanes beyond it idle)
    const uint32_t mine = ((pos + 1u) >> sh) - 1u;                             // A[lane]        (lane <= depth)
    const uint32_t par = ((pos + 1u) >> (sh + 1u)) - 1u;                       // A[lane + 1]    (lane <  depth)
    const bool in = lane < depth;
    StdPair u = {0.f, 0u};
    if (in) u = std_get(h, par);
    const uint64_t stop = __ballot(in && std_le(h, e.sim, u.sim));            // hole.element() <= hole.get(parent): break
    const uint32_t t = stop ? (uint32_t)__ffsll((unsigned long long)stop) - 1u : depth;
    if (lane < t) std_set(h, mine, u);                                         // the parents e passed move down
    const uint32_t land = ((pos + 1u) >> t) - 1u;
    std_set(h, land, e);
}
__device__ __forceinline__ void std_push(StdHeap &h, StdPair x, uint32_t *status)
{
    if (h.n >= h.cap) { *status = 1u; return; }
    h.n += 1;
    std_sift_up(h, h.n - 1u, x);
}
// pop of a heap whose array is in LDS and has at most 257 entries (W, the selections, the small working heaps): one pass for
// the greater child of every inner node, the path to the bottom followed in registers, one pass for the elements on it
__device__ __forceinline__ StdPair std_pop_small(StdHeap &h)
{
    const uint32_t lane = threadIdx.x & 63u;
    const StdPair last = g_std_lds[h.lo + --h.n];
    if (h.n == 0u) return last;
    const uint32_t end = h.n;
    const StdPair top = g_std_lds[h.lo];
    // greater child of inner node i (the right one when equal: `if hole.get(child) <= hole.get(child + 1) { child += 1 }`);
    // a node with a left child only (child == end - 1) goes there; 0 = a leaf.  Node 0's children are read with `last`
    // already standing at the root -- the root is never a child, so nothing changes for them.
    uint32_t ch[2];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const uint32_t i = (uint32_t)r * 64u + lane, c = 2u * i + 1u;
        uint32_t pick = 0u;
        if (c + 1u < end) {
            const float a = g_std_lds[h.lo + c].sim, b = g_std_lds[h.lo + c + 1u].sim;
            pick = std_le(h, a, b) ? c + 1u : c;
        } else if (c < end) pick = c;
        ch[r] = pick;
    }
    // the path 0 = P[0] -> P[1] -> ... -> P[len] (a leaf); lane k keeps P[k] and P[k + 1]
    uint32_t pk = 0u, pk1 = 0u, pos = 0u, len = 0u;
    for (;;) {
        const uint32_t c = pos < 64u ? (uint32_t)__builtin_amdgcn_readlane((int)ch[0], (int)pos)
                                     : (pos < 128u ? (uint32_t)__builtin_amdgcn_readlane((int)ch[1], (int)(pos - 64u)) : 0u);
        if (c == 0u) break;
        if (lane == len) { pk = pos; pk1 = c; }
        pos = c;
        len += 1u;
    }
    // sift_down_to_bottom moves old a[P[k + 1]] to P[k] for every k < len and leaves the hole at P[len]; sift_up then lifts
    // `last` from there while it is not <= its parent -- the parents it meets are exactly those moved elements, bottom first
    const bool in = lane < len;
    StdPair v = {0.f, 0u};
    if (in) v = g_std_lds[h.lo + pk1];
    const uint64_t stop = __ballot(in && std_le(h, last.sim, v.sim));          // at P[k + 1], parent (now at P[k]) stops it
    const int s_ = stop ? 63 - __builtin_clzll((unsigned long long)stop) : -1;  // the deepest one is met first
    if (in && (int)lane <= s_) g_std_lds[h.lo + pk] = v;                       // (the elements below it are back where they were)
    const uint32_t land = s_ < 0 ? 0u : (uint32_t)__builtin_amdgcn_readlane((int)pk1, s_);
    g_std_lds[h.lo + land] = last;
    return top;
}
__device__ __forceinline__ StdPair std_pop(StdHeap &h)
{
    if (h.n <= 257u && h.n <= h.lcap) return std_pop_small(h);
    StdPair item = std_get(h, --h.n);
    if (h.n) {
        const StdPair t = std_get(h, 0); item = t;
        const StdPair e = std_get(h, h.n);                                // (the former last element: sifted from the root)
        const uint32_t end = h.n;
        uint32_t pos = 0, child = 1;
        while (end >= 2 && child <= end - 2) {                        // sift_down_to_bottom
            const StdPair a = std_get(h, child), b = std_get(h, child + 1);
            const bool right = std_le(h, a.sim, b.sim);
            std_set(h, pos, right ? b : a);
            child += right ? 1u : 0u;
            pos = child;
            child = 2 * pos + 1;
        }
        if (child == end - 1) { std_set(h, pos, std_get(h, child)); pos = child; }
        std_sift_up(h, pos, e);
    }
    return item;
}
__device__ __forceinline__ void std_clone(StdHeap &dst, const StdHeap &src, uint32_t *status)
{
    if (src.n > dst.cap) { *status = 1u; dst.n = 0; return; }
    for (uint32_t i = 0; i < src.n; ++i) std_set(dst, i, std_get(src, i));
    dst.n = src.n; dst.reverse = src.reverse;
}

// All 64 lanes of the wavefront run the operation in lockstep with the same values (nothing but std_sim depends on the lane:
// every load, store and branch is uniform, a store writes the same word 64 times over); std_sim is the one place where the
// lanes share work.  metrics.rs:14-84 bit for bit: the AVX2 order iff dim % 32 == 0 -- lane s < 32 is accumulator s / 8, SIMD
// lane s % 8 (one FMA per 32-block), then (e1+e2)+(e3+e4), low128+high128, (s0+s1)+(s2+s3) as butterfly steps (additions
// commute, so every lane ends with the same bits) -- else the scalar left fold: squares side by side, the additions one by one.
__device__ __forceinline__ float std_sim(const float *a, const float *b, uint32_t dim)
{
    const uint32_t lane = threadIdx.x & 63u;
    float r;
    if (dim % 32u == 0u) {
        const uint32_t s = lane & 31u;
        float e = 0.f;
        for (uint32_t i = 0; i < dim; i += 32) {
            const float d = __fsub_rn(a[i + s], b[i + s]);
            e = __fmaf_rn(d, d, e);
        }
        e = __fadd_rn(e, __shfl_xor(e, 8));
        e = __fadd_rn(e, __shfl_xor(e, 16));
        e = __fadd_rn(e, __shfl_xor(e, 4));
        e = __fadd_rn(e, __shfl_xor(e, 1));
        e = __fadd_rn(e, __shfl_xor(e, 2));
        r = -e;
    } else {
        float acc = 0.f;
        for (uint32_t base = 0; base < dim; base += 64) {
            const uint32_t i = base + lane;
            float sq = 0.f;
            if (i < dim) { const float d = __fsub_rn(a[i], b[i]); sq = __fmul_rn(d, d); }
            const uint32_t cnt = dim - base < 64u ? dim - base : 64u;
            for (uint32_t j = 0; j < cnt; ++j) acc = __fadd_rn(acc, __int_as_float(__builtin_amdgcn_readlane(__float_as_int(sq), (int)j)));
        }
        r = -acc;
    }
    return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(r)));
}

struct StdCtx {
    GraphView g;
    StdScratch sc;
    StdHeap C, W, res, w, wd, ccopy, nbrs, econn, enew, t;
    uint32_t epoch;
    unsigned long long n_dist, n_ids, n_expand;
    uint32_t ovf;                 // a heap overflowed (written to sc.status when the operation ends)
    uint32_t *touched, touched_cap, nt;
#ifdef HNSW_STD_PROF
    unsigned long long tp[8], t0;
#endif
};
#ifdef HNSW_STD_PROF
#define STD_T0(x) (x).t0 = __builtin_amdgcn_s_memrealtime()
#define STD_MARK(x, i) do { const unsigned long long n_ = __builtin_amdgcn_s_memrealtime(); (x).tp[i] += n_ - (x).t0; (x).t0 = n_; } while (0)
#else
#define STD_T0(x)
#define STD_MARK(x, i)
#endif
__device__ __forceinline__ void std_ctx_init(StdCtx &x, const GraphView &g, const StdScratch &sc)
{
    x.g = g; x.sc = sc;
    // (each heap named: an array of pointers to them would pin the whole context in scratch memory)
    uint32_t off = 0, slot = 0;
    auto init = [&](StdHeap &h, uint32_t lcap) {
        h.a = sc.heaps + (size_t)slot * sc.hcap; h.n = 0; h.cap = sc.hcap; h.reverse = 0;
        h.lo = off; h.lcap = lcap; off += lcap; slot += 1;
    };
    init(x.C, 2048); init(x.W, 512); init(x.res, 512); init(x.w, kStdLdsW); init(x.wd, 512);
    init(x.ccopy, 512); init(x.nbrs, 64); init(x.econn, 128); init(x.enew, 64); init(x.t, 512);
    x.epoch = *sc.epoch;
    x.n_dist = x.n_ids = x.n_expand = 0;
    x.ovf = 0;
    x.touched = nullptr; x.touched_cap = 0; x.nt = 0;
}
__device__ __forceinline__ void std_visited_reset(StdCtx &x, uint32_t n)
{
    if (++x.epoch == 0u) { for (uint32_t i = 0; i < n; ++i) x.sc.stamp[i] = 0u; x.epoch = 1u; }
}
__device__ __forceinline__ bool std_test_and_set(StdCtx &x, uint32_t id)
{
    if (x.sc.stamp[id] == x.epoch) return true;
    x.sc.stamp[id] = x.epoch;
    return false;
}
__device__ __forceinline__ const float *std_vec(const StdCtx &x, uint32_t id) { return x.g.vec + (size_t)id * x.g.dim; }
// Similarities of one adjacency-row chunk against vector a: lane i holds the id of entry i, `mask` says which entries are
// wanted, lane i returns entry i's similarity.  The values are those of std_sim; computing them ahead of the heap decisions
// changes nothing (a similarity does not depend on the heaps) and takes the row's loads off the dependent chain: two entries
// per pass in the AVX order (one per half of the wavefront; the butterfly steps stay inside a half).
__device__ __forceinline__ float std_sims_row(const StdCtx &x, const float *a, uint32_t myid, uint64_t mask)
{
    const uint32_t lane = threadIdx.x & 63u, dim = x.g.dim;
    float mine = 0.f;
    if (dim % 32u == 0u) {
        const uint32_t s = lane & 31u, half = lane >> 5;
        constexpr int U = 4;                                          // pairs per pass: 8 vectors' loads in flight (the pass is one memory round trip)
        while (mask) {
            uint32_t bb[U][2];
            const float *bp[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                // (a pass that runs out of entries repeats its last one: same bits, written to the same lane again)
                uint32_t b0 = u ? bb[u - 1][1] : 0u, b1;
                if (mask) { b0 = (uint32_t)__ffsll((unsigned long long)mask) - 1u; mask &= mask - 1; }
                b1 = b0;
                if (mask) { b1 = (uint32_t)__ffsll((unsigned long long)mask) - 1u; mask &= mask - 1; }
                bb[u][0] = b0; bb[u][1] = b1;
                const uint32_t id0 = (uint32_t)__builtin_amdgcn_readlane((int)myid, (int)b0), id1 = (uint32_t)__builtin_amdgcn_readlane((int)myid, (int)b1);
                bp[u] = x.g.vec + (size_t)(half ? id1 : id0) * dim + s;
            }
            float e[U];
#pragma unroll
            for (int u = 0; u < U; ++u) e[u] = 0.f;
            for (uint32_t i = 0; i < dim; i += 32) {
                const float av = a[i + s];
                float bv[U];
#pragma unroll
                for (int u = 0; u < U; ++u) bv[u] = bp[u][i];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const float d = __fsub_rn(av, bv[u]);
                    e[u] = __fmaf_rn(d, d, e[u]);
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                float t = e[u];
                t = __fadd_rn(t, __shfl_xor(t, 8));
                t = __fadd_rn(t, __shfl_xor(t, 16));
                t = __fadd_rn(t, __shfl_xor(t, 4));
                t = __fadd_rn(t, __shfl_xor(t, 1));
                t = __fadd_rn(t, __shfl_xor(t, 2));
                const float r0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(t), 0));
                const float r1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(t), 32));
                if (lane == bb[u][0]) mine = -r0;
                if (lane == bb[u][1]) mine = -r1;
            }
        }
    } else {
        while (mask) {
            const uint32_t b0 = (uint32_t)__ffsll((unsigned long long)mask) - 1u;
            mask &= mask - 1;
            const uint32_t id0 = (uint32_t)__builtin_amdgcn_readlane((int)myid, (int)b0);
            const float r = std_sim(a, x.g.vec + (size_t)id0 * dim, dim);
            if (lane == b0) mine = r;
        }
    }
    return mine;
}
// rows above a node's top level behave as empty (push_levels, core.rs:127-135, 642)
__device__ __forceinline__ const uint32_t *std_row(const StdCtx &x, uint32_t id, uint32_t lc, uint32_t &cnt)
{
    if (lc > x.g.levels[id]) { cnt = 0; return nullptr; }
    const uint32_t *r = row_ptr(x.g, id, lc);
    const uint32_t stride = lc ? x.g.strideU : x.g.stride0;
    cnt = r[0] > stride - 1 ? stride - 1 : r[0];
    return r + 1;
}
__device__ __forceinline__ void std_touch(StdCtx &x, uint32_t id)
{
    if (x.touched && x.nt < x.touched_cap) x.touched[x.nt] = id;
    x.nt += 1;
}
// Row scans below: the lanes compare a chunk of the row at once (one load latency instead of one per entry); a shift moves
// a chunk per step, every lane's load before any lane's store (one instruction each, in order).
// position of the first entry of r[1 .. 1 + cnt) equal to nb, or kEmpty
__device__ __forceinline__ uint32_t std_row_find(const uint32_t *r, uint32_t cnt, uint32_t nb)
{
    const uint32_t lane = threadIdx.x & 63u;
    for (uint32_t base = 0; base < cnt; base += 64u) {
        const uint64_t hit = __ballot(base + lane < cnt && r[1 + base + lane] == nb);
        if (hit) return base + (uint32_t)__ffsll((unsigned long long)hit) - 1u;
    }
    return kEmpty;
}
// core.rs:137-143 add_neighbor: push iff not already present
__device__ __forceinline__ void std_add_neighbor(StdCtx &x, uint32_t id, uint32_t lc, uint32_t nb)
{
    uint32_t *r = row_ptr(x.g, id, lc);
    const uint32_t stride = lc ? x.g.strideU : x.g.stride0;
    const uint32_t cnt = r[0];
    if (std_row_find(r, cnt, nb) != kEmpty) return;
    if (cnt + 1 > stride - 1) { atomicOr(&x.g.hdr->status, ST_ROW_OVERFLOW); return; }
    r[1 + cnt] = nb; r[0] = cnt + 1;
    if ((threadIdx.x & 63u) == 0u) atomicMax(lc ? &x.g.hdr->max_degU : &x.g.hdr->max_deg0, cnt + 1);
}
// core.rs:145-152 rm_neighbor: position().unwrap() then Vec::remove
__device__ __forceinline__ void std_rm_neighbor(StdCtx &x, uint32_t id, uint32_t lc, uint32_t nb)
{
    uint32_t *r = row_ptr(x.g, id, lc);
    const uint32_t cnt = r[0], lane = threadIdx.x & 63u;
    const uint32_t pos = std_row_find(r, cnt, nb);
    if (pos == kEmpty) { atomicOr(&x.g.hdr->status, ST_ASYMMETRIC); return; }    // the reference panics here (core.rs:150)
    for (uint32_t b = pos; b + 1 < cnt; b += 64u) {
        const uint32_t j = b + lane;
        const bool in = j + 1 < cnt;
        const uint32_t v = in ? r[2 + j] : 0u;
        if (in) r[1 + j] = v;
    }
    r[0] = cnt - 1;
}

// core.rs:607-675; leaves the result heap (:670-674) in x.res
__device__ __forceinline__ void std_search_level(StdCtx &x, const float *query, uint32_t ep, uint32_t ef, uint32_t level)
{
    std_visited_reset(x, x.g.hdr->node_count);
    std_test_and_set(x, ep);
    const StdPair qpair = {std_sim(query, std_vec(x, ep), x.g.dim), ep};
    x.n_dist += 1;
    x.C.n = x.W.n = x.res.n = 0; x.C.reverse = 0; x.W.reverse = 1; x.res.reverse = 0;
    std_push(x.C, qpair, &x.ovf); std_push(x.W, qpair, &x.ovf);
    while (x.C.n) {
        const StdPair c = std_pop(x.C);
        StdPair f = std_get(x.W, 0);
        if (c.sim < f.sim) break;                                     // :635
        x.n_expand += 1;
        uint32_t cnt;
        const uint32_t *nb = std_row(x, c.id, level, cnt);
        const uint32_t lane = threadIdx.x & 63u;
        for (uint32_t base = 0; base < cnt; base += 64u) {            // :646 stored order, a chunk of the row at a time
            const uint32_t nn = cnt - base < 64u ? cnt - base : 64u;
            const bool in = lane < nn;
            const uint32_t myid = in ? nb[base + lane] : 0u;
            const uint64_t unvisited = __ballot(in && x.sc.stamp[myid] != x.epoch);
            const float mysim = std_sims_row(x, query, myid, unvisited);
            x.n_ids += nn;
            for (uint64_t todo = unvisited; todo; todo &= todo - 1) {
                const uint32_t i = (uint32_t)__ffsll((unsigned long long)todo) - 1u;
                const uint32_t e = (uint32_t)__builtin_amdgcn_readlane((int)myid, (int)i);
                if (__ballot(in && myid == e) & ((1ull << i) - 1ull)) continue;    // the same id earlier in this row: visited by now
                x.sc.stamp[e] = x.epoch;
                f = std_get(x.W, 0);
                const StdPair e2 = {__int_as_float(__builtin_amdgcn_readlane(__float_as_int(mysim), (int)i)), e};
                x.n_dist += 1;
                if (e2.sim > f.sim || x.W.n < ef) {                   // :657
                    std_push(x.C, e2, &x.ovf); std_push(x.W, e2, &x.ovf);
                    if (x.W.n > ef) std_pop(x.W);
                }
            }
        }
        if (x.ovf) return;
    }
    for (uint32_t i = 0; i < x.W.n; ++i) std_push(x.res, std_get(x.W, i), &x.ovf);   // :670-674: into_iter order, pushed one by one
}

// core.rs:724-754, the part of select_neighbors after the candidates' neighbourhood is gathered in w.  As written it is a
// sort: w pops nearest first; r takes the first element and then refuses everything (`e.sim > r.peek().sim` cannot hold for
// what pops later), so the loop DRAINS w into wd -- ~3 000 pops after a layer-0 search with ef 200 -- and the second loop pops
// wd back until r holds m.  Pushed in descending order neither heap sifts anything: r's array is the pop order.  Which
// elements pop first is the heaps' business only among EQUAL similarities; when the m nearest eligible entries of w and the
// one after them are pairwise different, r = those m, nearest first, whatever the heaps do.  That case is answered here by m
// rounds of "the largest similarity below the previous one, and it occurs once" over w's array (all lanes, LDS); a
// repeated similarity returns false with nothing touched, and the loops below decide as std decides.
__device__ __forceinline__ bool std_select_distinct(StdCtx &x, uint32_t query, uint32_t ignored, uint32_t m, StdHeap &r)
{
    const StdHeap &w = x.w;
    if (m < 2u || w.n > w.lcap || m > r.lcap) return false;
    const uint32_t lane = threadIdx.x & 63u;
    float prev = 0.f;
    uint32_t found = 0;
    while (found < m) {
        float best = 0.f;
        uint32_t bid = kEmpty, nbest = 0;                             // nbest: entries of this lane's share at `best`
        for (uint32_t i = lane; i < w.n; i += 64u) {
            const StdPair p = g_std_lds[w.lo + i];
            if (p.id == query || p.id == ignored || (found && !(p.sim < prev))) continue;
            if (!nbest || p.sim > best) { best = p.sim; bid = p.id; nbest = 1u; }
            else if (p.sim == best) nbest += 1u;
        }
        const uint64_t have = __ballot(nbest != 0u);
        if (!have) break;                                             // fewer than m eligible entries
        float wmax = best;
        bool any = nbest != 0u;
        for (int d = 32; d >= 1; d >>= 1) {
            const float o = __shfl_xor(wmax, d);
            const bool oa = __shfl_xor((int)any, d) != 0;
            if (oa && (!any || o > wmax)) wmax = o;
            any = any || oa;
        }
        const uint64_t at = __ballot(nbest != 0u && best == wmax);
        uint32_t total = (nbest != 0u && best == wmax) ? nbest : 0u;
        for (int d = 32; d >= 1; d >>= 1) total += (uint32_t)__shfl_xor((int)total, d);
        if (__builtin_amdgcn_readfirstlane(total) != 1u) return false;    // a repeated similarity: the heaps decide
        const int wl = __ffsll((unsigned long long)at) - 1;
        const StdPair win = {__int_as_float(__builtin_amdgcn_readlane(__float_as_int(wmax), wl)), (uint32_t)__builtin_amdgcn_readlane((int)bid, wl)};
        g_std_lds[r.lo + found] = win;
        found += 1u;
        prev = win.sim;
    }
    r.n = found; r.reverse = 0;
    return true;
}

// core.rs:677-757 (extend_candidates = keep_pruned_connections = true at every call site); result in r
__device__ __forceinline__ void std_select_neighbors(StdCtx &x, uint32_t query, const StdHeap &c, uint32_t m, uint32_t lc, uint32_t ignored, StdHeap &r)
{
    r.n = 0; r.reverse = 0;
    std_clone(x.w, c, &x.ovf);                                   // :685
    x.wd.n = 0; x.wd.reverse = 0;
    std_visited_reset(x, x.g.hdr->node_count);                        // :692
    for (uint32_t i = 0; i < c.n; ++i) x.sc.stamp[std_get(c, i).id] = x.epoch;   // :690-696 (a set)
    std_clone(x.ccopy, c, &x.ovf);                               // :698
    const float *qv = std_vec(x, query);
    while (x.ccopy.n) {
        const StdPair e = std_pop(x.ccopy);
        uint32_t cnt;
        const uint32_t *nb = std_row(x, e.id, lc, cnt);
        const uint32_t lane = threadIdx.x & 63u;
        for (uint32_t base = 0; base < cnt; base += 64u) {            // stored order, a chunk of the row at a time
            const uint32_t nn = cnt - base < 64u ? cnt - base : 64u;
            const bool in = lane < nn;
            const uint32_t myid = in ? nb[base + lane] : 0u;
            const uint64_t fresh = __ballot(in && myid != query && myid != ignored && x.sc.stamp[myid] != x.epoch);   // :704-710
            const float mysim = std_sims_row(x, qv, myid, fresh);
            x.n_ids += nn;
            for (uint64_t todo = fresh; todo; todo &= todo - 1) {
                const uint32_t i = (uint32_t)__ffsll((unsigned long long)todo) - 1u;
                const uint32_t en = (uint32_t)__builtin_amdgcn_readlane((int)myid, (int)i);
                if (__ballot(in && myid == en) & ((1ull << i) - 1ull)) continue;   // the same id earlier in this row: in the set by now
                const StdPair p = {__int_as_float(__builtin_amdgcn_readlane(__float_as_int(mysim), (int)i)), en};
                x.n_dist += 1;
                std_push(x.w, p, &x.ovf);                             // :717
                x.sc.stamp[en] = x.epoch;                             // :718
            }
        }
        if (x.ovf) return;
    }
    if (std_select_distinct(x, query, ignored, m, r)) return;         // :724-754 when no similarity among the selected (or at the cut) repeats
    while (x.w.n && r.n < m) {                                        // :724-738
        const StdPair e = std_pop(x.w);
        if (e.id == query || e.id == ignored) continue;
        if (r.n == 0 || e.sim > std_get(r, 0).sim) std_push(r, e, &x.ovf);   // :733
        else std_push(x.wd, e, &x.ovf);
    }
    while (x.wd.n && r.n < m) {                                       // :741-754
        const StdPair p = std_pop(x.wd);
        if (p.id == query || p.id == ignored) continue;
        std_push(r, p, &x.ovf);
    }
}

// core.rs:776-822
__device__ __forceinline__ void std_update_node_connections(StdCtx &x, uint32_t node, const StdHeap &new_neighbors, const StdHeap &old_neighbors, uint32_t level,
                                                   uint32_t ignored)
{
    std_clone(x.t, new_neighbors, &x.ovf);                      // :784
    // :785 into_vec: the old heap's array as it is -- its ids kept in the LDS part of x.wd (free here)
    uint32_t n_rm = old_neighbors.n;
    StdPair *rm = g_std_lds + x.wd.lo;
    if (n_rm > x.wd.lcap) { x.ovf = 1u; return; }
    const uint32_t lane = threadIdx.x & 63u;
    for (uint32_t i = 0; i < n_rm; ++i) rm[i] = std_get(old_neighbors, i);
    std_touch(x, node);                                               // :787
    while (x.t.n) {                                                   // :790
        const StdPair np = std_pop(x.t);
        std_add_neighbor(x, node, level, np.id);                      // :793
        std_add_neighbor(x, np.id, level, node);                      // :794-795
        std_touch(x, np.id);                                          // :796
        uint32_t pos = kEmpty;                                        // :799-801 position() + Vec::remove
        for (uint32_t base = 0; base < n_rm && pos == kEmpty; base += 64u) {
            const uint64_t hit = __ballot(base + lane < n_rm && rm[base + lane].id == np.id);
            if (hit) pos = base + (uint32_t)__ffsll((unsigned long long)hit) - 1u;
        }
        if (pos != kEmpty) {
            for (uint32_t b = pos; b + 1 < n_rm; b += 64u) {
                const uint32_t j = b + lane;
                const bool in = j + 1 < n_rm;
                StdPair v = {0.f, 0u};
                if (in) v = rm[j + 1];
                if (in) rm[j] = v;
            }
            n_rm--;
        }
    }
    while (n_rm) {                                                    // :805
        const StdPair rp = rm[--n_rm];
        std_rm_neighbor(x, node, level, rp.id);                       // :808
        if (rp.id == ignored) continue;                               // :810-813
        std_rm_neighbor(x, rp.id, level, node);                       // :815
        std_touch(x, rp.id);                                          // :816
    }
}

// core.rs:489-599 for node `query` (vector, level and empty rows already in place)
__device__ __forceinline__ void std_insert(StdCtx &x, uint32_t query, uint32_t mlinks, uint32_t ef)
{
    const uint32_t l = x.g.levels[query];
    const uint32_t l_max = x.g.hdr->max_layer;                        // :496
    const float *qv = std_vec(x, query);
    uint32_t ep = (uint32_t)x.g.hdr->enterpoint;                      // :508
    uint32_t lc = l_max;
    while (lc > l) {                                                  // :511-520
        std_search_level(x, qv, ep, 1, lc);
        if (x.ovf) return;
        ep = std_get(x.res, 0).id;                                           // :514 w.pop(): the root
        if (lc == 0) break;
        lc--;
    }
    const uint32_t top = l_max < l ? l_max : l;
    for (uint32_t lcc = top + 1; lcc-- > 0;) {                        // :523
        STD_T0(x);
        std_search_level(x, qv, ep, ef, lcc);                         // :524
        if (x.ovf) return;
        STD_MARK(x, 0);
        std_select_neighbors(x, query, x.res, mlinks, lcc, kEmpty, x.nbrs);   // :525-531
        if (x.ovf) return;
        STD_MARK(x, 1);
        std_clone(x.t, x.nbrs, &x.ovf);                          // :532 connect_neighbors (:759-774)
        while (x.t.n) { const StdPair n = std_pop(x.t); std_add_neighbor(x, query, lcc, n.id); std_add_neighbor(x, n.id, lcc, query); }
        for (uint32_t i = 0; i < x.nbrs.n; ++i) std_touch(x, std_get(x.nbrs, i).id);   // :535-537
        const uint32_t ep_next = std_get(x.res, 0).id;                       // :576 w.peek() (res is not touched below)
        STD_MARK(x, 2);
        while (x.nbrs.n) {                                            // :540
            STD_T0(x);
            const StdPair e = std_pop(x.nbrs);
            x.econn.n = 0; x.econn.reverse = 0;                       // :544-558
            uint32_t cnt;
            const uint32_t *er = std_row(x, e.id, lcc, cnt);
            const float *ev = std_vec(x, e.id);
            const uint32_t lane = threadIdx.x & 63u;
            for (uint32_t base = 0; base < cnt; base += 64u) {
                const uint32_t nn = cnt - base < 64u ? cnt - base : 64u;
                const bool in = lane < nn;
                const uint32_t myid = in ? er[base + lane] : 0u;
                const float mysim = std_sims_row(x, ev, myid, __ballot(in));                 // :550
                for (uint32_t i = 0; i < nn; ++i) {
                    const StdPair p = {__int_as_float(__builtin_amdgcn_readlane(__float_as_int(mysim), (int)i)),
                                       (uint32_t)__builtin_amdgcn_readlane((int)myid, (int)i)};
                    x.n_dist += 1; x.n_ids += 1;
                    std_push(x.econn, p, &x.ovf);
                }
            }
            const uint32_t m_max = lcc == 0 ? 2 * mlinks : mlinks;    // :560
            STD_MARK(x, 3);
            if (x.econn.n > m_max) {                                  // :561
                std_select_neighbors(x, e.id, x.econn, m_max, lcc, kEmpty, x.enew);   // :568
                if (x.ovf) return;
                STD_MARK(x, 4);
                std_update_node_connections(x, e.id, x.enew, x.econn, lcc, kEmpty);  // :569
                STD_MARK(x, 5);
            }
            if (x.ovf) return;
        }
        ep = ep_next;
    }
    if (l > l_max) { x.g.hdr->max_layer = l; x.g.hdr->enterpoint = (int32_t)query; }   // :587-593
    x.g.hdr->node_count = query + 1;
}

// HNSW.NODE.ADD in the reference binary's tie order: one wavefront, its lanes in lockstep (see std_sim)
__global__ __launch_bounds__(64) void k_insert_std_heap(GraphView g, StdScratch sc, uint32_t id, uint32_t mlinks, uint32_t ef, uint32_t *touched,
                                                        uint32_t touched_cap)
{
    if (blockIdx.x != 0) return;
    StdCtx x;
    std_ctx_init(x, g, sc);
    x.touched = touched; x.touched_cap = touched_cap;
#ifdef HNSW_STD_PROF
    for (int i = 0; i < 8; ++i) x.tp[i] = 0;
#endif
    std_insert(x, id, mlinks, ef);
#ifdef HNSW_STD_PROF
    if (threadIdx.x == 0)
        printf("STDPROF id %u search %llu select %llu connect %llu econn %llu shr_select %llu shr_update %llu dist %llu\n", id, x.tp[0], x.tp[1], x.tp[2],
               x.tp[3], x.tp[4], x.tp[5], x.n_dist);
#endif
    *sc.epoch = x.epoch;
    if (x.ovf) *sc.status = 1u;
    if (touched) g.hdr->n_touched = x.nt;
    if (threadIdx.x != 0) return;
    atomicAdd(&g.hdr->ctr_insert[0], x.n_dist);
    atomicAdd(&g.hdr->ctr_insert[1], x.n_ids);
    atomicAdd(&g.hdr->ctr_insert[2], x.n_expand);
}

// HNSW.NODE.DEL in the reference binary's tie order (core.rs:414-475 -> delete_node_from_neighbors, :824-863): every neighbour
// n of the node, layer by layer in stored order, gathers its connections in a heap (:832-844), re-selects with the node
// ignored (:853) and is rewired (:856).  The host keeps the tombstone and re-elects the enterpoint as for any delete.
__global__ __launch_bounds__(64) void k_delete_std_heap(GraphView g, StdScratch sc, uint32_t id, uint32_t mlinks, uint32_t *touched,
                                                        uint32_t touched_cap)
{
    if (blockIdx.x != 0) return;
    StdCtx x;
    std_ctx_init(x, g, sc);
    x.touched = touched; x.touched_cap = touched_cap;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t l = g.levels[id];
    for (uint32_t lc = 0; lc <= l && !x.ovf; ++lc) {                  // :434-439
        uint32_t dcnt;
        const uint32_t *drow = std_row(x, id, lc, dcnt);              // not modified while it is walked (the node is ignored everywhere)
        for (uint32_t kk = 0; kk < dcnt && !x.ovf; ++kk) {            // :829 stored order
            const uint32_t n = drow[kk];
            x.econn.n = 0; x.econn.reverse = 0;                       // :832-844
            uint32_t cnt;
            const uint32_t *er = std_row(x, n, lc, cnt);
            const float *nv = std_vec(x, n);
            for (uint32_t base = 0; base < cnt; base += 64u) {
                const uint32_t nn = cnt - base < 64u ? cnt - base : 64u;
                const bool in = lane < nn;
                const uint32_t myid = in ? er[base + lane] : 0u;
                const float mysim = std_sims_row(x, nv, myid, __ballot(in));
                for (uint32_t i = 0; i < nn; ++i) {
                    const StdPair p = {__int_as_float(__builtin_amdgcn_readlane(__float_as_int(mysim), (int)i)),
                                       (uint32_t)__builtin_amdgcn_readlane((int)myid, (int)i)};
                    x.n_dist += 1; x.n_ids += 1;
                    std_push(x.econn, p, &x.ovf);
                }
            }
            const uint32_t m_max = lc == 0 ? 2 * mlinks : mlinks;     // :846
            std_select_neighbors(x, n, x.econn, m_max, lc, id, x.enew);   // :853
            if (x.ovf) break;
            std_touch(x, n);                                          // :855
            std_update_node_connections(x, n, x.enew, x.econn, lc, id);   // :856
        }
        row_ptr(g, id, lc)[0] = 0u;                                   // the node is gone (core.rs:419)
    }
    *sc.epoch = x.epoch;
    if (x.ovf) *sc.status = 1u;
    g.hdr->n_touched = x.nt;
    if (threadIdx.x != 0) return;
    atomicAdd(&g.hdr->ctr_insert[0], x.n_dist);
    atomicAdd(&g.hdr->ctr_insert[1], x.n_ids);
}

// which[0 .. *count) = the queries whose tie flag is set (all = 1: every query of the batch); any order
__global__ void k_tie_compact(const uint32_t *flags, uint32_t B, uint32_t *which, uint32_t *count, uint32_t all)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < B && (all || flags[i])) which[atomicAdd(count, 1u)] = i;
}

// HNSW.SEARCH (core.rs:477-486, 865-892) in the reference binary's tie order for the queries which[0 .. *n_which): one
// wavefront per query (lanes in lockstep, see std_sim); block b serves entries b, b + gridDim.x, ... with scratch context b
__global__ __launch_bounds__(64) void k_search_std_heap(GraphView g, const StdScratch *scs, const float *Q, const uint32_t *which, const uint32_t *n_which, uint32_t k,
                                  uint32_t ef, uint32_t *out_ids, float *out_sims, uint32_t *out_n)
{
    const uint32_t n = *n_which;
    StdCtx x;
    std_ctx_init(x, g, scs[blockIdx.x]);
    for (uint32_t w = blockIdx.x; w < n; w += gridDim.x) {
        const uint32_t qi = which[w];
        const float *q = Q + (size_t)qi * g.dim;
        uint32_t ep = (uint32_t)g.hdr->enterpoint, lc = g.hdr->max_layer;
        while (lc > 0) {                                              // :869-874
            std_search_level(x, q, ep, 1, lc);
            ep = std_get(x.res, 0).id;                                       // :872 peek
            lc--;
        }
        std_search_level(x, q, ep, ef, 0);                            // :876
        uint32_t m = 0;
        while (m < k && x.res.n) {                                    // :878-890
            const StdPair p = std_pop(x.res);
            out_ids[(size_t)qi * k + m] = p.id;
            out_sims[(size_t)qi * k + m] = p.sim;
            m++;
        }
        for (uint32_t i = m; i < k; ++i) { out_ids[(size_t)qi * k + i] = kEmpty; out_sims[(size_t)qi * k + i] = -__builtin_inff(); }
        out_n[qi] = m;
    }
    *x.sc.epoch = x.epoch;
    if (x.ovf) { *scs[blockIdx.x].status = 1u; atomicOr(&g.hdr->status, ST_STD_OVERFLOW); }     // (the insert's launcher reads the word itself: std_status)
}

} // namespace hnsw
