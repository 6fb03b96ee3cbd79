// hnsw_device.hpp -- device-side building blocks of the MI355X HNSW engine.
//
// gfx950 only.  One 64-lane wavefront owns one query (or one insert).  Its state:
//   W        sorted (dist, id) keys -- the reference's W and C heaps in one list.
//            search_level_v2 (every dim % 32 == 0) keeps it in registers and uses
//            LDS only to scatter a merge; search_level_v1 (the reference's scalar
//            metric order, other dims) keeps it in LDS.
//   visited  exact hash set in LDS (core.rs:614 HashSet): 16-bit tags behind a
//            bijective hash, or 32-bit ids above 16 M nodes; moves to an HBM
//            table if it ever fills.
//   fresh / dsc / S / aux   LDS scratch of the v1 path and the insert kernels.
// The vector matrix is row-major f32 [N][dim]; 8 lanes stream one row with
// 16 B loads (one 128 B line per 8-lane group per load instruction).
// Everything is force-inlined into the kernels: a real call makes the compiler
// spill the state structs to scratch and turn LDS accesses into flat ones.
#pragma once
#include "hnsw_wave_sync.hpp"
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace hnsw {

constexpr uint32_t kEmpty = 0xFFFFFFFFu;
constexpr uint32_t kNoUpper = 0xFFFFFFFFu;
constexpr int MODE_AVX = 0;    // dim % 32 == 0 : metrics.rs:48-77 summation order
constexpr int MODE_SCALAR = 1; // otherwise     : metrics.rs:79-84 left fold

// sticky status bits in DevHeader::status
constexpr uint32_t ST_VISITED_OVERFLOW = 1u; // spill table full
constexpr uint32_t ST_ROW_OVERFLOW = 2u;     // adjacency row full (engine bug: host re-strides first)
constexpr uint32_t ST_ROW_DROPPED = 4u;      // fast build dropped a reverse link
constexpr uint32_t ST_ASYMMETRIC = 8u;       // rm of a non-neighbour (reference would panic, core.rs:150)
constexpr uint32_t ST_STD_OVERFLOW = 16u;    // tie_mode: a heap of the std-order search kernel overflowed (answers of that call void)

struct DevHeader {
    uint32_t node_count;
    uint32_t max_layer;
    int32_t enterpoint;
    uint32_t max_deg0;
    uint32_t max_degU;
    uint32_t status;
    uint32_t n_touched; // exact insert: length of the touched list
    uint32_t pad;
    unsigned long long ctr_search[4]; // n_dist, n_ids, n_expand, n_spill
    unsigned long long ctr_insert[4];
    unsigned long long prof[8];       // HNSW_PHASE_TIMERS builds: cycles per phase
    // Tie census (hnsw_get_tie_counters): decisions that compared EQUAL distances of two different nodes -- the only places
    // where the engine's (distance, id) order and the reference's sim-only order on std's BinaryHeap (core.rs:292-300, :635,
    // :657, :733) can part.  [0] events in searches (tuning "tie_census"), [1] queries with at least one, [2] events in
    // inserts (plan searches, select_neighbors cuts of plans / speculative records / recomputed shrinks), [3] plans with one.
    unsigned long long ctr_tie[4];
};

// The synchronisation point of the code in this file -- lanes of ONE wavefront handing each other data.  The file is
// compiled into two kinds of translation units, and each unit says which it is before it includes anything:
//   HNSW_SYNC_WAVE_FULL  the insert / delete units (hnsw_tu_insert, _occ, _occteam, _occpar, _planlean, _planduo): the
//                        code is run by one wavefront per copy and hands over through HBM as well as LDS (read-log
//                        entries, rows) -> wave_sync() of hnsw_wave_sync.hpp: vmcnt(0) + lgkmcnt(0) + wave barrier;
//   HNSW_SYNC_BLOCK      the search units and the engine (hnsw_engine, hnsw_tu_search, _lean, _duo): 64-thread
//                        workgroups whose lanes hand over through LDS only -- the visited set's HBM spill tables are
//                        written and read with agent-scope atomics, W / the scatter buffer / the query live in LDS, and
//                        no kernel of these units loads a word another lane of the same launch stored to HBM with a plain
//                        store -> __syncthreads() (for one wave: LDS wait + wave barrier, no s_barrier).
// A unit that does not say is refused; the meaning no longer depends on which header came first.
#if defined(HNSW_SYNC_WAVE_FULL) == defined(HNSW_SYNC_BLOCK)
#error "define exactly one of HNSW_SYNC_WAVE_FULL / HNSW_SYNC_BLOCK before including the engine's headers (see hnsw_device.hpp)"
#endif
__device__ __forceinline__ void dev_sync()
{
#ifdef HNSW_SYNC_WAVE_FULL
    wave_sync();
#else
    __syncthreads();
#endif
}

struct GraphView {
    const float *vec;           // [cap][dim]
    uint32_t *adj0;             // [cap][stride0]  word 0 = count, then ids in stored order
    uint32_t *adjU;             // [upper slots][strideU]
    const uint32_t *upper_base; // [cap] first upper slot (layer 1) or kNoUpper
    const uint32_t *levels;     // [cap]
    DevHeader *hdr;
    uint32_t dim, stride0, strideU;
    uint32_t tagcfg;            // 16-bit tag visited table: log2(buckets) | idbits << 8; 0 = 32-bit ids
    uint32_t plan_stride;       // words per (slot, layer) row of the insert plans: 1 + max(64, M)
    uint32_t selcap;            // keys select_neighbors' result list holds in LDS: kSelMax, kSelMaxWide when M > 64
};

// Make this wave's own global stores visible to its own later loads: wait for them (they are written through
// to the XCD's L2), then drop this CU's L1 lines.  Unlike __threadfence() it does not write the L2 back --
// nothing outside this CU reads the data before the kernel ends (MI355X_MICROARCH.md: acquire agent fence
// ~1.7 us against ~3.5 us for the acq_rel form).
__device__ __forceinline__ void fence_own_writes()
{
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
}

__device__ __forceinline__ uint32_t *row_ptr(const GraphView &g, uint32_t id, uint32_t lc)
{
    if (lc == 0) return g.adj0 + (size_t)id * g.stride0;
    return g.adjU + (size_t)(g.upper_base[id] + lc - 1) * g.strideU;
}

// a row about to be EDITED (wave-uniform id; every lane calls).  For the graph itself that is the row; a private
// overlay of the graph (hnsw_occ_par.hpp: OverlayView) copies the row on first touch and hands out the copy.
__device__ __forceinline__ uint32_t *row_mut(const GraphView &g, uint32_t id, uint32_t lc, int lane)
{
    (void)lane;
    return row_ptr(g, id, lc);
}
// ... for an edit that REWRITES the row's live words (count + ids) in one go: where to write, and in *src where the row's
// present content is read from.  An overlay hands out an empty scratch row on first touch and the graph's row as the
// source -- no copy, the edit's own store fills the scratch row.
__device__ __forceinline__ uint32_t *row_rewrite(const GraphView &g, uint32_t id, uint32_t lc, int lane, const uint32_t **src)
{
    (void)lane;
    uint32_t *r = row_ptr(g, id, lc);
    *src = r;
    return r;
}

// ---------------------------------------------------------------------------
// keys: (squared distance bits << 32) | (id << 1) | expanded.  Squared
// distances are >= +0 so their IEEE bits order like unsigned ints; smaller key
// = nearer = larger reference sim, ties to the smaller id (tie order).
// ---------------------------------------------------------------------------
__device__ __forceinline__ uint64_t pack_key(float d, uint32_t id)
{
    return ((uint64_t)__float_as_uint(d) << 32) | ((uint64_t)id << 1);
}
__device__ __forceinline__ uint32_t key_id(uint64_t k) { return (uint32_t)(k & 0xFFFFFFFFu) >> 1; }
__device__ __forceinline__ float key_dist(uint64_t k) { return __uint_as_float((uint32_t)(k >> 32)); }

__device__ __forceinline__ uint64_t readlane64(uint64_t v, int lane)
{
    uint32_t lo = __builtin_amdgcn_readlane((int)(uint32_t)v, lane);
    uint32_t hi = __builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), lane);
    return ((uint64_t)hi << 32) | lo;
}
__device__ __forceinline__ uint64_t lanemask_lt(int lane) { return (1ull << lane) - 1ull; }

// ---------------------------------------------------------------------------
// cross-lane moves inside an 8-lane group (DPP, no LDS traffic)
// ---------------------------------------------------------------------------
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float x)
{
    return __builtin_bit_cast(
        float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xf, 0xf, true));
}
constexpr int DPP_QUAD_XOR1 = 0xB1;    // quad_perm [1,0,3,2]
constexpr int DPP_QUAD_XOR2 = 0x4E;    // quad_perm [2,3,0,1]
constexpr int DPP_HALF_MIRROR = 0x141; // lane i <-> 7-i inside each 8

// Lane i of an 8-lane group reads the 16-byte piece pp(i) of every 128-byte
// block.  Pieces 0..3 on lanes 0..3, pieces 7..4 on lanes 4..7, so that the
// three exchanges the AVX2 order needs are all single DPP moves:
//   piece ^ 2 = quad xor 2, piece ^ 4 = half mirror, piece ^ 1 = quad xor 1.
__device__ __forceinline__ int piece_of_lane(int lane)
{
    const int i = lane & 7;
    return i < 4 ? i : 11 - i;
}

template <int T>
struct QReg {
    float4 q[T > 0 ? T : 1];
};

// The reference's AVX2 kernel (metrics.rs:48-77) keeps 4 accumulators x 8
// lanes; element 32t + 8a + j goes to lane j of accumulator a by one FMA per
// t.  A piece p holds j = 4(p%2)+c of accumulator a = p/2, c = 0..3.
typedef float f32x2 __attribute__((ext_vector_type(2)));

// Two lanes of the reference's AVX registers per packed instruction: v_pk_add_f32 (exact
// subtraction) and v_pk_fma_f32 (IEEE fused multiply-add per half) -- same bits as the scalar
// forms, half the instructions of a wave that can only issue one every four cycles.
template <int T>
__device__ __forceinline__ float4 avx_accumulate(const float4 (&q)[T], const float4 (&v)[T])
{
    f32x2 alo = {0.f, 0.f}, ahi = {0.f, 0.f};
#pragma unroll
    for (int t = 0; t < T; ++t) {
        const f32x2 qlo = {q[t].x, q[t].y}, qhi = {q[t].z, q[t].w};
        const f32x2 vlo = {v[t].x, v[t].y}, vhi = {v[t].z, v[t].w};
        const f32x2 dlo = qlo - vlo, dhi = qhi - vhi;
        alo = __builtin_elementwise_fma(dlo, dlo, alo);   // metrics.rs:57,60,64,68
        ahi = __builtin_elementwise_fma(dhi, dhi, ahi);
    }
    return make_float4(alo.x, alo.y, ahi.x, ahi.y);
}

// (e1+e2)+(e3+e4) per lane (metrics.rs:71-74), low128+high128 (:37-39),
// (s0+s1)+(s2+s3) (:27-31).  Every lane of the group ends with the result.
__device__ __forceinline__ float avx_reduce(float4 a)
{
    a.x = __fadd_rn(a.x, dpp_mov<DPP_QUAD_XOR2>(a.x)); a.y = __fadd_rn(a.y, dpp_mov<DPP_QUAD_XOR2>(a.y));
    a.z = __fadd_rn(a.z, dpp_mov<DPP_QUAD_XOR2>(a.z)); a.w = __fadd_rn(a.w, dpp_mov<DPP_QUAD_XOR2>(a.w));
    a.x = __fadd_rn(a.x, dpp_mov<DPP_HALF_MIRROR>(a.x)); a.y = __fadd_rn(a.y, dpp_mov<DPP_HALF_MIRROR>(a.y));
    a.z = __fadd_rn(a.z, dpp_mov<DPP_HALF_MIRROR>(a.z)); a.w = __fadd_rn(a.w, dpp_mov<DPP_HALF_MIRROR>(a.w));
    a.x = __fadd_rn(a.x, dpp_mov<DPP_QUAD_XOR1>(a.x)); a.y = __fadd_rn(a.y, dpp_mov<DPP_QUAD_XOR1>(a.y));
    a.z = __fadd_rn(a.z, dpp_mov<DPP_QUAD_XOR1>(a.z)); a.w = __fadd_rn(a.w, dpp_mov<DPP_QUAD_XOR1>(a.w));
    return __fadd_rn(__fadd_rn(a.x, a.y), __fadd_rn(a.z, a.w));
}

// per-wave LDS slices
struct WaveMem {
    uint64_t *W;     // [R*64] sorted keys
    uint64_t *S;     // [128]  select_neighbors result list (insert kernels): up to m_max0 = 2M ids, M <= 64
    uint32_t *fresh; // [64]
    float *dsc;      // [64]
    float *qlds;     // [dim]  query copy (generic / scalar modes; also float4 view)
    uint32_t *aux;   // [kAuxWords] scratch for the insert kernels
};

// Load the wave's query into registers (AVX mode, T > 0) or LDS (T == 0).
template <int MODE, int T>
__device__ __forceinline__ void load_query(const float *src, uint32_t dim, QReg<T> &qr, float *qlds, int lane)
{
    if constexpr (MODE == MODE_AVX && T > 0) {
        const float4 *s4 = reinterpret_cast<const float4 *>(src);
        const int pp = piece_of_lane(lane);
#pragma unroll
        for (int t = 0; t < T; ++t) qr.q[t] = s4[t * 8 + pp];
    } else {
        for (uint32_t i = lane; i < dim; i += 64) qlds[i] = src[i];
        dev_sync();
    }
}

// Squared distances from the query to fresh[0..nf) -> dsc[0..nf).
template <int MODE, int T>
__device__ __forceinline__ void compute_dists(const GraphView &g, const QReg<T> &qr, const WaveMem &m,
                                              uint32_t nf, int lane)
{
    if constexpr (MODE == MODE_SCALAR) {
        // metrics.rs:79-84 left fold; one lane per vector keeps the order exact
        if ((uint32_t)lane < nf) {
            const float *v = g.vec + (size_t)m.fresh[lane] * g.dim;
            float acc = 0.f;
            for (uint32_t i = 0; i < g.dim; ++i) {
                float d = __fsub_rn(m.qlds[i], v[i]);
                acc = __fadd_rn(acc, __fmul_rn(d, d));
            }
            m.dsc[lane] = acc;
        }
    } else if constexpr (T > 0) {
        const int grp = lane >> 3, pp = piece_of_lane(lane);
        constexpr int RB = (T <= 4) ? 4 : (T <= 8 ? 2 : 1); // rounds of 8 vectors in flight
        for (uint32_t base = 0; base < nf; base += 8 * RB) {
            float4 v[RB][T];
#pragma unroll
            for (int r = 0; r < RB; ++r) {
                uint32_t vi = base + r * 8 + grp;
                if (vi < nf) {
                    const float4 *p = reinterpret_cast<const float4 *>(g.vec + (size_t)m.fresh[vi] * g.dim) + pp;
#pragma unroll
                    for (int t = 0; t < T; ++t) v[r][t] = p[t * 8];
                }
            }
#pragma unroll
            for (int r = 0; r < RB; ++r) {
                uint32_t vi = base + r * 8 + grp;
                if (vi < nf) {
                    float d = avx_reduce(avx_accumulate<T>(qr.q, v[r]));
                    if ((lane & 7) == 0) m.dsc[vi] = d;
                }
            }
        }
    } else {
        // AVX order, any dim % 32 == 0: query pieces come from LDS
        const int grp = lane >> 3, pp = piece_of_lane(lane);
        const uint32_t Trt = g.dim >> 5;
        const float4 *q4 = reinterpret_cast<const float4 *>(m.qlds) + pp;
        for (uint32_t base = 0; base < nf; base += 8) {
            uint32_t vi = base + grp;
            if (vi < nf) {
                const float4 *p = reinterpret_cast<const float4 *>(g.vec + (size_t)m.fresh[vi] * g.dim) + pp;
                float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
                for (uint32_t t = 0; t < Trt; ++t) {
                    float4 x = p[t * 8], q = q4[t * 8];
                    float dx = __fsub_rn(q.x, x.x), dy = __fsub_rn(q.y, x.y);
                    float dz = __fsub_rn(q.z, x.z), dw = __fsub_rn(q.w, x.w);
                    acc.x = __fmaf_rn(dx, dx, acc.x); acc.y = __fmaf_rn(dy, dy, acc.y);
                    acc.z = __fmaf_rn(dz, dz, acc.z); acc.w = __fmaf_rn(dw, dw, acc.w);
                }
                float d = avx_reduce(acc);
                if ((lane & 7) == 0) m.dsc[vi] = d;
            }
        }
    }
}

// ---------------------------------------------------------------------------
// exact visited set (core.rs:614 HashSet): bucketed hash in LDS, spilling to an
// HBM table when it fills.  A bucket is 8 words = 32 bytes: word 0 counts the
// arrivals, words 1..7 hold ids.  Lookup = two ds_read_b128 + 7 compares;
// insert = one ds_add_rtn on the counter (which hands concurrent lanes distinct
// slots, so there is no retry) + one ds_write.  A bucket that has seen 7
// arrivals is full: later ids for it chain into the next bucket, and a lookup
// follows the chain only through full buckets.  Any bucket count works (the
// host sizes the table to the LDS budget, not to a power of two).
// ---------------------------------------------------------------------------
struct Visited {
    uint32_t *lds;
    uint32_t *glob;
    uint32_t lnb, gnb;       // buckets in the LDS / HBM table
    uint32_t lcap;           // ids the LDS table may hold before the set moves to HBM
    uint32_t tag_bb;         // 0: 32-bit ids in LDS; else log2(buckets) of the 16-bit tag table
    uint32_t idbits;         // tag mode: ids are < 2^idbits
    uint32_t count;          // ids held (wave-uniform)
    bool spilled;            // wave-uniform
    bool glob_dirty;         // wave-uniform
    bool bounded;            // search only: a full LDS table stops taking ids instead of moving to HBM
    bool lossy;              // wave-uniform: the bounded table is full, ids met from now on are not recorded
};
constexpr uint32_t kBucketIds = 7;

__device__ __forceinline__ uint32_t hash_bucket(uint32_t id, uint32_t nb)
{
    return __umulhi(id * 0x9E3779B1u, nb);
}
__device__ __forceinline__ bool any_eq7(const uint4 &lo, const uint4 &hi, uint32_t id)
{
    // min over the xors is 0 iff some slot equals id
    uint32_t a = min(min(lo.y ^ id, lo.z ^ id), lo.w ^ id);
    uint32_t c = min(min(hi.x ^ id, hi.y ^ id), min(hi.z ^ id, hi.w ^ id));
    return min(a, c) == 0u;
}

// true if id was not in the table (and is now).  Lanes of one wave call this
// concurrently with distinct ids; visited_reserve keeps the table <= ~7/8 full,
// the iteration cap only guards against a corrupted table.
__device__ __forceinline__ bool lds_set_insert(uint32_t *tab, uint32_t nb, uint32_t id)
{
    uint32_t b = hash_bucket(id, nb);
    for (uint32_t it = 0; it < nb; ++it) {
        uint32_t *bp = tab + (b << 3);
        const uint4 *p4 = reinterpret_cast<const uint4 *>(bp);
        const uint4 lo = p4[0], hi = p4[1];
        if (any_eq7(lo, hi, id)) return false;
        if (lo.x < kBucketIds) {
            const uint32_t pos = atomicAdd(bp, 1u);
            if (pos < kBucketIds) { bp[1 + pos] = id; return true; }
        }
        b = b + 1 == nb ? 0 : b + 1;   // full: the id lives further down the chain
    }
    return false;
}

// Same protocol on the HBM spill table.  Loads are agent-scope atomics so they
// are served by L2, where the atomic adds execute (a plain load could hit a
// stale L1 line); stores are atomic stores for the same reason.
__device__ __forceinline__ bool glob_set_insert(uint32_t *tab, uint32_t nb, uint32_t id)
{
    uint32_t b = hash_bucket(id, nb);
    for (uint32_t it = 0; it < nb; ++it) {
        uint32_t *bp = tab + ((size_t)b << 3);
        uint32_t w[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) w[i] = __hip_atomic_load(bp + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const uint4 lo = make_uint4(w[0], w[1], w[2], w[3]), hi = make_uint4(w[4], w[5], w[6], w[7]);
        if (any_eq7(lo, hi, id)) return false;
        if (w[0] < kBucketIds) {
            const uint32_t pos = atomicAdd(bp, 1u);
            if (pos < kBucketIds) {
                __hip_atomic_store(bp + 1 + pos, id, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                return true;
            }
        }
        b = b + 1 == nb ? 0 : b + 1;
    }
    return false;
}

// ---- 16-bit tag mode -------------------------------------------------------------------------
// Twice the ids per LDS byte, still exact: the id is run through a BIJECTION of [0, 2^idbits)
// (odd multiplier, then xor-shift), the low tag_bb bits pick a 16-byte bucket and the remaining
// bits (<= 13) are stored as a tag next to a 3-bit displacement 0..6 (how many buckets past its
// home the entry had to go).  (home bucket, tag) identifies the id uniquely, so equality of the 16-bit
// entry in bucket home+d is equality of ids.  Bucket = [u16 arrivals][u16 entry x 7]; one
// ds_read_b128 per lookup.  Free slots hold 0xFFFF, which no entry can equal (a displacement is
// never 7).  If an id finds seven consecutive full buckets the query falls back to the HBM table: entries are decoded back to ids (the hash is
// inverted) and moved there.
constexpr uint32_t kTagMul = 0x9E3779B1u;
constexpr uint32_t kTagMulInv = 0x0E8B2F51u;   // kTagMul * kTagMulInv == 1 (mod 2^32)
static_assert((uint32_t)(kTagMul * kTagMulInv) == 1u, "modular inverse");

__device__ __forceinline__ uint32_t tag_hash(uint32_t id, uint32_t idbits)
{
    const uint32_t mask = idbits >= 32 ? 0xFFFFFFFFu : ((1u << idbits) - 1u);
    const uint32_t x = (id * kTagMul) & mask;
    return x ^ (x >> ((idbits + 1) >> 1));
}
__device__ __forceinline__ uint32_t tag_unhash(uint32_t h, uint32_t idbits)
{
    const uint32_t mask = idbits >= 32 ? 0xFFFFFFFFu : ((1u << idbits) - 1u);
    const uint32_t x = h ^ (h >> ((idbits + 1) >> 1));   // shift >= idbits/2: self-inverse
    return (x * kTagMulInv) & mask;
}
__device__ __forceinline__ uint32_t haszero16(uint32_t x) { return (x - 0x00010001u) & ~x & 0x80008000u; }

// 1: inserted (was absent), 0: already present, 2: seven full buckets in a row (caller must spill)
__device__ __forceinline__ uint32_t tag_set_insert(uint32_t *tab, uint32_t bb, uint32_t idbits, uint32_t id)
{
    const uint32_t h = tag_hash(id, idbits);
    const uint32_t bmask = (1u << bb) - 1u;
    const uint32_t b0 = h & bmask, tag = h >> bb;
    for (uint32_t d = 0; d < 7; ++d) {
        const uint32_t b = (b0 + d) & bmask;
        uint32_t *bp = tab + (b << 2);
        const uint4 w = *reinterpret_cast<const uint4 *>(bp);
        const uint32_t want = (tag << 3) | d;
        const uint32_t ww = want | (want << 16);
        const uint32_t hit = haszero16(w.y ^ ww) | haszero16(w.z ^ ww) | haszero16(w.w ^ ww) |
                             ((w.x >> 16) == want ? 1u : 0u);
        if (hit) return 0;
        if ((w.x & 0xFFFFu) < kBucketIds) {
            const uint32_t old = atomicAdd(bp, 1u) & 0xFFFFu;      // arrivals live in the low half
            if (old < kBucketIds) {
                reinterpret_cast<unsigned short *>(bp)[1 + old] = (unsigned short)want;
                return 1;
            }
        }
    }
    return 2;
}

// lookup without insert (bounded table that is full): true if id is NOT in the table
__device__ __forceinline__ bool tag_set_absent(const uint32_t *tab, uint32_t bb, uint32_t idbits, uint32_t id)
{
    const uint32_t h = tag_hash(id, idbits);
    const uint32_t bmask = (1u << bb) - 1u;
    const uint32_t b0 = h & bmask, tag = h >> bb;
    for (uint32_t d = 0; d < 7; ++d) {
        const uint4 w = *reinterpret_cast<const uint4 *>(tab + (((b0 + d) & bmask) << 2));
        const uint32_t want = (tag << 3) | d;
        const uint32_t ww = want | (want << 16);
        const uint32_t hit = haszero16(w.y ^ ww) | haszero16(w.z ^ ww) | haszero16(w.w ^ ww) |
                             ((w.x >> 16) == want ? 1u : 0u);
        if (hit) return false;
        if ((w.x & 0xFFFFu) < kBucketIds) return true;       // an id only chains past FULL buckets
    }
    return true;
}
__device__ __forceinline__ bool lds_set_absent(const uint32_t *tab, uint32_t nb, uint32_t id)
{
    uint32_t b = hash_bucket(id, nb);
    for (uint32_t it = 0; it < nb; ++it) {
        const uint4 *p4 = reinterpret_cast<const uint4 *>(tab + (b << 3));
        const uint4 lo = p4[0], hi = p4[1];
        if (any_eq7(lo, hi, id)) return false;
        if (lo.x < kBucketIds) return true;
        b = b + 1 == nb ? 0 : b + 1;
    }
    return true;
}

// empty bucket = {0, kEmpty x 7}.  16-byte piece i is the head of a bucket iff i
// is even; i = lane + 64k keeps the lane's parity, so each lane stores one constant.
__device__ __forceinline__ void visited_clear(Visited &v, int lane)
{
    uint4 *t4 = reinterpret_cast<uint4 *>(v.lds);
    if (v.tag_bb) {
        // 16-byte buckets: low half of word 0 = arrivals (0), everything else 0xFFFF
        const uint4 e = make_uint4(0xFFFF0000u, kEmpty, kEmpty, kEmpty);
        for (uint32_t i = lane; i < (1u << v.tag_bb); i += 64) t4[i] = e;
    } else {
        const uint4 e = make_uint4((lane & 1) ? kEmpty : 0u, kEmpty, kEmpty, kEmpty);
        for (uint32_t i = lane; i < 2 * v.lnb; i += 64) t4[i] = e;
    }
    if (v.glob_dirty) {
        uint4 *g4 = reinterpret_cast<uint4 *>(v.glob);
        const uint4 e = make_uint4((lane & 1) ? kEmpty : 0u, kEmpty, kEmpty, kEmpty);
        for (uint32_t i = lane; i < 2 * v.gnb; i += 64) g4[i] = e;
        __threadfence();
        v.glob_dirty = false;
    }
    v.count = 0;
    v.spilled = false;
    v.lossy = false;
    dev_sync();
}

// Move the whole LDS set to the HBM table and continue there.
__device__ __forceinline__ void visited_spill(Visited &v, int lane, unsigned long long *spill_ctr)
{
    if (v.tag_bb) {
        const uint32_t nb = 1u << v.tag_bb, bmask = nb - 1u;
        const unsigned short *t16 = reinterpret_cast<const unsigned short *>(v.lds);
        for (uint32_t i = lane; i < nb * 8u; i += 64) {
            const uint32_t b = i >> 3, sl = i & 7u;
            const uint32_t arrivals = t16[b << 3];
            if (sl >= 1 && sl <= (arrivals < kBucketIds ? arrivals : kBucketIds)) {
                const uint32_t e = t16[i];
                const uint32_t home = (b - (e & 7u)) & bmask;
                const uint32_t id = tag_unhash(((e >> 3) << v.tag_bb) | home, v.idbits);
                glob_set_insert(v.glob, v.gnb, id);
            }
        }
    } else {
        for (uint32_t i = lane; i < v.lnb * 8u; i += 64) {
            const uint32_t id = v.lds[i];
            if ((i & 7u) != 0u && id != kEmpty) glob_set_insert(v.glob, v.gnb, id);
        }
    }
    __threadfence();
    dev_sync();
    v.spilled = true;
    v.glob_dirty = true;
    if (lane == 0 && spill_ctr) atomicAdd(spill_ctr, 1ull);
}

// Whole-wave insert of one id per lane flagged `valid`; returns this lane's "was absent" flag.
// (Tag mode can hit a run of full buckets: the set then moves to HBM and the lanes concerned
// retry there; lanes that already inserted keep their answer.)
__device__ __forceinline__ bool visited_insert_wave(Visited &v, bool valid, uint32_t id, int lane,
                                                    unsigned long long *spill_ctr)
{
    if (v.spilled) return valid && glob_set_insert(v.glob, v.gnb, id);
    if (v.lossy)           // bounded table, full: look up only (the caller drops re-met members of W itself)
        return valid && (v.tag_bb ? tag_set_absent(v.lds, v.tag_bb, v.idbits, id) : lds_set_absent(v.lds, v.lnb, id));
    if (!v.tag_bb) return valid && lds_set_insert(v.lds, v.lnb, id);
    const uint32_t r = valid ? tag_set_insert(v.lds, v.tag_bb, v.idbits, id) : 0u;
    if (v.bounded) {
        if (__ballot(r == 2u)) {                      // no room near an id's home bucket: stop recording
            v.lossy = true;
            if (lane == 0 && spill_ctr) atomicAdd(spill_ctr, 1ull);
        }
        return r != 0u;                               // 2 = absent (and not recorded)
    }
    if (__ballot(r == 2u)) {
        visited_spill(v, lane, spill_ctr);
        if (r == 2u) return glob_set_insert(v.glob, v.gnb, id);
    }
    return r == 1u;
}

// single-lane form (entry points): a fresh table cannot overflow
__device__ __forceinline__ bool visited_insert(const Visited &v, uint32_t id)
{
    if (v.spilled) return glob_set_insert(v.glob, v.gnb, id);
    if (v.tag_bb) return tag_set_insert(v.lds, v.tag_bb, v.idbits, id) == 1u;
    return lds_set_insert(v.lds, v.lnb, id);
}

// Make room for up to 64 more ids.  Returns false if even the HBM table is full.
__device__ __forceinline__ bool visited_reserve(Visited &v, int lane, unsigned long long *spill_ctr)
{
    const uint32_t gcap = v.gnb * 6u;                  // ids at ~6/7 of the slots
    if (!v.spilled) {
        if (v.count + 64 <= v.lcap || v.lossy) return true;
        if (v.bounded) {                               // search: keep what is recorded, forget the rest
            v.lossy = true;
            if (lane == 0 && spill_ctr) atomicAdd(spill_ctr, 1ull);
            return true;
        }
        if (v.count + 64 > gcap) return false;         // would not fit there either
        visited_spill(v, lane, spill_ctr);
    }
    return v.count + 64 <= gcap;
}

// ---------------------------------------------------------------------------
// W: merge up to 64 new keys (one per lane, `take` marks the lanes that carry
// one) into the sorted list W[0..nW), keep the best `cap`.  Rank-and-scatter:
// every old entry moves up by the number of new keys below it, every new key
// lands at (#old below) + (#new below).  Equivalent to the reference pushing
// the neighbours one by one (core.rs:657-664) -- the final W is the top-`cap`
// of the union either way.
// ---------------------------------------------------------------------------
template <int R>
__device__ __forceinline__ uint32_t merge_sorted(uint64_t *W, uint32_t nW, uint32_t cap, uint64_t nk,
                                                 bool take, int lane, uint32_t *ties = nullptr)
{
    // ties (tie census of a select_neighbors cut, core.rs:733 / :741-754): ties[1] keeps the nearest distance among the keys
    // pushed out of the list; the callers add the arrivals they reject and compare with the final last key (merge_S)
    const uint64_t tmask = __ballot(take);
    if (tmask == 0) return nW;
    uint64_t w[R];
    uint32_t up[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        uint32_t i = r * 64 + lane;
        w[r] = i < nW ? W[i] : ~0ull;
        up[r] = 0;
    }
    uint32_t mypos = 0;
    uint64_t mm = tmask;
    while (mm) {
        const int j = __ffsll((unsigned long long)mm) - 1;
        mm &= mm - 1;
        const uint64_t s = readlane64(nk, j);
        uint32_t rank = 0;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const bool below = w[r] < s;
            rank += __popcll(__ballot(below));
            up[r] += below ? 0u : 1u;
        }
        rank += __popcll(__ballot(take && nk < s));
        if (lane == j) mypos = rank;
    }
    dev_sync();
#pragma unroll
    for (int r = 0; r < R; ++r) {
        uint32_t i = r * 64 + lane;
        if (i < nW) {
            uint32_t np = i + up[r];
            if (np < cap) W[np] = w[r];
        }
    }
    if (take && mypos < cap) W[mypos] = nk;
    const uint32_t total = nW + (uint32_t)__popcll(tmask);
    dev_sync();
    if (ties && total > cap) {
        // ties[1] = min(ties[1], distances of the keys this merge pushed out): the caller compares it with the list's last
        // distance when the selection is complete (a cut between equal distances)
        uint32_t dmin = take && mypos >= cap ? (uint32_t)(nk >> 32) : 0xFFFFFFFFu;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const uint32_t i = r * 64 + lane;
            if (i < nW && i + up[r] >= cap) dmin = min(dmin, (uint32_t)(w[r] >> 32));
        }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) dmin = min(dmin, (uint32_t)__shfl_xor((int)dmin, o, 64));
        ties[1] = min(ties[1], dmin);
    }
    return total < cap ? total : cap;
}

// tie census: do two neighbouring keys of the sorted list L[0..n) have equal distances?  (Keys that the reference pops
// "nearest first" from a heap ordered by similarity alone: which of two equal ones comes first is the heap's choice.)
__device__ __forceinline__ bool any_adjacent_equal(const uint64_t *L, uint32_t n, int lane)
{
    bool eq = false;
    for (uint32_t i = lane; i + 1 < n; i += 64) eq |= (uint32_t)(L[i] >> 32) == (uint32_t)(L[i + 1] >> 32);
    return __ballot(eq) != 0;
}

// index of the nearest entry not yet expanded (core.rs:631 pop of C), or -1.
template <int R>
__device__ __forceinline__ int find_unexpanded(const uint64_t *W, uint32_t nW, int lane)
{
#pragma unroll
    for (int r = 0; r < R; ++r) {
        uint32_t i = r * 64 + lane;
        if (r * 64u >= nW) break;
        bool u = i < nW && !(W[i] & 1ull);
        uint64_t b = __ballot(u);
        if (b) return r * 64 + (__ffsll((unsigned long long)b) - 1);
    }
    return -1;
}

constexpr uint32_t kAuxWords = 1024; // insert scratch: one adjacency row (degree <= 1023; the reference does not bound degrees, core.rs:790-796)
constexpr uint32_t kSelMax = 128;   // select_neighbors result: m_max0 = 2M ids at most (M <= 64; g.selcap = 256 for an index with M > 64)
constexpr uint32_t kSelMaxWide = 256;
constexpr uint32_t kMaxM = 128;     // M above 64 is served by the serial insert / delete kernels only (hnsw_create)

// LDS carve-up.  Search: [W: R*64*8][fresh: 64*4][dsc: 64*4][qlds (T==0)][hash: nb*32].
// The insert kernels add [S: 64*8][aux: kAuxWords*4] after dsc.
__host__ __device__ inline size_t lds_fixed_bytes(int R, int T, uint32_t dim, bool ins, uint32_t selcap = kSelMax)
{
    size_t b = (size_t)R * 64 * 8 + 64 * 4 + 64 * 4;
    if (ins) b += (size_t)selcap * 8 + kAuxWords * 4;
    if (T == 0) b += ((size_t)dim * 4 + 15) & ~(size_t)15;
    return b;
}
__host__ __device__ inline size_t lds_bytes(int R, int T, uint32_t dim, uint32_t nbuckets, bool ins, uint32_t selcap = kSelMax)
{
    return lds_fixed_bytes(R, T, dim, ins, selcap) + (size_t)nbuckets * 32;
}

template <int R, int T, bool INS>
__device__ __forceinline__ void carve(unsigned char *smem, uint32_t dim, uint32_t nbuckets, uint32_t lcap, WaveMem &m,
                                      Visited &vis, uint32_t tagcfg = 0, uint32_t selcap = kSelMax)
{
    unsigned char *p = smem;
    m.W = reinterpret_cast<uint64_t *>(p); p += (size_t)R * 64 * 8;
    m.fresh = reinterpret_cast<uint32_t *>(p); p += 64 * 4;
    m.dsc = reinterpret_cast<float *>(p); p += 64 * 4;
    m.S = nullptr;
    m.aux = nullptr;
    if (INS) {
        m.S = reinterpret_cast<uint64_t *>(p); p += (size_t)selcap * 8;
        m.aux = reinterpret_cast<uint32_t *>(p); p += kAuxWords * 4;
    }
    m.qlds = reinterpret_cast<float *>(p);
    if (T == 0) p += ((size_t)dim * 4 + 15) & ~(size_t)15;
    vis.lds = reinterpret_cast<uint32_t *>(p);
    vis.lnb = nbuckets;
    vis.lcap = lcap;
    vis.tag_bb = tagcfg & 0xFFu;          // tagcfg = log2(16-byte buckets) | idbits << 8, or 0
    vis.idbits = tagcfg >> 8;
}

// One entry of a plan's read log (hnsw_occ.hpp: exact-order parallel insert).  A plan stays valid as long as
// no row it read changed in a way that matters: an id z added to / removed from the row matters iff
// dist(reader, z) <= `bound` (distance bits); `full` clear = everything matters.
//   select_neighbors (core.rs:724-754): bound = the last selected candidate.
//   search_level expansion (core.rs:630-668): bound = max(W's final furthest, the furthest candidate popped
//   AFTER this expansion) -- occ_finalize_search_log, where the argument is written out.
struct OccRead {
    uint32_t row;
    uint32_t meta;      // layer [0,5) | kind [5,7) | sub-operation [7,13) | full [13]
    uint32_t bound;
};
constexpr uint32_t OCC_SEARCH = 0, OCC_SELECT = 1, OCC_SHRINK_NB = 2, OCC_SHRINK_ROW = 3;
__host__ __device__ inline uint32_t occ_meta(uint32_t lc, uint32_t kind, uint32_t sub, bool full)
{
    return (lc & 31u) | (kind << 5) | ((sub & 63u) << 7) | (full ? 1u << 13 : 0u);
}

struct WorkCtr {
    uint32_t n_dist, n_ids, n_expand;
    uint32_t n_tie;          // tie census (DevHeader::ctr_tie): counted by the routines instantiated with TIES
    uint32_t tie_emin;       // ... its running minimum (&n_tie + 1: the routines take one pointer): the nearest distance among the keys that
                             // fell out of the list being kept but are still candidates (search_level), or among everything outside the
                             // selection (select_neighbors); all ones = none
    OccRead *log;            // nullptr: no read log
    uint32_t log_n, log_cap; // entries written / capacity (log_n keeps counting past the capacity)
#ifdef HNSW_PHASE_TIMERS
    unsigned long long ph[8];
#endif
};
// The threshold of a search_level read.  W(t) is always the ef nearest of everything evaluated so far, and the
// loop pops the nearest unexpanded member, stopping at the first pop farther than W's furthest (core.rs:635).
// Take an id z that appears in (or vanishes from) a row expanded at time t, with
//        dist(q, z) > T := max( W's furthest at the end, every candidate popped after t ).
// Present, z can enter W and the candidate heap for a while (the accept threshold of the moment may be far
// looser than T -- it is infinite until W fills), but: every later pop is nearer than z, so z is never the
// nearest unexpanded candidate before the loop ends, hence never expanded; the members it displaces from W,
// and the ones that its presence keeps from being accepted, are all farther than z, so none of them is ever
// popped either; the stop test compares pops (nearer than z) with a furthest that is >= dist(z) while z is in
// W and unchanged once z has been pushed out; and at the end the nearest unexpanded candidate, z or the
// original one, is farther than the final furthest, which z is not part of.  Pops, expansions and the final W
// are the same with and without z.  With dist(q, z) <= T nothing is claimed: the plan is redone.
// The accept threshold at time t (what the log carried before) is >= T: this bound is never looser, and for the
// rows expanded while W is still filling -- the hubs every search passes through -- it replaces "everything
// matters" by roughly the ef-th neighbour's distance.
// Entries [log_start, ctr.log_n) were written by lane 0 with the popped candidate's distance as `bound`.
__device__ __forceinline__ void occ_finalize_search_log(WorkCtr &ctr, uint32_t log_start, uint32_t lc, uint64_t worst, int lane)
{
    if (!ctr.log) return;
    const uint32_t end = ctr.log_n < ctr.log_cap ? ctr.log_n : ctr.log_cap;
    if (end <= log_start) return;
    fence_own_writes();                                   // lane 0's entries, read back by all lanes
    const bool full = worst != ~0ull;
    uint32_t running = (uint32_t)(worst >> 32);           // all ones while W never filled: everything matters
    const uint32_t meta = occ_meta(lc, OCC_SEARCH, 0, full);
    for (uint32_t hi = end; hi > log_start;) {
        const uint32_t n = hi - log_start < 64u ? hi - log_start : 64u;
        const bool on = (uint32_t)lane < n;
        const uint32_t idx = hi - 1u - (uint32_t)lane;    // lane 0 = the latest expansion of this chunk
        const uint32_t pop = on ? ctr.log[idx].bound : 0u;
        uint32_t incl = pop;                              // max over lanes <= this one (= this and later expansions)
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t o = (uint32_t)__shfl_up((int)incl, d, 64);
            if (lane >= d && o > incl) incl = o;
        }
        uint32_t excl = (uint32_t)__shfl_up((int)incl, 1, 64);
        if (lane == 0) excl = 0u;
        const uint32_t t = running > excl ? running : excl;
        if (on) { ctr.log[idx].bound = t; ctr.log[idx].meta = meta; }
        const uint32_t all = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
        if (all > running) running = all;
        hi -= n;
    }
}

#ifdef HNSW_PHASE_TIMERS
#define PH_T0() unsigned long long ph_t_ = __builtin_readcyclecounter()
#define PH_MARK(ctr, i)                                              \
    do {                                                             \
        unsigned long long n_ = __builtin_readcyclecounter();        \
        (ctr).ph[i] += n_ - ph_t_;                                   \
        ph_t_ = n_;                                                  \
    } while (0)
#else
#define PH_T0() do {} while (0)
#define PH_MARK(ctr, i) do {} while (0)
#endif

// ---------------------------------------------------------------------------
// search_level (core.rs:607-675).  W doubles as C: an entry is a live
// candidate while its `expanded` bit is clear.  The reference's C also keeps
// pairs already evicted from W, but those can never be expanded: W's furthest
// only improves, so `c.sim < f.sim` (core.rs:635) stops the loop the moment
// one is popped.  Leaves W[0..n) sorted nearest first; returns n.
// ---------------------------------------------------------------------------
template <int MODE, int T, int R>
__device__ __forceinline__ uint32_t search_level_v1(const GraphView &g, const WaveMem &m, Visited &vis, const QReg<T> &qr,
                                 uint32_t ep, uint32_t ef, uint32_t lc, WorkCtr &ctr, int lane, bool &fail)
{
    visited_clear(vis, lane);                    // core.rs:614
    if (lane == 0) {
        m.fresh[0] = ep;
        visited_insert(vis, ep);                 // core.rs:617
    }
    vis.count = 1;
    dev_sync();
    compute_dists<MODE, T>(g, qr, m, 1, lane);   // core.rs:621
    ctr.n_dist += 1;
    dev_sync();
    if (lane == 0) m.W[0] = pack_key(m.dsc[0], ep); // core.rs:627-628
    uint32_t nW = 1;
    dev_sync();
    const uint32_t stride = lc ? g.strideU : g.stride0;

    const uint32_t log_start = ctr.log_n;
    PH_T0();
    for (;;) {
        const int pos = find_unexpanded<R>(m.W, nW, lane); // core.rs:631
        if (pos < 0) break;                               // core.rs:630,635
        const uint64_t ckey = m.W[pos];
        const uint32_t c = key_id(ckey);
        dev_sync();
        if (lane == 0) m.W[pos] = ckey | 1ull;
        ctr.n_expand += 1;
        const uint32_t log_idx = ctr.log_n;
        if (ctr.log) ctr.log_n += 1;

        const uint32_t *row = row_ptr(g, c, lc);          // core.rs:645
        uint32_t word = (uint32_t)lane < stride ? row[lane] : 0u;
        uint32_t cnt = __builtin_amdgcn_readfirstlane(word);
        if (cnt > stride - 1) cnt = stride - 1;
        ctr.n_ids += cnt;
        PH_MARK(ctr, 0);  // pop + adjacency row fetch
        for (uint32_t wbase = 0; wbase <= cnt; wbase += 64) { // core.rs:646 stored order
            const uint32_t wi = wbase + lane;
            if (wbase) word = wi < stride ? row[wi] : 0u;
            const bool valid = wi >= 1 && wi <= cnt;
            if (!visited_reserve(vis, lane, &g.hdr->ctr_search[3])) { fail = true; return nW; }
            const bool fresh = visited_insert_wave(vis, valid, word, lane, &g.hdr->ctr_search[3]); // core.rs:648-649
            const uint64_t fm = __ballot(fresh);
            const uint32_t nf = __popcll(fm);
            PH_MARK(ctr, 1);  // visited filter
            if (nf == 0) continue;
            if (fresh) m.fresh[__popcll(fm & lanemask_lt(lane))] = word;
            vis.count += nf;
            ctr.n_dist += nf;
            dev_sync();
            compute_dists<MODE, T>(g, qr, m, nf, lane);    // core.rs:652
            dev_sync();
            const bool have = (uint32_t)lane < nf;
            const uint64_t key = have ? pack_key(m.dsc[lane], m.fresh[lane]) : ~0ull;
            PH_MARK(ctr, 2);  // vector gather + distances
            const uint64_t worst = nW == ef ? m.W[ef - 1] : ~0ull; // core.rs:651
            const bool take = have && key < worst;          // core.rs:657
            nW = merge_sorted<R>(m.W, nW, ef, key, take, lane); // core.rs:659-664
            PH_MARK(ctr, 3);  // merge into W
        }
        dev_sync();
        // read log: the accept threshold once this row's keys are in (what a later change of the row is judged by)
        if (ctr.log && lane == 0 && log_idx < ctr.log_cap)
            ctr.log[log_idx] = OccRead{c, occ_meta(lc, OCC_SEARCH, 0, false), (uint32_t)(ckey >> 32)};
    }
    occ_finalize_search_log(ctr, log_start, lc, nW == ef ? m.W[ef - 1] : ~0ull, lane);
    return nW;
}

// ---------------------------------------------------------------------------
// search_level, second generation: W lives in registers (lane L owns entries
// L, 64+L, ...), LDS only carries the scatter of the merge; neighbour ids reach
// the 8-lane gather groups with ds_bpermute instead of an LDS compaction, and
// each distance is kept by the lane that will own its key, so no LDS staging of
// ids / distances is left.  Two exact overlaps hide the dependent-access
// latencies: the next candidate is known before the last merge of an expansion,
// so its adjacency row is requested first and the merge's rank loop runs under
// that fetch; the merge's scatter is deferred until the next expansion's vector
// loads are in flight.  Results are identical to v1 and to the reference.
// ---------------------------------------------------------------------------
__device__ __forceinline__ uint32_t bperm(uint32_t v, int src_lane)
{
    return (uint32_t)__builtin_amdgcn_ds_bpermute(src_lane << 2, (int)v);
}

// Merge of up to 64 keys (one per lane flagged `take`) into the register-resident sorted list, in
// two halves so that each can run under a different memory wait:
//   merge_rank   how far every old entry moves up (number of new keys below it) and where every
//                new key lands (#old below + #new below)
//   merge_apply  scatter through LDS + read back
// The result equals the reference pushing the neighbours one by one (core.rs:657-664): the list
// ends as the top-`cap` of the union.
template <int R>
__device__ __forceinline__ void merge_rank(const uint64_t (&w)[R], uint64_t nk, bool take, uint32_t (&up)[R],
                                           uint32_t &mypos, int lane)
{
    uint32_t stay[R];                              // new keys ABOVE each old entry (it does not move for those)
#pragma unroll
    for (int r = 0; r < R; ++r) stay[r] = 0;
    mypos = 0;
    const uint64_t mm0 = __ballot(take);
    uint64_t mm = mm0;
    while (mm) {
        const int j = __ffsll((unsigned long long)mm) - 1;
        mm &= mm - 1;
        const uint64_t s = readlane64(nk, j);
        uint32_t rank = (uint32_t)__popcll(__ballot(nk < s) & mm0);
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const bool below = w[r] < s;          // slots past the end hold ~0: never below
            rank += __popcll(__ballot(below));
            stay[r] += below ? 1u : 0u;
        }
        mypos = lane == j ? rank : mypos;
    }
    const uint32_t n = (uint32_t)__popcll(mm0);
#pragma unroll
    for (int r = 0; r < R; ++r) up[r] = n - stay[r];
}

template <int R>
__device__ __forceinline__ uint32_t merge_apply(uint64_t (&w)[R], uint64_t *Wbuf, uint32_t nW, uint32_t cap,
                                                uint64_t nk, bool take, const uint32_t (&up)[R], uint32_t mypos,
                                                int lane, uint64_t *worst = nullptr)
{
    uint32_t total = nW + (uint32_t)__popcll(__ballot(take));
    if (total > cap) total = cap;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const uint32_t i = r * 64 + lane;
        if (i < nW) {
            const uint32_t np = i + up[r];
            if (np < cap) Wbuf[np] = w[r];
        }
    }
    if (take && mypos < cap) Wbuf[mypos] = nk;
    // One wave owns Wbuf and the LDS serves a wave's requests in issue order, so the reads below see
    // the scatter above; only the compiler has to be kept from reordering them.  (A dev_sync()
    // here would also drain the vector loads this merge is meant to run under.)
    lds_order();
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const uint32_t i = r * 64 + lane;
        w[r] = i < total ? Wbuf[i] : ~0ull;
    }
    // the accept threshold of the next expansion (core.rs:651): the cap-th key once the list is full
    if (worst) *worst = total == cap ? Wbuf[cap - 1] : ~0ull;
    lds_order();
    return total;
}

template <int R>
__device__ __forceinline__ uint32_t merge_regs(uint64_t (&w)[R], uint64_t *Wbuf, uint32_t nW, uint32_t cap,
                                               uint64_t nk, bool take, int lane, uint64_t *worst = nullptr)
{
    if (__ballot(take) == 0) return nW;
    uint32_t up[R], mypos;
    merge_rank<R>(w, nk, take, up, mypos, lane);
    return merge_apply<R>(w, Wbuf, nW, cap, nk, take, up, mypos, lane, worst);
}

// Bounded visited table that has stopped recording: a neighbour met again may still be a member of W
// (any other re-met id fails the accept test again, W's furthest only improves).  The same id always has
// the same distance, so a member shows up as an equal key (expanded bit aside): those are dropped here.
template <int R>
__device__ __forceinline__ bool drop_members(const uint64_t (&w)[R], uint64_t key, bool take, int lane)
{
    uint64_t mm = __ballot(take), dup = 0;
    while (mm) {
        const int j = __ffsll((unsigned long long)mm) - 1;
        mm &= mm - 1;
        const uint64_t s = readlane64(key, j) >> 1;
        bool eq = false;
#pragma unroll
        for (int r = 0; r < R; ++r) eq |= (w[r] >> 1) == s;
        if (__ballot(eq)) dup |= 1ull << j;
    }
    return take && !((dup >> lane) & 1ull);
}

// first entry whose expanded bit is clear (slots past nW hold ~0, bit set)
template <int R>
__device__ __forceinline__ bool first_unexpanded(const uint64_t (&w)[R], uint64_t &key, int &rsel, int &lsel)
{
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const uint64_t b = __ballot(!(w[r] & 1ull));
        if (b) {
            lsel = __ffsll((unsigned long long)b) - 1;
            rsel = r;
            key = readlane64(w[r], lsel);
            return true;
        }
    }
    return false;
}

// Squared distances from the query to NR vectors (ids idr[]), AVX2 summation order, for the 8-lane
// group this lane belongs to.  T > 0: query in registers, all T loads per vector issued at once.
// T == 0 (any dim % 32 == 0): query pieces come from LDS and the row is walked 128 floats at a
// time with the accumulators carried -- the per-lane FMA order in t is the same.  hook() runs once,
// right after the first loads have been issued (work to overlap with their latency).
// Storage formats of the vector matrix.  FMT_F32 is the reference's data.  FMT_BF16 / FMT_FP8 are the compressed,
// read-only serving copies (hnsw_set_tuning "compress_bf16" / "compress_fp8"): 2 / 1 bytes per component in the
// gather, widened back to f32 in registers (exactly), after which the arithmetic is the reference's f32 kernel --
// so results are bit-identical to the reference run on the stored (rounded) values.  A piece is the 4 components
// one lane owns of every 32-component block: 16, 8 or 4 bytes.
constexpr int FMT_F32 = 0, FMT_BF16 = 1, FMT_FP8 = 2;
typedef float f32x2_t __attribute__((ext_vector_type(2)));
template <int FMT>
__device__ __forceinline__ float4 load_piece(const float4 *vec4, size_t piece)
{
    if constexpr (FMT == FMT_F32) {
        return vec4[piece];
    } else if constexpr (FMT == FMT_BF16) {
        const uint2 u = reinterpret_cast<const uint2 *>(vec4)[piece];
        return make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xFFFF0000u),
                           __uint_as_float(u.y << 16), __uint_as_float(u.y & 0xFFFF0000u));
    } else {
        const uint32_t u = reinterpret_cast<const uint32_t *>(vec4)[piece];
        const f32x2_t lo = __builtin_amdgcn_cvt_pk_f32_fp8((int)u, false), hi = __builtin_amdgcn_cvt_pk_f32_fp8((int)u, true);
        return make_float4(lo.x, lo.y, hi.x, hi.y);
    }
}

template <int T, int NR, int FMT = FMT_F32, typename Hook>
__device__ __forceinline__ void dist_rounds(const float4 *vec4, uint32_t row4, const uint32_t (&idr)[NR],
                                            const QReg<T> &qr, const float *qlds, int pp, float (&d)[NR],
                                            Hook &&hook)
{
    if constexpr (T > 0) {
        float4 v[NR][T];
#pragma unroll
        for (int rr = 0; rr < NR; ++rr) {
            const size_t p = (size_t)idr[rr] * row4 + pp;
#pragma unroll
            for (int t = 0; t < T; ++t) v[rr][t] = load_piece<FMT>(vec4, p + t * 8);
        }
        hook();
#pragma unroll
        for (int rr = 0; rr < NR; ++rr) d[rr] = avx_reduce(avx_accumulate<T>(qr.q, v[rr])); // core.rs:652
    } else {
        const uint32_t Trt = row4 >> 3;               // 128-byte blocks per row
        const float4 *q4 = reinterpret_cast<const float4 *>(qlds) + pp;
        f32x2 alo[NR], ahi[NR];
#pragma unroll
        for (int rr = 0; rr < NR; ++rr) { alo[rr] = f32x2{0.f, 0.f}; ahi[rr] = f32x2{0.f, 0.f}; }
        bool first = true;
        for (uint32_t t0 = 0; t0 < Trt; t0 += 4) {
            float4 v[NR][4], q[4];
#pragma unroll
            for (int tb = 0; tb < 4; ++tb) {
                if (t0 + tb < Trt) {
                    q[tb] = q4[(t0 + tb) * 8];
#pragma unroll
                    for (int rr = 0; rr < NR; ++rr) v[rr][tb] = load_piece<FMT>(vec4, (size_t)idr[rr] * row4 + pp + (t0 + tb) * 8);
                }
            }
            if (first) { hook(); first = false; }
#pragma unroll
            for (int tb = 0; tb < 4; ++tb) {
                if (t0 + tb < Trt) {
                    const f32x2 qlo = {q[tb].x, q[tb].y}, qhi = {q[tb].z, q[tb].w};
#pragma unroll
                    for (int rr = 0; rr < NR; ++rr) {
                        const f32x2 dlo = qlo - f32x2{v[rr][tb].x, v[rr][tb].y};
                        const f32x2 dhi = qhi - f32x2{v[rr][tb].z, v[rr][tb].w};
                        alo[rr] = __builtin_elementwise_fma(dlo, dlo, alo[rr]);   // metrics.rs:57,60,64,68
                        ahi[rr] = __builtin_elementwise_fma(dhi, dhi, ahi[rr]);
                    }
                }
            }
        }
#pragma unroll
        for (int rr = 0; rr < NR; ++rr)
            d[rr] = avx_reduce(make_float4(alo[rr].x, alo[rr].y, ahi[rr].x, ahi[rr].y));
    }
}

template <int MODE, int T, int R, int FMT = FMT_F32>
__device__ __forceinline__ uint32_t search_level_v2(const GraphView &g, const WaveMem &m, Visited &vis, const QReg<T> &qr,
                                    uint32_t ep, uint32_t ef, uint32_t lc, WorkCtr &ctr, int lane, bool &fail)
{
    static_assert(MODE == MODE_AVX, "AVX2 summation order (dim % 32 == 0)");
    const int grp = lane >> 3, pp = piece_of_lane(lane), sub = lane & 7;
    const float4 *vec4 = reinterpret_cast<const float4 *>(g.vec);
    const uint32_t row4 = g.dim >> 2;                 // float4 per vector row
    const uint32_t stride = lc ? g.strideU : g.stride0;

    visited_clear(vis, lane);                          // core.rs:614
    if (lane == 0) visited_insert(vis, ep);            // core.rs:617
    vis.count = 1;
    // the entry point's row is requested before its distance is computed
    const uint32_t *row = row_ptr(g, ep, lc);
    uint32_t word = (uint32_t)lane < stride ? row[lane] : 0u;
    uint64_t w[R];
    uint64_t ckey;
    {
        const uint32_t id1[1] = {ep};
        float d1[1];
        dist_rounds<T, 1, FMT>(vec4, row4, id1, qr, m.qlds, pp, d1, [] {});   // core.rs:621
        const float d = d1[0];
        ctr.n_dist += 1;
#pragma unroll
        for (int r = 0; r < R; ++r) w[r] = ~0ull;
        ckey = pack_key(d, ep);
        if (lane == 0) w[0] = ckey | 1ull;             // core.rs:627-628; popped right away (:631)
    }
    uint32_t nW = 1;
    // keys of the previous expansion's last gather, not merged yet: the merge runs after the next
    // expansion's vector loads have been issued, i.e. under their latency
    uint64_t pkey = ~0ull;
    bool ptake = false;
    // W's ef-th key once it is full (accept threshold, core.rs:651); with ef = 1 the entry point fills it
    uint64_t worst = ef == 1 ? ckey : ~0ull;
    uint32_t pup[R], ppos = 0;     // its ranks, computed under the row-fetch latency
#pragma unroll
    for (int r = 0; r < R; ++r) pup[r] = 0;
    const uint32_t log_start = ctr.log_n;             // read log of this call (occ_finalize_search_log)
    dev_sync();
    PH_T0();

    for (;;) {
        // ckey is the candidate being expanded (already marked), `word` its adjacency row (core.rs:631-645)
        ctr.n_expand += 1;
        if (ctr.log) {
            // read log: the expanded row and, for now, the popped candidate's distance (thresholds at the end)
            if (lane == 0 && ctr.log_n < ctr.log_cap)
                ctr.log[ctr.log_n] = OccRead{key_id(ckey), occ_meta(lc, OCC_SEARCH, 0, false), (uint32_t)(ckey >> 32)};
            ctr.log_n += 1;
        }
        uint32_t cnt = __builtin_amdgcn_readfirstlane(word);
        if (cnt > stride - 1) cnt = stride - 1;
        ctr.n_ids += cnt;
        PH_MARK(ctr, 0);  // pop + adjacency row fetch

        // The next candidate is known BEFORE the last merge of this expansion: it is the nearest
        // accepted new key if that beats the runner-up (the first unexpanded entry of W), else the
        // runner-up, which no new key can displace in that case.  Its row is requested first and the
        // merge runs under that latency.
        bool next_issued = false, have_next = false;
        uint64_t nkey = ~0ull;
        uint32_t word_next = 0;
        const uint32_t *row_next = row;

        // last keys of an expansion: choose the next candidate now, request its row; the keys stay
        // pending and are merged under the latencies that follow
        auto choose_next = [&](uint64_t key, bool take, bool rkey_known = false, uint64_t rkey_pre = ~0ull) {
            // last merge of this expansion: choose the next candidate now
            uint64_t rkey = rkey_pre;
            if (!rkey_known) {
                int r2, l2;
                if (!first_unexpanded<R>(w, rkey, r2, l2)) rkey = ~0ull;
            }
            uint64_t bm = __ballot(take && key < rkey);
            nkey = rkey;
            while (bm) {
                const int j = __ffsll((unsigned long long)bm) - 1;
                bm &= bm - 1;
                const uint64_t kj = readlane64(key, j);
                if (kj < nkey) nkey = kj;
            }
            have_next = nkey != ~0ull;
            if (have_next) {
                row_next = row_ptr(g, key_id(nkey), lc);
                word_next = (uint32_t)lane < stride ? row_next[lane] : 0u;
            }
            next_issued = true;
            pkey = key;                                                   // merged next expansion
            ptake = take;
            PH_MARK(ctr, 4);  // choose next + request its row
        };

        // one pass = up to 32 fresh neighbours: gather, distances, accept test, then either the choice of
        // the next candidate (last pass of the expansion; its keys stay pending) or an immediate merge
        auto do_pass = [&](uint32_t word_, uint64_t fm_, int shift, int pass, bool is_last) {
            const uint64_t fms = fm_ >> shift;
            const uint32_t pm = (uint32_t)(fms >> (32 * pass));
            if (pm == 0) return;
            constexpr int RB = (T <= 4) ? 4 : (T <= 8 ? 2 : 1);   // rounds of 8 vectors in flight
            uint64_t key = ~0ull;
            bool have = false;
            // Slots without a fresh neighbour re-read the first fresh one (same lines, already in
            // flight): the four rounds then form one straight-line block the compiler interleaves.
            const uint32_t safe_id =
                (uint32_t)__builtin_amdgcn_readlane((int)word_, __ffsll((unsigned long long)fm_) - 1);
#pragma unroll
            for (int r0 = 0; r0 < 4; r0 += RB) {
                uint32_t idr[RB];
#pragma unroll
                for (int rr = 0; rr < RB; ++rr) {
                    const int r = r0 + rr;
                    const int s = pass * 32 + r * 8 + grp;
                    const uint32_t got = bperm(word_, (s + shift) & 63);
                    idr[rr] = ((pm >> (r * 8 + grp)) & 1u) ? got : safe_id;
                }
                float dd[RB];
                dist_rounds<T, RB, FMT>(vec4, row4, idr, qr, m.qlds, pp, dd, [&] {
                    if (r0 == 0 && __ballot(ptake)) {      // deferred scatter, under the loads just issued
                        nW = merge_apply<R>(w, m.W, nW, ef, pkey, ptake, pup, ppos, lane, &worst);
                        ptake = false;
                        PH_MARK(ctr, 3);
                    }
                });
#pragma unroll
                for (int rr = 0; rr < RB; ++rr) {
                    const int r = r0 + rr;
                    if (sub == r && ((pm >> (r * 8 + grp)) & 1u)) { key = pack_key(dd[rr], idr[rr]); have = true; }
                }
            }
            PH_MARK(ctr, 2);  // vector gather + distances
            bool take = have && key < worst;                                  // core.rs:657
            if (vis.lossy) take = drop_members<R>(w, key, take, lane);
            if (is_last) {
                choose_next(key, take);
            } else {
                nW = merge_regs<R>(w, m.W, nW, ef, key, take, lane, &worst);  // core.rs:659-664
                PH_MARK(ctr, 3);  // merge into W
            }
        };

        if (cnt <= 32) {
            // The common case, straight-line: one chunk, one pass (slots come from lanes 1..32).  The
            // vectors of ALL its neighbours are requested the moment the row is here; the visited
            // filter (core.rs:648-649) and the deferred merge run under that latency, and the filter's
            // answer only masks the keys afterwards.  A distance is a pure function of (query, vector),
            // so the result is unchanged; the price is the vectors of already-visited neighbours
            // (n_ids - n_dist, a few per cent of the traffic).
            if (!visited_reserve(vis, lane, &g.hdr->ctr_search[3])) { fail = true; return nW; }
            if (cnt) {
                const bool valid = lane >= 1 && (uint32_t)lane <= cnt;
                const uint32_t pm = cnt >= 32 ? 0xFFFFFFFFu : ((1u << cnt) - 1u);
                constexpr int RB = (T <= 4) ? 4 : (T <= 8 ? 2 : 1);   // rounds of 8 vectors in flight
                const uint32_t safe_id = (uint32_t)__builtin_amdgcn_readlane((int)word, 1);
                uint64_t key = ~0ull, rkey_pre = ~0ull;
                bool have = false, fresh_mine = false;
#pragma unroll
                for (int r0 = 0; r0 < 4; r0 += RB) {
                    uint32_t idr[RB];
#pragma unroll
                    for (int rr = 0; rr < RB; ++rr) {
                        const int s = (r0 + rr) * 8 + grp;
                        const uint32_t got = bperm(word, s + 1);
                        idr[rr] = ((pm >> s) & 1u) ? got : safe_id;
                    }
                    float dd[RB];
                    dist_rounds<T, RB, FMT>(vec4, row4, idr, qr, m.qlds, pp, dd, [&] {
                        if (r0 == 0) {
                            const uint64_t fm = __ballot(visited_insert_wave(vis, valid, word, lane, &g.hdr->ctr_search[3]));
                            const uint32_t nf = __popcll(fm);
                            vis.count += nf;
                            ctr.n_dist += nf;          // the reference evaluates the fresh ones (core.rs:652)
                            fresh_mine = (fm >> ((sub & 3) * 8 + grp + 1)) & 1ull;   // the key this lane will hold
                            PH_MARK(ctr, 1);  // visited filter
                            if (__ballot(ptake)) {      // deferred scatter of the previous expansion's keys
                                nW = merge_apply<R>(w, m.W, nW, ef, pkey, ptake, pup, ppos, lane, &worst);
                                ptake = false;
                                        PH_MARK(ctr, 3);
                            }
                            // W is complete now: its first unexpanded entry is known before the
                            // distances are (one scan less between the vectors and the next row request)
                            int r2, l2;
                            if (!first_unexpanded<R>(w, rkey_pre, r2, l2)) rkey_pre = ~0ull;
                        }
                    });
#pragma unroll
                    for (int rr = 0; rr < RB; ++rr) {
                        const int r = r0 + rr;
                        if (sub == r && ((pm >> (r * 8 + grp)) & 1u)) { key = pack_key(dd[rr], idr[rr]); have = true; }
                    }
                }
                PH_MARK(ctr, 2);  // vector gather + distances
                bool take = have && fresh_mine && key < worst;                    // core.rs:657
                if (vis.lossy) take = drop_members<R>(w, key, take, lane);
                choose_next(key, take, true, rkey_pre);
            }
        } else {
            for (uint32_t wbase = 0; wbase <= cnt; wbase += 64) {   // core.rs:646 stored order
                const uint32_t wi = wbase + lane;
                if (wbase) word = wi < stride ? row[wi] : 0u;
                const bool valid = wi >= 1 && wi <= cnt;
                if (!visited_reserve(vis, lane, &g.hdr->ctr_search[3])) { fail = true; return nW; }
                const bool fresh = visited_insert_wave(vis, valid, word, lane, &g.hdr->ctr_search[3]); // core.rs:648-649
                const uint64_t fm = __ballot(fresh);
                const uint32_t nf = __popcll(fm);
                PH_MARK(ctr, 1);  // visited filter
                if (nf == 0) continue;
                vis.count += nf;
                ctr.n_dist += nf;
                // Slot s of this chunk is lane s; in the first chunk lane 0 holds the degree, so slots
                // are taken from lane s+1 to keep 32 neighbours in 4 rounds.
                const int shift = wbase ? 0 : 1;
                const bool last_chunk = wbase + 64 > cnt;
                const bool two = ((fm >> shift) >> 32) != 0;
                if ((uint32_t)(fm >> shift)) do_pass(word, fm, shift, 0, last_chunk && !two);
                if (two) do_pass(word, fm, shift, 1, last_chunk);
            }
        }
        if (!next_issued) {
            // the last chunk had no unvisited neighbour: nothing is in flight to hide a merge under
            if (__ballot(ptake)) {
                nW = merge_apply<R>(w, m.W, nW, ef, pkey, ptake, pup, ppos, lane, &worst);
                ptake = false;
            }
            int r2, l2;
            have_next = first_unexpanded<R>(w, nkey, r2, l2);
            if (have_next) {
                row_next = row_ptr(g, key_id(nkey), lc);
                word_next = (uint32_t)lane < stride ? row_next[lane] : 0u;
            }
        }
        if (!have_next) break;                                   // core.rs:630,635
        // mark the chosen entry expanded (core.rs:631 pop)
#pragma unroll
        for (int r = 0; r < R; ++r)
            if (w[r] == nkey) w[r] |= 1ull;
        if (ptake && pkey == nkey) pkey |= 1ull;
        // ranks of the pending keys: pure ALU work, runs while the row just requested is in flight
        if (__ballot(ptake)) {
            merge_rank<R>(w, pkey, ptake, pup, ppos, lane);
            PH_MARK(ctr, 5);
        }
        ckey = nkey;
        row = row_next;
        word = word_next;
    }
    // the last expansion's keys were never ranked (the loop ended before that point)
    if (__ballot(ptake)) nW = merge_regs<R>(w, m.W, nW, ef, pkey, ptake, lane, &worst);
    occ_finalize_search_log(ctr, log_start, lc, worst, lane);
    // leave W in LDS for the callers (top-k output, select_neighbors)
#pragma unroll
    for (int r = 0; r < R; ++r) m.W[r * 64 + lane] = w[r];
    dev_sync();
    return nW;
}

template <int MODE, int T, int R, int FMT = FMT_F32>
__device__ __forceinline__ uint32_t search_level(const GraphView &g, const WaveMem &m, Visited &vis,
                                                 const QReg<T> &qr, uint32_t ep, uint32_t ef, uint32_t lc,
                                                 WorkCtr &ctr, int lane, bool &fail)
{
    static_assert(FMT == FMT_F32 || MODE == MODE_AVX, "compressed storage: AVX2 summation order only (dim % 32 == 0)");
    // W in registers up to 16 slices of 64 keys (ef <= 1024); beyond that (R = 64: ef up to 4096) W stays in LDS
    if constexpr (MODE == MODE_AVX && R <= 16) return search_level_v2<MODE, T, R, FMT>(g, m, vis, qr, ep, ef, lc, ctr, lane, fail);
    else {
        static_assert(FMT == FMT_F32 || R <= 16, "compressed storage with ef_construction > 1024 is not built");
        return search_level_v1<MODE, T, R>(g, m, vis, qr, ep, ef, lc, ctr, lane, fail);
    }
}

} // namespace hnsw
