// hnsw_search_lean.hpp -- the search kernel's inner loop, specialised for the shapes every BASELINE
// configuration with dim 128 uses (C1, C2, C4): AVX2 summation order with the query in registers (T = dim/32),
// adjacency rows of at most 63 ids (one 256-byte wave load; WIDE variant: 127 ids, two loads), the 16-bit tag table as the visited set with a
// compile-time bucket count, in its BOUNDED form only (a full table stops recording; re-met members of W are
// dropped by key equality -- see search_level_v2 and DESIGN.md 4.1).  Same algorithm, same results and the
// same work counters as search_level_v2; what is gone is everything that made the general routine's
// instruction stream long: the HBM spill path and its three table formats inside the loop, the second code
// path for rows wider than 32, and the closures the compiler spilled scalar registers for.  A row is walked
// in chunks of 40 ids; the rounds of 8 vectors a chunk does not need are skipped (uniform branches), which is
// what the narrow rows of C1 (M=5) want.
#pragma once
#include "hnsw_device.hpp"

namespace hnsw {

template <int BB, int DB = 3>
struct TagSet {
    uint32_t *tab;       // LDS, (1 << BB) buckets of 16 bytes: [u16 arrivals][u16 entry x 7]
    uint32_t idbits;     // ids are < 2^idbits, idbits - BB <= 16 - DB (DB displacement bits per entry: 3, or 2 for 2^24 ids at 1024 buckets)
    uint32_t count;      // ids recorded (wave-uniform)
    uint32_t lcap;       // ... before the table stops recording
    bool lossy;          // wave-uniform
};

template <int BB, int DB>
__device__ __forceinline__ void tagset_clear(TagSet<BB, DB> &v, int lane)
{
    uint4 *t4 = reinterpret_cast<uint4 *>(v.tab);
    const uint4 e = make_uint4(0xFFFF0000u, kEmpty, kEmpty, kEmpty);
#pragma unroll
    for (uint32_t i = 0; i < (1u << BB) / 64; ++i) t4[i * 64 + lane] = e;
    v.count = 0;
    v.lossy = false;
    lds_order();
}

// Test-and-set of one id per lane flagged `valid` (ids of one adjacency row: all distinct).  Returns "was not
// recorded before"; records it unless the table has stopped recording.  An id lives in the first bucket from
// its home bucket on that had room when it arrived, so a lookup may stop at the first bucket that is not full.
template <int BB, int DB>
__device__ __forceinline__ bool tagset_visit(TagSet<BB, DB> &v, bool valid, uint32_t id)
{
    const uint32_t mask = (1u << v.idbits) - 1u;
    const uint32_t x = (id * kTagMul) & mask;
    const uint32_t h = x ^ (x >> ((v.idbits + 1) >> 1));
    constexpr uint32_t bmask = (1u << BB) - 1u;
    const uint32_t b0 = h & bmask, tag = h >> BB;
    uint32_t state = valid ? 3u : 0u;            // 3 unresolved, 1 absent (fresh), 0 present / not asked
    // displacements 0 .. kMaxD-1; the all-ones entry is the free-slot marker, so the largest displacement
    // value of a DB-bit field is never used
    constexpr uint32_t kMaxD = (1u << DB) - 1u;
#pragma unroll 1
    for (uint32_t d = 0; d < kMaxD; ++d) {
        uint32_t *bp = v.tab + (((b0 + d) & bmask) << 2);
        const uint4 wv = *reinterpret_cast<const uint4 *>(bp);
        const uint32_t want = (tag << DB) | d;
        const uint32_t ww = want | (want << 16);
        const bool hit = ((haszero16(wv.y ^ ww) | haszero16(wv.z ^ ww) | haszero16(wv.w ^ ww)) != 0u) | ((wv.x >> 16) == want);
        const bool room = (wv.x & 0xFFFFu) < kBucketIds;
        if (state == 3u) {
            if (hit) state = 0u;
            else if (room) {
                if (v.lossy) state = 1u;
                else {
                    const uint32_t old = atomicAdd(bp, 1u) & 0xFFFFu;      // arrivals: hands out distinct slots
                    if (old < kBucketIds) {
                        reinterpret_cast<unsigned short *>(bp)[1 + old] = (unsigned short)want;
                        state = 1u;
                    }                                                       // else: filled meanwhile, chain on
                }
            }
        }
        if (!__ballot(state == 3u)) break;
    }
    if (__ballot(state == 3u)) v.lossy = true;   // kMaxD full buckets in a row: absent, and recording stops
    return state != 0u;
}

// ---- vector formats ------------------------------------------------------------------------------------
// f32 rows (the reference's data): 8 lanes per vector, each the 16-byte piece piece_of_lane() of every
// 128-byte block; 8 vectors per round, up to 5 rounds per chunk of 40 ids.  AVX2 order of metrics.rs:48-77.
template <int T>
struct VecF32 {
    static constexpr int LPV = 8, SPR = 8, NR = 5;      // 5 rounds: a chunk is 40 ids (rows of 33-40 ids are 10 % of the expansions)
    static constexpr int MIN_WAVES = 2;
    struct Q { float4 q[T]; };
    struct V { float4 v[T]; };
    static __device__ __forceinline__ void load_q(const float *src, Q &q, int lane)
    {
        const float4 *s4 = reinterpret_cast<const float4 *>(src);
        const int pp = piece_of_lane(lane);
#pragma unroll
        for (int t = 0; t < T; ++t) q.q[t] = s4[t * 8 + pp];
    }
    static __device__ __forceinline__ void load_v(const GraphView &g, uint32_t id, int lane, V &v)
    {
        const float4 *p = reinterpret_cast<const float4 *>(g.vec) + (size_t)id * (g.dim >> 2) + piece_of_lane(lane);
#pragma unroll
        for (int t = 0; t < T; ++t) v.v[t] = p[t * 8];
    }
    static __device__ __forceinline__ float dist(const Q &q, const V &v) { return avx_reduce(avx_accumulate<T>(q.q, v.v)); }
};

// bf16 rows (the compressed, read-only serving copy: hnsw_set_tuning "compress_bf16"): half the bytes per
// vector.  The arithmetic is the reference's f32 AVX2 kernel applied to the stored values widened back to
// f32 -- same 4 x 8 accumulators, same FMA order in t, same reduction tree -- so results are bit-identical
// to the reference run on the bf16-rounded vectors.  A 16-byte load is 8 consecutive elements = the 8 AVX
// lanes of ONE accumulator of one 32-element block, so a vector takes 4 lanes (lane a = accumulator a),
// 16 vectors per round, up to 3 rounds per chunk.
template <int T>
struct VecBF16 {
    static constexpr int LPV = 4, SPR = 16, NR = 3;     // a chunk is 48 ids
    static constexpr int MIN_WAVES = 2;                 // per SIMD: the register budget the kernel is compiled for (3 spills: measured 0.6x)
    struct Q { f32x2 q[T][4]; };          // q[t][k] = elements 32t + 8a + 2k, +1
    struct V { uint4 x[T]; };
    static __device__ __forceinline__ void load_q(const float *src, Q &q, int lane)
    {
        const int a = lane & 3;
#pragma unroll
        for (int t = 0; t < T; ++t) {
            const float4 lo = *reinterpret_cast<const float4 *>(src + 32 * t + 8 * a);
            const float4 hi = *reinterpret_cast<const float4 *>(src + 32 * t + 8 * a + 4);
            q.q[t][0] = f32x2{lo.x, lo.y}; q.q[t][1] = f32x2{lo.z, lo.w};
            q.q[t][2] = f32x2{hi.x, hi.y}; q.q[t][3] = f32x2{hi.z, hi.w};
        }
    }
    static __device__ __forceinline__ void load_v(const GraphView &g, uint32_t id, int lane, V &v)
    {
        // row = dim bf16 = dim/8 pieces of 16 bytes; block t is pieces 4t .. 4t+3
        const uint4 *p = reinterpret_cast<const uint4 *>(g.vec) + (size_t)id * (g.dim >> 3) + (lane & 3);
#pragma unroll
        for (int t = 0; t < T; ++t) v.x[t] = p[t * 4];
    }
    static __device__ __forceinline__ f32x2 widen(uint32_t w)      // two bf16 -> two f32 (exact)
    {
        return f32x2{__uint_as_float(w << 16), __uint_as_float(w & 0xFFFF0000u)};
    }
    static __device__ __forceinline__ float dist(const Q &q, const V &v)
    {
        f32x2 acc[4] = {f32x2{0.f, 0.f}, f32x2{0.f, 0.f}, f32x2{0.f, 0.f}, f32x2{0.f, 0.f}};   // AVX lanes j = 2k, 2k+1
#pragma unroll
        for (int t = 0; t < T; ++t) {
            const uint32_t w[4] = {v.x[t].x, v.x[t].y, v.x[t].z, v.x[t].w};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const f32x2 d = q.q[t][k] - widen(w[k]);
                acc[k] = __builtin_elementwise_fma(d, d, acc[k]);                            // metrics.rs:57,60,64,68
            }
        }
        return quad_reduce(acc);
    }
    static __device__ __forceinline__ float quad_reduce(const f32x2 (&acc)[4])
    {
        // (e1+e2)+(e3+e4) per AVX lane (metrics.rs:71-74): the accumulators are the 4 lanes of a quad
        float e[8] = {acc[0].x, acc[0].y, acc[1].x, acc[1].y, acc[2].x, acc[2].y, acc[3].x, acc[3].y};
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            e[j] = __fadd_rn(e[j], dpp_mov<DPP_QUAD_XOR1>(e[j]));
            e[j] = __fadd_rn(e[j], dpp_mov<DPP_QUAD_XOR2>(e[j]));
        }
        // low128 + high128 (metrics.rs:37-39), then (s0+s1)+(s2+s3) (:27-31)
        const float s0 = __fadd_rn(e[0], e[4]), s1 = __fadd_rn(e[1], e[5]), s2 = __fadd_rn(e[2], e[6]), s3 = __fadd_rn(e[3], e[7]);
        return __fadd_rn(__fadd_rn(s0, s1), __fadd_rn(s2, s3));
    }
};

// fp8 rows (e4m3, "compress_fp8"): a quarter of the bytes.  Same layout idea as bf16: lane a of a quad is the
// reference's accumulator a and loads its 8 components of every 32-component block -- 8 bytes; a quad reads 32
// contiguous bytes per block.  v_cvt_pk_f32_fp8 widens two components per instruction (exactly), so the widening costs
// half of what bf16's shift / mask pairs cost; the arithmetic after it is the reference's f32 kernel.
template <int T>
struct VecFP8 {
    static constexpr int LPV = 4, SPR = 16, NR = 3;     // a chunk is 48 ids
    static constexpr int MIN_WAVES = 2;
    struct Q { f32x2 q[T][4]; };          // q[t][k] = elements 32t + 8a + 2k, +1
    struct V { uint2 x[T]; };
    static __device__ __forceinline__ void load_q(const float *src, Q &q, int lane)
    {
        const int a = lane & 3;
#pragma unroll
        for (int t = 0; t < T; ++t) {
            const float4 lo = *reinterpret_cast<const float4 *>(src + 32 * t + 8 * a);
            const float4 hi = *reinterpret_cast<const float4 *>(src + 32 * t + 8 * a + 4);
            q.q[t][0] = f32x2{lo.x, lo.y}; q.q[t][1] = f32x2{lo.z, lo.w};
            q.q[t][2] = f32x2{hi.x, hi.y}; q.q[t][3] = f32x2{hi.z, hi.w};
        }
    }
    static __device__ __forceinline__ void load_v(const GraphView &g, uint32_t id, int lane, V &v)
    {
        // row = dim bytes = dim/8 pieces of 8 bytes; block t is pieces 4t .. 4t+3
        const uint2 *p = reinterpret_cast<const uint2 *>(g.vec) + (size_t)id * (g.dim >> 3) + (lane & 3);
#pragma unroll
        for (int t = 0; t < T; ++t) v.x[t] = p[t * 4];
    }
    static __device__ __forceinline__ float dist(const Q &q, const V &v)
    {
        f32x2 acc[4] = {f32x2{0.f, 0.f}, f32x2{0.f, 0.f}, f32x2{0.f, 0.f}, f32x2{0.f, 0.f}};   // AVX lanes j = 2k, 2k+1
#pragma unroll
        for (int t = 0; t < T; ++t) {
            const f32x2 w[4] = {__builtin_amdgcn_cvt_pk_f32_fp8((int)v.x[t].x, false), __builtin_amdgcn_cvt_pk_f32_fp8((int)v.x[t].x, true),
                                __builtin_amdgcn_cvt_pk_f32_fp8((int)v.x[t].y, false), __builtin_amdgcn_cvt_pk_f32_fp8((int)v.x[t].y, true)};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const f32x2 d = q.q[t][k] - w[k];
                acc[k] = __builtin_elementwise_fma(d, d, acc[k]);                            // metrics.rs:57,60,64,68
            }
        }
        return VecBF16<T>::quad_reduce(acc);
    }
};

// ---- W's merge, straight-line form ---------------------------------------------------------------------
// Same result as merge_apply / merge_regs of hnsw_device.hpp (the list ends as the top-`cap` of the union, the
// reference's one-by-one pushes of core.rs:657-664), without a predicated LDS access: Wbuf carries 64 slots of
// slack and one trash slot, so
//   * every lane scatters all its R entries -- slots past nW hold ~0 with up[r] = (number of new keys), land at
//     or beyond the merged length and so PAD the list with ~0;
//   * entries pushed past `cap` land in [cap, R*64 + 64) and are masked when read back;
//   * lanes without a new key write theirs to the trash slot.
// The accept threshold of the next expansion (core.rs:651) is slot cap-1: a key once the list is full, padding
// (~0 = "accept everything") before.
template <int R>
struct LeanW {
    static constexpr uint32_t kSlots = R * 64 + 64 + 8;        // + slack for the scatter + trash (keeps 64-byte multiples)
    static constexpr uint32_t kTrash = R * 64 + 64;
    static constexpr size_t kBytes = (size_t)kSlots * 8;
};

// TIES (the tie census, DevHeader::ctr_tie).  Every accept test the reference makes while it walks this row one neighbour
// at a time (core.rs:657, e.sim == f.sim with W full) compares an ARRIVING key with the key that is W's last at that
// moment; that key can only have moved outwards by the end of the row, so after the merge the two are neighbours in the
// stretch from W's last slot outwards (or the arrival was rejected against the row's first threshold: counted by the
// caller).  Counted here: arrivals with an equal neighbour there; tracked (ties[1]): the nearest key pushed out of W, for
// the eviction (core.rs:662-664) and the stop test of the search's last pop (core.rs:635).
template <int R, bool TIES = false>
__device__ __forceinline__ uint32_t merge_apply_lean(uint64_t (&w)[R], uint64_t *Wbuf, uint32_t nW, uint32_t cap, uint64_t nk,
                                                     bool take, const uint32_t (&up)[R], uint32_t mypos, uint32_t n_new,
                                                     int lane, uint64_t &worst, uint32_t *ties = nullptr)
{
    uint32_t total = nW + n_new;
    const uint32_t all = total;
    if (total > cap) total = cap;
#pragma unroll
    for (int r = 0; r < R; ++r) Wbuf[(uint32_t)(r * 64 + lane) + up[r]] = w[r];
    Wbuf[take ? mypos : LeanW<R>::kTrash] = nk;
    lds_order();                    // one wave owns Wbuf; the LDS serves it in issue order
    if constexpr (TIES) {
        if (all >= cap) {
            // (i) the nearest key pushed out of W (ties[1]).  If W's LAST key at the end of the search has that distance, which of
            // the two stayed was the heap's choice: either at the eviction itself (two equal furthest entries, one goes:
            // core.rs:662-664 -- expanded or not), or, for a key that is still a candidate, at the stop test of the last pop
            // (core.rs:635, c.sim == f.sim).  The caller compares at the end; a pop against it is tie_pop_test's.
            if (all > cap) ties[1] = min(ties[1], (uint32_t)(Wbuf[cap] >> 32));    // (the stretch beyond W is sorted)
            // (ii) an arriving key next to an equal one at or beyond W's last slot: the accept test it met (or set up for a
            // later arrival of the row) compared equal distances
            bool ev = false;
            if (take) {
                const uint32_t d = (uint32_t)(nk >> 32);
                if (mypos + 1u < all && mypos + 2u >= cap) ev = (uint32_t)(Wbuf[mypos + 1u] >> 32) == d;
                if (mypos >= cap) ev |= (uint32_t)(Wbuf[mypos - 1u] >> 32) == d;
            }
            *ties += (uint32_t)__popcll(__ballot(ev));
        }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const uint64_t v = Wbuf[r * 64 + lane];
        w[r] = (uint32_t)(r * 64 + lane) < cap ? v : ~0ull;
    }
    worst = Wbuf[cap - 1];
    lds_order();
    return total;
}

template <int R, bool TIES = false>
__device__ __forceinline__ uint32_t merge_regs_lean(uint64_t (&w)[R], uint64_t *Wbuf, uint32_t nW, uint32_t cap, uint64_t nk,
                                                    bool take, int lane, uint64_t &worst, uint32_t *ties = nullptr)
{
    const uint64_t mm = __ballot(take);
    if (mm == 0) return nW;
    uint32_t up[R], mypos;
    merge_rank<R>(w, nk, take, up, mypos, lane);
    return merge_apply_lean<R, TIES>(w, Wbuf, nW, cap, nk, take, up, mypos, (uint32_t)__popcll(mm), lane, worst, ties);
}
// tie census, stop test (core.rs:635, c.sim == f.sim): the candidate about to be expanded against W's last key as the
// list stands (every key of the previous expansion merged); W need not be full
template <int R>
__device__ __forceinline__ void tie_stop_test(const uint64_t *Wbuf, uint32_t nW, uint64_t ckey, uint32_t *ties)
{
    if (nW < 2u) return;
    const uint64_t f = Wbuf[nW - 1u];
    if ((uint32_t)(f >> 32) == (uint32_t)(ckey >> 32) && (uint32_t)f >> 1 != (uint32_t)ckey >> 1) *ties += 1u;
}

// tie census, the pop (core.rs:631): 1 if another UNEXPANDED candidate -- in W (w[], bit 0 clear), among the keys accepted
// but not merged yet (pkey / ptake), or the nearest one evicted from W (emin) -- has the chosen key's distance
template <int R>
__device__ __forceinline__ uint32_t tie_pop_test(const uint64_t (&w)[R], uint64_t pkey, bool ptake, uint64_t nkey, uint32_t emin)
{
    const uint32_t nd = (uint32_t)(nkey >> 32), nlo = (uint32_t)nkey & ~1u;
    bool other = ptake && (uint32_t)(pkey >> 32) == nd && ((uint32_t)pkey & ~1u) != nlo;
#pragma unroll
    for (int r = 0; r < R; ++r) other = other || ((uint32_t)(w[r] >> 32) == nd && !(w[r] & 1ull) && (uint32_t)w[r] != nlo);
    return (__ballot(other) != 0ull || emin == nd) ? 1u : 0u;
}

// search_level (core.rs:607-675) with W in registers; leaves w[] sorted (also copied to Wbuf) and returns |W|.
// WIDE: adjacency rows of up to 128 words (127 ids) -- two row words per lane; the ids of a chunk are first
// brought to lanes 0.. of one register (two ds_bpermute + a select per chunk), the rest is the narrow code with
// base 0.  The reference does not bound degrees (core.rs:790-796), M = 32 rows are 112 words, and a restride can
// widen an M = 16 index past 64: those stay on this kernel instead of falling back to the general one.
// LOG: the caller is a plan of the exact-order parallel insert (hnsw_occ.hpp) -- every expanded row goes to its
// read log with the popped candidate's distance, turned into the row's threshold by occ_finalize_search_log.
// TIES: the tie census (DevHeader::ctr_tie) -- ctr.n_tie += the decisions of this search that compared equal distances of
// two different nodes; always on for the plans of an insert (LOG), on request for searches (tuning "tie_census")
template <class VEC, int R, int BB, int DB, bool WIDE, bool LOG = false, bool TIES = LOG>
__device__ __forceinline__ uint32_t search_level_lean(const GraphView &g, uint64_t *Wbuf, TagSet<BB, DB> &vis,
                                                      const typename VEC::Q &qr, uint32_t ep, uint32_t ef, uint32_t lc,
                                                      WorkCtr &ctr, int lane, unsigned long long *lossy_ctr)
{
    constexpr int LPV = VEC::LPV, SPR = VEC::SPR, NR = VEC::NR;   // lanes per vector, vectors per round, rounds per chunk
    const int grp = lane / LPV, sub = lane % LPV;
    const uint32_t stride = lc ? g.strideU : g.stride0;          // <= 64 (WIDE: <= 128): a row is one (two) wave load(s)

    if constexpr (TIES) ctr.tie_emin = 0xFFFFFFFFu;
    tagset_clear<BB, DB>(vis, lane);                                  // core.rs:614
    (void)tagset_visit<BB, DB>(vis, lane == 0, ep);                   // core.rs:617
    vis.count = 1;
    const uint32_t *row = row_ptr(g, ep, lc);
    uint32_t word = (uint32_t)lane < stride ? row[lane] : 0u;     // requested before the distance is computed
    uint32_t word2 = 0u;
    if constexpr (WIDE) word2 = (uint32_t)lane + 64u < stride ? row[lane + 64] : 0u;
    uint64_t w[R];
#pragma unroll
    for (int r = 0; r < R; ++r) w[r] = ~0ull;
    uint64_t ckey;
    {
        typename VEC::V v0;
        VEC::load_v(g, ep, lane, v0);
        const float d = VEC::dist(qr, v0);                        // core.rs:621
        ctr.n_dist += 1;
        ckey = pack_key(d, ep);
        if (lane == 0) w[0] = ckey | 1ull;                        // core.rs:627-628, popped right away (:631)
    }
    uint32_t nW = 1;
    uint64_t worst = ef == 1 ? ckey : ~0ull;                      // W's ef-th key once it is full (core.rs:651)
    uint64_t pkey = ~0ull;                                        // keys of the last chunk, merged one expansion later
    bool ptake = false;
    uint32_t pup[R], ppos = 0, pn = 0;                            // ranks of the pending keys and their count
#pragma unroll
    for (int r = 0; r < R; ++r) pup[r] = 0;

    const uint32_t log_start = ctr.log_n;
    PH_T0();
    for (;;) {
        ctr.n_expand += 1;
        if constexpr (LOG) {
            if (lane == 0 && ctr.log_n < ctr.log_cap)
                ctr.log[ctr.log_n] = OccRead{key_id(ckey), occ_meta(lc, OCC_SEARCH, 0, false), (uint32_t)(ckey >> 32)};
            ctr.log_n += 1;
        }
        uint32_t cnt = __builtin_amdgcn_readfirstlane(word);
        PH_MARK(ctr, 0);  // waiting for the adjacency row
        if (cnt > stride - 1) cnt = stride - 1;
        ctr.n_ids += cnt;
        bool have_next = false;
        uint64_t nkey = ~0ull;
        uint32_t word_next = 0, word2_next = 0;

        uint32_t c0 = 0;
        constexpr uint32_t CH = (uint32_t)(NR * SPR);             // ids per chunk
        do {                                                      // chunks of CH ids (core.rs:646 stored order)
            const uint32_t nch = cnt - c0 < CH ? cnt - c0 : CH;
            const bool last = c0 + CH >= cnt;
            uint64_t key = ~0ull;
            bool take = false;
            uint64_t rkey = ~0ull;                                // first unexpanded entry of W (last chunk only)
            if (nch) {
                // slot s = r*SPR + grp of this chunk sits in lane base + s of cw; empty slots re-read the first id
                uint32_t cw, base;
                if constexpr (WIDE) {
                    const uint32_t idx = c0 + 1u + (uint32_t)lane;                  // row position this lane fetches
                    const uint32_t lo = bperm(word, (int)(idx & 63u)), hi = bperm(word2, (int)(idx & 63u));
                    cw = idx < 64u ? lo : hi;                                        // positions >= 128 are never valid (lane >= nch)
                    base = 0u;
                } else {
                    cw = word;
                    base = c0 + 1u;
                }
                const uint32_t safe = (uint32_t)__builtin_amdgcn_readlane((int)cw, (int)base);
                uint32_t idr[NR];
#pragma unroll
                for (int r = 0; r < NR; ++r) {
                    const uint32_t s = (uint32_t)(r * SPR + grp);
                    const uint32_t got = bperm(cw, (int)((base + s) & 63u));
                    idr[r] = s < nch ? got : safe;
                }
                // all NR rounds are requested: the slots a short chunk does not use re-read its first vector (one
                // line per group, already on its way), which is cheaper than predicating the loads -- the registers
                // of a skipped load would have to be given a value on the other path
                typename VEC::V v[NR];
#pragma unroll
                for (int r = 0; r < NR; ++r) VEC::load_v(g, idr[r], lane, v[r]);
                PH_MARK(ctr, 1);  // chunk set-up, ids to the groups, vector requests
                // ---- under those loads: visited filter (core.rs:648-649) and the deferred merge ----
                if (!vis.lossy && vis.count + CH > vis.lcap) {
                    vis.lossy = true;
                    if (lane == 0) atomicAdd(lossy_ctr, 1ull);
                }
                const uint32_t li = (uint32_t)lane - base;
                const bool was_lossy = vis.lossy;
                const uint64_t fm = __ballot(tagset_visit<BB, DB>(vis, li < nch, cw));
                if (vis.lossy && !was_lossy && lane == 0) atomicAdd(lossy_ctr, 1ull);
                const uint32_t nf = (uint32_t)__popcll(fm);
                vis.count += nf;
                ctr.n_dist += nf;                                 // the reference evaluates the fresh ones (core.rs:652)
                PH_MARK(ctr, 2);  // visited filter
                if (pn) {
                    nW = merge_apply_lean<R, TIES>(w, Wbuf, nW, ef, pkey, ptake, pup, ppos, pn, lane, worst, &ctr.n_tie);
                    ptake = false;
                    pn = 0;
                }
                if constexpr (TIES) { if (c0 == 0) tie_stop_test<R>(Wbuf, nW, ckey, &ctr.n_tie); }
                if (last) {
                    // W is complete: its first unexpanded entry is the next candidate unless one of this chunk's
                    // keys beats it.  (Requesting that entry's row here, one round trip early, was measured:
                    // no gain -- the rank loop already covers the row fetch -- so it is not done.)
                    int r2, l2;
                    if (!first_unexpanded<R>(w, rkey, r2, l2)) rkey = ~0ull;
                }
                PH_MARK(ctr, 3);  // deferred merge + first unexpanded
                // ---- distances ----
                float dsel = 0.f;
                uint32_t idsel = idr[0];
#pragma unroll
                for (int r = 0; r < NR; ++r)
                    if (r == 0 || nch > (uint32_t)(r * SPR)) {
                        const float dr = VEC::dist(qr, v[r]);                                        // core.rs:652
                        dsel = (r == 0 || sub == r) ? dr : dsel;
                        idsel = (r == 0 || sub == r) ? idr[r] : idsel;
                    }
                const uint32_t myslot = (uint32_t)(sub * SPR + grp);
                const bool mine = sub < NR && myslot < nch && ((fm >> ((base + myslot) & 63u)) & 1ull);
                key = pack_key(dsel, idsel);
                take = mine && key < worst;                       // core.rs:657
                if constexpr (TIES)                               // an arrival rejected at W's own last distance
                    ctr.n_tie += (uint32_t)__popcll(__ballot(mine && !take && (uint32_t)(key >> 32) == (uint32_t)(worst >> 32)));
                if (vis.lossy) take = drop_members<R>(w, key, take, lane);
                PH_MARK(ctr, 4);  // waiting for the vectors + distances + accept
            } else {
                if (pn) {
                    nW = merge_apply_lean<R, TIES>(w, Wbuf, nW, ef, pkey, ptake, pup, ppos, pn, lane, worst, &ctr.n_tie);
                    ptake = false;
                    pn = 0;
                }
                if constexpr (TIES) { if (c0 == 0) tie_stop_test<R>(Wbuf, nW, ckey, &ctr.n_tie); }
                int r2, l2;
                if (!first_unexpanded<R>(w, rkey, r2, l2)) rkey = ~0ull;
            }
            if (!last) {
                nW = merge_regs_lean<R, TIES>(w, Wbuf, nW, ef, key, take, lane, worst, &ctr.n_tie);   // core.rs:659-664
            } else {
                // The next candidate is known before these keys are merged: the nearest accepted new key if
                // it beats the first unexpanded entry of W, else that entry.  Its row is requested now; the
                // ranks of the pending keys are computed under that latency, the scatter one expansion later.
                nkey = rkey;
                uint64_t bm = __ballot(take && key < rkey);
                while (bm) {
                    const int j = __ffsll((unsigned long long)bm) - 1;
                    bm &= bm - 1;
                    const uint64_t kj = readlane64(key, j);
                    nkey = kj < nkey ? kj : nkey;
                }
                have_next = nkey != ~0ull;
                if (have_next) {
                    row = row_ptr(g, key_id(nkey), lc);
                    word_next = (uint32_t)lane < stride ? row[lane] : 0u;
                    if constexpr (WIDE) word2_next = (uint32_t)lane + 64u < stride ? row[lane + 64] : 0u;
                }
                pkey = key;
                ptake = take;
                PH_MARK(ctr, 5);  // choice of the next candidate + its row request
            }
            c0 += CH;
        } while (c0 < cnt);

        if (!have_next) break;                                    // core.rs:630,635
        // tie census, the pop itself (core.rs:631): C holds another candidate at the popped one's distance -- an unexpanded
        // entry of W, a pending key, or the nearest one that fell out of W -- and which of the two is expanded first is the
        // heap's choice (the other may never be: the expansion can move W's end past both)
        if constexpr (TIES) ctr.n_tie += tie_pop_test<R>(w, pkey, ptake, nkey, ctr.tie_emin);
        // mark the chosen entry expanded (core.rs:631 pop).  Ids are unique in W and among the pending keys, and the
        // chosen key is unexpanded: its low word (id << 1) identifies it, and adding the match sets bit 0
        {
            const uint32_t nlo = (uint32_t)nkey;
#pragma unroll
            for (int r = 0; r < R; ++r) w[r] += ((uint32_t)w[r] == nlo) ? 1ull : 0ull;
            pkey += (ptake && (uint32_t)pkey == nlo) ? 1ull : 0ull;
        }
        pn = (uint32_t)__popcll(__ballot(ptake));
        if (pn) merge_rank<R>(w, pkey, ptake, pup, ppos, lane);
        PH_MARK(ctr, 6);  // mark expanded + ranks of the pending keys
        ckey = nkey;
        word = word_next;
        word2 = word2_next;
    }
    nW = merge_regs_lean<R, TIES>(w, Wbuf, nW, ef, pkey, ptake, lane, worst, &ctr.n_tie);
    // tie census, the last pop (core.rs:631-635): the nearest candidate left is one that fell out of W; equal to W's last = a tie
    if constexpr (TIES) { if (nW == ef && ctr.tie_emin == (uint32_t)(worst >> 32)) ctr.n_tie += 1u; }
#pragma unroll
    for (int r = 0; r < R; ++r) Wbuf[r * 64 + lane] = w[r];
    lds_order();
    // tie census: W's two nearest members at equal distances -- which one is the next layer's entry point (core.rs:514,
    // :576, :872) is the heap's choice
    if constexpr (TIES) { if (nW >= 2u && (uint32_t)(Wbuf[0] >> 32) == (uint32_t)(Wbuf[1] >> 32)) ctr.n_tie += 1u; }
    (void)ckey;
    if constexpr (LOG) occ_finalize_search_log(ctr, log_start, lc, nW == ef ? Wbuf[ef - 1] : ~0ull, lane);
    else (void)log_start;
    return nW;
}

// HNSW.SEARCH (core.rs:477-486 -> :865-892): one wave per query, grid-stride over the batch.
template <class VEC, int R, int BB, int DB, bool WIDE, bool TIES = false>
__global__ __launch_bounds__(64, VEC::MIN_WAVES) void k_search_lean(GraphView g, const float *__restrict__ Q, uint32_t B, uint32_t k,
                                                    uint32_t ef, uint32_t lcap, uint32_t idbits,
                                                    uint32_t *__restrict__ out_ids, float *__restrict__ out_sims,
                                                    uint32_t *__restrict__ out_n, uint32_t *__restrict__ tie_flags = nullptr)
{
    // tie_flags (TIES kernels, tuning tie_mode): [B] 1 = this query met a tie (it is answered again in std's heap order)
    extern __shared__ __align__(16) unsigned char smem[];
    const int lane = threadIdx.x;
    uint64_t *Wbuf = reinterpret_cast<uint64_t *>(smem);                       // [LeanW<R>::kSlots]
    TagSet<BB, DB> vis;
    vis.tab = reinterpret_cast<uint32_t *>(smem + LeanW<R>::kBytes);
    vis.idbits = idbits;
    vis.lcap = lcap;
    vis.count = 0;
    vis.lossy = false;
    WorkCtr ctr = {};
    const int32_t ep0 = g.hdr->enterpoint;        // core.rs:866
    const uint32_t lmax = g.hdr->max_layer;       // core.rs:867
    for (uint32_t qi = blockIdx.x; qi < B; qi += gridDim.x) {
        typename VEC::Q qr;
        VEC::load_q(Q + (size_t)qi * g.dim, qr, lane);
        uint32_t ep = (uint32_t)ep0;
        const uint32_t tie0 = ctr.n_tie;
        for (uint32_t lc = lmax; lc >= 1; --lc) {  // core.rs:870-874
            search_level_lean<VEC, 1, BB, DB, WIDE, false, TIES>(g, Wbuf, vis, qr, ep, 1, lc, ctr, lane, &g.hdr->ctr_search[3]);
            ep = key_id(Wbuf[0]);                  // core.rs:872
            lds_order();
        }
        const uint32_t nW = search_level_lean<VEC, R, BB, DB, WIDE, false, TIES>(g, Wbuf, vis, qr, ep, ef, 0, ctr, lane, &g.hdr->ctr_search[3]); // core.rs:876
        // core.rs:878-890: nearest first, min(k, |W|) results; sim = -dist (metrics.rs:75)
        const uint32_t nres = nW < k ? nW : k;
        if constexpr (TIES) {
            // equal distances among the k + 1 nearest: which of them is returned, and in which order, is the heap's business
            for (uint32_t i0 = 0; i0 < nres && i0 + 1u < nW; i0 += 64u) {
                const uint32_t i = i0 + (uint32_t)lane;
                const bool in = i < nres && i + 1u < nW;
                ctr.n_tie += (uint32_t)__popcll(__ballot(in && (uint32_t)(Wbuf[in ? i : 0u] >> 32) == (uint32_t)(Wbuf[in ? i + 1u : 0u] >> 32)));
            }
            if (lane == 0 && ctr.n_tie != tie0) atomicAdd(&g.hdr->ctr_tie[1], 1ull);
            if (lane == 0 && tie_flags) tie_flags[qi] = ctr.n_tie != tie0 ? 1u : 0u;
        }
        for (uint32_t i = lane; i < k; i += 64) {
            const uint64_t key = i < nres ? Wbuf[i] : 0;
            out_ids[(size_t)qi * k + i] = i < nres ? key_id(key) : kEmpty;
            out_sims[(size_t)qi * k + i] = i < nres ? -key_dist(key) : -__builtin_inff();
        }
        if (lane == 0) out_n[qi] = nres;
        lds_order();
    }
    if (lane == 0) {
        atomicAdd(&g.hdr->ctr_search[0], (unsigned long long)ctr.n_dist);
        atomicAdd(&g.hdr->ctr_search[1], (unsigned long long)ctr.n_ids);
        atomicAdd(&g.hdr->ctr_search[2], (unsigned long long)ctr.n_expand);
        if constexpr (TIES) atomicAdd(&g.hdr->ctr_tie[0], (unsigned long long)ctr.n_tie);
#ifdef HNSW_PHASE_TIMERS
        for (int i = 0; i < 7; ++i) atomicAdd(&g.hdr->prof[i], ctr.ph[i]);
#endif
    }
}

} // namespace hnsw
