// hnsw_search_trio.hpp -- search_level (core.rs:607-675) with THREE wavefronts per query, for the loneliest shapes:
// one HNSW.SEARCH, a batch of a few hundred queries, one insert plan.
//
// hnsw_search_duo.hpp took W's upkeep off the walking wavefront; what its phase clocks leave on the walker's chain is
// row latency, the vector requests, the VISITED FILTER (~670 clocks of ~3 100 on C1), the distance arithmetic and the
// choice of the next candidate.  The filter does not depend on the vectors, only on the adjacency row -- and the row
// is known the moment the candidate is chosen.  So a third wavefront owns the visited set (core.rs:614, 648-649):
//
//   wave 0, the WALKER:  row, vector requests, distances, accept test (core.rs:657), choice of the next candidate
//                        (:631), request of its row.  No W, no visited set.
//   wave 1, the KEEPER:  W, exactly as in the two-wave form (marks the chosen candidate expanded, merges the accepted
//                        keys, replies with W's ef-th key and its first unexpanded entry).
//   wave 2, the FILTER:  per message it fetches the chosen candidate's row ITSELF (the same 256 bytes the walker
//                        requests, served once from HBM and once from this CU's L1 / L2), runs the test-and-set of
//                        the row's ids against the tag table and replies with the mask of fresh positions.  Ids of
//                        one row are distinct, so one pass over the whole row equals the one-wave kernel's pass
//                        per chunk.
//
// The walker needs both replies when it reaches the accept test of the next expansion; until then all three work in
// parallel.  Nothing is speculative: the accept test sees W's state and the visited set's state exactly as the
// one-wave kernel has them at that point, so W, the expansion order, the answers and the work counters are the
// reference's (the tests compare all forms bit for bit).
//
// Three parties cannot hand over with the workgroup barrier without waiting for each other (a parked wave is released
// ~600 clocks after the last arrival): the hand-overs are sequence numbers in LDS, polled with
// s_sleep between the trips and cut short by s_wakeup from the sender.
//
// As in the two-wave form, a search whose visited set outgrows the table is ABORTED and redone from the start by the
// walker alone (search_level_lean): the filter reports it instead of a mask.
#pragma once
#include "hnsw_search_duo.hpp"

namespace hnsw {

constexpr uint32_t TRIO_NEWROW = 16u;         // with DUO_INIT / DUO_FIN / DUO_ABORT / DUO_EXIT: `cand`'s row is expanded next

struct TrioBox {
    uint64_t key[64];          // walker -> keeper: lane l's accepted key of this chunk, ~0 = none
    uint64_t nkey;             // walker -> keeper: the candidate chosen for the next expansion (mark it expanded), ~0 = none
    uint32_t flags;
    uint32_t cand;             // walker -> filter (TRIO_NEWROW): the node whose row is expanded next; with DUO_INIT: the entry point
    uint32_t lc;               // layer of this search (DUO_INIT)
    uint32_t mseq;             // messages sent
    uint64_t worst, rkey;      // keeper -> walker
    uint32_t nW;
    uint32_t kseq;             // messages the keeper has processed
    uint64_t fm0, fm1;         // filter -> walker: bit p set = row word p (fm1: word 64 + p) is a node not met before
    uint32_t flossy;           // the table stopped recording: abort
    uint32_t fseq;             // messages the filter has processed
};
constexpr size_t kTrioBoxBytes = (sizeof(TrioBox) + 63) & ~(size_t)63;
typedef __attribute__((address_space(3))) volatile TrioBox *TrioBoxLds;
__device__ __forceinline__ TrioBoxLds trio_lds(TrioBox *box) { return (TrioBoxLds)box; }

__device__ __forceinline__ void trio_ping()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_wakeup" ::: "memory");     // after the sequence number has landed
}
__device__ __forceinline__ uint64_t trio_u64(uint64_t v)
{
    return ((uint64_t)__builtin_amdgcn_readfirstlane((uint32_t)(v >> 32)) << 32) | __builtin_amdgcn_readfirstlane((uint32_t)v);
}

// walker: one message for both helpers
__device__ __forceinline__ void trio_send(TrioBox *box, uint32_t &sent, uint64_t key, bool take, uint64_t nkey, uint32_t flags,
                                          uint32_t cand, uint32_t lc, int lane)
{
    TrioBoxLds vb = trio_lds(box);
    asm volatile("" ::: "memory");
    vb->key[lane] = take ? key : ~0ull;
    if (lane == 0) {
        vb->nkey = nkey;
        vb->flags = flags;
        vb->cand = cand;
        vb->lc = lc;
    }
    sent += 1;
    if (lane == 0) vb->mseq = sent;                 // after the data, in issue order
    trio_ping();
}

// walker: the keeper has processed everything sent
__device__ __forceinline__ void trio_wait_keeper(TrioBox *box, uint32_t sent, uint64_t &worst, uint64_t &rkey)
{
    TrioBoxLds vb = trio_lds(box);
    sent = __builtin_amdgcn_readfirstlane(sent);
    uint64_t wv, rv;
    for (;;) {
        const uint32_t sq = vb->kseq;               // issued first: if it is current, so is what follows
        wv = vb->worst;
        rv = vb->rkey;
        if (__builtin_amdgcn_readfirstlane(sq) == sent) break;
        __builtin_amdgcn_s_sleep(2);
    }
    asm volatile("" ::: "memory");
    worst = trio_u64(wv);
    rkey = trio_u64(rv);
}
// walker: the filter has processed everything sent
__device__ __forceinline__ void trio_wait_filter(TrioBox *box, uint32_t sent, uint64_t &fm0, uint64_t &fm1, bool &lossy)
{
    TrioBoxLds vb = trio_lds(box);
    sent = __builtin_amdgcn_readfirstlane(sent);
    uint64_t a, b;
    uint32_t ls;
    for (;;) {
        const uint32_t sq = vb->fseq;
        a = vb->fm0;
        b = vb->fm1;
        ls = vb->flossy;
        if (__builtin_amdgcn_readfirstlane(sq) == sent) break;
        __builtin_amdgcn_s_sleep(2);
    }
    asm volatile("" ::: "memory");
    fm0 = trio_u64(a);
    fm1 = trio_u64(b);
    lossy = __builtin_amdgcn_readfirstlane(ls) != 0u;
}
// helper: a message is there
__device__ __forceinline__ void trio_recv(TrioBox *box, uint32_t &have)
{
    TrioBoxLds vb = trio_lds(box);
    have = __builtin_amdgcn_readfirstlane(have);
    while (__builtin_amdgcn_readfirstlane(vb->mseq) == have) __builtin_amdgcn_s_sleep(2);
    have += 1;
}

// ---- the keeper: W (as duo_keep, polled) -------------------------------------------------------------------
template <int R, bool WIDE>
__device__ __forceinline__ bool trio_keep(const GraphView &g, uint64_t *Wbuf, TrioBox *box, uint32_t ef, uint32_t &seq, int lane)
{
    TrioBoxLds vb = trio_lds(box);
    uint64_t w[R];
#pragma unroll
    for (int r = 0; r < R; ++r) w[r] = ~0ull;
    uint32_t nW = 0;
    uint64_t worst = ~0ull;
    for (;;) {
        trio_recv(box, seq);
        uint64_t kk = vb->key[lane];
        const uint64_t nk = trio_u64(vb->nkey);
        const uint32_t fl = __builtin_amdgcn_readfirstlane(vb->flags);
        if (fl & DUO_EXIT) return false;
        if (fl & DUO_INIT) {
#pragma unroll
            for (int r = 0; r < R; ++r) w[r] = ~0ull;
            nW = 0;
            worst = ~0ull;
        }
        if (!(fl & DUO_ABORT)) {
            bool take = kk != ~0ull;
            if (nk != ~0ull) {
                // core.rs:631 pop: the chosen candidate is in W or among these keys; its low word (id << 1) identifies it
                const uint32_t nlo = (uint32_t)nk;
#pragma unroll
                for (int r = 0; r < R; ++r) w[r] += ((uint32_t)w[r] == nlo) ? 1ull : 0ull;
                kk += (take && (uint32_t)kk == nlo) ? 1ull : 0ull;
            }
            nW = merge_regs_lean<R>(w, Wbuf, nW, ef, kk, take, lane, worst);      // core.rs:659-664
        }
        if (fl & DUO_FIN) {
#pragma unroll
            for (int r = 0; r < R; ++r) Wbuf[r * 64 + lane] = w[r];
            __builtin_amdgcn_wave_barrier();
            if (lane == 0) {
                vb->nW = nW;
                vb->kseq = seq;
            }
            trio_ping();
            return true;
        }
        uint64_t rkey;
        int r2, l2;
        if (!first_unexpanded<R>(w, rkey, r2, l2)) rkey = ~0ull;
        if (lane == 0) {
            vb->worst = worst;
            vb->rkey = rkey;
            vb->kseq = seq;                            // after the data, in issue order
        }
        trio_ping();
    }
}

// ---- the filter: the visited set ----------------------------------------------------------------------------
template <int BB, int DB, bool WIDE>
__device__ __forceinline__ bool trio_filter(const GraphView &g, TagSet<BB, DB> &vis, TrioBox *box, uint32_t &seq, int lane)
{
    TrioBoxLds vb = trio_lds(box);
    uint32_t lc = 0, stride = g.stride0;
    for (;;) {
        trio_recv(box, seq);
        const uint32_t fl = __builtin_amdgcn_readfirstlane(vb->flags);
        if (fl & DUO_EXIT) return false;
        uint64_t fm0 = 0, fm1 = 0;
        if (fl & DUO_INIT) {
            lc = __builtin_amdgcn_readfirstlane(vb->lc);
            stride = lc ? g.strideU : g.stride0;
            tagset_clear<BB, DB>(vis, lane);                                  // core.rs:614
        }
        if ((fl & TRIO_NEWROW) && !(fl & DUO_ABORT)) {
            const uint32_t cand = __builtin_amdgcn_readfirstlane(vb->cand);
            const uint32_t *row = row_ptr(g, cand, lc);
            uint32_t word = (uint32_t)lane < stride ? row[lane] : 0u;       // requested first; the entry point's mark runs under it
            uint32_t word2 = 0u;
            if constexpr (WIDE) word2 = (uint32_t)lane + 64u < stride ? row[lane + 64] : 0u;
            if (fl & DUO_INIT) {
                (void)tagset_visit<BB, DB>(vis, lane == 0, cand);            // core.rs:617
                vis.count = 1;
            }
            uint32_t cnt = __builtin_amdgcn_readfirstlane(word);
            if (cnt > stride - 1) cnt = stride - 1;
            if (!vis.lossy && vis.count + cnt > vis.lcap) vis.lossy = true;  // would stop recording inside this row
            if (!vis.lossy) {
                // core.rs:648-649 for the whole row at once: its ids are distinct, the order inside a row does not matter
                fm0 = __ballot(tagset_visit<BB, DB>(vis, lane >= 1 && (uint32_t)lane <= cnt, word));
                if constexpr (WIDE) {
                    if (cnt >= 64u && !vis.lossy) fm1 = __ballot(tagset_visit<BB, DB>(vis, (uint32_t)lane + 64u <= cnt, word2));
                }
                vis.count += (uint32_t)__popcll(fm0) + (uint32_t)__popcll(fm1);
            }
        }
        if (lane == 0) {
            vb->fm0 = fm0;
            vb->fm1 = fm1;
            vb->flossy = vis.lossy ? 1u : 0u;
            vb->fseq = seq;                            // after the data, in issue order
        }
        trio_ping();
        if (fl & DUO_FIN) return true;
    }
}

// ---- the walker -------------------------------------------------------------------------------------------
// Returns |W| (W itself is in Wbuf once the keeper has answered DUO_FIN), or kEmpty when the visited table stopped
// recording: the caller redoes the search with search_level_lean (ctr is left as it was on entry).
template <class VEC, bool WIDE, bool LOG = false>
__device__ __forceinline__ uint32_t trio_walk(const GraphView &g, uint64_t *Wbuf, TrioBox *box, uint32_t &seq, const typename VEC::Q &qr,
                                              uint32_t ep, uint32_t ef, uint32_t lc, WorkCtr &ctr, int lane)
{
    constexpr int LPV = VEC::LPV, SPR = VEC::SPR, NR = VEC::NR;
    const int grp = lane / LPV, sub = lane % LPV;
    const uint32_t stride = lc ? g.strideU : g.stride0;
    const uint32_t nd0 = ctr.n_dist, ni0 = ctr.n_ids, ne0 = ctr.n_expand, nl0 = ctr.log_n;   // restored on an abort

    const uint32_t *row = row_ptr(g, ep, lc);
    uint32_t word = (uint32_t)lane < stride ? row[lane] : 0u;
    uint32_t word2 = 0u;
    if constexpr (WIDE) word2 = (uint32_t)lane + 64u < stride ? row[lane + 64] : 0u;
    uint64_t ckey;
    {
        typename VEC::V v0;
        VEC::load_v(g, ep, lane, v0);
        const float d = VEC::dist(qr, v0);                        // core.rs:621
        ctr.n_dist += 1;
        ckey = pack_key(d, ep);
    }
    // core.rs:614-628: the visited set starts with the entry point (the filter), W with its key, popped right away (the keeper)
    trio_send(box, seq, ckey | 1ull, lane == 0, ~0ull, DUO_INIT | TRIO_NEWROW, ep, lc, lane);

    const uint32_t log_start = ctr.log_n;
    auto abort_search = [&]() {
        uint64_t wv, rv, a, b;
        bool ls;
        trio_wait_keeper(box, seq, wv, rv);
        trio_wait_filter(box, seq, a, b, ls);
        trio_send(box, seq, ~0ull, false, ~0ull, DUO_FIN | DUO_ABORT, kEmpty, lc, lane);
        trio_wait_keeper(box, seq, wv, rv);
        trio_wait_filter(box, seq, a, b, ls);
        ctr.n_dist = nd0;
        ctr.n_ids = ni0;
        ctr.n_expand = ne0;
        ctr.log_n = nl0;
    };
    for (;;) {
        ctr.n_expand += 1;
        if constexpr (LOG) {
            if (lane == 0 && ctr.log_n < ctr.log_cap)
                ctr.log[ctr.log_n] = OccRead{key_id(ckey), occ_meta(lc, OCC_SEARCH, 0, false), (uint32_t)(ckey >> 32)};
            ctr.log_n += 1;
        }
        uint32_t cnt = __builtin_amdgcn_readfirstlane(word);
        if (cnt > stride - 1) cnt = stride - 1;
        ctr.n_ids += cnt;
        bool have_next = false;
        uint64_t nkey = ~0ull;
        uint32_t word_next = 0, word2_next = 0;
        uint64_t fm0 = 0, fm1 = 0;                                // this row's fresh positions, from the filter
        bool have_fm = false;

        uint32_t c0 = 0;
        constexpr uint32_t CH = (uint32_t)(NR * SPR);
        do {                                                      // chunks of CH ids (core.rs:646 stored order)
            const uint32_t nch = cnt - c0 < CH ? cnt - c0 : CH;
            const bool last = c0 + CH >= cnt;
            uint64_t key = ~0ull;
            bool mine = false;                                    // an empty row (nch == 0) sends only the choice
            if (nch) {
                uint32_t cw, base;
                if constexpr (WIDE) {
                    const uint32_t idx = c0 + 1u + (uint32_t)lane;
                    const uint32_t lo = bperm(word, (int)(idx & 63u)), hi = bperm(word2, (int)(idx & 63u));
                    cw = idx < 64u ? lo : hi;
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
                typename VEC::V v[NR];
#pragma unroll
                for (int r = 0; r < NR; ++r) VEC::load_v(g, idr[r], lane, v[r]);
                // ---- distances (core.rs:652); the visited filter of this row is the third wavefront's ----
                float dsel = 0.f;
                uint32_t idsel = idr[0];
#pragma unroll
                for (int r = 0; r < NR; ++r)
                    if (r == 0 || nch > (uint32_t)(r * SPR)) {
                        const float dr = VEC::dist(qr, v[r]);
                        dsel = (r == 0 || sub == r) ? dr : dsel;
                        idsel = (r == 0 || sub == r) ? idr[r] : idsel;
                    }
                key = pack_key(dsel, idsel);
            }
            if (!have_fm) {
                bool lossy;
                trio_wait_filter(box, seq, fm0, fm1, lossy);
                have_fm = true;
                if (lossy) {
                    abort_search();
                    return kEmpty;
                }
            }
            if (nch) {
                // this chunk covers row positions c0 + 1 .. c0 + nch; slot s of the chunk is position c0 + 1 + s
                const uint32_t myslot = (uint32_t)(sub * SPR + grp);
                const uint32_t pos = c0 + 1u + myslot;
                const uint64_t bit = pos < 64u ? (fm0 >> pos) : (fm1 >> (pos - 64u));
                mine = sub < NR && myslot < nch && (bit & 1ull);
                // the reference evaluates the fresh ones (core.rs:652): fresh positions inside this chunk
                const uint32_t lo = c0 + 1u, hi = c0 + nch;        // inclusive
                uint32_t nf = 0;
                {
                    const uint64_t m0 = lo < 64u ? (fm0 >> lo) << lo : 0ull;
                    const uint64_t m0c = hi < 63u ? m0 & ((2ull << hi) - 1ull) : m0;
                    nf += (uint32_t)__popcll(lo < 64u ? m0c : 0ull);
                    if constexpr (WIDE) {
                        if (hi >= 64u) {
                            const uint32_t l1 = lo > 64u ? lo - 64u : 0u, h1 = hi - 64u;
                            uint64_t m1 = (fm1 >> l1) << l1;
                            if (h1 < 63u) m1 &= (2ull << h1) - 1ull;
                            nf += (uint32_t)__popcll(m1);
                        }
                    }
                }
                ctr.n_dist += nf;
            }
            // everything sent so far is merged: W's ef-th key and its first unexpanded entry, as the one-wave kernel
            // has them at this point
            uint64_t worst, rkey;
            trio_wait_keeper(box, seq, worst, rkey);
            const bool take = mine && key < worst;                // core.rs:657
            if (!last) {
                trio_send(box, seq, key, take, ~0ull, 0u, kEmpty, lc, lane);      // core.rs:659-664, by the keeper
            } else {
                nkey = rkey;
                uint64_t bm = __ballot(take && key < rkey);
                while (bm) {
                    const int j = __ffsll((unsigned long long)bm) - 1;
                    bm &= bm - 1;
                    const uint64_t kj = readlane64(key, j);
                    nkey = kj < nkey ? kj : nkey;
                }
                have_next = nkey != ~0ull;                        // core.rs:630, 635
                if (have_next) {                                  // its row first: the message runs under it
                    row = row_ptr(g, key_id(nkey), lc);
                    word_next = (uint32_t)lane < stride ? row[lane] : 0u;
                    if constexpr (WIDE) word2_next = (uint32_t)lane + 64u < stride ? row[lane + 64] : 0u;
                }
                trio_send(box, seq, key, take, nkey, have_next ? TRIO_NEWROW : DUO_FIN, have_next ? key_id(nkey) : kEmpty, lc, lane);
            }
            c0 += CH;
        } while (c0 < cnt);
        if (!have_next) break;
        ckey = nkey;
        word = word_next;
        word2 = word2_next;
    }
    uint64_t wv, rv, a, b;
    bool ls;
    trio_wait_keeper(box, seq, wv, rv);                           // the keeper has written W out
    trio_wait_filter(box, seq, a, b, ls);
    const uint32_t nW = __builtin_amdgcn_readfirstlane(trio_lds(box)->nW);
    if constexpr (LOG) occ_finalize_search_log(ctr, log_start, lc, nW == ef ? Wbuf[ef - 1] : ~0ull, lane);
    else (void)log_start;
    return nW;
}

// HNSW.SEARCH (core.rs:477-486 -> :865-892): one workgroup of three waves per query.
template <class VEC, int R, int BB, int DB, bool WIDE>
__global__ __launch_bounds__(192, 2) void k_search_trio(GraphView g, const float *__restrict__ Q, uint32_t B, uint32_t k, uint32_t ef,
                                                      uint32_t lcap, uint32_t idbits, uint32_t *__restrict__ out_ids,
                                                      float *__restrict__ out_sims, uint32_t *__restrict__ out_n)
{
    extern __shared__ __align__(16) unsigned char smem[];
    const int lane = threadIdx.x & 63;
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)threadIdx.x) >> 6;   // wave-uniform, and known to be
    uint64_t *Wbuf = reinterpret_cast<uint64_t *>(smem);                       // [LeanW<R>::kSlots]
    TrioBox *box = reinterpret_cast<TrioBox *>(smem + LeanW<R>::kBytes);
    TagSet<BB, DB> vis;
    vis.tab = reinterpret_cast<uint32_t *>(smem + LeanW<R>::kBytes + kTrioBoxBytes);
    vis.idbits = idbits;
    vis.lcap = lcap;
    vis.count = 0;
    vis.lossy = false;
    if (threadIdx.x == 0) {
        box->mseq = 0;
        box->kseq = 0;
        box->fseq = 0;
    }
    __syncthreads();
    uint32_t seq = 0;
    WorkCtr ctr = {};
    const int32_t ep0 = g.hdr->enterpoint;        // core.rs:866
    const uint32_t lmax = g.hdr->max_layer;       // core.rs:867
    for (uint32_t qi = blockIdx.x; qi < B; qi += gridDim.x) {
        uint32_t nW = 0;
        if (wave == 0) {
            typename VEC::Q qr;
            VEC::load_q(Q + (size_t)qi * g.dim, qr, lane);
            uint32_t ep = (uint32_t)ep0;
            for (uint32_t lc = lmax; lc >= 1; --lc) {  // core.rs:870-874: a handful of expansions each, the walker alone
                search_level_lean<VEC, 1, BB, DB, WIDE>(g, Wbuf, vis, qr, ep, 1, lc, ctr, lane, &g.hdr->ctr_search[3]);
                ep = key_id(Wbuf[0]);                  // core.rs:872
                __builtin_amdgcn_wave_barrier();
            }
            nW = trio_walk<VEC, WIDE>(g, Wbuf, box, seq, qr, ep, ef, 0, ctr, lane);   // core.rs:876
            if (nW == kEmpty)
                nW = search_level_lean<VEC, R, BB, DB, WIDE>(g, Wbuf, vis, qr, ep, ef, 0, ctr, lane, &g.hdr->ctr_search[3]);
        } else if (wave == 1) {
            (void)trio_keep<R, WIDE>(g, Wbuf, box, ef, seq, lane);
        } else {
            (void)trio_filter<BB, DB, WIDE>(g, vis, box, seq, lane);
        }
        __syncthreads();
        if (wave == 0) {
            // core.rs:878-890: nearest first, min(k, |W|) results; sim = -dist (metrics.rs:75)
            const uint32_t nres = nW < k ? nW : k;
            for (uint32_t i = lane; i < k; i += 64) {
                const uint64_t key = i < nres ? Wbuf[i] : 0;
                out_ids[(size_t)qi * k + i] = i < nres ? key_id(key) : kEmpty;
                out_sims[(size_t)qi * k + i] = i < nres ? -key_dist(key) : -__builtin_inff();
            }
            if (lane == 0) out_n[qi] = nres;
        }
        __syncthreads();
    }
    if (wave == 0 && lane == 0) {
        atomicAdd(&g.hdr->ctr_search[0], (unsigned long long)ctr.n_dist);
        atomicAdd(&g.hdr->ctr_search[1], (unsigned long long)ctr.n_ids);
        atomicAdd(&g.hdr->ctr_search[2], (unsigned long long)ctr.n_expand);
    }
}

} // namespace hnsw
