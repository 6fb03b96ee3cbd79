// hnsw_search_duo.hpp -- search_level (core.rs:607-675) for a query that has the chip (almost) to itself: TWO
// wavefronts per query, on two SIMDs of one CU.
//
// With fewer waves than SIMDs -- one HNSW.SEARCH, one HNSW.NODE.ADD's plan, a lone launch of <= 1024 queries -- a
// wavefront issues one instruction every ~8 clocks whatever the instruction, and an expansion of
// search_level_lean is ~430-640 of them; more than half keep W in order (ranks of the accepted keys, the scatter,
// the first unexpanded entry: DESIGN.md 4.1c).  None of that is needed by the walk itself until it has to choose the
// next candidate, a full expansion later.  So the work is split:
//
//   wave 0, the WALKER: the reference's loop in the reference's order -- adjacency row (core.rs:642-646), visited
//     filter (:648-649), vector requests, distances (:652), accept test against W's ef-th key (:657), choice of the
//     next candidate (:631), request of its row.  It holds no W.
//   wave 1, the KEEPER: owns W (sorted keys in registers, as in the one-wave kernel).  Per message from the walker
//     -- the accepted keys of a chunk, and the candidate chosen for the next expansion -- it marks that candidate
//     expanded, merges the keys (merge_rank + scatter), finds the first unexpanded entry, and publishes
//     (W's ef-th key, first unexpanded entry).
//
// The walker sends expansion i's keys and starts expansion i+1 at once; it needs the keeper's answer only when it
// gets to the accept test of i+1, by which time the keeper has had the row fetch, the vector requests, the filter and
// the distance arithmetic of i+1 to work under.  Nothing is speculative and nothing is stale: the accept test and
// the choice use exactly the state the one-wave kernel has at that point (all earlier keys merged), so W, the
// expansion order, results and the work counters are the reference's -- the same as search_level_lean's, which the
// tests compare it with bit for bit.
//
// The mailbox is a few words of LDS; how the waves hand over is HNSW_DUO_SYNC below.
//
// One thing the walker cannot do without W: once the bounded visited table stops recording, re-met members of W are
// recognised by key equality (drop_members).  The two-wave form runs with the 2048-bucket table (14 336 ids); a
// search that outgrows it is ABORTED and redone from the start by the walker alone with search_level_lean (exact as
// ever); the counters of the aborted attempt are discarded.
#pragma once
#include "hnsw_search_lean.hpp"

namespace hnsw {

constexpr uint32_t DUO_INIT = 1u, DUO_FIN = 2u, DUO_ABORT = 4u, DUO_EXIT = 8u;

struct DuoBox {
    uint64_t key[64];          // walker -> keeper: lane l's accepted key of this chunk, ~0 = none
    uint64_t nkey;             // walker -> keeper: the candidate the walker expands next (mark it expanded), ~0 = none
    uint32_t flags;            // DUO_INIT: reset W first (nkey's low word = the layer); DUO_FIN: this search is over (write W out);
                               // DUO_EXIT (alone): no more searches, the keeper leaves
    uint32_t mseq;             // messages sent so far (polled forms)
    uint64_t worst;            // keeper -> walker: W's ef-th key (~0 while W is not full)  core.rs:651
    uint64_t rkey;             // keeper -> walker: W's first unexpanded entry, ~0 = none
    uint32_t nW;               // |W| (read by the walker after DUO_FIN)
    uint32_t sseq;             // messages processed so far (polled forms)
    uint32_t ties, pad;        // keeper -> walker with DUO_FIN: the tie census of this search's merges (TIES keepers)
};
constexpr size_t kDuoBoxBytes = (sizeof(DuoBox) + 63) & ~(size_t)63;
// The mailbox is polled: its accesses are volatile, and they must stay LDS instructions.  A volatile access through a
// generic pointer is compiled as a FLAT load with system scope and a wait for EVERY outstanding vector load of the
// wave (the gathers the polling is supposed to run under): the pointer carries the LDS address space explicitly.
typedef __attribute__((address_space(3))) volatile DuoBox *DuoBoxLds;
__device__ __forceinline__ DuoBoxLds duo_lds(DuoBox *box) { return (DuoBoxLds)box; }

// How the two waves hand over (HNSW_DUO_SYNC; measured on C1 / a lone 1024-query launch):
//   0  the workgroup barrier: a parked wave costs its SIMD nothing, but is released ~600 clocks after the other one's
//      arrival;
//   1  a spin on a sequence number in LDS (read, compare, branch);
//   2  the same poll with s_sleep between the trips and s_wakeup from the sender, which ends the sleep at once.
// Every message has exactly one reply and the walker collects it before it sends the next one.  LDS instructions of
// one wave execute in issue order, so "data, then sequence number" needs no fence -- only `volatile` against the
// compiler and the LDS address space on the pointer (above).
#ifndef HNSW_DUO_SYNC
#define HNSW_DUO_SYNC 0
#endif
struct DuoSeq {
    uint32_t n;                // walker: messages sent; keeper: messages processed (both waves count alike)
};

__device__ __forceinline__ void duo_bar()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // the LDS writes before it have landed
}
__device__ __forceinline__ void duo_pause()
{
#if HNSW_DUO_SYNC == 2
    __builtin_amdgcn_s_sleep(4);
#endif
}
__device__ __forceinline__ void duo_ping()
{
#if HNSW_DUO_SYNC == 2
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_wakeup" ::: "memory");   // after the sequence number has landed
#endif
}

// walker: the keeper has processed the message sent last: take its answer
__device__ __forceinline__ void duo_wait(DuoBox *box, DuoSeq &seq, uint64_t &worst, uint64_t &rkey)
{
    DuoBoxLds vb = duo_lds(box);
#if HNSW_DUO_SYNC == 0
    duo_bar();                                     // B2
    const uint64_t wv = vb->worst, rv = vb->rkey;
#else
    const uint32_t sent = __builtin_amdgcn_readfirstlane(seq.n);   // wave-uniform: say so, or the poll is compiled as a divergent loop
    uint64_t wv, rv;
    for (;;) {
        const uint32_t sq = vb->sseq;              // issued first: if it is current, so is what follows
        wv = vb->worst;
        rv = vb->rkey;
        if (__builtin_amdgcn_readfirstlane(sq) == sent) break;
        duo_pause();
    }
    asm volatile("" ::: "memory");                 // nothing of what follows (Wbuf reads after DUO_FIN) moves above the poll
#endif
    worst = ((uint64_t)__builtin_amdgcn_readfirstlane((uint32_t)(wv >> 32)) << 32) | __builtin_amdgcn_readfirstlane((uint32_t)wv);
    rkey = ((uint64_t)__builtin_amdgcn_readfirstlane((uint32_t)(rv >> 32)) << 32) | __builtin_amdgcn_readfirstlane((uint32_t)rv);
}

__device__ __forceinline__ void duo_send(DuoBox *box, DuoSeq &seq, uint64_t key, bool take, uint64_t nkey, uint32_t flags, int lane)
{
    DuoBoxLds vb = duo_lds(box);
    asm volatile("" ::: "memory");
    vb->key[lane] = take ? key : ~0ull;
    if (lane == 0) {
        vb->nkey = nkey;
        vb->flags = flags;
    }
    seq.n += 1;
#if HNSW_DUO_SYNC == 0
    duo_bar();                                     // B1
#else
    if (lane == 0) vb->mseq = seq.n;               // after the data, in issue order
    duo_ping();
#endif
}

// keeper: a message is there
__device__ __forceinline__ void duo_recv(DuoBox *box, DuoSeq &seq)
{
#if HNSW_DUO_SYNC == 0
    duo_bar();                                     // B1
#else
    DuoBoxLds vb = duo_lds(box);
    const uint32_t have = __builtin_amdgcn_readfirstlane(seq.n);
    while (__builtin_amdgcn_readfirstlane(vb->mseq) == have) duo_pause();
#endif
    seq.n += 1;
}

// keeper: the reply is written
__device__ __forceinline__ void duo_reply(DuoBox *box, DuoSeq &seq, int lane)
{
#if HNSW_DUO_SYNC == 0
    duo_bar();                                     // B2
#else
    DuoBoxLds vb = duo_lds(box);
    if (lane == 0) vb->sseq = seq.n;               // after the data, in issue order
    duo_ping();
#endif
}

// ---- the keeper ------------------------------------------------------------------------------------------
// Serves ONE search_level: from its DUO_INIT message to its DUO_FIN.  Leaves W sorted in Wbuf[0 .. nW) and returns
// true; returns false if the first message was DUO_EXIT instead.
template <int R, bool WIDE, bool TIES = false>
__device__ __forceinline__ bool duo_keep(const GraphView &g, uint64_t *Wbuf, DuoBox *box, uint32_t ef, DuoSeq &seq, int lane,
                                         WorkCtr &ctr)
{
    uint32_t ties[2] = {0u, 0xFFFFFFFFu};              // tie census of this search (merge_apply_lean, tie_stop_test): count, nearest evicted candidate
    PH_T0();
    DuoBoxLds vb = duo_lds(box);
    uint32_t lc = 0, stride = g.stride0;
    uint64_t w[R];
#pragma unroll
    for (int r = 0; r < R; ++r) w[r] = ~0ull;
    uint32_t nW = 0;
    uint64_t worst = ~0ull;
    uint32_t warm = 0;                                 // what the row prefetches brought (never looked at)
    for (;;) {
        duo_recv(box, seq);
        PH_MARK(ctr, 7);  // keeper: waiting for a message
        uint64_t kk = vb->key[lane];
        const uint64_t nkv = vb->nkey;
        const uint32_t fl = __builtin_amdgcn_readfirstlane(vb->flags);
        const uint64_t nk = ((uint64_t)__builtin_amdgcn_readfirstlane((uint32_t)(nkv >> 32)) << 32) | __builtin_amdgcn_readfirstlane((uint32_t)nkv);
        if (fl & DUO_EXIT) return false;
        if (fl & DUO_INIT) {
#pragma unroll
            for (int r = 0; r < R; ++r) w[r] = ~0ull;
            nW = 0;
            worst = ~0ull;
            lc = (uint32_t)nk;
            stride = lc ? g.strideU : g.stride0;
        }
        if (!(fl & DUO_ABORT)) {
            bool take = kk != ~0ull;
            if (nk != ~0ull && !(fl & DUO_INIT)) {
                // core.rs:631 pop: the chosen candidate is in W or among these keys; ids are unique, the chosen key is
                // unexpanded: its low word (id << 1) identifies it, adding the match sets bit 0
                const uint32_t nlo = (uint32_t)nk;
                if constexpr (TIES) ties[0] += tie_pop_test<R>(w, kk, take, nk, ties[1]);   // the pop itself (core.rs:631), see search_level_lean
#pragma unroll
                for (int r = 0; r < R; ++r) w[r] += ((uint32_t)w[r] == nlo) ? 1ull : 0ull;
                kk += (take && (uint32_t)kk == nlo) ? 1ull : 0ull;
            }
            nW = merge_regs_lean<R, TIES>(w, Wbuf, nW, ef, kk, take, lane, worst, ties);      // core.rs:659-664
            // the candidate the walker expands next, against W's last key with everything merged (core.rs:635)
            if constexpr (TIES) { if (nk != ~0ull && !(fl & DUO_INIT)) tie_stop_test<R>(Wbuf, nW, nk, ties); }
        }
        if (fl & DUO_FIN) {
#pragma unroll
            for (int r = 0; r < R; ++r) Wbuf[r * 64 + lane] = w[r];
            lds_order();
            if (lane == 0) vb->nW = nW + (warm == 0x9E3779B9u && lane == 64 ? 1u : 0u);   // (keeps the prefetches alive)
            if constexpr (TIES) { if (nW == ef && ties[1] == (uint32_t)(worst >> 32)) ties[0] += 1u; }   // the last pop (core.rs:635)
            if constexpr (TIES) { if (nW >= 2u && (uint32_t)(Wbuf[0] >> 32) == (uint32_t)(Wbuf[1] >> 32)) ties[0] += 1u; }   // the entry point of the next layer
            if (lane == 0) vb->ties = ties[0];
            PH_MARK(ctr, 6);
            duo_reply(box, seq, lane);
            return true;
        }
        uint64_t rkey;
        int r2, l2;
        if (!first_unexpanded<R>(w, rkey, r2, l2)) rkey = ~0ull;
        if (lane == 0) {
            vb->worst = worst;
            vb->rkey = rkey;
        }
        PH_MARK(ctr, 6);  // keeper: mark + ranks + scatter + first unexpanded + publish
        duo_reply(box, seq, lane);
        // W's first unexpanded entry is the walker's next candidate unless one of the keys it is evaluating right now
        // beats it (about every second time): touch that entry's adjacency row, so that the walker's own request --
        // one dependent round trip on its critical path -- finds it in this CU's L1 / in L2.  One 256-byte row per
        // expansion next to ~20 vectors of 512 bytes; the value is never used.
        if (rkey != ~0ull) {
            const uint32_t *prow = row_ptr(g, key_id(rkey), lc);
            warm ^= (uint32_t)lane < stride ? prow[lane] : 0u;
            if constexpr (WIDE) warm ^= (uint32_t)lane + 64u < stride ? prow[lane + 64] : 0u;
        }
    }
}

// ---- the walker ------------------------------------------------------------------------------------------
// Returns |W| (W itself is in Wbuf once the keeper has answered DUO_FIN), or kEmpty when the visited table stopped
// recording: the caller redoes the search with search_level_lean (ctr is left as it was on entry).
template <class VEC, int BB, int DB, bool WIDE, bool LOG = false, bool TIES = LOG>
__device__ __forceinline__ uint32_t duo_walk(const GraphView &g, uint64_t *Wbuf, DuoBox *box, DuoSeq &seq, TagSet<BB, DB> &vis,
                                             const typename VEC::Q &qr, uint32_t ep, uint32_t ef, uint32_t lc, WorkCtr &ctr,
                                             int lane)
{
    constexpr int LPV = VEC::LPV, SPR = VEC::SPR, NR = VEC::NR;
    const int grp = lane / LPV, sub = lane % LPV;
    const uint32_t stride = lc ? g.strideU : g.stride0;
    const uint32_t nd0 = ctr.n_dist, ni0 = ctr.n_ids, ne0 = ctr.n_expand, nl0 = ctr.log_n;   // restored on an abort

    tagset_clear<BB, DB>(vis, lane);                                  // core.rs:614
    (void)tagset_visit<BB, DB>(vis, lane == 0, ep);                   // core.rs:617
    vis.count = 1;
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
    duo_send(box, seq, ckey | 1ull, lane == 0, (uint64_t)lc, DUO_INIT, lane);   // core.rs:627-628, popped right away (:631)

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
                PH_MARK(ctr, 1);  // chunk set-up, ids to the groups, vector requests
                // ---- under those loads: the visited filter (core.rs:648-649) ----
                if (!vis.lossy && vis.count + CH > vis.lcap) vis.lossy = true;
                const uint32_t li = (uint32_t)lane - base;
                uint64_t fm = 0;
                if (!vis.lossy) fm = __ballot(tagset_visit<BB, DB>(vis, li < nch, cw));
                if (vis.lossy) {
                    // the table stopped recording: this form cannot tell re-met members of W apart.  Drain the loads,
                    // close the keeper's search and hand the whole search_level back to the one-wave routine.
                    uint64_t wv, rv;
                    duo_wait(box, seq, wv, rv);
                    duo_send(box, seq, ~0ull, false, ~0ull, DUO_FIN | DUO_ABORT, lane);
                    duo_wait(box, seq, wv, rv);
                    ctr.n_dist = nd0;
                    ctr.n_ids = ni0;
                    ctr.n_expand = ne0;
                    ctr.log_n = nl0;
                    return kEmpty;
                }
                const uint32_t nf = (uint32_t)__popcll(fm);
                vis.count += nf;
                ctr.n_dist += nf;                                 // the reference evaluates the fresh ones (core.rs:652)
                PH_MARK(ctr, 2);  // visited filter
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
                mine = sub < NR && myslot < nch && ((fm >> ((base + myslot) & 63u)) & 1ull);
                key = pack_key(dsel, idsel);
                PH_MARK(ctr, 3);  // waiting for the vectors + distances
            }
            // everything sent so far is merged: W's ef-th key and its first unexpanded entry, as the one-wave kernel
            // has them at this point
            uint64_t worst, rkey;
            duo_wait(box, seq, worst, rkey);
            PH_MARK(ctr, 4);  // waiting for the keeper
            const bool take = mine && key < worst;                // core.rs:657
            if constexpr (TIES)                                   // an arrival rejected at W's own last distance
                ctr.n_tie += (uint32_t)__popcll(__ballot(mine && !take && (uint32_t)(key >> 32) == (uint32_t)(worst >> 32)));
            if (!last) {
                duo_send(box, seq, key, take, ~0ull, 0u, lane);        // core.rs:659-664, by the keeper
            } else {
                // the next candidate (core.rs:631): the nearest accepted new key if it beats W's first unexpanded entry,
                // else that entry
                nkey = rkey;
                uint64_t bm = __ballot(take && key < rkey);
                while (bm) {
                    const int j = __ffsll((unsigned long long)bm) - 1;
                    bm &= bm - 1;
                    const uint64_t kj = readlane64(key, j);
                    nkey = kj < nkey ? kj : nkey;
                }
                have_next = nkey != ~0ull;                        // core.rs:630, 635
                if (have_next) {                                  // its row first: the message's ~70 instructions run under it
                    row = row_ptr(g, key_id(nkey), lc);
                    word_next = (uint32_t)lane < stride ? row[lane] : 0u;
                    if constexpr (WIDE) word2_next = (uint32_t)lane + 64u < stride ? row[lane + 64] : 0u;
                }
                duo_send(box, seq, key, take, nkey, have_next ? 0u : DUO_FIN, lane);
            }
            PH_MARK(ctr, 5);  // accept, choice of the next candidate, message, row request
            c0 += CH;
        } while (c0 < cnt);
        if (!have_next) break;
        ckey = nkey;
        word = word_next;
        word2 = word2_next;
    }
    uint64_t wv, rv;
    duo_wait(box, seq, wv, rv);                                        // the keeper has written W out
    const uint32_t nW = __builtin_amdgcn_readfirstlane(duo_lds(box)->nW);
    if constexpr (TIES) ctr.n_tie += __builtin_amdgcn_readfirstlane(duo_lds(box)->ties);
    if constexpr (LOG) occ_finalize_search_log(ctr, log_start, lc, nW == ef ? Wbuf[ef - 1] : ~0ull, lane);
    else (void)log_start;
    return nW;
}

// HNSW.SEARCH (core.rs:477-486 -> :865-892): one workgroup of two waves per query.
template <class VEC, int R, int BB, int DB, bool WIDE>
__global__ __launch_bounds__(128, 2) void k_search_duo(GraphView g, const float *__restrict__ Q, uint32_t B, uint32_t k, uint32_t ef,
                                                     uint32_t lcap, uint32_t idbits, uint32_t *__restrict__ out_ids,
                                                     float *__restrict__ out_sims, uint32_t *__restrict__ out_n)
{
    extern __shared__ __align__(16) unsigned char smem[];
    const int lane = threadIdx.x & 63;
    // wave-uniform, and the compiler must know it: under a branch it takes for divergent every loop of the walker and
    // the keeper would be compiled with exec-mask bookkeeping instead of scalar branches
    const bool walker = __builtin_amdgcn_readfirstlane((int)threadIdx.x) < 64;
    uint64_t *Wbuf = reinterpret_cast<uint64_t *>(smem);                       // [LeanW<R>::kSlots]
    DuoBox *box = reinterpret_cast<DuoBox *>(smem + LeanW<R>::kBytes);
    TagSet<BB, DB> vis;
    vis.tab = reinterpret_cast<uint32_t *>(smem + LeanW<R>::kBytes + kDuoBoxBytes);
    vis.idbits = idbits;
    vis.lcap = lcap;
    vis.count = 0;
    vis.lossy = false;
    if (threadIdx.x == 0) {
        box->mseq = 0;
        box->sseq = 0;
    }
    __syncthreads();
    DuoSeq seq = {0};
    WorkCtr ctr = {};
    const int32_t ep0 = g.hdr->enterpoint;        // core.rs:866
    const uint32_t lmax = g.hdr->max_layer;       // core.rs:867
    for (uint32_t qi = blockIdx.x; qi < B; qi += gridDim.x) {
        uint32_t nW = 0;
        if (walker) {
            typename VEC::Q qr;
            VEC::load_q(Q + (size_t)qi * g.dim, qr, lane);
            uint32_t ep = (uint32_t)ep0;
            for (uint32_t lc = lmax; lc >= 1; --lc) {  // core.rs:870-874: a handful of expansions each, the walker alone
                search_level_lean<VEC, 1, BB, DB, WIDE>(g, Wbuf, vis, qr, ep, 1, lc, ctr, lane, &g.hdr->ctr_search[3]);
                ep = key_id(Wbuf[0]);                  // core.rs:872
                lds_order();
            }
            nW = duo_walk<VEC, BB, DB, WIDE>(g, Wbuf, box, seq, vis, qr, ep, ef, 0, ctr, lane);   // core.rs:876
            if (nW == kEmpty)
                nW = search_level_lean<VEC, R, BB, DB, WIDE>(g, Wbuf, vis, qr, ep, ef, 0, ctr, lane, &g.hdr->ctr_search[3]);
        } else {
            (void)duo_keep<R, WIDE>(g, Wbuf, box, ef, seq, lane, ctr);
        }
        __syncthreads();
        if (walker) {
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
    if (walker && lane == 0) {
        atomicAdd(&g.hdr->ctr_search[0], (unsigned long long)ctr.n_dist);
        atomicAdd(&g.hdr->ctr_search[1], (unsigned long long)ctr.n_ids);
        atomicAdd(&g.hdr->ctr_search[2], (unsigned long long)ctr.n_expand);
    }
#ifdef HNSW_PHASE_TIMERS
    if (lane == 0)
        for (int i = walker ? 0 : 6; i < (walker ? 6 : 8); ++i) atomicAdd(&g.hdr->prof[i], ctr.ph[i]);
#endif
}

} // namespace hnsw
