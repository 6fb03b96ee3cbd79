// hnsw_insert.hpp -- HNSW.NODE.ADD on the GPU (core.rs:383-412, 489-599).
//
// An insert is two phases:
//   plan    read-only: greedy descent, per-layer search_level(ef_construction)
//           and select_neighbors(m) -> the link list of every layer
//           (core.rs:511-531).  One wave per new node; many nodes per launch in
//           the fast build.
//   commit  connect_neighbors + the shrink loop + enterpoint promotion
//           (core.rs:532-596).  The exact commit is one wave replaying the
//           reference's serial order link for link; the fast build commits a
//           whole batch with atomics and prunes over-full rows afterwards.
// Layers are independent structures, so planning every layer before committing
// any is equivalent to the reference's interleaving.
#pragma once
#include "hnsw_wave_sync.hpp"
#include "hnsw_device.hpp"

namespace hnsw {

constexpr uint32_t kMaxLayers = 32;       // levels 0..31
// plan rows: per (slot, layer) a count + up to M ids; g.plan_stride words each (65, or 1 + kMaxM for an index with M > 64)

// ---------------------------------------------------------------------------
// select_neighbors (core.rs:677-757) with extend_candidates = keep_pruned =
// true, which is every call site (:528-529, :565-566).  The r / wd dance of
// :724-754 returns exactly the m nearest of  C u N_lc(C) \ {query}  (the first
// pop seeds r, every later pop fails the strict `>` at :733 and waits in wd,
// keep_pruned then refills r nearest first), so that set is built directly:
// S starts as the m nearest of C, then every unvisited neighbour of every
// candidate is merged in.  cand[0..ncand) is sorted nearest first and is left
// untouched.  Result in m.S[0..n), sorted; returns n.
// ---------------------------------------------------------------------------
// S holds up to kSelMax keys (m_max0 = 2M): one register slice per 64; the four-slice form serves M > 64 only
// ties (tie census, DevHeader::ctr_tie): ties[1] = the nearest distance outside the selection so far (rejected arrivals
// here, evicted keys in merge_sorted); select_topm compares it with the last selected key at the end: equal = the cut of
// core.rs:733 / :741-754 fell between equal similarities, which of the two is selected is the heap's choice there
__device__ __forceinline__ uint32_t merge_S(uint64_t *S, uint32_t nS, uint32_t mcap, uint64_t key, bool have, uint64_t worst, int lane,
                                            uint32_t *ties)
{
    const bool take = have && key < worst;
    if (ties && __ballot(have && !take)) {                  // the nearest rejected arrival joins "everything outside the selection"
        uint32_t dmin = have && !take ? (uint32_t)(key >> 32) : 0xFFFFFFFFu;
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) dmin = min(dmin, (uint32_t)__shfl_xor((int)dmin, o, 64));
        ties[1] = min(ties[1], dmin);
    }
    if (mcap > 128) return merge_sorted<4>(S, nS, mcap, key, take, lane, ties);
    return mcap > 64 ? merge_sorted<2>(S, nS, mcap, key, take, lane, ties) : merge_sorted<1>(S, nS, mcap, key, take, lane, ties);
}
__device__ __forceinline__ uint32_t merge_S(uint64_t *S, uint32_t nS, uint32_t mcap, uint64_t key, bool take, int lane)
{
    return merge_S(S, nS, mcap, key, take, ~0ull, lane, nullptr);
}

// GV: the graph (GraphView) or a private overlay of it (OverlayView, hnsw_occ_par.hpp): rows are reached through
// row_ptr(g, ...) / row_mut(g, ...), everything else is the graph's
template <int MODE, int T, class GV>
__device__ __forceinline__ uint32_t select_topm(const GV &g, const WaveMem &m, Visited &vis, const QReg<T> &qr,
                                const uint64_t *cand, uint32_t ncand, uint32_t qid, uint32_t mcap,
                                uint32_t lc, WorkCtr &ctr, int lane, bool &fail, uint32_t ignored = kEmpty, bool final_cut = true)
{
    // final_cut (tie census): this call selects from the WHOLE pool, so its last key against the nearest key outside is the
    // reference's cut; a team's share (team_select) leaves ctr.tie_emin to the wave that merges the shares
    ctr.tie_emin = 0xFFFFFFFFu;                             // tie census: nothing outside the selection yet
    visited_clear(vis, lane);                               // core.rs:692
    for (uint32_t base = 0; base < ncand; base += 64) {     // core.rs:693-696
        if (!visited_reserve(vis, lane, nullptr)) { fail = true; return 0; }
        const uint32_t i = base + lane;
        visited_insert_wave(vis, i < ncand, i < ncand ? key_id(cand[i]) : 0u, lane, nullptr);
        vis.count += ncand - base < 64 ? ncand - base : 64;
    }
    // core.rs:685 w = c.clone(): S starts as the nearest mcap candidates, minus `ignored`
    // (core.rs:728-731 skips it when popping; HNSW.NODE.DEL passes the node being removed)
    uint32_t nS;
    {
        const uint32_t take = ncand < mcap + 1 ? ncand : mcap + 1;     // up to 65 when m_max0 = 64 (M = 32)
        nS = 0;
        for (uint32_t base = 0; base < take; base += 64) {
            const uint32_t i = base + (uint32_t)lane;
            const bool in = i < take;
            const uint64_t ck = in ? cand[i] : ~0ull;
            const uint64_t ign = __ballot(in && key_id(ck) == ignored);
            const uint32_t dst = nS + (uint32_t)lane - (uint32_t)__popcll(ign & lanemask_lt(lane));
            if (in && !((ign >> lane) & 1ull) && dst < mcap) m.S[dst] = ck & ~1ull;
            nS += (take - base < 64 ? take - base : 64) - (uint32_t)__popcll(ign);
        }
        if (nS > mcap) nS = mcap;
    }
    wave_sync();
    // tie census: the candidates beyond the first mcap are outside the selection (cand is sorted: the nearest of them)
    if (ncand > mcap && nS == mcap) {
        const uint32_t first_out = key_id(cand[mcap]) != ignored ? mcap : mcap + 1u;
        if (first_out < ncand) ctr.tie_emin = (uint32_t)(cand[first_out] >> 32);
    }

    const uint32_t stride = lc ? g.strideU : g.stride0;
    if constexpr (MODE == MODE_AVX) {
        // The pool is a set (core.rs:699-721 evaluates every unvisited neighbour of every candidate, whatever the
        // order) and the result its mcap nearest: the rows are fetched eight candidates at a time, their unvisited
        // ids queue up in LDS (fresh[64] + dsc[64], adjacent and unused on this path), and distances are evaluated
        // 64 ids at a time -- full rounds of eight vectors instead of one short pass per candidate.
        uint32_t *queue = m.fresh;
        uint32_t qn = 0;
        const int grp = lane >> 3, pp = piece_of_lane(lane), sub = lane & 7;
        const float4 *vec4 = reinterpret_cast<const float4 *>(g.vec);
        const uint32_t row4 = g.dim >> 2;
        auto evaluate = [&](uint32_t count) {               // the first `count` (<= 64) ids of the queue
            const uint32_t safe_id = queue[0];
            if constexpr (T > 0 && T <= 4) {
                // all eight rounds of eight vectors in flight at once (128 VGPRs at dim 128): one memory latency
                // and one merge per 64 ids; lane (grp, sub) ends up with the key of slot sub * 8 + grp
                uint32_t idr[8];
                bool live[8];
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    const uint32_t slot = (uint32_t)(r * 8 + grp);
                    live[r] = slot < count;
                    idr[r] = live[r] ? queue[slot] : safe_id;
                }
                float dd[8];
                dist_rounds<T, 8>(vec4, row4, idr, qr, m.qlds, pp, dd, [] {});       // core.rs:711
                uint64_t key = ~0ull;
                bool have = false;
#pragma unroll
                for (int r = 0; r < 8; ++r)
                    if (sub == r && live[r]) { key = pack_key(dd[r], idr[r]); have = true; }
                const uint64_t worst = nS == mcap ? m.S[mcap - 1] : ~0ull;
                nS = merge_S(m.S, nS, mcap, key, have, worst, lane, &ctr.n_tie);   // core.rs:717
                return;
            }
            for (uint32_t pass = 0; pass * 32 < count; ++pass) {
                constexpr int RB = (T <= 4) ? 4 : (T <= 8 ? 2 : 1);
                uint64_t key = ~0ull;
                bool have = false;
#pragma unroll
                for (int r0 = 0; r0 < 4; r0 += RB) {
                    uint32_t idr[RB];
                    bool live[RB];
#pragma unroll
                    for (int rr = 0; rr < RB; ++rr) {
                        const uint32_t slot = pass * 32 + (uint32_t)((r0 + rr) * 8 + grp);
                        live[rr] = slot < count;
                        idr[rr] = live[rr] ? queue[slot] : safe_id;
                    }
                    float dd[RB];
                    dist_rounds<T, RB>(vec4, row4, idr, qr, m.qlds, pp, dd, [] {});   // core.rs:711
#pragma unroll
                    for (int rr = 0; rr < RB; ++rr)
                        if (sub == r0 + rr && live[rr]) { key = pack_key(dd[rr], idr[rr]); have = true; }
                }
                const uint64_t worst = nS == mcap ? m.S[mcap - 1] : ~0ull;
                nS = merge_S(m.S, nS, mcap, key, have, worst, lane, &ctr.n_tie);   // core.rs:717
            }
        };
        auto drain = [&](uint32_t keep_below) {             // evaluate blocks of 64 until fewer than keep_below wait
            while (qn >= keep_below && qn) {
                const uint32_t n = qn < 64u ? qn : 64u;
                wave_sync();
                evaluate(n);
                wave_sync();
                const uint32_t rest = qn - n;                // < 64
                const uint32_t mv = (uint32_t)lane < rest ? queue[n + lane] : 0u;
                wave_sync();
                if ((uint32_t)lane < rest) queue[lane] = mv;
                qn = rest;
            }
        };
        constexpr uint32_t G = 8;
        for (uint32_t c0 = 0; c0 < ncand; c0 += G) {          // core.rs:699-700 (the order does not matter, see above)
            uint32_t wordp[G];
#pragma unroll
            for (uint32_t j = 0; j < G; ++j) {
                wordp[j] = 0u;
                if (c0 + j < ncand) {
                    const uint32_t *rj = row_ptr(g, key_id(cand[c0 + j]), lc);
                    wordp[j] = (uint32_t)lane < stride ? rj[lane] : 0u;
                }
            }
#pragma unroll
            for (uint32_t j = 0; j < G; ++j) {
                if (c0 + j >= ncand) break;
                const uint32_t *row = row_ptr(g, key_id(cand[c0 + j]), lc);
                uint32_t word = wordp[j];
                uint32_t cnt = __builtin_amdgcn_readfirstlane(word);
                if (cnt > stride - 1) cnt = stride - 1;
                ctr.n_ids += cnt;
                for (uint32_t wbase = 0; wbase <= cnt; wbase += 64) { // core.rs:702
                    const uint32_t wi = wbase + lane;
                    if (wbase) word = wi < stride ? row[wi] : 0u;
                    const bool valid = wi >= 1 && wi <= cnt && word != qid && word != ignored; // core.rs:704-708
                    if (!visited_reserve(vis, lane, nullptr)) { fail = true; return 0; }
                    const bool fresh = visited_insert_wave(vis, valid, word, lane, nullptr);  // core.rs:710,718
                    const uint64_t fm = __ballot(fresh);
                    const uint32_t nf = __popcll(fm);
                    if (nf == 0) continue;
                    vis.count += nf;
                    ctr.n_dist += nf;
                    if (fresh) queue[qn + (uint32_t)__popcll(fm & lanemask_lt(lane))] = word;   // qn < 64 here
                    qn += nf;
                    drain(64);
                }
            }
        }
        drain(1);
        wave_sync();
        if (final_cut && nS == mcap && nS && ctr.tie_emin == (uint32_t)(m.S[mcap - 1] >> 32)) ctr.n_tie += 1u;   // tie census
        if (final_cut && any_adjacent_equal(m.S, nS, lane)) ctr.n_tie += 1u;     // ... and equal distances INSIDE the selection (link / shrink order)
        return nS;
    }
    // software prefetch: the next candidate's row is requested before the
    // current one's distances are computed
    uint32_t word_next = 0;
    if (ncand) {
        const uint32_t *r0 = row_ptr(g, key_id(cand[0]), lc);
        word_next = (uint32_t)lane < stride ? r0[lane] : 0u;
    }
    for (uint32_t ci = 0; ci < ncand; ++ci) {               // core.rs:699-700 nearest first
        const uint32_t e = key_id(cand[ci]);
        const uint32_t *row = row_ptr(g, e, lc);
        uint32_t word = word_next;
        if (ci + 1 < ncand) {
            const uint32_t *rn = row_ptr(g, key_id(cand[ci + 1]), lc);
            word_next = (uint32_t)lane < stride ? rn[lane] : 0u;
        }
        uint32_t cnt = __builtin_amdgcn_readfirstlane(word);
        if (cnt > stride - 1) cnt = stride - 1;
        ctr.n_ids += cnt;
        for (uint32_t wbase = 0; wbase <= cnt; wbase += 64) { // core.rs:702
            const uint32_t wi = wbase + lane;
            if (wbase) word = wi < stride ? row[wi] : 0u;
            const bool valid = wi >= 1 && wi <= cnt && word != qid && word != ignored; // core.rs:704-708
            if (!visited_reserve(vis, lane, nullptr)) { fail = true; return 0; }
            const bool fresh = visited_insert_wave(vis, valid, word, lane, nullptr);  // core.rs:710,718
            const uint64_t fm = __ballot(fresh);
            const uint32_t nf = __popcll(fm);
            if (nf == 0) continue;
            vis.count += nf;
            ctr.n_dist += nf;
            if (fresh) m.fresh[__popcll(fm & lanemask_lt(lane))] = word;
            wave_sync();
            compute_dists<MODE, T>(g, qr, m, nf, lane);      // core.rs:711
            wave_sync();
            const bool have = (uint32_t)lane < nf;
            const uint64_t key = have ? pack_key(m.dsc[lane], m.fresh[lane]) : ~0ull;
            const uint64_t worst = nS == mcap ? m.S[mcap - 1] : ~0ull;
            nS = merge_S(m.S, nS, mcap, key, have, worst, lane, &ctr.n_tie); // core.rs:717
        }
    }
    wave_sync();
    if (final_cut && nS == mcap && nS && ctr.tie_emin == (uint32_t)(m.S[mcap - 1] >> 32)) ctr.n_tie += 1u;       // tie census
    if (final_cut && any_adjacent_equal(m.S, nS, lane)) ctr.n_tie += 1u;         // ... and equal distances INSIDE the selection (link / shrink order)
    return nS;
}

// ---------------------------------------------------------------------------
// select_neighbors right after search_level, the short way.  When search_level(ef) returns, every member
// of W has been expanded: the loop (core.rs:630-668) only stops when the nearest unexpanded candidate is
// strictly farther than W's furthest, which no member of W is.  So every neighbour of every member was
// visited by that search -- evaluated, and either kept in W or rejected / evicted against a threshold
// that is no farther than W's final furthest.  The extension of select_neighbors (core.rs:698-721) walks
// exactly those neighbours (same rows, nothing changed in between): everything it can add is farther than
// all of W, and the r / wd passes (:724-754) return the m nearest of the pool.  Hence, whenever ef >= m (or W
// never filled, in which case nothing was ever rejected and the pool IS W):
//        select_neighbors(query, W, m) = the m nearest of W,
// no distance evaluation needed (the reference spends about half of an insert's evaluations there:
// SURVEY 8a-6).  Checked on the CPU against the full procedure (0 differences in 51 k calls, ties and
// ef == m included) and by the graph-equality tests, which run with it.  `select_shortcut = 0` keeps the full
// extension (its work counters then equal the reference's).
// ---------------------------------------------------------------------------
__device__ __forceinline__ bool select_is_head_of_W(uint32_t ef, uint32_t mcap, uint32_t nW) { return ef >= mcap || nW < ef; }
__device__ __forceinline__ uint32_t select_head_of_W(const WaveMem &m, uint32_t nW, uint32_t mcap, int lane, uint32_t *ties = nullptr)
{
    const uint32_t nS = nW < mcap ? nW : mcap;
    for (uint32_t i = lane; i < nS; i += 64) m.S[i] = m.W[i] & ~1ull;
    // tie census: the cut falls between equal distances (core.rs:733: which of the two is linked is the heap's choice), or
    // two selected neighbours have equal distances (core.rs:765-772, :540-541: which is linked / shrunk first)
    if (ties && nW > mcap && mcap && (uint32_t)(m.W[mcap - 1] >> 32) == (uint32_t)(m.W[mcap] >> 32)) *ties += 1u;
    if (ties && any_adjacent_equal(m.W, nS, lane)) *ties += 1u;
    wave_sync();
    return nS;
}

// ---------------------------------------------------------------------------
// plan kernel: one wave per new node (ids first_id .. first_id+count-1, whose
// vectors / levels / upper slots are already in HBM and whose rows are empty).
// plan[(slot*kMaxLayers + lc)*g.plan_stride] = n, then n ids nearest first.
// ---------------------------------------------------------------------------
template <int MODE, int T, int R>
__global__ __launch_bounds__(64, 1) void k_insert_plan(GraphView g, uint32_t first_id, uint32_t count, uint32_t ef,
                                                    uint32_t mlinks, uint32_t lnb, uint32_t lcap, uint32_t *__restrict__ gspill,
                                                    uint32_t gnb, uint32_t *__restrict__ plan, uint32_t shortcut)
{
    extern __shared__ __align__(16) unsigned char smem[];
    const int lane = threadIdx.x;
    WaveMem m;
    Visited vis;
    carve<R, T, true>(smem, g.dim, lnb, lcap, m, vis, g.tagcfg, g.selcap);
    vis.glob = gspill + (size_t)blockIdx.x * gnb * 8;
    vis.gnb = gnb;
    vis.glob_dirty = false;
    vis.spilled = false;
    vis.count = 0;
    vis.bounded = false;
    vis.lossy = false;

    WorkCtr ctr = {};
    const uint32_t lmax = g.hdr->max_layer;                 // core.rs:496
    const uint32_t ep0 = (uint32_t)g.hdr->enterpoint;       // core.rs:508

    for (uint32_t s = blockIdx.x; s < count; s += gridDim.x) {
        const uint32_t id = first_id + s;
        const uint32_t l = g.levels[id];
        QReg<T> qr;
        load_query<MODE, T>(g.vec + (size_t)id * g.dim, g.dim, qr, m.qlds, lane);
        bool fail = false;
        uint32_t ep = ep0;
        for (uint32_t lc = lmax; lc > l && !fail; --lc) {   // core.rs:511-520
            search_level<MODE, T, 1>(g, m, vis, qr, ep, 1, lc, ctr, lane, fail);
            ep = key_id(m.W[0]);                            // core.rs:514
            wave_sync();
        }
        const uint32_t top = lmax < l ? lmax : l;
        for (uint32_t lc1 = top + 1; lc1-- > 0 && !fail;) { // core.rs:523
            const uint32_t lc = lc1;
            const uint32_t nW = search_level<MODE, T, R>(g, m, vis, qr, ep, ef, lc, ctr, lane, fail); // :524
            if (fail) break;
            const uint32_t wnearest = key_id(m.W[0]);
            const uint32_t nS = shortcut && select_is_head_of_W(ef, mlinks, nW)
                                    ? select_head_of_W(m, nW, mlinks, lane)
                                    : select_topm<MODE, T>(g, m, vis, qr, m.W, nW, id, mlinks, lc, ctr, lane, fail); // :531
            if (fail) break;
            uint32_t *pl = plan + ((size_t)s * kMaxLayers + lc) * g.plan_stride;
            if (lane == 0) pl[0] = nS;
            for (uint32_t i = lane; i < nS; i += 64) pl[1 + i] = key_id(m.S[i]);   // (more than 64 only when M > 64)
            ep = wnearest;                                  // core.rs:576
            wave_sync();
        }
        if (fail && lane == 0) atomicOr(&g.hdr->status, ST_VISITED_OVERFLOW);
    }
    if (vis.glob_dirty) visited_clear(vis, lane);
    if (lane == 0) {
        atomicAdd(&g.hdr->ctr_insert[0], (unsigned long long)ctr.n_dist);
        atomicAdd(&g.hdr->ctr_insert[1], (unsigned long long)ctr.n_ids);
        atomicAdd(&g.hdr->ctr_insert[2], (unsigned long long)ctr.n_expand);
    }
}

// ---------------------------------------------------------------------------
// update_node_connections (core.rs:776-822) for node e whose old row (cnt ids, stored order) is in
// m.aux and whose new neighbour set is m.S[0..nS) (nearest first).  Final row(e) = old entries that
// survive, in their stored order (add_neighbor leaves them where they are, :793; rm_neighbor keeps
// order, :808), followed by the brand-new ones nearest first (:790-796).  Brand-new neighbours get
// e appended; dropped ones lose e -- except `ignored` (the node HNSW.NODE.DEL is removing), whose
// rows are left alone (:810-813).
// ---------------------------------------------------------------------------
__device__ __forceinline__ void touch_push(uint32_t *touched, uint32_t cap, uint32_t &nt, uint32_t id, bool on,
                                           int lane);

// Journal of row changes (hnsw_occ.hpp): every entry says "id z was added to / removed from row (row, layer)".
// A ring; `n` counts entries ever written.  One wave writes it (the in-order commit).
struct OccDelta { uint32_t row, lc_add, z; };        // lc_add = layer | added << 8
constexpr uint32_t kOccJournalBits = 18;
struct OccJournal {
    OccDelta *ring;
    uint32_t n;                                      // the committing wave's copy of the counter (wave-uniform)
    OccDelta *own;                                   // optional LDS mirror of the entries from own_base on
    uint32_t own_base;
    uint32_t cap;                                    // 0: `ring` is the journal ring; else a private linear buffer of `cap` entries
                                                     // (a dry run's deltas, hnsw_occ_par.hpp): entries beyond it are counted, not written
};
// lanes flagged `on` each append one delta (wave-uniform call)
__device__ __forceinline__ void journal_push(OccJournal *jr, bool on, uint32_t row, uint32_t lc, uint32_t z, bool add, int lane)
{
    if (!jr) return;
    const uint64_t b = __ballot(on);
    if (!b) return;
    if (on) {
        const uint32_t p = (jr->n + (uint32_t)__popcll(b & lanemask_lt(lane))) & ((1u << kOccJournalBits) - 1u);
        const OccDelta d = OccDelta{row, lc | (add ? 256u : 0u), z};
        if (!jr->cap || p < jr->cap) jr->ring[p] = d;
        const uint32_t o = jr->n + (uint32_t)__popcll(b & lanemask_lt(lane)) - jr->own_base;
        if (jr->own && o < 256u) jr->own[o] = d;
    }
    jr->n += (uint32_t)__popcll(b);
}

template <class GV>
__device__ __forceinline__ void update_connections(const GV &g, const WaveMem &m, uint32_t e, uint32_t *erow, uint32_t cnt,
                                   uint32_t nS, uint32_t lc, uint32_t stride, uint32_t *maxdeg, uint32_t ignored,
                                   uint32_t *touched, uint32_t touched_cap, uint32_t &nt, int lane,
                                   OccJournal *jr = nullptr)
{
    // The rows of the lanes flagged in `mask` (x = each lane's row id) lose e (remove) or gain it (append).  Each row is
    // edited by the whole wave -- one row load, one ballot, one store of the row's live words -- and the loads of up to
    // four rows are in flight together (a lone wave pays a full memory round trip per dependent load; a shrink touches
    // two to four rows).  The live words are rewritten whole: row_rewrite hands an overlay's scratch row out empty and
    // this very store fills it.
    auto edit_rows = [&](uint64_t mask, uint32_t x, bool remove) {
        while (mask) {
            const uint32_t *xs[4];
            uint32_t *xr[4];
            uint32_t wxs[4];
            int n = 0;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                xs[u] = nullptr; xr[u] = nullptr;
                if (mask) {
                    const int j = __ffsll((unsigned long long)mask) - 1;
                    mask &= mask - 1;
                    const uint32_t xj = (uint32_t)__builtin_amdgcn_readlane((int)x, j);
                    xr[u] = row_rewrite(g, xj, lc, lane, &xs[u]);
                    n = u + 1;
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) wxs[u] = (u < n && (uint32_t)lane < stride) ? xs[u][lane] : kEmpty;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (u >= n) break;
                const uint32_t *xsrc = xs[u];
                uint32_t *xrow = xr[u];
                const uint32_t wx = wxs[u];
                uint32_t xc1 = (uint32_t)__builtin_amdgcn_readfirstlane((int)wx);   // (lane 0 = the count)
                if (xc1 > stride - 1) xc1 = stride - 1;
                if (remove) {
                    if (xc1 <= 63) {                                    // rows of <= 63 ids: the 64 words just loaded are the row
                        const uint64_t hit = __ballot(lane >= 1 && (uint32_t)lane <= xc1 && wx == e);
                        if (!hit) {                                                            // reference panics, :150
                            if (lane == 0) atomicOr(&g.hdr->status, ST_ASYMMETRIC);
                            if ((uint32_t)lane <= xc1) xrow[lane] = wx;
                        } else {
                            const int pos = __ffsll((unsigned long long)hit) - 1;              // word index of e
                            const uint32_t nxt = (uint32_t)__shfl_down((int)wx, 1, 64);         // word lane + 1
                            if ((uint32_t)lane < xc1) xrow[lane] = lane == 0 ? xc1 - 1 : (lane >= pos ? nxt : wx);
                        }
                        continue;
                    }
                    if (xsrc != xrow) { for (uint32_t i = lane; i < stride; i += 64) xrow[i] = xsrc[i]; wave_sync(); }   // wide rows: copy, then edit in place
                    const uint32_t xc = xc1;
                    bool found = false;
                    for (uint32_t b2 = 0; b2 < xc; b2 += 64) {          // rows wider than 63 ids: more than one pass
                        const uint32_t p = b2 + lane;
                        const uint32_t v = p < xc ? xrow[1 + p] : kEmpty;
                        const uint32_t vnext = p + 1 < xc ? xrow[2 + p] : kEmpty;
                        const uint64_t hit = __ballot(p < xc && v == e);
                        if (!found && hit) {
                            found = true;
                            const uint32_t pos = b2 + (uint32_t)(__ffsll((unsigned long long)hit) - 1);
                            wave_sync();
                            if (p >= pos && p + 1 < xc) xrow[1 + p] = vnext;
                        } else if (found) {
                            wave_sync();
                            if (p + 1 < xc) xrow[1 + p] = vnext;
                        }
                    }
                    if (!found) { if (lane == 0) atomicOr(&g.hdr->status, ST_ASYMMETRIC); }   // reference panics, :150
                    else if (lane == 0) xrow[0] = xc - 1;
                } else {
                    if (xc1 < 63) {                                     // the row and the appended slot are words 0..63
                        const bool there = __ballot(lane >= 1 && (uint32_t)lane <= xc1 && wx == e) != 0;
                        if (there || xc1 + 1 > stride - 1) {
                            if (!there && lane == 0) atomicOr(&g.hdr->status, ST_ROW_OVERFLOW);
                            if ((uint32_t)lane <= xc1) xrow[lane] = wx;
                        } else {
                            if ((uint32_t)lane <= xc1 + 1) xrow[lane] = lane == 0 ? xc1 + 1 : ((uint32_t)lane == xc1 + 1 ? e : wx);
                            if (lane == 0) atomicMax(maxdeg, xc1 + 1);
                        }
                        continue;
                    }
                    if (xsrc != xrow) { for (uint32_t i = lane; i < stride; i += 64) xrow[i] = xsrc[i]; wave_sync(); }   // wide rows: copy, then edit in place
                    const uint32_t xc = xc1;
                    bool present = false;
                    for (uint32_t b2 = 0; b2 < xc; b2 += 64) {
                        const uint32_t p = b2 + lane;
                        present |= __ballot(p < xc && xrow[1 + p] == e) != 0;
                    }
                    if (!present && lane == 0) {
                        if (xc + 1 > stride - 1) atomicOr(&g.hdr->status, ST_ROW_OVERFLOW);
                        else { xrow[1 + xc] = e; xrow[0] = xc + 1; atomicMax(maxdeg, xc + 1); }
                    }
                }
            }
        }
    };
    uint32_t kept = 0;
    for (uint32_t base = 0; base < cnt; base += 64) {
        const uint32_t i = base + lane;
        const uint32_t x = i < cnt ? m.aux[i] : kEmpty;
        bool inS = false;
        if (i < cnt)
            for (uint32_t j = 0; j < nS; ++j) inS |= key_id(m.S[j]) == x;
        const uint64_t kb = __ballot(inS);
        if (inS) erow[1 + kept + __popcll(kb & lanemask_lt(lane))] = x;
        kept += __popcll(kb);
        // bidirectionally remove old-but-not-new (:805-819)
        const bool drop = i < cnt && !inS && x != ignored;
        edit_rows(__ballot(drop), x, true);
        touch_push(touched, touched_cap, nt, x, drop, lane); // :816
        journal_push(jr, drop, e, lc, x, false, lane);
        journal_push(jr, drop, x, lc, e, false, lane);
    }
    // new neighbours that were not adjacent before (:790-796), nearest first (64 of S at a time: nS <= 2M <= 128)
    for (uint32_t sb = 0; sb < nS; sb += 64) {
        const bool inSel = sb + (uint32_t)lane < nS;
        const uint32_t x = inSel ? key_id(m.S[sb + lane]) : kEmpty;
        bool isNew = inSel;
        if (isNew)
            for (uint32_t i = 0; i < cnt; ++i) isNew &= m.aux[i] != x;
        const uint64_t nb = __ballot(isNew);
        if (isNew) erow[1 + kept + __popcll(nb & lanemask_lt(lane))] = x;
        edit_rows(nb, x, false);                                // e appended to each new neighbour's row
        kept += __popcll(nb);
        journal_push(jr, isNew, e, lc, x, true, lane);
        journal_push(jr, isNew, x, lc, e, true, lane);
        touch_push(touched, touched_cap, nt, x, inSel, lane); // :796
    }
    if (lane == 0) erow[0] = kept;
    touch_push(touched, touched_cap, nt, e, lane == 0, lane);               // :787
    fence_own_writes();
    wave_sync();
}

// ---------------------------------------------------------------------------
// exact commit (one wave, one node): core.rs:532-596 in the reference's order
// ---------------------------------------------------------------------------
__device__ __forceinline__ void touch_push(uint32_t *touched, uint32_t cap, uint32_t &nt, uint32_t id, bool on,
                                           int lane)
{
    const uint64_t b = __ballot(on);
    if (on) {
        uint32_t p = nt + __popcll(b & lanemask_lt(lane));
        if (p < cap) touched[p] = id;
    }
    nt += __popcll(b);
}

template <int MODE, int T, int R>
__global__ __launch_bounds__(64, 1) void k_insert_commit_exact(GraphView g, uint32_t id, uint32_t mlinks,
                                                            uint32_t lnb, uint32_t lcap, uint32_t *__restrict__ gspill,
                                                            uint32_t gnb, const uint32_t *__restrict__ plan,
                                                            uint32_t *__restrict__ touched, uint32_t touched_cap)
{
    extern __shared__ __align__(16) unsigned char smem[];
    const int lane = threadIdx.x;
    WaveMem m;
    Visited vis;
    carve<R, T, true>(smem, g.dim, lnb, lcap, m, vis, g.tagcfg, g.selcap);
    vis.glob = gspill;
    vis.gnb = gnb;
    vis.glob_dirty = false;
    vis.spilled = false;
    vis.count = 0;
    vis.bounded = false;
    vis.lossy = false;

    // the plan failed (its visited set outgrew even the HBM table): nothing is linked, the host reports it
    if (g.hdr->status & (ST_VISITED_OVERFLOW | ST_ROW_OVERFLOW)) return;

    WorkCtr ctr = {};
    uint32_t skipped = 0;  // econn evaluations the reference makes but whose result it never uses (:549-557 when deg <= m_max)
    uint32_t nt = 0;
    bool fail = false;
    const uint32_t lmax = g.hdr->max_layer;
    const uint32_t l = g.levels[id];
    const uint32_t top = lmax < l ? lmax : l;

    for (uint32_t lc1 = top + 1; lc1-- > 0 && !fail;) {     // core.rs:523
        const uint32_t lc = lc1;
        const uint32_t stride = lc ? g.strideU : g.stride0;
        const uint32_t mmax = lc ? mlinks : 2 * mlinks;     // core.rs:560
        uint32_t *maxdeg = lc ? &g.hdr->max_degU : &g.hdr->max_deg0;
        const uint32_t *pl = plan + (size_t)lc * g.plan_stride;
        const uint32_t nsel = pl[0];

        // connect_neighbors (core.rs:759-774): nearest first (64 at a time: more than one pass only when M > 64)
        uint32_t *qrow = row_ptr(g, id, lc);
        if (lane == 0) qrow[0] = nsel;
        for (uint32_t sb = 0; sb < nsel; sb += 64) {
            const uint32_t si = sb + (uint32_t)lane;
            const uint32_t myselid = si < nsel ? pl[1 + si] : kEmpty;
            if (si < nsel) {
                qrow[1 + si] = myselid;                     // :770
                uint32_t *nrow = row_ptr(g, myselid, lc);   // :771-772 (id is new: never present)
                uint32_t c = nrow[0];
                if (c + 1 > stride - 1) atomicOr(&g.hdr->status, ST_ROW_OVERFLOW);
                else { nrow[1 + c] = id; nrow[0] = c + 1; atomicMax(maxdeg, c + 1); }
            }
            touch_push(touched, touched_cap, nt, myselid, si < nsel, lane); // :535-537
        }
        if (lane == 0) atomicMax(maxdeg, nsel);
        __threadfence();
        wave_sync();

        // shrink loop (core.rs:540-574), e nearest first
        for (uint32_t si = 0; si < nsel && !fail; ++si) {
            const uint32_t e = pl[1 + si];
            uint32_t *erow = row_ptr(g, e, lc);
            uint32_t cnt = erow[0];
            if (cnt > stride - 1) cnt = stride - 1;
            if (cnt <= mmax) { skipped += cnt; continue; }  // :561
            ctr.n_ids += cnt;

            // econn (core.rs:544-558): sims from e to each of its neighbours
            QReg<T> qe;
            load_query<MODE, T>(g.vec + (size_t)e * g.dim, g.dim, qe, m.qlds, lane);
            uint32_t nE = 0;
            for (uint32_t base = 0; base < cnt; base += 64) {
                const uint32_t i = base + lane;
                const uint32_t nf = cnt - base < 64 ? cnt - base : 64;
                if (i < cnt) { uint32_t x = erow[1 + i]; m.fresh[lane] = x; m.aux[i] = x; }
                wave_sync();
                compute_dists<MODE, T>(g, qe, m, nf, lane);  // :550
                ctr.n_dist += nf;
                wave_sync();
                const bool have = (uint32_t)lane < nf;
                const uint64_t key = have ? pack_key(m.dsc[lane], m.fresh[lane]) : ~0ull;
                nE = merge_sorted<R>(m.W, nE, R * 64, key, have, lane);
            }
            // select_neighbors(e, econn, m_max) (core.rs:568)
            const uint32_t nS = select_topm<MODE, T>(g, m, vis, qe, m.W, nE, e, mmax, lc, ctr, lane, fail);
            if (fail) break;

            update_connections(g, m, e, erow, cnt, nS, lc, stride, maxdeg, kEmpty, touched, touched_cap, nt, lane);
        }
    }

    if (fail && lane == 0) atomicOr(&g.hdr->status, ST_VISITED_OVERFLOW);
    if (lane == 0) {
        if (l > lmax) {                                     // core.rs:587-593
            g.hdr->max_layer = l;
            g.hdr->enterpoint = (int32_t)id;
        }
        g.hdr->node_count = id + 1;
        g.hdr->n_touched = nt;
        atomicAdd(&g.hdr->ctr_insert[0], (unsigned long long)ctr.n_dist);
        atomicAdd(&g.hdr->ctr_insert[1], (unsigned long long)ctr.n_ids);
        atomicAdd(&g.hdr->ctr_insert[3], (unsigned long long)skipped);
    }
    if (vis.glob_dirty) visited_clear(vis, lane);
}

// ---------------------------------------------------------------------------
// HNSW.NODE.DEL (core.rs:414-475): for every layer of the node (ascending, :434) and every
// neighbour n of it in stored order (:829), n re-selects its links from its two-hop neighbourhood
// with the node ignored (delete_node_from_neighbors, :824-863).  One wave, the reference's serial
// order.  The host re-elects the enterpoint afterwards.
// ---------------------------------------------------------------------------
template <int MODE, int T, int R>
__global__ __launch_bounds__(64, 1) void k_delete_exact(GraphView g, uint32_t id, uint32_t mlinks, uint32_t lnb,
                                                     uint32_t lcap, uint32_t *__restrict__ gspill, uint32_t gnb,
                                                     uint32_t *__restrict__ touched, uint32_t touched_cap)
{
    extern __shared__ __align__(16) unsigned char smem[];
    const int lane = threadIdx.x;
    WaveMem m;
    Visited vis;
    carve<R, T, true>(smem, g.dim, lnb, lcap, m, vis, g.tagcfg, g.selcap);
    vis.glob = gspill;
    vis.gnb = gnb;
    vis.glob_dirty = false;
    vis.spilled = false;
    vis.count = 0;
    vis.bounded = false;
    vis.lossy = false;
    WorkCtr ctr = {};
    uint32_t nt = 0;
    bool fail = false;
    const uint32_t l = g.levels[id];

    for (uint32_t lc = 0; lc <= l && !fail; ++lc) {              // core.rs:434
        const uint32_t stride = lc ? g.strideU : g.stride0;
        const uint32_t mmax = lc ? mlinks : 2 * mlinks;           // core.rs:846
        uint32_t *maxdeg = lc ? &g.hdr->max_degU : &g.hdr->max_deg0;
        uint32_t *drow = row_ptr(g, id, lc);                      // not modified while we walk it
        uint32_t dcnt = drow[0];
        if (dcnt > stride - 1) dcnt = stride - 1;
        for (uint32_t kk = 0; kk < dcnt && !fail; ++kk) {         // core.rs:829 stored order
            const uint32_t n = drow[1 + kk];
            uint32_t *erow = row_ptr(g, n, lc);
            uint32_t cnt = erow[0];
            if (cnt > stride - 1) cnt = stride - 1;
            ctr.n_ids += cnt;
            // nconn (core.rs:832-844): sims from n to each of its neighbours (the node included)
            QReg<T> qe;
            load_query<MODE, T>(g.vec + (size_t)n * g.dim, g.dim, qe, m.qlds, lane);
            uint32_t nE = 0;
            for (uint32_t base = 0; base < cnt; base += 64) {
                const uint32_t i = base + lane;
                const uint32_t nf = cnt - base < 64 ? cnt - base : 64;
                if (i < cnt) { uint32_t x = erow[1 + i]; m.fresh[lane] = x; m.aux[i] = x; }
                wave_sync();
                compute_dists<MODE, T>(g, qe, m, nf, lane);   // :840-841
                ctr.n_dist += nf;
                wave_sync();
                const bool have = (uint32_t)lane < nf;
                const uint64_t key = have ? pack_key(m.dsc[lane], m.fresh[lane]) : ~0ull;
                nE = merge_sorted<R>(m.W, nE, R * 64, key, have, lane);
            }
            // select_neighbors(n, nconn, m_max, ignored = node) (core.rs:853)
            const uint32_t nS = select_topm<MODE, T>(g, m, vis, qe, m.W, nE, n, mmax, lc, ctr, lane, fail, id);
            if (fail) break;
            // update_node_connections(n, new, old, ignored = node) (core.rs:856); :855 is covered by :787
            update_connections(g, m, n, erow, cnt, nS, lc, stride, maxdeg, id, touched, touched_cap, nt, lane);
        }
        if (lane == 0) drow[0] = 0;                               // the node is gone (core.rs:419)
        __threadfence();
        wave_sync();
    }
    if (fail && lane == 0) atomicOr(&g.hdr->status, ST_VISITED_OVERFLOW);
    if (lane == 0) {
        g.hdr->n_touched = nt;
        atomicAdd(&g.hdr->ctr_insert[0], (unsigned long long)ctr.n_dist);
        atomicAdd(&g.hdr->ctr_insert[1], (unsigned long long)ctr.n_ids);
    }
    if (vis.glob_dirty) visited_clear(vis, lane);
}

#ifdef HNSW_UTILITY_KERNELS   // plain (non-template) kernels: defined once, in hnsw_engine.hip
// ---------------------------------------------------------------------------
// fast build: commit a planned batch.  One wave per new node links it (both
// directions, atomics on the row counters); rows pushed past m_max go to a
// worklist and are pruned to their m_max nearest by k_shrink_batch.  This is
// NOT the reference's serial order (see include/hnsw_mi355x.h, mode 1).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_link_batch(GraphView g, uint32_t first_id, uint32_t count,
                                                   uint32_t mlinks, uint32_t lmax_snapshot,
                                                   const uint32_t *__restrict__ plan, uint32_t *pending0,
                                                   uint32_t *pendingU, uint32_t *worklist, uint32_t *work_n,
                                                   uint32_t work_cap)
{
    const int lane = threadIdx.x;
    for (uint32_t s = blockIdx.x; s < count; s += gridDim.x) {
        const uint32_t id = first_id + s;
        const uint32_t l = g.levels[id];
        const uint32_t top = lmax_snapshot < l ? lmax_snapshot : l;
        for (uint32_t lc = 0; lc <= top; ++lc) {
            const uint32_t stride = lc ? g.strideU : g.stride0;
            const uint32_t mmax = lc ? mlinks : 2 * mlinks;
            const uint32_t *pl = plan + ((size_t)s * kMaxLayers + lc) * g.plan_stride;
            const uint32_t nsel = pl[0];
            uint32_t *qrow = row_ptr(g, id, lc);
            if (lane == 0) qrow[0] = nsel;
            if ((uint32_t)lane < nsel) {
                const uint32_t e = pl[1 + lane];
                qrow[1 + lane] = e;
                uint32_t *erow = row_ptr(g, e, lc);
                const uint32_t c = atomicAdd(&erow[0], 1u);
                if (c + 1 > stride - 1) {
                    atomicSub(&erow[0], 1u);
                    atomicOr(&g.hdr->status, ST_ROW_DROPPED);
                } else {
                    erow[1 + c] = id;
                    if (c + 1 > mmax) {
                        uint32_t *pend = lc ? &pendingU[g.upper_base[e] + lc - 1] : &pending0[e];
                        if (atomicExch(pend, 1u) == 0u) {
                            uint32_t w = atomicAdd(work_n, 1u);
                            if (w < work_cap) { worklist[2 * w] = e; worklist[2 * w + 1] = lc; }
                        }
                    }
                }
            }
        }
    }
}

#endif // HNSW_UTILITY_KERNELS

// prune row (e, lc) to its m_max nearest (by the engine's (dist, id) order)
template <int MODE, int T>
__global__ __launch_bounds__(64) void k_shrink_batch(GraphView g, uint32_t mlinks, const uint32_t *__restrict__ worklist,
                                                     const uint32_t *__restrict__ work_n, uint32_t work_cap,
                                                     uint32_t *pending0, uint32_t *pendingU)
{
    extern __shared__ __align__(16) unsigned char smem[];
    const int lane = threadIdx.x;
    WaveMem m;
    m.W = reinterpret_cast<uint64_t *>(smem);            // [128] keys
    m.fresh = reinterpret_cast<uint32_t *>(smem + 128 * 8);
    m.dsc = reinterpret_cast<float *>(smem + 128 * 8 + 64 * 4);
    m.qlds = reinterpret_cast<float *>(smem + 128 * 8 + 64 * 8);
    m.S = nullptr;
    m.aux = nullptr;
    uint32_t nwork = *work_n;
    if (nwork > work_cap) nwork = work_cap;
    unsigned long long ndist = 0;
    for (uint32_t w = blockIdx.x; w < nwork; w += gridDim.x) {
        const uint32_t e = worklist[2 * w], lc = worklist[2 * w + 1];
        const uint32_t stride = lc ? g.strideU : g.stride0;
        const uint32_t mmax = lc ? mlinks : 2 * mlinks;
        uint32_t *erow = row_ptr(g, e, lc);
        uint32_t cnt = erow[0];
        if (cnt > stride - 1) cnt = stride - 1;
        if (lane == 0) {
            if (lc) pendingU[g.upper_base[e] + lc - 1] = 0u; else pending0[e] = 0u;
        }
        if (cnt <= mmax) continue;
        QReg<T> qe;
        wave_sync();
        load_query<MODE, T>(g.vec + (size_t)e * g.dim, g.dim, qe, m.qlds, lane);
        uint32_t nE = 0;
        for (uint32_t base = 0; base < cnt; base += 64) {
            const uint32_t i = base + lane;
            const uint32_t nf = cnt - base < 64 ? cnt - base : 64;
            if (i < cnt) m.fresh[lane] = erow[1 + i];
            wave_sync();
            compute_dists<MODE, T>(g, qe, m, nf, lane);
            ndist += nf;
            wave_sync();
            const bool have = (uint32_t)lane < nf;
            const uint64_t key = have ? pack_key(m.dsc[lane], m.fresh[lane]) : ~0ull;
            nE = merge_sorted<2>(m.W, nE, 128, key, have, lane);
        }
        const uint32_t keep = nE < mmax ? nE : mmax;
        for (uint32_t i = lane; i < keep; i += 64) erow[1 + i] = key_id(m.W[i]);
        if (lane == 0) {
            erow[0] = keep;
            atomicMax(lc ? &g.hdr->max_degU : &g.hdr->max_deg0, keep);
        }
        wave_sync();
    }
    if (lane == 0 && ndist) atomicAdd(&g.hdr->ctr_insert[0], ndist);
}

#ifdef HNSW_UTILITY_KERNELS
__global__ void k_batch_finish(DevHeader *hdr, uint32_t node_count, uint32_t max_layer, int32_t enterpoint,
                               uint32_t *work_n)
{
    hdr->node_count = node_count;
    hdr->max_layer = max_layer;
    hdr->enterpoint = enterpoint;
    *work_n = 0;
}

#endif // HNSW_UTILITY_KERNELS

} // namespace hnsw
