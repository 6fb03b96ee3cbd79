// hnsw_plan_lean.hpp -- the PLAN half of HNSW.NODE.ADD (core.rs:511-531) with the specialised dim-128 search
// routine (hnsw_search_lean.hpp) doing the descent and the per-layer search_level(ef_construction).
//
// A plan is one wavefront's dependent chain (~230 expansions); the general routine the insert kernels were
// written with (search_level_v2) costs a lone wave 3.5-5.6 us per expansion, the specialised one 1.5-2.2 us
// (DESIGN.md 4.1c).  Both produce the same W, the same work counters and -- for the exact-order parallel insert --
// the same read log; everything after the search (select_neighbors, the node's own rows, the list of shrinks)
// is the code of k_insert_plan / k_occ_plan.  Used when the index has the shape the specialised routine serves
// (f32 rows of dim 128, adjacency rows of at most 127 ids, ef_construction <= 512, ids < 2^24): single
// hnsw_add calls (the only form the Redis command can issue, src/lib.rs:356) and the windowed exact build.
//
// LDS: the insert kernels' carve-up (W, S, fresh, dsc, aux, the general visited table -- select_neighbors and the
// shrinks still use it), then the specialised routine's Wbuf and its 32 KB tag table.
//
// DUO = true: the two-wave form (hnsw_search_duo.hpp).  The workgroup has a second wavefront that does nothing but
// keep W for the layer searches of this plan (core.rs:524) -- a plan is always a lone chain, so the split that
// serves lone queries serves every plan.  Everything else is the first wave's, exactly as in the one-wave kernel; the
// translation unit that instantiates the DUO kernels (hnsw_tu_planduo.hip) turns the block-level synchronisations
// of the shared insert code into wave-level ones, so that the keeper takes no part in them and the only s_barriers
// the walker executes are the two of each hand-over.
#pragma once
#include "hnsw_occ.hpp"
#include "hnsw_search_lean.hpp"
#include "hnsw_search_duo.hpp"

namespace hnsw {

constexpr int kPlanLeanBB = 11;                       // one plan per workgroup: the largest tag table (12 288 ids before it stops recording)
template <int R>
constexpr size_t plan_lean_bytes() { return LeanW<R>::kBytes + ((size_t)16 << kPlanLeanBB) + kDuoBoxBytes; }

template <int R, int DB, bool WIDE, bool LOG, bool DUO = false>
struct PlanLean {
    using VEC = VecF32<4>;
    uint64_t *Wbuf;
    DuoBox *box;
    DuoSeq seq;
    TagSet<kPlanLeanBB, DB> ts;
    typename VEC::Q q;

    static __device__ __forceinline__ uint64_t *wbuf_of(unsigned char *lds) { return reinterpret_cast<uint64_t *>(lds); }
    static __device__ __forceinline__ DuoBox *box_of(unsigned char *lds)
    {
        return reinterpret_cast<DuoBox *>(lds + LeanW<R>::kBytes + ((size_t)16 << kPlanLeanBB));
    }
    __device__ __forceinline__ void init(unsigned char *lds, uint32_t idbits, const QReg<4> &qr)
    {
        Wbuf = wbuf_of(lds);
        box = box_of(lds);
        seq.n = 0;
        ts.tab = reinterpret_cast<uint32_t *>(lds + LeanW<R>::kBytes);
        ts.idbits = idbits;
        ts.lcap = (1u << kPlanLeanBB) * 6u;
        ts.count = 0;
        ts.lossy = false;
#pragma unroll
        for (int t = 0; t < 4; ++t) q.q[t] = qr.q[t];   // same pieces, same lanes (load_query / VecF32::load_q)
    }
    // descent step (core.rs:511-520): the nearest node of layer lc
    __device__ __forceinline__ uint32_t nearest(const GraphView &g, uint32_t ep, uint32_t lc, WorkCtr &ctr, int lane)
    {
        search_level_lean<VEC, 1, kPlanLeanBB, DB, WIDE, LOG>(g, Wbuf, ts, q, ep, 1, lc, ctr, lane, &g.hdr->ctr_search[3]);
        const uint32_t n = key_id(Wbuf[0]);
        lds_order();
        return n;
    }
    // search_level(ef) (core.rs:524): W sorted in Wbuf[0..nW), expanded bits set
    __device__ __forceinline__ uint32_t search(const GraphView &g, uint32_t ep, uint32_t ef, uint32_t lc, WorkCtr &ctr, int lane)
    {
        if constexpr (DUO) {
            const uint32_t nW = duo_walk<VEC, kPlanLeanBB, DB, WIDE, LOG>(g, Wbuf, box, seq, ts, q, ep, ef, lc, ctr, lane);
            if (nW != kEmpty) return nW;                    // else: the table stopped recording, the walker redoes it alone
        }
        return search_level_lean<VEC, R, kPlanLeanBB, DB, WIDE, LOG>(g, Wbuf, ts, q, ep, ef, lc, ctr, lane, &g.hdr->ctr_search[3]);
    }
    // no more searches: the keeper leaves (DUO only)
    __device__ __forceinline__ void finish(int lane)
    {
        if constexpr (DUO) duo_send(box, seq, ~0ull, false, ~0ull, DUO_EXIT, lane);
    }
    // the second wavefront of a DUO plan: W's keeper for every layer search of the plan
    static __device__ __forceinline__ void keep(const GraphView &g, unsigned char *lds, uint32_t ef, int lane)
    {
        DuoSeq ks = {0};
        WorkCtr kc = {};
        while (duo_keep<R, WIDE, LOG>(g, wbuf_of(lds), box_of(lds), ef, ks, lane, kc)) {}
    }
};

// ---------------------------------------------------------------------------
// k_insert_plan with the specialised search: one wave per new node.
// ---------------------------------------------------------------------------
template <int R, int DB, bool WIDE, bool DUO = false>
__global__ __launch_bounds__(DUO ? 128 : 64, 1) void k_insert_plan_lean(GraphView g, uint32_t first_id, uint32_t count, uint32_t ef,
                                                         uint32_t mlinks, uint32_t lnb, uint32_t lcap, uint32_t *__restrict__ gspill,
                                                         uint32_t gnb, uint32_t *__restrict__ plan, uint32_t shortcut,
                                                         uint32_t lean_off, uint32_t idbits)
{
    extern __shared__ __align__(16) unsigned char smem[];
    const int lane = threadIdx.x & 63;
    if constexpr (DUO) {
        if (__builtin_amdgcn_readfirstlane((int)threadIdx.x) >= 64) {
            // one keeper session per node of the grid-stride loop below
            for (uint32_t s = blockIdx.x; s < count; s += gridDim.x) PlanLean<R, DB, WIDE, false, true>::keep(g, smem + lean_off, ef, lane);
            return;
        }
    }
    WaveMem m;
    Visited vis;
    carve<R, 4, true>(smem, g.dim, lnb, lcap, m, vis, g.tagcfg, g.selcap);
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
        QReg<4> qr;
        load_query<MODE_AVX, 4>(g.vec + (size_t)id * g.dim, g.dim, qr, m.qlds, lane);
        PlanLean<R, DB, WIDE, false, DUO> pl_;
        pl_.init(smem + lean_off, idbits, qr);
        m.W = pl_.Wbuf;                                     // the search leaves W there; select_* read it through m.W
        bool fail = false;
        uint32_t ep = ep0;
        for (uint32_t lc = lmax; lc > l; --lc) ep = pl_.nearest(g, ep, lc, ctr, lane);   // core.rs:511-520
        const uint32_t top = lmax < l ? lmax : l;
        for (uint32_t lc1 = top + 1; lc1-- > 0 && !fail;) { // core.rs:523
            const uint32_t lc = lc1;
            const uint32_t nW = pl_.search(g, ep, ef, lc, ctr, lane);                    // :524
            wave_sync();
            const uint32_t wnearest = key_id(m.W[0]);
            const uint32_t nS = shortcut && select_is_head_of_W(ef, mlinks, nW)
                                    ? select_head_of_W(m, nW, mlinks, lane)
                                    : select_topm<MODE_AVX, 4>(g, m, vis, qr, m.W, nW, id, mlinks, lc, ctr, lane, fail); // :531
            if (fail) break;
            uint32_t *pl = plan + ((size_t)s * kMaxLayers + lc) * g.plan_stride;
            if (lane == 0) pl[0] = nS;
            if ((uint32_t)lane < nS) pl[1 + lane] = key_id(m.S[lane]);
            ep = wnearest;                                  // core.rs:576
            wave_sync();
        }
        if (fail && lane == 0) atomicOr(&g.hdr->status, ST_VISITED_OVERFLOW);
        pl_.finish(lane);
    }
    if (vis.glob_dirty) visited_clear(vis, lane);
    if (lane == 0) {
        atomicAdd(&g.hdr->ctr_insert[0], (unsigned long long)ctr.n_dist);
        atomicAdd(&g.hdr->ctr_insert[1], (unsigned long long)ctr.n_ids);
        atomicAdd(&g.hdr->ctr_insert[2], (unsigned long long)ctr.n_expand);
    }
}

// ---------------------------------------------------------------------------
// k_occ_plan with the specialised search (read log included): one wave per window node without a valid plan.
// ---------------------------------------------------------------------------
template <int R, int DB, bool WIDE, bool DUO = false>
__global__ __launch_bounds__(DUO ? 128 : 64, 1) void k_occ_plan_lean(GraphView g, OccBufs ob, uint32_t first_node, uint32_t count, uint32_t ef,
                                                      uint32_t mlinks, uint32_t lnb, uint32_t lcap, uint32_t *__restrict__ gspill,
                                                      uint32_t gnb, uint32_t *__restrict__ plan, uint32_t shortcut, uint32_t log_cap,
                                                      uint32_t lean_off, uint32_t idbits, uint32_t split_pos = kEmpty, uint32_t far = 0)
{
    // split_pos: a node at window position >= split_pos with a level >= 1 gets its UPPER layers planned now and layer 0 in
    // the next round's launch (OccSlot::stage), so that no launch lasts two full searches for one node's sake.
    // far (round 6): the launch carries `far` more workgroups, for the nodes BEYOND the window (blockIdx.x in
    // [count, count + far)): everything ABOVE layer 0 -- the greedy descent, and for a node with a level >= 1 the layer
    // searches down to layer 1 with their plan rows -- is done for them now, rounds before they enter the window, so
    // that the plan a round waits for is one layer-0 search whatever the node's level.  Upper layers change only when
    // a node with a level >= 1 commits (one in M), and k_occ_validate keeps checking a staged slot's reads every round.
    extern __shared__ __align__(16) unsigned char smem[];
    const int lane = threadIdx.x & 63;
    if (!occ_round_window(ob, first_node, count)) return;
    const uint32_t id = first_node + blockIdx.x;
    const bool is_far = blockIdx.x >= count;
    if (is_far && (blockIdx.x >= count + far || id >= ob.ctl->end_node || g.hdr->max_layer == 0u)) return;
    const uint32_t slot = id % ob.W;
    OccSlot *sl = &ob.slots[slot];
    if (sl->planned && sl->node == id) return;              // (both waves of a DUO plan take the same way out)
    if (is_far && sl->node == id && sl->stage == 1u && sl->epoch == ob.ctl->epoch) return;   // staged already (k_occ_validate resets a stale one)
    if constexpr (DUO) {
        if (__builtin_amdgcn_readfirstlane((int)threadIdx.x) >= 64) {
            PlanLean<R, DB, WIDE, true, true>::keep(g, smem + lean_off, ef, lane);
            return;
        }
    }

    WaveMem m;
    Visited vis;
    carve<R, 4, true>(smem, g.dim, lnb, lcap, m, vis, g.tagcfg, g.selcap);
    vis.glob = gspill + (size_t)blockIdx.x * gnb * 8;
    vis.gnb = gnb;
    vis.glob_dirty = false;
    vis.spilled = false;
    vis.count = 0;
    vis.bounded = false;
    vis.lossy = false;

    OccRead *reads = ob.reads + (size_t)slot * kOccMaxReads;
    OccShr *shr = ob.shr + (size_t)slot * kOccMaxShr;
    WorkCtr ctr = {};
    ctr.log = reads;
    ctr.log_cap = kOccMaxReads;
    const uint32_t snap = ob.ctl->nJ, epoch = ob.ctl->epoch;
    const uint32_t lmax = g.hdr->max_layer;                 // core.rs:496
    const uint32_t ep0 = (uint32_t)g.hdr->enterpoint;       // core.rs:508
    const uint32_t l = g.levels[id];
    uint32_t *pl0 = plan + (size_t)slot * kMaxLayers * g.plan_stride;
    const uint32_t top = lmax < l ? lmax : l;
    // stage 1 done a round ago and still valid (k_occ_validate checked it against the journal): layer 0 only
    const bool resume = sl->stage == 1u && sl->node == id && sl->epoch == epoch && lmax >= 1;
    // far from the head, two or more layers to search: the upper ones now, layer 0 next round; beyond the window:
    // whatever lies above layer 0 (the descent alone for a node of level 0)
    const bool split = !resume && (is_far ? lmax >= 1 : (top >= 1 && blockIdx.x >= split_pos));
    uint32_t snapU = snap;

    QReg<4> qr;
    load_query<MODE_AVX, 4>(g.vec + (size_t)id * g.dim, g.dim, qr, m.qlds, lane);
    PlanLean<R, DB, WIDE, true, DUO> pl_;
    pl_.init(smem + lean_off, idbits, qr);
    m.W = pl_.Wbuf;
    bool fail = false, staged = false;
    uint32_t ep = ep0;
    if (resume) {
        ep = sl->ep_l0;
        ctr.log_n = sl->n_reads_u;
        snapU = sl->snapU;
    } else {
        for (uint32_t lc = lmax; lc > l; --lc) ep = pl_.nearest(g, ep, lc, ctr, lane);   // core.rs:511-520
    }
    for (uint32_t lc1 = resume ? 1u : top + 1; lc1-- > 0 && !fail;) {     // core.rs:523
        const uint32_t lc = lc1;
        if (split && lc == 0 && ctr.log_n <= log_cap) { staged = true; break; }
        const uint32_t nW = pl_.search(g, ep, ef, lc, ctr, lane);                        // :524
        wave_sync();
        const uint32_t wnearest = key_id(m.W[0]);
        uint32_t nS;
        if (shortcut && select_is_head_of_W(ef, mlinks, nW)) {
            nS = select_head_of_W(m, nW, mlinks, lane, &ctr.n_tie);
        } else {
            // select's reads: the rows of all members of W (logged before S exists; the bound is patched in below)
            const uint32_t sel_log0 = ctr.log_n;
            for (uint32_t i = lane; i < nW; i += 64)
                if (sel_log0 + i < kOccMaxReads) reads[sel_log0 + i] = OccRead{key_id(m.W[i]), occ_meta(lc, OCC_SELECT, 0, false), 0u};
            ctr.log_n += nW;
            nS = select_topm<MODE_AVX, 4>(g, m, vis, qr, m.W, nW, id, mlinks, lc, ctr, lane, fail); // :531
            if (fail) break;
            const bool sfull = nS >= mlinks;
            const uint32_t sbound = nS ? (uint32_t)(m.S[nS - 1] >> 32) : 0u;
            for (uint32_t i = lane; i < nW; i += 64)
                if (sel_log0 + i < kOccMaxReads) { reads[sel_log0 + i].meta = occ_meta(lc, OCC_SELECT, 0, sfull); reads[sel_log0 + i].bound = sbound; }
        }
        uint32_t *pl = pl0 + (size_t)lc * g.plan_stride;
        if (lane == 0) pl[0] = nS;
        if ((uint32_t)lane < nS) pl[1 + lane] = key_id(m.S[lane]);
        // the node's own row: what connect_neighbors will make it (core.rs:770); nobody can reach it yet
        uint32_t *qrow = row_ptr(g, id, lc);
        if (lane == 0) qrow[0] = nS;
        if ((uint32_t)lane < nS) qrow[1 + lane] = key_id(m.S[lane]);
        ep = wnearest;                                      // core.rs:576
        wave_sync();
    }
    if (staged && !fail) {
        // stage 1 ends here: the upper layers' plan rows, own rows and log entries are written; layer 0 starts from ep
        pl_.finish(lane);
        if (vis.glob_dirty) visited_clear(vis, lane);
        __threadfence();
        wave_sync();
        if (lane == 0) {
            sl->node = id; sl->planned = 0u; sl->epoch = epoch; sl->fail = 0u;
            sl->snapU = snap; sl->ep_l0 = ep; sl->n_reads_u = ctr.log_n; sl->top = top;
            sl->tie = ctr.n_tie ? 1u : 0u;
            sl->stage = 1u;
            atomicAdd(&g.hdr->ctr_insert[0], (unsigned long long)ctr.n_dist);
            atomicAdd(&g.hdr->ctr_insert[1], (unsigned long long)ctr.n_ids);
            atomicAdd(&g.hdr->ctr_insert[2], (unsigned long long)ctr.n_expand);
            if (ctr.n_tie) { atomicAdd(&g.hdr->ctr_tie[2], (unsigned long long)ctr.n_tie); atomicAdd(&g.hdr->ctr_tie[3], 1ull); }
        }
        return;
    }
    pl_.finish(lane);
    __threadfence();
    wave_sync();
    occ_plan_finish(g, ob, sl, shr, pl0, ctr, id, top, mlinks, log_cap, snap, epoch, fail, vis, lane, snapU);
}

} // namespace hnsw
