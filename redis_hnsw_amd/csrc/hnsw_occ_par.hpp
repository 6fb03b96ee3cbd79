// hnsw_occ_par.hpp -- the commits of a round applied in VALIDATED PARALLEL GROUPS instead of one by one.
//
// hnsw_occ.hpp plans a window of inserts in parallel and commits them strictly in id order on ONE wavefront
// (k_occ_commit): at 1 M nodes that wave was two thirds of a round -- ~12 commits of ~130 us each, one after the
// other (core.rs:532-574 per node: connect_neighbors, the shrink loop with its select_neighbors).  Here every
// window node gets a workgroup of its own and the round's commits go in GROUPS:
//
//   dry run   every window node whose link plan still holds runs its WHOLE commit -- connect, the shrink loop, each
//             speculative record validated against the node's own deltas and recomputed when stale, exactly the code
//             of k_occ_commit -- against the graph as it stands, all nodes at once, each in a private OVERLAY: a row is
//             copied into the workgroup's scratch rows the first time it is edited and every later access goes to the
//             copy (row_ptr / row_mut on an OverlayView).  The graph itself is read-only during the phase.  A dry run
//             leaves its deltas in journal order (private buffer), the rows it rewrote, read-log entries for every
//             select_neighbors it had to recompute (the rows of e's members, bound = the last selected: the same
//             entries a speculative record carries), and which speculative records it used.
//   validate  node j of the iteration may commit together with the nodes before it iff, for every earlier node i,
//               (1) no delta of i is RELEVANT (the journal's rules, occ_check_range) to anything j read: its link plan,
//                   the speculative records it used, the recomputations it made, and
//               (2) no delta of i is on a row j rewrote (j's rows are written back whole).
//             Then i's commit changes nothing j's commit saw, j's changes touch no row of i's, and applying both is
//             what the in-order wave would have produced -- rows, stored order and journal alike.
//   apply     the longest prefix of conflict-free nodes writes its overlay rows into the graph and its deltas into
//             the journal ring at the offsets the in-order wave would have used.  The first node that fails ends the
//             group and is dry-run again in the next iteration, now against a graph that holds the group -- which is
//             what the in-order wave would have given it.  A node that raises max_layer (core.rs:587-593) closes its
//             group.  The head of an iteration has no predecessor: it commits whenever its link plan holds, so the
//             scheme makes progress exactly where the in-order wave does, and the round ends where it did (the head's
//             link plan is stale: re-plan).
//
// Proven on the CPU first (tests/experiments/occ_model.c with PAR=1, under test: graphs identical to the serial
// oracle's; without rule (2) they are not) -- 6.3 nodes per group and 3 groups per round on the 1 M reference graph.
// One kernel launch per round: the iterations are separated by grid barriers (a monotonic counter, <= 64 workgroups
// of one wavefront, all resident).  A node whose dry run does not fit its buffers (overlay rows, deltas, read log) is
// left to the serial kernels, as a plan that could not be logged always was.
#pragma once
#include "hnsw_wave_sync.hpp"
#include "hnsw_occ.hpp"

namespace hnsw {

// the graph seen through a workgroup's overlay
struct OverlayView : GraphView {
    uint32_t *ovkey;                      // LDS [kParTab]: (row << 5 | layer) of the scratch row in that slot, or kEmpty
    uint32_t *ovctl;                      // LDS: [0] rows in the table, [1] set when the table overflowed
    uint32_t *ovrows;                     // HBM [kParTab + 1][ovstride]; row kParTab swallows the edits of an overflowed table
    uint32_t ovstride;
};
__device__ __forceinline__ uint32_t ov_hash(uint32_t key) { return (key * 0x9E3779B1u) >> (32 - 9); }
static_assert(kParTab == (1u << 9), "ov_hash");

// where a row is READ from (any lane, any id)
__device__ __forceinline__ uint32_t *row_ptr(const OverlayView &g, uint32_t id, uint32_t lc)
{
    const uint32_t key = (id << 5) | lc;
    uint32_t h = ov_hash(key);
    for (;;) {
        const uint32_t k = g.ovkey[h];
        if (k == key) return g.ovrows + (size_t)h * g.ovstride;
        if (k == kEmpty) break;
        h = (h + 1) & (kParTab - 1);
    }
    return row_ptr(static_cast<const GraphView &>(g), id, lc);
}
// a row about to be EDITED (wave-uniform id, every lane calls): copied into the overlay on first touch
__device__ __forceinline__ uint32_t *row_mut(const OverlayView &g, uint32_t id, uint32_t lc, int lane)
{
    const uint32_t key = (id << 5) | lc;
    uint32_t h = ov_hash(key);
    for (;;) {
        const uint32_t k = g.ovkey[h];
        if (k == key) return g.ovrows + (size_t)h * g.ovstride;
        if (k == kEmpty) break;
        h = (h + 1) & (kParTab - 1);
    }
    const uint32_t *src = row_ptr(static_cast<const GraphView &>(g), id, lc);
    const uint32_t stride = lc ? g.strideU : g.stride0;
    if (g.ovctl[0] >= kParMaxRows) {                      // no room: the dry run is void (the node goes to the serial kernels)
        uint32_t *trash = g.ovrows + (size_t)kParTab * g.ovstride;
        for (uint32_t i = lane; i < stride; i += 64) trash[i] = src[i];
        if (lane == 0) g.ovctl[1] = 1u;
        wave_sync_full();
        return trash;
    }
    uint32_t *dst = g.ovrows + (size_t)h * g.ovstride;
    for (uint32_t i = lane; i < stride; i += 64) dst[i] = src[i];
    if (lane == 0) { g.ovkey[h] = key; g.ovctl[0] += 1u; }
    wave_sync_full();                                     // the copy has landed before anybody reads it back
    return dst;
}
// a row about to be REWRITTEN (count + ids in one store, update_connections): on first touch the scratch row is handed
// out EMPTY and the graph's row is where the present content is read from -- no copy, no round trip
__device__ __forceinline__ uint32_t *row_rewrite(const OverlayView &g, uint32_t id, uint32_t lc, int lane, const uint32_t **src)
{
    const uint32_t key = (id << 5) | lc;
    uint32_t h = ov_hash(key);
    for (;;) {
        const uint32_t k = g.ovkey[h];
        if (k == key) { uint32_t *r = g.ovrows + (size_t)h * g.ovstride; *src = r; return r; }
        if (k == kEmpty) break;
        h = (h + 1) & (kParTab - 1);
    }
    *src = row_ptr(static_cast<const GraphView &>(g), id, lc);
    if (g.ovctl[0] >= kParMaxRows) {                      // no room: the dry run is void
        if (lane == 0) g.ovctl[1] = 1u;
        return g.ovrows + (size_t)kParTab * g.ovstride;
    }
    if (lane == 0) { g.ovkey[h] = key; g.ovctl[0] += 1u; }
    lds_order();
    return g.ovrows + (size_t)h * g.ovstride;
}
// connect_neighbors: lane i < n is about to append to the row of its own selected neighbour -- all of them enter the
// overlay, the copies four rows at a time.  Returns this lane's scratch row.
__device__ __forceinline__ uint32_t *ov_own_lanes(const OverlayView &g, bool valid, uint32_t id, uint32_t lc, int lane)
{
    const uint32_t stride = lc ? g.strideU : g.stride0;
    uint32_t slot = kParTab;
    bool fresh = false;
    const bool room = g.ovctl[0] + 64u <= kParMaxRows;
    if (valid && room) {
        const uint32_t key = (id << 5) | lc;
        uint32_t h = ov_hash(key);
        for (;;) {
            const uint32_t old = atomicCAS(&g.ovkey[h], kEmpty, key);
            if (old == kEmpty) { fresh = true; break; }
            if (old == key) break;
            h = (h + 1) & (kParTab - 1);
        }
        slot = h;
    }
    uint64_t fm = __ballot(fresh);
    if (lane == 0) {
        g.ovctl[0] += (uint32_t)__popcll(fm);
        if (!room) g.ovctl[1] = 1u;
    }
    while (fm) {
        uint32_t w[4];
        uint32_t *dst[4];
        const uint32_t *src[4];
        int n = 0;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            dst[u] = nullptr; src[u] = nullptr;
            if (fm) {
                const int j = __ffsll((unsigned long long)fm) - 1;
                fm &= fm - 1;
                const uint32_t sid = (uint32_t)__builtin_amdgcn_readlane((int)id, j);
                const uint32_t ss = (uint32_t)__builtin_amdgcn_readlane((int)slot, j);
                src[u] = row_ptr(static_cast<const GraphView &>(g), sid, lc);
                dst[u] = g.ovrows + (size_t)ss * g.ovstride;
                n = u + 1;
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) w[u] = (u < n && (uint32_t)lane < stride) ? src[u][lane] : 0u;
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (u < n && (uint32_t)lane < stride) dst[u][lane] = w[u];
        for (int u = 0; u < n; ++u)
            for (uint32_t i = 64 + lane; i < stride; i += 64) dst[u][i] = src[u][i];
    }
    wave_sync_full();
    return g.ovrows + (size_t)slot * g.ovstride;
}

// one more entry in the LDS hash of what the node read (the same structure occ_build_hash fills from the plan's log)
__device__ __forceinline__ void occ_hash_add(const OccScratch &sc, uint32_t idx, uint32_t row, uint32_t meta, uint32_t bound)
{
    sc.rmeta[idx] = meta;
    sc.rbound[idx] = bound;
    const uint32_t key = (row << 5) | (meta & 31u);
    uint32_t h = occ_hash(key);
    for (;;) {
        const uint32_t old = atomicCAS(&sc.hkey[h], kEmpty, key);
        if (old == kEmpty || old == key) break;
        h = (h + 1) & (kOccHash - 1);
    }
    sc.hslot[idx] = h;
    sc.hnext[idx] = atomicExch(&sc.hhead[h], idx);
}

// grid barrier of <= 64 one-wave workgroups: one monotonic counter (MI355X_MICROARCH.md, barrier-counter)
__device__ __forceinline__ void par_barrier(uint32_t *ctr, uint32_t &target, uint32_t nwg, int lane)
{
    wave_sync_full();                                     // every lane's stores have completed
    target += nwg;
    if (lane == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        while ((int32_t)(__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target) < 0) __builtin_amdgcn_s_sleep(4);
    }
    lds_order();                                          // (the wave reconverges; the fence below does the waiting)
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");    // what the others wrote before they arrived, and no stale L1 lines
    asm volatile("" ::: "memory");
}

// ---------------------------------------------------------------------------------------------------------
// One workgroup (one wavefront) per window node; grid = the window's look-ahead.  Node id belongs to workgroup
// id % grid for as long as it waits, so a dry run that the group just committed did not touch is KEPT: after a group
// only the nodes that conflicted with one of its members run again (the node that closed the group among them), and
// an iteration after the first costs one or two dry runs, not the slowest of the whole window.  Iterates dry run /
// validate / apply until the head's link plan is stale, the window is exhausted, or the head needs the host
// (ctl->stop).
// ---------------------------------------------------------------------------------------------------------
// HW > 0: every workgroup has HW helper wavefronts that share each recomputed select_neighbors with its committing
// wave (team_select, hnsw_occ.hpp), reading through the same overlay; they take no part in the grid barriers.
template <int MODE, int T, int R, int HW = 0>
__global__ __launch_bounds__(64 * (1 + HW), 1) void k_occ_commit_par(GraphView g, OccBufs ob, ParBufs pb, uint32_t end_node, uint32_t mlinks,
                                                       uint32_t lnb, uint32_t lcap, uint32_t *__restrict__ gspill, uint32_t gnb,
                                                       const uint32_t *__restrict__ plan, uint32_t slack, uint32_t own_lds = 0,
                                                       TeamCfg tc = TeamCfg{}, uint32_t chained = 0, uint32_t *__restrict__ touched = nullptr,
                                                       uint32_t touched_cap = 0)
{
    // touched (a window of ONE node: a single hnsw_add under tuning tie_mode 1): the update_fn list of core.rs:535-537, 787-816,
    // written by the dry run in the reference's order and published when the node is applied
    extern __shared__ __align__(16) unsigned char smem[];
    const int lane = threadIdx.x & 63;
    if (chained && (ob.ctl->stop >= OCC_STOP_RESTRIDE || ob.ctl->head >= end_node)) return;   // a round enqueued ahead of the host (occ_round_window)
    const uint32_t b = blockIdx.x, nwg = gridDim.x;
    OccScratch sc = occ_carve(smem);
    OverlayView ov;
    static_cast<GraphView &>(ov) = g;
    ov.ovkey = reinterpret_cast<uint32_t *>(smem + kOccScratchBytes);
    ov.ovctl = ov.ovkey + kParTab;
    ov.ovrows = pb.rows + (size_t)b * (kParTab + 1) * pb.ovstride;
    ov.ovstride = pb.ovstride;
    uint32_t *lmax_deg = ov.ovctl + 2;                       // [2], [3]: the largest degrees this dry run produced (layer 0 / upper)
    WaveMem m;
    Visited vis;
    carve<R, T, true>(smem + kOccScratchBytes + kParLdsBytes, g.dim, lnb, lcap, m, vis, g.tagcfg, g.selcap);
    volatile TeamTask *task = reinterpret_cast<volatile TeamTask *>(smem + kOccScratchBytes + kParLdsBytes + own_lds);
    uint64_t *W0sub = reinterpret_cast<uint64_t *>(smem + kOccScratchBytes + kParLdsBytes + own_lds + sizeof(TeamTask));
    unsigned char *hmem0 = smem + kOccScratchBytes + kParLdsBytes + own_lds + sizeof(TeamTask) + (size_t)kTeamCand * 8;
    if constexpr (HW > 0) {
        const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)threadIdx.x) >> 6;
        if (wave > 0) {                                      // HBM spill regions: [0, 64) the committing waves', then HW per workgroup
            team_helper<MODE, T>(ov, task, m.W, hmem0 + (size_t)(wave - 1) * tc.hbytes, tc, gspill + (size_t)(64u + b * HW) * gnb * 8, wave,
                                 1u + HW, lane);
            return;
        }
    }
    vis.glob = gspill + (size_t)b * gnb * 8;
    vis.gnb = gnb;
    vis.glob_dirty = false;
    vis.spilled = false;
    vis.count = 0;
    vis.bounded = false;
    vis.lossy = false;

    OccPar *me = &pb.par[b];
    OccDelta *mydelta = pb.delta + (size_t)b * kParMaxDelta;
    uint32_t bar_target = ob.ctl->bar_start;
    occ_init_hash(sc, lane);
    unsigned long long n_groups = 0, n_dry = 0, n_conf_link = 0, n_conf_rec = 0, n_conf_row = 0;   // (groups: workgroup 0 keeps the round's)
    unsigned long long n_early = 0;                          // rounds ended right after a group (no iteration spent on finding the head not ready)
    unsigned long long prof[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // workgroup 0, 100 MHz ticks: dry run, wait, validate, wait, apply, wait; [6] iterations, [7] launches
    unsigned long long t_ = wall_clock64();
#define PAR_T(i) do { const unsigned long long n_ = wall_clock64(); prof[i] += n_ - t_; t_ = n_; } while (0)

    unsigned long long dprof[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long d_ = 0;
#define DRY_T(i) do { const unsigned long long n_ = wall_clock64(); dprof[i] += n_ - d_; d_ = n_; } while (0)
    // the dry run this workgroup holds (kept across iterations while nothing committed touches it)
    bool have = false, redo = false;
    uint32_t cur_id = kEmpty, kept_state = PAR_NONE, kept_nt = 0;
    uint64_t live = 0;                                       // sub-operations whose reads stand for the dry run
    uint32_t n_hash = 0;                                     // entries in the LDS hash
    uint32_t n_delta = 0, promotes = 0;
    uint32_t n_spec = 0, n_fallback = 0, n_norec = 0, n_tie = 0;
    unsigned long long w_dist = 0, w_ids = 0, w_skipped = 0;

    for (;;) {
        const uint32_t head = __hip_atomic_load(&ob.ctl->head, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const uint32_t nJ = __hip_atomic_load(&ob.ctl->nJ, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const uint32_t epoch = __hip_atomic_load(&ob.ctl->epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const uint32_t pos = (b + nwg - head % nwg) % nwg;   // this workgroup's node is the pos-th of the window
        const uint32_t id = head + pos;
        const uint32_t slot = id % ob.W;
        OccSlot *sl = &ob.slots[slot];
        const OccRead *reads = ob.reads + (size_t)slot * kOccMaxReads;
        const OccShr *shr = ob.shr + (size_t)slot * kOccMaxShr;

        // ------------------------------------------------------------------ dry run (or the one kept from before)
        if (have && (cur_id != id || redo)) {                // the node went into the last group, or the group touched what it read
            if (n_hash) occ_clear_hash(sc, n_hash, lane);
            n_hash = 0;
            have = false;
        }
        uint32_t state = PAR_NONE;
        unsigned long long dry_ticks = 0;
        if (id < end_node) {
            if (!sl->planned || sl->node != id) state = PAR_REPLAN;
            else if (sl->fail) state = PAR_SERIAL;
            else if (sl->epoch != epoch) { if (lane == 0) sl->planned = 0; state = PAR_REPLAN; }
            else if (g.hdr->max_deg0 + slack > g.stride0 - 1 || g.hdr->max_degU + slack > g.strideU - 1) state = PAR_RESTRIDE;
            else if (have) state = kept_state;
            else {
                const uint32_t n_shr = sl->n_shr;
                d_ = wall_clock64();
                dry_ticks = d_;
                live = 0; n_delta = 0; promotes = 0; n_spec = n_fallback = n_norec = n_tie = 0; w_dist = w_ids = w_skipped = 0;
                n_hash = sl->n_reads;
                occ_build_hash(sc, reads, n_hash, shr, n_shr, lane);
                occ_check_range<MODE, T>(g, m, sc, ob, reads, shr, id, sl->snapU, nJ, lane, false, kEmpty, nullptr, sl->snap);
                if (sc.flags[0]) {                           // the link plan is stale against what is committed: re-plan
                    if (lane == 0) { sl->planned = 0; sl->stage = 0u; }
                    state = PAR_REPLAN;
                } else {
                    n_dry += 1;
                    DRY_T(0);
                    for (uint32_t i = lane; i < kParTab; i += 64) ov.ovkey[i] = kEmpty;
                    if (lane < 4) ov.ovctl[lane] = 0u;
                    wave_sync_full();
                    OccJournal jr;
                    jr.ring = mydelta;
                    jr.n = 0;
                    jr.own = sc.own;
                    jr.own_base = 0;
                    jr.cap = kParMaxDelta;
                    uint32_t checked = 0;                    // own deltas already checked against the records
                    uint64_t done_rec = 0;                   // speculative records this dry run has already consumed
                    uint32_t n_tie_used = 0;
                    uint32_t nsub = n_shr;                   // the next recomputation's sub-operation
                    bool fail = false, overflow = false;
                    uint32_t nt = 0;
                    const uint32_t lmax = g.hdr->max_layer;
                    const uint32_t l = g.levels[id];
                    const uint32_t top = sl->top;
                    const uint32_t *pl0 = plan + (size_t)slot * kMaxLayers * g.plan_stride;
                    for (uint32_t lc1 = top + 1; lc1-- > 0 && !fail && !overflow;) {     // core.rs:523
                        const uint32_t lc = lc1;
                        const uint32_t stride = lc ? g.strideU : g.stride0;
                        const uint32_t mmax = lc ? mlinks : 2 * mlinks;     // core.rs:560
                        uint32_t *maxdeg = lmax_deg + (lc ? 1 : 0);
                        const uint32_t *pl = pl0 + (size_t)lc * g.plan_stride;
                        const uint32_t nsel = pl[0];
                        const uint32_t myselid = (uint32_t)lane < nsel ? pl[1 + lane] : kEmpty;
                        // connect_neighbors (core.rs:759-774), nearest first
                        uint32_t *qrow = row_mut(ov, id, lc, lane);
                        uint32_t *nrow = ov_own_lanes(ov, (uint32_t)lane < nsel, myselid, lc, lane);
                        if (lane == 0) qrow[0] = nsel;
                        if ((uint32_t)lane < nsel) {
                            qrow[1 + lane] = myselid;
                            const uint32_t c = row_ptr(g, myselid, lc)[0];     // (the graph's count: the copy holds the same)
                            if (c + 1 > stride - 1) atomicOr(&g.hdr->status, ST_ROW_OVERFLOW);
                            else { nrow[1 + c] = id; nrow[0] = c + 1; atomicMax(maxdeg, c + 1); }
                        }
                        if (lane == 0) atomicMax(maxdeg, nsel);
                        touch_push(touched, touched_cap, nt, myselid, touched != nullptr && (uint32_t)lane < nsel, lane);   // :535-537
                        journal_push(&jr, (uint32_t)lane < nsel, myselid, lc, id, true, lane);
                        fence_own_writes();
                        wave_sync_full();
                        DRY_T(1);

                        for (uint32_t si = 0; si < nsel && !fail && !overflow; ++si) {   // shrink loop (core.rs:540-574), e nearest first
                            // more deltas than the private buffer holds (journal_push counts them, it does not write them): the
                            // dry run is void from here on -- and nothing may read mydelta[] past its end
                            if (jr.n > kParMaxDelta) { overflow = true; break; }
                            const uint32_t e = pl[1 + si];
                            uint32_t *erow = row_mut(ov, e, lc, lane);       // (in the overlay since the connect)
                            uint32_t cnt = erow[0];
                            if (cnt > stride - 1) cnt = stride - 1;
                            if (cnt <= mmax) { w_skipped += cnt; continue; }   // :561
                            int k = -1;
                            {
                                const bool mine = (uint32_t)lane < n_shr && shr[lane].e == e && shr[lane].lc == lc && shr[lane].nS != 0;
                                const uint64_t mb = __ballot(mine);
                                if (mb) k = 63 - __builtin_clzll((unsigned long long)mb);
                            }
                            if (k >= 0 && checked != jr.n) {
                                occ_check_range<MODE, T>(g, m, sc, ob, reads, shr, id, checked, jr.n, lane, true, 0u, mydelta, kEmpty, done_rec);
                                checked = jr.n;
                            }
                            if (k >= 0) done_rec |= 1ull << k;       // (its verdict is taken below; later deltas need not look at it)
                            DRY_T(2);
                            for (uint32_t i = lane; i < cnt; i += 64) m.aux[i] = erow[1 + i];
                            wave_sync_full();
                            uint32_t nS;
                            if (k >= 0 && !sc.flags[2 + k]) {
                                const uint32_t sv = shr[k].S[lane], sv2 = shr[k].S[64 + lane];
                                nS = shr[k].nS;
                                if ((uint32_t)lane < nS) m.S[lane] = (uint64_t)sv << 1;
                                if (64u + (uint32_t)lane < nS) m.S[64 + lane] = (uint64_t)sv2 << 1;
                                wave_sync_full();
                                n_spec += 1;
                                w_dist += shr[k].w_dist;
                                w_ids += shr[k].w_ids;
                                n_tie_used += shr[k].tie;            // (tie gate: a record whose selection met a tie)
                                live |= 1ull << k;
                                DRY_T(3);
                            } else {
                                // recompute on the spot (core.rs:544-568), reading through the overlay
                                QReg<T> qe;
                                load_query<MODE, T>(g.vec + (size_t)e * g.dim, g.dim, qe, m.qlds, lane);
                                uint32_t nE = 0;
                                for (uint32_t base = 0; base < cnt; base += 64) {
                                    const uint32_t i = base + lane;
                                    const uint32_t nf = cnt - base < 64 ? cnt - base : 64;
                                    if (i < cnt) m.fresh[lane] = m.aux[i];
                                    wave_sync_full();
                                    compute_dists<MODE, T>(g, qe, m, nf, lane);
                                    wave_sync_full();
                                    const bool hv = (uint32_t)lane < nf;
                                    const uint64_t key = hv ? pack_key(m.dsc[lane], m.fresh[lane]) : ~0ull;
                                    nE = merge_sorted<R>(m.W, nE, R * 64, key, hv, lane);
                                }
                                WorkCtr nolog = {};
                                if constexpr (HW > 0)
                                    nS = team_select<MODE, T>(ov, m, vis, qe, nE, e, mmax, lc, nolog, lane, fail, kEmpty, task, W0sub, hmem0, tc, 1u + HW);
                                else
                                    nS = select_topm<MODE, T>(ov, m, vis, qe, m.W, nE, e, mmax, lc, nolog, lane, fail);
                                if (fail) break;
                                w_dist += cnt + nolog.n_dist;
                                w_ids += cnt + nolog.n_ids;
                                n_tie += nolog.n_tie;            // tie census (counted when the dry run commits)
                                n_fallback += 1;
                                if (k < 0) n_norec += 1;
                                // what it read, for the nodes committed alongside: the rows of e's members, bound = the last
                                // selected (row e itself is one of the rows this dry run rewrites)
                                if (nsub >= kParMaxSub || n_hash + cnt > kOccMaxReads || nS == 0) overflow = true;
                                else {
                                    const uint32_t bound = (uint32_t)(m.S[nS - 1] >> 32);
                                    const uint32_t meta = occ_meta(lc, OCC_SHRINK_NB, nsub, nS >= mmax);
                                    for (uint32_t i = lane; i < cnt; i += 64) occ_hash_add(sc, n_hash + i, m.aux[i], meta, bound);
                                    if (lane == 0) { sc.flags[2 + nsub] = 0u; sc.flags[2 + kOccMaxShr + nsub] = e; }
                                    n_hash += cnt;
                                    live |= 1ull << nsub;
                                    nsub += 1;
                                    wave_sync_full();
                                }
                                DRY_T(4);
                            }
                            update_connections(ov, m, e, erow, cnt, nS, lc, stride, maxdeg, kEmpty, touched, touched_cap, nt, lane, &jr);
                            DRY_T(5);
                        }
                    }
                    if (vis.glob_dirty) visited_clear(vis, lane);
                    wave_sync_full();
                    // tie gate (tuning tie_mode 1): a commit whose selections met a tie is not applied: the host redoes the insert in
                    // the reference binary's own tie order (hnsw_std_heap.hpp)
                    const bool tied = ob.ctl->tie_gate && (n_tie || n_tie_used);
                    if (fail || overflow || tied || ov.ovctl[1] || jr.n > kParMaxDelta || n_hash + ov.ovctl[0] > kOccMaxReads) state = PAR_SERIAL;
                    else {
                        // the rows this dry run rewrote join what it "read": any delta of an earlier node on one of them is a conflict
                        const uint32_t meta = occ_meta(0, OCC_SHRINK_ROW, kParSubRows, true);
                        uint32_t nrows = 0;
                        for (uint32_t base = 0; base < kParTab; base += 64) {
                            const uint32_t key = ov.ovkey[base + lane];
                            const bool used = key != kEmpty;
                            const uint64_t um = __ballot(used);
                            if (used) occ_hash_add(sc, n_hash + nrows + (uint32_t)__popcll(um & lanemask_lt(lane)), key >> 5,
                                                   (meta & ~31u) | (key & 31u), 0u);
                            nrows += (uint32_t)__popcll(um);
                        }
                        n_hash += nrows;
                        live |= 1ull << kParSubRows;
                        state = PAR_READY;
                        n_delta = jr.n;
                        promotes = l > lmax ? 1u : 0u;
                    }
                    wave_sync_full();
                    DRY_T(6);
                    dry_ticks = d_ - dry_ticks;
                    have = true;                             // kept until a group touches it (a void dry run, PAR_SERIAL, as well)
                    cur_id = id;
                    kept_state = state;
                    kept_nt = nt;
                }
            }
        }
        redo = false;
        if (lane == 0) {
            me->state = state; me->conflict = 0u; me->promotes = state == PAR_READY ? promotes : 0u; me->n_delta = state == PAR_READY ? n_delta : 0u;
            me->why = 0u; me->pad0 = (uint32_t)(dry_ticks < 0xFFFFFFu ? dry_ticks : 0u); me->pad1 = 0u;
        }
        PAR_T(0);
        par_barrier(&ob.ctl->bar, bar_target, nwg, lane);
        PAR_T(1);
        {   // the head of the window is not ready (its link plan is stale, or it needs the host): the round ends here
            const uint32_t hs = pb.par[head % nwg].state;
            if (hs != PAR_READY) {
                if (b == 0 && lane == 0)
                    ob.ctl->stop = hs == PAR_SERIAL ? OCC_STOP_SERIAL : hs == PAR_RESTRIDE ? OCC_STOP_RESTRIDE : OCC_STOP_REPLAN;
                prof[6] += 1;
                break;
            }
        }

        // ------------------------------------------------------------------ validate against the nodes before this one
        uint32_t conflict = 0, why = 0, fc = kEmpty;        // fc: position of the first earlier node this one cannot commit beside
        if (state == PAR_READY && pos > 0) {
            for (uint32_t i = lane; i < 2 + kOccMaxShr; i += 64) sc.flags[i] = 0u;
            wave_sync_full();
            for (uint32_t x = 0; x < pos && !conflict; ++x) {
                const uint32_t wx = (head + x) % nwg;
                const OccPar *pp = &pb.par[wx];
                if (pp->state != PAR_READY) { conflict = 1; why = 8; fc = x; break; }   // the group cannot reach this node anyway
                const uint32_t nd = pp->n_delta;
                occ_check_range<MODE, T>(g, m, sc, ob, reads, shr, id, 0u, nd, lane, false, kEmpty, pb.delta + (size_t)wx * kParMaxDelta);
                if (sc.flags[0]) { conflict = 1; why = 1; fc = x; }
                else {
                    const uint32_t f = (uint32_t)lane < kOccMaxShr && ((live >> lane) & 1ull) ? sc.flags[2 + lane] : 0u;
                    const uint64_t fm = __ballot(f != 0u);
                    if (fm) { conflict = 1; why = (fm >> kParSubRows) & 1ull ? 4 : 2; fc = x; }
                }
            }
        }
        if (lane == 0) { me->conflict = conflict; me->why = why; }
        PAR_T(2);
        par_barrier(&ob.ctl->bar, bar_target, nwg, lane);
        PAR_T(3);

        // ------------------------------------------------------------------ the group: the longest conflict-free prefix
        uint32_t p = 0, my_off = 0, total = 0, promoter = kEmpty, close_why = 0, st0 = PAR_NONE, stp = PAR_NONE;
        {
            const bool in = (uint32_t)lane < nwg;
            const OccPar *pp = &pb.par[in ? (head + (uint32_t)lane) % nwg : 0u];     // lane = position in the window
            const uint32_t st = in ? pp->state : PAR_NONE, cf = in ? pp->conflict : 1u, pr = in ? pp->promotes : 0u;
            const uint32_t nd = in ? pp->n_delta : 0u, wy = in ? pp->why : 0u;
            {   // the slowest dry run of the iteration
                uint32_t mx = in ? pp->pad0 : 0u;
                for (int d = 32; d >= 1; d >>= 1) { const uint32_t o = (uint32_t)__shfl_xor((int)mx, d, 64); mx = o > mx ? o : mx; }
                if (b == 0) dprof[7] += mx;
            }
            const uint64_t bad = __ballot(!in || st != PAR_READY || cf != 0u);
            p = bad ? (uint32_t)(__ffsll((unsigned long long)bad) - 1) : nwg;
            const uint64_t prm = __ballot(in && pr != 0u) & (p >= 64 ? ~0ull : lanemask_lt((int)p));
            if (prm) { promoter = (uint32_t)(__ffsll((unsigned long long)prm) - 1); p = promoter + 1; }
            // journal offsets: prefix sum of the group's delta counts
            uint32_t run = ((uint32_t)lane < p) ? nd : 0u;
            for (int d = 1; d < 64; d <<= 1) { const uint32_t o = (uint32_t)__shfl_up((int)run, d, 64); if (lane >= d) run += o; }
            total = (uint32_t)__shfl((int)run, 63, 64);
            const uint32_t excl = run - (((uint32_t)lane < p) ? nd : 0u);
            my_off = (uint32_t)__shfl((int)excl, (int)(pos < 64 ? pos : 0), 64);
            close_why = p < nwg ? (uint32_t)__shfl((int)wy, (int)p, 64) : 0u;
            st0 = (uint32_t)__shfl((int)st, 0, 64);
            stp = p < nwg ? (uint32_t)__shfl((int)st, (int)p, 64) : PAR_NONE;   // the node that closed the group: the next head
        }
        // ---- does the round end with this group?  The next iteration would start by finding out whether the node that
        // closed the group -- the next head -- can still commit, after every window node has made its dry run (~250 us
        // nobody needs: every round ends that way).  Most of the answer is known now: a next head that needs a new plan or
        // wider rows ends the round whatever happens, and whether its LINK PLAN survives the group is one more look at the
        // group's deltas -- the validate phase stopped at the first member it conflicts with, the members after that are
        // checked here, by that node's own workgroup, while the members write their rows.
        if (p && pos == p && state == PAR_READY) {
            bool stale = why == 1u;
            for (uint32_t x = fc + 1u; x < p && !stale; ++x) {
                const uint32_t wx = (head + x) % nwg;
                occ_check_range<MODE, T>(g, m, sc, ob, reads, shr, id, 0u, pb.par[wx].n_delta, lane, false, kEmpty, pb.delta + (size_t)wx * kParMaxDelta);
                stale = sc.flags[0] != 0u;
            }
            if (stale && lane == 0) { me->pad1 = 1u; sl->planned = 0; sl->stage = 0u; }
        }
        if (state == PAR_READY && pos < p) {
            if (touched && lane == 0) g.hdr->n_touched = kept_nt;
            // write the overlay rows back, eight at a time
            for (uint32_t base = 0; base < kParTab; base += 64) {
                const uint32_t key = ov.ovkey[base + lane];
                uint64_t um = __ballot(key != kEmpty);
                while (um) {
                    uint32_t w[8];
                    uint32_t *dst[8];
                    const uint32_t *src[8];
                    uint32_t st[8];
                    int n = 0;
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        dst[u] = nullptr; src[u] = nullptr; st[u] = 0;
                        if (um) {
                            const int j = __ffsll((unsigned long long)um) - 1;
                            um &= um - 1;
                            const uint32_t kk = (uint32_t)__builtin_amdgcn_readlane((int)key, j);
                            src[u] = ov.ovrows + (size_t)(base + j) * ov.ovstride;
                            dst[u] = row_ptr(g, kk >> 5, kk & 31u);
                            st[u] = (kk & 31u) ? g.strideU : g.stride0;
                            n = u + 1;
                        }
                    }
#pragma unroll
                    for (int u = 0; u < 8; ++u) w[u] = (u < n && (uint32_t)lane < st[u]) ? src[u][lane] : 0u;
#pragma unroll
                    for (int u = 0; u < 8; ++u) {                     // the live words only: count + ids (the rest of a scratch row is undefined)
                        uint32_t cu = (uint32_t)__builtin_amdgcn_readfirstlane((int)w[u]);
                        if (cu > st[u] - 1) cu = st[u] - 1;
                        st[u] = u < n ? cu + 1 : 0u;
                        if ((uint32_t)lane < st[u]) dst[u][lane] = w[u];
                    }
                    for (int u = 0; u < n; ++u)
                        for (uint32_t i = 64 + lane; i < st[u]; i += 64) dst[u][i] = src[u][i];
                }
            }
            // ... and the deltas into the journal ring, where the in-order wave would have put them
            for (uint32_t i = lane; i < n_delta; i += 64) ob.ring[(nJ + my_off + i) & ((1u << kOccJournalBits) - 1u)] = mydelta[i];
            if (lane == 0) {
                sl->planned = 0;
                atomicMax(&g.hdr->max_deg0, lmax_deg[0]);
                atomicMax(&g.hdr->max_degU, lmax_deg[1]);
                atomicAdd(&g.hdr->ctr_insert[0], w_dist);
                atomicAdd(&g.hdr->ctr_insert[1], w_ids);
                atomicAdd(&g.hdr->ctr_insert[3], w_skipped);
                if (n_tie) atomicAdd(&g.hdr->ctr_tie[2], (unsigned long long)n_tie);
                atomicAdd(&ob.ctl->n_spec, (unsigned long long)n_spec);
                atomicAdd(&ob.ctl->n_fallback, (unsigned long long)n_fallback);
                atomicAdd(&ob.ctl->n_norec, (unsigned long long)n_norec);
            }
        } else if (have && fc < p) redo = true;            // a member of the group touched what this dry run read or rewrote
        if (b == 0) {
            if (p) n_groups += 1;
            if (close_why == 1) n_conf_link += 1; else if (close_why == 2) n_conf_rec += 1; else if (close_why == 4) n_conf_row += 1;
            if (lane == 0) {
                if (p) {
                    ob.ctl->n_commit += p;
                    g.hdr->node_count = head + p;
                    if (promoter != kEmpty) {                    // core.rs:587-593
                        g.hdr->max_layer = g.levels[head + promoter];
                        g.hdr->enterpoint = (int32_t)(head + promoter);
                        ob.ctl->epoch = epoch + 1;
                    }
                    __hip_atomic_store(&ob.ctl->nJ, nJ + total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(&ob.ctl->head, head + p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    ob.ctl->stop = OCC_STOP_NONE;
                } else {
                    ob.ctl->stop = st0 == PAR_SERIAL ? OCC_STOP_SERIAL : st0 == PAR_RESTRIDE ? OCC_STOP_RESTRIDE : OCC_STOP_REPLAN;
                }
            }
        }
        PAR_T(4);
        par_barrier(&ob.ctl->bar, bar_target, nwg, lane);
        PAR_T(5);
        prof[6] += 1;
        if (p == 0 || head + p >= end_node) break;
        {   // the next head cannot commit: the round ends here (see above)
            uint32_t nstop = stp == PAR_REPLAN ? (uint32_t)OCC_STOP_REPLAN : stp == PAR_RESTRIDE ? (uint32_t)OCC_STOP_RESTRIDE : (uint32_t)OCC_STOP_NONE;
            if (nstop == OCC_STOP_NONE && p < nwg && stp == PAR_READY && pb.par[(head + p) % nwg].pad1) nstop = OCC_STOP_REPLAN;
            if (nstop != OCC_STOP_NONE) {
                if (b == 0 && lane == 0) ob.ctl->stop = nstop;
                if (b == 0) n_early += 1;
                break;
            }
        }
    }
    if constexpr (HW > 0) {
        if (lane == 0) task->op = TEAM_EXIT;
        team_bar();
    }
    if (lane == 0 && n_dry) {
        atomicAdd(&ob.ctl->n_dry, n_dry);
        for (int i = 0; i < 7; ++i) atomicAdd(&ob.ctl->dry_prof[i], dprof[i]);
    }
    if (b == 0 && lane == 0) atomicAdd(&ob.ctl->dry_prof[7], dprof[7]);
    if (b == 0 && lane == 0) {
        ob.ctl->bar_start = bar_target;
        ob.ctl->rounds += 1;
        ob.ctl->n_groups += n_groups;
        ob.ctl->n_conf_link += n_conf_link;
        ob.ctl->n_conf_rec += n_conf_rec;
        ob.ctl->n_conf_row += n_conf_row;
        ob.ctl->n_early += n_early;
        prof[7] = 1;
        for (int i = 0; i < 8; ++i) ob.ctl->par_prof[i] += prof[i];
    }
#undef PAR_T
#undef DRY_T
}

} // namespace hnsw
