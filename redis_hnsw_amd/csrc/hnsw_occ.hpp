// hnsw_occ.hpp -- HNSW.NODE.ADD in the reference's serial order (core.rs:489-599), executed optimistically:
// a window of consecutive inserts is PLANNED in parallel against one snapshot of the graph, then COMMITTED
// strictly in id order by one wave, which first checks that nothing the plan read has changed in a way that
// matters.  The result is the serial graph, row for row (tests: graphs_equal against the oracle); the scheme
// and its validation rules were first proven on the CPU (tests/experiments/occ_model.c).
//
//   plan      k_occ_plan: per node the read-only part of insert(): descent, per layer search_level(efc) +
//             select_neighbors(m) (core.rs:511-531) -- with a READ LOG (row, layer, accept bound in force) --
//             the node's own rows, and speculatively the select_neighbors(m_max) of every selected neighbour e
//             that the connect will push over m_max (core.rs:560-568), computed as if only this node's connect
//             had happened since the snapshot.
//   journal   every committed change of a row is appended as (row, layer, id, added | removed).
//   validate  a plan is still the reference's plan if no journal entry since its snapshot is RELEVANT to a
//             row it read.  Entry z on row h is relevant to
//               a search_level expansion of h   iff W was not full then, or dist(q,z) <= W's furthest  (core.rs:657)
//               select_neighbors over h          iff fewer than m were selected, or dist(q,z) <= the last selected
//               a speculative shrink of e        iff h == e (any change but this node's own append), or h is one of
//                                                e's neighbours and dist(e,z) <= the last selected
//             (an irrelevant z is rejected by the same comparison whenever it is met, now or later: the accept
//             threshold only improves).  Ties count as relevant.
//   commit    k_occ_commit: in id order -- validate, connect (core.rs:532), then per selected neighbour the
//             shrink: the speculative result if still valid, else recomputed on the spot (core.rs:540-574).
//             The first node whose link plan is stale ends the round; the host re-plans what is stale.
#pragma once
#include "hnsw_insert.hpp"

namespace hnsw {

constexpr uint32_t kOccMaxW = 64;           // window slots
constexpr uint32_t kOccMaxReads = 3072;     // read-log entries per slot
constexpr uint32_t kOccMaxShr = 64;         // speculative records per slot (array stride; one per lane of the committing wave)
constexpr uint32_t kOccInsShr = 32;         // ... of which an INSERT lists at most this many (window slots x this <= HBM spill slots);
                                            // a delete may list all 64 (one re-selection per neighbour of the node, all its layers)
constexpr uint32_t kOccHash = 4096;         // validation hash slots
constexpr uint32_t kOccOwn = 256;           // own deltas of one commit mirrored in LDS
constexpr uint32_t kOccMaxHits = 1024;      // (reader, z) pairs whose distance must be evaluated per validation

struct OccShr {
    uint32_t lc, e, nS, bound;              // bound = distance bits of the last selected; nS = 0: not computed yet
    uint32_t log0, cnt, w_dist, w_ids;      // its read-log range starts at log0: 1 + cnt + 1 entries (row e, its members, this node); w_*: evaluations / ids scanned computing it
    uint32_t tie, pad0, pad1, pad2;         // tie census: computing it met equal distances (a cut, or an order inside the selection)
    uint32_t S[kSelMax];                    // selected ids, nearest first (up to m_max0 = 2M)
};
struct OccSlot {
    uint32_t node, planned, snap, epoch;
    uint32_t n_reads, n_shr, top, fail;     // fail: the plan could not be logged (log overflow, visited overflow)
    // A plan in two stages (k_occ_plan_lean): a node with a level >= 1 that is far from the window's head has its upper
    // layers searched a round EARLY (stage 1: descent, layers top..1, their plan rows and read-log entries, the entry
    // point of layer 0); the next round's plan kernel does layer 0 only -- so no launch waits for one node's two full
    // searches.  The reads of layers >= 1 date from snapU, those of layer 0 (and the speculative records) from snap:
    // occ_check_range ignores layer-0 deltas older than snap.
    uint32_t stage, snapU, ep_l0, n_reads_u;
    uint32_t tie, pad0, pad1, pad2;         // tie census of the stages planned so far (tie gate: OccCtl::tie_gate)
};
enum { OCC_WHY_ROW = 1, OCC_WHY_OPEN = 2, OCC_WHY_ADD = 4, OCC_WHY_REMOVE = 8 };   // why a speculative shrink is stale (flags[2 + sub]; the first cause met, later entries are not looked at)
enum { OCC_STOP_NONE = 0, OCC_STOP_REPLAN = 1, OCC_STOP_RESTRIDE = 2, OCC_STOP_SERIAL = 3 };
struct OccCtl {
    uint32_t head;                          // next node id to commit
    uint32_t nJ;                            // journal entries ever written
    uint32_t epoch;                         // bumped when enterpoint / max_layer change (every plan reads them)
    uint32_t stop;                          // why the last commit kernel stopped
    unsigned long long n_commit, n_spec, n_fallback, n_stale, n_norec, n_rowstale;
    unsigned long long n_cls[8];            // recomputed shrinks by cause: no record, row changed, pool not full, a relevant removal, relevant additions only
    unsigned long long prof[8];             // commit kernel, shader clocks: hash, first check, connect, shrink checks, apply, recompute, finish, total
    // parallel validated commits (hnsw_occ_par.hpp)
    uint32_t bar, bar_start;                // grid barrier of k_occ_commit_par: arrivals ever / the count the next launch starts from
    unsigned long long n_groups, n_dry, n_conf_link, n_conf_rec, n_conf_row;   // groups committed, dry runs made, groups closed by: a stale link plan / a record used / a changed row
    unsigned long long n_early;             // rounds that ended right after a group: the next head was known not to be able to commit
    unsigned long long dry_prof[8];         // all workgroups' dry runs, 100 MHz ticks: hash + journal check, connect, record checks, row load + spec apply, recompute, update_connections, finish; [7] = sum over iterations of the slowest dry run
    uint32_t tie_gate, pad_tg;              // tuning tie_mode 1: a node whose plan or commit met a tie (the census) is handed to the host untouched (OCC_STOP_SERIAL: the std-order kernel)
    uint32_t end_node, rounds;              // rounds enqueued ahead of the host (occ_round_window): where the chunk ends; commit launches that ran
    unsigned long long par_prof[8];         // k_occ_commit_par, workgroup 0, 100 MHz ticks: dry run, wait, validate, wait, apply, wait; [6] iterations, [7] launches
};

struct OccBufs {
    OccSlot *slots;
    OccRead *reads;                         // [W][kOccMaxReads]
    OccShr *shr;                            // [W][kOccMaxShr]
    OccDelta *ring;                         // [1 << kOccJournalBits]
    OccCtl *ctl;
    uint32_t W;
};

// A round enqueued AHEAD of the host (first_node == kEmpty): the host queues a few rounds back to back and synchronises
// once; such a round starts where the previous one stopped (ctl->head), and does nothing once a round has asked for the
// host (restride, a serial insert) or the chunk is done.  False = nothing to do for this launch.
__device__ __forceinline__ bool occ_round_window(const OccBufs &ob, uint32_t &first_node, uint32_t &count)
{
    if (first_node != kEmpty) return true;
    if (ob.ctl->stop >= OCC_STOP_RESTRIDE) return false;
    first_node = ob.ctl->head;
    const uint32_t end_node = ob.ctl->end_node;
    if (first_node >= end_node) return false;
    if (count > end_node - first_node) count = end_node - first_node;
    return true;
}

// LDS scratch of a validation
struct OccScratch {
    uint32_t *hkey;     // [kOccHash]
    uint32_t *hhead;    // [kOccHash]
    uint32_t *hnext;    // [kOccMaxReads]
    uint32_t *hslot;    // [kOccMaxReads] the slot each read went to (so that only used slots are cleared)
    uint32_t *rmeta;    // [kOccMaxReads] copies of the reads' meta / bound words (the chain walk stays in LDS)
    uint32_t *rbound;   // [kOccMaxReads]
    OccDelta *own;      // [kOccOwn] the committing node's own deltas (mirror of the journal tail)
    uint32_t *hits;     // [kOccMaxHits][3]: reader (0 = the node itself, 1 + sub = shrink sub), z, bound
    uint32_t *flags;    // [0] link plan stale, [1] hit count, [2 + sub] shrink sub stale
};
constexpr size_t kOccScratchBytes = (size_t)(kOccHash * 2 + kOccMaxReads * 4 + kOccOwn * 3 + kOccMaxHits * 3 + 2 + 2 * kOccMaxShr) * 4;

__device__ __forceinline__ OccScratch occ_carve(unsigned char *p)
{
    OccScratch s;
    s.hkey = reinterpret_cast<uint32_t *>(p);
    s.hhead = s.hkey + kOccHash;
    s.hnext = s.hhead + kOccHash;
    s.hslot = s.hnext + kOccMaxReads;
    s.rmeta = s.hslot + kOccMaxReads;
    s.rbound = s.rmeta + kOccMaxReads;
    s.own = reinterpret_cast<OccDelta *>(s.rbound + kOccMaxReads);
    s.hits = s.rbound + kOccMaxReads + kOccOwn * 3;
    s.flags = s.hits + kOccMaxHits * 3;
    return s;
}
__device__ __forceinline__ uint32_t occ_hash(uint32_t key) { return (key * 0x9E3779B1u) >> (32 - 12); }   // kOccHash = 2^12
static_assert(kOccHash == (1u << 12), "occ_hash");

// hash of the slot's read rows: (row, layer) -> chain of read indices
__device__ __forceinline__ void occ_init_hash(const OccScratch &sc, int lane)       // once per kernel
{
    for (uint32_t i = lane; i < kOccHash; i += 64) { sc.hkey[i] = kEmpty; sc.hhead[i] = kEmpty; }
    wave_sync();
}
__device__ __forceinline__ void occ_clear_hash(const OccScratch &sc, uint32_t n_reads, int lane)   // after a slot is done
{
    for (uint32_t i = lane; i < n_reads; i += 64) { const uint32_t h = sc.hslot[i]; sc.hkey[h] = kEmpty; sc.hhead[h] = kEmpty; }
    wave_sync();
}
__device__ __forceinline__ void occ_build_hash(const OccScratch &sc, const OccRead *reads, uint32_t n_reads, const OccShr *shr,
                                               uint32_t n_shr, int lane)
{
    for (uint32_t i = lane; i < 2 + 2 * kOccMaxShr; i += 64) sc.flags[i] = 0;
    wave_sync();
    // the shrinks' rows, for the chain walk: flags[2 + kOccMaxShr + sub] = shr[sub].e
    if ((uint32_t)lane < n_shr && (uint32_t)lane < kOccMaxShr) sc.flags[2 + kOccMaxShr + lane] = shr[lane].e;
    for (uint32_t i = lane; i < n_reads; i += 64) {
        const OccRead r = reads[i];
        sc.rmeta[i] = r.meta;
        sc.rbound[i] = r.bound;
        const uint32_t key = (r.row << 5) | (r.meta & 31u);
        uint32_t h = occ_hash(key);
        for (;;) {
            const uint32_t old = atomicCAS(&sc.hkey[h], kEmpty, key);
            if (old == kEmpty || old == key) break;
            h = (h + 1) & (kOccHash - 1);
        }
        sc.hslot[i] = h;
        sc.hnext[i] = atomicExch(&sc.hhead[h], i);
    }
    wave_sync();
}

// Check journal[from, to) against the hashed reads.  q = the slot's node; own_connect: entries (row,+q) are
// this node's own appends (its shrinks were planned with them in place).  Accumulates into sc.flags.
template <int MODE, int T>
__device__ __forceinline__ void occ_check_range(const GraphView &g, const WaveMem &m, const OccScratch &sc, const OccBufs &ob,
                                                const OccRead *reads, const OccShr *shr, uint32_t q, uint32_t from,
                                                uint32_t to, int lane, bool shrinks_only = false, uint32_t own_base = kEmpty,
                                                const OccDelta *src = nullptr, uint32_t from0 = kEmpty, uint64_t skip = 0)
{
    // skip: speculative records (bit = sub-operation) whose verdict nobody will ask for any more -- the commit has passed
    // their neighbour: their hits are not collected, their distances not evaluated
    // from0 != kEmpty: the plan read layer 0 later than the upper layers (OccSlot): layer-0 deltas before journal index
    // from0 were already in the graph when it did
    // src: the deltas come from this linear buffer (another node's dry run, hnsw_occ_par.hpp) instead of the journal ring
    (void)reads;
    const OccDelta *jsrc = src ? src : ob.ring;
    const uint32_t jmask = src ? 0xFFFFFFFFu : (1u << kOccJournalBits) - 1u;
    if (to - from > (1u << kOccJournalBits) - 4096u) {      // the ring has wrapped past this plan: stale
        if (lane == 0) sc.flags[0] = 1;
        wave_sync();
        return;
    }
    // ---- collect: entries on rows the plan read ----
    for (uint32_t base = from; base < to; base += 64) {
        const uint32_t j = base + lane;
        if (j < to) {
            // the committing node's own entries are mirrored in LDS (own_base = journal index of own[0])
            const OccDelta d = (j >= own_base && j - own_base < kOccOwn) ? sc.own[j - own_base] : jsrc[j & jmask];
            const bool old0 = from0 != kEmpty && (d.lc_add & 31u) == 0u && (int32_t)(j - from0) < 0;
            const uint32_t key = old0 ? kEmpty - 1u : (d.row << 5) | (d.lc_add & 31u);   // (a key no read has)
            const bool add = (d.lc_add & 256u) != 0;
            uint32_t h = occ_hash(key);
            while (sc.hkey[h] != kEmpty && sc.hkey[h] != key) h = (h + 1) & (kOccHash - 1);
            if (sc.hkey[h] == key) {
                for (uint32_t i = sc.hhead[h]; i != kEmpty; i = sc.hnext[i]) {
                    const uint32_t rm = sc.rmeta[i];
                    const uint32_t kind = (rm >> 5) & 3u, sub = (rm >> 7) & 63u;
                    const bool full = (rm >> 13) & 1u;
                    if (kind != OCC_SEARCH && kind != OCC_SELECT && ((skip >> sub) & 1ull)) continue;
                    if (kind == OCC_SHRINK_ROW) {
                        if (d.z == q && add) continue;                   // this node's own connect
                        if (!sc.flags[2 + sub]) atomicAdd(&ob.ctl->n_rowstale, 1ull);
                        atomicOr(&sc.flags[2 + sub], OCC_WHY_ROW);
                        continue;
                    }
                    if (kind == OCC_SHRINK_NB) {
                        if (d.z == sc.flags[2 + kOccMaxShr + sub] || d.z == q) continue;   // e is excluded (core.rs:704); q is in econn already
                        if (sc.flags[2 + sub]) continue;
                        if (!full) { atomicOr(&sc.flags[2 + sub], OCC_WHY_OPEN); continue; }
                    } else {
                        if (shrinks_only) continue;                      // the link plan was already accepted
                        if (!full) { sc.flags[0] = 1; continue; }
                    }
                    const uint32_t p = atomicAdd(&sc.flags[1], 1u);
                    if (p < kOccMaxHits) {
                        sc.hits[3 * p] = (kind == OCC_SHRINK_NB ? 1u + sub : 0u) | (add ? 0x100u : 0u);
                        sc.hits[3 * p + 1] = d.z;
                        sc.hits[3 * p + 2] = sc.rbound[i];
                    } else if (kind == OCC_SHRINK_NB) atomicOr(&sc.flags[2 + sub], OCC_WHY_ROW);   // too many to evaluate: that shrink is recomputed
                    else sc.flags[0] = 1;                                // ... the link plan is treated as stale
                }
            }
        }
    }
    wave_sync();
    // ---- evaluate: dist(reader, z) <= bound ? ----
    uint32_t nh = sc.flags[1];
    if (nh > kOccMaxHits) nh = kOccMaxHits;
    for (uint32_t base = 0; base < nh; base += 64) {
        const uint32_t i = base + lane;
        bool active = i < nh;
        const uint32_t reader_w = active ? sc.hits[3 * i] : 0u, z = active ? sc.hits[3 * i + 1] : 0u;
        const uint32_t reader = reader_w & 0xFFu;
        const bool was_add = (reader_w & 0x100u) != 0;
        const uint32_t bound = active ? sc.hits[3 * i + 2] : 0u;
        // a shrink already known stale needs no more distances
        if (active && reader && sc.flags[1 + reader]) active = false;
        uint64_t am = __ballot(active);
        while (am) {
            const int first = __ffsll((unsigned long long)am) - 1;
            const uint32_t rd = (uint32_t)__builtin_amdgcn_readlane((int)reader, first);
            const bool sel = active && reader == rd;
            const uint64_t sm = __ballot(sel);
            const uint32_t nf = (uint32_t)__popcll(sm);
            const uint32_t idx = (uint32_t)__popcll(sm & lanemask_lt(lane));
            lds_order();                                  // (everything handed over in this loop lives in LDS or registers)
            if (sel) m.fresh[idx] = z;
            const uint32_t rid = rd ? sc.flags[2 + kOccMaxShr + rd - 1] : q;
            QReg<T> qr;
            load_query<MODE, T>(g.vec + (size_t)rid * g.dim, g.dim, qr, m.qlds, lane);
            lds_order();                                  // the reader's vector and the gather below are in flight together
            compute_dists<MODE, T>(g, qr, m, nf, lane);
            lds_order();
            if (sel && __float_as_uint(m.dsc[idx]) <= bound) {
                if (rd) atomicOr(&sc.flags[1 + rd], was_add ? OCC_WHY_ADD : OCC_WHY_REMOVE);
                else sc.flags[0] = 1;
            }
            am &= ~sm;
        }
    }
    if (lane == 0) sc.flags[1] = 0;
    wave_sync();
}

// The tail of a plan: the shrinks the connect will trigger (core.rs:560-561) are listed here (k_occ_shrinks computes
// them speculatively, one wave each), then the slot is published.  Shared by k_occ_plan and k_occ_plan_lean.
__device__ __forceinline__ void occ_plan_finish(const GraphView &g, const OccBufs &ob, OccSlot *sl, OccShr *shr, const uint32_t *pl0,
                                                WorkCtr &ctr, uint32_t id, uint32_t top, uint32_t mlinks, uint32_t log_cap,
                                                uint32_t snap, uint32_t epoch, bool fail, Visited &vis, int lane, uint32_t snapU = kEmpty)
{
    // ---- the shrinks the connect will trigger (core.rs:560-561): listed here, computed speculatively by
    // k_occ_shrinks, one wave each ----
    uint32_t n_shr = 0;
    for (uint32_t lc1 = top + 1; lc1-- > 0 && !fail;) {
        const uint32_t lc = lc1;
        const uint32_t stride = lc ? g.strideU : g.stride0;
        const uint32_t mmax = lc ? mlinks : 2 * mlinks;     // core.rs:560
        const uint32_t *pl = pl0 + (size_t)lc * g.plan_stride;
        const uint32_t nsel = pl[0];
        const uint32_t e = (uint32_t)lane < nsel ? pl[1 + lane] : 0u;
        uint32_t cnt = (uint32_t)lane < nsel ? row_ptr(g, e, lc)[0] : 0u;
        if (cnt > stride - 1) cnt = stride - 1;
        const bool need = (uint32_t)lane < nsel && cnt + 1 > mmax;   // :561 (the row with this node appended)
        const uint64_t nm = __ballot(need);
        const uint32_t k = n_shr + (uint32_t)__popcll(nm & lanemask_lt(lane));
        // log ranges: running sum of (cnt + 2) over the lanes that need a shrink, in selection order
        uint32_t off = 0;
        uint32_t run = ctr.log_n;
        for (uint64_t mm = nm; mm; mm &= mm - 1) {
            const int j = __ffsll((unsigned long long)mm) - 1;
            const uint32_t cj = (uint32_t)__builtin_amdgcn_readlane((int)cnt, j);
            if (lane == j) off = run;
            run += cj + 2;
        }
        if (need) {
            if (k >= kOccInsShr || cnt + 1 > kAuxWords) fail = true;
            else {
                OccShr *sp = &shr[k];
                sp->lc = lc; sp->e = e; sp->nS = 0; sp->bound = 0; sp->log0 = off; sp->cnt = cnt;
            }
        }
        fail = __ballot(fail) != 0;
        ctr.log_n = run;
        n_shr += (uint32_t)__popcll(nm);
    }
    if (ctr.log_n > log_cap) fail = true;           // (log_cap < kOccMaxReads only in tests: forces the serial path)
    if (ob.ctl->tie_gate && (ctr.n_tie || (sl->node == id && sl->stage == 1u && sl->tie))) fail = true;   // the census flagged this plan: not for the window
    if (vis.glob_dirty) visited_clear(vis, lane);
    __threadfence();
    if (lane == 0) {
        sl->node = id; sl->snap = snap; sl->epoch = epoch;
        sl->snapU = snapU == kEmpty ? snap : snapU; sl->stage = 0u;
        sl->n_reads = ctr.log_n < kOccMaxReads ? ctr.log_n : kOccMaxReads;
        sl->n_shr = n_shr; sl->top = top; sl->fail = fail ? 1u : 0u;
        sl->planned = 1;
        atomicAdd(&g.hdr->ctr_insert[0], (unsigned long long)ctr.n_dist);
        atomicAdd(&g.hdr->ctr_insert[1], (unsigned long long)ctr.n_ids);
        atomicAdd(&g.hdr->ctr_insert[2], (unsigned long long)ctr.n_expand);
        if (ctr.n_tie) { atomicAdd(&g.hdr->ctr_tie[2], (unsigned long long)ctr.n_tie); atomicAdd(&g.hdr->ctr_tie[3], 1ull); }   // tie census
    }
}

// ---------------------------------------------------------------------------------------------------------
// plan: one wave per window node that has no valid plan
// ---------------------------------------------------------------------------------------------------------
template <int MODE, int T, int R>
__global__ __launch_bounds__(64, 1) void k_occ_plan(GraphView g, OccBufs ob, uint32_t first_node, uint32_t count, uint32_t ef,
                                                 uint32_t mlinks, uint32_t lnb, uint32_t lcap, uint32_t *__restrict__ gspill,
                                                 uint32_t gnb, uint32_t *__restrict__ plan, uint32_t shortcut, uint32_t log_cap)
{
    extern __shared__ __align__(16) unsigned char smem[];
    const int lane = threadIdx.x;
    if (!occ_round_window(ob, first_node, count)) return;
    const uint32_t id = first_node + blockIdx.x;
    if (blockIdx.x >= count) return;
    const uint32_t slot = id % ob.W;
    OccSlot *sl = &ob.slots[slot];
    if (sl->planned && sl->node == id) return;

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

    QReg<T> qr;
    load_query<MODE, T>(g.vec + (size_t)id * g.dim, g.dim, qr, m.qlds, lane);
    bool fail = false;
    uint32_t ep = ep0;
    for (uint32_t lc = lmax; lc > l && !fail; --lc) {       // core.rs:511-520
        search_level<MODE, T, 1>(g, m, vis, qr, ep, 1, lc, ctr, lane, fail);
        ep = key_id(m.W[0]);                                // core.rs:514
        wave_sync();
    }
    const uint32_t top = lmax < l ? lmax : l;
    for (uint32_t lc1 = top + 1; lc1-- > 0 && !fail;) {     // core.rs:523
        const uint32_t lc = lc1;
        const uint32_t nW = search_level<MODE, T, R>(g, m, vis, qr, ep, ef, lc, ctr, lane, fail); // :524
        if (fail) break;
        const uint32_t wnearest = key_id(m.W[0]);
        uint32_t nS;
        if (shortcut && select_is_head_of_W(ef, mlinks, nW)) {
            // the selection is the head of W (see select_head_of_W): it reads nothing the search did not read
            nS = select_head_of_W(m, nW, mlinks, lane, &ctr.n_tie);
        } else {
            // select's reads: the rows of all members of W (logged before S exists; the bound is patched in below)
            const uint32_t sel_log0 = ctr.log_n;
            for (uint32_t i = lane; i < nW; i += 64)
                if (sel_log0 + i < kOccMaxReads) reads[sel_log0 + i] = OccRead{key_id(m.W[i]), occ_meta(lc, OCC_SELECT, 0, false), 0u};
            ctr.log_n += nW;
            nS = select_topm<MODE, T>(g, m, vis, qr, m.W, nW, id, mlinks, lc, ctr, lane, fail); // :531
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
    __threadfence();
    wave_sync();

    occ_plan_finish(g, ob, sl, shr, pl0, ctr, id, top, mlinks, log_cap, snap, epoch, fail, vis, lane);
}

// ---------------------------------------------------------------------------------------------------------
// speculative shrinks (core.rs:540-574): one wave per (window node, listed shrink), each as if only that
// node's connect had happened since the snapshot.  The node's own rows hold its selection already (k_occ_plan).
// ---------------------------------------------------------------------------------------------------------
template <int MODE, int T, int R>
__global__ __launch_bounds__(64, 1) void k_occ_shrinks(GraphView g, OccBufs ob, uint32_t first_node, uint32_t count, uint32_t mlinks,
                                                    uint32_t lnb, uint32_t lcap, uint32_t *__restrict__ gspill, uint32_t gnb,
                                                    uint32_t del_id, uint32_t per)
{
    // del_id != kEmpty: the slot lists the re-selections of HNSW.NODE.DEL (k_occ_del_list): the row as it is, del_id ignored
    extern __shared__ __align__(16) unsigned char smem[];
    const int lane = threadIdx.x;
    if (!occ_round_window(ob, first_node, count)) return;
    const uint32_t b = blockIdx.x / per, k = blockIdx.x % per;   // `per` records launched per slot (kOccInsShr / kOccMaxShr)
    if (b >= count) return;
    const uint32_t id = first_node + b;
    const uint32_t slot = id % ob.W;
    OccSlot *sl = &ob.slots[slot];
    if (!sl->planned || sl->node != id || sl->fail || k >= sl->n_shr) return;
    OccShr *sp = &ob.shr[(size_t)slot * kOccMaxShr + k];
    if (sp->nS != 0) return;                                 // computed in an earlier round, plan still valid
    OccRead *reads = ob.reads + (size_t)slot * kOccMaxReads;

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

    const uint32_t lc = sp->lc, e = sp->e, cnt = sp->cnt, log0 = sp->log0;
    const uint32_t mmax = lc ? mlinks : 2 * mlinks;         // core.rs:560
    const uint32_t *erow = row_ptr(g, e, lc);
    bool fail = false;
    // econn: e's row in stored order, this node last (core.rs:544-558); nconn of a delete: the row as it is (:832-844)
    QReg<T> qe;
    load_query<MODE, T>(g.vec + (size_t)e * g.dim, g.dim, qe, m.qlds, lane);
    uint32_t nE = 0;
    const uint32_t tot = del_id == kEmpty ? cnt + 1 : cnt;
    for (uint32_t base = 0; base < tot; base += 64) {
        const uint32_t i = base + lane;
        const uint32_t nf = tot - base < 64 ? tot - base : 64;
        if (i < tot) { const uint32_t x = i < cnt ? erow[1 + i] : id; m.fresh[lane] = x; m.aux[i] = x; }
        wave_sync();
        compute_dists<MODE, T>(g, qe, m, nf, lane);
        wave_sync();
        const bool have = (uint32_t)lane < nf;
        const uint64_t key = have ? pack_key(m.dsc[lane], m.fresh[lane]) : ~0ull;
        nE = merge_sorted<R>(m.W, nE, R * 64, key, have, lane);
    }
    WorkCtr nolog = {};
    nolog.n_dist = tot;                                      // the econn evaluations (core.rs:550)
    nolog.n_ids = tot;
    const uint32_t nS = select_topm<MODE, T>(g, m, vis, qe, m.W, nE, e, mmax, lc, nolog, lane, fail, del_id); // :568 / :853
    if (vis.glob_dirty) visited_clear(vis, lane);
    if (fail || (nS == 0 && del_id == kEmpty)) { if (lane == 0) sl->fail = 1; return; }
    if (nS == 0) {
        // an empty re-selection: the commit computes it itself.  Its share of the read log is hashed with the rest
        // (k_occ_del_list counts the whole range), so it must not keep entries of an earlier operation: they could
        // flag unrelated re-selections stale -- harmless for the graph (a stale one is recomputed), but the
        // speculation's yield and the counters would depend on history.  Row kEmpty matches no journal entry.
        const OccRead none = OccRead{kEmpty, occ_meta(0, OCC_SHRINK_NB, k, true), 0u};
        for (uint32_t i = lane; i < tot + 1; i += 64)
            if (log0 + i < kOccMaxReads) reads[log0 + i] = none;
        return;
    }
    if (lane == 0) { sp->w_dist = nolog.n_dist; sp->w_ids = nolog.n_ids; }
    if (lane == 0 && nolog.n_tie) atomicAdd(&g.hdr->ctr_tie[2], (unsigned long long)nolog.n_tie);   // tie census: a cut between equal distances
    if (lane == 0) sp->tie = nolog.n_tie ? 1u : 0u;
    const uint32_t bound = (uint32_t)(m.S[nS - 1] >> 32);
    for (uint32_t i = lane; i < nS; i += 64) sp->S[i] = key_id(m.S[i]);
    // reads: row e itself, and the rows of econn's members
    if (lane == 0 && log0 < kOccMaxReads) reads[log0] = OccRead{e, occ_meta(lc, OCC_SHRINK_ROW, k, true), 0u};
    for (uint32_t i = lane; i < tot; i += 64)
        if (log0 + 1 + i < kOccMaxReads) reads[log0 + 1 + i] = OccRead{m.aux[i], occ_meta(lc, OCC_SHRINK_NB, k, nS >= mmax), bound};
    // (an insert's shrink always fills S -- its row is over-full; a delete's re-selection may not: then any id that
    // appears in a member's row would be selected, whatever its distance: "not full" = everything matters)
    __threadfence();
    if (lane == 0) { sp->bound = bound; sp->nS = nS; }
}

// ---------------------------------------------------------------------------------------------------------
// validate every planned slot of the window against the journal (parallel; stale ones are re-planned before
// they reach the head).  A slot that passes moves its snapshot forward.
// ---------------------------------------------------------------------------------------------------------
template <int MODE, int T>
__global__ __launch_bounds__(64, 1) void k_occ_validate(GraphView g, OccBufs ob, uint32_t first_node, uint32_t count, uint32_t far = 0)
{
    // far: the slots of the nodes beyond the window whose upper layers were planned ahead (k_occ_plan_lean) are checked too
    extern __shared__ __align__(16) unsigned char smem[];
    const int lane = threadIdx.x;
    if (!occ_round_window(ob, first_node, count)) return;
    const uint32_t id = first_node + blockIdx.x;
    if (blockIdx.x >= count && (blockIdx.x >= count + far || id >= ob.ctl->end_node)) return;
    OccSlot *sl = &ob.slots[id % ob.W];
    const uint32_t nJ = ob.ctl->nJ;
    OccScratch sc = occ_carve(smem);
    WaveMem m = {};
    unsigned char *p = smem + kOccScratchBytes;
    m.fresh = reinterpret_cast<uint32_t *>(p); p += 64 * 4;
    m.dsc = reinterpret_cast<float *>(p); p += 64 * 4;
    m.qlds = reinterpret_cast<float *>(p);
    const OccRead *reads = ob.reads + (size_t)(id % ob.W) * kOccMaxReads;
    const OccShr *shr = ob.shr + (size_t)(id % ob.W) * kOccMaxShr;
    if (!sl->planned || sl->node != id) {
        // a plan whose upper layers were searched a round early (stage 1): what it read so far against the journal
        if (sl->node != id || sl->stage != 1u) return;
        if (sl->epoch != ob.ctl->epoch) { if (lane == 0) sl->stage = 0u; return; }
        if (sl->snapU == nJ) return;
        occ_init_hash(sc, lane);
        occ_build_hash(sc, reads, sl->n_reads_u, shr, 0u, lane);
        occ_check_range<MODE, T>(g, m, sc, ob, reads, shr, id, sl->snapU, nJ, lane, false, kEmpty, nullptr, nJ);   // nothing of layer 0 was read yet
        if (lane == 0) {
            if (sc.flags[0]) { sl->stage = 0u; atomicAdd(&ob.ctl->n_stale, 1ull); }
            else sl->snapU = nJ;
        }
        return;
    }
    if (sl->fail) return;                                   // the commit hands it to the serial path
    if (sl->epoch != ob.ctl->epoch) { if (lane == 0) { sl->planned = 0; sl->stage = 0u; } return; }
    if (sl->snap == nJ && sl->snapU == nJ) return;
    occ_init_hash(sc, lane);
    occ_build_hash(sc, reads, sl->n_reads, shr, sl->n_shr, lane);
    occ_check_range<MODE, T>(g, m, sc, ob, reads, shr, id, sl->snapU, nJ, lane, false, kEmpty, nullptr, sl->snap);
    bool bad = sc.flags[0] != 0;
    for (uint32_t k = 0; k < sl->n_shr; ++k) bad |= sc.flags[2 + k] != 0;
    if (lane == 0) {
        if (bad) { sl->planned = 0; sl->stage = 0u; atomicAdd(&ob.ctl->n_stale, 1ull); }
        else { sl->snap = nJ; sl->snapU = nJ; }
    }
}

// ---------------------------------------------------------------------------------------------------------
// A recomputed select_neighbors (core.rs:544-568 / :832-853) spread over the commit workgroup's wavefronts.
// The committing wave is one wavefront walking the reference's order; what it cannot take from a speculative
// record it recomputes on the spot -- about one select_neighbors per commit at 1 M nodes, ~60 us of one lone wave's
// instruction stream, a third of the commit.  select_neighbors returns the m nearest of a POOL (the candidates and
// their neighbours, core.rs:685-721: a set, whatever the order it is visited in), so the pool splits: wave w of the
// team takes every (1 + HW)-th candidate of econn, runs the SAME select_topm on that share with a visited set of its
// own, and the committing wave merges the shares' results -- the m nearest of the union of the shares' m nearest are
// the m nearest of the pool.  An id reachable through two shares is evaluated twice and shows up as two equal keys
// (same id, same distance): the merge drops keys it already holds.  The result is the one-wave result, key for key.
// Helpers sleep at the workgroup barrier between tasks.  Built only in the translation unit that turns the shared
// code's block-level synchronisations into wave-level ones (hnsw_tu_occteam.hip), like the two-wave plans.
// ---------------------------------------------------------------------------------------------------------
constexpr uint32_t TEAM_SELECT = 1u, TEAM_EXIT = 2u;
constexpr uint32_t kTeamCand = 256;                   // a share of econn: a row holds at most 1 023 ids, a team has >= 4 waves
struct TeamTask {
    uint32_t op, e, lc, mmax, nE, ignored, pad0, pad1;
    uint32_t nS[4], fail[4], n_dist[4], n_ids[4], n_tie[4];
};
struct TeamCfg {                                      // helper LDS layout (the same for every helper), set by the host
    uint32_t hbytes;                                  // bytes per helper
    uint32_t lnb, lcap, tagcfg;                       // its visited table
    uint32_t gnb;                                     // HBM spill buckets per helper
};
__host__ __device__ inline size_t team_fixed_bytes(int T, uint32_t dim)
{
    return (size_t)kTeamCand * 8 + 64 * 4 + 64 * 4 + kSelMax * 8 + (T == 0 ? (((size_t)dim * 4 + 15) & ~(size_t)15) : 0);
}
__device__ __forceinline__ void team_bar()
{
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
}
template <int T>
__device__ __forceinline__ void team_carve(unsigned char *p, uint32_t dim, const TeamCfg &tc, WaveMem &m, Visited &vis)
{
    m.W = reinterpret_cast<uint64_t *>(p); p += (size_t)kTeamCand * 8;
    m.fresh = reinterpret_cast<uint32_t *>(p); p += 64 * 4;
    m.dsc = reinterpret_cast<float *>(p); p += 64 * 4;
    m.S = reinterpret_cast<uint64_t *>(p); p += kSelMax * 8;
    m.aux = nullptr;
    m.qlds = reinterpret_cast<float *>(p);
    if (T == 0) p += ((size_t)dim * 4 + 15) & ~(size_t)15;
    vis.lds = reinterpret_cast<uint32_t *>(p);
    vis.lnb = tc.lnb;
    vis.lcap = tc.lcap;
    vis.tag_bb = tc.tagcfg & 0xFFu;
    vis.idbits = tc.tagcfg >> 8;
    vis.glob_dirty = false;
    vis.spilled = false;
    vis.count = 0;
    vis.bounded = false;
    vis.lossy = false;
}
// this wave's share of econn (sorted nearest first in W0[0..nE)): every nw-th candidate, starting at `wave`
__device__ __forceinline__ uint32_t team_share(uint64_t *dst, const uint64_t *W0, uint32_t nE, uint32_t wave, uint32_t nw, int lane)
{
    const uint32_t n = nE > wave ? (nE - wave + nw - 1) / nw : 0u;
    for (uint32_t i = lane; i < n; i += 64) dst[i] = W0[i * nw + wave];
    wave_sync();
    return n;
}

// a helper wavefront of the commit workgroup: serves TEAM_SELECT tasks until TEAM_EXIT
template <int MODE, int T, class GV>
__device__ __forceinline__ void team_helper(const GV &g, volatile TeamTask *task, const uint64_t *W0, unsigned char *hmem,
                                            const TeamCfg &tc, uint32_t *hspill, uint32_t wave, uint32_t nw, int lane)
{
    WaveMem m;
    Visited vis;
    team_carve<T>(hmem, g.dim, tc, m, vis);
    vis.glob = hspill + (size_t)(wave - 1) * tc.gnb * 8;
    vis.gnb = tc.gnb;
    for (;;) {
        team_bar();                                         // a task is posted
        const uint32_t op = __builtin_amdgcn_readfirstlane(task->op);
        if (op == TEAM_EXIT) break;
        const uint32_t e = __builtin_amdgcn_readfirstlane(task->e), lc = __builtin_amdgcn_readfirstlane(task->lc);
        const uint32_t mmax = __builtin_amdgcn_readfirstlane(task->mmax), nE = __builtin_amdgcn_readfirstlane(task->nE);
        const uint32_t ignored = __builtin_amdgcn_readfirstlane(task->ignored);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");  // the committing wave has rewritten rows since the last task
        const uint32_t n = team_share(m.W, W0, nE, wave, nw, lane);
        uint32_t nS = 0;
        bool fail = false;
        WorkCtr c = {};
        if (n) {
            QReg<T> qe;
            load_query<MODE, T>(g.vec + (size_t)e * g.dim, g.dim, qe, m.qlds, lane);
            nS = select_topm<MODE, T>(g, m, vis, qe, m.W, n, e, mmax, lc, c, lane, fail, ignored, false);
        } else c.tie_emin = 0xFFFFFFFFu;
        if (lane == 0) {
            task->nS[wave] = nS;
            task->fail[wave] = fail ? 1u : 0u;
            task->n_dist[wave] = c.n_dist;
            task->n_ids[wave] = c.n_ids;
            task->n_tie[wave] = c.tie_emin;                  // tie census: the nearest distance outside this share's selection
        }
        team_bar();                                         // the results are there
    }
    if (vis.glob_dirty) visited_clear(vis, lane);
}

// the committing wave's side: econn is sorted in m.W[0..nE); result in m.S[0..nS)
template <int MODE, int T, class GV>
__device__ __forceinline__ uint32_t team_select(const GV &g, const WaveMem &m, Visited &vis, const QReg<T> &qe, uint32_t nE,
                                                uint32_t e, uint32_t mmax, uint32_t lc, WorkCtr &ctr, int lane, bool &fail,
                                                uint32_t ignored, volatile TeamTask *task, uint64_t *W0sub, unsigned char *hmem0,
                                                const TeamCfg &tc, uint32_t nw)
{
    if (lane == 0) {
        task->op = TEAM_SELECT; task->e = e; task->lc = lc; task->mmax = mmax; task->nE = nE; task->ignored = ignored;
    }
    fence_own_writes();                                     // the rows this wave has rewritten, before the helpers read them
    team_bar();
    const uint32_t n0 = team_share(W0sub, m.W, nE, 0u, nw, lane);
    if (!n0) ctr.tie_emin = 0xFFFFFFFFu;
    uint32_t nS = n0 ? select_topm<MODE, T>(g, m, vis, qe, W0sub, n0, e, mmax, lc, ctr, lane, fail, ignored, false) : 0u;
    team_bar();
    for (uint32_t w = 1; w < nw; ++w) {
        const uint32_t nSw = __builtin_amdgcn_readfirstlane(task->nS[w]);
        fail |= __builtin_amdgcn_readfirstlane(task->fail[w]) != 0u;
        ctr.n_dist += __builtin_amdgcn_readfirstlane(task->n_dist[w]);
        ctr.n_ids += __builtin_amdgcn_readfirstlane(task->n_ids[w]);
        ctr.tie_emin = min(ctr.tie_emin, (uint32_t)__builtin_amdgcn_readfirstlane(task->n_tie[w]));
        const uint64_t *Sw = reinterpret_cast<const uint64_t *>(hmem0 + (size_t)(w - 1) * tc.hbytes + (size_t)kTeamCand * 8 + 64 * 4 + 64 * 4);
        for (uint32_t base = 0; base < nSw; base += 64) {
            const bool have = base + (uint32_t)lane < nSw;
            const uint64_t key = have ? Sw[base + lane] : ~0ull;
            // already held (the same id reached through another share)?  m.S is sorted: lower bound, then compare
            uint32_t lo = 0, hi = nS;
            while (__ballot(have && lo < hi)) {
                const uint32_t mid = (lo + hi) >> 1;
                const uint64_t mv = (have && lo < hi) ? m.S[mid] : 0ull;
                if (have && lo < hi) { if (mv < key) lo = mid + 1; else hi = mid; }
            }
            const bool dup = have && lo < nS && m.S[lo] == key;
            const uint64_t worst = nS == mmax ? m.S[mmax - 1] : ~0ull;
            wave_sync();
            nS = merge_S(m.S, nS, mmax, key, have && !dup, worst, lane, &ctr.n_tie);
        }
    }
    // tie census: the cut of the whole pool (core.rs:733 / :741-754) against the nearest key left outside by any share
    if (nS == mmax && nS && ctr.tie_emin == (uint32_t)(m.S[mmax - 1] >> 32)) ctr.n_tie += 1u;
    if (any_adjacent_equal(m.S, nS, lane)) ctr.n_tie += 1u;                      // equal distances inside the selection
    return nS;
}

// ---------------------------------------------------------------------------------------------------------
// commit: one wave, strictly in id order
// ---------------------------------------------------------------------------------------------------------
// HW > 0: the workgroup has HW helper wavefronts that share every recomputed select_neighbors with the committing
// wave (team_select above); LDS after the committing wave's own carve-up: [TeamTask][its share of econn][helpers].
template <int MODE, int T, int R, int HW = 0>
__global__ __launch_bounds__(64 * (1 + HW), 1) void k_occ_commit(GraphView g, OccBufs ob, uint32_t end_node, uint32_t mlinks, uint32_t lnb,
                                                   uint32_t lcap, uint32_t *__restrict__ gspill, uint32_t gnb,
                                                   const uint32_t *__restrict__ plan, uint32_t slack,
                                                   uint32_t *__restrict__ touched, uint32_t touched_cap,
                                                   uint32_t own_lds = 0, TeamCfg tc = TeamCfg{}, uint32_t *__restrict__ hspill = nullptr,
                                                   uint32_t chained = 0)
{
    extern __shared__ __align__(16) unsigned char smem[];
    const int lane = threadIdx.x & 63;
    if (chained && (ob.ctl->stop >= OCC_STOP_RESTRIDE || ob.ctl->head >= end_node)) return;   // a round enqueued ahead of the host (occ_round_window)
    OccScratch sc = occ_carve(smem);
    WaveMem m;
    Visited vis;
    carve<R, T, true>(smem + kOccScratchBytes, g.dim, lnb, lcap, m, vis, g.tagcfg, g.selcap);
    volatile TeamTask *task = reinterpret_cast<volatile TeamTask *>(smem + kOccScratchBytes + own_lds);
    uint64_t *W0sub = reinterpret_cast<uint64_t *>(smem + kOccScratchBytes + own_lds + sizeof(TeamTask));
    unsigned char *hmem0 = smem + kOccScratchBytes + own_lds + sizeof(TeamTask) + (size_t)kTeamCand * 8;
    if constexpr (HW > 0) {
        const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)threadIdx.x) >> 6;
        if (wave > 0) {
            team_helper<MODE, T>(g, task, m.W, hmem0 + (size_t)(wave - 1) * tc.hbytes, tc, hspill, wave, 1u + HW, lane);
            return;
        }
    }
    vis.glob = gspill;
    vis.gnb = gnb;
    vis.glob_dirty = false;
    vis.spilled = false;
    vis.count = 0;
    vis.bounded = false;
    vis.lossy = false;

    OccJournal jr;
    jr.ring = ob.ring;
    jr.n = ob.ctl->nJ;
    jr.own = nullptr;
    jr.own_base = 0;
    jr.cap = 0;
    occ_init_hash(sc, lane);
    uint32_t head = ob.ctl->head;
    uint32_t stop = OCC_STOP_NONE;
    unsigned long long n_commit = 0, n_spec = 0, n_fallback = 0, n_norec = 0;
    unsigned long long n_cls[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long w_dist = 0, w_ids = 0, w_skipped = 0;   // the shrink loop's work: evaluations made (speculative results
                                                               // that were used + recomputations), econn evaluations skipped (:561)
    uint32_t nt = 0;
    unsigned long long prof[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const unsigned long long t_begin = __builtin_readcyclecounter();
    unsigned long long t_ = t_begin;
#define OCC_T(i) do { const unsigned long long n_ = __builtin_readcyclecounter(); prof[i] += n_ - t_; t_ = n_; } while (0)

    while (head < end_node) {
        const uint32_t id = head;
        const uint32_t slot = id % ob.W;
        OccSlot *sl = &ob.slots[slot];
        if (!sl->planned || sl->node != id) { stop = OCC_STOP_REPLAN; break; }
        if (sl->fail) { stop = OCC_STOP_SERIAL; break; }
        if (sl->epoch != ob.ctl->epoch) { if (lane == 0) sl->planned = 0; stop = OCC_STOP_REPLAN; break; }
        // one insert raises any row by at most m + 1 per layer (connect + third-party appends of its shrinks)
        if (g.hdr->max_deg0 + slack > g.stride0 - 1 || g.hdr->max_degU + slack > g.strideU - 1) { stop = OCC_STOP_RESTRIDE; break; }
        const OccRead *reads = ob.reads + (size_t)slot * kOccMaxReads;
        const OccShr *shr = ob.shr + (size_t)slot * kOccMaxShr;
        const uint32_t n_shr = sl->n_shr;
        OCC_T(6);
        occ_build_hash(sc, reads, sl->n_reads, shr, n_shr, lane);
        OCC_T(0);
        uint32_t checked = sl->snapU;
        occ_check_range<MODE, T>(g, m, sc, ob, reads, shr, id, checked, jr.n, lane, false, kEmpty, nullptr, sl->snap);
        checked = jr.n;
        OCC_T(1);
        if (sc.flags[0]) {                                   // the link plan is stale: re-plan (end of the round)
            wave_sync();
            occ_clear_hash(sc, sl->n_reads, lane);
            if (lane == 0) { sl->planned = 0; sl->stage = 0u; }
            stop = OCC_STOP_REPLAN;
            break;
        }
        jr.own = sc.own;                                     // this node's deltas are mirrored in LDS from here on
        jr.own_base = jr.n;
        uint64_t done_rec = 0;                               // speculative records already consumed by this commit
        const uint32_t lmax = g.hdr->max_layer;
        const uint32_t l = g.levels[id];
        const uint32_t top = sl->top;
        const uint32_t *pl0 = plan + (size_t)slot * kMaxLayers * g.plan_stride;
        bool fail = false;

        for (uint32_t lc1 = top + 1; lc1-- > 0 && !fail;) { // core.rs:523
            const uint32_t lc = lc1;
            const uint32_t stride = lc ? g.strideU : g.stride0;
            const uint32_t mmax = lc ? mlinks : 2 * mlinks; // core.rs:560
            uint32_t *maxdeg = lc ? &g.hdr->max_degU : &g.hdr->max_deg0;
            const uint32_t *pl = pl0 + (size_t)lc * g.plan_stride;
            const uint32_t nsel = pl[0];
            const uint32_t myselid = (uint32_t)lane < nsel ? pl[1 + lane] : kEmpty;
            // connect_neighbors (core.rs:759-774), nearest first; the node's own row was written by its plan
            uint32_t *qrow = row_ptr(g, id, lc);
            if (lane == 0) qrow[0] = nsel;
            if ((uint32_t)lane < nsel) {
                qrow[1 + lane] = myselid;
                uint32_t *nrow = row_ptr(g, myselid, lc);
                const uint32_t c = nrow[0];
                if (c + 1 > stride - 1) atomicOr(&g.hdr->status, ST_ROW_OVERFLOW);
                else { nrow[1 + c] = id; nrow[0] = c + 1; atomicMax(maxdeg, c + 1); }
            }
            if (lane == 0) atomicMax(maxdeg, nsel);
            touch_push(touched, touched_cap, nt, myselid, touched != nullptr && (uint32_t)lane < nsel, lane);   // :535-537
            journal_push(&jr, (uint32_t)lane < nsel, myselid, lc, id, true, lane);
            fence_own_writes();
            wave_sync();
            OCC_T(2);

            for (uint32_t si = 0; si < nsel && !fail; ++si) {   // shrink loop (core.rs:540-574), e nearest first
                const uint32_t e = pl[1 + si];
                uint32_t *erow = row_ptr(g, e, lc);
                uint32_t cnt = erow[0];
                if (cnt > stride - 1) cnt = stride - 1;
                if (cnt <= mmax) { w_skipped += cnt; continue; }   // :561
                // the speculative result, if nothing relevant happened since it was planned
                int k = -1;
                {
                    const bool mine = (uint32_t)lane < n_shr && shr[lane].e == e && shr[lane].lc == lc && shr[lane].nS != 0;
                    const uint64_t mb = __ballot(mine);
                    if (mb) k = 63 - __builtin_clzll((unsigned long long)mb);
                }
                if (k >= 0 && checked != jr.n) {
                    occ_check_range<MODE, T>(g, m, sc, ob, reads, shr, id, checked, jr.n, lane, true, jr.own_base, nullptr, kEmpty, done_rec);
                    checked = jr.n;
                }
                if (k >= 0) done_rec |= 1ull << k;                       // (its verdict is taken below; later deltas need not look at it)
                OCC_T(3);
                for (uint32_t i = lane; i < cnt; i += 64) m.aux[i] = erow[1 + i];
                wave_sync();
                uint32_t nS;
                if (k >= 0 && !sc.flags[2 + k]) {
                    const uint32_t sv = shr[k].S[lane], sv2 = shr[k].S[64 + lane];   // loaded alongside nS, not after it
                    nS = shr[k].nS;
                    if ((uint32_t)lane < nS) m.S[lane] = (uint64_t)sv << 1;
                    if (64u + (uint32_t)lane < nS) m.S[64 + lane] = (uint64_t)sv2 << 1;
                    wave_sync();
                    n_spec += 1;
                    w_dist += shr[k].w_dist;
                    w_ids += shr[k].w_ids;
                } else {
                    // recompute on the spot (core.rs:544-568)
                    QReg<T> qe;
                    load_query<MODE, T>(g.vec + (size_t)e * g.dim, g.dim, qe, m.qlds, lane);
                    uint32_t nE = 0;
                    for (uint32_t base = 0; base < cnt; base += 64) {
                        const uint32_t i = base + lane;
                        const uint32_t nf = cnt - base < 64 ? cnt - base : 64;
                        if (i < cnt) m.fresh[lane] = m.aux[i];
                        wave_sync();
                        compute_dists<MODE, T>(g, qe, m, nf, lane);
                        wave_sync();
                        const bool have = (uint32_t)lane < nf;
                        const uint64_t key = have ? pack_key(m.dsc[lane], m.fresh[lane]) : ~0ull;
                        nE = merge_sorted<R>(m.W, nE, R * 64, key, have, lane);
                    }
                    WorkCtr nolog = {};
                    if constexpr (HW > 0)
                        nS = team_select<MODE, T>(g, m, vis, qe, nE, e, mmax, lc, nolog, lane, fail, kEmpty, task, W0sub, hmem0, tc, 1u + HW);
                    else
                        nS = select_topm<MODE, T>(g, m, vis, qe, m.W, nE, e, mmax, lc, nolog, lane, fail);
                    if (fail) break;
                    w_dist += cnt + nolog.n_dist;
                    w_ids += cnt + nolog.n_ids;
                    if (lane == 0 && nolog.n_tie) atomicAdd(&g.hdr->ctr_tie[2], (unsigned long long)nolog.n_tie);   // tie census
                    n_fallback += 1;
                    if (k < 0) n_norec += 1;
                    {
                        const uint32_t why = k >= 0 ? sc.flags[2 + k] : 0u;
                        const int c = k < 0 ? 0 : (why & OCC_WHY_ROW) ? 1 : (why & OCC_WHY_OPEN) ? 2 : (why & OCC_WHY_REMOVE) ? 3 : 4;
                        n_cls[c] += 1;
                    }
                    OCC_T(5);
                }
                update_connections(g, m, e, erow, cnt, nS, lc, stride, maxdeg, kEmpty, touched, touched_cap, nt, lane, &jr);
                OCC_T(4);
            }
        }
        if (fail) { if (lane == 0) atomicOr(&g.hdr->status, ST_VISITED_OVERFLOW); stop = OCC_STOP_SERIAL; break; }
        if (lane == 0) {
            if (l > lmax) {                                 // core.rs:587-593
                g.hdr->max_layer = l;
                g.hdr->enterpoint = (int32_t)id;
                ob.ctl->epoch += 1;
            }
            g.hdr->node_count = id + 1;
            sl->planned = 0;
        }
        fence_own_writes();
        wave_sync();
        occ_clear_hash(sc, sl->n_reads, lane);
        n_commit += 1;
        head += 1;
    }
    if constexpr (HW > 0) {
        if (lane == 0) task->op = TEAM_EXIT;
        team_bar();
    }
    if (vis.glob_dirty) visited_clear(vis, lane);
    if (lane == 0) {
        ob.ctl->head = head;
        ob.ctl->nJ = jr.n;
        ob.ctl->stop = stop;
        ob.ctl->rounds += 1;
        ob.ctl->n_commit += n_commit;
        ob.ctl->n_spec += n_spec;
        ob.ctl->n_fallback += n_fallback;
        ob.ctl->n_norec += n_norec;
        for (int i = 0; i < 8; ++i) ob.ctl->n_cls[i] += n_cls[i];
        if (touched) g.hdr->n_touched = nt;
        atomicAdd(&g.hdr->ctr_insert[0], w_dist);
        atomicAdd(&g.hdr->ctr_insert[1], w_ids);
        atomicAdd(&g.hdr->ctr_insert[3], w_skipped);
        OCC_T(6);
        prof[7] = __builtin_readcyclecounter() - t_begin;
        for (int i = 0; i < 8; ++i) ob.ctl->prof[i] += prof[i];
    }
#undef OCC_T
}

// ---------------------------------------------------------------------------------------------------------
// HNSW.NODE.DEL through the same machinery (core.rs:414-475, 824-863): every neighbour n of the deleted node
// re-selects its links from its two-hop neighbourhood with the node ignored -- ~33 select_neighbors one after the
// other on one wavefront in k_delete_exact.  Here they are LISTED (k_occ_del_list), computed speculatively against
// the graph as it stands, one wave each (k_occ_shrinks with del_id), and applied in the reference's order by one
// wave that validates each result against the row changes the delete itself has journalled so far and recomputes
// the stale ones (k_occ_del_commit).  Same rules as the insert's speculative shrinks: a change of n's own row, or a
// change of a member's row by an id no farther than the last selected, makes the result stale.  A node with more
// than kOccMaxShr neighbours over all its layers (or a read log that does not fit) is left to k_delete_exact.
// ---------------------------------------------------------------------------------------------------------
template <int MODE, int T>
__global__ __launch_bounds__(64, 1) void k_occ_del_list(GraphView g, OccBufs ob, uint32_t id)
{
    const int lane = threadIdx.x;
    const uint32_t slot = id % ob.W;
    OccSlot *sl = &ob.slots[slot];
    OccShr *shr = ob.shr + (size_t)slot * kOccMaxShr;
    const uint32_t l = g.levels[id];
    uint32_t n_shr = 0, log_n = 0;
    bool fail = false;
    for (uint32_t lc = 0; lc <= l && !fail; ++lc) {              // core.rs:434
        const uint32_t stride = lc ? g.strideU : g.stride0;
        const uint32_t *drow = row_ptr(g, id, lc);
        uint32_t dcnt = drow[0];
        if (dcnt > stride - 1) dcnt = stride - 1;
        if (n_shr + dcnt > kOccMaxShr) { fail = true; break; }   // (kOccMaxShr <= 64: one pass of the wave per layer)
        const bool on = (uint32_t)lane < dcnt;
        const uint32_t n = on ? drow[1 + lane] : 0u;
        uint32_t cnt = on ? row_ptr(g, n, lc)[0] : 0u;
        if (cnt > stride - 1) cnt = stride - 1;
        if (__ballot(on && cnt > kAuxWords)) { fail = true; break; }
        // log ranges: running sum of (1 + cnt) in stored order (core.rs:829)
        uint32_t incl = on ? cnt + 1 : 0u;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t o = (uint32_t)__shfl_up((int)incl, d, 64);
            if (lane >= d) incl += o;
        }
        if (on) {
            OccShr *sp = &shr[n_shr + lane];
            sp->lc = lc; sp->e = n; sp->nS = 0; sp->bound = 0; sp->log0 = log_n + incl - (cnt + 1); sp->cnt = cnt;
        }
        log_n += (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
        n_shr += dcnt;
    }
    if (log_n > kOccMaxReads) fail = true;
    __threadfence();
    if (lane == 0) {
        sl->node = id; sl->snap = ob.ctl->nJ; sl->epoch = ob.ctl->epoch;
        sl->n_reads = log_n < kOccMaxReads ? log_n : kOccMaxReads;
        sl->n_shr = fail ? 0u : n_shr; sl->top = l; sl->fail = fail ? 1u : 0u;
        sl->planned = 1;
    }
}

template <int MODE, int T, int R, int HW = 0>
__global__ __launch_bounds__(64 * (1 + HW), 1) void k_occ_del_commit(GraphView g, OccBufs ob, uint32_t id, uint32_t mlinks, uint32_t lnb,
                                                       uint32_t lcap, uint32_t *__restrict__ gspill, uint32_t gnb,
                                                       uint32_t *__restrict__ touched, uint32_t touched_cap,
                                                       uint32_t own_lds = 0, TeamCfg tc = TeamCfg{}, uint32_t *__restrict__ hspill = nullptr)
{
    extern __shared__ __align__(16) unsigned char smem[];
    const int lane = threadIdx.x & 63;
    OccScratch sc = occ_carve(smem);
    WaveMem m;
    Visited vis;
    carve<R, T, true>(smem + kOccScratchBytes, g.dim, lnb, lcap, m, vis, g.tagcfg, g.selcap);
    volatile TeamTask *task = reinterpret_cast<volatile TeamTask *>(smem + kOccScratchBytes + own_lds);
    uint64_t *W0sub = reinterpret_cast<uint64_t *>(smem + kOccScratchBytes + own_lds + sizeof(TeamTask));
    unsigned char *hmem0 = smem + kOccScratchBytes + own_lds + sizeof(TeamTask) + (size_t)kTeamCand * 8;
    if constexpr (HW > 0) {
        const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)threadIdx.x) >> 6;
        if (wave > 0) {
            team_helper<MODE, T>(g, task, m.W, hmem0 + (size_t)(wave - 1) * tc.hbytes, tc, hspill, wave, 1u + HW, lane);
            return;
        }
    }
    vis.glob = gspill;
    vis.gnb = gnb;
    vis.glob_dirty = false;
    vis.spilled = false;
    vis.count = 0;
    vis.bounded = false;
    vis.lossy = false;

    const uint32_t slot = id % ob.W;
    OccSlot *sl = &ob.slots[slot];
    if (!sl->planned || sl->node != id || sl->fail) {           // nothing is touched: k_delete_exact takes it
        if (lane == 0) ob.ctl->stop = OCC_STOP_SERIAL;
        if constexpr (HW > 0) {
            if (lane == 0) task->op = TEAM_EXIT;
            team_bar();
        }
        return;
    }
    const OccRead *reads = ob.reads + (size_t)slot * kOccMaxReads;
    const OccShr *shr = ob.shr + (size_t)slot * kOccMaxShr;
    const uint32_t n_shr = sl->n_shr;
    OccJournal jr;
    jr.ring = ob.ring;
    jr.n = ob.ctl->nJ;
    jr.cap = 0;
    occ_init_hash(sc, lane);
    occ_build_hash(sc, reads, sl->n_reads, shr, n_shr, lane);
    jr.own = sc.own;                                            // the delete's deltas are mirrored in LDS
    jr.own_base = jr.n;
    uint32_t checked = jr.n;
    unsigned long long n_spec = 0, n_fallback = 0, w_dist = 0, w_ids = 0;
#ifdef HNSW_OCC_DEBUG
    unsigned long long dbg_hash = 0, dbg_rec = 0, dbg_stale = 0, dbg_why = 0, dbg_miss = 0, dbg_over = 0;
#endif
    uint32_t nt = 0;
    bool fail = false;
    const uint32_t l = g.levels[id];

    for (uint32_t lc = 0; lc <= l && !fail; ++lc) {              // core.rs:434
        const uint32_t stride = lc ? g.strideU : g.stride0;
        const uint32_t mmax = lc ? mlinks : 2 * mlinks;          // core.rs:846
        uint32_t *maxdeg = lc ? &g.hdr->max_degU : &g.hdr->max_deg0;
        uint32_t *drow = row_ptr(g, id, lc);                     // not modified while we walk it
        uint32_t dcnt = drow[0];
        if (dcnt > stride - 1) dcnt = stride - 1;
        for (uint32_t kk = 0; kk < dcnt && !fail; ++kk) {        // core.rs:829 stored order
            const uint32_t n = drow[1 + kk];
            uint32_t *erow = row_ptr(g, n, lc);
            uint32_t cnt = erow[0];
            if (cnt > stride - 1) cnt = stride - 1;
            int k = -1;
            {
                const bool mine = (uint32_t)lane < n_shr && shr[lane].e == n && shr[lane].lc == lc && shr[lane].nS != 0;
                const uint64_t mb = __ballot(mine);
                if (mb) k = 63 - __builtin_clzll((unsigned long long)mb);
            }
            if (k >= 0 && checked != jr.n) {
                occ_check_range<MODE, T>(g, m, sc, ob, reads, shr, id, checked, jr.n, lane, true, jr.own_base);
                checked = jr.n;
            }
            for (uint32_t i = lane; i < cnt; i += 64) m.aux[i] = erow[1 + i];
            wave_sync();
            uint32_t nS;
#ifdef HNSW_OCC_DEBUG
            // ground truth beside the validation: every record is also recomputed; a record whose list differs from the
            // recomputed one without having been flagged is a validation miss
            unsigned long long dbg_spec_hash = 0;
            if (k >= 0) {
                const uint32_t ns0 = shr[k].nS;
                for (uint32_t i = lane; i < ns0; i += 64) dbg_spec_hash += (unsigned long long)shr[k].S[i] * (unsigned long long)(2 * i + 1);
                for (int o = 32; o; o >>= 1) dbg_spec_hash += __shfl_xor(dbg_spec_hash, o, 64);
                dbg_spec_hash += ns0;
            }
            if (false) {
#else
            if (k >= 0 && !sc.flags[2 + k]) {
#endif
                const uint32_t sv = shr[k].S[lane], sv2 = shr[k].S[64 + lane];
                nS = shr[k].nS;
                if ((uint32_t)lane < nS) m.S[lane] = (uint64_t)sv << 1;
                if (64u + (uint32_t)lane < nS) m.S[64 + lane] = (uint64_t)sv2 << 1;
                wave_sync();
                n_spec += 1;
                w_dist += shr[k].w_dist;
                w_ids += shr[k].w_ids;
            } else {
                // recompute on the spot (core.rs:832-853)
                QReg<T> qe;
                load_query<MODE, T>(g.vec + (size_t)n * g.dim, g.dim, qe, m.qlds, lane);
                uint32_t nE = 0;
                for (uint32_t base = 0; base < cnt; base += 64) {
                    const uint32_t i = base + lane;
                    const uint32_t nf = cnt - base < 64 ? cnt - base : 64;
                    if (i < cnt) m.fresh[lane] = m.aux[i];
                    wave_sync();
                    compute_dists<MODE, T>(g, qe, m, nf, lane);
                    wave_sync();
                    const bool have = (uint32_t)lane < nf;
                    const uint64_t key = have ? pack_key(m.dsc[lane], m.fresh[lane]) : ~0ull;
                    nE = merge_sorted<R>(m.W, nE, R * 64, key, have, lane);
                }
                WorkCtr nolog = {};
                if constexpr (HW > 0)
                {
                    nS = team_select<MODE, T>(g, m, vis, qe, nE, n, mmax, lc, nolog, lane, fail, id, task, W0sub, hmem0, tc, 1u + HW);
                }
                else
                    nS = select_topm<MODE, T>(g, m, vis, qe, m.W, nE, n, mmax, lc, nolog, lane, fail, id);
                if (fail) break;
                w_dist += cnt + nolog.n_dist;
                w_ids += cnt + nolog.n_ids;
                if (lane == 0 && nolog.n_tie) atomicAdd(&g.hdr->ctr_tie[2], (unsigned long long)nolog.n_tie);   // tie census
                n_fallback += 1;
            }
#ifdef HNSW_OCC_DEBUG
            {   // what was decided for this neighbour, into the control block (hnsw_debug_occ_ctl)
                wave_sync();
                unsigned long long hs = 0;
                for (uint32_t i = lane; i < nS; i += 64) hs += (unsigned long long)key_id(m.S[i]) * (unsigned long long)(2 * i + 1);
                for (int o = 32; o; o >>= 1) hs += __shfl_xor(hs, o, 64);
                dbg_hash = dbg_hash * 1000003ull + hs + nS;
                if (k >= 0) {
                    const bool same = dbg_spec_hash == hs + nS;
                    if (!same && !sc.flags[2 + k]) dbg_miss += 1;
                    if (same && sc.flags[2 + k]) dbg_over += 1;
                }
                if (k >= 0) { dbg_rec |= 1ull << k; if (sc.flags[2 + k]) dbg_stale |= 1ull << k; if (k < 16) dbg_why |= (unsigned long long)(sc.flags[2 + k] & 15u) << (4 * k); }
            }
#endif
            // update_node_connections(n, new, old, ignored = node) (core.rs:856)
            update_connections(g, m, n, erow, cnt, nS, lc, stride, maxdeg, id, touched, touched_cap, nt, lane, &jr);
        }
        if (lane == 0) drow[0] = 0;                              // the node is gone (core.rs:419)
        __threadfence();
        wave_sync();
    }
    if (fail && lane == 0) atomicOr(&g.hdr->status, ST_VISITED_OVERFLOW);
    if constexpr (HW > 0) {
        if (lane == 0) task->op = TEAM_EXIT;
        team_bar();
    }
    if (vis.glob_dirty) visited_clear(vis, lane);
    if (lane == 0) {
        g.hdr->n_touched = nt;
        atomicAdd(&g.hdr->ctr_insert[0], w_dist);
        atomicAdd(&g.hdr->ctr_insert[1], w_ids);
        ob.ctl->head = id + 1;                                   // done (the host's completion test)
        ob.ctl->nJ = jr.n;
        ob.ctl->stop = OCC_STOP_NONE;
        ob.ctl->n_spec += n_spec;
        ob.ctl->n_fallback += n_fallback;
#ifdef HNSW_OCC_DEBUG
        ob.ctl->prof[0] = dbg_hash; ob.ctl->prof[1] = dbg_rec; ob.ctl->prof[2] = dbg_stale; ob.ctl->prof[3] = n_spec; ob.ctl->prof[4] = n_fallback;
        ob.ctl->n_cls[0] = dbg_why; ob.ctl->n_cls[1] = dbg_miss; ob.ctl->n_cls[2] = dbg_over;
#endif
        sl->planned = 0;
    }
}

// ---- sizes and records of the parallel group commit (kernels: hnsw_occ_par.hpp) ----------------------------------
constexpr uint32_t kParTab = 512;         // overlay table slots = scratch rows of a workgroup
constexpr uint32_t kParMaxRows = 400;     // rows a dry run may rewrite
constexpr uint32_t kParMaxDelta = 1024;   // deltas it may journal
constexpr uint32_t kParSubRows = 63;      // the read log's sub-operation that stands for "a row this dry run rewrote"
constexpr uint32_t kParMaxSub = 63;       // speculative records use 0 .. n_shr-1 (<= kOccInsShr), recomputations the ones up to 62
enum { PAR_NONE = 0, PAR_READY = 1, PAR_REPLAN = 2, PAR_SERIAL = 3, PAR_RESTRIDE = 4 };
constexpr size_t kParLdsBytes = (size_t)kParTab * 4 + 64;

struct OccPar {                           // what a workgroup publishes about its node, per iteration
    uint32_t state, conflict, promotes, n_delta;
    uint32_t n_spec, n_fallback, n_norec, maxdeg0, maxdegU, why, pad0, pad1;
    unsigned long long w_dist, w_ids, w_skipped;
};
struct ParBufs {
    OccPar *par;                          // [workgroups]
    OccDelta *delta;                      // [workgroups][kParMaxDelta]
    uint32_t *rows;                       // [workgroups][kParTab + 1][ovstride]
    uint32_t ovstride;                    // words per scratch row = max(stride0, strideU)
};


} // namespace hnsw
