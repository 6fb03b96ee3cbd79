// hnsw_host.hpp -- what the translation units of libhnsw_mi355x.so share on the host side: the handle, the
// helpers the kernel launchers need, and the launchers' declarations.  The library is built from one
// translation unit per kernel family and metric variant (redis_hnsw_amd/build.py compiles them in parallel):
//   hnsw_engine.hip      C ABI, capacity / staging / pipeline bookkeeping, the small utility kernels
//   hnsw_tu_search.hip   k_search<MODE,T,R>                    (general search kernel)         x 4 variants
//   hnsw_tu_lean.hip     k_search_lean<VEC,R,BB,DB,WIDE>       (dim-128 specialisation)        x 4 variants
//   hnsw_tu_insert.hip   k_insert_plan / k_insert_commit_exact / k_delete_exact / k_shrink_batch x 4 variants
//   hnsw_tu_occ.hip      k_occ_validate / plan / shrinks / commit                              x 4 variants
//   hnsw_tu_planlean.hip k_insert_plan_lean / k_occ_plan_lean  (dim-128 plans, specialised search) x 2 row widths
// A variant is one (metric order, query placement) pair, HNSW_VARIANT = 0..3:
//   0 MODE_SCALAR,T=0   1 MODE_AVX,T=4 (dim 128)   2 MODE_AVX,T=24 (dim 768)   3 MODE_AVX,T=0 (any dim % 32 == 0)
// Each launcher template is defined in its family's file and explicitly instantiated there for the variant
// being compiled; the engine only sees the declarations below.
#pragma once
#include "../../include/hnsw_mi355x.h"
#include "hnsw_occ.hpp"   // -> hnsw_insert.hpp -> hnsw_device.hpp (types and constants; kernels are templates)

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <mutex>
#include <string>
#include <vector>

struct hnsw_index {
    uint32_t dim = 0, m = 0, m_max = 0, m_max0 = 0, efc = 0;
    double level_mult = 0;
    int device = 0;
    int mode = hnsw::MODE_AVX, T = 0;
    uint32_t cap = 0, n = 0, max_layer = 0;
    int64_t enterpoint = -1;
    uint32_t stride0 = 0, strideU = 0, upper_cap = 0, upper_used = 0;
    uint32_t max_deg0 = 0, max_degU = 0;
    float *d_vec = nullptr;
    uint32_t *d_adj0 = nullptr, *d_adjU = nullptr, *d_upper_base = nullptr, *d_levels = nullptr;
    hnsw::DevHeader *d_hdr = nullptr;
    std::vector<uint32_t> h_levels, h_upper_base;
    std::vector<uint8_t> h_dead;   // tombstones (HNSW.NODE.DEL); ids are never reused
    uint32_t n_dead = 0;
    std::vector<uint32_t> purged_owners;   // HNSW.NODE.DEL on a one-directional graph: owners of the rows the inbound sweep edited
    // search scratch
    // HBM spill tables of the visited sets: kSpillRegions regions, handed out round-robin so that
    // launches overlapping on different streams never share one (an event per region orders reuse)
    uint32_t *d_spill = nullptr;
    uint32_t *d_spill_one = nullptr;   // one table for the single-wave exact insert / delete kernels: holds every id of the index
    uint32_t spill_one_gnb = 0;
    uint32_t spill_gnb = 0, spill_slots = 0;
    hipEvent_t spill_ev[4] = {nullptr, nullptr, nullptr, nullptr};
    bool spill_busy[4] = {false, false, false, false};
    // the specialised search kernel uses no spill region: one "last search" event per caller stream instead
    static constexpr uint32_t kSearchStreams = 16;
    hipEvent_t search_ev[kSearchStreams] = {};
    hipStream_t search_st[kSearchStreams] = {};
    bool search_busy[kSearchStreams] = {};
    uint32_t search_rr = 0;
    uint32_t spill_rr = 0;
    // ---- the search pipeline (hnsw_search_batch / hnsw_search_batch_device): kPipe engine-owned streams,
    // each with its own staging (device queries + results, pinned host mirrors) and fork / join events
    static constexpr uint32_t kPipe = 3;
    hipStream_t pipe_st[kPipe] = {};
    hipEvent_t pipe_done[kPipe] = {};      // recorded after a lane's last operation of a call
    hipEvent_t pipe_fork = nullptr;        // the caller's stream at entry (_device form)
    float *pipe_dq[kPipe] = {};            // device queries of one chunk
    uint32_t *pipe_dres[kPipe] = {};       // device [ids c*k][sims c*k][n_out c]
    float *pipe_hq[kPipe] = {};            // pinned mirrors
    uint32_t *pipe_hres[kPipe] = {};
    size_t pipe_q_words = 0, pipe_r_words = 0;
    uint32_t pipe_chunk = 1024;            // queries per chunk (tuning "pipe_chunk")
    uint32_t pipe_min_batch = 1536;        // batches at least this large are pipelined (tuning "pipe_min_batch")
    int pipe_overlap = -1;                 // measured at first use: 1 the lanes run concurrently, 0 they serialise (hardware queues alias)
    float pipe_probe_ratio = 0.f;          // (all lanes together) / (one lane alone), spin-kernel probe
    bool pipe_copy_warm = false;           // each lane has done one copy in each direction
    bool pipe_prio = false;                // lanes were re-created with distinct priorities to get queues of their own
    float *d_Q = nullptr;
    uint32_t *d_res = nullptr;       // [ids B*k][sims B*k][n_out B] of the host-buffer entry points
    size_t stage_q = 0, stage_r = 0;
    uint32_t *h_pinned = nullptr;    // pinned mirror for batches <= kPinnedBatch
    size_t pinned_words = 0;
    // insert scratch
    uint32_t *d_plan = nullptr;     // [plan_slots][kMaxLayers][1 + 64]
    uint32_t plan_slots = 0;
    uint32_t *d_touched = nullptr;  // exact insert touched list
    uint32_t touched_cap = 0;
    // single hnsw_add: pinned staging for the vector going in and the header + the head of the touched list coming
    // back (one stream synchronisation per call instead of pageable copies that each wait)
    static constexpr uint32_t kInsTouched = 4096;
    uint32_t *h_ins = nullptr;      // [dim floats][DevHeader][kInsTouched ids]
    uint32_t ins_dim_words = 0;
    uint32_t ins_touched_have = 0;  // ids of the last insert's touched list already on the host
    uint32_t *d_work = nullptr;     // fast build: shrink worklist
    // exact-order parallel insert (hnsw_occ.hpp)
    hnsw::OccSlot *d_occ_slots = nullptr;
    hnsw::OccRead *d_occ_reads = nullptr;
    hnsw::OccShr *d_occ_shr = nullptr;
    hnsw::OccDelta *d_occ_ring = nullptr;
    hnsw::OccCtl *d_occ_ctl = nullptr;
    // ... its commits in validated parallel groups (hnsw_occ_par.hpp): per-workgroup state, deltas and scratch rows
    void *d_par = nullptr, *d_par_delta = nullptr;
    uint32_t *d_par_rows = nullptr;
    uint32_t par_ovstride = 0;
    // k_occ_commit_par's grid barrier needs every workgroup resident: the co-resident count of the kernel / LDS size last asked for
    const void *par_res_kernel = nullptr;
    size_t par_res_lds = 0;
    uint32_t par_res_n = 0;
    uint32_t par_max_resident = 0xFFFFFFFFu;   // tests: pretend the device holds at most this many (tuning "par_max_resident")
    bool plan_split = true;         // tuning: a far node's upper layers are planned a round early (OccSlot::stage, hnsw_plan_lean.hpp)
    uint32_t plan_split_x10 = 7;   // ... for nodes at window positions >= this/10 x the running yield + 2
    uint32_t plan_split_pos = 0xFFFFFFFFu;   // ... from this window position on, for the round being launched (set by add_exact_window)
    int commit_par = 1;             // tuning: a window's commits go in validated parallel groups (hnsw_occ_par.hpp): 0 never (the in-order commit
                                    // wave only), 1 when the window has been committing at least commit_par_min_x10 / 10 nodes per round (a group
                                    // costs one dry run whatever its size: below ~5 nodes per round the in-order wave is as fast), 2 always
    uint32_t commit_par_min_x10 = 45;
    bool occ_chained = false;       // the round being launched was enqueued ahead of the host: its kernels take the window from the control block
    uint32_t occ_chain = 4;         // tuning: rounds of the windowed insert enqueued per host synchronisation
    bool occ_fresh_slots = false;   // the round about to be launched starts from cleared slots (single hnsw_add)
    bool occ_want_touched = false;  // the commit kernel records the update_fn list (a single hnsw_add through a one-node window)
    bool single_window = true;      // tuning: a single hnsw_add runs as a one-node window (speculative shrinks in parallel) instead of the serial kernels
    uint32_t occ_window = 64;       // tuning: window slots (0 = the serial path only)
    uint32_t occ_min_batch = 64;    // batches smaller than this take the serial path
    uint32_t occ_log_cap = hnsw::kOccMaxReads;   // tests: a tiny read log sends every node of the window to the serial kernels
    uint32_t occ_slack_extra = 0;   // tests: demand this much more free room per row (exercises the restride stop)
    uint32_t occ_slack_base = 0;    // measurements only: free room per row an insert is assumed to need instead of m + 2 (its worst case); a
                                    // row that does overflow is reported (ST_ROW_OVERFLOW), never silent
    uint32_t occ_ahead_x10 = 15;    // tuning: FRONT of the group commit (nodes dry-run side by side) = this/10 x running yield + 3, at most occ_front_max
    uint32_t occ_front_max = 64;
    uint32_t occ_depth_x10 = 0;     // tuning: DEPTH of the planned window = this/10 x running yield + 6 (>= the front, <= occ_window); 0 = the front
    uint32_t occ_stage_ahead = 32;  // tuning: nodes beyond the window whose layers above 0 are planned ahead (k_occ_plan_lean `far`; 0 = off)
    uint32_t occ_far = 0;           // ... for the round being launched (set by add_exact_window)
    uint32_t occ_front = 0;         // the front of the round being launched (0: the whole window, as a single hnsw_add / a delete have it)
    double occ_yield = 4.0;         // commits per round, running average (sizes the look-ahead)
    uint64_t occ_rounds = 0;
    hnsw::OccCtl occ_last = {};     // counters of the last windowed build (hnsw_debug_occ)
    uint32_t work_cap = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr, ev_sync = nullptr;
    bool ev_valid = false;
    bool time_launches = false;     // tuning: bracket every search launch with events (hnsw_last_search_kernel_ms)
    int lds_buckets_override = -1;
    bool tag_table = true;          // 16-bit tag visited table when the id range allows it
    int tag_bb_override = -1;       // tests: force log2(buckets) of the tag table
    int idbits_override = -1;       // tests: hash ids as an index of 2^idbits nodes would
    bool asymmetric = false;        // links may be one-directional (the fast build prunes that way; imports are checked)
    uint32_t lds_fill_x2 = 13;       // LDS visited table holds lnb * fill/2 ids (of 7 per bucket) before spilling
    int grid_override = -1;
    int fmt = 0;                     // storage format of d_vec: FMT_F32, or a compressed read-only serving copy (FMT_BF16 / FMT_FP8)
    bool bf16 = false;               // fmt == FMT_BF16 (the specialised dim-128 kernel has a bf16 form)
    bool select_shortcut = true;     // select_neighbors after search_level = the head of W (hnsw_insert.hpp); 0 = the full extension
    bool plan_lean = true;           // dim-128 insert plans (single adds, the windowed exact build) search with the specialised routine (hnsw_plan_lean.hpp)
    // tuning "tie_mode": 0 off; 1 an insert / a query the tie census flags is redone in the reference binary's own tie order
    // (hnsw_std_heap.hpp: one wavefront per operation, std's BinaryHeap restated); 2 EVERY insert and query runs there (tests: the port itself)
    int tie_mode = 0;
    uint32_t *d_tie_flags = nullptr;  // [cap] per-query flags of the census kernel, then [4] a count, then [cap] the flagged queries
    uint32_t tie_flags_cap = 0;
    hipEvent_t std_ev = nullptr;     // the last std-order search launch (they share scratch contexts)
    bool std_ev_valid = false;
    uint64_t tie_redone = 0;         // inserts of windowed builds handed to the std-order kernel (hnsw_debug_tie_redone)
    void *d_std_stamp = nullptr, *d_std_heaps = nullptr, *d_std_ctx = nullptr, *d_std_misc = nullptr;
    uint32_t std_cap = 0, std_hcap = 0;
    struct { void *stamp, *epoch, *heaps; uint32_t hcap; void *status; } std_ctx0 = {};   // (layout of hnsw::StdScratch: checked where it is used)
    bool tie_uncounted = false;      // a search ran without a census kernel while tie_census was on (since the last hnsw_reset_counters)
    bool tie_census = false;         // tuning: searches run the census form of the specialised kernel (hnsw_get_tie_counters; f32 rows, one wave per query)
    bool lean = true;                // dim-128 searches use the specialised kernel (hnsw_search_lean.hpp) when its preconditions hold
    bool duo = true;                 // ... in its two-wave form (hnsw_search_duo.hpp) when at most duo_max queries are in flight
    uint32_t duo_max = 1024;         // 4 workgroups of two waves per CU: every query gets two SIMD slots
    bool last_search_duo = false;
    bool wide_m = false;             // M > 64: serial insert / delete kernels only
    int commit_team = 1;             // the commit kernels run with three helper wavefronts (hnsw_tu_occteam.hip)
    bool plan_duo = true;            // insert plans (always lone chains) run in the two-wave form when at most plan_duo_max are launched at once
    uint32_t plan_duo_max = 256;     // one two-wave workgroup per CU
    size_t lds_reserve = 0;          // LDS a kernel needs besides the wave's own share (the OCC kernels' validation scratch)
    bool grid_stride = false;        // specialised kernel: cap the grid at the resident waves and walk the batch grid-stride (tuning, for comparison)
    bool visited_bounded = true;     // k_search: a full LDS visited table stops recording (exact results)
    bool wpc_user = false;           // waves_per_cu was set by the caller (else the kernel's own best residency is used)
    uint32_t max_waves_per_cu = 8;   // residency the LDS visited table is sized for (tuning: waves_per_cu)
    uint32_t launch_concurrency = 0; // tuning: search launches the CALLER keeps in flight at once (0 = observed per launch, see search_concurrency)
    uint32_t pipe_inflight = 1;      // ... and how many the engine's own pipeline has in flight right now
    uint32_t cur_conc = 1;           // what the launch being enqueued sizes its LDS share for
    bool last_search_lean = false;   // the latest search launch was the specialised kernel's (hnsw_debug_last_search_path)
    bool pipe_device = false;        // tuning: the _device entry point splits large batches over the lanes too (measured slower than one launch: a lane's next chunk waits for its previous chunk's last wave)
    uint32_t fast_seed = 512, fast_batch_max = 4096, fast_batch_div = 8;
    uint64_t rng[4] = {0, 0, 0, 0};
    uint64_t hbm_bytes = 0;
    std::string err;
};

namespace hnsw_host {

using namespace hnsw;

// free words per row a commit demands before it starts: one insert raises any row by at most m + 1 per layer (connect: +1;
// each of its m shrinks may append itself to the same third party, core.rs:794)
inline uint32_t occ_slack(const hnsw_index *h) { return (h->occ_slack_base ? h->occ_slack_base : h->m + 2) + h->occ_slack_extra; }

// Every C-ABI entry runs on its index's device and leaves the calling thread's current HIP device as it found it
// (a Redis module shares its threads with whatever else the process does with HIP).
struct DeviceScope {
    int prev = -1, dev;
    hipError_t err = hipSuccess;
    explicit DeviceScope(int d) : dev(d)
    {
        if (hipGetDevice(&prev) != hipSuccess) { prev = -1; (void)hipGetLastError(); }
        if (prev != dev) err = hipSetDevice(dev);
    }
    ~DeviceScope()
    {
        if (prev >= 0 && prev != dev) (void)hipSetDevice(prev);
    }
    DeviceScope(const DeviceScope &) = delete;
    DeviceScope &operator=(const DeviceScope &) = delete;
};
#define ON_DEVICE(h)                       \
    DeviceScope dev_scope_((h)->device);   \
    HIP_TRY(h, dev_scope_.err)

#define HIP_TRY(h, expr)                                                                       \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess) {                                                                \
            (h)->err = std::string(#expr) + ": " + hipGetErrorString(e_);                      \
            return HNSW_ERR_DEVICE;                                                            \
        }                                                                                      \
    } while (0)

// Variant of the translation unit being compiled (see the file header).
#if defined(HNSW_VARIANT)
#if HNSW_VARIANT == 0
constexpr int kVarMode = MODE_SCALAR, kVarT = 0;
#elif HNSW_VARIANT == 1
constexpr int kVarMode = MODE_AVX, kVarT = 4;
#elif HNSW_VARIANT == 2
constexpr int kVarMode = MODE_AVX, kVarT = 24;
#elif HNSW_VARIANT == 3
constexpr int kVarMode = MODE_AVX, kVarT = 0;
#else
constexpr int kVarMode = MODE_AVX, kVarT = 0;     // 4, 5: the compressed-storage kernels of hnsw_tu_search.hip
#endif
#endif

// LDS visited-set configuration of one launch
struct VisCfg {
    uint32_t lnb;      // 32-byte buckets (32-bit id mode); also sizes LDS in tag mode via `bytes`
    uint32_t lcap;     // ids before the set moves to HBM
    uint32_t tagcfg;   // 0, or log2(16-byte buckets) | idbits << 8
    size_t bytes;      // LDS bytes of the table
};

struct InsertCfg {
    int R;
    uint32_t lnb, lcap, tagcfg;
    size_t lds;
};

// ---- defined in hnsw_engine.hip ---------------------------------------------------------------------
hnsw_status fail(hnsw_index *h, hnsw_status s, const std::string &msg);
GraphView view(const hnsw_index *h);
GraphView view_tag(const hnsw_index *h, uint32_t tagcfg);
VisCfg pick_vis(const hnsw_index *h, int R, int T, bool ins, uint32_t nwaves);
hnsw_status spill_acquire(hnsw_index *h, hipStream_t st, uint32_t *region, uint32_t **base);
hnsw_status spill_release(hnsw_index *h, hipStream_t st, uint32_t region);
hnsw_status note_search(hnsw_index *h, hipStream_t st);
hnsw_status wait_inflight_searches(hnsw_index *h);

// The dynamic-LDS attribute sticks to the kernel function (process-wide), so it is raised once per size class
// instead of once per launch; `have` is the launcher's static high-water mark per device.  Handles on different
// threads may launch the same kernel: the check-and-set is serialised.
template <typename Kern>
hnsw_status raise_lds_attr(hnsw_index *h, Kern kern, size_t lds, size_t (&have)[16])
{
    static std::mutex mu;
    std::lock_guard<std::mutex> lock(mu);
    size_t &cur = have[h->device & 15];
    if (lds > cur) {
        HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        cur = lds;
    }
    return HNSW_OK;
}

// ---- the launchers: defined + explicitly instantiated in the family files ----------------------------------
// hnsw_tu_search.hip
template <int MODE, int T>
hnsw_status launch_search_r(hnsw_index *h, int R, const float *dQ, uint32_t B, uint32_t k, uint32_t *d_ids,
                            float *d_sims, uint32_t *d_nout, hipStream_t st);
template <int FMT>
hnsw_status launch_search_fmt(hnsw_index *h, int R, const float *dQ, uint32_t B, uint32_t k, uint32_t *d_ids,
                              float *d_sims, uint32_t *d_nout, hipStream_t st);
// hnsw_tu_lean.hip: launches k_search_lean<VEC,R,BB,DB> if that instantiation exists (*done), else leaves *done false
template <class VEC, bool WIDE>
hnsw_status launch_lean_v(hnsw_index *h, int R, uint32_t bb, uint32_t db, const float *dQ, uint32_t B, uint32_t k,
                          uint32_t idbits, uint32_t per_cu, uint32_t *d_ids, float *d_sims, uint32_t *d_nout,
                          hipStream_t st, bool *done);
// hnsw_tu_duo.hip: the two-wave form (a walker and a W-keeper wavefront per query, hnsw_search_duo.hpp), f32 rows
template <bool WIDE>
hnsw_status launch_duo_v(hnsw_index *h, int R, uint32_t db, const float *dQ, uint32_t B, uint32_t k, uint32_t idbits,
                         uint32_t *d_ids, float *d_sims, uint32_t *d_nout, hipStream_t st, bool *done);
// hnsw_tu_insert.hip
template <int MODE, int T>
hnsw_status launch_insert_r(hnsw_index *h, const InsertCfg &c, bool plan, uint32_t first, uint32_t count);
template <int MODE, int T>
hnsw_status launch_shrink_t(hnsw_index *h, uint32_t *pending0, uint32_t *pendingU, uint32_t *work_n);
template <int MODE, int T>
hnsw_status launch_delete_r(hnsw_index *h, const InsertCfg &c, uint32_t id);
// hnsw_tu_planlean.hip: the plans that search with the specialised dim-128 routine (narrow / wide adjacency rows)
template <bool WIDE>
hnsw_status launch_plan_lean_v(hnsw_index *h, const InsertCfg &c, uint32_t first, uint32_t count, uint32_t idbits);
template <bool WIDE>
hnsw_status launch_occ_plan_lean_v(hnsw_index *h, const InsertCfg &c, const OccBufs &ob, uint32_t head, uint32_t count, uint32_t idbits);
// hnsw_tu_occteam.hip: the commit kernels with three helper wavefronts sharing every recomputed select_neighbors
// (*done stays false when the CU's LDS has no room for the helpers: the caller launches the one-wave kernel)
template <int MODE, int T>
hnsw_status occ_commit_team_r(hnsw_index *h, const InsertCfg &c, const OccBufs &ob, uint32_t end_node, bool *done);
template <int MODE, int T>
hnsw_status occ_del_commit_team_r(hnsw_index *h, const InsertCfg &c, const OccBufs &ob, uint32_t id, bool *done);
// hnsw_tu_occpar.hip: the round's commits in validated parallel groups, one workgroup per window node (*done stays false
// when the kernel cannot serve the round: the caller launches the in-order commit)
template <int MODE, int T>
hnsw_status occ_commit_par_r(hnsw_index *h, const InsertCfg &c, const OccBufs &ob, uint32_t count, uint32_t end_node, bool *done);
// hnsw_tu_planduo.hip: the same plans with a second wavefront keeping W for their layer searches
template <bool WIDE>
hnsw_status launch_plan_duo_v(hnsw_index *h, const InsertCfg &c, uint32_t first, uint32_t count, uint32_t idbits);
template <bool WIDE>
hnsw_status launch_occ_plan_duo_v(hnsw_index *h, const InsertCfg &c, const OccBufs &ob, uint32_t head, uint32_t count, uint32_t idbits);
// hnsw_engine.hip: 0 when the specialised plan cannot serve this index, else the id-hash width to launch it with
uint32_t plan_lean_idbits(const hnsw_index *h, const InsertCfg &c);
hnsw_status ensure_par(hnsw_index *h);
size_t plan_lean_lds(int R);
hnsw_status launch_occ_plan_lean(hnsw_index *h, const InsertCfg &c, const OccBufs &ob, uint32_t head, uint32_t count, bool *done);
// hnsw_tu_std.hip: the reference binary's own tie order on one wavefront per operation (tuning "tie_mode")
hnsw_status launch_insert_std(hnsw_index *h, uint32_t id, bool want_touched);
hnsw_status launch_search_std(hnsw_index *h, const float *dQ, uint32_t B, uint32_t k, uint32_t *d_ids, float *d_sims, uint32_t *d_nout, bool all,
                              hipStream_t st);
hnsw_status ensure_tie_flags(hnsw_index *h, uint32_t B);
hnsw_status launch_delete_std(hnsw_index *h, uint32_t id);
hnsw_status std_status(hnsw_index *h, uint32_t *out);
// hnsw_tu_occ.hip
template <int MODE, int T>
hnsw_status occ_round_r(hnsw_index *h, const InsertCfg &c, const OccBufs &ob, uint32_t head, uint32_t count, uint32_t end_node);
template <int MODE, int T>
hnsw_status occ_delete_r(hnsw_index *h, const InsertCfg &c, const OccBufs &ob, uint32_t id);

} // namespace hnsw_host
