/*
 * hnsw_mi355x.h -- C ABI of the MI355X-native HNSW engine (libhnsw_mi355x.so).
 *
 * This is the drop-in boundary for the hot path of zhao-lang/redis_hnsw: the
 * Rust module keeps its command layer (src/lib.rs) and its name <-> node map,
 * and binds these entry points with `extern "C"` where it calls
 * Index<f32,f32> today (INTEGRATION.md shows the shim).  Citations are to the
 * reference tree.
 *
 * Conventions
 *  - Nodes are dense u32 ids in insertion order; names never cross the ABI.
 *  - Similarity is the reference's: sim = -(squared L2), larger is closer
 *    (src/hnsw/metrics.rs:75,80), bit-identical summation order.
 *  - All pointers are caller owned.  "host" entry points take host memory and
 *    return when the result is in the output buffers.  "_device" entry points
 *    take device (HBM) pointers + a hipStream_t and only enqueue work.
 *  - One caller at a time per handle (the reference runs on the Redis command
 *    thread and takes try_write/try_read, src/lib.rs:349,474).
 *  - Every function returns an hnsw_status; hnsw_last_error() gives the text.
 *  - There is no CPU fallback: without a usable gfx950 device hnsw_create
 *    fails with HNSW_ERR_DEVICE.
 *  - Vector components must be finite: the host entry points return
 *    HNSW_ERR_INVALID otherwise (the reference's OrderedFloat NaN order,
 *    core.rs:4,241, is not reproduced); the _device entry point does not check.
 */
#ifndef HNSW_MI355X_H
#define HNSW_MI355X_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct hnsw_index hnsw_index;

typedef enum {
    HNSW_OK = 0,
    HNSW_ERR_DIM_MISMATCH = 1, /* "data dimension: {} does not match Index" core.rs:390,479 */
    HNSW_ERR_DUPLICATE = 2,    /* "Node: {:?} already exists" core.rs:408 (raised by the host mirror) */
    HNSW_ERR_NOT_FOUND = 3,    /* "Node: {:?} does not exist" core.rs:421 */
    HNSW_ERR_DEVICE = 4,       /* HIP failure / no gfx950 device */
    HNSW_ERR_INVALID = 5,      /* bad argument or unsupported parameter */
    HNSW_ERR_CAPACITY = 6      /* internal table overflow (reported, never silent) */
} hnsw_status;

/* work counters, summed over the calls since the last reset    */
typedef struct {
    uint64_t n_dist;   /* vectors fetched + metric evaluations               */
    uint64_t n_ids;    /* neighbour ids scanned                              */
    uint64_t n_expand; /* candidates expanded                                */
    uint64_t n_spill;  /* queries whose visited set spilled from LDS to HBM  */
} hnsw_counters;

typedef struct {
    uint32_t dim, m, m_max, m_max0, ef_construction;
    uint32_t node_count, max_layer;
    int64_t enterpoint;        /* -1 when the index is empty                 */
    uint32_t stride0, stride_upper; /* words per adjacency row (slot 0 = count) */
    uint32_t max_degree0, max_degree_upper;
    uint64_t hbm_bytes;        /* device memory held by the handle           */
    uint32_t allocated_ids;    /* ids handed out so far (deleted ids are not reused) */
} hnsw_info;

/* Index::new (core.rs:322-347): m_max = m, m_max0 = 2m, level_mult = 1/ln m.
 * seed feeds the engine's own level generator (used when level < 0 is passed
 * to hnsw_add; the reference seeds from entropy, core.rs:344).  device is a
 * HIP device ordinal.                                                        */
hnsw_status hnsw_create(uint32_t dim, uint32_t m, uint32_t ef_construction,
                        uint64_t seed, int device, hnsw_index **out);
void hnsw_destroy(hnsw_index *h);
const char *hnsw_last_error(const hnsw_index *h); /* valid until the next call on h */

/* Index::add_node (core.rs:383-412 -> insert :489-599), executed on the GPU,
 * the reference's serial algorithm link for link (one call runs as a one-node window of
 * csrc/hnsw_occ.hpp: the shrinks its connect triggers are computed in parallel, then
 * validated and applied in the reference's order; "single_window" = 0 runs the one-wave
 * serial kernels instead -- same graph, same update_fn ids).  level < 0 draws
 * floor(-ln U / ln m) (core.rs:601-605); the first node ignores it
 * (core.rs:393-405).  touched (may be NULL) receives the ids the reference
 * would pass to update_fn (core.rs:580-584), each once, unordered; *n_touched
 * is their number even when it exceeds touched_cap (only touched_cap are
 * written then: the caller must treat that as an error, not skip the rest --
 * AFTER recording the new id, because the insert itself is complete whenever
 * the status is HNSW_OK; UINT32_MAX = the engine could not list them at all). */
hnsw_status hnsw_add(hnsw_index *h, const float *v, uint32_t dim, int32_t level,
                     uint32_t *out_id, uint32_t *touched, uint32_t touched_cap,
                     uint32_t *n_touched);

/* Bulk build (BASELINE.json config 5).  mode 0 = exact: the graph of n sequential
 * hnsw_add() calls (core.rs:489-599 in insert order), link for link.  Batches of
 * 64 or more are planned in parallel and committed strictly in id order after
 * each plan has been validated against the rows changed since it was made
 * (csrc/hnsw_occ.hpp; "occ_window" = 0 runs the inserts one after the other).
 * mode 1 = fast: inserts are planned in data-parallel batches against a
 * snapshot and committed together -- NOT link-for-link identical to the
 * reference's serial order; judged by recall parity only.  levels may be NULL. */
hnsw_status hnsw_add_batch(hnsw_index *h, const float *V, uint32_t n, uint32_t dim,
                           const int32_t *levels, uint32_t mode);

/* Index::delete_node (core.rs:414-475 -> delete_node_from_neighbors :824-863), executed on the GPU
 * in the reference's serial order (the neighbours' re-selections are computed in parallel, then validated
 * and applied in that order; "single_window" = 0: one after the other).  The id becomes a tombstone (never reused); node_count drops.
 * If the node was the enterpoint, the new one is the smallest id of the highest non-empty layer
 * (the reference takes an arbitrary node of that layer, core.rs:453).  HNSW_ERR_NOT_FOUND if id is
 * not a live node (core.rs:421).  touched as for hnsw_add.                                        */
hnsw_status hnsw_delete(hnsw_index *h, uint32_t id, uint32_t *touched, uint32_t touched_cap,
                        uint32_t *n_touched);

/* Index::search_knn (core.rs:477-486 -> :865-892); ef = ef_construction
 * (core.rs:485).  *n_out = min(k, ef, reachable) results, nearest first; ids and
 * sims must have room for k entries (entries past *n_out are padded like the
 * batch form's).  An empty index returns HNSW_OK with *n_out = 0
 * (core.rs:481-483).                                                          */
hnsw_status hnsw_search(hnsw_index *h, const float *q, uint32_t dim, uint32_t k,
                        uint32_t *ids, float *sims, uint32_t *n_out);

/* B independent queries, Q row-major [B][dim]; ids/sims are [B][k] (rows are
 * padded with id 0xFFFFFFFF / sim -inf past n_out[b]); n_out is [B].          */
hnsw_status hnsw_search_batch(hnsw_index *h, const float *Q, uint32_t B, uint32_t dim,
                              uint32_t k, uint32_t *ids, float *sims, uint32_t *n_out);

/* Same, with every buffer already resident in HBM; enqueues on `stream`
 * (a hipStream_t, NULL = the null stream) and returns without synchronising.
 * One call is one kernel launch with one workgroup per query: the dispatcher hands the next query to
 * whichever of the chip's 2048 wave slots frees first, which is the best a single call can do (measured:
 * splitting a call into chunks on several streams is slower -- a stream's next chunk waits for the last
 * wave of its previous one; "pipe_device" = 1 keeps that form for comparison).  What one call cannot hide
 * is its own drain: the last queries run on a nearly empty chip (0.76 of the steady rate at 4096 queries,
 * 0.82 at 8192).  A caller with more than one batch hides it by keeping calls in flight on two or three
 * streams of its own; nothing needs tuning for that -- the engine sizes each launch's LDS share from the
 * launches it sees in flight.  hnsw_search_batch (host buffers) pipelines its copies and kernels on the
 * engine's own lanes.  Inserts and deletes wait for every search enqueued before them, whatever its stream. */
hnsw_status hnsw_search_batch_device(hnsw_index *h, const float *dQ, uint32_t B,
                                     uint32_t dim, uint32_t k, uint32_t *d_ids,
                                     float *d_sims, uint32_t *d_n_out, void *stream);

/* The search pipeline's lanes need a hardware queue each.  The library asks the HIP runtime for 8
 * (GPU_MAX_HW_QUEUES, set at load time unless the operator exported a value; effective when the library is
 * loaded before the runtime starts) and measures, when the lanes are created, whether they really run
 * concurrently: overlap = 1 yes, 0 they serialise (reported once on stderr; fatal with
 * HNSW_REQUIRE_OVERLAP=1 in the environment), -1 not created yet.  probe_ratio = (a 200 us spin kernel on
 * every lane at once) / (on one lane); priorities = 1 when the lanes had to be given distinct stream
 * priorities to get queues of their own.  Creates the lanes if no batched search has done so yet.          */
typedef struct {
    uint32_t lanes;
    int32_t overlap;
    float probe_ratio;
    uint32_t priorities;
    uint32_t chunk, min_batch;
    uint32_t hw_queues_env;    /* GPU_MAX_HW_QUEUES as this process sees it (0 = unset) */
} hnsw_pipeline;
hnsw_status hnsw_pipeline_info(hnsw_index *h, hnsw_pipeline *out);

/* Replaces make_index (src/lib.rs:252-315): load a frozen graph straight into
 * HBM.  vectors [n][dim]; levels [n]; per layer l < n_layers a CSR
 * (row_ptr[l] has n+1 entries, col[l] the neighbour ids in stored order).    */
hnsw_status hnsw_import(hnsw_index *h, uint32_t n, const float *vectors,
                        const uint32_t *levels, int64_t enterpoint, uint32_t n_layers,
                        const uint64_t *const *row_ptr, const uint32_t *const *col);

/* One-time index distribution to replicas (search_knn takes &self, core.rs:477, so every GPU may
 * hold a copy), without a host hop: the tables are handed out as DEVICE pointers, so that a collective
 * (RCCL broadcast over xGMI) or a peer copy moves them HBM to HBM.
 *   source       hnsw_replica_view(src, &r)          r = sizes + pointers into src's HBM (valid until the next
 *                                                     call that grows, restrides, inserts into or destroys src)
 *   destination  copy r's scalar fields from the source's, then
 *                hnsw_replica_prepare(dst, &r)       allocates for them on dst's device, fills in dst's pointers
 *                <move vec_bytes / adj0_bytes / adj_upper_bytes / 4n / 4n bytes into vec, adj0, adj_upper,
 *                 upper_base, levels -- in the source's row layout: stride0 / stride_upper words per row>
 *                hnsw_replica_commit(dst, &r, dead)  adopts the tables; dead = n tombstone bytes (0/1, host
 *                                                     memory) or NULL when n_dead = 0
 * dst must be an empty index created with the source's dim, M and ef_construction.  The replica is an exact copy
 * (same rows in the same stored order, same enterpoint): it answers every search bit for bit like the source,
 * and continues like it under exact inserts and deletes (its level generator keeps its own seed).            */
typedef struct {
    uint32_t n;                /* ids handed out (rows of vec / adj0 / levels / upper_base)                   */
    uint32_t dim;
    uint32_t upper_used;       /* rows of adj_upper                                                            */
    uint32_t stride0, stride_upper;
    uint32_t max_layer, max_degree0, max_degree_upper;
    uint32_t n_dead, asymmetric, format;   /* format: 0 f32, 1 bf16, 2 fp8 (compressed serving copies)          */
    uint32_t reserved;
    int64_t enterpoint;
    uint64_t vec_bytes, adj0_bytes, adj_upper_bytes;
    void *vec, *adj0, *adj_upper, *upper_base, *levels;    /* device memory                                    */
} hnsw_replica;
hnsw_status hnsw_replica_view(hnsw_index *h, hnsw_replica *out);
hnsw_status hnsw_replica_prepare(hnsw_index *h, hnsw_replica *inout);
hnsw_status hnsw_replica_commit(hnsw_index *h, const hnsw_replica *r, const uint8_t *dead);
hnsw_status hnsw_get_tombstones(hnsw_index *h, uint8_t *dead /*[allocated_ids]*/);

/* ---- one process, several GPUs ("one process, 8 devices, one stream each") ----------------------
 * What a Redis module -- one process -- uses instead of one rank per GPU: a GROUP is the primary index plus one
 * replica per further device.  Replicas are made by hnsw_replica_view / _prepare / _commit with peer copies
 * (HBM to HBM over xGMI, no host hop).  Searches are independent (search_knn takes &self, core.rs:477): a batch is
 * split contiguously, member g takes queries [g*B/G, (g+1)*B/G), every member runs its slice through
 * hnsw_search_batch on its own device at the same time, results land in the caller's buffers in query order --
 * no collective, no merge (disjoint queries, not disjoint data).  Writes keep the members identical by being
 * REPLAYED: insert (core.rs:489-599) and delete_node (:414-475) are deterministic given the level, so
 * hnsw_group_add / _delete / _add_batch(mode 0) run the same exact operation on every member concurrently, with
 * the level drawn once by the group (floor(-ln U / ln m), core.rs:601-605, its own seeded generator).  The fast
 * build (mode 1) is not reproducible link for link: it runs on the primary and the replicas are re-copied
 * (hnsw_group_refresh, also the way back after writing to the primary directly).
 * A device may be listed more than once (several members on one GPU: how the path is tested on a one-GPU box).
 * One caller at a time per group; do not use the primary's handle concurrently with group calls.             */
typedef struct hnsw_group hnsw_group;
hnsw_status hnsw_group_create(hnsw_index *primary, const int *devices, uint32_t n_devices, uint64_t seed,
                              hnsw_group **out);             /* devices: where the REPLICAS go (may be empty);
                                                                * on failure *out still holds a group to ask
                                                                * hnsw_group_last_error and to destroy          */
void hnsw_group_destroy(hnsw_group *g);                      /* destroys the replicas, not the primary        */
const char *hnsw_group_last_error(const hnsw_group *g);
uint32_t hnsw_group_size(const hnsw_group *g);               /* members, primary included                     */
hnsw_index *hnsw_group_member(hnsw_group *g, uint32_t i);    /* 0 = the primary                               */
hnsw_status hnsw_group_refresh(hnsw_group *g);               /* re-copy every replica from the primary        */
/* Index::search_knn for B queries, sharded over the members (arguments as hnsw_search_batch)                  */
hnsw_status hnsw_group_search_batch(hnsw_group *g, const float *Q, uint32_t B, uint32_t dim, uint32_t k,
                                    uint32_t *ids, float *sims, uint32_t *n_out);
/* Index::add_node / delete_node on every member (arguments as hnsw_add / hnsw_delete; touched from the primary) */
hnsw_status hnsw_group_add(hnsw_group *g, const float *v, uint32_t dim, int32_t level, uint32_t *out_id,
                           uint32_t *touched, uint32_t touched_cap, uint32_t *n_touched);
hnsw_status hnsw_group_delete(hnsw_group *g, uint32_t id, uint32_t *touched, uint32_t touched_cap,
                              uint32_t *n_touched);
hnsw_status hnsw_group_add_batch(hnsw_group *g, const float *V, uint32_t n, uint32_t dim, const int32_t *levels,
                                 uint32_t mode);

/* Export for IndexRedis/NodeRedis write-through (src/types.rs:62-91,292-309). */
hnsw_status hnsw_get_info(hnsw_index *h, hnsw_info *info);
hnsw_status hnsw_get_levels(hnsw_index *h, uint32_t *levels /*[allocated_ids]*/);
hnsw_status hnsw_get_level(hnsw_index *h, uint32_t id, uint32_t *level);  /* one node's top layer (core.rs:596), O(1) */
hnsw_status hnsw_get_vector(hnsw_index *h, uint32_t id, float *out /*[dim]*/);
hnsw_status hnsw_get_neighbors(hnsw_index *h, uint32_t id, uint32_t layer,
                               uint32_t *out, uint32_t cap, uint32_t *n);
hnsw_status hnsw_layer_nnz(hnsw_index *h, uint32_t layer, uint64_t *nnz);
hnsw_status hnsw_export_layer(hnsw_index *h, uint32_t layer, uint64_t *row_ptr /*[n+1]*/,
                              uint32_t *col);

/* Snapshot of the whole index (parameters, level generator state, vectors, levels, tombstones,
 * per-layer adjacency in stored order) as one byte string: what the module's RDB callbacks
 * (save_index / load_index, src/types.rs:176-284) stream instead of per-node keys.  A restored
 * index continues exactly where the saved one was (same graph, same future level draws).       */
hnsw_status hnsw_serialize_size(hnsw_index *h, uint64_t *bytes);
hnsw_status hnsw_serialize(hnsw_index *h, void *buf, uint64_t cap, uint64_t *written);
hnsw_status hnsw_deserialize(const void *buf, uint64_t bytes, uint64_t seed, int device,
                             hnsw_index **out);

/* Engine knobs (not part of the reference surface).
 *   search    "launch_concurrency" (search launches the caller keeps in flight: sizes the LDS share; 0 =
 *             default: observed per launch), "pipe_chunk" / "pipe_min_batch" / "pipe_device" (the engine's
 *             own pipelining of large batches, see hnsw_search_batch_device),
 *             "waves_per_cu" (residency the visited table is sized for, default 8), "visited_bounded"
 *             (1: a full LDS visited table stops recording -- exact results, distance evaluations may
 *             exceed the reference's; 0: the table continues in HBM -- counters equal the reference's),
 *             "lean" (specialised dim-128 kernel on/off), "tie_census" (1: searches run the census form of that
 *             kernel, see hnsw_get_tie_counters), "tie_mode" (0: answers and links in the total order (similarity, id),
 *             the default; 1: every insert and every query the tie census flags is redone in the REFERENCE BINARY's own
 *             order -- core.rs:489-892 statement by statement on std::collections::BinaryHeap restated, one wavefront
 *             per operation (csrc/hnsw_std_heap.hpp): the graph and the answers are the Rust binary's, row for row;
 *             implies tie_census; a single hnsw_add is gated the same way (one-node window, its commit dry-run first);
 *             shapes other than dim-128 f32 rows: every insert runs there; 2: every insert / query runs there (tests);
 *             every hnsw_delete in tie mode runs there too (core.rs:824-863 on the same heap); a heap overflow of
 *             those kernels is HNSW_ERR_CAPACITY; f32 rows only: a search on a compress_bf16 / compress_fp8 index
 *             under tie_mode is HNSW_ERR_INVALID),
 *             "grid_stride", "query_in_lds", "time_launches",
 *             "lds_buckets" / "lds_hash_bits" / "tag_table" / "tag_bb" / "idbits" / "grid" (tests)
 *   build     "occ_window" (slots of the exact parallel insert, 0 = serial), "occ_min_batch",
 *             "occ_ahead_x10" / "occ_front_max" (nodes the group commit dry-runs side by side), "occ_depth_x10" (how far
 *             ahead of them nodes are planned; 0 = the same), "occ_stage_ahead" (nodes beyond the window whose layers
 *             above 0 are planned ahead), "commit_par" / "commit_par_min_x10" / "par_max_resident" (commits in validated
 *             parallel groups; the last one pretends a smaller device, tests), "plan_split" / "plan_split_x10",
 *             "occ_chain" (rounds enqueued per host synchronisation), "select_shortcut" (1: select_neighbors after search_level is the head
 *             of W, see csrc/hnsw_insert.hpp), "plan_lean" (1: dim-128 insert plans -- single hnsw_add
 *             calls and the windowed exact build -- search with the specialised routine of the search
 *             kernel, csrc/hnsw_plan_lean.hpp; same graph either way), "single_window" (1: a single hnsw_add
 *             runs as a one-node window -- its shrinks computed in parallel, then validated -- instead of
 *             the serial kernels; same graph and update_fn list), "fast_seed" / "fast_batch_max" /
 *             "fast_batch_div"
 *   storage   "compress_bf16" / "compress_fp8" (one way: 2 / 1 bytes per component in the gather, the index becomes
 *             read-only; results are the reference's on the stored, rounded values; any dim % 32 == 0),
 *             "force_restride" (tests)                                                              */
hnsw_status hnsw_set_tuning(hnsw_index *h, const char *key, int64_t value);

hnsw_status hnsw_get_counters(hnsw_index *h, hnsw_counters *search, hnsw_counters *insert);
hnsw_status hnsw_reset_counters(hnsw_index *h);

/* Tie census (since the last hnsw_reset_counters).  The reference orders SimPair by similarity ALONE (core.rs:292-300)
 * and leaves equal similarities to std::collections::BinaryHeap; the engine (like the parity oracle) breaks them by
 * the smaller id.  The two can only part where a DECISION compared equal distances of two different nodes: the stop
 * test (core.rs:635), the accept test with W full (:657), a select_neighbors cut (:733, :741-754), equal distances
 * among the k + 1 nearest of an answer -- or where an ORDER is decided between equal distances: inside a selection (which
 * neighbour is linked / shrunk / appended first, :540-541, :765-772, :790-796), W's two nearest (the next entry point),
 * the pop of :631 when another candidate is as similar as the popped one, and the eviction of :662-664 when W's furthest
 * entry at the end of a search is as similar as one that was pushed out.  The kernels count every such comparison they make -- a superset of the
 * reference's own (a whole adjacency row is merged at once where the reference walks it id by id), never fewer:
 *   out[0] events in searches      (only while tuning "tie_census" = 1: the census form of the dim-128 search kernel,
 *          f32 rows, ef_construction <= 256; if a search of the period ran on another kernel while the tuning was on,
 *          out[0] = out[1] = UINT64_MAX: unknown)
 *   out[1] queries with at least one
 *   out[2] events in inserts / deletes: the plans' search_level + select_neighbors, the speculative and the recomputed
 *          select_neighbors of the shrink loop (always counted by the dim-128 plan kernels and by every select)
 *   out[3] insert plans with at least one (a node planned twice counts twice)
 * Zero means: every answer / every link is what the reference's binary produces, whatever its heap does with ties.  */
hnsw_status hnsw_get_tie_counters(hnsw_index *h, uint64_t *out4);

/* Timing of the most recent search kernel launch measured with HIP events on
 * the stream it ran on (milliseconds); synchronises that stream.             */
hnsw_status hnsw_last_search_kernel_ms(hnsw_index *h, float *ms);

/* Device metric on its own: sims[i] = euclidean(a[i], b[i]) for n pairs of
 * host vectors (metrics.rs:14-23 dispatch: AVX2 order iff dim % 32 == 0).
 * Used by the metric known-answer tests.                                      */
hnsw_status hnsw_metric_pairs(int device, const float *a, const float *b, uint32_t n,
                              uint32_t dim, float *sims);

#ifdef __cplusplus
}
#endif
#endif
