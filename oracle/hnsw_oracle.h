/*
 * hnsw_oracle.h -- CPU parity oracle for the redis_hnsw hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  This is a plain-C restatement of the reference's
 * src/hnsw/core.rs + src/hnsw/metrics.rs (zhao-lang/redis_hnsw v0.2.1).  Only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it,
 * and there only as the checker / the timed CPU baseline.  The product
 * (redis_hnsw_amd/, libhnsw_mi355x.so) never links, imports or calls it.
 *
 * Pinning: the reference is Rust and cannot be compiled in this image (no
 * rustc/cargo), so oracle/_ref does not exist.  The oracle is pinned against
 * every known-answer test the reference holds for this path
 * (src/hnsw/metrics_tests.rs:3-33, src/hnsw/core_tests.rs:12-80, the delete
 * loop included); see tests/test_oracle_kat.py.
 *
 * Deliberate, documented restatement choices (none observable on tie-free
 * data; the fourth only when the enterpoint itself is deleted):
 *   1. Ids are dense u32 in insertion order; names stay on the caller's side.
 *   2. Rust's BinaryHeap leaves the order of EQUAL similarities unspecified
 *      (core_tests.rs:50-53 does not assert it).  The oracle fixes a total
 *      order: larger sim first, then smaller id.  Every comparison the
 *      reference makes on `sim` is made on that (sim, id) key here.
 *   3. The level RNG is entropy seeded in the reference (core.rs:344), so
 *      levels are an explicit input (or drawn from a seeded generator using
 *      the same formula floor(-ln U * 1/ln m), core.rs:338,601-605).
 *   4. delete_node re-elects the enterpoint with HashSet::iter().next()
 *      (core.rs:453), i.e. arbitrarily among the nodes of the highest non-empty
 *      layer; the oracle takes the smallest id of that layer.
 */
#ifndef HNSW_ORACLE_H
#define HNSW_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct hnsw_oracle hnsw_oracle;

/* per-call work counters */
typedef struct {
    uint64_t n_dist;   /* metric evaluations (core.rs:550,621,652,711)       */
    uint64_t n_ids;    /* neighbour ids scanned (core.rs:646,702)             */
    uint64_t n_expand; /* candidates expanded (core.rs:631 pops that expand)  */
} hnsw_oracle_counters;

/* ---- metric kernels: src/hnsw/metrics.rs -------------------------------- */
/* metrics.rs:79-84 : sequential left fold of (x-y)*(x-y), no FMA, negated   */
float hnsw_oracle_sim_scalar(const float *a, const float *b, size_t n);
/* metrics.rs:48-77 : 4x8-lane FMA accumulators, AVX2 hsum order, negated.
 * Uses real AVX2+FMA when the host has them, else a lane-exact fmaf
 * emulation (bit-identical).  n must be a multiple of 32.                   */
float hnsw_oracle_sim_avx(const float *a, const float *b, size_t n);
/* the lane-exact emulation on its own (for testing the two agree)          */
float hnsw_oracle_sim_avx_emulated(const float *a, const float *b, size_t n);
/* metrics.rs:14-23 : dispatch; AVX order iff n % 32 == 0                    */
float hnsw_oracle_euclidean(const float *a, const float *b, size_t n);

/* test switch: compare on sim alone at core.rs:635, :657, :733 (see hnsw_oracle.c); default off */
void hnsw_oracle_set_strict_ties(int on);

/* ---- index: src/hnsw/core.rs -------------------------------------------- */
/* core.rs:322-347. seed feeds the oracle's own level generator (used only
 * when add() is called with level < 0).                                     */
hnsw_oracle *hnsw_oracle_new(uint32_t dim, uint32_t m, uint32_t ef_construction,
                             uint64_t seed);
void hnsw_oracle_free(hnsw_oracle *o);

/* core.rs:383-412 + 489-599.  level < 0 => draw (core.rs:601-605).  The first
 * node ignores `level` (core.rs:393-405: no draw, level 0).  touched (may be
 * NULL) receives the ids update_fn would be called for (core.rs:580-584), in
 * unspecified order.  Returns the new node id (>= 0) or -1 on error.        */
int64_t hnsw_oracle_add(hnsw_oracle *o, const float *v, int32_t level,
                        uint32_t *touched, uint32_t touched_cap,
                        uint32_t *n_touched);

/* core.rs:477-486 + 865-892 : ef = ef_construction.  Returns the number of
 * results written (min(k, ef, reachable)), nearest first.  ctrs may be NULL. */
uint32_t hnsw_oracle_search(const hnsw_oracle *o, const float *q, uint32_t k,
                            uint32_t *ids, float *sims,
                            hnsw_oracle_counters *ctrs);

/* B independent searches on `threads` pthreads (baseline B of BASELINE.md);
 * ids/sims are [B][k], n_out is [B]; ctrs (may be NULL) is summed.          */
void hnsw_oracle_search_batch(const hnsw_oracle *o, const float *Q, uint32_t B,
                              uint32_t k, uint32_t *ids, float *sims,
                              uint32_t *n_out, uint32_t threads,
                              hnsw_oracle_counters *ctrs);

/* Tie census of B searches (one thread): how many decisions of search_level met EQUAL similarities of two
 * different nodes -- the only places where the reference's sim-only order (core.rs:292-300, :635, :657) and the
 * oracle's (sim, id) order can part.  out[0] queries, [1] stop-test ties, [2] accept-test ties (W full),
 * [3] queries with either, [4] queries whose k + 1 nearest hold equal sims, [5] queries with any of these.      */
void hnsw_oracle_tie_census(const hnsw_oracle *o, const float *Q, uint32_t B, uint32_t k, uint64_t out[6]);

/* HNSW.SEARCH in the Rust binary's own tie order: the reference's sim-only comparisons (core.rs:635, :657) on
 * std::collections::BinaryHeap restated (sift_up / sift_down_to_bottom); what hnsw_oracle_search answers whenever
 * no decision ties (hnsw_oracle_tie_census), and what the reference's binary answers when one does.              */
uint32_t hnsw_oracle_search_std_heap(const hnsw_oracle *o, const float *q, uint32_t k, uint32_t *ids, float *sims);

/* HNSW.NODE.ADD in the Rust binary's own tie order (core.rs:489-599 on std's BinaryHeap restated, sim-only comparisons
 * at :635, :657, :733): what the reference's binary links once a decision meets equal similarities.  ties (may be NULL)
 * accumulates the decisions of this insert that met a tie: [0] stop test, [1] accept test with W full, [2] a
 * select_neighbors cut, [3] equal similarities where only an ORDER is decided -- inside a selection (which of two equal
 * neighbours is linked / shrunk / appended first: the stored order of rows) or W's two nearest (the next layer's entry
 * point).  An insert with all four at zero links the same rows in the same order under ANY heap (tests/test_golden_cpu.py:
 * a build that takes the total-order insert for those and this one for the rest IS this build).  ties has 4 entries.
 * Same levels / first-node rules as hnsw_oracle_add; no touched list.                                                */
int64_t hnsw_oracle_add_std_heap(hnsw_oracle *o, const float *v, int32_t level, uint64_t *ties);
/* Tie census of the LAST hnsw_oracle_add (the total-order build): the same four counts for that insert.              */
void hnsw_oracle_last_add_ties(const hnsw_oracle *o, uint64_t out[4]);

/* core.rs:414-475 + 824-863 (HNSW.NODE.DEL).  Ids are never reused; node_count() keeps counting
 * allocated ids, live_count() is the reference's node_count.  The new enterpoint, which the
 * reference picks arbitrarily from the highest non-empty layer (HashSet order, core.rs:453), is the
 * smallest id of that layer.  Returns 0, -1 if id is not a live node.                          */
int hnsw_oracle_delete(hnsw_oracle *o, uint32_t id, uint32_t *touched, uint32_t touched_cap,
                       uint32_t *n_touched);
/* the same in the Rust binary's own tie order (std BinaryHeap restated; hnsw_oracle_add_std_heap's counterpart) */
int hnsw_oracle_delete_std_heap(hnsw_oracle *o, uint32_t id, uint32_t *touched, uint32_t touched_cap, uint32_t *n_touched);
uint32_t hnsw_oracle_live_count(const hnsw_oracle *o);
int hnsw_oracle_is_live(const hnsw_oracle *o, uint32_t id);

/* ---- introspection / bulk transfer --------------------------------------- */
uint32_t hnsw_oracle_node_count(const hnsw_oracle *o);
uint32_t hnsw_oracle_max_layer(const hnsw_oracle *o);
int64_t hnsw_oracle_enterpoint(const hnsw_oracle *o); /* -1 if none          */
uint32_t hnsw_oracle_level(const hnsw_oracle *o, uint32_t id);
uint32_t hnsw_oracle_degree(const hnsw_oracle *o, uint32_t id, uint32_t layer);
/* copies min(cap, degree) ids in stored order (core.rs:646 iteration order) */
uint32_t hnsw_oracle_neighbors(const hnsw_oracle *o, uint32_t id, uint32_t layer,
                               uint32_t *out, uint32_t cap);
const float *hnsw_oracle_vector(const hnsw_oracle *o, uint32_t id);
void hnsw_oracle_insert_counters(const hnsw_oracle *o, hnsw_oracle_counters *c);

/* per-layer CSR export: row_ptr has n+1 entries, col has row_ptr[n] entries */
uint64_t hnsw_oracle_layer_nnz(const hnsw_oracle *o, uint32_t layer);
void hnsw_oracle_export_layer(const hnsw_oracle *o, uint32_t layer,
                              uint64_t *row_ptr, uint32_t *col);
void hnsw_oracle_export_levels(const hnsw_oracle *o, uint32_t *levels);

/* rebuild an oracle index from a frozen graph (mirror of make_index,
 * src/lib.rs:252-315): vectors [n][dim], levels [n], per-layer CSR.          */
hnsw_oracle *hnsw_oracle_import(uint32_t dim, uint32_t m, uint32_t ef_construction,
                                uint32_t n, const float *vectors,
                                const uint32_t *levels, int64_t enterpoint,
                                uint32_t n_layers,
                                const uint64_t *const *row_ptr,
                                const uint32_t *const *col);

#ifdef __cplusplus
}
#endif
#endif
