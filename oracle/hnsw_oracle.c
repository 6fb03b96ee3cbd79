/*
 * hnsw_oracle.c -- CPU parity oracle (TEST INFRASTRUCTURE ONLY, see header).
 *
 * A plain-C restatement of zhao-lang/redis_hnsw v0.2.1
 *   src/hnsw/metrics.rs   (negated squared-L2, scalar + AVX2/FMA orders)
 *   src/hnsw/core.rs      (Index::new, add_node/insert, search_level,
 *                          select_neighbors, connect_neighbors,
 *                          update_node_connections, search_knn)
 * Every function cites the reference lines it follows.  Nothing here is used
 * by the shipped library.
 *
 * Build: gcc -O3 -ffp-contract=off -fPIC -shared -pthread (see Makefile).
 * -ffp-contract=off matters: Rust never contracts a*b+c, so the scalar fold
 * must stay mul-then-add; the AVX order uses explicit FMA only where the
 * reference does (metrics.rs:57,60,64,68).
 */
#include "hnsw_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#if defined(__x86_64__) || defined(__i386__)
#include <immintrin.h>
#define ORACLE_X86 1
#endif

/* ========================================================================= */
/* metrics.rs                                                                */
/* ========================================================================= */

/* metrics.rs:79-84: -fold(0.0, acc + (x-y)*(x-y)), strictly left to right.   */
float hnsw_oracle_sim_scalar(const float *a, const float *b, size_t n)
{
    float acc = 0.0f;
    for (size_t i = 0; i < n; i++) {
        float d = a[i] - b[i];
        float sq = d * d;
        acc = acc + sq;
    }
    return -acc;
}

/* Lane-exact emulation of metrics.rs:48-77.  Virtual lane (acc, j) handles
 * elements 32*t + 8*acc + j, t ascending, with one fused multiply-add each
 * (metrics.rs:55-69); then (e1+e2)+(e3+e4) per lane (metrics.rs:71-74);
 * hsum256: low128 + high128 (metrics.rs:37-39); hsum_ps_sse3:
 * (s0+s1) + (s2+s3) (metrics.rs:27-31); negate (metrics.rs:75).             */
float hnsw_oracle_sim_avx_emulated(const float *a, const float *b, size_t n)
{
    float e[4][8];
    memset(e, 0, sizeof e);
    for (size_t i = 0; i + 32 <= n; i += 32)
        for (int acc = 0; acc < 4; acc++)
            for (int j = 0; j < 8; j++) {
                float d = a[i + 8 * acc + j] - b[i + 8 * acc + j];
                e[acc][j] = fmaf(d, d, e[acc][j]);
            }
    float v[8], s[4];
    for (int j = 0; j < 8; j++)
        v[j] = (e[0][j] + e[1][j]) + (e[2][j] + e[3][j]);
    for (int j = 0; j < 4; j++)
        s[j] = v[j] + v[j + 4];
    float res = (s[0] + s[1]) + (s[2] + s[3]);
    return -res;
}

#ifdef ORACLE_X86
/* The same computation with the reference's own intrinsics, statement for
 * statement what metrics.rs:25-77 executes on an AVX2 host.                 */
__attribute__((target("avx2,fma"))) static float sim_avx2_hw(const float *a,
                                                             const float *b,
                                                             size_t n)
{
    __m256 e1 = _mm256_setzero_ps(), e2 = _mm256_setzero_ps();
    __m256 e3 = _mm256_setzero_ps(), e4 = _mm256_setzero_ps();
    for (size_t i = 0; i + 32 <= n; i += 32) {
        __m256 v1 = _mm256_sub_ps(_mm256_loadu_ps(a + i), _mm256_loadu_ps(b + i));
        e1 = _mm256_fmadd_ps(v1, v1, e1);
        __m256 v2 = _mm256_sub_ps(_mm256_loadu_ps(a + i + 8), _mm256_loadu_ps(b + i + 8));
        e2 = _mm256_fmadd_ps(v2, v2, e2);
        __m256 v3 = _mm256_sub_ps(_mm256_loadu_ps(a + i + 16), _mm256_loadu_ps(b + i + 16));
        e3 = _mm256_fmadd_ps(v3, v3, e3);
        __m256 v4 = _mm256_sub_ps(_mm256_loadu_ps(a + i + 24), _mm256_loadu_ps(b + i + 24));
        e4 = _mm256_fmadd_ps(v4, v4, e4);
    }
    __m256 t = _mm256_add_ps(_mm256_add_ps(e1, e2), _mm256_add_ps(e3, e4));
    __m128 lo = _mm256_castps256_ps128(t);
    __m128 hi = _mm256_extractf128_ps(t, 1);
    lo = _mm_add_ps(lo, hi);
    __m128 shuf = _mm_movehdup_ps(lo);
    __m128 sums = _mm_add_ps(lo, shuf);
    shuf = _mm_movehl_ps(shuf, sums);
    sums = _mm_add_ss(sums, shuf);
    return -_mm_cvtss_f32(sums);
}
static int g_have_avx2 = -1;
#endif

float hnsw_oracle_sim_avx(const float *a, const float *b, size_t n)
{
#ifdef ORACLE_X86
    if (g_have_avx2 < 0)
        g_have_avx2 = __builtin_cpu_supports("avx2") && __builtin_cpu_supports("fma");
    if (g_have_avx2)
        return sim_avx2_hw(a, b, n);
#endif
    return hnsw_oracle_sim_avx_emulated(a, b, n);
}

/* metrics.rs:14-23.  The reference takes the AVX branch iff the host has avx2
 * and len % 32 == 0; the oracle models an AVX2 host (every x86-64 server
 * since 2013) so the summation ORDER is host independent.                   */
float hnsw_oracle_euclidean(const float *a, const float *b, size_t n)
{
    if (n % 32 == 0)
        return hnsw_oracle_sim_avx(a, b, n);
    return hnsw_oracle_sim_scalar(a, b, n);
}

/* ========================================================================= */
/* core.rs data model                                                        */
/* ========================================================================= */

typedef struct { float sim; uint32_t id; } simpair;      /* core.rs:233-249  */

/* core.rs:292-300 orders SimPair by sim only; ties are unspecified in the
 * reference.  Oracle total order: larger sim, then smaller id, is "nearer". */
static inline int nearer(simpair a, simpair b)
{
    return a.sim > b.sim || (a.sim == b.sim && a.id < b.id);
}

/* "Reference-strict ties" (a test switch, off by default).  The reference compares on sim ALONE at
 * core.rs:635 (stop), :657 (accept) and :733 (select); the oracle's default applies its (sim, id) total
 * order there too.  With the switch on those three tests use sim only, exactly as written in the
 * reference, and the id order is left to the heaps (where Rust leaves it unspecified).  The two variants
 * can only differ when similarities tie; tests/test_oracle_kat.py shows whether they do.            */
static int g_strict_ties = 0;
void hnsw_oracle_set_strict_ties(int on) { g_strict_ties = on; }
static inline int stop_test(simpair c, simpair f)      /* core.rs:635  c.sim < f.sim */
{
    return g_strict_ties ? c.sim < f.sim : nearer(f, c);
}
static inline int accept_test(simpair e, simpair f)    /* core.rs:657  esim > f.sim  */
{
    return g_strict_ties ? e.sim > f.sim : nearer(e, f);
}

/* binary heap; top = nearest (BinaryHeap<SimPair>) or top = furthest
 * (BinaryHeap<Reverse<SimPair>>), core.rs:625-628                            */
typedef struct { simpair *a; uint32_t n, cap; int furthest_top; } heap;

static inline int heap_above(const heap *h, simpair x, simpair y)
{
    return h->furthest_top ? nearer(y, x) : nearer(x, y);
}
static void heap_init(heap *h, int furthest_top)
{
    h->a = NULL; h->n = 0; h->cap = 0; h->furthest_top = furthest_top;
}
static void heap_free(heap *h) { free(h->a); h->a = NULL; h->n = h->cap = 0; }
static void heap_clear(heap *h) { h->n = 0; }
static void heap_push(heap *h, simpair x)
{
    if (h->n == h->cap) {
        h->cap = h->cap ? h->cap * 2 : 64;
        h->a = (simpair *)realloc(h->a, (size_t)h->cap * sizeof(simpair));
    }
    uint32_t i = h->n++;
    while (i > 0) {
        uint32_t p = (i - 1) / 2;
        if (!heap_above(h, x, h->a[p])) break;
        h->a[i] = h->a[p];
        i = p;
    }
    h->a[i] = x;
}
static inline simpair heap_peek(const heap *h) { return h->a[0]; }
static simpair heap_pop(heap *h)
{
    simpair top = h->a[0];
    simpair x = h->a[--h->n];
    uint32_t i = 0;
    for (;;) {
        uint32_t c = 2 * i + 1;
        if (c >= h->n) break;
        if (c + 1 < h->n && heap_above(h, h->a[c + 1], h->a[c])) c++;
        if (!heap_above(h, h->a[c], x)) break;
        h->a[i] = h->a[c];
        i = c;
    }
    if (h->n) h->a[i] = x;
    return top;
}
static void heap_copy(heap *dst, const heap *src, int furthest_top)
{
    heap_clear(dst);
    dst->furthest_top = furthest_top;
    for (uint32_t i = 0; i < src->n; i++) heap_push(dst, src->a[i]);
}

/* core.rs:96-100 `neighbors: Vec<Vec<NodeWeak>>`                             */
typedef struct { uint32_t *ids; uint32_t n, cap; } nrow;
typedef struct { uint32_t level; nrow *rows; /* [level+1] */ } onode;

/* per-thread scratch: visited stamps (HashSet v, core.rs:614,692) + heaps   */
typedef struct {
    uint32_t *stamp; uint32_t stamp_cap; uint32_t epoch;
    heap C, W, res, w2, wd, r, ccopy, econn, enew, nbrs;
    /* tie census (hnsw_oracle_tie_census): decisions of search_level that met EQUAL similarities of two
     * different nodes -- the only places where the reference's sim-only order (core.rs:292-300) and this
     * file's (sim, id) order can part */
    uint64_t tie_stop, tie_accept, tie_select;
    float evicted_max; int have_evicted;   /* census: the most similar entry W has evicted in this search_level */
    uint64_t tie_order;   /* equal similarities where only an ORDER is decided: inside a selection (link / shrink / append order), W's two nearest (entry point) */
} scratch;

struct hnsw_oracle {                                   /* core.rs:303-319  */
    uint32_t dim, m, m_max, m_max0, ef_construction;
    double level_mult;
    uint32_t node_count, max_layer;
    int64_t enterpoint;
    float *data; uint32_t cap;
    onode *nodes;
    uint8_t *dead;                                     /* tombstones: ids are never reused      */
    uint32_t n_dead;
    uint64_t rng[4];
    scratch sc;
    uint32_t *touch; uint32_t n_touch, touch_cap;      /* `updated` sets    */
    uint32_t *touch_stamp; uint32_t touch_epoch, touch_stamp_cap;
    hnsw_oracle_counters ins;
};

static void scratch_init(scratch *s)
{
    memset(s, 0, sizeof *s);
    heap_init(&s->C, 0); heap_init(&s->W, 1); heap_init(&s->res, 0);
    heap_init(&s->w2, 0); heap_init(&s->wd, 0); heap_init(&s->r, 0);
    heap_init(&s->ccopy, 0); heap_init(&s->econn, 0); heap_init(&s->enew, 0);
    heap_init(&s->nbrs, 0);
}
static void scratch_free(scratch *s)
{
    free(s->stamp);
    heap_free(&s->C); heap_free(&s->W); heap_free(&s->res); heap_free(&s->w2);
    heap_free(&s->wd); heap_free(&s->r); heap_free(&s->ccopy);
    heap_free(&s->econn); heap_free(&s->enew); heap_free(&s->nbrs);
}
/* a fresh HashSet: bump the epoch, grow the stamp array if the index grew   */
static void visited_reset(scratch *s, uint32_t n)
{
    if (s->stamp_cap < n) {
        uint32_t nc = s->stamp_cap ? s->stamp_cap : 1024;
        while (nc < n) nc *= 2;
        s->stamp = (uint32_t *)realloc(s->stamp, (size_t)nc * 4);
        memset(s->stamp + s->stamp_cap, 0, (size_t)(nc - s->stamp_cap) * 4);
        s->stamp_cap = nc;
    }
    if (++s->epoch == 0) { memset(s->stamp, 0, (size_t)s->stamp_cap * 4); s->epoch = 1; }
}
static inline int visited_test_and_set(scratch *s, uint32_t id)
{
    if (s->stamp[id] == s->epoch) return 1;
    s->stamp[id] = s->epoch;
    return 0;
}
static inline int visited_test(const scratch *s, uint32_t id) { return s->stamp[id] == s->epoch; }

static inline const float *vec(const hnsw_oracle *o, uint32_t id)
{
    return o->data + (size_t)id * o->dim;
}
/* rows above a node's top level behave as empty (push_levels, core.rs:127-135,642) */
static inline const nrow *row_of(const hnsw_oracle *o, uint32_t id, uint32_t level)
{
    static const nrow empty = { NULL, 0, 0 };
    const onode *nd = &o->nodes[id];
    return level <= nd->level ? &nd->rows[level] : &empty;
}

/* core.rs:137-143 add_neighbor: push iff not already present                 */
static void add_neighbor(hnsw_oracle *o, uint32_t id, uint32_t level, uint32_t nb)
{
    onode *nd = &o->nodes[id];
    if (level > nd->level) {
        /* push_levels (core.rs:127-135) would grow the rows here.  It cannot
         * happen: a node is only ever linked at layers <= its own top level
         * (the searches that find it start from nodes of a higher layer).     */
        fprintf(stderr, "hnsw_oracle: add_neighbor above a node's level (%u > %u)\n", level, nd->level);
        abort();
    }
    nrow *r = &nd->rows[level];
    for (uint32_t i = 0; i < r->n; i++) if (r->ids[i] == nb) return;
    if (r->n == r->cap) {
        r->cap = r->cap ? r->cap * 2 : (level == 0 ? o->m_max0 : o->m_max) + 1;
        r->ids = (uint32_t *)realloc(r->ids, (size_t)r->cap * 4);
    }
    r->ids[r->n++] = nb;
}
/* core.rs:145-152 rm_neighbor: position().unwrap() then Vec::remove           */
static void rm_neighbor(hnsw_oracle *o, uint32_t id, uint32_t level, uint32_t nb)
{
    nrow *r = &o->nodes[id].rows[level];
    for (uint32_t i = 0; i < r->n; i++)
        if (r->ids[i] == nb) {
            memmove(r->ids + i, r->ids + i + 1, (size_t)(r->n - i - 1) * 4);
            r->n--;
            return;
        }
    fprintf(stderr, "hnsw_oracle: rm_neighbor(%u, L%u, %u): not a neighbour "
                    "(the reference panics here, core.rs:150)\n", id, level, nb);
    abort();
}

static void touch_reset(hnsw_oracle *o)
{
    if (o->touch_stamp_cap < o->cap) {
        o->touch_stamp = (uint32_t *)realloc(o->touch_stamp, (size_t)o->cap * 4);
        memset(o->touch_stamp + o->touch_stamp_cap, 0, (size_t)(o->cap - o->touch_stamp_cap) * 4);
        o->touch_stamp_cap = o->cap;
    }
    if (++o->touch_epoch == 0) { memset(o->touch_stamp, 0, (size_t)o->touch_stamp_cap * 4); o->touch_epoch = 1; }
    o->n_touch = 0;
}
static void touch_add(hnsw_oracle *o, uint32_t id)
{
    if (o->touch_stamp[id] == o->touch_epoch) return;
    o->touch_stamp[id] = o->touch_epoch;
    if (o->n_touch == o->touch_cap) {
        o->touch_cap = o->touch_cap ? o->touch_cap * 2 : 256;
        o->touch = (uint32_t *)realloc(o->touch, (size_t)o->touch_cap * 4);
    }
    o->touch[o->n_touch++] = id;
}

/* ========================================================================= */
/* core.rs:607-675 search_level                                              */
/* ========================================================================= */
/* Leaves W (furthest-top heap, <= ef pairs) in s->W.                         */
static void search_level(const hnsw_oracle *o, scratch *s, const float *query,
                         uint32_t ep, uint32_t ef, uint32_t level,
                         hnsw_oracle_counters *ct)
{
    visited_reset(s, o->node_count);                    /* :614 */
    visited_test_and_set(s, ep);                        /* :617 */
    simpair qpair = { hnsw_oracle_euclidean(query, vec(o, ep), o->dim), ep }; /* :621 */
    ct->n_dist++;
    heap *C = &s->C, *W = &s->W;
    heap_clear(C); heap_clear(W);
    heap_push(C, qpair);                                /* :627 */
    heap_push(W, qpair);                                /* :628 */
    s->have_evicted = 0;

    while (C->n) {                                      /* :630 */
        simpair c = heap_pop(C);                        /* :631 nearest        */
        simpair f = heap_peek(W);                       /* :632 furthest       */
        if (c.sim == f.sim && c.id != f.id) s->tie_stop++;   /* census: :635 decided by something other than sim */
        if (stop_test(c, f)) break;                     /* :635 c.sim < f.sim  */
        if (C->n && heap_peek(C).sim == c.sim) s->tie_order++;   /* census: the pop itself (:631) chose between equal candidates */
        ct->n_expand++;
        const nrow *nb = row_of(o, c.id, level);        /* :642-645            */
        for (uint32_t i = 0; i < nb->n; i++) {          /* :646 stored order   */
            uint32_t e = nb->ids[i];
            ct->n_ids++;
            if (visited_test_and_set(s, e)) continue;   /* :648-649            */
            f = heap_peek(W);                           /* :651                */
            simpair ep2 = { hnsw_oracle_euclidean(query, vec(o, e), o->dim), e }; /* :652 */
            ct->n_dist++;
            if (W->n >= ef && ep2.sim == f.sim) s->tie_accept++;  /* census: :657 decided by something other than sim */
            if (accept_test(ep2, f) || W->n < ef) {     /* :657                */
                heap_push(C, ep2);                      /* :659                */
                heap_push(W, ep2);                      /* :660                */
                if (W->n > ef) {                        /* :662-664: which of two equal furthest entries goes is the heap's choice; it   */
                    simpair out = heap_pop(W);          /* matters if the other one is still W's furthest when the search ends (census below) */
                    if (!s->have_evicted || out.sim > s->evicted_max) { s->evicted_max = out.sim; s->have_evicted = 1; }
                }
            }
        }
    }
    /* census: the most similar entry W ever evicted is as similar as W's furthest now: which of the two stayed was the heap's choice */
    if (s->have_evicted && W->n && heap_peek(W).sim == s->evicted_max) s->tie_order++;
}

/* nearest member of s->W (core.rs:670-674 rebuilds a nearest-top heap; :514,
 * :576, :872 then take its top)                                              */
static simpair nearest_of_W(const scratch *s)
{
    simpair best = s->W.a[0];
    for (uint32_t i = 1; i < s->W.n; i++) if (nearer(s->W.a[i], best)) best = s->W.a[i];
    /* census: two nearest members with equal similarities -- which one is the next layer's entry point is the heap's choice */
    for (uint32_t i = 0; i < s->W.n; i++) if (s->W.a[i].sim == best.sim && s->W.a[i].id != best.id) { ((scratch *)s)->tie_order++; break; }
    return best;
}

/* ========================================================================= */
/* core.rs:677-757 select_neighbors (extend_candidates = keep_pruned = true   */
/* at every call site: :528-529, :565-566, :850-851)                          */
/* ========================================================================= */
/* c: nearest-top heap of candidates.  Result: nearest-top heap in *r.         */
static void select_neighbors(hnsw_oracle *o, scratch *s, uint32_t query,
                             const heap *c, uint32_t m, uint32_t lc,
                             int64_t ignored, heap *r, hnsw_oracle_counters *ct)
{
    heap *w = &s->w2, *wd = &s->wd, *ccopy = &s->ccopy;
    heap_clear(r); r->furthest_top = 0;                 /* :684 */
    heap_copy(w, c, 0);                                 /* :685 */
    heap_clear(wd); wd->furthest_top = 0;               /* :686 */

    /* :689-722 extend candidates by their neighbours */
    visited_reset(s, o->node_count);                    /* :692 */
    for (uint32_t i = 0; i < c->n; i++) visited_test_and_set(s, c->a[i].id); /* :693-696 */
    heap_copy(ccopy, c, 0);                             /* :698 */
    const float *qv = vec(o, query);
    while (ccopy->n) {                                  /* :699 */
        simpair e = heap_pop(ccopy);                    /* :700 nearest first  */
        const nrow *nb = row_of(o, e.id, lc);
        for (uint32_t i = 0; i < nb->n; i++) {          /* :702 */
            uint32_t en = nb->ids[i];
            ct->n_ids++;
            if (en == query || (ignored >= 0 && en == (uint32_t)ignored)) continue; /* :704-708 */
            if (!visited_test(s, en)) {                 /* :710 */
                simpair p = { hnsw_oracle_euclidean(qv, vec(o, en), o->dim), en }; /* :711 */
                ct->n_dist++;
                heap_push(w, p);                        /* :717 */
                visited_test_and_set(s, en);            /* :718 */
            }
        }
    }

    /* :724-738 */
    while (w->n && r->n < m) {
        simpair e = heap_pop(w);
        if (e.id == query || (ignored >= 0 && e.id == (uint32_t)ignored)) continue; /* :728-731 */
        /* :733  `enr.sim > r.peek().sim`: r is nearest-top, so after the first
         * element this never holds on sim; on the (sim,id) key it cannot hold
         * either because w pops in key order.                                 */
        if (r->n == 0 || accept_test(e, heap_peek(r))) heap_push(r, e);
        else heap_push(wd, e);
    }
    /* :741-754 keep_pruned_connections */
    while (wd->n && r->n < m) {
        simpair p = heap_pop(wd);
        if (p.id == query || (ignored >= 0 && p.id == (uint32_t)ignored)) continue;
        heap_push(r, p);
    }
    /* census: equal similarities INSIDE the selection: connect_neighbors (:765-772), the shrink loop (:540-541) and
     * update_node_connections (:790-796) pop it nearest first, so which of two equal ones is linked / shrunk / appended first
     * -- the stored order of rows -- is the heap's choice */
    for (uint32_t i = 0; i < r->n; i++)
        for (uint32_t j = i + 1; j < r->n; j++)
            if (r->a[i].sim == r->a[j].sim) { s->tie_order++; i = r->n; break; }
    if (r->n == m && m) {                               /* census: equal similarities across the cut (:733 / :741) */
        float worst = r->a[0].sim;
        for (uint32_t i = 1; i < r->n; i++) if (r->a[i].sim < worst) worst = r->a[i].sim;
        int tie = 0;
        for (uint32_t i = 0; i < wd->n && !tie; i++) tie = wd->a[i].sim == worst && wd->a[i].id != query && !(ignored >= 0 && wd->a[i].id == (uint32_t)ignored);
        for (uint32_t i = 0; i < w->n && !tie; i++) tie = w->a[i].sim == worst && w->a[i].id != query && !(ignored >= 0 && w->a[i].id == (uint32_t)ignored);
        s->tie_select += (uint64_t)tie;
    }
}

/* core.rs:759-774 */
static void connect_neighbors(hnsw_oracle *o, scratch *s, uint32_t query,
                              const heap *neighbors, uint32_t level)
{
    heap *t = &s->ccopy;
    heap_copy(t, neighbors, 0);
    while (t->n) {
        simpair n = heap_pop(t);                        /* nearest first        */
        add_neighbor(o, query, level, n.id);            /* :770 */
        add_neighbor(o, n.id, level, query);            /* :771-772 */
    }
}

/* core.rs:776-822; `updated` goes to the touched set                          */
static void update_node_connections(hnsw_oracle *o, scratch *s, uint32_t node,
                                    const heap *new_neighbors,
                                    const heap *old_neighbors, uint32_t level,
                                    int64_t ignored)
{
    heap *newconn = &s->ccopy;
    heap_copy(newconn, new_neighbors, 0);               /* :784 */
    uint32_t n_rm = old_neighbors->n;                   /* :785 into_vec        */
    simpair *rmconn = (simpair *)malloc((size_t)(n_rm ? n_rm : 1) * sizeof(simpair));
    memcpy(rmconn, old_neighbors->a, (size_t)n_rm * sizeof(simpair));
    touch_add(o, node);                                 /* :787 */

    while (newconn->n) {                                /* :790 */
        simpair np = heap_pop(newconn);
        add_neighbor(o, node, level, np.id);            /* :793 */
        add_neighbor(o, np.id, level, node);            /* :794-795 */
        touch_add(o, np.id);                            /* :796 */
        for (uint32_t i = 0; i < n_rm; i++)             /* :799-801 */
            if (rmconn[i].id == np.id) {
                memmove(rmconn + i, rmconn + i + 1, (size_t)(n_rm - i - 1) * sizeof(simpair));
                n_rm--;
                break;
            }
    }
    while (n_rm) {                                      /* :805 */
        simpair rp = rmconn[--n_rm];                    /* :806 Vec::pop        */
        rm_neighbor(o, node, level, rp.id);             /* :808 */
        if (ignored >= 0 && rp.id == (uint32_t)ignored) continue; /* :810-813 */
        rm_neighbor(o, rp.id, level, node);             /* :815 */
        touch_add(o, rp.id);                            /* :816 */
    }
    free(rmconn);
}

/* ========================================================================= */
/* core.rs:322-347 Index::new                                                 */
/* ========================================================================= */
static uint64_t splitmix64(uint64_t *x)
{
    uint64_t z = (*x += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
static inline uint64_t rotl64(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }
static uint64_t xoshiro_next(uint64_t *s)
{
    uint64_t result = rotl64(s[1] * 5, 7) * 9, t = s[1] << 17;
    s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3]; s[2] ^= t; s[3] = rotl64(s[3], 45);
    return result;
}

hnsw_oracle *hnsw_oracle_new(uint32_t dim, uint32_t m, uint32_t ef_construction, uint64_t seed)
{
    hnsw_oracle *o = (hnsw_oracle *)calloc(1, sizeof *o);
    o->dim = dim;
    o->m = m;
    o->m_max = m;                                       /* :335 */
    o->m_max0 = m * 2;                                  /* :336 */
    o->ef_construction = ef_construction;               /* :337 */
    o->level_mult = 1.0 / log(1.0 * (double)m);         /* :338 */
    o->node_count = 0;
    o->max_layer = 0;
    o->enterpoint = -1;                                 /* :343 None */
    uint64_t x = seed;
    for (int i = 0; i < 4; i++) o->rng[i] = splitmix64(&x);
    scratch_init(&o->sc);
    return o;
}

void hnsw_oracle_free(hnsw_oracle *o)
{
    if (!o) return;
    for (uint32_t i = 0; i < o->node_count; i++) {
        onode *nd = &o->nodes[i];
        if (nd->rows) {
            for (uint32_t l = 0; l <= nd->level; l++) free(nd->rows[l].ids);
            free(nd->rows);
        }
    }
    free(o->nodes); free(o->data); free(o->touch); free(o->touch_stamp); free(o->dead);
    scratch_free(&o->sc);
    free(o);
}

/* core.rs:601-605: r ~ U[0,1); (-ln(r) * level_mult) as usize.  r == 0 gives
 * +inf which Rust saturates; the oracle caps the level at 31 instead.        */
static uint32_t gen_random_level(hnsw_oracle *o)
{
    double r = (double)(xoshiro_next(o->rng) >> 11) * (1.0 / 9007199254740992.0);
    double l = -log(r) * o->level_mult;
    if (!(l < 31.0)) return 31;
    return (uint32_t)l;
}

static void ensure_cap(hnsw_oracle *o)
{
    if (o->node_count < o->cap) return;
    o->cap = o->cap ? o->cap * 2 : 1024;
    o->data = (float *)realloc(o->data, (size_t)o->cap * o->dim * sizeof(float));
    o->nodes = (onode *)realloc(o->nodes, (size_t)o->cap * sizeof(onode));
    o->dead = (uint8_t *)realloc(o->dead, (size_t)o->cap);
    memset(o->dead + o->node_count, 0, (size_t)(o->cap - o->node_count));
}

static uint32_t store_node(hnsw_oracle *o, const float *v, uint32_t level)
{
    ensure_cap(o);
    uint32_t id = o->node_count++;
    memcpy(o->data + (size_t)id * o->dim, v, (size_t)o->dim * sizeof(float));
    o->nodes[id].level = level;
    o->nodes[id].rows = (nrow *)calloc(level + 1, sizeof(nrow));
    return id;
}

/* core.rs:489-599 insert                                                     */
static void insert(hnsw_oracle *o, const float *data, uint32_t l)
{
    scratch *s = &o->sc;
    hnsw_oracle_counters *ct = &o->ins;
    uint32_t l_max = o->max_layer;                      /* :496 */
    uint32_t query = store_node(o, data, l);            /* :498-507 */
    const float *qv = vec(o, query);
    uint32_t ep = (uint32_t)o->enterpoint;              /* :508 */

    uint32_t lc = l_max;                                /* :511 */
    while (lc > l) {                                    /* :512 */
        search_level(o, s, qv, ep, 1, lc, ct);          /* :513 */
        ep = nearest_of_W(s).id;                        /* :514 */
        if (lc == 0) break;
        lc--;
    }

    uint32_t top = l_max < l ? l_max : l;
    for (uint32_t lcc = top + 1; lcc-- > 0;) {          /* :523 */
        search_level(o, s, qv, ep, o->ef_construction, lcc, ct); /* :524 */
        heap_copy(&s->res, &s->W, 0);                   /* :670-674 */
        simpair w_nearest = heap_peek(&s->res);
        select_neighbors(o, s, query, &s->res, o->m, lcc, -1, &s->nbrs, ct); /* :525-531 */
        connect_neighbors(o, s, query, &s->nbrs, lcc);  /* :532 */
        for (uint32_t i = 0; i < s->nbrs.n; i++) touch_add(o, s->nbrs.a[i].id); /* :535-537 */

        while (s->nbrs.n) {                             /* :540 */
            simpair e = heap_pop(&s->nbrs);             /* :541 nearest first   */
            heap *econn = &s->econn;                    /* :544-558 */
            heap_clear(econn); econn->furthest_top = 0;
            {
                const nrow *er = row_of(o, e.id, lcc);
                const float *ev = vec(o, e.id);
                for (uint32_t i = 0; i < er->n; i++) {
                    simpair p = { hnsw_oracle_euclidean(ev, vec(o, er->ids[i]), o->dim), er->ids[i] }; /* :550 */
                    ct->n_dist++;
                    ct->n_ids++;
                    heap_push(econn, p);
                }
            }
            uint32_t m_max = lcc == 0 ? o->m_max0 : o->m_max; /* :560 */
            if (econn->n > m_max) {                     /* :561 */
                select_neighbors(o, s, e.id, econn, m_max, lcc, -1, &s->enew, ct); /* :568 */
                update_node_connections(o, s, e.id, &s->enew, econn, lcc, -1);     /* :569 */
            }
        }
        ep = w_nearest.id;                              /* :576 */
    }

    if (l > l_max) {                                    /* :587-593 */
        o->max_layer = l;
        o->enterpoint = query;
    }
}

/* core.rs:383-412 add_node (names/duplicate check live with the caller)      */
int64_t hnsw_oracle_add(hnsw_oracle *o, const float *v, int32_t level,
                        uint32_t *touched, uint32_t touched_cap, uint32_t *n_touched)
{
    if (n_touched) *n_touched = 0;
    if (o->node_count - o->n_dead == 0) {               /* :393-405 (node_count is the live count there) */
        uint32_t id = store_node(o, v, 0);
        o->enterpoint = id;
        return id;
    }
    uint32_t l = level >= 0 ? (uint32_t)level : gen_random_level(o); /* :495 */
    uint32_t id = o->node_count;
    ensure_cap(o);                                      /* before touch_reset sizes its stamps */
    touch_reset(o);
    o->sc.tie_stop = o->sc.tie_accept = o->sc.tie_select = o->sc.tie_order = 0;   /* census of this insert (hnsw_oracle_last_add_ties) */
    insert(o, v, l);
    if (touched) {
        uint32_t n = o->n_touch < touched_cap ? o->n_touch : touched_cap;
        memcpy(touched, o->touch, (size_t)n * 4);
    }
    if (n_touched) *n_touched = o->n_touch;
    return id;
}

/* ========================================================================= */
/* core.rs:477-486, 865-892 search_knn                                        */
/* ========================================================================= */
static int cmp_nearer(const void *a, const void *b)
{
    simpair x = *(const simpair *)a, y = *(const simpair *)b;
    return nearer(x, y) ? -1 : nearer(y, x) ? 1 : 0;
}

static uint32_t search_knn_internal(const hnsw_oracle *o, scratch *s, const float *q,
                                    uint32_t k, uint32_t ef, uint32_t *ids,
                                    float *sims, hnsw_oracle_counters *ct)
{
    uint32_t ep = (uint32_t)o->enterpoint;              /* :866 */
    uint32_t lc = o->max_layer;                         /* :867-869 */
    while (lc > 0) {                                    /* :870 */
        search_level(o, s, q, ep, 1, lc, ct);           /* :871 */
        ep = nearest_of_W(s).id;                        /* :872 */
        lc--;
    }
    search_level(o, s, q, ep, ef, 0, ct);               /* :876 */
    /* :878-890 pop nearest-first until k or empty */
    qsort(s->W.a, s->W.n, sizeof(simpair), cmp_nearer);
    uint32_t n = s->W.n < k ? s->W.n : k;
    for (uint32_t i = 0; i < n; i++) { ids[i] = s->W.a[i].id; sims[i] = s->W.a[i].sim; }
    return n;
}

uint32_t hnsw_oracle_search(const hnsw_oracle *o, const float *q, uint32_t k,
                            uint32_t *ids, float *sims, hnsw_oracle_counters *ctrs)
{
    hnsw_oracle_counters local = { 0, 0, 0 };
    if (o->enterpoint < 0 || o->node_count - o->n_dead == 0) { /* :481-483 */
        if (ctrs) *ctrs = local;
        return 0;
    }
    uint32_t n = search_knn_internal(o, (scratch *)&o->sc, q, k, o->ef_construction, /* :485 */
                                     ids, sims, &local);
    if (ctrs) *ctrs = local;
    return n;
}

/* Tie census of HNSW.SEARCH (test infrastructure for the parity claim).  The reference orders SimPair by sim
 * alone (core.rs:292-300) and leaves equal sims to std's BinaryHeap; this file and the engine break them by id.
 * The two can only answer differently on a query where a DECISION met equal sims of two different nodes:
 *   the stop test    core.rs:635  c.sim == f.sim  (the reference expands c, the total order may stop),
 *   the accept test  core.rs:657  e.sim == f.sim with W full (the reference rejects e, the total order may accept),
 *   the answer       core.rs:878-890: two of the k + 1 nearest of W have equal sims (which one is returned, or in
 *                    which order, is the heap's business).
 * (Which of two equal furthest members W evicts, :662-664, does not change f.sim and so changes no later decision.)
 * out[0] queries, [1] stop-test ties, [2] accept-test ties, [3] queries with a stop or accept tie,
 * out[4] queries whose k + 1 nearest hold equal sims, [5] queries with any of the three.  One thread.           */
void hnsw_oracle_tie_census(const hnsw_oracle *o, const float *Q, uint32_t B, uint32_t k, uint64_t out[6])
{
    memset(out, 0, 6 * sizeof(uint64_t));
    if (o->enterpoint < 0 || o->node_count - o->n_dead == 0) return;
    scratch *s = (scratch *)&o->sc;
    uint32_t *ids = (uint32_t *)malloc(((size_t)k + 1) * 4);
    float *sims = (float *)malloc(((size_t)k + 1) * 4);
    for (uint32_t b = 0; b < B; b++) {
        hnsw_oracle_counters ct = { 0, 0, 0 };
        s->tie_stop = s->tie_accept = s->tie_order = 0;
        uint32_t n = search_knn_internal(o, s, Q + (size_t)b * o->dim, k + 1, o->ef_construction, ids, sims, &ct);
        int rt = 0;
        for (uint32_t i = 1; i < n; i++) rt |= sims[i] == sims[i - 1];
        out[0]++;
        out[1] += s->tie_stop; out[2] += s->tie_accept;
        out[3] += (s->tie_stop + s->tie_accept) != 0;
        out[4] += rt != 0;
        out[5] += (s->tie_stop + s->tie_accept + s->tie_order) != 0 || rt;   /* (tie_order: two nearest of an upper layer equal: the entry point) */
    }
    free(ids); free(sims);
}

/* ------------------------------------------------------------------------------------------------------------ */
/* HNSW.SEARCH in the RUST BINARY's tie order (test infrastructure).  SimPair compares by sim only (core.rs:292-300)  */
/* and the reference keeps C, W and the result in std::collections::BinaryHeap (core.rs:625-628, :670-674), so which   */
/* of two equal sims pops, is evicted or is returned first is decided by that heap's sift procedures.  They are      */
/* restated here from the standard library's published source (library/alloc/src/collections/binary_heap.rs:         */
/* push = sift_up(0, len-1), stopping at a parent that is >= the element; pop = swap the last element into the root,  */
/* sift_down_to_bottom(0) -- always towards the greater child, the RIGHT one when the two are equal -- then sift_up;   */
/* into_vec = the array as it is), as tests/transcription/hnsw_transcription.py (RustHeap) does independently in      */
/* Python; tests/golden/tiecase_rust_lattice.npz pins one against the other on tie-heavy lattice data.               */
/* ------------------------------------------------------------------------------------------------------------ */
typedef struct { simpair *a; uint32_t n, cap; int reverse; } rheap;
static inline int rh_le(const rheap *h, float x, float y) { return h->reverse ? y <= x : x <= y; }   /* x <= y in the heap's order */
static uint32_t rh_sift_up(rheap *h, uint32_t start, uint32_t pos)
{
    simpair e = h->a[pos];
    while (pos > start) {
        uint32_t parent = (pos - 1) / 2;
        if (rh_le(h, e.sim, h->a[parent].sim)) break;      /* hole.element() <= hole.get(parent) */
        h->a[pos] = h->a[parent];
        pos = parent;
    }
    h->a[pos] = e;
    return pos;
}
static void rh_push(rheap *h, simpair x)
{
    if (h->n == h->cap) { h->cap = h->cap ? h->cap * 2 : 64; h->a = (simpair *)realloc(h->a, (size_t)h->cap * sizeof(simpair)); }
    h->a[h->n++] = x;
    rh_sift_up(h, 0, h->n - 1);
}
static simpair rh_pop(rheap *h)
{
    simpair item = h->a[--h->n];
    if (h->n) {
        simpair t = h->a[0]; h->a[0] = item; item = t;
        const uint32_t end = h->n;
        uint32_t pos = 0, child = 1;
        simpair e = h->a[0];
        while (end >= 2 && child <= end - 2) {             /* sift_down_to_bottom */
            if (rh_le(h, h->a[child].sim, h->a[child + 1].sim)) child++;
            h->a[pos] = h->a[child];
            pos = child;
            child = 2 * pos + 1;
        }
        if (child == end - 1) { h->a[pos] = h->a[child]; pos = child; }
        h->a[pos] = e;
        rh_sift_up(h, 0, pos);
    }
    return item;
}
/* core.rs:607-675 with the reference's own comparisons (:635 c.sim < f.sim, :657 e.sim > f.sim) on std's heaps; res = :670-674 */
static void search_level_std(const hnsw_oracle *o, scratch *s, const float *query, uint32_t ep, uint32_t ef, uint32_t level,
                             rheap *C, rheap *W, rheap *res)
{
    visited_reset(s, o->node_count);
    visited_test_and_set(s, ep);
    simpair qpair = { hnsw_oracle_euclidean(query, vec(o, ep), o->dim), ep };
    C->n = W->n = res->n = 0; C->reverse = 0; W->reverse = 1; res->reverse = 0;
    rh_push(C, qpair); rh_push(W, qpair);
    while (C->n) {
        simpair c = rh_pop(C);
        simpair f = W->a[0];
        if (c.sim < f.sim) break;                           /* :635 */
        const nrow *nb = row_of(o, c.id, level);
        for (uint32_t i = 0; i < nb->n; i++) {
            uint32_t e = nb->ids[i];
            if (visited_test_and_set(s, e)) continue;
            f = W->a[0];
            simpair e2 = { hnsw_oracle_euclidean(query, vec(o, e), o->dim), e };
            if (e2.sim > f.sim || W->n < ef) {              /* :657 */
                rh_push(C, e2); rh_push(W, e2);
                if (W->n > ef) rh_pop(W);
            }
        }
    }
    for (uint32_t i = 0; i < W->n; i++) rh_push(res, W->a[i]);   /* :670-674: into_vec order, pushed one by one */
}
uint32_t hnsw_oracle_search_std_heap(const hnsw_oracle *o, const float *q, uint32_t k, uint32_t *ids, float *sims)
{
    if (o->enterpoint < 0 || o->node_count - o->n_dead == 0) return 0;
    scratch *s = (scratch *)&o->sc;
    rheap C = {0}, W = {0}, res = {0};
    uint32_t ep = (uint32_t)o->enterpoint, lc = o->max_layer;
    while (lc > 0) {                                        /* :869-874 */
        search_level_std(o, s, q, ep, 1, lc, &C, &W, &res);
        ep = res.a[0].id;                                   /* :872 peek */
        lc--;
    }
    search_level_std(o, s, q, ep, o->ef_construction, 0, &C, &W, &res);   /* :876 */
    uint32_t n = 0;
    while (n < k && res.n) { simpair p = rh_pop(&res); ids[n] = p.id; sims[n] = p.sim; n++; }   /* :878-890 */
    free(C.a); free(W.a); free(res.a);
    return n;
}

/* ------------------------------------------------------------------------------------------------------------ */
/* HNSW.NODE.ADD in the RUST BINARY's tie order (test infrastructure): core.rs:489-599 with every heap a std           */
/* BinaryHeap (rheap above) and every comparison the reference's own -- :635 c.sim < f.sim, :657 e.sim > f.sim, :733  */
/* e.sim > r.peek().sim -- so that equal similarities are ordered by the heap's sift procedures exactly as the binary */
/* orders them: BinaryHeap::clone = the array copied as it is (:685, :690, :698, :765, :784), into_vec / into_iter =   */
/* the array in place (:670-673, :785).  Pinned row for row against the transcription's "rust" golden on 20 k x 128     */
/* (tests/golden/transcribed_20k_dim128.npz: 18 decision ties), tests/test_golden_cpu.py.                              */
/* ------------------------------------------------------------------------------------------------------------ */
static void rh_clone(rheap *dst, const rheap *src)
{
    if (dst->cap < src->n + 1) { dst->cap = src->n + 64; dst->a = (simpair *)realloc(dst->a, (size_t)dst->cap * sizeof(simpair)); }
    memcpy(dst->a, src->a, (size_t)src->n * sizeof(simpair));
    dst->n = src->n; dst->reverse = src->reverse;
}
typedef struct { rheap C, W, res, w, wd, ccopy, nbrs, econn, enew, t; } rscratch;
static void rscratch_free(rscratch *z)
{
    free(z->C.a); free(z->W.a); free(z->res.a); free(z->w.a); free(z->wd.a); free(z->ccopy.a);
    free(z->nbrs.a); free(z->econn.a); free(z->enew.a); free(z->t.a);
}
/* core.rs:677-757; c: a nearest-top std heap; result in *r.  Counts a tie when the cut of :733 / :741-754 fell between
 * equal similarities of different nodes (which one is selected is then the heap's choice). */
static void select_neighbors_std(hnsw_oracle *o, scratch *s, rscratch *z, uint32_t query, const rheap *c, uint32_t m, uint32_t lc,
                                 int64_t ignored, rheap *r, hnsw_oracle_counters *ct)
{
    rheap *w = &z->w, *wd = &z->wd, *ccopy = &z->ccopy;
    r->n = 0; r->reverse = 0;                               /* :684 */
    rh_clone(w, c);                                         /* :685 */
    wd->n = 0; wd->reverse = 0;                             /* :686 */
    visited_reset(s, o->node_count);                        /* :692 */
    for (uint32_t i = 0; i < c->n; i++) visited_test_and_set(s, c->a[i].id);   /* :690-696 (a set: the pop order is immaterial) */
    rh_clone(ccopy, c);                                     /* :698 */
    const float *qv = vec(o, query);
    while (ccopy->n) {                                      /* :699 */
        simpair e = rh_pop(ccopy);
        const nrow *nb = row_of(o, e.id, lc);
        for (uint32_t i = 0; i < nb->n; i++) {              /* :702 */
            uint32_t en = nb->ids[i];
            ct->n_ids++;
            if (en == query || (ignored >= 0 && en == (uint32_t)ignored)) continue;   /* :704-708 */
            if (!visited_test(s, en)) {                     /* :710 */
                simpair p = { hnsw_oracle_euclidean(qv, vec(o, en), o->dim), en };
                ct->n_dist++;
                rh_push(w, p);                              /* :717 */
                visited_test_and_set(s, en);                /* :718 */
            }
        }
    }
    while (w->n && r->n < m) {                              /* :724-738 */
        simpair e = rh_pop(w);
        if (e.id == query || (ignored >= 0 && e.id == (uint32_t)ignored)) continue;
        if (r->n == 0 || e.sim > r->a[0].sim) rh_push(r, e);    /* :733 (r.peek() is r's NEAREST: only the first passes) */
        else rh_push(wd, e);
    }
    while (wd->n && r->n < m) {                             /* :741-754 */
        simpair p = rh_pop(wd);
        if (p.id == query || (ignored >= 0 && p.id == (uint32_t)ignored)) continue;
        rh_push(r, p);
    }
    /* census: equal similarities INSIDE the selection: connect_neighbors (:765-772), the shrink loop (:540-541) and
     * update_node_connections (:790-796) pop it nearest first, so which of two equal ones is linked / shrunk / appended first
     * -- the stored order of rows -- is the heap's choice */
    for (uint32_t i = 0; i < r->n; i++)
        for (uint32_t j = i + 1; j < r->n; j++)
            if (r->a[i].sim == r->a[j].sim) { s->tie_order++; i = r->n; break; }
    if (r->n == m && m) {                                   /* census: equal similarities across the cut */
        float worst = r->a[0].sim;
        for (uint32_t i = 1; i < r->n; i++) if (r->a[i].sim < worst) worst = r->a[i].sim;
        int tie = 0;
        for (uint32_t i = 0; i < wd->n && !tie; i++) tie = wd->a[i].sim == worst && wd->a[i].id != query && !(ignored >= 0 && wd->a[i].id == (uint32_t)ignored);
        for (uint32_t i = 0; i < w->n && !tie; i++) tie = w->a[i].sim == worst && w->a[i].id != query && !(ignored >= 0 && w->a[i].id == (uint32_t)ignored);
        s->tie_select += (uint64_t)tie;
    }
}
/* core.rs:776-822 with std heaps (the removal order, :805-806, is result-neutral: Vec::remove keeps the rest in order) */
static void update_node_connections_std(hnsw_oracle *o, rscratch *z, uint32_t node, const rheap *new_neighbors, const rheap *old_neighbors,
                                        uint32_t level, int64_t ignored)
{
    rheap *newconn = &z->t;
    rh_clone(newconn, new_neighbors);                       /* :784 */
    uint32_t n_rm = old_neighbors->n;                       /* :785 into_vec: the array as it is */
    simpair *rmconn = (simpair *)malloc((size_t)(n_rm ? n_rm : 1) * sizeof(simpair));
    memcpy(rmconn, old_neighbors->a, (size_t)n_rm * sizeof(simpair));
    touch_add(o, node);                                     /* :787 */
    while (newconn->n) {                                    /* :790 */
        simpair np = rh_pop(newconn);
        add_neighbor(o, node, level, np.id);                /* :793 */
        add_neighbor(o, np.id, level, node);                /* :794-795 */
        touch_add(o, np.id);
        for (uint32_t i = 0; i < n_rm; i++)                 /* :799-801 */
            if (rmconn[i].id == np.id) { memmove(rmconn + i, rmconn + i + 1, (size_t)(n_rm - i - 1) * sizeof(simpair)); n_rm--; break; }
    }
    while (n_rm) {                                          /* :805 */
        simpair rp = rmconn[--n_rm];
        rm_neighbor(o, node, level, rp.id);                 /* :808 */
        if (ignored >= 0 && rp.id == (uint32_t)ignored) continue;
        rm_neighbor(o, rp.id, level, node);                 /* :815 */
        touch_add(o, rp.id);
    }
    free(rmconn);
}
/* search_level_std with the insert's counters and the tie census of :635 / :657 */
static void search_level_std_ct(const hnsw_oracle *o, scratch *s, const float *query, uint32_t ep, uint32_t ef, uint32_t level,
                                rheap *C, rheap *W, rheap *res, hnsw_oracle_counters *ct)
{
    visited_reset(s, o->node_count);
    visited_test_and_set(s, ep);
    simpair qpair = { hnsw_oracle_euclidean(query, vec(o, ep), o->dim), ep };
    ct->n_dist++;
    C->n = W->n = res->n = 0; C->reverse = 0; W->reverse = 1; res->reverse = 0;
    rh_push(C, qpair); rh_push(W, qpair);
    s->have_evicted = 0;
    while (C->n) {
        simpair c = rh_pop(C);
        simpair f = W->a[0];
        if (c.sim == f.sim && c.id != f.id) s->tie_stop++;
        if (c.sim < f.sim) break;                           /* :635 */
        if (C->n && C->a[0].sim == c.sim) s->tie_order++;   /* census: the pop itself (:631) chose between equal candidates */
        ct->n_expand++;
        const nrow *nb = row_of(o, c.id, level);
        for (uint32_t i = 0; i < nb->n; i++) {
            uint32_t e = nb->ids[i];
            ct->n_ids++;
            if (visited_test_and_set(s, e)) continue;
            f = W->a[0];
            simpair e2 = { hnsw_oracle_euclidean(query, vec(o, e), o->dim), e };
            ct->n_dist++;
            if (W->n >= ef && e2.sim == f.sim) s->tie_accept++;
            if (e2.sim > f.sim || W->n < ef) {              /* :657 */
                rh_push(C, e2); rh_push(W, e2);
                if (W->n > ef) {
                    simpair out = rh_pop(W);
                    if (!s->have_evicted || out.sim > s->evicted_max) { s->evicted_max = out.sim; s->have_evicted = 1; }
                }
            }
        }
    }
    if (s->have_evicted && W->n && W->a[0].sim == s->evicted_max) s->tie_order++;   /* census: see search_level */
    for (uint32_t i = 0; i < W->n; i++) rh_push(res, W->a[i]);   /* :670-674 */
}
/* core.rs:489-599 */
static void insert_std(hnsw_oracle *o, const float *data, uint32_t l)
{
    scratch *s = &o->sc;
    hnsw_oracle_counters *ct = &o->ins;
    rscratch z; memset(&z, 0, sizeof z);
    uint32_t l_max = o->max_layer;                          /* :496 */
    uint32_t query = store_node(o, data, l);                /* :498-507 */
    const float *qv = vec(o, query);
    uint32_t ep = (uint32_t)o->enterpoint;                  /* :508 */
    uint32_t lc = l_max;                                    /* :511 */
    while (lc > l) {                                        /* :512 */
        search_level_std_ct(o, s, qv, ep, 1, lc, &z.C, &z.W, &z.res, ct);   /* :513 */
        ep = z.res.a[0].id;                                 /* :514 w.pop(): the root */
        for (uint32_t i = 1; i < z.res.n; i++) if (z.res.a[i].sim == z.res.a[0].sim) { s->tie_order++; break; }
        if (lc == 0) break;
        lc--;
    }
    uint32_t top = l_max < l ? l_max : l;
    for (uint32_t lcc = top + 1; lcc-- > 0;) {              /* :523 */
        search_level_std_ct(o, s, qv, ep, o->ef_construction, lcc, &z.C, &z.W, &z.res, ct);   /* :524 */
        select_neighbors_std(o, s, &z, query, &z.res, o->m, lcc, -1, &z.nbrs, ct);          /* :525-531 */
        {                                                   /* :532 connect_neighbors (core.rs:759-774) */
            rheap *t = &z.t;
            rh_clone(t, &z.nbrs);
            while (t->n) { simpair n = rh_pop(t); add_neighbor(o, query, lcc, n.id); add_neighbor(o, n.id, lcc, query); }
        }
        for (uint32_t i = 0; i < z.nbrs.n; i++) touch_add(o, z.nbrs.a[i].id);   /* :535-537 */
        while (z.nbrs.n) {                                  /* :540 */
            simpair e = rh_pop(&z.nbrs);
            rheap *econn = &z.econn;                        /* :544-558 */
            econn->n = 0; econn->reverse = 0;
            const nrow *er = row_of(o, e.id, lcc);
            const float *ev = vec(o, e.id);
            for (uint32_t i = 0; i < er->n; i++) {
                simpair p = { hnsw_oracle_euclidean(ev, vec(o, er->ids[i]), o->dim), er->ids[i] };   /* :550 */
                ct->n_dist++; ct->n_ids++;
                rh_push(econn, p);
            }
            uint32_t m_max = lcc == 0 ? o->m_max0 : o->m_max;   /* :560 */
            if (econn->n > m_max) {                         /* :561 */
                select_neighbors_std(o, s, &z, e.id, econn, m_max, lcc, -1, &z.enew, ct);   /* :568 */
                update_node_connections_std(o, &z, e.id, &z.enew, econn, lcc, -1);          /* :569 */
            }
        }
        ep = z.res.a[0].id;                                 /* :576 w.peek() */
        for (uint32_t i = 1; i < z.res.n; i++) if (z.res.a[i].sim == z.res.a[0].sim) { s->tie_order++; break; }
    }
    if (l > l_max) { o->max_layer = l; o->enterpoint = query; }   /* :587-593 */
    rscratch_free(&z);
}
/* core.rs:383-412 in the Rust binary's tie order; ties[4] (may be NULL) += decisions of THIS insert that met equal
 * similarities of two different nodes: [0] the stop test :635, [1] the accept test :657 with W full, [2] a select cut :733, [3] order-only ties (inside a selection, W's two nearest) */
int64_t hnsw_oracle_add_std_heap(hnsw_oracle *o, const float *v, int32_t level, uint64_t *ties)
{
    if (o->node_count - o->n_dead == 0) {                   /* :393-405 */
        uint32_t id = store_node(o, v, 0);
        o->enterpoint = id;
        return id;
    }
    uint32_t l = level >= 0 ? (uint32_t)level : gen_random_level(o);
    uint32_t id = o->node_count;
    ensure_cap(o);
    touch_reset(o);
    o->sc.tie_stop = o->sc.tie_accept = o->sc.tie_select = o->sc.tie_order = 0;
    insert_std(o, v, l);
    if (ties) { ties[0] += o->sc.tie_stop; ties[1] += o->sc.tie_accept; ties[2] += o->sc.tie_select; ties[3] += o->sc.tie_order; }
    return id;
}

/* Baseline B: T independent searches at a time over the shared read-only graph.  The
 * workers are persistent and each keeps its own scratch (visited stamps + heaps) across calls -- a
 * fresh scratch per call costs a node_count-sized memset per thread, which at 256 threads and 1024
 * queries is more work than the searches themselves.                                              */
typedef struct {
    const hnsw_oracle *o; const float *Q; uint32_t B, k;
    uint32_t *ids; float *sims; uint32_t *n_out;
} batch_job;

typedef struct pool_worker {
    pthread_t th; uint32_t idx; scratch sc; hnsw_oracle_counters ct; uint64_t seen_gen;
} pool_worker;

static struct {
    pthread_mutex_t mu; pthread_cond_t go, done;
    pool_worker *w; uint32_t n;           /* workers alive                          */
    uint32_t active;                      /* workers taking part in the current job */
    uint64_t gen; uint32_t pending;
    volatile uint32_t next;               /* next query index (dynamic chunks)      */
    batch_job job;
} g_pool = { PTHREAD_MUTEX_INITIALIZER, PTHREAD_COND_INITIALIZER, PTHREAD_COND_INITIALIZER, NULL, 0, 0, 0, 0, 0, {0} };

static void run_slice(pool_worker *w, const batch_job *j)
{
    memset(&w->ct, 0, sizeof w->ct);
    for (;;) {
        /* chunks of 4 queries: balances the uneven cost of queries without contention */
        uint32_t lo = __atomic_fetch_add(&g_pool.next, 4, __ATOMIC_RELAXED);
        if (lo >= j->B) break;
        uint32_t hi = lo + 4 < j->B ? lo + 4 : j->B;
        for (uint32_t i = lo; i < hi; i++) {
            if (j->o->enterpoint < 0 || j->o->node_count - j->o->n_dead == 0) { j->n_out[i] = 0; continue; }
            j->n_out[i] = search_knn_internal(j->o, &w->sc, j->Q + (size_t)i * j->o->dim, j->k,
                                              j->o->ef_construction, j->ids + (size_t)i * j->k,
                                              j->sims + (size_t)i * j->k, &w->ct);
        }
    }
}

static void *pool_main(void *arg)
{
    pool_worker *w = (pool_worker *)arg;
    pthread_mutex_lock(&g_pool.mu);
    for (;;) {
        while (w->seen_gen == g_pool.gen) pthread_cond_wait(&g_pool.go, &g_pool.mu);
        w->seen_gen = g_pool.gen;
        int take = w->idx < g_pool.active;
        batch_job j = g_pool.job;
        pthread_mutex_unlock(&g_pool.mu);
        if (take) run_slice(w, &j);
        pthread_mutex_lock(&g_pool.mu);
        if (take && --g_pool.pending == 0) pthread_cond_signal(&g_pool.done);
    }
    return NULL;
}

/* grow the pool to `threads` workers (never shrinks; idle workers sleep on the condition variable) */
static void pool_ensure(uint32_t threads)
{
    if (g_pool.n >= threads) return;
    /* workers hold pointers into the array: allocate it once, large enough */
    if (!g_pool.w) g_pool.w = (pool_worker *)calloc(1024, sizeof(pool_worker));
    if (threads > 1024) threads = 1024;
    for (uint32_t t = g_pool.n; t < threads; t++) {
        pool_worker *w = &g_pool.w[t];
        w->idx = t; w->seen_gen = g_pool.gen;
        scratch_init(&w->sc);
        pthread_create(&w->th, NULL, pool_main, w);
    }
    g_pool.n = threads;
}

void hnsw_oracle_search_batch(const hnsw_oracle *o, const float *Q, uint32_t B,
                              uint32_t k, uint32_t *ids, float *sims, uint32_t *n_out,
                              uint32_t threads, hnsw_oracle_counters *ctrs)
{
    hnsw_oracle_counters sum = { 0, 0, 0 };
    if (threads < 1) threads = 1;
    if (threads > 1024) threads = 1024;
    if (threads > B) threads = B ? B : 1;
    if (threads == 1) {
        /* Baseline A: the reference's single command thread; the index's own scratch persists */
        for (uint32_t i = 0; i < B; i++) {
            if (o->enterpoint < 0 || o->node_count - o->n_dead == 0) { n_out[i] = 0; continue; }
            n_out[i] = search_knn_internal(o, (scratch *)&o->sc, Q + (size_t)i * o->dim, k, o->ef_construction,
                                           ids + (size_t)i * k, sims + (size_t)i * k, &sum);
        }
        if (ctrs) *ctrs = sum;
        return;
    }
    pthread_mutex_lock(&g_pool.mu);
    pool_ensure(threads);
    g_pool.job = (batch_job){ o, Q, B, k, ids, sims, n_out };
    g_pool.active = threads;
    g_pool.pending = threads;
    g_pool.next = 0;
    g_pool.gen++;
    pthread_cond_broadcast(&g_pool.go);
    while (g_pool.pending) pthread_cond_wait(&g_pool.done, &g_pool.mu);
    for (uint32_t t = 0; t < threads; t++) {
        sum.n_dist += g_pool.w[t].ct.n_dist; sum.n_ids += g_pool.w[t].ct.n_ids; sum.n_expand += g_pool.w[t].ct.n_expand;
    }
    pthread_mutex_unlock(&g_pool.mu);
    if (ctrs) *ctrs = sum;
}

/* ========================================================================= */
/* core.rs:414-475 delete_node, :824-863 delete_node_from_neighbors            */
/* ========================================================================= */
/* core.rs:824-863: every neighbour n of `node` at layer lc re-selects its links
 * from its own neighbourhood (two hops) with `node` ignored.                  */
static void delete_node_from_neighbors(hnsw_oracle *o, scratch *s, uint32_t node, uint32_t lc,
                                       hnsw_oracle_counters *ct)
{
    /* the reference iterates a borrowed view of node's row (:826,829); the row is not
     * modified during the loop (node is `ignored` everywhere), copy it for safety      */
    const nrow *nr = row_of(o, node, lc);
    uint32_t cnt = nr->n;
    uint32_t *nbrs = (uint32_t *)malloc((size_t)(cnt ? cnt : 1) * 4);
    memcpy(nbrs, nr->ids, (size_t)cnt * 4);
    for (uint32_t k = 0; k < cnt; k++) {                /* :829 stored order   */
        uint32_t n = nbrs[k];
        heap *nconn = &s->econn;                        /* :832-844 */
        heap_clear(nconn); nconn->furthest_top = 0;
        const nrow *r = row_of(o, n, lc);
        const float *nv = vec(o, n);
        for (uint32_t i = 0; i < r->n; i++) {
            simpair p = { hnsw_oracle_euclidean(nv, vec(o, r->ids[i]), o->dim), r->ids[i] }; /* :840-841 */
            ct->n_dist++;
            ct->n_ids++;
            heap_push(nconn, p);
        }
        uint32_t m_max = lc == 0 ? o->m_max0 : o->m_max; /* :846 */
        select_neighbors(o, s, n, nconn, m_max, lc, (int64_t)node, &s->enew, ct); /* :853 */
        touch_add(o, n);                                /* :855 */
        update_node_connections(o, s, n, &s->enew, nconn, lc, (int64_t)node); /* :856 */
    }
    free(nbrs);
}

/* core.rs:414-475.  Enterpoint re-election (:449-472) takes `layers[lc].iter().next()` of a
 * HashSet in the reference, i.e. an arbitrary node of the highest non-empty layer; the oracle
 * (and the engine) take the one with the SMALLEST ID.  Returns 0, or -1 if id is not a live node
 * ("Node: {:?} does not exist", :421).                                                         */
static void delete_node_from_neighbors_std(hnsw_oracle *o, scratch *s, rscratch *z, uint32_t node, uint32_t lc, hnsw_oracle_counters *ct);
static int delete_impl(hnsw_oracle *o, uint32_t id, uint32_t *touched, uint32_t touched_cap, uint32_t *n_touched, int std_heap);
int hnsw_oracle_delete(hnsw_oracle *o, uint32_t id, uint32_t *touched, uint32_t touched_cap,
                       uint32_t *n_touched)
{
    return delete_impl(o, id, touched, touched_cap, n_touched, 0);
}
static int delete_impl(hnsw_oracle *o, uint32_t id, uint32_t *touched, uint32_t touched_cap, uint32_t *n_touched, int std_heap)
{
    if (n_touched) *n_touched = 0;
    if (id >= o->node_count || o->dead[id]) return -1;  /* :419-422 */
    touch_reset(o);
    uint32_t top = o->nodes[id].level;                  /* :434 node.neighbors.len() rows */
    if (std_heap) {
        rscratch z; memset(&z, 0, sizeof z);
        for (uint32_t lc = 0; lc <= top; lc++) delete_node_from_neighbors_std(o, &o->sc, &z, id, lc, &o->ins);
        rscratch_free(&z);
    } else
    for (uint32_t lc = 0; lc <= top; lc++)              /* :434-439 ascending         */
        delete_node_from_neighbors(o, &o->sc, id, lc, &o->ins);
    o->dead[id] = 1;                                    /* :419 nodes.remove, :424     */
    o->n_dead++;
    for (uint32_t lc = 0; lc <= top; lc++) o->nodes[id].rows[lc].n = 0;
    if (touched) {
        uint32_t n = o->n_touch < touched_cap ? o->n_touch : touched_cap;
        memcpy(touched, o->touch, (size_t)n * 4);
    }
    if (n_touched) *n_touched = o->n_touch;
    if (o->enterpoint == (int64_t)id) {                 /* :449-472 */
        int64_t best = -1;
        uint32_t best_level = 0;
        for (uint32_t i = 0; i < o->node_count; i++)
            if (!o->dead[i] && (best < 0 || o->nodes[i].level > best_level)) { best = i; best_level = o->nodes[i].level; }
        o->enterpoint = best;
        /* empty top layers are popped and max_layer decremented, never below 0 (:458-465) */
        o->max_layer = best >= 0 ? best_level : 0;
    }
    return 0;
}

/* HNSW.NODE.DEL in the RUST BINARY's tie order (test infrastructure): core.rs:824-863 with nconn a std BinaryHeap filled in
 * the row's stored order (:832-844), select_neighbors / update_node_connections as insert_std runs them, the deleted node
 * ignored (:853, :856).  Pinned against the transcription's "rust"-mode run with deletes (tests/golden/tiecase_rust_lattice_del.npz). */
static void delete_node_from_neighbors_std(hnsw_oracle *o, scratch *s, rscratch *z, uint32_t node, uint32_t lc, hnsw_oracle_counters *ct)
{
    const nrow *nr = row_of(o, node, lc);
    uint32_t cnt = nr->n;
    uint32_t *nbrs = (uint32_t *)malloc((size_t)(cnt ? cnt : 1) * 4);
    memcpy(nbrs, nr->ids, (size_t)cnt * 4);
    for (uint32_t k = 0; k < cnt; k++) {                    /* :829 stored order */
        uint32_t n = nbrs[k];
        rheap *nconn = &z->econn;                           /* :832-844 */
        nconn->n = 0; nconn->reverse = 0;
        const nrow *r = row_of(o, n, lc);
        const float *nv = vec(o, n);
        for (uint32_t i = 0; i < r->n; i++) {
            simpair p = { hnsw_oracle_euclidean(nv, vec(o, r->ids[i]), o->dim), r->ids[i] };
            ct->n_dist++; ct->n_ids++;
            rh_push(nconn, p);
        }
        uint32_t m_max = lc == 0 ? o->m_max0 : o->m_max;    /* :846 */
        select_neighbors_std(o, s, z, n, nconn, m_max, lc, (int64_t)node, &z->enew, ct);    /* :853 */
        touch_add(o, n);                                    /* :855 */
        update_node_connections_std(o, z, n, &z->enew, nconn, lc, (int64_t)node);           /* :856 */
    }
    free(nbrs);
}
static int delete_impl(hnsw_oracle *o, uint32_t id, uint32_t *touched, uint32_t touched_cap, uint32_t *n_touched, int std_heap);
int hnsw_oracle_delete_std_heap(hnsw_oracle *o, uint32_t id, uint32_t *touched, uint32_t touched_cap, uint32_t *n_touched)
{
    return delete_impl(o, id, touched, touched_cap, n_touched, 1);
}

void hnsw_oracle_last_add_ties(const hnsw_oracle *o, uint64_t out[4])
{
    out[0] = o->sc.tie_stop; out[1] = o->sc.tie_accept; out[2] = o->sc.tie_select; out[3] = o->sc.tie_order;
}

uint32_t hnsw_oracle_live_count(const hnsw_oracle *o) { return o->node_count - o->n_dead; }
int hnsw_oracle_is_live(const hnsw_oracle *o, uint32_t id) { return id < o->node_count && !o->dead[id]; }

/* ========================================================================= */
/* introspection / bulk transfer                                             */
/* ========================================================================= */
uint32_t hnsw_oracle_node_count(const hnsw_oracle *o) { return o->node_count; }
uint32_t hnsw_oracle_max_layer(const hnsw_oracle *o) { return o->max_layer; }
int64_t hnsw_oracle_enterpoint(const hnsw_oracle *o) { return o->enterpoint; }
uint32_t hnsw_oracle_level(const hnsw_oracle *o, uint32_t id) { return o->nodes[id].level; }
uint32_t hnsw_oracle_degree(const hnsw_oracle *o, uint32_t id, uint32_t layer) { return row_of(o, id, layer)->n; }
uint32_t hnsw_oracle_neighbors(const hnsw_oracle *o, uint32_t id, uint32_t layer, uint32_t *out, uint32_t cap)
{
    const nrow *r = row_of(o, id, layer);
    uint32_t n = r->n < cap ? r->n : cap;
    if (n) memcpy(out, r->ids, (size_t)n * 4);
    return r->n;
}
const float *hnsw_oracle_vector(const hnsw_oracle *o, uint32_t id) { return vec(o, id); }
void hnsw_oracle_insert_counters(const hnsw_oracle *o, hnsw_oracle_counters *c) { *c = o->ins; }

uint64_t hnsw_oracle_layer_nnz(const hnsw_oracle *o, uint32_t layer)
{
    uint64_t nnz = 0;
    for (uint32_t i = 0; i < o->node_count; i++) nnz += row_of(o, i, layer)->n;
    return nnz;
}
void hnsw_oracle_export_layer(const hnsw_oracle *o, uint32_t layer, uint64_t *row_ptr, uint32_t *col)
{
    uint64_t p = 0;
    for (uint32_t i = 0; i < o->node_count; i++) {
        const nrow *r = row_of(o, i, layer);
        row_ptr[i] = p;
        if (r->n) memcpy(col + p, r->ids, (size_t)r->n * 4);
        p += r->n;
    }
    row_ptr[o->node_count] = p;
}
void hnsw_oracle_export_levels(const hnsw_oracle *o, uint32_t *levels)
{
    for (uint32_t i = 0; i < o->node_count; i++) levels[i] = o->nodes[i].level;
}

hnsw_oracle *hnsw_oracle_import(uint32_t dim, uint32_t m, uint32_t ef_construction,
                                uint32_t n, const float *vectors, const uint32_t *levels,
                                int64_t enterpoint, uint32_t n_layers,
                                const uint64_t *const *row_ptr, const uint32_t *const *col)
{
    hnsw_oracle *o = hnsw_oracle_new(dim, m, ef_construction, 0);
    o->cap = n ? n : 1;
    o->data = (float *)malloc((size_t)o->cap * dim * sizeof(float));
    o->nodes = (onode *)calloc(o->cap, sizeof(onode));
    o->dead = (uint8_t *)calloc(o->cap, 1);
    memcpy(o->data, vectors, (size_t)n * dim * sizeof(float));
    o->node_count = n;
    for (uint32_t i = 0; i < n; i++) {
        uint32_t L = levels[i];
        o->nodes[i].level = L;
        o->nodes[i].rows = (nrow *)calloc(L + 1, sizeof(nrow));
        for (uint32_t l = 0; l <= L && l < n_layers; l++) {
            uint64_t b = row_ptr[l][i], e = row_ptr[l][i + 1];
            nrow *r = &o->nodes[i].rows[l];
            r->n = r->cap = (uint32_t)(e - b);
            if (r->n) {
                r->ids = (uint32_t *)malloc((size_t)r->n * 4);
                memcpy(r->ids, col[l] + b, (size_t)r->n * 4);
            }
        }
    }
    o->enterpoint = enterpoint;
    o->max_layer = enterpoint >= 0 ? levels[enterpoint] : 0;
    return o;
}
