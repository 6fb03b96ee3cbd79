/*
 * occ_model.c -- CPU model of the exact-order PARALLEL insert (TEST INFRASTRUCTURE; includes the
 * oracle's source).  It answers two questions before any HIP is written:
 *   1. soundness: do the validation rules below reproduce the reference's serial graph exactly?
 *      (the model's graph is compared row by row with the plain oracle's)
 *   2. yield: how many commits per planning round, how many shrinks survive speculation?
 *
 * Scheme (optimistic concurrency, deterministic order).  A window of W consecutive HNSW.NODE.ADDs
 * is PLANNED against one snapshot of the graph: the read-only part of core.rs:489-599, i.e. per
 * layer search_level(ef_construction) + select_neighbors(m) (:511-531), and -- speculatively -- the
 * select_neighbors(m_max) of every selected neighbour e the connect would push over m_max
 * (:560-568), computed as if only this node's connect had happened since the snapshot.  Plans are
 * COMMITTED strictly in id order.  Every commit appends its row changes to a journal of deltas
 * (row, layer, id, added/removed).  Before a plan (or one of its shrinks) is applied, the journal
 * entries since its snapshot are checked against what the plan READ:
 *   search_level  expanded row h, B = W's furthest after that expansion (or "W not full"):
 *                 a delta z on row h matters iff W was not full or key(q,z) <= B      (core.rs:657)
 *   select        rows of the members of W; B = the m-th selected:                     (core.rs:724-754)
 *                 a delta z matters iff fewer than m were selected or key(q,z) <= B
 *   shrink of e   row e itself must be untouched (except this node's own append); rows of e's
 *                 neighbours: a delta z matters iff key(e,z) <= the m_max-th selected
 * A link plan with a relevant delta is re-planned (end of the round); a shrink with a relevant
 * delta is recomputed serially at commit time.
 *
 * gcc -O3 -mavx2 -mfma -ffp-contract=off -o /tmp/occ_model tests/experiments/occ_model.c -lm -lpthread
 * /tmp/occ_model N0 K W [dim M ef]
 */
#include "../../oracle/hnsw_oracle.c"

typedef struct { uint32_t row, lc, z; int add; } delta;
static delta *J; static uint32_t nJ, capJ;
static uint32_t hdr_epoch;   /* bumped when enterpoint/max_layer change */
static void jpush(uint32_t row, uint32_t lc, uint32_t z, int add)
{
    if (nJ == capJ) { capJ = capJ ? capJ * 2 : 4096; J = realloc(J, capJ * sizeof(delta)); }
    J[nJ++] = (delta){row, lc, z, add};
}

static uint64_t why[2][2][2]; static int g_count_why; static int g_refine = 1; static int g_tight = 1;
enum { RK_SEARCH = 0, RK_SELECT = 1, RK_SHRINK_NB = 2, RK_SHRINK_ROW = 3 };
typedef struct { uint32_t row, lc; simpair bound; int full, kind, sub; int32_t t; } rd;
typedef struct { uint32_t lc, e; simpair S[64]; uint32_t nS; int valid; } shr;
typedef struct {
    int planned; uint32_t snap, snap_hdr;
    rd *r; uint32_t nr, capr;
    int32_t *hhead, *hnext; uint32_t hsize;
    uint64_t *vkey; int32_t *vt; uint32_t vsize, vcount; int32_t tclock;
    uint32_t top; simpair sel[32][64]; uint32_t nsel[32];
    shr *sh; uint32_t nsh, capsh;
} txn;

static void rd_push(txn *t, uint32_t row, uint32_t lc, simpair b, int full, int kind, int sub)
{
    if (t->nr == t->capr) { t->capr = t->capr ? t->capr * 2 : 1024; t->r = realloc(t->r, t->capr * sizeof(rd)); }
    t->r[t->nr++] = (rd){row, lc, b, full, kind, sub, t->tclock};
}


static void vmap_reset(txn *t)
{
    if (!t->vsize) { t->vsize = 1u << 16; t->vkey = malloc(t->vsize * 8); t->vt = malloc(t->vsize * 4); }
    memset(t->vkey, 0xFF, t->vsize * 8); t->vcount = 0; t->tclock = 0;
}
static void vmap_put(txn *t, uint32_t id, uint32_t lc, int32_t tm)
{
    uint64_t k = ((uint64_t)lc << 32) | id; uint32_t h = (uint32_t)((k * 0x9E3779B97F4A7C15ull) >> 40) & (t->vsize - 1);
    while (t->vkey[h] != ~0ull && t->vkey[h] != k) h = (h + 1) & (t->vsize - 1);
    if (t->vkey[h] == k) return;
    t->vkey[h] = k; t->vt[h] = tm; t->vcount++;
    if (t->vcount * 2 > t->vsize) { fprintf(stderr, "vmap full\n"); abort(); }
}
static int32_t vmap_get(const txn *t, uint32_t id, uint32_t lc)   /* first-visit time or INT32_MAX */
{
    uint64_t k = ((uint64_t)lc << 32) | id; uint32_t h = (uint32_t)((k * 0x9E3779B97F4A7C15ull) >> 40) & (t->vsize - 1);
    while (t->vkey[h] != ~0ull) { if (t->vkey[h] == k) return t->vt[h]; h = (h + 1) & (t->vsize - 1); }
    return 0x7fffffff;
}

static void search_level_log(const hnsw_oracle *o, scratch *s, const float *query, uint32_t ep, uint32_t ef,
                             uint32_t level, txn *t)
{
    visited_reset(s, o->node_count);
    visited_test_and_set(s, ep);
    vmap_put(t, ep, level, t->tclock);
    simpair qpair = { hnsw_oracle_euclidean(query, vec(o, ep), o->dim), ep };
    heap *C = &s->C, *W = &s->W;
    heap_clear(C); heap_clear(W);
    heap_push(C, qpair); heap_push(W, qpair);
    const uint32_t log_start = t->nr;
    while (C->n) {
        simpair c = heap_pop(C);
        simpair f = heap_peek(W);
        if (nearer(f, c)) break;
        t->tclock++;
        const nrow *nb = row_of(o, c.id, level);
        for (uint32_t i = 0; i < nb->n; i++) {
            uint32_t e = nb->ids[i];
            if (visited_test_and_set(s, e)) continue;
            vmap_put(t, e, level, t->tclock);
            f = heap_peek(W);
            simpair ep2 = { hnsw_oracle_euclidean(query, vec(o, e), o->dim), e };
            if (nearer(ep2, f) || W->n < ef) { heap_push(C, ep2); heap_push(W, ep2); if (W->n > ef) heap_pop(W); }
        }
        /* g_tight: the popped candidate for now (thresholds in the backward pass below); else the accept
           threshold of the moment (the rule of the first version) */
        if (g_tight) rd_push(t, c.id, level, c, 0, RK_SEARCH, 0);
        else rd_push(t, c.id, level, heap_peek(W), W->n >= ef, RK_SEARCH, 0);
    }
    if (g_tight) {
        /* bound of an expansion = the farther of W's final furthest and every candidate popped after it
           (hnsw_device.hpp, occ_finalize_search_log) */
        int full = W->n >= ef;
        simpair running = heap_peek(W);
        for (uint32_t j = t->nr; j-- > log_start;) {
            simpair pop = t->r[j].bound;
            t->r[j].bound = running; t->r[j].full = full;
            if (nearer(running, pop)) running = pop;
        }
    }
}

static simpair worst_of(const heap *h)
{
    simpair w = h->a[0];
    for (uint32_t i = 1; i < h->n; i++) if (nearer(w, h->a[i])) w = h->a[i];
    return w;
}

/* plan node q (already stored, rows empty, unreachable) against the current graph */
static void plan_txn(hnsw_oracle *o, uint32_t q, txn *t)
{
    scratch *s = &o->sc;
    hnsw_oracle_counters ct = {0, 0, 0};
    t->nr = 0; t->nsh = 0; t->planned = 1; t->snap = nJ; t->snap_hdr = hdr_epoch; vmap_reset(t);
    const float *qv = vec(o, q);
    uint32_t l = o->nodes[q].level, l_max = o->max_layer, ep = (uint32_t)o->enterpoint, lc = l_max;
    while (lc > l) { search_level_log(o, s, qv, ep, 1, lc, t); ep = nearest_of_W(s).id; if (lc == 0) break; lc--; }
    uint32_t top = l_max < l ? l_max : l;
    t->top = top;
    for (uint32_t lcc = top + 1; lcc-- > 0;) {
        search_level_log(o, s, qv, ep, o->ef_construction, lcc, t);
        heap_copy(&s->res, &s->W, 0);
        simpair w_nearest = heap_peek(&s->res);
        select_neighbors(o, s, q, &s->res, o->m, lcc, -1, &s->nbrs, &ct);
        simpair worst = s->nbrs.n ? worst_of(&s->nbrs) : (simpair){0, 0};
        for (uint32_t i = 0; i < s->res.n; i++) rd_push(t, s->res.a[i].id, lcc, worst, s->nbrs.n >= o->m, RK_SELECT, 0);
        heap *tt = &s->ccopy; heap_copy(tt, &s->nbrs, 0);
        t->nsel[lcc] = 0;
        while (tt->n) t->sel[lcc][t->nsel[lcc]++] = heap_pop(tt);
        ep = w_nearest.id;
    }
    /* speculative shrinks: apply this node's connect as an overlay, compute, undo */
    for (uint32_t lcc = top + 1; lcc-- > 0;) {
        uint32_t m_max = lcc == 0 ? o->m_max0 : o->m_max;
        for (uint32_t i = 0; i < t->nsel[lcc]; i++) { add_neighbor(o, q, lcc, t->sel[lcc][i].id); add_neighbor(o, t->sel[lcc][i].id, lcc, q); }
        for (uint32_t i = 0; i < t->nsel[lcc]; i++) {
            uint32_t e = t->sel[lcc][i].id;
            const nrow *er = row_of(o, e, lcc);
            if (er->n <= m_max) continue;
            heap *econn = &s->econn; heap_clear(econn); econn->furthest_top = 0;
            const float *ev = vec(o, e);
            for (uint32_t a = 0; a < er->n; a++) { simpair p = { hnsw_oracle_euclidean(ev, vec(o, er->ids[a]), o->dim), er->ids[a] }; heap_push(econn, p); }
            select_neighbors(o, s, e, econn, m_max, lcc, -1, &s->enew, &ct);
            if (t->nsh == t->capsh) { t->capsh = t->capsh ? t->capsh * 2 : 16; t->sh = realloc(t->sh, t->capsh * sizeof(shr)); }
            shr *sp = &t->sh[t->nsh];
            sp->lc = lcc; sp->e = e; sp->nS = 0; sp->valid = 1;
            heap *tt = &s->ccopy; heap_copy(tt, &s->enew, 0);
            while (tt->n) sp->S[sp->nS++] = heap_pop(tt);
            simpair worst = sp->S[sp->nS - 1];
            rd_push(t, e, lcc, worst, 1, RK_SHRINK_ROW, (int)t->nsh);
            for (uint32_t a = 0; a < er->n; a++) rd_push(t, er->ids[a], lcc, worst, 1, RK_SHRINK_NB, (int)t->nsh);
            t->nsh++;
        }
        /* undo the overlay */
        for (uint32_t i = 0; i < t->nsel[lcc]; i++) { rm_neighbor(o, t->sel[lcc][i].id, lcc, q); }
        o->nodes[q].rows[lcc].n = 0;
    }
    uint32_t hs = 1024; while (hs < 2 * t->nr) hs *= 2;
    if (hs > t->hsize) { t->hhead = realloc(t->hhead, hs * 4); t->hsize = hs; }
    t->hnext = realloc(t->hnext, (t->nr + 1) * 4);
    for (uint32_t i = 0; i < t->hsize; i++) t->hhead[i] = -1;
    for (uint32_t i = 0; i < t->nr; i++) { uint32_t h = ((t->r[i].row * 2654435761u) ^ (t->r[i].lc * 40503u)) & (t->hsize - 1); t->hnext[i] = t->hhead[h]; t->hhead[h] = (int32_t)i; }
}

/* validate txn t (node q) against journal[from, nJ): returns 1 if the link plan is still valid; marks shrinks */
static int validate(hnsw_oracle *o, uint32_t q, txn *t, uint32_t from, int own_from_valid)
{
    if (t->snap_hdr != hdr_epoch) return 0;
    const float *qv = vec(o, q);
    int ok = 1;
    for (uint32_t ji = from; ji < nJ; ji++) {
        const delta *d = &J[ji];
        uint32_t hh = ((d->row * 2654435761u) ^ (d->lc * 40503u)) & (t->hsize - 1);
        for (int32_t i = t->hhead[hh]; i >= 0; i = t->hnext[i]) {
            const rd *r = &t->r[i];
            if (r->row != d->row || r->lc != d->lc) continue;
            if (r->kind == RK_SHRINK_ROW) {
                if (d->z == q && d->add && own_from_valid) continue;   /* this node's own connect */
                t->sh[r->sub].valid = 0;
                continue;
            }
            const float *rv = r->kind == RK_SHRINK_NB ? vec(o, t->sh[r->sub].e) : qv;
            if (r->kind == RK_SHRINK_NB && !t->sh[r->sub].valid) continue;
            if (r->kind == RK_SHRINK_NB && (d->z == t->sh[r->sub].e || d->z == q)) continue; /* e itself is excluded (core.rs:704); q is in econn already */
            if (r->kind == RK_SEARCH && g_refine) {
                int32_t tv = vmap_get(t, d->z, d->lc);
                if (!d->add && tv < r->t) continue;      /* z was not fresh in this row: its removal changes nothing */
                if (d->add && tv <= r->t) continue;      /* z was already visited when this row was expanded */
            }
            simpair pk = { hnsw_oracle_euclidean(rv, vec(o, d->z), o->dim), d->z };
            int relevant = !r->full || !nearer(r->bound, pk);     /* key(z) <= bound */
            if (!relevant) continue;
            if (r->kind == RK_SHRINK_NB) t->sh[r->sub].valid = 0;
            else { ok = 0; if (g_count_why) why[r->kind][r->full][d->add]++; }
        }
    }
    return ok;
}

static uint64_t st_spec_applied, st_fallback, st_noshrink_spec_unused, st_commits, st_rounds, st_replans, st_plans;

static void apply_shrink(hnsw_oracle *o, uint32_t e, uint32_t lcc, const simpair *S, uint32_t nS)
{
    scratch *s = &o->sc;
    const nrow *er = row_of(o, e, lcc);
    uint32_t on = er->n, oldr[600];
    memcpy(oldr, er->ids, on * 4);
    heap *econn = &s->econn; heap_clear(econn); econn->furthest_top = 0;
    const float *ev = vec(o, e);
    for (uint32_t a = 0; a < on; a++) { simpair p = { hnsw_oracle_euclidean(ev, vec(o, oldr[a]), o->dim), oldr[a] }; heap_push(econn, p); }
    heap *enew = &s->enew; heap_clear(enew); enew->furthest_top = 0;
    if (S) for (uint32_t a = 0; a < nS; a++) heap_push(enew, S[a]);
    else { hnsw_oracle_counters ct = {0,0,0}; uint32_t m_max = lcc == 0 ? o->m_max0 : o->m_max; select_neighbors(o, s, e, econn, m_max, lcc, -1, enew, &ct); }
    update_node_connections(o, s, e, enew, econn, lcc, -1);
    const nrow *nr = row_of(o, e, lcc);
    for (uint32_t a = 0; a < on; a++) { int f = 0; for (uint32_t b = 0; b < nr->n; b++) f |= nr->ids[b] == oldr[a]; if (!f) { jpush(e, lcc, oldr[a], 0); jpush(oldr[a], lcc, e, 0); } }
    for (uint32_t b = 0; b < nr->n; b++) { int f = 0; for (uint32_t a = 0; a < on; a++) f |= nr->ids[b] == oldr[a]; if (!f) { jpush(e, lcc, nr->ids[b], 1); jpush(nr->ids[b], lcc, e, 1); } }
}

/* commit node q with its (link-valid) plan */
static void commit_txn(hnsw_oracle *o, uint32_t q, txn *t)
{
    scratch *s = &o->sc;
    touch_reset(o);
    uint32_t l = o->nodes[q].level, l_max = o->max_layer;
    for (uint32_t lcc = t->top + 1; lcc-- > 0;) {
        uint32_t m_max = lcc == 0 ? o->m_max0 : o->m_max;
        uint32_t j0 = nJ;
        for (uint32_t i = 0; i < t->nsel[lcc]; i++) {
            uint32_t e = t->sel[lcc][i].id;
            add_neighbor(o, q, lcc, e); add_neighbor(o, e, lcc, q);
            jpush(e, lcc, q, 1);
        }
        (void)j0;
        for (uint32_t i = 0; i < t->nsel[lcc]; i++) {
            uint32_t e = t->sel[lcc][i].id;
            if (row_of(o, e, lcc)->n <= m_max) continue;
            /* find the speculative shrink, re-validate against everything journalled since the snapshot */
            shr *sp = NULL;
            for (uint32_t k = 0; k < t->nsh; k++) if (t->sh[k].e == e && t->sh[k].lc == lcc) sp = &t->sh[k];
            if (sp) { validate(o, q, t, t->snap, 1); }
            if (sp && sp->valid) { apply_shrink(o, e, lcc, sp->S, sp->nS); st_spec_applied++; }
            else { apply_shrink(o, e, lcc, NULL, 0); st_fallback++; }
            t->snap = t->snap;  /* journal keeps growing; later shrinks are validated against all of it */
        }
        (void)s;
    }
    if (l > l_max) { o->max_layer = l; o->enterpoint = q; hdr_epoch++; }
    st_commits++;
}

static int rows_equal(const hnsw_oracle *a, const hnsw_oracle *b)
{
    if (a->node_count != b->node_count || a->enterpoint != b->enterpoint || a->max_layer != b->max_layer) return 0;
    for (uint32_t i = 0; i < a->node_count; i++) {
        if (a->nodes[i].level != b->nodes[i].level) return 0;
        for (uint32_t l = 0; l <= a->nodes[i].level; l++) {
            const nrow *ra = &a->nodes[i].rows[l], *rb = &b->nodes[i].rows[l];
            if (ra->n != rb->n || memcmp(ra->ids, rb->ids, ra->n * 4)) { fprintf(stderr, "row %u L%u differs\n", i, l); return 0; }
        }
    }
    return 1;
}

int main(int argc, char **argv)
{
    uint32_t N0 = argc > 1 ? atoi(argv[1]) : 20000, K = argc > 2 ? atoi(argv[2]) : 2048, Wn = argc > 3 ? atoi(argv[3]) : 64;
    uint32_t dim = argc > 4 ? atoi(argv[4]) : 128, M = argc > 5 ? atoi(argv[5]) : 16, ef = argc > 6 ? atoi(argv[6]) : 200;
    if (getenv("NOREFINE")) g_refine = 0;
    if (getenv("LOOSE")) g_tight = 0;            /* the first version's rule: accept threshold at expansion time */
    hnsw_oracle *A = hnsw_oracle_new(dim, M, ef, 7);
    uint32_t total = N0 + K;
    float *V = malloc((size_t)total * dim * 4);
    uint64_t x = 12345;
    for (size_t i = 0; i < (size_t)total * dim; i++) V[i] = (float)((splitmix64(&x) >> 40) * (1.0 / 16777216.0));
    uint32_t *lv = malloc(total * 4);
    for (uint32_t i = 0; i < total; i++) lv[i] = gen_random_level(A);
    for (uint32_t i = 0; i < N0; i++) { hnsw_oracle_add(A, V + (size_t)i * dim, (int32_t)lv[i], NULL, 0, NULL); if (i % 20000 == 0) fprintf(stderr, "built %u\n", i); }
    /* clone A -> B through export/import */
    uint32_t L = A->max_layer + 1;
    uint64_t **rp = malloc(L * sizeof *rp); uint32_t **cl = malloc(L * sizeof *cl);
    uint32_t *lev = malloc(N0 * 4); hnsw_oracle_export_levels(A, lev);
    for (uint32_t l = 0; l < L; l++) { rp[l] = malloc(((size_t)N0 + 1) * 8); cl[l] = malloc((hnsw_oracle_layer_nnz(A, l) + 1) * 4); hnsw_oracle_export_layer(A, l, rp[l], cl[l]); }
    hnsw_oracle *B = hnsw_oracle_import(dim, M, ef, N0, A->data, lev, A->enterpoint, L, (const uint64_t *const *)rp, (const uint32_t *const *)cl);
    /* reference: plain serial inserts on A */
    for (uint32_t i = N0; i < total; i++) hnsw_oracle_add(A, V + (size_t)i * dim, (int32_t)lv[i], NULL, 0, NULL);
    fprintf(stderr, "reference done\n");

    /* model on B: store all K nodes (unlinked) up front */
    for (uint32_t i = N0; i < total; i++) { ensure_cap(B); store_node(B, V + (size_t)i * dim, lv[i]); }
    /* node_count now counts unlinked nodes too; they are unreachable, visited_reset sizes by node_count: fine */
    txn *T = calloc(K, sizeof(txn));
    uint32_t head = 0;
    uint64_t run_hist[8] = {0};
    while (head < K) {
        st_rounds++;
        uint32_t wend = head + Wn < K ? head + Wn : K;
        /* (re)plan everything in the window that has no valid plan */
        for (uint32_t j = head; j < wend; j++) {
            if (T[j].planned) {
                int ok = validate(B, N0 + j, &T[j], T[j].snap, 0);
                int shok = 1; for (uint32_t k = 0; k < T[j].nsh; k++) shok &= T[j].sh[k].valid;
                if (ok && shok) continue;
                st_replans++;
            }
            plan_txn(B, N0 + j, &T[j]); st_plans++;
        }
        uint32_t run = 0;
        while (head < wend) {
            txn *t = &T[head];
            for (uint32_t k = 0; k < t->nsh; k++) t->sh[k].valid = 1;
            g_count_why = 1; int vok = validate(B, N0 + head, t, t->snap, 0); g_count_why = 0;
            if (!vok) break;
            commit_txn(B, N0 + head, t);
            head++; run++;
        }
        int b = run <= 1 ? 0 : run <= 2 ? 1 : run <= 4 ? 2 : run <= 8 ? 3 : run <= 16 ? 4 : run <= 32 ? 5 : run <= 64 ? 6 : 7;
        run_hist[b]++;
    }
    int same = rows_equal(A, B);
    printf("N0=%u K=%u W=%u dim=%u M=%u ef=%u : graphs %s\n", N0, K, Wn, dim, M, ef, same ? "IDENTICAL" : "DIFFER");
    printf("  rounds=%lu commits/round=%.1f plans=%lu (replans %lu) plans/commit=%.2f\n", (unsigned long)st_rounds, (double)K / st_rounds, (unsigned long)st_plans, (unsigned long)st_replans, (double)st_plans / K);
    printf("  shrinks/commit=%.2f  speculative applied=%.3f fallback=%.3f\n", (double)(st_spec_applied + st_fallback) / K, (double)st_spec_applied / (st_spec_applied + st_fallback + 1e-9), (double)st_fallback / (st_spec_applied + st_fallback + 1e-9));
    printf("  journal deltas/commit=%.1f\n", (double)nJ / K);
    printf("  run-length histogram (<=1,2,4,8,16,32,64,>64):"); for (int i = 0; i < 8; i++) printf(" %lu", (unsigned long)run_hist[i]); printf("\n");
    printf("  head-invalid reasons [kind][full][add]: search nf- %lu nf+ %lu f- %lu f+ %lu | select nf- %lu nf+ %lu f- %lu f+ %lu\n", (unsigned long)why[0][0][0],(unsigned long)why[0][0][1],(unsigned long)why[0][1][0],(unsigned long)why[0][1][1],(unsigned long)why[1][0][0],(unsigned long)why[1][0][1],(unsigned long)why[1][1][0],(unsigned long)why[1][1][1]);
    /* predicted build rate: round overhead 1.5 ms, commit 15 us, spec shrink 2 us, fallback 52 us */
    double tsec = st_rounds * 1.5e-3 + K * 15e-6 + st_spec_applied * 2e-6 + st_fallback * 52e-6;
    printf("  predicted %.0f inserts/s (1.5 ms/round, 15 us/commit, 2 us/spec shrink, 52 us/fallback)\n", K / tsec);
    return same ? 0 : 1;
}
