/*
 * spec_conflicts.c -- design experiment (TEST INFRASTRUCTURE, links the oracle's source).
 *
 * Question: if W consecutive HNSW.NODE.ADDs are PLANNED (core.rs:511-531, read-only) against one
 * snapshot of the graph and then COMMITTED in id order (core.rs:532-596), how often is a plan
 * still the plan the reference's serial order would have produced?  A plan is a deterministic
 * function of the rows it read, so it is certainly still valid if no earlier commit of the window
 * wrote one of them.  Two refinements are measured:
 *   any   : plan j read a row that a commit k<j of the window wrote (in any way)
 *   hard  : as above, but an APPEND of a new id k to a row r that plan j read is ignored when k
 *           provably changes nothing: in search_level the row was expanded when W's furthest was
 *           already nearer than k (core.rs:657 rejects it, it only joins the visited set), and in
 *           select_neighbors k is not nearer than the m-th selected (core.rs:724-754 never takes it).
 * For every window the program also re-plans each node in true serial order and checks that every
 * plan without a hard conflict is identical to it (the soundness of the rule).
 *
 * Build: gcc -O3 -mavx2 -mfma -ffp-contract=off -o /tmp/spec_conflicts tests/experiments/spec_conflicts.c -lm -lpthread
 */
#include "../../oracle/hnsw_oracle.c"

typedef struct { uint32_t id, lc; simpair bound; int full; } rd;   /* one row read + the accept bound after it */
typedef struct {
    rd *r; uint32_t n, cap;
    uint32_t sel[32][64]; uint32_t nsel[32]; uint32_t top;
} plan_t;

static void rd_push(plan_t *p, uint32_t id, uint32_t lc, simpair b, int full)
{
    if (p->n == p->cap) { p->cap = p->cap ? p->cap * 2 : 1024; p->r = realloc(p->r, p->cap * sizeof(rd)); }
    p->r[p->n].id = id; p->r[p->n].lc = lc; p->r[p->n].bound = b; p->r[p->n].full = full; p->n++;
}

/* search_level with a read log */
static void search_level_log(const hnsw_oracle *o, scratch *s, const float *query, uint32_t ep, uint32_t ef,
                             uint32_t level, plan_t *p)
{
    hnsw_oracle_counters ct = {0, 0, 0};
    visited_reset(s, o->node_count);
    visited_test_and_set(s, ep);
    simpair qpair = { hnsw_oracle_euclidean(query, vec(o, ep), o->dim), ep };
    heap *C = &s->C, *W = &s->W;
    heap_clear(C); heap_clear(W);
    heap_push(C, qpair); heap_push(W, qpair);
    while (C->n) {
        simpair c = heap_pop(C);
        simpair f = heap_peek(W);
        if (nearer(f, c)) break;
        const nrow *nb = row_of(o, c.id, level);
        for (uint32_t i = 0; i < nb->n; i++) {
            uint32_t e = nb->ids[i];
            if (visited_test_and_set(s, e)) continue;
            f = heap_peek(W);
            simpair ep2 = { hnsw_oracle_euclidean(query, vec(o, e), o->dim), e };
            if (nearer(ep2, f) || W->n < ef) {
                heap_push(C, ep2); heap_push(W, ep2);
                if (W->n > ef) heap_pop(W);
            }
        }
        rd_push(p, c.id, level, heap_peek(W), W->n >= ef);
    }
    (void)ct;
}

static void plan_node(hnsw_oracle *o, uint32_t query, plan_t *p)
{
    scratch *s = &o->sc;
    hnsw_oracle_counters ct = {0, 0, 0};
    p->n = 0;
    const float *qv = vec(o, query);
    uint32_t l = o->nodes[query].level, l_max = o->max_layer;
    uint32_t ep = (uint32_t)o->enterpoint;
    uint32_t lc = l_max;
    while (lc > l) {
        search_level_log(o, s, qv, ep, 1, lc, p);
        ep = nearest_of_W(s).id;
        if (lc == 0) break;
        lc--;
    }
    uint32_t top = l_max < l ? l_max : l;
    p->top = top;
    for (uint32_t lcc = top + 1; lcc-- > 0;) {
        search_level_log(o, s, qv, ep, o->ef_construction, lcc, p);
        heap_copy(&s->res, &s->W, 0);
        simpair w_nearest = heap_peek(&s->res);
        select_neighbors(o, s, query, &s->res, o->m, lcc, -1, &s->nbrs, &ct);
        /* select's reads: the row of every member of W; bound = the m-th selected */
        simpair worst = {0, 0}; int full = s->nbrs.n >= o->m;
        for (uint32_t i = 0; i < s->nbrs.n; i++) if (i == 0 || nearer(worst, s->nbrs.a[i])) worst = s->nbrs.a[i];
        for (uint32_t i = 0; i < s->res.n; i++) rd_push(p, s->res.a[i].id, lcc | 0x80000000u, worst, full);
        heap *t = &s->ccopy; heap_copy(t, &s->nbrs, 0);
        p->nsel[lcc] = 0;
        while (t->n) p->sel[lcc][p->nsel[lcc]++] = heap_pop(t).id;
        ep = w_nearest.id;
    }
}

static int plans_equal(const plan_t *a, const plan_t *b)
{
    if (a->top != b->top) return 0;
    for (uint32_t l = 0; l <= a->top; l++) {
        if (a->nsel[l] != b->nsel[l]) return 0;
        if (memcmp(a->sel[l], b->sel[l], a->nsel[l] * 4)) return 0;
    }
    return 1;
}

/* write log of one commit: diff of rows before/after is expensive; instead snapshot degrees+content of rows via hooks:
 * we detect writes by comparing row contents of the touched set before and after the insert */
typedef struct { uint32_t id, lc; int hard; uint32_t appended[64]; uint32_t napp; } wr;

int main(int argc, char **argv)
{
    uint32_t N0 = argc > 1 ? atoi(argv[1]) : 20000, Wn = argc > 2 ? atoi(argv[2]) : 128, nwin = argc > 3 ? atoi(argv[3]) : 8;
    uint32_t dim = argc > 4 ? atoi(argv[4]) : 128, M = argc > 5 ? atoi(argv[5]) : 16, ef = argc > 6 ? atoi(argv[6]) : 200;
    uint32_t total = N0 + Wn * nwin;
    hnsw_oracle *o = hnsw_oracle_new(dim, M, ef, 7);
    float *v = malloc(dim * 4);
    uint64_t x = 12345;
    /* uniform [0,1) vectors */
    #define NEXTV() do { for (uint32_t d_ = 0; d_ < dim; d_++) v[d_] = (float)((splitmix64(&x) >> 40) * (1.0 / 16777216.0)); } while (0)
    for (uint32_t i = 0; i < N0; i++) { NEXTV(); hnsw_oracle_add(o, v, -1, NULL, 0, NULL); if (i % 20000 == 0) fprintf(stderr, "built %u\n", i); }

    plan_t *spec = calloc(Wn, sizeof(plan_t));
    plan_t truth; memset(&truth, 0, sizeof truth);
    double sum_first_any = 0, sum_first_hard = 0, sum_first_true = 0;
    uint64_t nshrink=0,nshrinkw=0; uint64_t why[6]={0,0,0,0,0,0}; uint64_t n_any = 0, n_hard = 0, n_true = 0, n_pairs = 0, unsound = 0, n_nodes = 0, n_any_node = 0, n_hard_node = 0, n_true_node = 0;
    for (uint32_t w = 0; w < nwin; w++) {
        uint32_t first = o->node_count;
        /* store the window's nodes (vector + empty rows), unreachable until committed */
        uint32_t *lv = malloc(Wn * 4);
        for (uint32_t j = 0; j < Wn; j++) { NEXTV(); lv[j] = gen_random_level(o); ensure_cap(o); store_node(o, v, lv[j]); }
        uint32_t lmax0 = o->max_layer;
        for (uint32_t j = 0; j < Wn; j++) plan_node(o, first + j, &spec[j]);
        /* now un-store and insert for real, one at a time, diffing rows */
        /* (rows of nodes >= first are empty; we keep them stored and emulate insert() without store_node) */
        int first_any = -1, first_hard = -1, first_true = -1;
        /* per-row write marks for this window: map (id,lc) -> last hard writer / appended list; use simple arrays */
        uint32_t cap = o->cap;
        /* layer-0 only tables + a small list for upper layers */
        uint8_t *hard0 = calloc(cap, 1);
        uint8_t *soft0 = calloc(cap, 1);
        typedef struct { uint32_t id, lc, k; int hard; } uw;
        uw *uws = NULL; uint32_t nuw = 0, capuw = 0;
        uint32_t (*app0)[8] = calloc(cap, sizeof(uint32_t[8]));
        int closed = 0;
        for (uint32_t j = 0; j < Wn; j++) {
            uint32_t q = first + j;
            /* ground truth plan against the true serial graph */
            plan_node(o, q, &truth);
            int same = plans_equal(&truth, &spec[j]);
            /* conflicts of spec[j] with writes of earlier commits of this window */
            int any = closed, hard = closed;
            const float *qv = vec(o, q);
            for (uint32_t i = 0; i < spec[j].n && !hard; i++) {
                rd *r = &spec[j].r[i];
                uint32_t lc = r->lc & 0x7fffffffu;
                if (lc == 0) {
                    if (hard0[r->id]) { any = 1; hard = 1; why[0]++; break; }
                    if (soft0[r->id]) {
                        any = 1;
                        for (uint32_t a = 0; a < soft0[r->id] && a < 8; a++) {
                            uint32_t k = app0[r->id][a];
                            simpair pk = { hnsw_oracle_euclidean(qv, vec(o, k), o->dim), k };
                            if (!r->full || nearer(pk, r->bound)) { hard = 1; why[(r->lc >> 31) ? 2 : 1]++; if (!r->full) why[3]++; break; }
                        }
                        if (soft0[r->id] > 8) hard = 1;
                    }
                } else {
                    for (uint32_t u = 0; u < nuw; u++) if (uws[u].id == r->id && uws[u].lc == lc) {
                        any = 1;
                        if (uws[u].hard) { hard = 1; why[4]++; break; }
                        simpair pk = { hnsw_oracle_euclidean(qv, vec(o, uws[u].k), o->dim), uws[u].k };
                        if (!r->full || nearer(pk, r->bound)) { hard = 1; why[5]++; break; }
                    }
                }
            }
            if (!hard && !same) unsound++;
            n_nodes++; n_any_node += any; n_hard_node += hard; n_true_node += !same;
            if (any && first_any < 0) first_any = j;
            if (hard && first_hard < 0) first_hard = j;
            if (!same && first_true < 0) first_true = j;
            /* commit for real: snapshot rows that may change = we diff via touched set */
            /* run the reference insert on the already-stored node: emulate insert() body */
            {
                scratch *s = &o->sc; hnsw_oracle_counters *ct = &o->ins;
                uint32_t l = o->nodes[q].level, l_max = o->max_layer;
                uint32_t ep = (uint32_t)o->enterpoint, lc = l_max;
                touch_reset(o);
                while (lc > l) { search_level(o, s, qv, ep, 1, lc, ct); ep = nearest_of_W(s).id; if (lc == 0) break; lc--; }
                uint32_t top = l_max < l ? l_max : l;
                for (uint32_t lcc = top + 1; lcc-- > 0;) {
                    search_level(o, s, qv, ep, o->ef_construction, lcc, ct);
                    heap_copy(&s->res, &s->W, 0);
                    simpair w_nearest = heap_peek(&s->res);
                    select_neighbors(o, s, q, &s->res, o->m, lcc, -1, &s->nbrs, ct);
                    connect_neighbors(o, s, q, &s->nbrs, lcc);
                    /* appends: q appended to each selected row */
                    for (uint32_t i = 0; i < s->nbrs.n; i++) {
                        uint32_t e = s->nbrs.a[i].id;
                        if (lcc == 0) { if (soft0[e] < 8) app0[e][soft0[e]] = q; if (soft0[e] < 255) soft0[e]++; }
                        else { if (nuw == capuw) { capuw = capuw ? capuw * 2 : 256; uws = realloc(uws, capuw * sizeof(uw)); } uws[nuw++] = (uw){e, lcc, q, 0}; }
                    }
                    while (s->nbrs.n) {
                        simpair e = heap_pop(&s->nbrs);
                        heap *econn = &s->econn; heap_clear(econn); econn->furthest_top = 0;
                        const nrow *er = row_of(o, e.id, lcc); const float *ev = vec(o, e.id);
                        for (uint32_t i = 0; i < er->n; i++) { simpair pp = { hnsw_oracle_euclidean(ev, vec(o, er->ids[i]), o->dim), er->ids[i] }; heap_push(econn, pp); }
                        uint32_t m_max = lcc == 0 ? o->m_max0 : o->m_max;
                        if (econn->n > m_max) {
                            /* precise write set of the shrink: e's row, rows that gain e, rows that lose e */
                            uint32_t on = er->n, oldr[512];
                            memcpy(oldr, er->ids, on * 4);
                            select_neighbors(o, s, e.id, econn, m_max, lcc, -1, &s->enew, ct);
                            update_node_connections(o, s, e.id, &s->enew, econn, lcc, -1);
                            const nrow *nr = row_of(o, e.id, lcc);
                            uint32_t wl[1100], nw = 0;
                            wl[nw++] = e.id;
                            for (uint32_t a = 0; a < on; a++) { int f = 0; for (uint32_t b = 0; b < nr->n; b++) f |= nr->ids[b] == oldr[a]; if (!f) wl[nw++] = oldr[a]; }
                            for (uint32_t b = 0; b < nr->n; b++) { int f = 0; for (uint32_t a = 0; a < on; a++) f |= nr->ids[b] == oldr[a]; if (!f) wl[nw++] = nr->ids[b]; }
                            nshrink++; nshrinkw += nw;
                            for (uint32_t t = 0; t < nw; t++) {
                                uint32_t id = wl[t];
                                if (lcc == 0) hard0[id] = 1;
                                else { if (nuw == capuw) { capuw = capuw ? capuw * 2 : 256; uws = realloc(uws, capuw * sizeof(uw)); } uws[nuw++] = (uw){id, lcc, 0, 1}; }
                            }
                        }
                    }
                    ep = w_nearest.id;
                }
                if (l > l_max) { o->max_layer = l; o->enterpoint = q; closed = 1; }
            }
        }
        (void)lmax0;
        sum_first_any += first_any < 0 ? Wn : first_any;
        sum_first_hard += first_hard < 0 ? Wn : first_hard;
        sum_first_true += first_true < 0 ? Wn : first_true;
        free(hard0); free(soft0); free(app0); free(uws); free(lv);
        (void)n_any; (void)n_hard; (void)n_true; (void)n_pairs;
    }
    printf("N0=%u W=%u windows=%u dim=%u M=%u ef=%u\n", N0, Wn, nwin, dim, M, ef);
    printf("  mean first-conflict position: any=%.1f hard=%.1f true-difference=%.1f\n", sum_first_any / nwin, sum_first_hard / nwin, sum_first_true / nwin);
    printf("  per-node rates over the window: any=%.3f hard=%.3f true=%.3f  (nodes=%lu)\n", (double)n_any_node / n_nodes, (double)n_hard_node / n_nodes, (double)n_true_node / n_nodes, (unsigned long)n_nodes);
    printf("  why hard: shrink-row0=%lu search-accept0=%lu select0=%lu (notfull=%lu) upper-shrink=%lu upper-append=%lu\n",(unsigned long)why[0],(unsigned long)why[1],(unsigned long)why[2],(unsigned long)why[3],(unsigned long)why[4],(unsigned long)why[5]);
    printf("  shrinks per insert %.3f, rows written per shrink %.1f\n",(double)nshrink/n_nodes,(double)nshrinkw/(nshrink?nshrink:1));
    printf("  UNSOUND (no hard conflict but plan differs): %lu\n", (unsigned long)unsound);
    return 0;
}
