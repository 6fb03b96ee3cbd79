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

static int g_diag; static uint32_t g_diag_n0;
static uint64_t why[2][2][2]; static int g_count_why; static int g_refine = 1; static int g_tight = 1; static uint64_t st_hot_hits;
enum { RK_SEARCH = 0, RK_SELECT = 1, RK_SHRINK_NB = 2, RK_SHRINK_ROW = 3 };
typedef struct { uint32_t row, lc; simpair bound; int full, kind, sub; int32_t t; } rd;
typedef struct { uint32_t lc, e; simpair S[64]; uint32_t nS; int valid; } shr;
typedef struct {
    int planned; uint32_t snap, snap_hdr;
    int hot_seen;
    uint32_t hot_end;              /* AHEAD: the plan was made WHILE the commits journalled in [snap, hot_end) were being applied */
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
    t->nr = 0; t->nsh = 0; t->planned = 1; t->snap = nJ; t->snap_hdr = hdr_epoch; t->hot_end = 0; vmap_reset(t);
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
static int validate_range(hnsw_oracle *o, uint32_t q, txn *t, const delta *arr, uint32_t from, uint32_t to, int own_from_valid)
{
    if (t->snap_hdr != hdr_epoch) return 0;
    const float *qv = vec(o, q);
    int ok = 1;
    for (uint32_t ji = from; ji < to; ji++) {
        const delta *d = &arr[ji];
        uint32_t hh = ((d->row * 2654435761u) ^ (d->lc * 40503u)) & (t->hsize - 1);
        for (int32_t i = t->hhead[hh]; i >= 0; i = t->hnext[i]) {
            const rd *r = &t->r[i];
            if (r->row != d->row || r->lc != d->lc) continue;
            if (arr == J && ji < t->hot_end) {
                /* a HOT delta: the row was being rewritten while the plan ran -- the plan may have seen it before, after or
                   torn (two cache lines).  Whatever it saw of that row is void: no distance test */
                if (r->kind == RK_SHRINK_ROW || r->kind == RK_SHRINK_NB) t->sh[r->sub].valid = 0;
                else ok = 0;
                st_hot_hits++;
                continue;
            }
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
            else {
                ok = 0; if (g_count_why) why[r->kind][r->full][d->add]++;
                if (g_diag) fprintf(stderr, "stale kind=%d lc=%u add=%d t=%d/%d rowdeg=%u z_new=%d row_new=%d age=%u\n", r->kind, d->lc, d->add, r->t, t->tclock,
                                    row_of(o, d->row, d->lc)->n, d->z >= g_diag_n0, d->row >= g_diag_n0, ji - from);
            }
        }
    }
    return ok;
}

static int validate(hnsw_oracle *o, uint32_t q, txn *t, uint32_t from, int own_from_valid)
{
    return validate_range(o, q, t, J, from, nJ, own_from_valid);
}

static uint64_t st_spec_applied, st_fallback, st_noshrink_spec_unused, st_commits, st_rounds, st_replans, st_plans;
static uint64_t st_hot_plans, st_hot_killed, st_hot_rec_killed, st_crit_plans, st_rounds_nocrit, st_crit_head;

/* ---- parallel validated commits (PAR=1) -------------------------------------------------------
 * Every window node whose link plan holds runs its WHOLE commit (connect, shrink loop with its own validations and
 * recomputations, core.rs:532-574) against the graph as it stands at the start of the iteration, all of them at once:
 * on the GPU each in a private overlay of the rows it rewrites; here one after the other, every row saved before it
 * is touched and restored afterwards (a `dry` run).  A dry run leaves: its deltas in journal order, the rows it
 * changed (before / after), read-log entries for the select_neighbors it had to recompute, and which speculative
 * records it used.  Node j of the iteration may then commit together with the nodes before it iff, for every earlier
 * node i of the iteration,
 *   (1) no delta of i is relevant to anything j read: its link plan, the speculative records it used, the
 *       recomputations it made (the same relevance rules as for the journal), and
 *   (2) no delta of i is on a row j changed (j's rows are written back whole).
 * The first node that fails ends the group; it is dry-run again in the next iteration, now against a graph that holds
 * the group, which is exactly what the in-order commit wave would have given it.  A node that raises max_layer
 * (core.rs:587-593) closes its group.  The head of an iteration has no predecessor and always commits if its link
 * plan holds, so the scheme makes progress exactly where the in-order wave does. */
typedef struct { uint32_t row, lc, npre, npost; uint32_t *pre, *post; int modified; } rowimg;
typedef struct {
    int ready, promotes;
    delta *d; uint32_t nd;
    rowimg *img; uint32_t nimg, capimg;
    uint32_t nr0, nsh0;            /* the plan's own read log / records end here; beyond: this dry run's recomputations */
    uint8_t live[256];             /* per shrink record: used (speculative result applied, or recomputed here) */
    uint32_t n_spec, n_fallback;
} dry;
static dry *g_dry;                 /* the dry run in progress (NULL: the in-order commit of the serial scheme) */
static txn *g_dry_txn;

static void save_row(hnsw_oracle *o, uint32_t row, uint32_t lc)
{
    dry *d = g_dry;
    if (!d || lc > o->nodes[row].level) return;
    for (uint32_t i = 0; i < d->nimg; i++) if (d->img[i].row == row && d->img[i].lc == lc) return;
    if (d->nimg == d->capimg) { d->capimg = d->capimg ? d->capimg * 2 : 64; d->img = realloc(d->img, d->capimg * sizeof(rowimg)); }
    const nrow *r = row_of(o, row, lc);
    rowimg *im = &d->img[d->nimg++];
    im->row = row; im->lc = lc; im->npre = r->n; im->pre = malloc((r->n + 1) * 4); memcpy(im->pre, r->ids, r->n * 4);
    im->post = NULL; im->npost = 0; im->modified = 0;
}
static void set_row(hnsw_oracle *o, uint32_t row, uint32_t lc, const uint32_t *ids, uint32_t n)
{
    nrow *r = &o->nodes[row].rows[lc];
    if (r->cap < n) { r->cap = n + 8; r->ids = realloc(r->ids, (size_t)r->cap * 4); }
    memcpy(r->ids, ids, n * 4); r->n = n;
}
static void hash_rebuild(txn *t)
{
    uint32_t hs = 1024; while (hs < 2 * t->nr) hs *= 2;
    if (hs > t->hsize) { t->hhead = realloc(t->hhead, hs * 4); t->hsize = hs; }
    t->hnext = realloc(t->hnext, (t->nr + 1) * 4);
    for (uint32_t i = 0; i < t->hsize; i++) t->hhead[i] = -1;
    for (uint32_t i = 0; i < t->nr; i++) { uint32_t h = ((t->r[i].row * 2654435761u) ^ (t->r[i].lc * 40503u)) & (t->hsize - 1); t->hnext[i] = t->hhead[h]; t->hhead[h] = (int32_t)i; }
}

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
    else {
        hnsw_oracle_counters ct = {0,0,0}; uint32_t m_max = lcc == 0 ? o->m_max0 : o->m_max; select_neighbors(o, s, e, econn, m_max, lcc, -1, enew, &ct);
        if (g_dry) {
            /* a recomputation inside a dry run read row e and the rows of its members as they stand (the snapshot plus this
               node's own changes): logged like a speculative record, so that the deltas of the nodes committed alongside can
               be checked against it */
            txn *t = g_dry_txn;
            if (t->nsh == t->capsh) { t->capsh = t->capsh ? t->capsh * 2 : 16; t->sh = realloc(t->sh, t->capsh * sizeof(shr)); }
            if (t->nsh >= 256) { fprintf(stderr, "too many records\n"); abort(); }
            shr *sp = &t->sh[t->nsh];
            sp->lc = lcc; sp->e = e; sp->nS = 0; sp->valid = 1;
            heap *tt = &s->ccopy; heap_copy(tt, enew, 0);
            while (tt->n) sp->S[sp->nS++] = heap_pop(tt);
            simpair worst = sp->S[sp->nS - 1];
            int full = sp->nS >= m_max;
            rd_push(t, e, lcc, worst, 1, RK_SHRINK_ROW, (int)t->nsh);
            for (uint32_t a = 0; a < on; a++) rd_push(t, oldr[a], lcc, worst, full, RK_SHRINK_NB, (int)t->nsh);
            g_dry->live[t->nsh] = 1;
            t->nsh++;
            hash_rebuild(t);
        }
    }
    if (g_dry) {
        save_row(o, e, lcc);
        for (uint32_t a = 0; a < on; a++) save_row(o, oldr[a], lcc);
        for (uint32_t a = 0; a < enew->n; a++) save_row(o, enew->a[a].id, lcc);
    }
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
            save_row(o, q, lcc); save_row(o, e, lcc);
            add_neighbor(o, q, lcc, e); add_neighbor(o, e, lcc, q);
            jpush(e, lcc, q, 1);
        }
        (void)j0;
        for (uint32_t i = 0; i < t->nsel[lcc]; i++) {
            uint32_t e = t->sel[lcc][i].id;
            if (row_of(o, e, lcc)->n <= m_max) continue;
            /* find the speculative shrink, re-validate against everything journalled since the snapshot */
            shr *sp = NULL;
            const uint32_t nplan = g_dry ? g_dry->nsh0 : t->nsh;      /* (records beyond: this dry run's own recomputations) */
            for (uint32_t k = 0; k < nplan; k++) if (t->sh[k].e == e && t->sh[k].lc == lcc) sp = &t->sh[k];
            if (sp) { validate(o, q, t, t->snap, 1); }
            if (sp && sp->valid) { apply_shrink(o, e, lcc, sp->S, sp->nS); if (g_dry) { g_dry->live[sp - t->sh] = 1; g_dry->n_spec++; } else st_spec_applied++; }
            else { apply_shrink(o, e, lcc, NULL, 0); if (g_dry) g_dry->n_fallback++; else st_fallback++; }
            t->snap = t->snap;  /* journal keeps growing; later shrinks are validated against all of it */
        }
        (void)s;
    }
    if (g_dry) { g_dry->promotes = l > l_max; return; }
    if (l > l_max) { o->max_layer = l; o->enterpoint = q; hdr_epoch++; }
    st_commits++;
}

/* the whole commit of node q against the graph as it stands, leaving the graph as it was */
static void dry_run(hnsw_oracle *o, uint32_t q, txn *t, dry *d)
{
    memset(d->live, 0, sizeof d->live);
    d->nd = 0; d->nimg = 0; d->n_spec = d->n_fallback = 0; d->promotes = 0;
    d->nr0 = t->nr; d->nsh0 = t->nsh;
    for (uint32_t k = 0; k < t->nsh; k++) t->sh[k].valid = 1;
    d->ready = t->planned && validate(o, q, t, t->snap, 0);
    if (!d->ready) return;
    const uint32_t nJ0 = nJ;
    g_dry = d; g_dry_txn = t;
    commit_txn(o, q, t);
    g_dry = NULL; g_dry_txn = NULL;
    d->nd = nJ - nJ0;
    d->d = realloc(d->d, (d->nd + 1) * sizeof(delta));
    memcpy(d->d, J + nJ0, d->nd * sizeof(delta));
    nJ = nJ0;
    for (uint32_t i = 0; i < d->nimg; i++) {
        rowimg *im = &d->img[i];
        const nrow *r = row_of(o, im->row, im->lc);
        im->modified = r->n != im->npre || memcmp(r->ids, im->pre, r->n * 4);
        im->npost = r->n; im->post = malloc((r->n + 1) * 4); memcpy(im->post, r->ids, r->n * 4);
        set_row(o, im->row, im->lc, im->pre, im->npre);
    }
}
static void dry_free(txn *t, dry *d, int committed)
{
    for (uint32_t i = 0; i < d->nimg; i++) { free(d->img[i].pre); free(d->img[i].post); }
    d->nimg = 0;
    if (!committed && d->ready && (t->nr != d->nr0 || t->nsh != d->nsh0)) { t->nr = d->nr0; t->nsh = d->nsh0; hash_rebuild(t); }
    /* what the dry run's own deltas did to the records' flags is forgotten with it (the round's validation against
       the journal starts from the plan's snapshot again) */
    if (!committed) for (uint32_t k = 0; k < t->nsh; k++) t->sh[k].valid = 1;
}
static uint64_t st_par_iter, st_par_conf_link, st_par_conf_rec, st_par_conf_row, st_par_conf_promote, st_par_notready;
static int g_par_norowcheck, g_par_noreclog;
/* may node j (dry run dj, transaction tj) commit in the same group as the earlier node whose dry run is di? */
static int par_conflict(hnsw_oracle *o, uint32_t qj, txn *tj, const dry *dj, const dry *di)
{
    for (uint32_t k = 0; k < tj->nsh; k++) tj->sh[k].valid = 1;
    if (g_par_noreclog) { uint32_t keep = tj->nr; tj->nr = dj->nr0; hash_rebuild(tj); int ok0 = validate_range(o, qj, tj, di->d, 0, di->nd, 0); tj->nr = keep; hash_rebuild(tj);
        if (!ok0) { st_par_conf_link++; return 1; } }
    else if (!validate_range(o, qj, tj, di->d, 0, di->nd, 0)) { st_par_conf_link++; return 1; }
    for (uint32_t k = 0; k < tj->nsh; k++) if (dj->live[k] && !tj->sh[k].valid) { st_par_conf_rec++; return 1; }
    if (!g_par_norowcheck)
        for (uint32_t a = 0; a < di->nd; a++)
            for (uint32_t i = 0; i < dj->nimg; i++)
                if (dj->img[i].modified && dj->img[i].row == di->d[a].row && dj->img[i].lc == di->d[a].lc) {
                    st_par_conf_row++;
                    if (getenv("PAR_ROWSTATS")) {
                        /* what did j do to that row?  appended itself only (a selected neighbour that needed no shrink) / more */
                        const rowimg *im = &dj->img[i];
                        int append_only = im->npost == im->npre + 1 && !memcmp(im->pre, im->post, im->npre * 4) && im->post[im->npre] == qj;
                        int i_adds_only = 1; uint32_t i_n = 0;
                        for (uint32_t b = 0; b < di->nd; b++) if (di->d[b].row == im->row && di->d[b].lc == im->lc) { i_n++; i_adds_only &= di->d[b].add; }
                        uint32_t mm = im->lc == 0 ? o->m_max0 : o->m_max;
                        fprintf(stderr, "rowconf lc=%u j_append_only=%d i_adds_only=%d i_n=%u npre=%u mmax=%u room=%d\n", im->lc, append_only, i_adds_only, i_n, im->npre, mm, (int)(im->npre + i_n + 1 <= mm));
                    }
                    return 1;
                }
    return 0;
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
    const char *fix = getenv("FIX");                          /* a graph dumped by tests/experiments/dump_fixture.py instead of a build */
    hnsw_oracle *A = fix ? NULL : hnsw_oracle_new(dim, M, ef, 7);
    uint32_t total = N0 + K;
    float *V = malloc((size_t)total * dim * 4);
    uint64_t x = 12345;
    for (size_t i = 0; i < (size_t)total * dim; i++) V[i] = (float)((splitmix64(&x) >> 40) * (1.0 / 16777216.0));
    uint32_t *lv = malloc(total * 4);
    hnsw_oracle *B = NULL;
    if (fix) {
        /* <fix>/meta.txt: nodes layers enterpoint; vec.f32, levels.u32, rp<l>.u64, col<l>.u32 */
        char path[512]; uint32_t fn, fl; long long fep;
        snprintf(path, sizeof path, "%s/meta.txt", fix); FILE *f = fopen(path, "r");
        if (!f || fscanf(f, "%u %u %lld", &fn, &fl, &fep) != 3 || fn != N0) { fprintf(stderr, "bad fixture (N0 must be %u)\n", fn); return 2; }
        fclose(f);
        snprintf(path, sizeof path, "%s/vec.f32", fix); f = fopen(path, "rb"); if (fread(V, 4, (size_t)N0 * dim, f) != (size_t)N0 * dim) return 2; fclose(f);
        uint32_t *lev0 = malloc((size_t)N0 * 4);
        snprintf(path, sizeof path, "%s/levels.u32", fix); f = fopen(path, "rb"); if (fread(lev0, 4, N0, f) != N0) return 2; fclose(f);
        uint64_t **rp0 = malloc(fl * sizeof *rp0); uint32_t **cl0 = malloc(fl * sizeof *cl0);
        for (uint32_t l = 0; l < fl; l++) {
            rp0[l] = malloc(((size_t)N0 + 1) * 8);
            snprintf(path, sizeof path, "%s/rp%u.u64", fix, l); f = fopen(path, "rb"); if (fread(rp0[l], 8, (size_t)N0 + 1, f) != (size_t)N0 + 1) return 2; fclose(f);
            cl0[l] = malloc((rp0[l][N0] + 1) * 4);
            snprintf(path, sizeof path, "%s/col%u.u32", fix, l); f = fopen(path, "rb"); if (fread(cl0[l], 4, rp0[l][N0], f) != rp0[l][N0]) return 2; fclose(f);
        }
        A = hnsw_oracle_import(dim, M, ef, N0, V, lev0, fep, fl, (const uint64_t *const *)rp0, (const uint32_t *const *)cl0);
        B = hnsw_oracle_import(dim, M, ef, N0, V, lev0, fep, fl, (const uint64_t *const *)rp0, (const uint32_t *const *)cl0);
        for (uint32_t i = 0; i < N0; i++) lv[i] = lev0[i];
        A->rng[0] ^= 0x1234; 
        for (uint32_t i = N0; i < total; i++) lv[i] = gen_random_level(A);
        fprintf(stderr, "fixture loaded\n");
    } else
    for (uint32_t i = 0; i < total; i++) lv[i] = gen_random_level(A);
    for (uint32_t i = 0; i < (fix ? 0 : N0); i++) { hnsw_oracle_add(A, V + (size_t)i * dim, (int32_t)lv[i], NULL, 0, NULL); if (i % 20000 == 0) fprintf(stderr, "built %u\n", i); }
    /* clone A -> B through export/import */
    if (!fix) {
    uint32_t L = A->max_layer + 1;
    uint64_t **rp = malloc(L * sizeof *rp); uint32_t **cl = malloc(L * sizeof *cl);
    uint32_t *lev = malloc(N0 * 4); hnsw_oracle_export_levels(A, lev);
    for (uint32_t l = 0; l < L; l++) { rp[l] = malloc(((size_t)N0 + 1) * 8); cl[l] = malloc((hnsw_oracle_layer_nnz(A, l) + 1) * 4); hnsw_oracle_export_layer(A, l, rp[l], cl[l]); }
    B = hnsw_oracle_import(dim, M, ef, N0, A->data, lev, A->enterpoint, L, (const uint64_t *const *)rp, (const uint32_t *const *)cl);
    }
    /* reference: plain serial inserts on A */
    for (uint32_t i = N0; i < total; i++) hnsw_oracle_add(A, V + (size_t)i * dim, (int32_t)lv[i], NULL, 0, NULL);
    fprintf(stderr, "reference done\n");

    /* model on B: store all K nodes (unlinked) up front */
    for (uint32_t i = N0; i < total; i++) { ensure_cap(B); store_node(B, V + (size_t)i * dim, lv[i]); }
    /* node_count now counts unlinked nodes too; they are unreachable, visited_reset sizes by node_count: fine */
    txn *T = calloc(K, sizeof(txn));
    uint32_t head = 0;
    uint64_t run_hist[8] = {0}, grp_hist[8] = {0};
    const int g_par = getenv("PAR") != NULL;                 /* commits in validated parallel groups instead of one by one */
    g_par_norowcheck = getenv("PAR_NOROWCHECK") != NULL;     /* (unsound on purpose: drops rule 2) */
    g_par_noreclog = getenv("PAR_NORECLOG") != NULL;         /* (unsound on purpose: recomputations inside a dry run are not logged) */
    dry *D = calloc(Wn + 1, sizeof(dry));
    /* AHEAD=f: the nodes that will ENTER the window next round are planned while this round's commits are
       being applied (on the GPU: a second stream).  Such a plan saw every row in some state between "before the round's
       commits" and "after" -- possibly torn -- so its snapshot is the journal position at the START of the commits, and
       every delta journalled during them that sits on a row it read voids that read outright (validate_range, hot rule).
       Modelled here by planning them after the round's FIRST group has been applied (a state in between). */
    const double g_ahead = getenv("AHEAD") ? atof(getenv("AHEAD")) : 0.0;
    double yield_ema = 4.0;
    while (head < K) {
        st_rounds++;
        uint32_t wend = head + Wn < K ? head + Wn : K;
        /* (re)plan everything in the window that has no valid plan */
        uint32_t crit = 0;
        for (uint32_t j = head; j < wend; j++) {
            if (T[j].planned) {
                int ok = validate(B, N0 + j, &T[j], T[j].snap, 0);
                int shok = 1; for (uint32_t k = 0; k < T[j].nsh; k++) shok &= T[j].sh[k].valid;
                if (T[j].hot_end && !T[j].hot_seen) { T[j].hot_seen = 1; if (!ok) st_hot_killed++; else if (!shok) st_hot_rec_killed++; }
                if (ok && shok) continue;
                st_replans++;
            }
            plan_txn(B, N0 + j, &T[j]); st_plans++; crit++;
            if (j == head) st_crit_head++;
        }
        st_crit_plans += crit;
        if (!crit) st_rounds_nocrit++;
        const uint32_t j0 = nJ;                            /* journal position at the start of this round's commits */
        uint32_t hot_lo = wend, hot_hi = wend;
        int hot_done = g_ahead <= 0.0;
        uint32_t run = 0;
        while (g_par && head < wend) {
            /* one iteration of the parallel commit: dry runs of the whole window, then the longest conflict-free group */
            uint32_t n = wend - head, nd = 0, pfx = 0;
            st_par_iter++;
            for (uint32_t b = 0; b < n; b++) { dry_run(B, N0 + head + b, &T[head + b], &D[b]); nd++; if (!D[b].ready) break; }
            if (getenv("DIAG") && !D[0].ready && T[head].planned) { g_diag = 1; g_diag_n0 = N0; validate(B, N0 + head, &T[head], T[head].snap, 0); g_diag = 0; }
            for (uint32_t b = 0; b < nd; b++) {
                if (!D[b].ready) { if (b) st_par_notready++; break; }
                int conflict = 0;
                for (uint32_t i = 0; i < b && !conflict; i++) conflict = par_conflict(B, N0 + head + b, &T[head + b], &D[b], &D[i]);
                if (conflict) break;
                pfx++;
                if (D[b].promotes) { st_par_conf_promote++; break; }
            }
            for (uint32_t b = 0; b < pfx; b++) {
                dry *d = &D[b];
                for (uint32_t i = 0; i < d->nimg; i++) if (d->img[i].modified) set_row(B, d->img[i].row, d->img[i].lc, d->img[i].post, d->img[i].npost);
                for (uint32_t a = 0; a < d->nd; a++) jpush(d->d[a].row, d->d[a].lc, d->d[a].z, d->d[a].add);
                if (d->promotes) { B->max_layer = B->nodes[N0 + head + b].level; B->enterpoint = N0 + head + b; hdr_epoch++; }
                st_spec_applied += d->n_spec; st_fallback += d->n_fallback; st_commits++;
            }
            for (uint32_t b = 0; b < nd; b++) dry_free(&T[head + b], &D[b], b < pfx);
            if (!hot_done && pfx) {
                /* the look-ahead plans, made against the graph as it stands in the middle of the round's commits */
                hot_done = 1;
                uint32_t X = (uint32_t)(yield_ema * g_ahead) + 2;
                hot_hi = wend + X < K ? wend + X : K;
                for (uint32_t j = hot_lo; j < hot_hi; j++) {
                    if (T[j].planned) continue;
                    plan_txn(B, N0 + j, &T[j]); st_plans++; st_hot_plans++;
                    T[j].snap = j0; T[j].hot_end = 0xFFFFFFFFu; T[j].hot_seen = 0;     /* (the end is set when the commits are over) */
                }
            }
            if (!pfx) break;
            head += pfx; run += pfx;
            { int gb = pfx <= 1 ? 0 : pfx <= 2 ? 1 : pfx <= 4 ? 2 : pfx <= 8 ? 3 : pfx <= 16 ? 4 : pfx <= 32 ? 5 : pfx <= 64 ? 6 : 7; grp_hist[gb]++; }
        }
        while (!g_par && head < wend) {
            txn *t = &T[head];
            for (uint32_t k = 0; k < t->nsh; k++) t->sh[k].valid = 1;
            g_count_why = 1; int vok = validate(B, N0 + head, t, t->snap, 0); g_count_why = 0;
            if (!vok) break;
            commit_txn(B, N0 + head, t);
            head++; run++;
        }
        for (uint32_t j = hot_lo; j < hot_hi; j++) if (T[j].hot_end == 0xFFFFFFFFu) T[j].hot_end = nJ;
        yield_ema = 0.8 * yield_ema + 0.2 * run;
        int b = run <= 1 ? 0 : run <= 2 ? 1 : run <= 4 ? 2 : run <= 8 ? 3 : run <= 16 ? 4 : run <= 32 ? 5 : run <= 64 ? 6 : 7;
        run_hist[b]++;
    }
    int same = rows_equal(A, B);
    printf("N0=%u K=%u W=%u dim=%u M=%u ef=%u : graphs %s\n", N0, K, Wn, dim, M, ef, same ? "IDENTICAL" : "DIFFER");
    printf("  rounds=%lu commits/round=%.1f plans=%lu (replans %lu) plans/commit=%.2f\n", (unsigned long)st_rounds, (double)K / st_rounds, (unsigned long)st_plans, (unsigned long)st_replans, (double)st_plans / K);
    printf("  shrinks/commit=%.2f  speculative applied=%.3f fallback=%.3f\n", (double)(st_spec_applied + st_fallback) / K, (double)st_spec_applied / (st_spec_applied + st_fallback + 1e-9), (double)st_fallback / (st_spec_applied + st_fallback + 1e-9));
    printf("  journal deltas/commit=%.1f\n", (double)nJ / K);
    printf("  run-length histogram (<=1,2,4,8,16,32,64,>64):"); for (int i = 0; i < 8; i++) printf(" %lu", (unsigned long)run_hist[i]); printf("\n");
    if (g_par) {
        printf("  parallel commits: %.2f group commits per round, %.2f nodes per group; groups closed by: stale link plan %lu, a record used %lu, a row changed %lu, max_layer raised %lu, next plan not ready %lu\n",
               (double)st_par_iter / st_rounds, (double)K / (st_par_iter ? st_par_iter : 1), (unsigned long)st_par_conf_link, (unsigned long)st_par_conf_rec,
               (unsigned long)st_par_conf_row, (unsigned long)st_par_conf_promote, (unsigned long)st_par_notready);
        printf("  group-size histogram (<=1,2,4,8,16,32,64,>64):"); for (int i = 0; i < 8; i++) printf(" %lu", (unsigned long)grp_hist[i]); printf("\n");
    }
    if (g_ahead > 0.0)
        printf("  look-ahead plans (made during the commits): %lu, of which voided by a hot delta on a row they read: link plan %lu, a record only %lu; plans on the critical path: %.2f per round (%.1f %% of the rounds none; the head itself in %.1f %%)\n",
               (unsigned long)st_hot_plans, (unsigned long)st_hot_killed, (unsigned long)st_hot_rec_killed, (double)st_crit_plans / st_rounds,
               100.0 * st_rounds_nocrit / st_rounds, 100.0 * st_crit_head / st_rounds);
    else
        printf("  plans on the critical path: %.2f per round (%.1f %% of the rounds none; the head itself in %.1f %%)\n", (double)st_crit_plans / st_rounds,
               100.0 * st_rounds_nocrit / st_rounds, 100.0 * st_crit_head / st_rounds);
    printf("  head-invalid reasons [kind][full][add]: search nf- %lu nf+ %lu f- %lu f+ %lu | select nf- %lu nf+ %lu f- %lu f+ %lu\n", (unsigned long)why[0][0][0],(unsigned long)why[0][0][1],(unsigned long)why[0][1][0],(unsigned long)why[0][1][1],(unsigned long)why[1][0][0],(unsigned long)why[1][0][1],(unsigned long)why[1][1][0],(unsigned long)why[1][1][1]);
    /* predicted build rate: round overhead 1.5 ms, commit 15 us, spec shrink 2 us, fallback 52 us */
    double tsec = st_rounds * 1.5e-3 + K * 15e-6 + st_spec_applied * 2e-6 + st_fallback * 52e-6;
    printf("  predicted %.0f inserts/s (1.5 ms/round, 15 us/commit, 2 us/spec shrink, 52 us/fallback)\n", K / tsec);
    return same ? 0 : 1;
}
