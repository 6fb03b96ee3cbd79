/*
 * del_model.c -- CPU model of HNSW.NODE.DEL with its re-selections SPECULATED (TEST INFRASTRUCTURE; includes
 * the oracle's source).  It states the rules hnsw_occ.hpp's k_occ_del_list / k_occ_shrinks / k_occ_del_commit
 * implement and checks that they reproduce the reference's serial delete (core.rs:414-475, 824-863) exactly.
 *
 * The reference walks the deleted node's neighbours n in stored order (:829); each runs
 *     nconn = its row with similarities (:832-844);  new = select_neighbors(n, nconn, m_max, ignored = node) (:853);
 *     update_node_connections(n, new, nconn, ignored = node) (:856)
 * one after the other, every step seeing the rows the earlier ones rewrote.  The model computes ALL the
 * select_neighbors against the graph as it stands before the delete (what one wave per neighbour does on the
 * GPU), recording per neighbour what it read -- n's own row, and the rows of that row's members with the key of
 * the last selected id -- and then applies them in the reference's order.  Every row change an applied step makes
 * is journalled as (row, layer, id, added | removed); before a speculative result is used, the journal so far is
 * checked against its reads:
 *     n's own row changed                                   -> stale
 *     a member's row gained / lost id z (z != n, z != node) -> stale iff the selection did not fill its list
 *                                                              (anything that appears would be selected) or
 *                                                              key(n, z) <= key of the last selected
 * A stale result is recomputed on the spot, on the current graph.  An id farther than the last selected cannot
 * enter the top-m_max of the pool (select_neighbors returns exactly that: core.rs:724-754), and a
 * farther id that vanishes was not in it.
 *
 * gcc -O2 -mavx2 -mfma -ffp-contract=off -o /tmp/del_model tests/experiments/del_model.c -lm -lpthread
 * /tmp/del_model N K dim M ef [norule]     (norule: drop the "did not fill its list" rule -- the model must then FAIL
 *                                           on small m, which is how the rule was found on the GPU)
 */
#include "../../oracle/hnsw_oracle.c"

typedef struct { uint32_t row, lc, z; int add; } delta;
static delta *J; static uint32_t nJ, capJ;
static void jpush(uint32_t row, uint32_t lc, uint32_t z, int add)
{
    if (nJ == capJ) { capJ = capJ ? capJ * 2 : 4096; J = realloc(J, capJ * sizeof(delta)); }
    J[nJ++] = (delta){row, lc, z, add};
}

typedef struct {
    uint32_t lc, n;
    uint32_t *members; uint32_t nmem;          /* n's row when the result was computed */
    simpair *S; uint32_t nS;                   /* the speculative selection, as popped from the result heap */
    simpair bound; int full;
} spec;

static uint64_t st_used, st_recomputed, st_subs;
static int g_norule;

static void nconn_of(hnsw_oracle *o, uint32_t n, uint32_t lc, heap *nconn)
{
    heap_clear(nconn); nconn->furthest_top = 0;
    const nrow *r = row_of(o, n, lc);
    const float *nv = vec(o, n);
    for (uint32_t i = 0; i < r->n; i++) {
        simpair p = { hnsw_oracle_euclidean(nv, vec(o, r->ids[i]), o->dim), r->ids[i] };
        heap_push(nconn, p);
    }
}

/* the reference's delete with speculated re-selections; same tail as hnsw_oracle_delete */
static void model_delete(hnsw_oracle *o, uint32_t id)
{
    scratch *s = &o->sc;
    hnsw_oracle_counters ct = {0, 0, 0};
    uint32_t top = o->nodes[id].level;
    /* ---- speculate every re-selection on the graph as it stands ---- */
    uint32_t nsub = 0;
    for (uint32_t lc = 0; lc <= top; lc++) nsub += row_of(o, id, lc)->n;
    spec *sp = calloc(nsub ? nsub : 1, sizeof(spec));
    uint32_t k = 0;
    for (uint32_t lc = 0; lc <= top; lc++) {
        const nrow *dr = row_of(o, id, lc);
        uint32_t m_max = lc == 0 ? o->m_max0 : o->m_max;
        for (uint32_t kk = 0; kk < dr->n; kk++, k++) {
            uint32_t n = dr->ids[kk];
            spec *x = &sp[k];
            x->lc = lc; x->n = n;
            const nrow *r = row_of(o, n, lc);
            x->nmem = r->n; x->members = malloc((size_t)(r->n ? r->n : 1) * 4);
            memcpy(x->members, r->ids, (size_t)r->n * 4);
            nconn_of(o, n, lc, &s->econn);
            select_neighbors(o, s, n, &s->econn, m_max, lc, (int64_t)id, &s->enew, &ct);
            x->nS = s->enew.n; x->S = malloc((size_t)(x->nS ? x->nS : 1) * sizeof(simpair));
            memcpy(x->S, s->enew.a, (size_t)x->nS * sizeof(simpair));
            x->full = x->nS >= m_max;
            x->bound = x->nS ? x->S[0] : (simpair){0.f, 0};
            for (uint32_t i = 1; i < x->nS; i++) if (nearer(x->bound, x->S[i])) x->bound = x->S[i];   /* the farthest selected */
        }
    }
    /* ---- apply in the reference's order, validating against the delete's own journal ---- */
    nJ = 0;
    k = 0;
    touch_reset(o);
    for (uint32_t lc = 0; lc <= top; lc++) {
        const nrow *dr = row_of(o, id, lc);                 /* not modified during the walk (node is ignored) */
        uint32_t m_max = lc == 0 ? o->m_max0 : o->m_max;
        for (uint32_t kk = 0; kk < dr->n; kk++, k++) {
            spec *x = &sp[k];
            uint32_t n = x->n;
            int stale = 0;
            for (uint32_t j = 0; j < nJ && !stale; j++) {
                const delta *d = &J[j];
                if (d->lc != lc) continue;
                if (d->row == n) { stale = 1; break; }
                int member = 0;
                for (uint32_t i = 0; i < x->nmem; i++) member |= x->members[i] == d->row;
                if (!member || d->z == n || d->z == id) continue;
                if (!x->full && !g_norule) { stale = 1; break; }
                simpair pk = { hnsw_oracle_euclidean(vec(o, n), vec(o, d->z), o->dim), d->z };
                if (!nearer(x->bound, pk)) stale = 1;        /* key(n, z) <= key of the last selected */
            }
            st_subs++;
            /* the row before, for the journal */
            const nrow *r0 = row_of(o, n, lc);
            uint32_t on = r0->n, *oldr = malloc((size_t)(on ? on : 1) * 4);
            memcpy(oldr, r0->ids, (size_t)on * 4);
            nconn_of(o, n, lc, &s->econn);                   /* update_node_connections wants the old row with sims */
            heap *enew = &s->enew;
            if (stale) {
                select_neighbors(o, s, n, &s->econn, m_max, lc, (int64_t)id, enew, &ct);
                st_recomputed++;
            } else {
                heap_clear(enew); enew->furthest_top = 0;
                for (uint32_t i = 0; i < x->nS; i++) heap_push(enew, x->S[i]);
                st_used++;
            }
            touch_add(o, n);
            update_node_connections(o, s, n, enew, &s->econn, lc, (int64_t)id);
            const nrow *r1 = row_of(o, n, lc);
            for (uint32_t a = 0; a < on; a++) {
                int f = 0; for (uint32_t b = 0; b < r1->n; b++) f |= r1->ids[b] == oldr[a];
                if (!f && oldr[a] != id) { jpush(n, lc, oldr[a], 0); jpush(oldr[a], lc, n, 0); }   /* the node's own rows are left alone (:810-813) */
            }
            for (uint32_t b = 0; b < r1->n; b++) {
                int f = 0; for (uint32_t a = 0; a < on; a++) f |= r1->ids[b] == oldr[a];
                if (!f) { jpush(n, lc, r1->ids[b], 1); jpush(r1->ids[b], lc, n, 1); }
            }
            free(oldr);
        }
    }
    for (uint32_t i = 0; i < nsub; i++) { free(sp[i].members); free(sp[i].S); }
    free(sp);
    /* ---- the tail of hnsw_oracle_delete (core.rs:419-472) ---- */
    o->dead[id] = 1;
    o->n_dead++;
    for (uint32_t lc = 0; lc <= top; lc++) o->nodes[id].rows[lc].n = 0;
    if (o->enterpoint == (int64_t)id) {
        int64_t best = -1; uint32_t best_level = 0;
        for (uint32_t i = 0; i < o->node_count; i++)
            if (!o->dead[i] && (best < 0 || o->nodes[i].level > best_level)) { best = i; best_level = o->nodes[i].level; }
        o->enterpoint = best;
        o->max_layer = best >= 0 ? best_level : 0;
    }
}

static int rows_equal(const hnsw_oracle *a, const hnsw_oracle *b)
{
    if (a->node_count != b->node_count || a->enterpoint != b->enterpoint || a->max_layer != b->max_layer) return 0;
    for (uint32_t i = 0; i < a->node_count; i++) {
        if (a->nodes[i].level != b->nodes[i].level || a->dead[i] != b->dead[i]) return 0;
        for (uint32_t l = 0; l <= a->nodes[i].level; l++) {
            const nrow *ra = &a->nodes[i].rows[l], *rb = &b->nodes[i].rows[l];
            if (ra->n != rb->n || memcmp(ra->ids, rb->ids, (size_t)ra->n * 4)) { fprintf(stderr, "row %u L%u differs\n", i, l); return 0; }
        }
    }
    return 1;
}

int main(int argc, char **argv)
{
    uint32_t N = argc > 1 ? atoi(argv[1]) : 1500, K = argc > 2 ? atoi(argv[2]) : 300;
    uint32_t dim = argc > 3 ? atoi(argv[3]) : 32, M = argc > 4 ? atoi(argv[4]) : 8, ef = argc > 5 ? atoi(argv[5]) : 40;
    g_norule = argc > 6 && !strcmp(argv[6], "norule");
    uint64_t st = 12345;
    float *V = malloc((size_t)(N + K) * dim * 4);
    for (size_t i = 0; i < (size_t)(N + K) * dim; i++) V[i] = (float)((splitmix64(&st) >> 40) / 16777216.0);
    hnsw_oracle *A = hnsw_oracle_new(dim, M, ef, 7), *B = hnsw_oracle_new(dim, M, ef, 7);   /* same seed: same drawn levels */
    for (uint32_t i = 0; i < N; i++) {
        hnsw_oracle_add(A, V + (size_t)i * dim, -1, NULL, 0, NULL);
        hnsw_oracle_add(B, V + (size_t)i * dim, -1, NULL, 0, NULL);
    }
    if (!rows_equal(A, B)) { printf("setup: graphs DIFFER\n"); return 2; }
    uint32_t done = 0, checked = 0, next_add = N;
    int same = 1;
    for (uint32_t t = 0; t < K && same; t++) {
        /* mostly deletes (sometimes the enterpoint), now and then an insert so that rows refill */
        uint64_t r = splitmix64(&st);
        if (r % 5 == 4) {
            hnsw_oracle_add(A, V + (size_t)next_add * dim, -1, NULL, 0, NULL);
            hnsw_oracle_add(B, V + (size_t)next_add * dim, -1, NULL, 0, NULL);
            next_add++;
            continue;
        }
        uint32_t id = (r % 7 == 0 && A->enterpoint >= 0) ? (uint32_t)A->enterpoint : (uint32_t)((r >> 8) % A->node_count);
        if (A->dead[id] || A->node_count - A->n_dead < 8) continue;
        hnsw_oracle_delete(A, id, NULL, 0, NULL);            /* the reference's serial order */
        model_delete(B, id);                                 /* speculated, validated, applied in order */
        done++;
        if (t % 8 == 0 || t + 1 == K) { same = rows_equal(A, B); checked++; }
    }
    same = same && rows_equal(A, B);
    printf("N=%u ops=%u dim=%u M=%u ef=%u%s : %u deletes, graphs %s\n", N, K, dim, M, ef, g_norule ? " [rule off]" : "", done,
           same ? "IDENTICAL" : "DIFFER");
    printf("  re-selections %lu: speculative result used %.3f, recomputed %.3f\n", (unsigned long)st_subs,
           (double)st_used / (st_subs + 1e-9), (double)st_recomputed / (st_subs + 1e-9));
    return same ? 0 : 1;
}
