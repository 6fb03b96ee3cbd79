/*
 * select_head.c -- design experiment (TEST INFRASTRUCTURE, includes the oracle's source).
 *
 * Claim used by the engine's plan kernels (redis_hnsw_amd/csrc/hnsw_insert.hpp, select_head_of_W): right after
 * search_level(ef) (core.rs:607-675), select_neighbors(query, W, m, extend, keep) (core.rs:677-757) returns the
 * m nearest members of W whenever ef >= m or W never filled -- the extension can only add nodes the search
 * already evaluated and found farther than all of W.  This program runs the reference's insert() with the full
 * select_neighbors and counts the calls whose result is NOT the head of W.
 *
 * gcc -O3 -mavx2 -mfma -ffp-contract=off -w -o /tmp/select_head tests/experiments/select_head.c -lm -lpthread
 * /tmp/select_head N dim M ef [dup]      (dup: every vector four times, i.e. heavy ties)
 */
#include "../../oracle/hnsw_oracle.c"
static uint64_t n_calls, n_diff;
static void insert_chk(hnsw_oracle *o, const float *data, uint32_t l)
{
    scratch *s = &o->sc; hnsw_oracle_counters *ct = &o->ins;
    uint32_t l_max = o->max_layer; uint32_t query = store_node(o, data, l); const float *qv = vec(o, query);
    uint32_t ep = (uint32_t)o->enterpoint, lc = l_max;
    while (lc > l) { search_level(o, s, qv, ep, 1, lc, ct); ep = nearest_of_W(s).id; if (lc == 0) break; lc--; }
    uint32_t top = l_max < l ? l_max : l;
    for (uint32_t lcc = top + 1; lcc-- > 0;) {
        search_level(o, s, qv, ep, o->ef_construction, lcc, ct);
        heap_copy(&s->res, &s->W, 0);
        simpair w_nearest = heap_peek(&s->res);
        select_neighbors(o, s, query, &s->res, o->m, lcc, -1, &s->nbrs, ct);
        /* compare with the m nearest of W */
        simpair tmp[2048]; uint32_t nw = s->W.n; memcpy(tmp, s->W.a, nw * sizeof(simpair)); qsort(tmp, nw, sizeof(simpair), cmp_nearer);
        simpair sel[64]; uint32_t ns = s->nbrs.n; memcpy(sel, s->nbrs.a, ns * sizeof(simpair)); qsort(sel, ns, sizeof(simpair), cmp_nearer);
        n_calls++;
        uint32_t want = nw < o->m ? nw : o->m; int d = ns != want;
        for (uint32_t i = 0; i < ns && i < want && !d; i++) d = sel[i].id != tmp[i].id;
        if (nw >= o->m) n_diff += d;     /* only claim it when W holds at least m */
        connect_neighbors(o, s, query, &s->nbrs, lcc);
        while (s->nbrs.n) {
            simpair e = heap_pop(&s->nbrs); heap *econn = &s->econn; heap_clear(econn); econn->furthest_top = 0;
            const nrow *er = row_of(o, e.id, lcc); const float *ev = vec(o, e.id);
            for (uint32_t i = 0; i < er->n; i++) { simpair p = { hnsw_oracle_euclidean(ev, vec(o, er->ids[i]), o->dim), er->ids[i] }; heap_push(econn, p); }
            uint32_t m_max = lcc == 0 ? o->m_max0 : o->m_max;
            if (econn->n > m_max) { select_neighbors(o, s, e.id, econn, m_max, lcc, -1, &s->enew, ct); update_node_connections(o, s, e.id, &s->enew, econn, lcc, -1); }
        }
        ep = w_nearest.id;
    }
    if (l > l_max) { o->max_layer = l; o->enterpoint = query; }
}
int main(int argc, char **argv)
{
    uint32_t N = atoi(argv[1]), dim = atoi(argv[2]), M = atoi(argv[3]), ef = atoi(argv[4]); int dup = argc > 5;
    hnsw_oracle *o = hnsw_oracle_new(dim, M, ef, 7); float *v = malloc(dim * 4); uint64_t x = 99;
    for (uint32_t i = 0; i < N; i++) {
        if (!dup || (i & 3) == 0) for (uint32_t d = 0; d < dim; d++) v[d] = (float)((splitmix64(&x) >> 40) * (1.0 / 16777216.0));
        if (o->node_count == 0) { hnsw_oracle_add(o, v, -1, NULL, 0, NULL); continue; }
        ensure_cap(o); touch_reset(o); insert_chk(o, v, gen_random_level(o));
    }
    printf("N=%u dim=%u M=%u ef=%u%s: select calls %lu, differing from the m nearest of W: %lu\n", N, dim, M, ef, dup ? " (every vector 4x)" : "", (unsigned long)n_calls, (unsigned long)n_diff);
    return n_diff != 0;
}
