/*
 * visit_skew.c -- how concentrated are the vector reads of a batch of HNSW.SEARCHes?  (TEST INFRASTRUCTURE; includes
 * the oracle's source.)  Runs B queries on a graph dumped by tests/experiments/dump_fixture.py and histograms the
 * metric evaluations (core.rs:621, :652) per node: "the hottest X % of the nodes receive Y % of the evaluations".
 * Why it matters: roofline.achieved counts every evaluation's 4 x dim bytes; a node evaluated by many queries of the
 * launches in flight is served from the 256 MB Infinity Cache after the first time, which is how the search kernel's
 * algorithmic rate can exceed what HBM delivers for UNIFORM random gathers (profiles/r5_gather_bw.txt).
 *
 * gcc -O3 -mavx2 -mfma -ffp-contract=off -w -o /tmp/visit_skew tests/experiments/visit_skew.c -lm -lpthread
 * /tmp/visit_skew <fixture dir> <queries> [dim M ef]
 */
#include "../../oracle/hnsw_oracle.c"

static uint32_t *g_hits;
static int cmp_desc(const void *a, const void *b) { uint32_t x = *(const uint32_t *)a, y = *(const uint32_t *)b; return x < y ? 1 : x > y ? -1 : 0; }

/* search_level (core.rs:607-675) with a counter per evaluated node */
static void search_level_count(const hnsw_oracle *o, scratch *s, const float *query, uint32_t ep, uint32_t ef, uint32_t level)
{
    visited_reset(s, o->node_count);
    visited_test_and_set(s, ep);
    simpair qpair = { hnsw_oracle_euclidean(query, vec(o, ep), o->dim), ep };
    g_hits[ep]++;
    heap *C = &s->C, *W = &s->W;
    heap_clear(C); heap_clear(W);
    heap_push(C, qpair); heap_push(W, qpair);
    while (C->n) {
        simpair c = heap_pop(C), f = heap_peek(W);
        if (stop_test(c, f)) break;
        const nrow *nb = row_of(o, c.id, level);
        for (uint32_t i = 0; i < nb->n; i++) {
            uint32_t e = nb->ids[i];
            if (visited_test_and_set(s, e)) continue;
            f = heap_peek(W);
            simpair e2 = { hnsw_oracle_euclidean(query, vec(o, e), o->dim), e };
            g_hits[e]++;
            if (accept_test(e2, f) || W->n < ef) { heap_push(C, e2); heap_push(W, e2); if (W->n > ef) heap_pop(W); }
        }
    }
}

int main(int argc, char **argv)
{
    if (argc < 3) { fprintf(stderr, "usage: visit_skew <fixture dir> <queries> [dim M ef]\n"); return 2; }
    const char *fix = argv[1];
    uint32_t B = atoi(argv[2]), dim = argc > 3 ? atoi(argv[3]) : 128, M = argc > 4 ? atoi(argv[4]) : 16, ef = argc > 5 ? atoi(argv[5]) : 200;
    char path[512]; uint32_t N, L; long long ep0;
    snprintf(path, sizeof path, "%s/meta.txt", fix); FILE *f = fopen(path, "r");
    if (!f || fscanf(f, "%u %u %lld", &N, &L, &ep0) != 3) return 2;
    fclose(f);
    float *V = malloc((size_t)N * dim * 4);
    snprintf(path, sizeof path, "%s/vec.f32", fix); f = fopen(path, "rb"); if (fread(V, 4, (size_t)N * dim, f) != (size_t)N * dim) return 2; fclose(f);
    uint32_t *lev = malloc((size_t)N * 4);
    snprintf(path, sizeof path, "%s/levels.u32", fix); f = fopen(path, "rb"); if (fread(lev, 4, N, f) != N) return 2; fclose(f);
    uint64_t **rp = malloc(L * sizeof *rp); uint32_t **cl = malloc(L * sizeof *cl);
    for (uint32_t l = 0; l < L; l++) {
        rp[l] = malloc(((size_t)N + 1) * 8);
        snprintf(path, sizeof path, "%s/rp%u.u64", fix, l); f = fopen(path, "rb"); if (fread(rp[l], 8, (size_t)N + 1, f) != (size_t)N + 1) return 2; fclose(f);
        cl[l] = malloc((rp[l][N] + 1) * 4);
        snprintf(path, sizeof path, "%s/col%u.u32", fix, l); f = fopen(path, "rb"); if (fread(cl[l], 4, rp[l][N], f) != rp[l][N]) return 2; fclose(f);
    }
    hnsw_oracle *o = hnsw_oracle_import(dim, M, ef, N, V, lev, ep0, L, (const uint64_t *const *)rp, (const uint32_t *const *)cl);
    g_hits = calloc(N, 4);
    uint64_t x = 777;
    float *q = malloc(dim * 4);
    for (uint32_t b = 0; b < B; b++) {
        for (uint32_t j = 0; j < dim; j++) q[j] = (float)((splitmix64(&x) >> 40) * (1.0 / 16777216.0));
        uint32_t ep = (uint32_t)o->enterpoint, lc = o->max_layer;
        while (lc > 0) { search_level_count(o, &o->sc, q, ep, 1, lc); ep = nearest_of_W(&o->sc).id; lc--; }
        search_level_count(o, &o->sc, q, ep, ef, 0);
    }
    uint64_t total = 0, distinct = 0;
    for (uint32_t i = 0; i < N; i++) { total += g_hits[i]; distinct += g_hits[i] != 0; }
    qsort(g_hits, N, 4, cmp_desc);
    printf("%u queries on %u nodes (dim %u, M %u, ef %u): %.1f evaluations per query, %lu distinct nodes evaluated (%.1f %% of the index)\n",
           B, N, dim, M, ef, (double)total / B, (unsigned long)distinct, 100.0 * distinct / N);
    const double fr[] = { 0.001, 0.01, 0.05, 0.10, 0.25, 0.50 };
    uint64_t run = 0; uint32_t fi = 0;
    for (uint32_t i = 0; i < N && fi < 6; i++) {
        run += g_hits[i];
        if (i + 1 >= (uint32_t)(fr[fi] * N)) { printf("  the hottest %5.1f %% of the nodes (%8u vectors, %7.1f MB) receive %5.1f %% of the evaluations\n", 100 * fr[fi], i + 1, (double)(i + 1) * dim * 4 / 1e6, 100.0 * run / total); fi++; }
    }
    return 0;
}
