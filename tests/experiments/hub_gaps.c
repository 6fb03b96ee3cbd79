/* hub_gaps.c -- design experiment (TEST INFRASTRUCTURE): how long before a shrink (core.rs:560-569)
 * of row e was row e last written?  Decides whether shrinks can be planned ahead of their commit.
 * gcc -O3 -mavx2 -mfma -ffp-contract=off -o /tmp/hub_gaps tests/experiments/hub_gaps.c -lm -lpthread */
#include "../../oracle/hnsw_oracle.c"
int main(int argc, char **argv)
{
    uint32_t N = argc > 1 ? atoi(argv[1]) : 20000, dim = 128, M = 16, ef = 200;
    hnsw_oracle *o = hnsw_oracle_new(dim, M, ef, 7);
    float *v = malloc(dim * 4);
    uint64_t x = 12345;
    uint32_t *last = calloc(N, 4);      /* last insert that wrote layer-0 row */
    uint32_t *deg_before = malloc(N * 4);
    uint64_t hist[8] = {0}; uint64_t nshr = 0; uint64_t tot_touch=0;
    uint32_t *tb = malloc(65536 * 4);
    uint64_t sel_count_cap = N; uint32_t *selcnt = calloc(sel_count_cap, 4);
    for (uint32_t i = 0; i < N; i++) {
        for (uint32_t d = 0; d < dim; d++) v[d] = (float)((splitmix64(&x) >> 40) * (1.0 / 16777216.0));
        uint32_t nt = 0;
        hnsw_oracle_add(o, v, -1, tb, 65536, &nt);
        /* rows whose degree was > m_max0 before... approximate: any touched row e whose degree now == 32 and that is in new node's row => was shrunk */
        if (i >= N / 2) {
            const nrow *r = row_of(o, i, 0);
            for (uint32_t a = 0; a < r->n; a++) {
                uint32_t e = r->ids[a];
                selcnt[e]++;
            }
            /* shrunk rows: neighbours e of i (selected) that no longer... or whose degree == 32 */
            /* we mark every touched id as written now; for gap stats use selected e with degree==32 (just shrunk) */
        }
        /* selected = ids touched that contain i or were candidates; simpler: all touched ids are "written" */
        for (uint32_t t = 0; t < nt; t++) {
            uint32_t e = tb[t];
            if (e == i) continue;
            const nrow *er = row_of(o, e, 0);
            if (i >= N / 2 && er->n == 2 * M) { /* likely just shrunk (or exactly full) */
                uint32_t gap = i - last[e];
                int b = gap <= 4 ? 0 : gap <= 16 ? 1 : gap <= 64 ? 2 : gap <= 256 ? 3 : gap <= 1024 ? 4 : gap <= 4096 ? 5 : 6;
                hist[b]++; nshr++;
            }
            last[e] = i;
        }
        tot_touch += nt;
    }
    printf("N=%u: touched per insert %.1f; full-row writes (second half) %lu = %.2f per insert\n", N, (double)tot_touch / N, (unsigned long)nshr, (double)nshr / (N / 2));
    const char *lab[] = {"<=4", "<=16", "<=64", "<=256", "<=1024", "<=4096", ">4096"};
    for (int b = 0; b < 7; b++) printf("  gap %-7s %.3f\n", lab[b], (double)hist[b] / nshr);
    /* hub popularity: selection counts over the second half */
    uint32_t mx = 0; uint64_t tot = 0; for (uint32_t e = 0; e < N; e++) { if (selcnt[e] > mx) mx = selcnt[e]; tot += selcnt[e]; }
    /* top-k share */
    uint32_t *c = malloc(N * 4); memcpy(c, selcnt, N * 4);
    int cmpd(const void *a, const void *b) { return *(const uint32_t *)b > *(const uint32_t *)a ? 1 : -1; }
    qsort(c, N, 4, cmpd);
    uint64_t acc = 0; 
    for (uint32_t k = 0; k < N; k++) { acc += c[k]; if (k == 9 || k == 99 || k == 999 || k == 9999) printf("  top-%u nodes hold %.3f of link selections (max single %.4f of inserts)\n", k + 1, (double)acc / tot, (double)c[0] / (N / 2)); }
    return 0;
}
