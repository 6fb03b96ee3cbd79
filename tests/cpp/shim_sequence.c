/* shim_sequence.c -- the call sequence of the Rust FFI shim, in plain C against
 * the C ABI, with a stand-in for the Redis keyspace.
 *
 * The reference keeps every node twice: in the in-memory Index and as an `hnswnodet` key that it rewrites
 * through update_fn whenever a node's links change (src/lib.rs:351-353 `up`, :361-362; src/types.rs:292-309
 * NodeRedis::from).  With the engine behind the module, `up(name, id)` builds that value from
 * hnsw_get_neighbors + hnsw_get_vector.  This program plays HNSW.NEW / HNSW.NODE.ADD / HNSW.NODE.DEL /
 * HNSW.SEARCH exactly that way -- names and the key-value copies live here, the engine sees dense ids -- and
 * then checks the property the write-through exists for: the keyspace copy of every live node equals what
 * the engine holds (so make_index, src/lib.rs:252-315, would rebuild the same graph), and the graph rebuilt
 * from the keyspace through hnsw_import answers HNSW.SEARCH identically -- and, when the golden file is given
 * (argv[1] = tests/golden/shim_sequence.txt, written by the CPU oracle playing the same commands), that every
 * answer is the reference algorithm's: names in reply form (last '.' segment, core.rs:885-887) and similarity bits.
 * Exit code 0 = all assertions hold.                                                                    */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/hnsw_mi355x.h"

#define CHECK(c)                                                                         \
    do {                                                                                 \
        if (!(c)) { fprintf(stderr, "%s:%d: check failed: %s\n", __FILE__, __LINE__, #c); exit(1); } \
    } while (0)
#define OK(call)                                                                          \
    do {                                                                                  \
        hnsw_status st_ = (call);                                                         \
        if (st_ != HNSW_OK) { fprintf(stderr, "%s:%d: %s -> %d (%s)\n", __FILE__, __LINE__, #call, st_, hnsw_last_error(H)); exit(1); } \
    } while (0)

enum { DIM = 16, M = 4, EFC = 24, MAXN = 600, MAXL = 16, MAXDEG = 512 };

/* "Redis": one hnswnodet value per node key (src/types.rs:286-290) */
typedef struct {
    int present;
    char key[48];                        /* hnsw.{idx}.{node}, src/lib.rs:342-343 */
    float data[DIM];
    uint32_t n_layers;
    uint32_t deg[MAXL];
    char nbr[MAXL][64][48];              /* neighbour NAMES per layer, stored order */
} node_value;

static hnsw_index *H;
static node_value store[MAXN];           /* indexed by engine id for convenience; looked up by key below */
static char names[MAXN][48];             /* the shim's id -> name table (was Index.nodes keys, core.rs:316) */
static int alive[MAXN];
static uint32_t n_names;
static uint32_t levels[MAXN];
static unsigned long n_writes;

static int id_of(const char *key)
{
    for (uint32_t i = 0; i < n_names; i++)
        if (alive[i] && strcmp(names[i], key) == 0) return (int)i;
    return -1;
}

/* write_node(ctx, name, (&node).into())  (src/lib.rs:351-353, 409-420): rebuild the value from the engine */
static void write_node(uint32_t id)
{
    node_value *v = &store[id];
    uint32_t all[MAXN];
    OK(hnsw_get_levels(H, all));
    v->present = 1;
    strcpy(v->key, names[id]);
    OK(hnsw_get_vector(H, id, v->data));
    v->n_layers = all[id] + 1;
    levels[id] = all[id];
    for (uint32_t l = 0; l < v->n_layers; l++) {
        uint32_t ids[MAXDEG], n = 0;
        OK(hnsw_get_neighbors(H, id, l, ids, MAXDEG, &n));
        CHECK(n <= 64);
        v->deg[l] = n;
        for (uint32_t a = 0; a < n; a++) strcpy(v->nbr[l][a], names[ids[a]]);
    }
    n_writes++;
}

/* HNSW.NODE.ADD (src/lib.rs:334-368) */
static int add_node(const char *key, const float *data)
{
    if (n_names && id_of(key) >= 0) return -1;                    /* core.rs:407-409 */
    uint32_t id = 0, nt = 0, touched[8192];
    OK(hnsw_add(H, data, DIM, -1, &id, touched, 8192, &nt));
    CHECK(id == n_names);                                          /* dense ids in insertion order */
    strcpy(names[id], key);
    alive[id] = 1;
    n_names++;
    for (uint32_t i = 0; i < nt; i++) write_node(touched[i]);     /* update_fn, core.rs:580-584 */
    write_node(id);                                                /* src/lib.rs:361-362 */
    return 0;
}

/* HNSW.NODE.DEL (src/lib.rs:370-407) */
static int delete_node(const char *key)
{
    int id = id_of(key);
    if (id < 0) return -1;                                         /* core.rs:419-422 */
    uint32_t nt = 0, touched[8192];
    OK(hnsw_delete(H, (uint32_t)id, touched, 8192, &nt));
    alive[id] = 0;
    store[id].present = 0;                                         /* key deleted, src/lib.rs:402-404 */
    for (uint32_t i = 0; i < nt; i++)
        if (alive[touched[i]]) write_node(touched[i]);             /* core.rs:441-446 */
    return 0;
}

static float frand(unsigned *s) { *s = *s * 1664525u + 1013904223u; return (float)(*s >> 8) / 16777216.0f; }

int main(int argc, char **argv)
{
    FILE *golden = argc > 1 ? fopen(argv[1], "r") : NULL;
    if (argc > 1 && !golden) { fprintf(stderr, "cannot open %s\n", argv[1]); return 1; }
    unsigned n_golden = 0;
    OK(hnsw_create(DIM, M, EFC, 12345, 0, &H));                    /* HNSW.NEW, src/lib.rs:131-171 */
    unsigned seed = 7;
    static float V[MAXN][DIM];
    char key[48];
    const uint32_t N = 400;
    for (uint32_t i = 0; i < N; i++) {
        for (int d = 0; d < DIM; d++) V[i][d] = frand(&seed);
        snprintf(key, sizeof key, "hnsw.idx.n%u", i);
        CHECK(add_node(key, V[i]) == 0);
        if (i == 50) CHECK(add_node(key, V[i]) == -1);             /* duplicate name */
        if (i % 9 == 8) {                                          /* interleaved deletes */
            snprintf(key, sizeof key, "hnsw.idx.n%u", i - 5);
            CHECK(delete_node(key) == 0);
            CHECK(delete_node(key) == -1);                         /* already gone */
        }
    }
    /* 1. the keyspace copy of every live node equals the engine's rows */
    hnsw_info info;
    OK(hnsw_get_info(H, &info));
    uint32_t live = 0;
    for (uint32_t id = 0; id < n_names; id++) {
        if (!alive[id]) continue;
        live++;
        CHECK(store[id].present && strcmp(store[id].key, names[id]) == 0);
        for (uint32_t l = 0; l <= levels[id]; l++) {
            uint32_t ids[MAXDEG], n = 0;
            OK(hnsw_get_neighbors(H, id, l, ids, MAXDEG, &n));
            if (n != store[id].deg[l]) { fprintf(stderr, "node %u layer %u: keyspace has %u links, engine %u\n", id, l, store[id].deg[l], n); return 1; }
            for (uint32_t a = 0; a < n; a++) {
                CHECK(alive[ids[a]]);                                  /* no link to a deleted node */
                CHECK(strcmp(store[id].nbr[l][a], names[ids[a]]) == 0);
            }
        }
    }
    CHECK(live == info.node_count);

    /* 2. make_index (src/lib.rs:252-315): rebuild from the keyspace alone, through hnsw_import */
    static uint32_t new_id[MAXN], old_id[MAXN], lv2[MAXN];
    static float V2[MAXN][DIM];
    uint32_t n2 = 0;
    for (uint32_t id = 0; id < n_names; id++)
        if (alive[id]) { new_id[id] = n2; old_id[n2] = id; memcpy(V2[n2], store[id].data, sizeof V2[n2]); lv2[n2] = store[id].n_layers - 1; n2++; }
    uint32_t n_layers = info.max_layer + 1;
    uint64_t *rp[MAXL]; uint32_t *cl[MAXL];
    for (uint32_t l = 0; l < n_layers; l++) {
        rp[l] = calloc(n2 + 1, sizeof(uint64_t));
        cl[l] = calloc((size_t)n2 * 64 + 1, sizeof(uint32_t));
        uint64_t p = 0;
        for (uint32_t i = 0; i < n2; i++) {
            const node_value *v = &store[old_id[i]];
            rp[l][i] = p;
            if (l < v->n_layers)
                for (uint32_t a = 0; a < v->deg[l]; a++) {
                    int o = id_of(v->nbr[l][a]);                   /* names -> ids, src/lib.rs:277-283 */
                    CHECK(o >= 0);
                    cl[l][p++] = new_id[o];
                }
        }
        rp[l][n2] = p;
    }
    hnsw_index *H1 = H, *H2 = NULL;
    H = NULL;
    CHECK(hnsw_create(DIM, M, EFC, 1, 0, &H2) == HNSW_OK);
    H = H2;
    OK(hnsw_import(H2, n2, &V2[0][0], lv2, (int64_t)new_id[info.enterpoint], n_layers,
                   (const uint64_t *const *)rp, (const uint32_t *const *)cl));

    /* 3. HNSW.SEARCH (src/lib.rs:462-496) answers identically from both, by NAME */
    for (int q = 0; q < 60; q++) {
        float query[DIM];
        for (int d = 0; d < DIM; d++) query[d] = frand(&seed);
        uint32_t a_ids[5], b_ids[5], na = 0, nb = 0;
        float a_sims[5], b_sims[5];
        H = H1; OK(hnsw_search(H1, query, DIM, 5, a_ids, a_sims, &na));
        H = H2; OK(hnsw_search(H2, query, DIM, 5, b_ids, b_sims, &nb));
        CHECK(na == nb && na > 0);
        for (uint32_t i = 0; i < na; i++) {
            CHECK(strcmp(names[a_ids[i]], names[old_id[b_ids[i]]]) == 0);
            CHECK(memcmp(&a_sims[i], &b_sims[i], 4) == 0);
            CHECK(a_sims[i] <= 0.0f);                              /* sim = -(squared L2), metrics.rs:75 */
        }
        if (golden) {                                              /* the oracle's answer to the same command */
            unsigned gn = 0;
            CHECK(fscanf(golden, "%u", &gn) == 1 && gn == na);
            for (uint32_t i = 0; i < na; i++) {
                char gname[48]; unsigned gbits = 0, bits;
                CHECK(fscanf(golden, " %47[^:]:%x", gname, &gbits) == 2);
                const char *dot = strrchr(names[a_ids[i]], '.');   /* reply name = last '.' segment */
                memcpy(&bits, &a_sims[i], 4);
                if (strcmp(dot ? dot + 1 : names[a_ids[i]], gname) != 0 || bits != gbits) {
                    fprintf(stderr, "query %d rank %u: engine %s:%08x, oracle %s:%08x\n", q, i, dot ? dot + 1 : "?", bits, gname, gbits);
                    return 1;
                }
            }
            n_golden++;
        }
    }
    /* dimension mismatch surfaces the reference's message (core.rs:478-480) */
    {
        float bad[3] = {0, 0, 0};
        uint32_t ids[1], n = 0; float sims[1];
        H = H1;
        CHECK(hnsw_search(H1, bad, 3, 1, ids, sims, &n) == HNSW_ERR_DIM_MISMATCH);
        CHECK(strcmp(hnsw_last_error(H1), "data dimension: 3 does not match Index") == 0);
    }
    hnsw_destroy(H1);
    hnsw_destroy(H2);
    if (golden) fclose(golden);
    printf("shim_sequence ok: %u adds, %u live, %lu node writes, %u answers equal to the oracle's golden file\n", n_names, live, n_writes, n_golden);
    return 0;
}
