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
 *
 * The stand-in keyspace is REFERENCE-SHAPED: a node value holds the rows the reference would have written, not
 * level + 1 of them (a node promoted to enterpoint with l > l_max is saved with rows 0..=l_max only, core.rs:523;
 * the first node of an index with none, core.rs:393-405), and the hnswindex value (IndexRedis, src/types.rs:46-60)
 * keeps every node in the set of its TOP layer only (core.rs:596).  That value is maintained the way
 * GpuIndex::sync_redis (integration/rust/src/hnsw/gpu_index.rs) does it -- one name appended / swap-removed per
 * command, positions tracked per id -- and make_index takes levels from the layer sets and the layer count from
 * max_layer (src/lib.rs:287-299), as GpuIndex::from_keys does.  After the reload three more multi-level adds
 * (one of them a promotion) run on the reloaded index; every row and 20 answers must equal what the oracle, which
 * never reloaded, holds (argv[2] = tests/golden/shim_reload.txt).
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

/* "Redis": the hnswindex value (IndexRedis, src/types.rs:46-60).  Names are represented by engine ids here; the
 * strings are names[id].  Maintained incrementally exactly like GpuIndex::sync_redis. */
#define NOT_STORED 0xFFFFFFFFu
static struct {
    uint32_t node_count, max_layer;
    int64_t enterpoint;                  /* id or -1 ("null", src/types.rs:233-236) */
    uint32_t n_layers;
    uint32_t layer_len[MAXL];
    uint32_t layers[MAXL][MAXN];         /* layer l: the nodes whose TOP layer is l (core.rs:596) */
    uint32_t n_nodes;
    uint32_t nodes[MAXN];
} ir;
static uint32_t pos_nodes[MAXN], pos_layer[MAXN];

static void sync_redis_added(uint32_t id)
{
    uint32_t l = 0;
    OK(hnsw_get_level(H, id, &l));
    levels[id] = l;
    pos_nodes[id] = ir.n_nodes;
    ir.nodes[ir.n_nodes++] = id;
    while (ir.n_layers < l + 1) ir.layer_len[ir.n_layers++] = 0;   /* core.rs:590-592 */
    pos_layer[id] = ir.layer_len[l];
    ir.layers[l][ir.layer_len[l]++] = id;                          /* core.rs:596 */
}

static void sync_redis_removed(uint32_t id)
{
    uint32_t p = pos_nodes[id], l = levels[id], q = pos_layer[id];
    CHECK(p != NOT_STORED && ir.nodes[p] == id && ir.layers[l][q] == id);
    pos_nodes[id] = NOT_STORED;
    ir.nodes[p] = ir.nodes[--ir.n_nodes];                          /* swap_remove */
    if (p < ir.n_nodes) pos_nodes[ir.nodes[p]] = p;
    ir.layers[l][q] = ir.layers[l][--ir.layer_len[l]];             /* core.rs:426-430: the one set that holds it */
    if (q < ir.layer_len[l]) pos_layer[ir.layers[l][q]] = q;
}

static void sync_redis_header(void)
{
    hnsw_info info;
    OK(hnsw_get_info(H, &info));
    ir.node_count = info.node_count;
    ir.max_layer = info.max_layer;
    uint32_t want = info.node_count ? info.max_layer + 1 : 0;       /* core.rs:453-466: empty top layers are popped */
    while (ir.n_layers > want) CHECK(ir.layer_len[--ir.n_layers] == 0);
    ir.enterpoint = info.enterpoint;
}

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
    levels[id] = all[id];
    v->n_layers = 0;
    for (uint32_t l = 0; l <= all[id]; l++) {
        uint32_t ids[MAXDEG], n = 0;
        OK(hnsw_get_neighbors(H, id, l, ids, MAXDEG, &n));
        CHECK(n <= 64);
        v->deg[l] = n;
        for (uint32_t a = 0; a < n; a++) strcpy(v->nbr[l][a], names[ids[a]]);
        /* the reference's rows exist up to the highest layer something connected at (add_neighbor -> push_levels,
         * core.rs:127-143): a promoted enterpoint's rows above the old l_max, and the first node's only row, are
         * not there until a neighbour arrives */
        if (n) v->n_layers = l + 1;
    }
    n_writes++;
}

/* HNSW.NODE.ADD (src/lib.rs:334-368) */
static int add_node_level(const char *key, const float *data, int32_t level)
{
    if (n_names && id_of(key) >= 0) return -1;                    /* core.rs:407-409 */
    uint32_t id = 0, nt = 0, touched[8192];
    OK(hnsw_add(H, data, DIM, level, &id, touched, 8192, &nt));
    CHECK(id == n_names);                                          /* dense ids in insertion order */
    CHECK(nt <= 8192);
    strcpy(names[id], key);
    alive[id] = 1;
    n_names++;
    for (uint32_t i = 0; i < nt; i++) write_node(touched[i]);     /* update_fn, core.rs:580-584 */
    write_node(id);                                                /* src/lib.rs:361-362 */
    sync_redis_added(id);                                          /* update_index, src/lib.rs:365 */
    sync_redis_header();
    return 0;
}
static int add_node(const char *key, const float *data) { return add_node_level(key, data, -1); }

/* HNSW.NODE.DEL (src/lib.rs:370-407) */
static int delete_node(const char *key)
{
    int id = id_of(key);
    if (id < 0) return -1;                                         /* core.rs:419-422 */
    uint32_t nt = 0, touched[8192];
    OK(hnsw_delete(H, (uint32_t)id, touched, 8192, &nt));
    alive[id] = 0;
    store[id].present = 0;                                         /* key deleted, src/lib.rs:402-404 */
    CHECK(nt <= 8192);
    for (uint32_t i = 0; i < nt; i++)
        if (alive[touched[i]]) write_node(touched[i]);             /* core.rs:441-446 */
    sync_redis_removed((uint32_t)id);                              /* update_index, src/lib.rs:404 */
    sync_redis_header();
    return 0;
}

static float frand(unsigned *s) { *s = *s * 1664525u + 1013904223u; return (float)(*s >> 8) / 16777216.0f; }

/* one golden line "R name layer n nbr..." per (live node, layer) of the oracle's graph, any order */
static void check_rows_against(FILE *g, hnsw_index *h, const char (*nm)[48], uint32_t n_ids, const uint8_t *dead)
{
    unsigned n_rows = 0, want_rows = 0;
    uint32_t lv[MAXN];
    OK(hnsw_get_levels(h, lv));
    for (uint32_t i = 0; i < n_ids; i++) if (!dead[i]) want_rows += lv[i] + 1;
    CHECK(fscanf(g, " ROWS %u", &n_rows) == 1);
    if (n_rows != want_rows) { fprintf(stderr, "oracle holds %u rows, the engine %u\n", n_rows, want_rows); exit(1); }
    for (unsigned r = 0; r < n_rows; r++) {
        char gname[48]; unsigned layer = 0, deg = 0;
        CHECK(fscanf(g, " R %47s %u %u", gname, &layer, &deg) == 3);
        int id = -1;
        for (uint32_t i = 0; i < n_ids && id < 0; i++) {
            const char *dot = strrchr(nm[i], '.');
            if (!dead[i] && strcmp(dot ? dot + 1 : nm[i], gname) == 0) id = (int)i;
        }
        if (id < 0) { fprintf(stderr, "the oracle has node %s, the engine does not\n", gname); exit(1); }
        CHECK(layer <= lv[id]);
        uint32_t ids[MAXDEG], n = 0;
        OK(hnsw_get_neighbors(h, (uint32_t)id, layer, ids, MAXDEG, &n));
        if (n != deg) { fprintf(stderr, "%s layer %u: oracle %u links, engine %u\n", gname, layer, deg, n); exit(1); }
        for (uint32_t a = 0; a < deg; a++) {                       /* stored order is part of the semantics, core.rs:646 */
            char gn[48];
            CHECK(fscanf(g, " %47s", gn) == 1);
            const char *dot = strrchr(nm[ids[a]], '.');
            if (strcmp(dot ? dot + 1 : nm[ids[a]], gn) != 0) {
                fprintf(stderr, "%s layer %u link %u: oracle %s, engine %s\n", gname, layer, a, gn, nm[ids[a]]);
                exit(1);
            }
        }
    }
}

static unsigned check_answers_against(FILE *g, hnsw_index *h, const char (*nm)[48], int n_queries, unsigned *seed)
{
    unsigned n_golden = 0;
    for (int q = 0; q < n_queries; q++) {
        float query[DIM];
        for (int d = 0; d < DIM; d++) query[d] = frand(seed);
        uint32_t a_ids[5], na = 0;
        float a_sims[5];
        OK(hnsw_search(h, query, DIM, 5, a_ids, a_sims, &na));
        unsigned gn = 0;
        CHECK(fscanf(g, "%u", &gn) == 1 && gn == na);
        for (uint32_t i = 0; i < na; i++) {
            char gname[48]; unsigned gbits = 0, bits;
            CHECK(fscanf(g, " %47[^:]:%x", gname, &gbits) == 2);
            const char *dot = strrchr(nm[a_ids[i]], '.');          /* reply name = last '.' segment */
            memcpy(&bits, &a_sims[i], 4);
            if (strcmp(dot ? dot + 1 : nm[a_ids[i]], gname) != 0 || bits != gbits) {
                fprintf(stderr, "query %d rank %u: engine %s:%08x, oracle %s:%08x\n", q, i, dot ? dot + 1 : "?", bits, gname, gbits);
                exit(1);
            }
        }
        n_golden++;
    }
    return n_golden;
}

int main(int argc, char **argv)
{
    FILE *golden = argc > 1 ? fopen(argv[1], "r") : NULL;
    if (argc > 1 && !golden) { fprintf(stderr, "cannot open %s\n", argv[1]); return 1; }
    FILE *golden2 = argc > 2 ? fopen(argv[2], "r") : NULL;
    if (argc > 2 && !golden2) { fprintf(stderr, "cannot open %s\n", argv[2]); return 1; }
    unsigned n_golden = 0, n_golden2 = 0;
    OK(hnsw_create(DIM, M, EFC, 12345, 0, &H));                    /* HNSW.NEW, src/lib.rs:131-171 */
    unsigned seed = 7;
    static float V[MAXN][DIM];
    char key[48];
    const uint32_t N = 400;
    for (uint32_t i = 0; i < N; i++) {
        for (int d = 0; d < DIM; d++) V[i][d] = frand(&seed);
        snprintf(key, sizeof key, "hnsw.idx.n%u", i);
        CHECK(add_node(key, V[i]) == 0);
        if (i == 0) CHECK(store[0].n_layers == 0);                 /* the first node is saved without a row, core.rs:393-405 */
        if (i == 50) CHECK(add_node(key, V[i]) == -1);             /* duplicate name */
        if (i % 9 == 8) {                                          /* interleaved deletes */
            snprintf(key, sizeof key, "hnsw.idx.n%u", i - 5);
            CHECK(delete_node(key) == 0);
            CHECK(delete_node(key) == -1);                         /* already gone */
        }
    }
    /* the last command before the "restart": a node that lands ABOVE the current top layer (l > l_max) becomes the
     * enterpoint (core.rs:587-593) and is saved with its pre-promotion rows only */
    hnsw_info info;
    OK(hnsw_get_info(H, &info));
    const uint32_t old_top = info.max_layer;
    for (int d = 0; d < DIM; d++) V[N][d] = frand(&seed);
    snprintf(key, sizeof key, "hnsw.idx.n%u", N);
    CHECK(add_node_level(key, V[N], (int32_t)old_top + 2) == 0);
    OK(hnsw_get_info(H, &info));
    CHECK(info.max_layer == old_top + 2 && info.enterpoint == (int64_t)N);
    CHECK(store[N].n_layers == old_top + 1);                       /* fewer rows than level + 1: the reference's shape */
    CHECK(ir.max_layer == old_top + 2 && ir.n_layers == old_top + 3 && ir.layer_len[old_top + 1] == 0);

    /* 1. the keyspace copy of every live node equals the engine's rows; the hnswindex value is the reference's */
    uint32_t live = 0, all_levels[MAXN], in_sets = 0;
    OK(hnsw_get_levels(H, all_levels));
    for (uint32_t id = 0; id < n_names; id++) {
        if (!alive[id]) continue;
        live++;
        CHECK(store[id].present && strcmp(store[id].key, names[id]) == 0);
        CHECK(store[id].n_layers <= all_levels[id] + 1);
        for (uint32_t l = 0; l <= all_levels[id]; l++) {
            uint32_t ids[MAXDEG], n = 0;
            OK(hnsw_get_neighbors(H, id, l, ids, MAXDEG, &n));
            uint32_t have = l < store[id].n_layers ? store[id].deg[l] : 0;   /* a missing row is an empty row */
            if (n != have) { fprintf(stderr, "node %u layer %u: keyspace has %u links, engine %u\n", id, l, have, n); return 1; }
            for (uint32_t a = 0; a < n; a++) {
                CHECK(alive[ids[a]]);                                  /* no link to a deleted node */
                CHECK(strcmp(store[id].nbr[l][a], names[ids[a]]) == 0);
            }
        }
        CHECK(pos_nodes[id] < ir.n_nodes && ir.nodes[pos_nodes[id]] == id);
        CHECK(levels[id] == all_levels[id] && ir.layers[levels[id]][pos_layer[id]] == id);
    }
    CHECK(live == info.node_count && ir.n_nodes == live && ir.node_count == live);
    for (uint32_t l = 0; l < ir.n_layers; l++) {                   /* every node in exactly ONE set: its top layer */
        in_sets += ir.layer_len[l];
        for (uint32_t a = 0; a < ir.layer_len[l]; a++) CHECK(alive[ir.layers[l][a]] && all_levels[ir.layers[l][a]] == l);
    }
    CHECK(in_sets == live && ir.enterpoint == info.enterpoint);

    /* 2. make_index (src/lib.rs:252-315) as GpuIndex::from_keys does it: ids in the order of IndexRedis.nodes,
     * levels from the layer sets, layer count from max_layer, a missing row = an empty row; ONE hnsw_import */
    static uint32_t new_id[MAXN], old_id[MAXN], lv2[MAXN];
    static float V2[MAXN][DIM];
    static char names2[MAXN][48];
    static uint8_t dead2[MAXN];
    uint32_t n2 = ir.n_nodes;
    for (uint32_t i = 0; i < n2; i++) {
        uint32_t id = ir.nodes[i];
        new_id[id] = i; old_id[i] = id;
        memcpy(V2[i], store[id].data, sizeof V2[i]);
        strcpy(names2[i], store[id].key);
        lv2[i] = NOT_STORED;
    }
    for (uint32_t l = 0; l < ir.n_layers; l++)
        for (uint32_t a = 0; a < ir.layer_len[l]; a++) lv2[new_id[ir.layers[l][a]]] = l;    /* src/lib.rs:287-299 */
    for (uint32_t i = 0; i < n2; i++) CHECK(lv2[i] != NOT_STORED);
    uint32_t n_layers = ir.max_layer + 1;
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
    OK(hnsw_import(H2, n2, &V2[0][0], lv2, (int64_t)new_id[ir.enterpoint], n_layers,
                   (const uint64_t *const *)rp, (const uint32_t *const *)cl));
    OK(hnsw_get_info(H2, &info));
    CHECK(info.max_layer == old_top + 2 && info.node_count == live);   /* not "rows - 1" of the widest node */

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
            CHECK(strcmp(names[a_ids[i]], names2[b_ids[i]]) == 0);
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

    /* 4. the module goes on after the restart: three multi-level HNSW.NODE.ADDs on the RELOADED index (a level-1
     * node, a second promotion, a level-2 node) and one HNSW.NODE.DEL.  Had the reload taken levels from row counts,
     * max_layer would be old_top here, the descents would start two layers low and these inserts would link
     * differently from the reference (which the oracle, playing the same commands without a restart, stands for). */
    H = H2;
    const int32_t lv_more[3] = {1, (int32_t)old_top + 3, 2};
    uint32_t n_ids2 = n2;
    for (int a = 0; a < 3; a++) {
        float v[DIM];
        for (int d = 0; d < DIM; d++) v[d] = frand(&seed);
        uint32_t id = 0, nt = 0, touched[8192];
        OK(hnsw_add(H2, v, DIM, lv_more[a], &id, touched, 8192, &nt));
        CHECK(id == n_ids2 && nt <= 8192);
        snprintf(names2[id], sizeof names2[id], "hnsw.idx.n%u", N + 1 + (uint32_t)a);
        n_ids2++;
    }
    {
        uint32_t nt = 0, touched[8192];
        OK(hnsw_delete(H2, new_id[17], touched, 8192, &nt));       /* hnsw.idx.n17 */
        dead2[new_id[17]] = 1;
    }
    OK(hnsw_get_info(H2, &info));
    CHECK(info.max_layer == old_top + 3 && info.enterpoint == (int64_t)(n2 + 1) && info.node_count == live + 2);
    if (golden2) {
        check_rows_against(golden2, H2, names2, n_ids2, dead2);
        n_golden2 = check_answers_against(golden2, H2, names2, 20, &seed);
    }
    hnsw_destroy(H1);
    hnsw_destroy(H2);
    if (golden) fclose(golden);
    if (golden2) fclose(golden2);
    printf("shim_sequence ok: %u adds, %u live, %lu node writes, %u answers equal to the oracle's golden file; "
           "reference-shaped reload: every row and %u answers equal to the oracle's after 3 more adds and a delete\n",
           n_names, live, n_writes, n_golden, n_golden2);
    return 0;
}
