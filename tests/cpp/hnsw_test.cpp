// hnsw_test.cpp -- src/hnsw/core_tests.rs:6-53 (hnsw_test: create, add, search, delete) against the MI355X engine
// through the C++ host mirror.  Also src/hnsw/metrics_tests.rs through
// hnsw_metric_pairs.  Exit code 0 = all assertions hold.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <limits>
#include <string>
#include <vector>

#include "../../redis_hnsw_amd/host/hnsw_index.hpp"

#define ASSERT(c)                                                          \
    do {                                                                   \
        if (!(c)) {                                                        \
            std::fprintf(stderr, "%s:%d: assertion failed: %s\n", __FILE__, __LINE__, #c); \
            return 1;                                                      \
        }                                                                  \
    } while (0)

using redis_hnsw::HNSWError;
using redis_hnsw::Index;

static float metric(const std::vector<float> &a, const std::vector<float> &b)
{
    float s = 0;
    if (hnsw_metric_pairs(0, a.data(), b.data(), 1, (uint32_t)a.size(), &s) != HNSW_OK) std::abort();
    return s;
}

int main()
{
    const float EPS = std::numeric_limits<float>::epsilon();
    // ---- metrics_tests.rs:3-33 -------------------------------------------------
    ASSERT(std::fabs(metric(std::vector<float>(512, 1.f), std::vector<float>(512, 1.f)) - 0.0f) < EPS);
    ASSERT(std::fabs(metric(std::vector<float>(512, 0.f), std::vector<float>(512, 1.f)) - -512.0f) < EPS);
    ASSERT(std::fabs(metric(std::vector<float>(512, 0.f), std::vector<float>(512, 512.f)) - -134217728.0f) < EPS);
    ASSERT(std::fabs(metric(std::vector<float>(33, 0.f), std::vector<float>(33, 1.f)) - -33.0f) < EPS);

    // ---- core_tests.rs:8-19 index creation --------------------------------------
    const size_t n = 100, data_dim = 4;
    Index index("foo", data_dim, 5, 16);
    ASSERT(index.name == "foo");
    ASSERT(index.data_dim == data_dim);
    ASSERT(index.m == 5);
    ASSERT(index.ef_construction == 16);
    ASSERT(index.node_count() == 0);
    ASSERT(index.max_layer() == 0);
    ASSERT(!index.has_enterpoint());

    size_t updates = 0;
    auto mock_fn = [&](const std::string &, uint32_t) { ++updates; };   // core_tests.rs:21

    // ---- core_tests.rs:23-42 add node ---------------------------------------------
    for (size_t i = 0; i < n; ++i) {
        std::string name = "node" + std::to_string(i);
        std::vector<float> data(data_dim, (float)i);
        index.add_node(name, data, mock_fn);
    }
    ASSERT(index.node_count() == n);
    ASSERT(index.has_enterpoint());
    ASSERT(updates > 0);

    // ---- core_tests.rs:44-53 search ----------------------------------------------
    std::vector<float> query(4, 10.0f);
    auto res = index.search_knn(query, 5);
    ASSERT(res.size() == 5);
    ASSERT(std::fabs(res[0].sim - 0.0f) < EPS);
    ASSERT(res[0].name == "node10");
    ASSERT(std::fabs(res[1].sim - -4.0f) < EPS);
    ASSERT(std::fabs(res[2].sim - -4.0f) < EPS);
    ASSERT(std::fabs(res[3].sim - -16.0f) < EPS);
    ASSERT(std::fabs(res[4].sim - -16.0f) < EPS);

    // ---- error behaviour (core.rs:390,408,479) --------------------------------------
    try {
        index.add_node("bad", std::vector<float>(3, 0.f));
        ASSERT(false);
    } catch (const HNSWError &e) {
        ASSERT(e.error_string() == "String(\"data dimension: 3 does not match Index\")");
    }
    try {
        index.add_node("node7", std::vector<float>(4, 0.f));
        ASSERT(false);
    } catch (const HNSWError &e) {
        ASSERT(e.error_string() == "String(\"Node: \\\"node7\\\" already exists\")");
    }
    try {
        index.search_knn(std::vector<float>(5, 0.f), 1);
        ASSERT(false);
    } catch (const HNSWError &e) {
        ASSERT(e.error_string() == "String(\"data dimension: 5 does not match Index\")");
    }
    // a graph the reference would accept: symmetric links (core.rs:770-772)
    for (const auto &nb : index.neighbors("node10", 0)) {
        bool back = false;
        for (const auto &x : index.neighbors(nb, 0)) back |= x == "node10";
        ASSERT(back);
    }
    // ---- core_tests.rs:55-80 delete node ------------------------------------------------
    for (size_t i = 0; i < n; ++i) {
        std::string name = "node" + std::to_string(i);
        index.delete_node(name, mock_fn);
        ASSERT(index.node_count() == n - i - 1);
        ASSERT(!index.contains(name));
        // no remaining node lists it as a neighbour (sampled: the survivors next to it on the line)
        for (size_t jn = i + 1; jn < n && jn < i + 8; ++jn)
            for (size_t lc = 0; lc <= index.max_layer(); ++lc)
                for (const auto &x : index.neighbors("node" + std::to_string(jn), lc)) ASSERT(x != name);
    }
    ASSERT(!index.has_enterpoint());
    try {
        index.delete_node("node3");
        ASSERT(false);
    } catch (const HNSWError &e) {
        ASSERT(e.error_string() == "String(\"Node: \\\"node3\\\" does not exist\")");   // core.rs:421
    }
    std::printf("hnsw_test ok (%zu update_fn calls)\n", updates);
    return 0;
}
