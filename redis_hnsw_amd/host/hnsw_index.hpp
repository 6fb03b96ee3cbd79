// hnsw_index.hpp -- C++ host-side mirror of the reference's Index<f32,f32>
// (src/hnsw/core.rs:303-486) over the C ABI of include/hnsw_mi355x.h.
//
// The reference is Rust and there is no Rust toolchain in this image, so this is
// the host side "in the reference's own shape": same method names, argument
// meaning and error behaviour as Index::new / add_node / search_knn, so a test
// written against it reads like src/hnsw/core_tests.rs.  INTEGRATION.md shows
// the equivalent `extern "C"` shim for the Rust module.
#pragma once
#include <cstdint>
#include <functional>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/hnsw_mi355x.h"

namespace redis_hnsw {

// core.rs:25-46.  error_string() is the `{:?}` rendering the Redis client sees.
struct HNSWError : std::runtime_error {
    hnsw_status status;
    HNSWError(const std::string &m, hnsw_status s) : std::runtime_error(m), status(s) {}
    std::string error_string() const
    {
        std::string out = "String(\"";
        for (char c : std::string(what())) {
            if (c == '"' || c == '\\') out.push_back('\\');
            out.push_back(c);
        }
        return out + "\")";
    }
};

// core.rs:48-52 (the vector copy is dropped: src/lib.rs:486-492 never reads it)
struct SearchResult {
    float sim;
    std::string name;
    uint32_t id;
};

class Index {
public:
    // Index::new(name, Box::new(euclidean), data_dim, m, ef_construction)   core.rs:322-347
    Index(const std::string &name_, size_t data_dim_, size_t m_, size_t ef_construction_, uint64_t seed = 0,
          int device = 0)
        : name(name_), data_dim(data_dim_), m(m_), m_max(m_), m_max_0(2 * m_), ef_construction(ef_construction_)
    {
        hnsw_status s = hnsw_create((uint32_t)data_dim, (uint32_t)m, (uint32_t)ef_construction, seed, device, &h_);
        if (s != HNSW_OK) {
            std::string msg = h_ ? hnsw_last_error(h_) : "hnsw_create failed";
            hnsw_destroy(h_);
            h_ = nullptr;
            throw HNSWError(msg, s);
        }
    }
    ~Index() { hnsw_destroy(h_); }
    Index(const Index &) = delete;
    Index &operator=(const Index &) = delete;

    // pub fields of the reference struct (core.rs:303-319)
    std::string name;
    size_t data_dim, m, m_max, m_max_0, ef_construction;
    size_t node_count() const { return info().node_count; }
    size_t max_layer() const { return info().max_layer; }
    // name of the enterpoint or "" for None (core.rs:317)
    std::string enterpoint() const
    {
        int64_t e = info().enterpoint;
        return e < 0 ? std::string() : names_[(size_t)e];
    }
    bool has_enterpoint() const { return info().enterpoint >= 0; }

    // add_node(&mut self, name, data, update_fn)   core.rs:383-412
    void add_node(const std::string &node, const std::vector<float> &data,
                  const std::function<void(const std::string &, uint32_t)> &update_fn = nullptr, int32_t level = -1)
    {
        if (data.size() != data_dim)                                   // core.rs:389-391
            throw HNSWError("data dimension: " + std::to_string(data.size()) + " does not match Index",
                            HNSW_ERR_DIM_MISMATCH);
        if (node_count() != 0 && ids_.count(node))                     // core.rs:393-409
            throw HNSWError("Node: \"" + node + "\" already exists", HNSW_ERR_DUPLICATE);
        uint32_t id = 0, nt = 0;
        std::vector<uint32_t> touched(update_fn ? 65536 : 0);
        check(hnsw_add(h_, data.data(), (uint32_t)data.size(), level, &id, update_fn ? touched.data() : nullptr,
                       (uint32_t)touched.size(), update_fn ? &nt : nullptr));
        names_.push_back(node);                                        // the insert is complete whenever the status is OK
        ids_[node] = id;
        if (update_fn && nt > touched.size())
            throw HNSWError("update_fn list of " + std::to_string(nt) + " ids does not fit the buffer", HNSW_ERR_CAPACITY);
        if (update_fn)                                                 // core.rs:580-584
            for (uint32_t i = 0; i < nt; ++i) update_fn(names_[touched[i]], touched[i]);
    }

    // delete_node(&mut self, name, update_fn)   core.rs:414-475
    void delete_node(const std::string &node, const std::function<void(const std::string &, uint32_t)> &update_fn = nullptr)
    {
        auto it = ids_.find(node);
        if (it == ids_.end()) throw HNSWError("Node: \"" + node + "\" does not exist", HNSW_ERR_NOT_FOUND);  // :421
        const uint32_t id = it->second;
        uint32_t nt = 0;
        std::vector<uint32_t> touched(65536);
        check(hnsw_delete(h_, id, touched.data(), (uint32_t)touched.size(), &nt));
        ids_.erase(it);
        if (update_fn && nt > touched.size())
            throw HNSWError("update_fn list of " + std::to_string(nt) + " ids does not fit the buffer", HNSW_ERR_CAPACITY);
        if (update_fn)                                                 // core.rs:441-446
            for (uint32_t i = 0; i < nt; ++i) update_fn(names_[touched[i]], touched[i]);
    }
    bool contains(const std::string &node) const { return ids_.count(node) != 0; }

    // search_knn(&self, data, k)   core.rs:477-486
    std::vector<SearchResult> search_knn(const std::vector<float> &data, size_t k) const
    {
        if (data.size() != data_dim)                                   // core.rs:478-480
            throw HNSWError("data dimension: " + std::to_string(data.size()) + " does not match Index",
                            HNSW_ERR_DIM_MISMATCH);
        std::vector<uint32_t> ids(k ? k : 1);
        std::vector<float> sims(k ? k : 1);
        uint32_t n = 0;
        check(hnsw_search(h_, data.data(), (uint32_t)data.size(), (uint32_t)k, ids.data(), sims.data(), &n));
        std::vector<SearchResult> res;
        for (uint32_t i = 0; i < n; ++i) {
            const std::string &full = names_[ids[i]];
            size_t dot = full.rfind('.');                              // core.rs:885-887 last '.' segment
            res.push_back({sims[i], dot == std::string::npos ? full : full.substr(dot + 1), ids[i]});
        }
        return res;
    }

    // adjacency of one node for NodeRedis write-through (src/types.rs:292-309)
    std::vector<std::string> neighbors(const std::string &node, size_t layer) const
    {
        auto it = ids_.find(node);
        if (it == ids_.end()) throw HNSWError("Node: \"" + node + "\" does not exist", HNSW_ERR_NOT_FOUND);
        hnsw_info inf = info();
        std::vector<uint32_t> out(inf.stride0 > inf.stride_upper ? inf.stride0 : inf.stride_upper);
        uint32_t n = 0;
        check(hnsw_get_neighbors(h_, it->second, (uint32_t)layer, out.data(), (uint32_t)out.size(), &n));
        std::vector<std::string> r;
        for (uint32_t i = 0; i < n; ++i) r.push_back(names_[out[i]]);
        return r;
    }

    hnsw_index *handle() const { return h_; }

private:
    hnsw_info info() const
    {
        hnsw_info i;
        check(hnsw_get_info(h_, &i));
        return i;
    }
    void check(hnsw_status s) const
    {
        if (s != HNSW_OK) throw HNSWError(hnsw_last_error(h_), s);
    }
    hnsw_index *h_ = nullptr;
    std::vector<std::string> names_;                 // id -> name (the reference's `nodes` map, core.rs:316)
    std::unordered_map<std::string, uint32_t> ids_;  // name -> id
};

} // namespace redis_hnsw
