// hnsw_group.hip -- one process, several GPUs: the primary index plus one replica per further device,
// written entirely ABOVE the C ABI of hnsw_mi355x.h (it calls nothing but the entry points a host program has) plus
// HIP peer copies.  Searches shard over the members; writes are replayed on every member (the exact insert and delete
// are deterministic given the level, core.rs:489-599 / :414-475), so the members stay identical row for row without
// moving rows around after every HNSW.NODE.ADD.  No kernels here.
#include "../../include/hnsw_mi355x.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <condition_variable>
#include <cstdint>
#include <cstring>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace {

// one worker thread per replica: a member's host-side calls (staging copies, launches, the final wait) must not
// serialise behind another member's
struct Worker {
    std::thread th;
    std::mutex mu;
    std::condition_variable cv;
    std::function<void()> job;
    bool has_job = false, done = true, quit = false;

    void start()
    {
        th = std::thread([this] {
            std::unique_lock<std::mutex> lk(mu);
            for (;;) {
                cv.wait(lk, [this] { return has_job || quit; });
                if (quit) return;
                std::function<void()> j = std::move(job);
                has_job = false;
                lk.unlock();
                j();
                lk.lock();
                done = true;
                cv.notify_all();
            }
        });
    }
    void post(std::function<void()> j)
    {
        std::lock_guard<std::mutex> lk(mu);
        job = std::move(j);
        has_job = true;
        done = false;
        cv.notify_all();
    }
    void wait()
    {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [this] { return done; });
    }
    void stop()
    {
        {
            std::lock_guard<std::mutex> lk(mu);
            quit = true;
            cv.notify_all();
        }
        if (th.joinable()) th.join();
    }
};

struct Rng {                                           // xoshiro256**, seeded through splitmix64
    uint64_t s[4];
    static uint64_t splitmix(uint64_t &x)
    {
        uint64_t z = (x += 0x9E3779B97F4A7C15ull);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return z ^ (z >> 31);
    }
    void seed(uint64_t x) { for (auto &v : s) v = splitmix(x); }
    static uint64_t rotl(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }
    uint64_t next()
    {
        const uint64_t r = rotl(s[1] * 5, 7) * 9, t = s[1] << 17;
        s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3]; s[2] ^= t; s[3] = rotl(s[3], 45);
        return r;
    }
    double uniform() { return (double)(next() >> 11) * (1.0 / 9007199254740992.0); }   // [0, 1)
};

} // namespace

struct hnsw_group {
    std::vector<hnsw_index *> member;                  // [0] = the primary (not owned)
    std::vector<int> device;                           // device of every member
    std::vector<Worker *> worker;                      // [0] unused: the primary's share runs on the calling thread
    std::vector<hnsw_status> st;
    hnsw_info info = {};
    uint64_t seed = 0;
    Rng rng;
    bool diverged = false;                             // a replayed write failed on some member: refresh first
    std::string err;
};

namespace {

hnsw_status gfail(hnsw_group *g, hnsw_status s, const std::string &msg)
{
    g->err = msg;
    return s;
}

// set the current device for a scope and put the caller's back (this file sits above the C ABI and shares nothing
// with the engine's translation units)
struct DeviceScope {
    int prev = -1, dev;
    explicit DeviceScope(int d) : dev(d)
    {
        if (hipGetDevice(&prev) != hipSuccess) { prev = -1; (void)hipGetLastError(); }
        if (prev != dev) (void)hipSetDevice(dev);
    }
    ~DeviceScope()
    {
        if (prev >= 0 && prev != dev) (void)hipSetDevice(prev);
    }
};

int device_of(hnsw_index *h, const hnsw_replica &view)
{
    // the device a member's tables live on, from the tables themselves
    hipPointerAttribute_t a;
    if (view.adj0 && hipPointerGetAttributes(&a, view.adj0) == hipSuccess) return a.device;
    (void)hipGetLastError();
    (void)h;
    return 0;
}

hnsw_status copy_table(hnsw_group *g, void *dst, int ddev, const void *src, int sdev, uint64_t bytes)
{
    if (!bytes) return HNSW_OK;
    hipError_t e = ddev == sdev ? hipMemcpy(dst, src, bytes, hipMemcpyDeviceToDevice) : hipMemcpyPeer(dst, ddev, src, sdev, bytes);
    if (e != hipSuccess) return gfail(g, HNSW_ERR_DEVICE, std::string("group: peer copy failed: ") + hipGetErrorString(e));
    return HNSW_OK;
}

// a fresh replica of the primary on `dev`
hnsw_status make_replica(hnsw_group *g, int dev, uint64_t seed, hnsw_index **out)
{
    *out = nullptr;
    hnsw_index *p = g->member[0];
    hnsw_index *r = nullptr;
    hnsw_status s = hnsw_create(g->info.dim, g->info.m, g->info.ef_construction, seed, dev, &r);
    if (s != HNSW_OK) {                                // hnsw_create hands a handle back even when it fails (for the message)
        const std::string why = r ? hnsw_last_error(r) : "no handle";
        hnsw_destroy(r);
        return gfail(g, s, std::string("group: hnsw_create on device ") + std::to_string(dev) + ": " + why);
    }
    hnsw_replica src;
    if ((s = hnsw_replica_view(p, &src)) != HNSW_OK) { g->err = hnsw_last_error(p); hnsw_destroy(r); return s; }
    if (src.n) {
        const int pdev = g->device[0];
        if (pdev != dev) {
            // direct xGMI copies when the devices can reach each other (hipMemcpyPeer stages through the host otherwise)
            int can = 0;
            if (hipDeviceCanAccessPeer(&can, dev, pdev) == hipSuccess && can) {
                DeviceScope on_dev(dev);                    // peer access is enabled from the device that reads; the caller's
                (void)hipDeviceEnablePeerAccess(pdev, 0);   // current device is put back when the scope ends
            }
            (void)hipGetLastError();
        }
        hnsw_replica dst = src;
        if ((s = hnsw_replica_prepare(r, &dst)) != HNSW_OK) { g->err = hnsw_last_error(r); hnsw_destroy(r); return s; }
        if ((s = copy_table(g, dst.vec, dev, src.vec, pdev, src.vec_bytes)) != HNSW_OK ||
            (s = copy_table(g, dst.adj0, dev, src.adj0, pdev, src.adj0_bytes)) != HNSW_OK ||
            (s = copy_table(g, dst.adj_upper, dev, src.adj_upper, pdev, src.adj_upper_bytes)) != HNSW_OK ||
            (s = copy_table(g, dst.upper_base, dev, src.upper_base, pdev, (uint64_t)src.n * 4)) != HNSW_OK ||
            (s = copy_table(g, dst.levels, dev, src.levels, pdev, (uint64_t)src.n * 4)) != HNSW_OK) {
            hnsw_destroy(r);
            return s;
        }
        std::vector<uint8_t> dead;
        if (src.n_dead) {
            dead.resize(src.n);
            if ((s = hnsw_get_tombstones(p, dead.data())) != HNSW_OK) { g->err = hnsw_last_error(p); hnsw_destroy(r); return s; }
        }
        if ((s = hnsw_replica_commit(r, &dst, src.n_dead ? dead.data() : nullptr)) != HNSW_OK) { g->err = hnsw_last_error(r); hnsw_destroy(r); return s; }
    }
    *out = r;
    return HNSW_OK;
}

// run fn(i) for every member at once: replicas on their workers, the primary on the calling thread
template <class F>
hnsw_status on_all(hnsw_group *g, F fn, uint32_t first = 0)
{
    const uint32_t G = (uint32_t)g->member.size();
    for (uint32_t i = std::max(first, 1u); i < G; ++i) {
        g->st[i] = HNSW_OK;
        g->worker[i]->post([g, i, fn] { g->st[i] = fn(i); });
    }
    if (first == 0) g->st[0] = fn(0);
    for (uint32_t i = std::max(first, 1u); i < G; ++i) g->worker[i]->wait();
    for (uint32_t i = first; i < G; ++i)
        if (g->st[i] != HNSW_OK)
            return gfail(g, g->st[i], "group member " + std::to_string(i) + " (device " + std::to_string(g->device[i]) + "): " + hnsw_last_error(g->member[i]));
    return HNSW_OK;
}

int32_t draw_level(hnsw_group *g)                      // core.rs:601-605
{
    double u = g->rng.uniform();
    if (u <= 0.0) u = 1e-300;
    const double l = std::floor(-std::log(u) * (1.0 / std::log((double)g->info.m)));
    return (int32_t)std::min(l, 31.0);
}

hnsw_status check_in_step(hnsw_group *g)
{
    if (g->diverged) return gfail(g, HNSW_ERR_INVALID, "group: a replayed write failed on a member; call hnsw_group_refresh");
    hnsw_info a, b;
    hnsw_status s = hnsw_get_info(g->member[0], &a);
    if (s != HNSW_OK) return gfail(g, s, hnsw_last_error(g->member[0]));
    for (size_t i = 1; i < g->member.size(); ++i) {
        if ((s = hnsw_get_info(g->member[i], &b)) != HNSW_OK) return gfail(g, s, hnsw_last_error(g->member[i]));
        if (a.allocated_ids != b.allocated_ids || a.node_count != b.node_count || a.enterpoint != b.enterpoint)
            return gfail(g, HNSW_ERR_INVALID, "group: replica " + std::to_string(i) + " is not in step with the primary (written to directly?); call hnsw_group_refresh");
    }
    g->info = a;
    return HNSW_OK;
}

} // namespace

extern "C" {

hnsw_status hnsw_group_create(hnsw_index *primary, const int *devices, uint32_t n_devices, uint64_t seed, hnsw_group **out)
{
    if (!primary || !out || (n_devices && !devices) || n_devices > 63) return HNSW_ERR_INVALID;
    *out = nullptr;
    hnsw_group *g = new hnsw_group;
    *out = g;                                          // handed back even on failure: hnsw_group_last_error says why
    g->member.push_back(primary);
    g->worker.push_back(nullptr);
    g->st.push_back(HNSW_OK);
    g->seed = seed;
    g->rng.seed(seed ^ 0x6A09E667F3BCC909ull);
    hnsw_status s = hnsw_get_info(primary, &g->info);
    if (s != HNSW_OK) return gfail(g, s, hnsw_last_error(primary));
    hnsw_replica view;
    if ((s = hnsw_replica_view(primary, &view)) != HNSW_OK) return gfail(g, s, hnsw_last_error(primary));
    g->device.push_back(device_of(primary, view));
    for (uint32_t i = 0; i < n_devices; ++i) {
        hnsw_index *r = nullptr;
        if ((s = make_replica(g, devices[i], seed + 1 + i, &r)) != HNSW_OK) return s;
        g->member.push_back(r);
        g->device.push_back(devices[i]);
        g->st.push_back(HNSW_OK);
        Worker *w = new Worker;
        w->start();
        g->worker.push_back(w);
    }
    return HNSW_OK;
}

void hnsw_group_destroy(hnsw_group *g)
{
    if (!g) return;
    for (size_t i = 1; i < g->worker.size(); ++i) {
        g->worker[i]->stop();
        delete g->worker[i];
    }
    for (size_t i = 1; i < g->member.size(); ++i) hnsw_destroy(g->member[i]);
    delete g;
}

const char *hnsw_group_last_error(const hnsw_group *g) { return g ? g->err.c_str() : "null group"; }
uint32_t hnsw_group_size(const hnsw_group *g) { return g ? (uint32_t)g->member.size() : 0u; }
hnsw_index *hnsw_group_member(hnsw_group *g, uint32_t i) { return g && i < g->member.size() ? g->member[i] : nullptr; }

hnsw_status hnsw_group_refresh(hnsw_group *g)
{
    if (!g) return HNSW_ERR_INVALID;
    hnsw_status s = hnsw_get_info(g->member[0], &g->info);
    if (s != HNSW_OK) return gfail(g, s, hnsw_last_error(g->member[0]));
    for (size_t i = 1; i < g->member.size(); ++i) {
        hnsw_index *r = nullptr;
        // the old copy goes first: two replicas of a large index need not fit one device side by side
        hnsw_destroy(g->member[i]);
        g->member[i] = nullptr;
        if ((s = make_replica(g, g->device[i], g->seed + 1 + i, &r)) != HNSW_OK) {
            // keep the group usable: an empty stand-in that check_in_step() will report as behind
            g->diverged = true;
            if (hnsw_create(g->info.dim, g->info.m, g->info.ef_construction, g->seed, g->device[i], &g->member[i]) != HNSW_OK) {
                hnsw_destroy(g->member[i]);                 // no half-initialised member: hnsw_group_member() hands out NULL,
                g->member[i] = nullptr;                     // and every group call refuses until a refresh succeeds
            }
            return s;
        }
        g->member[i] = r;
    }
    g->diverged = false;
    return HNSW_OK;
}

hnsw_status hnsw_group_search_batch(hnsw_group *g, const float *Q, uint32_t B, uint32_t dim, uint32_t k, uint32_t *ids,
                                    float *sims, uint32_t *n_out)
{
    if (!g || (B && (!Q || !ids || !sims || !n_out))) return HNSW_ERR_INVALID;
    hnsw_status s = check_in_step(g);
    if (s != HNSW_OK) return s;
    const uint32_t G = (uint32_t)g->member.size();
    if (G == 1 || B < G) {                              // nothing to shard
        s = hnsw_search_batch(g->member[0], Q, B, dim, k, ids, sims, n_out);
        return s == HNSW_OK ? s : gfail(g, s, hnsw_last_error(g->member[0]));
    }
    // contiguous split: member i takes queries [i*B/G, (i+1)*B/G)
    return on_all(g, [=](uint32_t i) {
        const uint64_t lo = (uint64_t)i * B / G, hi = (uint64_t)(i + 1) * B / G;
        if (hi == lo) return HNSW_OK;
        return hnsw_search_batch(g->member[i], Q + lo * dim, (uint32_t)(hi - lo), dim, k, ids + lo * k, sims + lo * k, n_out + lo);
    });
}

hnsw_status hnsw_group_add(hnsw_group *g, const float *v, uint32_t dim, int32_t level, uint32_t *out_id, uint32_t *touched,
                           uint32_t touched_cap, uint32_t *n_touched)
{
    if (!g || !v) return HNSW_ERR_INVALID;
    hnsw_status s = check_in_step(g);
    if (s != HNSW_OK) return s;
    if (level < 0) level = draw_level(g);              // drawn once: every member inserts at the same level
    // the primary first: argument errors (dimension, non-finite data) surface before any replica is touched
    s = hnsw_add(g->member[0], v, dim, level, out_id, touched, touched_cap, n_touched);
    if (s != HNSW_OK) return gfail(g, s, hnsw_last_error(g->member[0]));
    s = on_all(g, [=](uint32_t i) { return hnsw_add(g->member[i], v, dim, level, nullptr, nullptr, 0, nullptr); }, 1);
    if (s != HNSW_OK) g->diverged = true;
    return s;
}

hnsw_status hnsw_group_delete(hnsw_group *g, uint32_t id, uint32_t *touched, uint32_t touched_cap, uint32_t *n_touched)
{
    if (!g) return HNSW_ERR_INVALID;
    hnsw_status s = check_in_step(g);
    if (s != HNSW_OK) return s;
    s = hnsw_delete(g->member[0], id, touched, touched_cap, n_touched);
    if (s != HNSW_OK) return gfail(g, s, hnsw_last_error(g->member[0]));
    s = on_all(g, [=](uint32_t i) { return hnsw_delete(g->member[i], id, nullptr, 0, nullptr); }, 1);
    if (s != HNSW_OK) g->diverged = true;
    return s;
}

hnsw_status hnsw_group_add_batch(hnsw_group *g, const float *V, uint32_t n, uint32_t dim, const int32_t *levels, uint32_t mode)
{
    if (!g || (n && !V)) return HNSW_ERR_INVALID;
    hnsw_status s = check_in_step(g);
    if (s != HNSW_OK) return s;
    if (mode != 0) {
        // the fast build is not reproducible link for link: build once, copy
        s = hnsw_add_batch(g->member[0], V, n, dim, levels, mode);
        if (s != HNSW_OK) return gfail(g, s, hnsw_last_error(g->member[0]));
        return hnsw_group_refresh(g);
    }
    std::vector<int32_t> lv;
    if (!levels || std::any_of(levels, levels + n, [](int32_t l) { return l < 0; })) {
        lv.resize(n);
        for (uint32_t i = 0; i < n; ++i) lv[i] = levels && levels[i] >= 0 ? levels[i] : draw_level(g);
        levels = lv.data();
    }
    s = on_all(g, [=](uint32_t i) { return hnsw_add_batch(g->member[i], V, n, dim, levels, 0); });
    if (s != HNSW_OK) g->diverged = true;
    return s;
}

} // extern "C"
