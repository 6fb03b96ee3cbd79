// hnsw_engine.hip -- host side of libhnsw_mi355x.so: the C ABI of
// include/hnsw_mi355x.h over the gfx950 kernels.  No CPU compute path exists
// here: every search / insert is a kernel launch, and creation fails without
// a device.  The heavy kernel templates are instantiated in their own
// translation units (hnsw_tu_*.hip, see hnsw_host.hpp); this file holds the
// handle's bookkeeping, the entry points and the small utility kernels.
#define HNSW_UTILITY_KERNELS 1
#define HNSW_SYNC_BLOCK   // search / engine unit: 64-thread workgroups handing over through LDS only (hnsw_device.hpp)
#include "hnsw_host.hpp"
#include "hnsw_kernels.hpp"
#include "hnsw_search_lean.hpp"
#include "hnsw_plan_lean.hpp"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <utility>
#include <vector>

using namespace hnsw;

namespace hnsw_host {

hnsw_status fail(hnsw_index *h, hnsw_status s, const std::string &msg)
{
    h->err = msg;
    return s;
}

uint32_t round_up(uint32_t x, uint32_t a) { return (x + a - 1) / a * a; }

// The reference orders similarities with OrderedFloat (NaN sorts as the nearest, core.rs:4,241); the
// engine's keys are IEEE bit patterns of finite distances, so non-finite components are refused at the
// host entry points instead of producing a different order.
bool all_finite(const float *v, size_t n)
{
    for (size_t i = 0; i < n; ++i)
        if (!std::isfinite(v[i])) return false;
    return true;
}
uint32_t ceil_log2(uint64_t x)
{
    uint32_t b = 0;
    while ((1ull << b) < x) ++b;
    return b;
}

uint64_t splitmix64(uint64_t &x)
{
    uint64_t z = (x += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
inline uint64_t rotl64(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }
uint64_t xoshiro_next(uint64_t *s)
{
    uint64_t result = rotl64(s[1] * 5, 7) * 9, t = s[1] << 17;
    s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3]; s[2] ^= t; s[3] = rotl64(s[3], 45);
    return result;
}
// core.rs:601-605: floor(-ln(U) * level_mult), U in [0,1)
uint32_t draw_level(hnsw_index *h)
{
    double r = (double)(xoshiro_next(h->rng) >> 11) * (1.0 / 9007199254740992.0);
    double l = -std::log(r) * h->level_mult;
    if (!(l < (double)(kMaxLayers - 1))) return kMaxLayers - 1;
    return (uint32_t)l;
}

// device scratch that lives for one API call: freed on every way out of the scope (hipFree waits
// for work still using it)
template <typename Tp>
struct DevScratch {
    Tp *p = nullptr;
    DevScratch() = default;
    DevScratch(const DevScratch &) = delete;
    DevScratch &operator=(const DevScratch &) = delete;
    ~DevScratch() { if (p) (void)hipFree(p); }
    hipError_t alloc(size_t count) { return hipMalloc((void **)&p, std::max<size_t>(count, 1) * sizeof(Tp)); }
};

template <typename Tp>
hnsw_status dev_alloc(hnsw_index *h, Tp **p, size_t count, int fill = -1)
{
    size_t bytes = count * sizeof(Tp);
    if (bytes == 0) bytes = sizeof(Tp);
    HIP_TRY(h, hipMalloc((void **)p, bytes));
    h->hbm_bytes += bytes;
    if (fill >= 0) HIP_TRY(h, hipMemsetAsync(*p, fill, bytes, h->stream));
    return HNSW_OK;
}
template <typename Tp>
void dev_free(hnsw_index *h, Tp *&p, size_t count)
{
    if (!p) return;
    (void)hipFree(p);
    size_t bytes = count * sizeof(Tp);
    h->hbm_bytes -= std::min<uint64_t>(h->hbm_bytes, bytes ? bytes : sizeof(Tp));
    p = nullptr;
}

uint32_t plan_stride(const hnsw_index *h) { return 1u + (h->m > 64 ? kMaxM : 64u); }

GraphView view(const hnsw_index *h)
{
    GraphView g;
    g.vec = h->d_vec;
    g.adj0 = h->d_adj0;
    g.adjU = h->d_adjU;
    g.upper_base = h->d_upper_base;
    g.levels = h->d_levels;
    g.hdr = h->d_hdr;
    g.dim = h->dim;
    g.stride0 = h->stride0;
    g.strideU = h->strideU;
    g.tagcfg = 0;
    g.plan_stride = plan_stride(h);
    g.selcap = h->m > 64 ? kSelMaxWide : kSelMax;
    return g;
}

uint32_t default_stride(uint32_t mmax, uint32_t m, uint32_t maxdeg)
{
    // slot 0 = count; room for m_max plus the over-degree the reference's
    // shrink step leaves on third parties, in 64-byte units
    uint32_t want = std::max(mmax, maxdeg) + std::max(m + 2, mmax / 2);
    return round_up(1 + want, 16);
}

// grow node capacity (vectors, layer-0 rows, levels, upper_base)
hnsw_status ensure_node_cap(hnsw_index *h, uint32_t need)
{
    if (need <= h->cap) return HNSW_OK;
    uint32_t ncap = h->cap ? h->cap : 1024;
    while (ncap < need) ncap *= 2;
    float *nvec = nullptr;
    uint32_t *nadj0 = nullptr, *nub = nullptr, *nlv = nullptr;
    hnsw_status s;
    if ((s = dev_alloc(h, &nvec, (size_t)ncap * h->dim)) != HNSW_OK ||
        (s = dev_alloc(h, &nadj0, (size_t)ncap * h->stride0, 0)) != HNSW_OK ||
        (s = dev_alloc(h, &nub, (size_t)ncap, 0xFF)) != HNSW_OK ||
        (s = dev_alloc(h, &nlv, (size_t)ncap, 0)) != HNSW_OK) {
        // out of HBM part-way: give back what was taken, the index stays as it was
        dev_free(h, nvec, (size_t)ncap * h->dim);
        dev_free(h, nadj0, (size_t)ncap * h->stride0);
        dev_free(h, nub, (size_t)ncap);
        dev_free(h, nlv, (size_t)ncap);
        return s;
    }
    if (h->n) {
        HIP_TRY(h, hipMemcpyAsync(nvec, h->d_vec, (size_t)h->n * h->dim * 4, hipMemcpyDeviceToDevice, h->stream));
        HIP_TRY(h, hipMemcpyAsync(nadj0, h->d_adj0, (size_t)h->n * h->stride0 * 4, hipMemcpyDeviceToDevice, h->stream));
        HIP_TRY(h, hipMemcpyAsync(nub, h->d_upper_base, (size_t)h->n * 4, hipMemcpyDeviceToDevice, h->stream));
        HIP_TRY(h, hipMemcpyAsync(nlv, h->d_levels, (size_t)h->n * 4, hipMemcpyDeviceToDevice, h->stream));
    }
    HIP_TRY(h, hipDeviceSynchronize());   // searches enqueued on caller streams may still read the old tables
    dev_free(h, h->d_vec, (size_t)h->cap * h->dim);
    dev_free(h, h->d_adj0, (size_t)h->cap * h->stride0);
    dev_free(h, h->d_upper_base, (size_t)h->cap);
    dev_free(h, h->d_levels, (size_t)h->cap);
    h->d_vec = nvec; h->d_adj0 = nadj0; h->d_upper_base = nub; h->d_levels = nlv;
    h->cap = ncap;
    return HNSW_OK;
}

hnsw_status ensure_upper_cap(hnsw_index *h, uint32_t need)
{
    if (need <= h->upper_cap) return HNSW_OK;
    uint32_t ncap = h->upper_cap ? h->upper_cap : 256;
    while (ncap < need) ncap *= 2;
    uint32_t *nadj = nullptr;
    hnsw_status s;
    if ((s = dev_alloc(h, &nadj, (size_t)ncap * h->strideU, 0)) != HNSW_OK) return s;
    if (h->upper_used)
        HIP_TRY(h, hipMemcpyAsync(nadj, h->d_adjU, (size_t)h->upper_used * h->strideU * 4, hipMemcpyDeviceToDevice, h->stream));
    HIP_TRY(h, hipDeviceSynchronize());   // searches enqueued on caller streams may still read the old tables
    dev_free(h, h->d_adjU, (size_t)h->upper_cap * h->strideU);
    h->d_adjU = nadj;
    h->upper_cap = ncap;
    return HNSW_OK;
}

// widen adjacency rows (layer 0 and/or upper) keeping their content
hnsw_status restride(hnsw_index *h, uint32_t nstride0, uint32_t nstrideU)
{
    if (nstride0 > h->stride0) {
        uint32_t *nadj = nullptr;
        hnsw_status s = dev_alloc(h, &nadj, (size_t)h->cap * nstride0, 0);
        if (s != HNSW_OK) return s;
        if (h->n) {
            uint64_t rows = h->n;
            uint32_t blocks = (uint32_t)((rows * 64 + 255) / 256);
            hipLaunchKernelGGL(k_restride, dim3(blocks), dim3(256), 0, h->stream, h->d_adj0, h->stride0, nadj, nstride0, rows);
        }
        HIP_TRY(h, hipDeviceSynchronize());   // searches enqueued on caller streams may still read the old tables
        dev_free(h, h->d_adj0, (size_t)h->cap * h->stride0);
        h->d_adj0 = nadj;
        h->stride0 = nstride0;
    }
    if (nstrideU > h->strideU) {
        uint32_t *nadj = nullptr;
        hnsw_status s = dev_alloc(h, &nadj, (size_t)std::max(h->upper_cap, 1u) * nstrideU, 0);
        if (s != HNSW_OK) return s;
        if (h->upper_used) {
            uint64_t rows = h->upper_used;
            uint32_t blocks = (uint32_t)((rows * 64 + 255) / 256);
            hipLaunchKernelGGL(k_restride, dim3(blocks), dim3(256), 0, h->stream, h->d_adjU, h->strideU, nadj, nstrideU, rows);
        }
        HIP_TRY(h, hipDeviceSynchronize());   // searches enqueued on caller streams may still read the old tables
        dev_free(h, h->d_adjU, (size_t)h->upper_cap * h->strideU);
        h->d_adjU = nadj;
        h->strideU = nstrideU;
    }
    return HNSW_OK;
}

int pick_R(uint32_t need)
{
    if (need <= 64) return 1;
    if (need <= 256) return 4;
    if (need <= 512) return 8;
    if (need <= 1024) return 16;
    if (need <= 4096) return 64;                 // W in LDS (search_level_v1): the slow, exact form for what nobody runs in production
    return 0;
}

// Visited-table size (in 32-byte buckets of 7 ids) for one wave.  LDS is what
// limits residency (160 KiB per CU, one wave per query), and a launch ends with
// its slowest wave, so the table takes the whole LDS share of the residency the
// batch needs: all `nwaves` queries resident at once if that is possible at
// <= 4 waves per CU (256 CUs), else 4 per CU.  A query whose set outgrows the
// table continues in HBM (slower, exact).  Expected ids of a layer-0 search
// ~ 0.8 * ef * m_max0 (measured 3.9 k at ef 200 / m_max0 32 on 20 k nodes,
// 5.9 k on 1 M); there is no point in more than 3x that.
uint32_t pick_lnb(const hnsw_index *h, int R, int T, bool ins, uint32_t nwaves)
{
    const size_t fixed = lds_fixed_bytes(R, T, h->dim, ins, h->m > 64 ? kSelMaxWide : kSelMax);
    if (h->lds_buckets_override >= 2) return (uint32_t)h->lds_buckets_override;
    // measured on MI355X: 40448 B per 64-thread block still gives 4 blocks per CU, 40960 B does not
    // (163840 / n - 512); 5..8 per CU follow the same rule, rounded down to 256 B
    const size_t tiers[16] = {160 * 1024 - 2048, 80896, 53760, 40448, 32256, 26624, 22784, 19968,
                              17664, 15872, 14336, 13056, 12032, 11008, 10240, 9728};
    // the insert kernels keep the exact HBM spill path: give them the larger table
    // ... and so does the dim-768 search: its kernel holds the query and 24 loads in registers (> 256 VGPRs:
    // one wave per SIMD whatever the LDS share), so a smaller table would only forget more
    const uint32_t max_per_cu = (ins || T == 24) ? std::min(h->max_waves_per_cu, 4u) : h->max_waves_per_cu;
    if (!ins) nwaves *= std::max(h->cur_conc, 1u);   // search launches in flight at once share the CUs (search_concurrency)
    uint32_t per_cu = (nwaves + 255) / 256;
    per_cu = std::min(std::max(per_cu, 1u), max_per_cu);
    size_t budget = tiers[per_cu - 1];
    while (budget <= fixed + 64 && per_cu > 1) budget = tiers[--per_cu - 1];
    if (h->lds_reserve) budget = std::min(budget, tiers[0] - h->lds_reserve);   // one block per CU still has to fit
    if (budget <= fixed + 64) return 2u;
    const uint32_t fit = (uint32_t)((budget - fixed) / 32);
    const uint32_t useful = (uint32_t)(0.8 * (double)h->efc * (double)h->m_max0 * 3.0 / 6.0) + 2;
    return std::max(std::min(fit, useful), 2u);
}

VisCfg pick_vis(const hnsw_index *h, int R, int T, bool ins, uint32_t nwaves)
{
    VisCfg c;
    c.lnb = pick_lnb(h, R, T, ins, nwaves);
    c.lcap = c.lnb * h->lds_fill_x2 / 2;
    c.tagcfg = 0;
    c.bytes = (size_t)c.lnb * 32;
    if (!h->tag_table || (h->lds_buckets_override >= 2 && h->tag_bb_override < 2)) return c;
    // tag mode: the largest power-of-two count of 16-byte buckets in the same LDS share, if the
    // id range fits 13 tag bits (ids < 2^(bb+13): 16 M nodes at 2048 buckets)
    uint32_t bb = 0;
    while (((size_t)16 << (bb + 1)) <= c.bytes && bb + 1 <= 12) ++bb;
    if (h->tag_bb_override >= 2) bb = std::min<uint32_t>(bb, (uint32_t)h->tag_bb_override);
    if (bb < 2) return c;
    uint32_t idbits = std::max(ceil_log2(std::max(h->cap, 2u)), bb);
    // tests: any idbits >= ceil_log2(cap) is a valid bijection domain (what a larger index would use)
    if (h->idbits_override > (int)idbits && h->idbits_override <= 32) idbits = (uint32_t)h->idbits_override;
    if (idbits - bb > 13) return c;
    c.tagcfg = bb | (idbits << 8);
    c.lcap = (1u << bb) * 6u;                     // 6 of the 7 entries per bucket
    c.bytes = (size_t)16 << bb;
    return c;
}

constexpr uint32_t kSpillRegions = 4;
constexpr uint32_t kPinnedBatch = 64;

// Pick the next spill region for a launch on `st`; the stream first waits for the launch that used
// that region last.  Call spill_release() right after enqueueing the kernel.
hnsw_status spill_acquire(hnsw_index *h, hipStream_t st, uint32_t *region, uint32_t **base)
{
    const uint32_t r = h->spill_rr++ % kSpillRegions;
    if (h->spill_busy[r]) HIP_TRY(h, hipStreamWaitEvent(st, h->spill_ev[r], 0));
    *region = r;
    *base = h->d_spill + (size_t)r * h->spill_slots * h->spill_gnb * 8;
    return HNSW_OK;
}
hnsw_status spill_release(hnsw_index *h, hipStream_t st, uint32_t region)
{
    HIP_TRY(h, hipEventRecord(h->spill_ev[region], st));
    h->spill_busy[region] = true;
    return HNSW_OK;
}
// a search without a spill region: remember its stream's latest launch (one slot per distinct stream)
hnsw_status note_search(hnsw_index *h, hipStream_t st)
{
    uint32_t slot = hnsw_index::kSearchStreams;
    for (uint32_t i = 0; i < hnsw_index::kSearchStreams; ++i)
        if (h->search_busy[i] && h->search_st[i] == st) { slot = i; break; }
    if (slot == hnsw_index::kSearchStreams)
        for (uint32_t i = 0; i < hnsw_index::kSearchStreams; ++i)
            if (!h->search_busy[i]) { slot = i; break; }
    if (slot == hnsw_index::kSearchStreams) {            // more distinct streams than slots: retire one
        slot = h->search_rr++ % hnsw_index::kSearchStreams;
        HIP_TRY(h, hipEventSynchronize(h->search_ev[slot]));
    }
    if (!h->search_ev[slot]) HIP_TRY(h, hipEventCreateWithFlags(&h->search_ev[slot], hipEventDisableTiming));
    HIP_TRY(h, hipEventRecord(h->search_ev[slot], st));
    h->search_st[slot] = st;
    h->search_busy[slot] = true;
    return HNSW_OK;
}
// an insert must not start while searches enqueued on other streams are still reading the graph
hnsw_status wait_inflight_searches(hnsw_index *h)
{
    for (uint32_t r = 0; r < kSpillRegions; ++r)
        if (h->spill_busy[r]) HIP_TRY(h, hipStreamWaitEvent(h->stream, h->spill_ev[r], 0));
    for (uint32_t i = 0; i < hnsw_index::kSearchStreams; ++i)
        if (h->search_busy[i]) {
            HIP_TRY(h, hipStreamWaitEvent(h->stream, h->search_ev[i], 0));
            h->search_busy[i] = false;                   // ordered behind it from here on (h->stream is in order)
        }
    return HNSW_OK;
}

__global__ void k_fill_buckets(uint4 *t, size_t n16)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    // piece i heads a bucket iff i is even; the stride is even, so a thread's parity is constant
    const uint4 e = make_uint4((i & 1) ? kEmpty : 0u, kEmpty, kEmpty, kEmpty);
    for (; i < n16; i += (size_t)gridDim.x * blockDim.x) t[i] = e;
}

hnsw_status ensure_spill(hnsw_index *h)
{
    // A visited set never holds more ids than the index has nodes, nor (much)
    // more than ef * row width; twice the smaller bound keeps the load <= 1/2.
    // Sized from the node CAPACITY so that it stays valid while the index grows.
    uint64_t want = 4ull * h->efc * std::max(h->stride0, 16u);
    want = std::min<uint64_t>(want, 2ull * std::max(h->cap, 1024u));
    uint32_t gnb = (uint32_t)(want / 6) + 16;           // 6 of the 7 id slots per bucket
    const uint32_t slots = 2048;
    if (h->d_spill && h->spill_gnb >= gnb && h->spill_slots >= slots) return HNSW_OK;
    HIP_TRY(h, hipDeviceSynchronize());          // nothing may still be using the old tables
    dev_free(h, h->d_spill, (size_t)kSpillRegions * h->spill_slots * h->spill_gnb * 8);
    hnsw_status s = dev_alloc(h, &h->d_spill, (size_t)kSpillRegions * slots * gnb * 8);
    if (s != HNSW_OK) return s;
    hipLaunchKernelGGL(k_fill_buckets, dim3(4096), dim3(256), 0, h->stream, reinterpret_cast<uint4 *>(h->d_spill),
                       (size_t)kSpillRegions * slots * gnb * 2);
    HIP_TRY(h, hipGetLastError());
    for (uint32_t r = 0; r < kSpillRegions; ++r) h->spill_busy[r] = false;
    h->spill_gnb = gnb;
    h->spill_slots = slots;
    return HNSW_OK;
}

// The exact insert / delete kernels are one wave: their table is sized so that it cannot overflow
// (a visited set never holds more ids than the index has nodes; load <= 1/2).
hnsw_status ensure_spill_one(hnsw_index *h)
{
    const uint32_t gnb = (uint32_t)(2ull * std::max(h->cap, 1024u) / 6) + 16;
    if (h->d_spill_one && h->spill_one_gnb >= gnb) return HNSW_OK;
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    dev_free(h, h->d_spill_one, (size_t)h->spill_one_gnb * 8);
    h->spill_one_gnb = 0;
    hnsw_status s = dev_alloc(h, &h->d_spill_one, (size_t)gnb * 8);
    if (s != HNSW_OK) return s;
    hipLaunchKernelGGL(k_fill_buckets, dim3(1024), dim3(256), 0, h->stream, reinterpret_cast<uint4 *>(h->d_spill_one),
                       (size_t)gnb * 2);
    HIP_TRY(h, hipGetLastError());
    h->spill_one_gnb = gnb;
    return HNSW_OK;
}

// The specialised kernel's preconditions that do not depend on the launch (nullptr = all hold).  Indexes it cannot
// serve (other dims, ef > 512, rows wider than 127 ids, test overrides, fp8 storage) use the general kernel.
const char *lean_blocker(const hnsw_index *h)
{
    const int R = pick_R(h->efc);
    if (!h->lean) return "tuning lean = 0";
    if (h->mode != MODE_AVX || h->dim != 128) return "dim != 128";
    if (!h->visited_bounded) return "tuning visited_bounded = 0";
    if (!h->tag_table || h->tag_bb_override >= 0 || h->lds_buckets_override >= 0) return "visited-table test overrides (tag_table / tag_bb / lds_buckets)";
    if (h->stride0 > 128 || h->strideU > 128) return "adjacency rows wider than 127 ids";
    if (R != 1 && R != 4 && R != 8) return "ef_construction > 512";
    // 16-bit entries: tag (idbits - bb bits) + >= 2 displacement bits; bb is 9..11 (try_launch_lean picks the largest table the id range needs)
    uint32_t idbits = std::max(ceil_log2(std::max(h->cap, 2u)), 11u);
    if (h->idbits_override > (int)idbits && h->idbits_override <= 31) idbits = (uint32_t)h->idbits_override;
    if (idbits > 25) return "more than 2^25 node ids";   // 2048 buckets + 14 tag bits
    return nullptr;
}

// The insert plans' form of the same question (hnsw_plan_lean.hpp): 0 = the general plan kernels, else the width of
// the id hash.  One plan per workgroup: always the 2048-bucket table with 3 displacement bits (ids < 2^24).
uint32_t plan_lean_idbits(const hnsw_index *h, const InsertCfg &c)
{
    if (!h->plan_lean || h->mode != MODE_AVX || h->dim != 128 || h->fmt != FMT_F32) return 0;
    if (h->stride0 > 128 || h->strideU > 128) return 0;
    if (c.R != 1 && c.R != 4 && c.R != 8) return 0;
    if ((uint32_t)c.R * 64u < h->efc) return 0;
    uint32_t idbits = std::max(ceil_log2(std::max(h->cap, 2u)), 11u);
    if (h->idbits_override > (int)idbits && h->idbits_override <= 31) idbits = (uint32_t)h->idbits_override;
    if (idbits > 24) return 0;
    if (c.lds + plan_lean_lds(c.R) > 160 * 1024 - 2048) return 0;
    return idbits;
}
size_t plan_lean_lds(int R)
{
    return R == 1 ? plan_lean_bytes<1>() : (R == 4 ? plan_lean_bytes<4>() : plan_lean_bytes<8>());
}
hnsw_status launch_occ_plan_lean(hnsw_index *h, const InsertCfg &c, const OccBufs &ob, uint32_t head, uint32_t count, bool *done)
{
    *done = false;
    const uint32_t idbits = plan_lean_idbits(h, c);
    if (!idbits) return HNSW_OK;
    *done = true;
    const bool wide = h->stride0 > 64 || h->strideU > 64;
    if (h->plan_duo && count <= h->plan_duo_max)
        return wide ? launch_occ_plan_duo_v<true>(h, c, ob, head, count, idbits) : launch_occ_plan_duo_v<false>(h, c, ob, head, count, idbits);
    return wide ? launch_occ_plan_lean_v<true>(h, c, ob, head, count, idbits) : launch_occ_plan_lean_v<false>(h, c, ob, head, count, idbits);
}

// returns HNSW_OK and sets *done when the specialised kernel was launched
hnsw_status try_launch_lean(hnsw_index *h, const float *dQ, uint32_t B, uint32_t k, uint32_t *d_ids, float *d_sims,
                            uint32_t *d_nout, hipStream_t st, bool *done)
{
    *done = false;
    const int R = pick_R(h->efc);
    if (lean_blocker(h)) return HNSW_OK;
    uint32_t per_cu = ((uint64_t)B * std::max(h->cur_conc, 1u) + 255) / 256;
    // residency the table is sized for: 8 waves per CU (two per SIMD: the f32 kernel's 177 VGPRs allow no more, and
    // it is at the memory system's gather ceiling there); the bf16 kernel needs 159 VGPRs and is bound by its
    // instruction stream, not by HBM -- three waves per SIMD (12 per CU, 8 KB table) measured 3.24 -> 4.0-4.1 M QPS
    // at C2 (f32 with a 168-VGPR build: 2.54 -> 2.55, and 2 % slower alone on the chip)
    const uint32_t max_wpc = (h->fmt && !h->wpc_user) ? 12u : h->max_waves_per_cu;   // the bf16 / fp8 forms fit three waves per SIMD
    per_cu = std::min(std::max(per_cu, 1u), max_wpc);
    // 32 KB table at <= 4 waves per CU, 16 KB at <= 8, 8 KB beyond
    uint32_t bb = per_cu >= 9 ? 9 : (per_cu >= 5 ? 10 : 11);
    uint32_t idbits = std::max(ceil_log2(std::max(h->cap, 2u)), 11u);
    if (h->idbits_override > (int)idbits && h->idbits_override <= 31) idbits = (uint32_t)h->idbits_override;
    // 16-bit entries: tag (idbits - bb bits) + displacement.  3 displacement bits normally; 14 tag bits leave 2
    // (chains of at most 3 buckets: more ids go unrecorded, see tagset_visit -- results are unaffected).  An id
    // range too wide for the smaller tables takes the next larger one (fewer waves per CU).
    while (bb < 11 && idbits - bb > ((bb == 9) ? 13u : 14u)) ++bb;
    uint32_t db = 3;
    if (idbits - bb > 13) {
        if (idbits - bb == 14) db = 2;
        else return HNSW_OK;
    }
    const bool wide = h->stride0 > 64 || h->strideU > 64;   // rows of 64..127 ids: two row words per lane
    h->last_search_duo = false;
    if (h->duo && !h->tie_census && h->fmt == FMT_F32 && (uint64_t)B * std::max(h->cur_conc, 1u) <= h->duo_max && idbits - 11 <= 14) {
        // few enough queries in flight that each can have two SIMDs: a walker wave and a W-keeper wave per query
        const uint32_t db2 = idbits - 11 > 13 ? 2u : 3u;
        hnsw_status s2 = wide ? launch_duo_v<true>(h, R, db2, dQ, B, k, idbits, d_ids, d_sims, d_nout, st, done)
                              : launch_duo_v<false>(h, R, db2, dQ, B, k, idbits, d_ids, d_sims, d_nout, st, done);
        if (s2 != HNSW_OK || *done) { h->last_search_duo = *done; return s2; }
    }
#define LEAN_GO(VEC)                                                                                                          \
    return wide ? launch_lean_v<VEC, true>(h, R, bb, db, dQ, B, k, idbits, per_cu, d_ids, d_sims, d_nout, st, done)          \
                : launch_lean_v<VEC, false>(h, R, bb, db, dQ, B, k, idbits, per_cu, d_ids, d_sims, d_nout, st, done)
    if (h->fmt == FMT_BF16) LEAN_GO(VecBF16<4>);
    if (h->fmt == FMT_FP8) LEAN_GO(VecFP8<4>);
    LEAN_GO(VecF32<4>);
#undef LEAN_GO
}

// How many search launches share the CUs with the one about to be enqueued on `st` (it sizes the LDS visited
// table: 32 KB at <= 4 waves per CU, 16 KB at <= 8).  The engine's own pipeline knows (pipe_inflight); an
// explicit "launch_concurrency" tuning is taken as the caller's promise; otherwise it is observed: this launch
// plus every OTHER stream whose latest search has not finished yet (launches on one stream run one after the
// other).  A wrong guess costs speed only, never exactness.
uint32_t search_concurrency(hnsw_index *h, hipStream_t st)
{
    if (h->pipe_inflight > 1) return std::max(h->pipe_inflight, h->launch_concurrency);
    if (h->launch_concurrency) return h->launch_concurrency;
    uint32_t n = 1;
    for (uint32_t i = 0; i < hnsw_index::kSearchStreams; ++i)
        if (h->search_busy[i] && h->search_st[i] != st && hipEventQuery(h->search_ev[i]) == hipErrorNotReady) ++n;
    (void)hipGetLastError();                           // hipErrorNotReady is an answer, not a failure
    return std::min(n, 8u);
}

static hnsw_status launch_search_kernels(hnsw_index *h, const float *dQ, uint32_t B, uint32_t k, uint32_t *d_ids,
                                         float *d_sims, uint32_t *d_nout, hipStream_t st);

// tuning tie_mode: after the search kernel, the queries its census flagged (1), or all of them (2, or a shape without a
// census kernel), are answered again in the reference binary's own tie order (hnsw_std_heap.hpp); same stream, asynchronous
hnsw_status launch_search(hnsw_index *h, const float *dQ, uint32_t B, uint32_t k, uint32_t *d_ids,
                          float *d_sims, uint32_t *d_nout, hipStream_t st)
{
    if (!h->tie_mode) return launch_search_kernels(h, dQ, B, k, d_ids, d_sims, d_nout, st);
    if (h->fmt) return fail(h, HNSW_ERR_INVALID, "tie_mode serves f32 rows only (a bf16 / fp8 index keeps no f32 vectors: there is no reference arithmetic to reproduce)");
    hnsw_status s = ensure_tie_flags(h, B);
    if (s != HNSW_OK) return s;
    const bool was = h->tie_uncounted;
    h->tie_uncounted = false;
    if ((s = launch_search_kernels(h, dQ, B, k, d_ids, d_sims, d_nout, st)) != HNSW_OK) return s;
    const bool all = h->tie_mode == 2 || h->tie_uncounted;
    h->tie_uncounted = h->tie_uncounted || was;
    return launch_search_std(h, dQ, B, k, d_ids, d_sims, d_nout, all, st);
}

static hnsw_status launch_search_kernels(hnsw_index *h, const float *dQ, uint32_t B, uint32_t k, uint32_t *d_ids,
                                         float *d_sims, uint32_t *d_nout, hipStream_t st)
{
    hnsw_status s = ensure_spill(h);
    if (s != HNSW_OK) return s;
    h->cur_conc = search_concurrency(h, st);
    bool done = false;
    h->last_search_lean = false;
    if ((s = try_launch_lean(h, dQ, B, k, d_ids, d_sims, d_nout, st, &done)) != HNSW_OK) return s;
    if (done) { h->last_search_lean = true; return HNSW_OK; }
    if (h->tie_census) h->tie_uncounted = true;           // the general kernel has no census form
    const int R = pick_R(h->efc);
    if (h->fmt == FMT_BF16) s = launch_search_fmt<FMT_BF16>(h, R, dQ, B, k, d_ids, d_sims, d_nout, st);   // any dim % 32 == 0, any M / ef
    else if (h->fmt == FMT_FP8) s = launch_search_fmt<FMT_FP8>(h, R, dQ, B, k, d_ids, d_sims, d_nout, st);
    else if (h->mode == MODE_SCALAR) s = launch_search_r<MODE_SCALAR, 0>(h, R, dQ, B, k, d_ids, d_sims, d_nout, st);
    else if (h->T == 4) s = launch_search_r<MODE_AVX, 4>(h, R, dQ, B, k, d_ids, d_sims, d_nout, st);
    else if (h->T == 24) s = launch_search_r<MODE_AVX, 24>(h, R, dQ, B, k, d_ids, d_sims, d_nout, st);
    else s = launch_search_r<MODE_AVX, 0>(h, R, dQ, B, k, d_ids, d_sims, d_nout, st);
    return s != HNSW_OK ? s : note_search(h, st);
}

#include "hnsw_pipeline.inc"   // the engine's own search pipeline (lanes, overlap probe, pinned staging)


hnsw_status push_header(hnsw_index *h)
{
    DevHeader hd;
    HIP_TRY(h, hipMemcpyAsync(&hd, h->d_hdr, sizeof hd, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    hd.node_count = h->n;
    hd.max_layer = h->max_layer;
    hd.enterpoint = (int32_t)h->enterpoint;
    hd.max_deg0 = h->max_deg0;
    hd.max_degU = h->max_degU;
    HIP_TRY(h, hipMemcpyAsync(h->d_hdr, &hd, sizeof hd, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return HNSW_OK;
}

hnsw_status pull_header(hnsw_index *h, DevHeader *out = nullptr)
{
    DevHeader hd;
    HIP_TRY(h, hipMemcpyAsync(&hd, h->d_hdr, sizeof hd, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    h->n = hd.node_count;
    h->max_layer = hd.max_layer;
    h->enterpoint = hd.enterpoint;
    h->max_deg0 = hd.max_deg0;
    h->max_degU = hd.max_degU;
    if (out) *out = hd;
    return HNSW_OK;
}

// The status word is per call: whatever a call reports is cleared on the device, so one failed (or merely
// informational) call does not fail every later one.
hnsw_status check_dev_status(hnsw_index *h, const DevHeader &hd)
{
    if (hd.status == 0) return HNSW_OK;
    HIP_TRY(h, hipMemsetAsync((char *)h->d_hdr + offsetof(DevHeader, status), 0, sizeof(uint32_t), h->stream));
    char buf[200];
    snprintf(buf, sizeof buf, "device status 0x%x:%s%s%s%s%s", hd.status,
             (hd.status & ST_STD_OVERFLOW) ? " std-order search heap overflow" : "",
             (hd.status & ST_VISITED_OVERFLOW) ? " visited-set overflow" : "",
             (hd.status & ST_ROW_OVERFLOW) ? " adjacency row overflow" : "",
             (hd.status & ST_ROW_DROPPED) ? " reverse link dropped" : "",
             (hd.status & ST_ASYMMETRIC) ? " asymmetric link" : "");
    // ROW_DROPPED is informational for the fast build, and so is a missing back link on a graph the
    // fast build produced (the reference would panic there, core.rs:150; its own graphs are symmetric)
    uint32_t benign = ST_ROW_DROPPED | (h->asymmetric ? ST_ASYMMETRIC : 0u);
    if ((hd.status & ~benign) == 0) return HNSW_OK;
    return fail(h, HNSW_ERR_CAPACITY, buf);
}

#include "hnsw_insert_host.inc"

} // namespace hnsw_host

using namespace hnsw_host;

// ===========================================================================
// C ABI
// ===========================================================================
extern "C" {

hnsw_status hnsw_create(uint32_t dim, uint32_t m, uint32_t ef_construction, uint64_t seed, int device,
                        hnsw_index **out)
{
    if (!out) return HNSW_ERR_INVALID;
    *out = nullptr;
    hnsw_index *h = new hnsw_index();
    *out = h; // returned even on failure so the caller can read hnsw_last_error()
    if (dim == 0 || m < 2 || m > kMaxM || ef_construction == 0 || ef_construction > 4096)
        return fail(h, HNSW_ERR_INVALID, "supported: dim >= 1, 2 <= M <= 128, 1 <= EFCON <= 4096");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(h, HNSW_ERR_DEVICE, "no HIP device: this engine has no CPU path");
    if (device < 0 || device >= ndev) return fail(h, HNSW_ERR_DEVICE, "bad device ordinal");
    DeviceScope dev_scope_(device);
    HIP_TRY(h, dev_scope_.err);
    hipDeviceProp_t prop;
    HIP_TRY(h, hipGetDeviceProperties(&prop, device));
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0 && !std::getenv("HNSW_ALLOW_ANY_ARCH"))
        return fail(h, HNSW_ERR_DEVICE, std::string("built for gfx950, device is ") + prop.gcnArchName);
    h->device = device;
    h->dim = dim;
    h->m = m;
    h->m_max = m;                               // core.rs:335
    h->m_max0 = 2 * m;                          // core.rs:336
    h->efc = ef_construction;                   // core.rs:337
    h->level_mult = 1.0 / std::log((double)m);  // core.rs:338
    h->mode = (dim % 32 == 0) ? MODE_AVX : MODE_SCALAR; // metrics.rs:18
    h->T = (h->mode == MODE_AVX && (dim == 128 || dim == 768)) ? (int)(dim / 32) : 0;
    uint64_t x = seed;
    for (int i = 0; i < 4; ++i) h->rng[i] = splitmix64(x);
    HIP_TRY(h, hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
    HIP_TRY(h, hipEventCreate(&h->ev0));
    HIP_TRY(h, hipEventCreate(&h->ev1));
    HIP_TRY(h, hipEventCreateWithFlags(&h->ev_sync, hipEventDisableTiming));
    for (uint32_t r = 0; r < kSpillRegions; ++r) HIP_TRY(h, hipEventCreateWithFlags(&h->spill_ev[r], hipEventDisableTiming));
    h->stride0 = default_stride(h->m_max0, m, 0);
    h->strideU = default_stride(h->m_max, m, 0);
    if (m > 64) {
        // the reference does not bound M (core.rs:322-347): above 64 the plans' lists no longer fit one id per lane, so
        // every insert and delete takes the serial kernels (k_insert_plan / k_insert_commit_exact / k_delete_exact),
        // which walk such lists 64 at a time; searches take the general kernel (rows wider than 127 ids)
        h->wide_m = true;
        h->occ_window = 0;
        h->plan_lean = 0;
    }
    hnsw_status s;
    if ((s = dev_alloc(h, &h->d_hdr, 1, 0)) != HNSW_OK) return s;
    h->enterpoint = -1;
    if ((s = ensure_node_cap(h, 1024)) != HNSW_OK) return s;
    if ((s = ensure_upper_cap(h, 256)) != HNSW_OK) return s;
    if ((s = push_header(h)) != HNSW_OK) return s;
    return HNSW_OK;
}

void hnsw_destroy(hnsw_index *h)
{
    if (!h) return;
    DeviceScope dev_scope_(h->device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    (void)hipFree(h->d_vec); (void)hipFree(h->d_adj0); (void)hipFree(h->d_adjU);
    (void)hipFree(h->d_upper_base); (void)hipFree(h->d_levels); (void)hipFree(h->d_hdr);
    (void)hipFree(h->d_spill); (void)hipFree(h->d_spill_one); (void)hipFree(h->d_Q); (void)hipFree(h->d_res);
    if (h->h_ins) (void)hipHostFree(h->h_ins);
    if (h->h_pinned) (void)hipHostFree(h->h_pinned);
    for (uint32_t l = 0; l < hnsw_index::kPipe; ++l) {
        if (h->pipe_st[l]) { (void)hipStreamSynchronize(h->pipe_st[l]); (void)hipStreamDestroy(h->pipe_st[l]); }
        if (h->pipe_done[l]) (void)hipEventDestroy(h->pipe_done[l]);
        (void)hipFree(h->pipe_dq[l]); (void)hipFree(h->pipe_dres[l]);
        if (h->pipe_hq[l]) (void)hipHostFree(h->pipe_hq[l]);
        if (h->pipe_hres[l]) (void)hipHostFree(h->pipe_hres[l]);
    }
    if (h->pipe_fork) (void)hipEventDestroy(h->pipe_fork);
    (void)hipFree(h->d_plan); (void)hipFree(h->d_touched); (void)hipFree(h->d_work);
    (void)hipFree(h->d_occ_slots); (void)hipFree(h->d_occ_reads); (void)hipFree(h->d_occ_shr); (void)hipFree(h->d_occ_ring); (void)hipFree(h->d_occ_ctl);
    (void)hipFree(h->d_par); (void)hipFree(h->d_par_delta); (void)hipFree(h->d_par_rows);
    if (h->ev0) (void)hipEventDestroy(h->ev0);
    if (h->ev1) (void)hipEventDestroy(h->ev1);
    if (h->ev_sync) (void)hipEventDestroy(h->ev_sync);
    for (uint32_t r = 0; r < kSpillRegions; ++r) if (h->spill_ev[r]) (void)hipEventDestroy(h->spill_ev[r]);
    for (uint32_t i = 0; i < hnsw_index::kSearchStreams; ++i) if (h->search_ev[i]) (void)hipEventDestroy(h->search_ev[i]);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
}

const char *hnsw_last_error(const hnsw_index *h) { return h ? h->err.c_str() : "null handle"; }

hnsw_status hnsw_set_tuning(hnsw_index *h, const char *key, int64_t value)
{
    if (!h || !key) return HNSW_ERR_INVALID;
    if (!std::strcmp(key, "force_restride")) {   // tests: widen both adjacency tables by `value` words now
        ON_DEVICE(h);
        return restride(h, h->stride0 + (uint32_t)value, h->strideU + (uint32_t)value);
    }
    if (!std::strcmp(key, "tag_table")) { h->tag_table = value != 0; return HNSW_OK; }
    if (!std::strcmp(key, "tag_bb")) { h->tag_bb_override = (int)value; return HNSW_OK; }
    if (!std::strcmp(key, "idbits")) { h->idbits_override = (int)value; return HNSW_OK; }
    if (!std::strcmp(key, "lds_fill_x2")) { h->lds_fill_x2 = (uint32_t)std::min<int64_t>(std::max<int64_t>(value, 2), 13); return HNSW_OK; }
    if (!std::strcmp(key, "lds_buckets")) { h->lds_buckets_override = (int)value; return HNSW_OK; }
    if (!std::strcmp(key, "lds_hash_bits")) { h->lds_buckets_override = std::max<int>(2, (int)((1ll << value) / 8)); return HNSW_OK; }
    if (!std::strcmp(key, "grid")) { h->grid_override = (int)value; return HNSW_OK; }
    if (!std::strcmp(key, "time_launches")) { h->time_launches = value != 0; return HNSW_OK; }
    if (!std::strcmp(key, "launch_concurrency")) { h->launch_concurrency = (uint32_t)std::min<int64_t>(std::max<int64_t>(value, 0), 8); return HNSW_OK; }
    if (!std::strcmp(key, "pipe_chunk")) { h->pipe_chunk = (uint32_t)std::min<int64_t>(std::max<int64_t>(value, 64), 1 << 20); return HNSW_OK; }
    if (!std::strcmp(key, "pipe_min_batch")) { h->pipe_min_batch = (uint32_t)std::max<int64_t>(value, 2); return HNSW_OK; }
    if (!std::strcmp(key, "pipe_device")) { h->pipe_device = value != 0; return HNSW_OK; }
    if (!std::strcmp(key, "compress_bf16") || !std::strcmp(key, "compress_fp8")) {
        // One way: the f32 vector matrix becomes a bf16 (2 bytes per component, round to nearest even) or fp8 (1 byte,
        // e4m3) one and the index read-only.  Separate, clearly-labelled serving modes: a half / a
        // quarter of the bytes of the gather; the arithmetic stays the reference's f32 kernel on the stored values
        // widened back (exactly), so results are those of the reference run on the ROUNDED vectors, not on the
        // original ones (hnsw_get_vector reports the stored values).  Any dim % 32 == 0 (the AVX2 summation order),
        // any M / ef_construction: dim-128 indexes the specialised kernel can serve use its bf16 form, everything
        // else the general kernel's.
        const int want = !std::strcmp(key, "compress_bf16") ? FMT_BF16 : FMT_FP8;
        if (!value || h->fmt == want) return HNSW_OK;
        if (h->fmt) return fail(h, HNSW_ERR_INVALID, "the index is already compressed (the original vectors are gone)");
        if (h->mode != MODE_AVX) return fail(h, HNSW_ERR_INVALID, std::string(key) + " needs dim % 32 == 0 (the AVX2 summation order, metrics.rs:18)");
        if (h->efc > 1024) return fail(h, HNSW_ERR_INVALID, std::string(key) + " serves ef_construction <= 1024");
        ON_DEVICE(h);
        HIP_TRY(h, hipDeviceSynchronize());
        const size_t nel = (size_t)h->cap * h->dim, esz = want == FMT_BF16 ? 2 : 1;
        void *dnew = nullptr;
        HIP_TRY(h, hipMalloc(&dnew, std::max<size_t>(nel, 1) * esz));
        if (want == FMT_BF16)
            hipLaunchKernelGGL(k_f32_to_bf16, dim3(4096), dim3(256), 0, h->stream, h->d_vec, (unsigned short *)dnew, (size_t)h->n * h->dim);
        else
            hipLaunchKernelGGL(k_f32_to_fp8, dim3(4096), dim3(256), 0, h->stream, h->d_vec, (unsigned char *)dnew, (size_t)h->n * h->dim);
        HIP_TRY(h, hipGetLastError());
        HIP_TRY(h, hipStreamSynchronize(h->stream));
        (void)hipFree(h->d_vec);
        h->hbm_bytes -= std::min<uint64_t>(h->hbm_bytes, nel * (4 - esz));
        h->d_vec = reinterpret_cast<float *>(dnew);
        h->fmt = want;
        h->bf16 = want == FMT_BF16;
        h->T = 0;                                    // the compressed kernels take the query from LDS
        return HNSW_OK;
    }
    if (!std::strcmp(key, "select_shortcut")) { h->select_shortcut = value != 0; return HNSW_OK; }
    if (h->wide_m && (!std::strcmp(key, "occ_window") || !std::strcmp(key, "plan_lean") || !std::strcmp(key, "single_window")))
        return HNSW_OK;                                  // M > 64: the serial kernels only (hnsw_create); these knobs stay off
    if (!std::strcmp(key, "plan_lean")) { h->plan_lean = value != 0; return HNSW_OK; }
    if (!std::strcmp(key, "single_window")) { h->single_window = value != 0; return HNSW_OK; }
    if (!std::strcmp(key, "occ_window")) { h->occ_window = (uint32_t)std::min<int64_t>(std::max<int64_t>(value, 0), kOccMaxW); return HNSW_OK; }
    if (!std::strcmp(key, "occ_ahead_x10")) { h->occ_ahead_x10 = (uint32_t)std::max<int64_t>(value, 5); return HNSW_OK; }
    if (!std::strcmp(key, "occ_stage_ahead")) { h->occ_stage_ahead = (uint32_t)std::min<int64_t>(std::max<int64_t>(value, 0), kOccMaxW); return HNSW_OK; }
    if (!std::strcmp(key, "occ_depth_x10")) { h->occ_depth_x10 = (uint32_t)std::max<int64_t>(value, 0); return HNSW_OK; }
    if (!std::strcmp(key, "occ_front_max")) { h->occ_front_max = (uint32_t)std::min<int64_t>(std::max<int64_t>(value, 2), kOccMaxW); return HNSW_OK; }
    if (!std::strcmp(key, "occ_slack_base")) { h->occ_slack_base = (uint32_t)std::min<int64_t>(std::max<int64_t>(value, 0), 64); return HNSW_OK; }
    if (!std::strcmp(key, "occ_slack_extra")) { h->occ_slack_extra = (uint32_t)std::min<int64_t>(std::max<int64_t>(value, 0), 64); return HNSW_OK; }
    if (!std::strcmp(key, "occ_log_cap")) { h->occ_log_cap = (uint32_t)std::min<int64_t>(std::max<int64_t>(value, 1), kOccMaxReads); return HNSW_OK; }
    if (!std::strcmp(key, "occ_min_batch")) { h->occ_min_batch = (uint32_t)std::max<int64_t>(value, 1); return HNSW_OK; }
    if (!std::strcmp(key, "lean")) { h->lean = value != 0; return HNSW_OK; }
    if (!std::strcmp(key, "tie_census")) { h->tie_census = value != 0; return HNSW_OK; }
    if (!std::strcmp(key, "tie_mode")) { h->tie_mode = value < 0 ? 0 : (value > 2 ? 2 : (int)value); if (h->tie_mode == 1) h->tie_census = true; return HNSW_OK; }
    if (!std::strcmp(key, "duo")) { h->duo = value != 0; return HNSW_OK; }
    if (!std::strcmp(key, "plan_duo")) { h->plan_duo = value != 0; return HNSW_OK; }
    if (!std::strcmp(key, "occ_chain")) { h->occ_chain = (uint32_t)std::min<int64_t>(std::max<int64_t>(value, 1), 16); return HNSW_OK; }
    if (!std::strcmp(key, "plan_split")) { h->plan_split = value != 0; return HNSW_OK; }
    if (!std::strcmp(key, "plan_split_x10")) { h->plan_split_x10 = (uint32_t)std::max<int64_t>(value, 0); return HNSW_OK; }
    if (!std::strcmp(key, "commit_par")) { h->commit_par = value < 0 ? 0 : (value > 2 ? 2 : (int)value); return HNSW_OK; }
    if (!std::strcmp(key, "commit_par_min_x10")) { h->commit_par_min_x10 = (uint32_t)std::max<int64_t>(value, 0); return HNSW_OK; }
    if (!std::strcmp(key, "par_max_resident")) { h->par_max_resident = value <= 0 ? 0xFFFFFFFFu : (uint32_t)value; return HNSW_OK; }
    if (!std::strcmp(key, "commit_team")) { h->commit_team = value < 0 ? 0 : (value > 2 ? 2 : (int)value); return HNSW_OK; }
    if (!std::strcmp(key, "plan_duo_max")) { h->plan_duo_max = (uint32_t)std::max<int64_t>(value, 0); return HNSW_OK; }
    if (!std::strcmp(key, "duo_max")) { h->duo_max = (uint32_t)std::max<int64_t>(value, 0); return HNSW_OK; }
    if (!std::strcmp(key, "grid_stride")) { h->grid_stride = value != 0; return HNSW_OK; }
    if (!std::strcmp(key, "query_in_lds")) {     // dims 128 / 768 normally keep the query in registers (T = dim/32)
        if (h->mode == MODE_AVX && !h->fmt) h->T = value ? 0 : ((h->dim == 128 || h->dim == 768) ? (int)(h->dim / 32) : 0);
        return HNSW_OK;
    }
    if (!std::strcmp(key, "visited_bounded")) { h->visited_bounded = value != 0; return HNSW_OK; }
    if (!std::strcmp(key, "waves_per_cu")) { h->wpc_user = true; h->max_waves_per_cu = (uint32_t)std::min<int64_t>(std::max<int64_t>(value, 1), 16); return HNSW_OK; }
    if (!std::strcmp(key, "fast_seed")) { h->fast_seed = (uint32_t)std::max<int64_t>(value, 1); return HNSW_OK; }
    if (!std::strcmp(key, "fast_batch_max")) { h->fast_batch_max = (uint32_t)std::max<int64_t>(value, 1); return HNSW_OK; }
    if (!std::strcmp(key, "fast_batch_div")) { h->fast_batch_div = (uint32_t)std::max<int64_t>(value, 1); return HNSW_OK; }
    return fail(h, HNSW_ERR_INVALID, std::string("unknown tuning key ") + key);
}

hnsw_status hnsw_add(hnsw_index *h, const float *v, uint32_t dim, int32_t level, uint32_t *out_id,
                     uint32_t *touched, uint32_t touched_cap, uint32_t *n_touched)
{
    if (!h || !v) return HNSW_ERR_INVALID;
    if (h->fmt) return fail(h, HNSW_ERR_INVALID, "the index is read-only in compressed (bf16 / fp8) storage mode");
    if (dim != h->dim) {                         // core.rs:389-391
        char buf[96];
        snprintf(buf, sizeof buf, "data dimension: %u does not match Index", dim);
        return fail(h, HNSW_ERR_DIM_MISMATCH, buf);
    }
    if (!all_finite(v, dim)) return fail(h, HNSW_ERR_INVALID, "non-finite vector component");
    ON_DEVICE(h);
    uint32_t nt = 0;
    hnsw_status s = add_exact(h, v, nullptr, level, out_id, touched != nullptr, &nt);
    if (s != HNSW_OK) return s;
    if (touched && nt) {
        // the device list may repeat ids (the reference's `updated` is a HashSet, core.rs:522)
        // The device list is sized for the operation's worst case before the kernel runs; should it overflow all
        // the same, the INSERT IS COMPLETE: reporting an error would make every host mirror skip its name / id
        // bookkeeping for a node the graph already holds.  The call succeeds and *n_touched = UINT32_MAX tells the
        // caller that the list is unavailable (more than any buffer: it must treat every node as touched).
        if (nt > h->touched_cap) {
            h->err = "update_fn list overflow (the insert itself is complete)";
            if (n_touched) *n_touched = 0xFFFFFFFFu;
            return HNSW_OK;
        }
        uint32_t have = std::min(nt, h->touched_cap);
        std::vector<uint32_t> tmp(have);
        if (h->h_ins && have <= h->ins_touched_have) {           // it came back with the header (add_exact)
            const uint32_t *tp = h->h_ins + ins_touched_off(h);
            std::copy(tp, tp + have, tmp.begin());
        } else {
            HIP_TRY(h, hipMemcpyAsync(tmp.data(), h->d_touched, (size_t)have * 4, hipMemcpyDeviceToHost, h->stream));
            HIP_TRY(h, hipStreamSynchronize(h->stream));
        }
        std::sort(tmp.begin(), tmp.end());
        tmp.erase(std::unique(tmp.begin(), tmp.end()), tmp.end());
        nt = (uint32_t)tmp.size();
        for (uint32_t i = 0; i < nt && i < touched_cap; ++i) touched[i] = tmp[i];
    }
    if (n_touched) *n_touched = nt;
    return HNSW_OK;
}

hnsw_status hnsw_add_batch(hnsw_index *h, const float *V, uint32_t n, uint32_t dim, const int32_t *levels,
                           uint32_t mode)
{
    if (!h || (!V && n)) return HNSW_ERR_INVALID;
    if (h->fmt) return fail(h, HNSW_ERR_INVALID, "the index is read-only in compressed (bf16 / fp8) storage mode");
    if (dim != h->dim) {
        char buf[96];
        snprintf(buf, sizeof buf, "data dimension: %u does not match Index", dim);
        return fail(h, HNSW_ERR_DIM_MISMATCH, buf);
    }
    if (n == 0) return HNSW_OK;
    if (mode > 1) return fail(h, HNSW_ERR_INVALID, "mode must be 0 (exact) or 1 (fast)");
    if (!all_finite(V, (size_t)n * dim)) return fail(h, HNSW_ERR_INVALID, "non-finite vector component");
    ON_DEVICE(h);
    hnsw_status s;
    uint32_t done = 0;
    if (h->wide_m) mode = 0;                                 // M > 64: the reference's order through the serial kernels
    // exact inserts: all of them (mode 0) or the seed prefix of the fast build.  Large exact batches go
    // through the optimistic window (same graph, planned in parallel, committed in order: hnsw_occ.hpp).
    // (tie_mode 2: every insert on the std-order kernel; tie_mode 1 on a shape whose plans do not count ties -- anything but
    // dim-128 f32 rows in the AVX order -- likewise: add_exact decides per node)
    const bool tie_gate_shape = h->plan_lean && h->mode == MODE_AVX && h->dim == 128 && h->fmt == FMT_F32;
    if (mode == 0 && h->occ_window >= 2 && n >= h->occ_min_batch && h->tie_mode != 2 && !(h->tie_mode == 1 && !tie_gate_shape)) {
        while (done < n && h->n - h->n_dead < 2) {           // the first nodes: serial (no graph to plan against)
            if ((s = add_exact(h, V + (size_t)done * dim, nullptr, levels ? levels[done] : -1, nullptr, false, nullptr)) != HNSW_OK)
                return s;
            ++done;
        }
        if (done < n && (s = add_exact_window(h, V + (size_t)done * dim, n - done, levels ? levels + done : nullptr)) != HNSW_OK) return s;
        return HNSW_OK;
    }
    while (done < n && (mode == 0 || h->n < h->fast_seed)) {
        if ((s = add_exact(h, V + (size_t)done * dim, nullptr, levels ? levels[done] : -1, nullptr, false, nullptr)) != HNSW_OK)
            return s;
        ++done;
    }
    if (done == n) return HNSW_OK;

    // ---- fast build ---------------------------------------------------------
    h->asymmetric = true;
    const uint32_t first = h->n, rest = n - done;
    std::vector<uint32_t> lv(rest);
    for (uint32_t i = 0; i < rest; ++i) {
        uint32_t l = levels ? (levels[done + i] >= 0 ? (uint32_t)levels[done + i] : draw_level(h)) : draw_level(h);
        lv[i] = std::min(l, kMaxLayers - 1);
    }
    if ((s = ensure_node_cap(h, first + rest)) != HNSW_OK) return s;
    uint32_t up_need = h->upper_used;
    for (uint32_t i = 0; i < rest; ++i) up_need += lv[i];
    if ((s = ensure_upper_cap(h, std::max(up_need, 1u))) != HNSW_OK) return s;
    for (uint32_t i = 0; i < rest; ++i)
        if ((s = reserve_node(h, first + i, lv[i])) != HNSW_OK) return s;
    HIP_TRY(h, hipMemcpyAsync(h->d_vec + (size_t)first * dim, V + (size_t)done * dim, (size_t)rest * dim * 4, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(h, hipMemcpyAsync(h->d_levels + first, h->h_levels.data() + first, (size_t)rest * 4, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(h, hipMemcpyAsync(h->d_upper_base + first, h->h_upper_base.data() + first, (size_t)rest * 4, hipMemcpyHostToDevice, h->stream));
    if ((s = ensure_spill(h)) != HNSW_OK) return s;
    if ((s = ensure_plan(h, h->fast_batch_max)) != HNSW_OK) return s;
    DevScratch<uint32_t> sc0, scU, scN;
    HIP_TRY(h, sc0.alloc(h->cap));
    HIP_TRY(h, scU.alloc(std::max(h->upper_cap, 1u)));
    HIP_TRY(h, scN.alloc(1));
    uint32_t *pending0 = sc0.p, *pendingU = scU.p, *work_n = scN.p;
    HIP_TRY(h, hipMemsetAsync(pending0, 0, (size_t)h->cap * 4, h->stream));
    HIP_TRY(h, hipMemsetAsync(pendingU, 0, (size_t)std::max(h->upper_cap, 1u) * 4, h->stream));
    HIP_TRY(h, hipMemsetAsync(work_n, 0, 4, h->stream));
    const uint32_t want_work = h->fast_batch_max * (h->m + 1) * 2;
    if (want_work > h->work_cap) {
        dev_free(h, h->d_work, (size_t)h->work_cap * 2);
        if ((s = dev_alloc(h, &h->d_work, (size_t)want_work * 2)) != HNSW_OK) return s;
        h->work_cap = want_work;
    }
    uint32_t pos = 0;
    while (pos < rest) {
        uint32_t bs = std::min(h->fast_batch_max, std::max(1u, h->n / h->fast_batch_div));
        bs = std::min(bs, rest - pos);
        uint32_t new_max = h->max_layer;
        int64_t new_ep = h->enterpoint;
        // a node that raises max_layer closes its batch: the next batch must see it as the enterpoint
        for (uint32_t i = 0; i < bs; ++i)
            if (lv[pos + i] > new_max) { new_max = lv[pos + i]; new_ep = first + pos + i; bs = i + 1; break; }
        if ((s = add_fast_batch(h, first + pos, bs, new_max, new_ep, pending0, pendingU, work_n)) != HNSW_OK) break;
        pos += bs;
    }
    hipError_t e = hipStreamSynchronize(h->stream);   // the kernels above use the scratch
    if (s != HNSW_OK) return s;
    HIP_TRY(h, e);
    DevHeader hd;
    if ((s = pull_header(h, &hd)) != HNSW_OK) return s;
    return check_dev_status(h, hd);
}

hnsw_status hnsw_delete(hnsw_index *h, uint32_t id, uint32_t *touched, uint32_t touched_cap, uint32_t *n_touched)
{
    if (!h) return HNSW_ERR_INVALID;
    if (n_touched) *n_touched = 0;
    if (h->fmt) return fail(h, HNSW_ERR_INVALID, "the index is read-only in compressed (bf16 / fp8) storage mode");
    if (id >= h->n || h->h_dead[id]) {                          // core.rs:419-422
        char buf[64];
        snprintf(buf, sizeof buf, "Node: %u does not exist", id);
        return fail(h, HNSW_ERR_NOT_FOUND, buf);
    }
    ON_DEVICE(h);
    uint32_t nt = 0;
    hnsw_status s = delete_exact(h, id, &nt);
    if (s != HNSW_OK) return s;
    if (touched && (nt || !h->purged_owners.empty())) {
        if (nt > h->touched_cap) {                           // as in hnsw_add: the delete is complete, the list is not
            h->err = "update_fn list overflow (the delete itself is complete)";
            if (n_touched) *n_touched = 0xFFFFFFFFu;
            return HNSW_OK;
        }
        uint32_t have = std::min(nt, h->touched_cap);
        std::vector<uint32_t> tmp(have);
        if (have) {
            HIP_TRY(h, hipMemcpyAsync(tmp.data(), h->d_touched, (size_t)have * 4, hipMemcpyDeviceToHost, h->stream));
            HIP_TRY(h, hipStreamSynchronize(h->stream));
        }
        for (uint32_t o : h->purged_owners)
            if (o != id) tmp.push_back(o);                   // rows the inbound sweep edited (one-directional graphs)
        std::sort(tmp.begin(), tmp.end());
        tmp.erase(std::unique(tmp.begin(), tmp.end()), tmp.end());
        nt = (uint32_t)tmp.size();
        for (uint32_t i = 0; i < nt && i < touched_cap; ++i) touched[i] = tmp[i];
    }
    if (n_touched) *n_touched = nt;
    return HNSW_OK;
}

hnsw_status hnsw_search_batch_device(hnsw_index *h, const float *dQ, uint32_t B, uint32_t dim, uint32_t k,
                                     uint32_t *d_ids, float *d_sims, uint32_t *d_n_out, void *stream)
{
    if (!h) return HNSW_ERR_INVALID;
    if (dim != h->dim) {                         // core.rs:478-480
        char buf[96];
        snprintf(buf, sizeof buf, "data dimension: %u does not match Index", dim);
        return fail(h, HNSW_ERR_DIM_MISMATCH, buf);
    }
    if (B == 0) return HNSW_OK;
    if (k == 0) return fail(h, HNSW_ERR_INVALID, "k must be >= 1");
    ON_DEVICE(h);
    hipStream_t st = (hipStream_t)stream;
    hnsw_status s0 = ensure_spill(h);
    if (s0 != HNSW_OK) return s0;
    if (st != h->stream) {
        // the index was written (build / import / scratch fills) on the engine's
        // own stream: order the caller's stream after it
        HIP_TRY(h, hipEventRecord(h->ev_sync, h->stream));
        HIP_TRY(h, hipStreamWaitEvent(st, h->ev_sync, 0));
    }
    if (h->n == h->n_dead || h->enterpoint < 0) { // core.rs:481-483
        HIP_TRY(h, hipMemsetAsync(d_n_out, 0, (size_t)B * 4, st));
        HIP_TRY(h, hipMemsetAsync(d_ids, 0xFF, (size_t)B * k * 4, st));
        hipLaunchKernelGGL(k_fill_f32, dim3(std::min<uint32_t>(1024, (uint32_t)(((size_t)B * k + 255) / 256))), dim3(256), 0, st,
                           d_sims, (size_t)B * k, -__builtin_inff());
        HIP_TRY(h, hipGetLastError());
        return HNSW_OK;
    }
    if (h->pipe_device && B >= h->pipe_min_batch) return search_device_pipelined(h, dQ, B, k, d_ids, d_sims, d_n_out, st);
    return launch_search(h, dQ, B, k, d_ids, d_sims, d_n_out, st);
}

hnsw_status hnsw_search_batch(hnsw_index *h, const float *Q, uint32_t B, uint32_t dim, uint32_t k,
                              uint32_t *ids, float *sims, uint32_t *n_out)
{
    if (!h) return HNSW_ERR_INVALID;
    if (dim != h->dim) {
        char buf[96];
        snprintf(buf, sizeof buf, "data dimension: %u does not match Index", dim);
        return fail(h, HNSW_ERR_DIM_MISMATCH, buf);
    }
    if (B == 0) return HNSW_OK;
    if (k == 0) return fail(h, HNSW_ERR_INVALID, "k must be >= 1");
    ON_DEVICE(h);
    if (h->n == h->n_dead || h->enterpoint < 0) {
        if (!all_finite(Q, (size_t)B * dim)) return fail(h, HNSW_ERR_INVALID, "non-finite query component");
        for (uint32_t b = 0; b < B; ++b) n_out[b] = 0;
        return HNSW_OK;
    }
    if (B > kPinnedBatch) return search_batch_pipelined(h, Q, B, k, ids, sims, n_out);   // checks finiteness while staging
    // HNSW.SEARCH-sized calls: one launch on the engine's stream through one pinned block
    if (!all_finite(Q, (size_t)B * dim)) return fail(h, HNSW_ERR_INVALID, "non-finite query component");
    hnsw_status s = ensure_stage(h, B, k);
    if (s != HNSW_OK) return s;
    const size_t nq = (size_t)B * dim, nk = (size_t)B * k;
    uint32_t *d_ids = h->d_res, *d_nout = h->d_res + 2 * nk;
    float *d_sims = reinterpret_cast<float *>(h->d_res + nk);
    std::memcpy(h->h_pinned, Q, nq * 4);
    HIP_TRY(h, hipMemcpyAsync(h->d_Q, h->h_pinned, nq * 4, hipMemcpyHostToDevice, h->stream));
    if ((s = launch_search(h, h->d_Q, B, k, d_ids, d_sims, d_nout, h->stream)) != HNSW_OK) return s;
    uint32_t *back = h->h_pinned + nq;
    HIP_TRY(h, hipMemcpyAsync(back, h->d_res, (2 * nk + B) * 4, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    std::memcpy(ids, back, nk * 4);
    std::memcpy(sims, back + nk, nk * 4);
    std::memcpy(n_out, back + 2 * nk, (size_t)B * 4);
    for (uint32_t b = 0; b < B; ++b)
        if (n_out[b] == kEmpty) return fail(h, HNSW_ERR_CAPACITY, "visited-set spill table overflow");
    return HNSW_OK;
}

hnsw_status hnsw_search(hnsw_index *h, const float *q, uint32_t dim, uint32_t k, uint32_t *ids, float *sims,
                        uint32_t *n_out)
{
    if (!h || !n_out) return HNSW_ERR_INVALID;
    // the caller's buffers hold k entries (include/hnsw_mi355x.h): the batch form writes straight into them
    if (!ids || !sims) return HNSW_ERR_INVALID;
    return hnsw_search_batch(h, q, 1, dim, k, ids, sims, n_out);
}

#include "hnsw_transfer.inc"   // hnsw_import, replicas, hnsw_get_*, hnsw_export_layer
#include "hnsw_snapshot.inc"   // hnsw_serialize / hnsw_deserialize

hnsw_status hnsw_get_counters(hnsw_index *h, hnsw_counters *search, hnsw_counters *insert)
{
    if (!h) return HNSW_ERR_INVALID;
    ON_DEVICE(h);
    DevHeader hd;
    HIP_TRY(h, hipMemcpy(&hd, h->d_hdr, sizeof hd, hipMemcpyDeviceToHost));
    if (search) { search->n_dist = hd.ctr_search[0]; search->n_ids = hd.ctr_search[1]; search->n_expand = hd.ctr_search[2]; search->n_spill = hd.ctr_search[3]; }
    if (insert) { insert->n_dist = hd.ctr_insert[0]; insert->n_ids = hd.ctr_insert[1]; insert->n_expand = hd.ctr_insert[2]; insert->n_spill = hd.ctr_insert[3]; }
    return HNSW_OK;
}

hnsw_status hnsw_reset_counters(hnsw_index *h)
{
    if (!h) return HNSW_ERR_INVALID;
    ON_DEVICE(h);
    HIP_TRY(h, hipDeviceSynchronize());
    static_assert(offsetof(DevHeader, ctr_tie) == offsetof(DevHeader, ctr_search) + sizeof(unsigned long long) * 16, "counters are contiguous");
    HIP_TRY(h, hipMemset((char *)h->d_hdr + offsetof(DevHeader, ctr_search), 0, sizeof(unsigned long long) * 20));
    h->tie_uncounted = false;
    return HNSW_OK;
}

hnsw_status hnsw_get_tie_counters(hnsw_index *h, uint64_t *out)
{
    if (!h || !out) return HNSW_ERR_INVALID;
    ON_DEVICE(h);
    DevHeader hd;
    HIP_TRY(h, hipMemcpy(&hd, h->d_hdr, sizeof hd, hipMemcpyDeviceToHost));
    for (int i = 0; i < 4; ++i) out[i] = hd.ctr_tie[i];
    if (h->tie_uncounted) out[0] = out[1] = ~0ull;        // a search of this period ran on a kernel without a census form: unknown
    return HNSW_OK;
}

// counters of the last windowed exact build: commits, speculative shrinks applied, shrinks recomputed at commit,
// plans found stale by the parallel validation, journal entries (development aid; not in the public header)
hnsw_status hnsw_debug_occ(hnsw_index *h, uint64_t *out5 /* [16] */)
{
    if (!h || !out5) return HNSW_ERR_INVALID;
    out5[0] = h->occ_last.n_commit; out5[1] = h->occ_last.n_spec; out5[2] = h->occ_last.n_fallback;
    out5[3] = h->occ_last.n_stale; out5[4] = h->occ_last.nJ; out5[5] = h->occ_last.stop;   // [5] = rounds
    for (int i = 0; i < 8; ++i) out5[6 + i] = h->occ_last.prof[i];   // commit kernel phases, shader clocks
    out5[14] = h->occ_last.n_norec; out5[15] = h->occ_last.n_rowstale;
    return HNSW_OK;
}

// the device's control block of the exact-order insert / delete as it stands (development aid; not in the public header):
// [0..7] prof, [8..15] n_cls, [16] n_spec, [17] n_fallback
hnsw_status hnsw_debug_occ_ctl(hnsw_index *h, uint64_t *out18)
{
    if (!h || !out18 || !h->d_occ_ctl) return HNSW_ERR_INVALID;
    ON_DEVICE(h);
    OccCtl c;
    HIP_TRY(h, hipDeviceSynchronize());
    HIP_TRY(h, hipMemcpy(&c, h->d_occ_ctl, sizeof(OccCtl), hipMemcpyDeviceToHost));
    for (int i = 0; i < 8; ++i) { out18[i] = c.prof[i]; out18[8 + i] = c.n_cls[i]; }
    out18[16] = c.n_spec; out18[17] = c.n_fallback;
    return HNSW_OK;
}

// the parallel group commit of the last windowed build (hnsw_occ_par.hpp; development aid, not in the public header):
// [0] groups committed, [1] dry runs made, groups closed by [2] a stale link plan, [3] a record the node used, [4] a row it rewrote
hnsw_status hnsw_debug_occ_par(hnsw_index *h, uint64_t *out5 /* [21] */)
{
    if (!h || !out5) return HNSW_ERR_INVALID;
    out5[0] = h->occ_last.n_groups; out5[1] = h->occ_last.n_dry; out5[2] = h->occ_last.n_conf_link;
    out5[3] = h->occ_last.n_conf_rec; out5[4] = h->occ_last.n_conf_row;
    for (int i = 0; i < 8; ++i) out5[5 + i] = h->occ_last.par_prof[i];   // workgroup 0's phase clocks (100 MHz), iterations, launches
    for (int i = 0; i < 8; ++i) out5[13 + i] = h->occ_last.dry_prof[i];  // all dry runs' phase clocks; [7] sum of the slowest per iteration
    return HNSW_OK;
}

// ... more of the same (development aid): [0] rounds that ended right after a group, without an iteration spent on finding the head not ready
hnsw_status hnsw_debug_occ_par2(hnsw_index *h, uint64_t *out4)
{
    if (!h || !out4) return HNSW_ERR_INVALID;
    out4[0] = h->occ_last.n_early; out4[1] = out4[2] = out4[3] = 0;
    return HNSW_OK;
}

// inserts of windowed builds that tie_mode 1 handed to the std-order kernel since the handle was created (development aid)
hnsw_status hnsw_debug_tie_redone(hnsw_index *h, uint64_t *out)
{
    if (!h || !out) return HNSW_ERR_INVALID;
    *out = h->tie_redone;
    return HNSW_OK;
}

// recomputed shrinks of the last windowed build by cause (development aid; not in the public header)
hnsw_status hnsw_debug_occ_causes(hnsw_index *h, uint64_t *out8)
{
    if (!h || !out8) return HNSW_ERR_INVALID;
    for (int i = 0; i < 8; ++i) out8[i] = h->occ_last.n_cls[i];
    return HNSW_OK;
}

// cycles per search phase (only filled by -DHNSW_PHASE_TIMERS builds; not in the public header)
hnsw_status hnsw_debug_phase_cycles(hnsw_index *h, uint64_t *out8)
{
    if (!h || !out8) return HNSW_ERR_INVALID;
    ON_DEVICE(h);
    DevHeader hd;
    HIP_TRY(h, hipMemcpy(&hd, h->d_hdr, sizeof hd, hipMemcpyDeviceToHost));
    for (int i = 0; i < 8; ++i) out8[i] = hd.prof[i];
    return HNSW_OK;
}

// The search pipeline wants a hardware queue per lane; the HIP runtime reads GPU_MAX_HW_QUEUES once, when it
// starts.  A process that loads this library before touching HIP (a Redis module does) gets 8 without exporting
// anything; a value the operator set is left alone, and ensure_pipe() measures what the lanes really got.
__attribute__((constructor)) static void hnsw_library_loaded() { (void)setenv("GPU_MAX_HW_QUEUES", "8", 0); }

hnsw_status hnsw_pipeline_info(hnsw_index *h, hnsw_pipeline *out)
{
    if (!h || !out) return HNSW_ERR_INVALID;
    ON_DEVICE(h);
    hnsw_status s = ensure_pipe(h);
    out->lanes = hnsw_index::kPipe;
    out->overlap = h->pipe_overlap;
    out->probe_ratio = h->pipe_probe_ratio;
    out->priorities = h->pipe_prio ? 1u : 0u;
    out->chunk = h->pipe_chunk;
    out->min_batch = h->pipe_min_batch;
    const char *e = std::getenv("GPU_MAX_HW_QUEUES");
    out->hw_queues_env = e ? (uint32_t)std::atoi(e) : 0u;
    return s;
}

// 1 when the latest search launch was the specialised dim-128 kernel's, 0 the general one's (not in the public header)
hnsw_status hnsw_debug_last_search_path(hnsw_index *h, uint32_t *lean)
{
    if (!h || !lean) return HNSW_ERR_INVALID;
    *lean = (h->last_search_lean ? 1u : 0u) | (h->last_search_duo ? 2u : 0u);   // bit 1: its two-wave form
    return HNSW_OK;
}

// why the specialised kernel cannot serve this index ("" = it can); development aid, not in the public header
const char *hnsw_debug_lean_blocker(hnsw_index *h)
{
    const char *why = h ? lean_blocker(h) : "null handle";
    return why ? why : "";
}

hnsw_status hnsw_last_search_kernel_ms(hnsw_index *h, float *ms)
{
    if (!h || !ms) return HNSW_ERR_INVALID;
    if (!h->ev_valid) return fail(h, HNSW_ERR_INVALID, "no timed search launch yet (hnsw_set_tuning(\"time_launches\", 1) first)");
    ON_DEVICE(h);
    HIP_TRY(h, hipEventSynchronize(h->ev1));
    HIP_TRY(h, hipEventElapsedTime(ms, h->ev0, h->ev1));
    return HNSW_OK;
}

hnsw_status hnsw_metric_pairs(int device, const float *a, const float *b, uint32_t n, uint32_t dim, float *sims)
{
    if (!a || !b || !sims || dim == 0) return HNSW_ERR_INVALID;
    if (n == 0) return HNSW_OK;
    DeviceScope dev_scope_(device);
    if (dev_scope_.err != hipSuccess) return HNSW_ERR_DEVICE;
    float *da = nullptr, *db = nullptr, *ds = nullptr;
    size_t bytes = (size_t)n * dim * 4;
    auto release = [&] { (void)hipFree(da); (void)hipFree(db); (void)hipFree(ds); };
    if (hipMalloc((void **)&da, bytes) != hipSuccess || hipMalloc((void **)&db, bytes) != hipSuccess ||
        hipMalloc((void **)&ds, (size_t)n * 4) != hipSuccess ||
        hipMemcpy(da, a, bytes, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(db, b, bytes, hipMemcpyHostToDevice) != hipSuccess) {
        release();
        return HNSW_ERR_DEVICE;
    }
    uint32_t grid = std::min(n, 1024u);
    if (dim % 32 == 0 && (dim == 128 || dim == 768)) {
        // the register-resident kernels the search uses for these dims
        uint32_t g2 = std::min((n + 63) / 64, 1024u);
        if (dim == 128) hipLaunchKernelGGL(k_metric_pairs_reg<4>, dim3(g2), dim3(64), 0, 0, da, db, n, ds);
        else hipLaunchKernelGGL(k_metric_pairs_reg<24>, dim3(g2), dim3(64), 0, 0, da, db, n, ds);
    } else {
        size_t lds = (((size_t)dim * 4 + 15) & ~(size_t)15) + 64 * 8;
        if (dim % 32 == 0)
            hipLaunchKernelGGL(k_metric_pairs<MODE_AVX>, dim3(grid), dim3(64), lds, 0, da, db, n, dim, ds);
        else
            hipLaunchKernelGGL(k_metric_pairs<MODE_SCALAR>, dim3(grid), dim3(64), lds, 0, da, db, n, dim, ds);
    }
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e == hipSuccess) e = hipMemcpy(sims, ds, (size_t)n * 4, hipMemcpyDeviceToHost);
    release();
    return e == hipSuccess ? HNSW_OK : HNSW_ERR_DEVICE;
}

} // extern "C"
