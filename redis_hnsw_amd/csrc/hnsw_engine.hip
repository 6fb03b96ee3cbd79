// hnsw_engine.hip -- host side of libhnsw_mi355x.so: the C ABI of
// include/hnsw_mi355x.h over the gfx950 kernels.  No CPU compute path exists
// here: every search / insert is a kernel launch, and creation fails without
// a device.  The heavy kernel templates are instantiated in their own
// translation units (hnsw_tu_*.hip, see hnsw_host.hpp); this file holds the
// handle's bookkeeping, the entry points and the small utility kernels.
#define HNSW_UTILITY_KERNELS 1
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
    const size_t fixed = lds_fixed_bytes(R, T, h->dim, ins);
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
    if (h->duo && h->fmt == FMT_F32 && (uint64_t)B * std::max(h->cur_conc, 1u) <= h->duo_max && idbits - 11 <= 14) {
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

hnsw_status launch_search(hnsw_index *h, const float *dQ, uint32_t B, uint32_t k, uint32_t *d_ids,
                          float *d_sims, uint32_t *d_nout, hipStream_t st)
{
    hnsw_status s = ensure_spill(h);
    if (s != HNSW_OK) return s;
    h->cur_conc = search_concurrency(h, st);
    bool done = false;
    h->last_search_lean = false;
    if ((s = try_launch_lean(h, dQ, B, k, d_ids, d_sims, d_nout, st, &done)) != HNSW_OK) return s;
    if (done) { h->last_search_lean = true; return HNSW_OK; }
    const int R = pick_R(h->efc);
    if (h->fmt == FMT_BF16) s = launch_search_fmt<FMT_BF16>(h, R, dQ, B, k, d_ids, d_sims, d_nout, st);   // any dim % 32 == 0, any M / ef
    else if (h->fmt == FMT_FP8) s = launch_search_fmt<FMT_FP8>(h, R, dQ, B, k, d_ids, d_sims, d_nout, st);
    else if (h->mode == MODE_SCALAR) s = launch_search_r<MODE_SCALAR, 0>(h, R, dQ, B, k, d_ids, d_sims, d_nout, st);
    else if (h->T == 4) s = launch_search_r<MODE_AVX, 4>(h, R, dQ, B, k, d_ids, d_sims, d_nout, st);
    else if (h->T == 24) s = launch_search_r<MODE_AVX, 24>(h, R, dQ, B, k, d_ids, d_sims, d_nout, st);
    else s = launch_search_r<MODE_AVX, 0>(h, R, dQ, B, k, d_ids, d_sims, d_nout, st);
    return s != HNSW_OK ? s : note_search(h, st);
}

// ---- the engine's own search pipeline --------------------------------------------------------------------
// A batch larger than one chunk is split into chunks that run round-robin on kPipe lanes: lane 0 is the stream
// the call is ordered on (the engine's for the host-buffer form, the caller's for the _device form), lanes
// 1.. are engine-owned streams.  Several chunks in flight keep the chip's 2048 wave slots full while the long
// queries of a chunk drain; in the host-buffer form a lane's H2D copy, kernel and D2H copy also
// overlap the other lanes', and the host stages chunk i+1 into pinned memory while chunk i runs.
// Lanes only overlap if each sits on a hardware queue of its own.  The HIP runtime hands new streams distinct
// queues until GPU_MAX_HW_QUEUES (default 4) are in use and multiplexes after that; the library asks for 8 in
// its load-time constructor (effective when it is loaded before the runtime starts), and -- because that cannot
// be relied on -- MEASURES the overlap when the lanes are created: a spin kernel on one lane against the same
// kernel on all lanes at once.  Lanes that serialise are re-created with distinct stream priorities (each
// priority level has its own queue pool) and measured again; what was found is reported by
// hnsw_pipeline_info(), printed once on stderr when the lanes still serialise, and fatal under
// HNSW_REQUIRE_OVERLAP=1.
__global__ void k_spin(unsigned long long ticks)
{
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
}

hnsw_status pipe_probe(hnsw_index *h, float *ratio)
{
    auto lane = [&](uint32_t l) { return l == 0 ? h->stream : h->pipe_st[l]; };
    auto run = [&](uint32_t nl, double *secs) -> hnsw_status {
        for (uint32_t l = 0; l < hnsw_index::kPipe; ++l) HIP_TRY(h, hipStreamSynchronize(lane(l)));
        const auto t0 = std::chrono::steady_clock::now();
        for (uint32_t l = 0; l < nl; ++l) hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, lane(l), 20000ull);   // 200 us at 100 MHz
        HIP_TRY(h, hipGetLastError());
        for (uint32_t l = 0; l < nl; ++l) HIP_TRY(h, hipStreamSynchronize(lane(l)));
        *secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        return HNSW_OK;
    };
    double t1 = 0, tn = 0, scratch = 0;
    hnsw_status s;
    if ((s = run(hnsw_index::kPipe, &scratch)) != HNSW_OK) return s;   // first launch loads the code object
    if ((s = run(1, &t1)) != HNSW_OK || (s = run(hnsw_index::kPipe, &tn)) != HNSW_OK) return s;
    *ratio = (float)(tn / std::max(t1, 1e-9));
    return HNSW_OK;
}

hnsw_status ensure_pipe(hnsw_index *h)
{
    if (h->pipe_overlap >= 0) return HNSW_OK;
    for (uint32_t l = 1; l < hnsw_index::kPipe; ++l) {
        HIP_TRY(h, hipStreamCreateWithFlags(&h->pipe_st[l], hipStreamNonBlocking));
        HIP_TRY(h, hipEventCreateWithFlags(&h->pipe_done[l], hipEventDisableTiming));
    }
    HIP_TRY(h, hipEventCreateWithFlags(&h->pipe_done[0], hipEventDisableTiming));
    HIP_TRY(h, hipEventCreateWithFlags(&h->pipe_fork, hipEventDisableTiming));
    float ratio = 0.f;
    hnsw_status s = pipe_probe(h, &ratio);
    if (s != HNSW_OK) return s;
    if (ratio > 1.6f) {
        // the lanes share a hardware queue: one stream per priority level instead
        int least = 0, greatest = 0;
        HIP_TRY(h, hipDeviceGetStreamPriorityRange(&least, &greatest));
        if (least != greatest) {
            for (uint32_t l = 1; l < hnsw_index::kPipe; ++l) {
                HIP_TRY(h, hipStreamDestroy(h->pipe_st[l]));
                h->pipe_st[l] = nullptr;
                HIP_TRY(h, hipStreamCreateWithPriority(&h->pipe_st[l], hipStreamNonBlocking, l == 1 ? greatest : least));
            }
            h->pipe_prio = true;
            if ((s = pipe_probe(h, &ratio)) != HNSW_OK) return s;
        }
    }
    h->pipe_probe_ratio = ratio;
    h->pipe_overlap = ratio <= 1.6f ? 1 : 0;
    if (!h->pipe_overlap) {
        static bool warned = false;
        if (!warned) {
            warned = true;
            fprintf(stderr, "libhnsw_mi355x: the search pipeline's %u streams do not overlap (probe ratio %.2f): hardware queues are "
                            "shared -- export GPU_MAX_HW_QUEUES=8 before the HIP runtime starts; batched searches run at about "
                            "2/3 of their throughput until then\n", hnsw_index::kPipe, (double)ratio);
        }
        if (std::getenv("HNSW_REQUIRE_OVERLAP"))
            return fail(h, HNSW_ERR_DEVICE, "search pipeline streams share a hardware queue (HNSW_REQUIRE_OVERLAP is set)");
    }
    return HNSW_OK;
}

hnsw_status ensure_pipe_stage(hnsw_index *h, uint32_t chunk, uint32_t k)
{
    const size_t nq = (size_t)chunk * h->dim, nr = 2 * (size_t)chunk * k + chunk;
    if (nq <= h->pipe_q_words && nr <= h->pipe_r_words) return HNSW_OK;
    const size_t wq = std::max(nq, h->pipe_q_words), wr = std::max(nr, h->pipe_r_words);
    for (uint32_t l = 0; l < hnsw_index::kPipe; ++l) {
        hipStream_t st = l == 0 ? h->stream : h->pipe_st[l];
        HIP_TRY(h, hipStreamSynchronize(st));
        dev_free(h, h->pipe_dq[l], h->pipe_q_words);
        dev_free(h, h->pipe_dres[l], h->pipe_r_words);
        if (h->pipe_hq[l]) (void)hipHostFree(h->pipe_hq[l]);
        if (h->pipe_hres[l]) (void)hipHostFree(h->pipe_hres[l]);
        h->pipe_hq[l] = nullptr; h->pipe_hres[l] = nullptr;
    }
    h->pipe_q_words = h->pipe_r_words = 0;
    hnsw_status s;
    for (uint32_t l = 0; l < hnsw_index::kPipe; ++l) {
        if ((s = dev_alloc(h, &h->pipe_dq[l], wq)) != HNSW_OK || (s = dev_alloc(h, &h->pipe_dres[l], wr)) != HNSW_OK) return s;
        HIP_TRY(h, hipHostMalloc((void **)&h->pipe_hq[l], wq * 4, hipHostMallocDefault));
        HIP_TRY(h, hipHostMalloc((void **)&h->pipe_hres[l], wr * 4, hipHostMallocDefault));
    }
    h->pipe_q_words = wq;
    h->pipe_r_words = wr;
    // A stream's first LARGE copy sets up its copy path (measured: 5-6 ms per lane at the first 512 KB H2D copy, in
    // the middle of the first large batch otherwise; a 256-byte copy does not trigger it): one full-size copy in
    // each direction per lane now, once per handle.
    if (!h->pipe_copy_warm) {
        for (uint32_t l = 0; l < hnsw_index::kPipe; ++l) {
            hipStream_t st = l == 0 ? h->stream : h->pipe_st[l];
            HIP_TRY(h, hipMemcpyAsync(h->pipe_dq[l], h->pipe_hq[l], wq * 4, hipMemcpyHostToDevice, st));
            HIP_TRY(h, hipMemcpyAsync(h->pipe_hres[l], h->pipe_dres[l], wr * 4, hipMemcpyDeviceToHost, st));
        }
        for (uint32_t l = 0; l < hnsw_index::kPipe; ++l) HIP_TRY(h, hipStreamSynchronize(l == 0 ? h->stream : h->pipe_st[l]));
        h->pipe_copy_warm = true;
    }
    return HNSW_OK;
}

// chunk sizes: as few chunks of <= pipe_chunk queries as cover the batch, equal to within one query
inline uint32_t pipe_chunks(const hnsw_index *h, uint32_t B, uint32_t *per)
{
    const uint32_t n = (B + h->pipe_chunk - 1) / h->pipe_chunk;
    *per = (B + n - 1) / n;
    return n;
}

// copy while checking: the host entry points refuse non-finite components (see all_finite)
bool copy_finite(float *dst, const float *src, size_t n)
{
    uint32_t bad = 0;
    const uint32_t *s32 = reinterpret_cast<const uint32_t *>(src);
    uint32_t *d32 = reinterpret_cast<uint32_t *>(dst);
    for (size_t i = 0; i < n; ++i) {
        const uint32_t u = s32[i];
        d32[i] = u;
        bad |= ((u & 0x7F800000u) == 0x7F800000u) ? 1u : 0u;
    }
    return bad == 0;
}

// hnsw_search_batch for B > kPinnedBatch: chunks through pinned staging on the lanes, copies and kernels overlapped
hnsw_status search_batch_pipelined(hnsw_index *h, const float *Q, uint32_t B, uint32_t k, uint32_t *ids, float *sims,
                                   uint32_t *n_out)
{
    hnsw_status s;
    if ((s = ensure_pipe(h)) != HNSW_OK) return s;
    uint32_t per = 0;
    const uint32_t nch = pipe_chunks(h, B, &per);
    if ((s = ensure_pipe_stage(h, per, k)) != HNSW_OK) return s;
    const uint32_t lanes = std::min(nch, hnsw_index::kPipe);
    auto lane_st = [&](uint32_t l) { return l == 0 ? h->stream : h->pipe_st[l]; };
    // Whatever a launch may still enqueue on the engine's stream must be there BEFORE the other lanes take their
    // dependency on it: the HBM visited tables are (re)allocated and filled on h->stream when the index has grown
    // (or on the very first search), and a chunk on lane 1 that started under that fill read and lost visited
    // marks -- duplicate ids in its answers (found by scripts/fuzz_search.py; launch_search's own call is a no-op then)
    if ((s = ensure_spill(h)) != HNSW_OK) return s;
    // the graph was written on the engine's stream: the other lanes start after it
    if (lanes > 1) {
        HIP_TRY(h, hipEventRecord(h->ev_sync, h->stream));
        for (uint32_t l = 1; l < lanes; ++l) HIP_TRY(h, hipStreamWaitEvent(h->pipe_st[l], h->ev_sync, 0));
    }
    h->pipe_inflight = lanes;
    struct Restore { hnsw_index *h; ~Restore() { h->pipe_inflight = 1; } } restore{h};
    bool overflow = false;
    const bool trace = std::getenv("HNSW_PIPE_TRACE") != nullptr;
    const auto tr0 = std::chrono::steady_clock::now();
    auto tr = [&](const char *what, uint32_t c) {
        if (trace) fprintf(stderr, "[pipe %8.1f us] %s %u\n", std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - tr0).count(), what, c);
    };
    auto collect = [&](uint32_t c) -> hnsw_status {       // chunk c's results: pinned -> the caller's buffers
        const uint32_t l = c % hnsw_index::kPipe, off = c * per, cb = std::min(per, B - off);
        tr("wait", c);
        HIP_TRY(h, hipEventSynchronize(h->pipe_done[l]));
        tr("done", c);
        const uint32_t *r = h->pipe_hres[l];
        const size_t nk = (size_t)cb * k;
        std::memcpy(ids + (size_t)off * k, r, nk * 4);
        std::memcpy(sims + (size_t)off * k, r + nk, nk * 4);
        std::memcpy(n_out + off, r + 2 * nk, (size_t)cb * 4);
        for (uint32_t b = 0; b < cb; ++b) overflow |= n_out[off + b] == kEmpty;
        return HNSW_OK;
    };
    for (uint32_t c = 0; c < nch; ++c) {
        const uint32_t l = c % hnsw_index::kPipe, off = c * per, cb = std::min(per, B - off);
        if (c >= hnsw_index::kPipe && (s = collect(c - hnsw_index::kPipe)) != HNSW_OK) return s;   // frees lane l's staging
        const size_t nq = (size_t)cb * h->dim, nk = (size_t)cb * k;
        tr("stage", c);
        if (!copy_finite(h->pipe_hq[l], Q + (size_t)off * h->dim, nq)) {
            for (uint32_t d = 0; d < lanes; ++d) (void)hipStreamSynchronize(lane_st(d));
            return fail(h, HNSW_ERR_INVALID, "non-finite query component");
        }
        hipStream_t st = lane_st(l);
        tr("enqueue", c);
        HIP_TRY(h, hipMemcpyAsync(h->pipe_dq[l], h->pipe_hq[l], nq * 4, hipMemcpyHostToDevice, st));
        tr("h2d", c);
        uint32_t *d_ids = h->pipe_dres[l], *d_nout = h->pipe_dres[l] + 2 * nk;
        float *d_sims = reinterpret_cast<float *>(h->pipe_dres[l] + nk);
        if ((s = launch_search(h, h->pipe_dq[l], cb, k, d_ids, d_sims, d_nout, st)) != HNSW_OK) return s;
        tr("launched", c);
        HIP_TRY(h, hipMemcpyAsync(h->pipe_hres[l], h->pipe_dres[l], (2 * nk + cb) * 4, hipMemcpyDeviceToHost, st));
        tr("d2h", c);
        HIP_TRY(h, hipEventRecord(h->pipe_done[l], st));
        tr("enqueued", c);
    }
    for (uint32_t c = nch > hnsw_index::kPipe ? nch - hnsw_index::kPipe : 0; c < nch; ++c)
        if ((s = collect(c)) != HNSW_OK) return s;
    if (overflow) return fail(h, HNSW_ERR_CAPACITY, "visited-set spill table overflow");
    return HNSW_OK;
}

// hnsw_search_batch_device for large batches: chunks on the caller's stream (lane 0) and the engine's lanes,
// joined back into the caller's stream before returning (nothing is synchronised)
hnsw_status search_device_pipelined(hnsw_index *h, const float *dQ, uint32_t B, uint32_t k, uint32_t *d_ids, float *d_sims,
                                    uint32_t *d_nout, hipStream_t st)
{
    hnsw_status s;
    if ((s = ensure_pipe(h)) != HNSW_OK) return s;
    uint32_t per = 0;
    const uint32_t nch = pipe_chunks(h, B, &per);
    const uint32_t lanes = std::min(nch, hnsw_index::kPipe);
    HIP_TRY(h, hipEventRecord(h->pipe_fork, st));        // inputs are ready in `st` order; so is the graph (ev_sync above)
    for (uint32_t l = 1; l < lanes; ++l) HIP_TRY(h, hipStreamWaitEvent(h->pipe_st[l], h->pipe_fork, 0));
    h->pipe_inflight = lanes;
    struct Restore { hnsw_index *h; ~Restore() { h->pipe_inflight = 1; } } restore{h};
    for (uint32_t c = 0; c < nch; ++c) {
        const uint32_t l = c % hnsw_index::kPipe, off = c * per, cb = std::min(per, B - off);
        if ((s = launch_search(h, dQ + (size_t)off * h->dim, cb, k, d_ids + (size_t)off * k, d_sims + (size_t)off * k,
                               d_nout + off, l == 0 ? st : h->pipe_st[l])) != HNSW_OK)
            return s;
    }
    for (uint32_t l = 1; l < lanes; ++l) {
        HIP_TRY(h, hipEventRecord(h->pipe_done[l], h->pipe_st[l]));
        HIP_TRY(h, hipStreamWaitEvent(st, h->pipe_done[l], 0));
    }
    return HNSW_OK;
}

// Staging of the host-buffer entry points: queries in, one result block [ids B*k][sims B*k][n_out B] out
// (a single copy back), and a pinned host mirror for small batches -- a HNSW.SEARCH command is one query,
// where the pageable-memory staging the runtime would do costs more than the transfers themselves.
hnsw_status ensure_stage(hnsw_index *h, uint32_t B, uint32_t k)
{
    size_t nq = (size_t)B * h->dim, nr = 2 * (size_t)B * k + B;
    hnsw_status s;
    if (nq > h->stage_q) {
        dev_free(h, h->d_Q, h->stage_q);
        if ((s = dev_alloc(h, &h->d_Q, nq)) != HNSW_OK) return s;
        h->stage_q = nq;
    }
    if (nr > h->stage_r) {
        dev_free(h, h->d_res, h->stage_r);
        if ((s = dev_alloc(h, &h->d_res, nr)) != HNSW_OK) return s;
        h->stage_r = nr;
    }
    const size_t pin_words = nq + nr;
    if (pin_words > h->pinned_words) {
        if (h->h_pinned) (void)hipHostFree(h->h_pinned);
        h->h_pinned = nullptr;
        h->pinned_words = 0;
        HIP_TRY(h, hipHostMalloc((void **)&h->h_pinned, pin_words * 4, hipHostMallocDefault));
        h->pinned_words = pin_words;
    }
    return HNSW_OK;
}

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
    char buf[160];
    snprintf(buf, sizeof buf, "device status 0x%x:%s%s%s%s", hd.status,
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
    if (dim == 0 || m < 2 || m > 64 || ef_construction == 0 || ef_construction > 4096)
        return fail(h, HNSW_ERR_INVALID, "supported: dim >= 1, 2 <= M <= 64, 1 <= EFCON <= 4096");
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
    if (!std::strcmp(key, "plan_lean")) { h->plan_lean = value != 0; return HNSW_OK; }
    if (!std::strcmp(key, "single_window")) { h->single_window = value != 0; return HNSW_OK; }
    if (!std::strcmp(key, "occ_window")) { h->occ_window = (uint32_t)std::min<int64_t>(std::max<int64_t>(value, 0), kOccMaxW); return HNSW_OK; }
    if (!std::strcmp(key, "occ_ahead_x10")) { h->occ_ahead_x10 = (uint32_t)std::max<int64_t>(value, 5); return HNSW_OK; }
    if (!std::strcmp(key, "occ_slack_extra")) { h->occ_slack_extra = (uint32_t)std::min<int64_t>(std::max<int64_t>(value, 0), 64); return HNSW_OK; }
    if (!std::strcmp(key, "occ_log_cap")) { h->occ_log_cap = (uint32_t)std::min<int64_t>(std::max<int64_t>(value, 1), kOccMaxReads); return HNSW_OK; }
    if (!std::strcmp(key, "occ_min_batch")) { h->occ_min_batch = (uint32_t)std::max<int64_t>(value, 1); return HNSW_OK; }
    if (!std::strcmp(key, "lean")) { h->lean = value != 0; return HNSW_OK; }
    if (!std::strcmp(key, "duo")) { h->duo = value != 0; return HNSW_OK; }
    if (!std::strcmp(key, "plan_duo")) { h->plan_duo = value != 0; return HNSW_OK; }
    if (!std::strcmp(key, "commit_team")) { h->commit_team = value != 0; return HNSW_OK; }
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
    // exact inserts: all of them (mode 0) or the seed prefix of the fast build.  Large exact batches go
    // through the optimistic window (same graph, planned in parallel, committed in order: hnsw_occ.hpp).
    if (mode == 0 && h->occ_window >= 2 && n >= h->occ_min_batch) {
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

hnsw_status hnsw_import(hnsw_index *h, uint32_t n, const float *vectors, const uint32_t *levels,
                        int64_t enterpoint, uint32_t n_layers, const uint64_t *const *row_ptr,
                        const uint32_t *const *col)
{
    if (!h) return HNSW_ERR_INVALID;
    if (h->fmt) return fail(h, HNSW_ERR_INVALID, "the index is read-only in compressed (bf16 / fp8) storage mode");
    if (h->n != 0) return fail(h, HNSW_ERR_INVALID, "hnsw_import needs an empty index");
    if (n == 0) return HNSW_OK;
    if (enterpoint < 0 || enterpoint >= (int64_t)n || n_layers == 0 || n_layers > kMaxLayers)
        return fail(h, HNSW_ERR_INVALID, "bad enterpoint / layer count");
    ON_DEVICE(h);
    // Validate everything before any state changes: the blob may come from a file.
    if (!vectors || !levels || !row_ptr || !col) return fail(h, HNSW_ERR_INVALID, "null argument");
    for (uint32_t i = 0; i < n; ++i)
        if (levels[i] >= kMaxLayers) return fail(h, HNSW_ERR_INVALID, "level too large");
    if (levels[enterpoint] + 1 > n_layers) return fail(h, HNSW_ERR_INVALID, "the enterpoint's level exceeds the layer count");
    for (uint32_t i = 0; i < n; ++i)
        if (levels[i] > levels[enterpoint]) return fail(h, HNSW_ERR_INVALID, "a node above the enterpoint's level (core.rs:587-593 keeps the enterpoint on top)");
    if (!all_finite(vectors, (size_t)n * h->dim)) return fail(h, HNSW_ERR_INVALID, "non-finite vector component");
    // degrees decide the row strides
    uint32_t md0 = 0, mdU = 0;
    for (uint32_t l = 0; l < n_layers; ++l) {
        if (!row_ptr[l] || (!col[l] && row_ptr[l][n] != 0)) return fail(h, HNSW_ERR_INVALID, "null layer arrays");
        if (row_ptr[l][0] != 0) return fail(h, HNSW_ERR_INVALID, "row_ptr must start at 0");
        for (uint32_t i = 0; i < n; ++i) {
            if (row_ptr[l][i + 1] < row_ptr[l][i]) return fail(h, HNSW_ERR_INVALID, "row_ptr is not monotonic");
            uint64_t d = row_ptr[l][i + 1] - row_ptr[l][i];
            if (d > 0 && levels[i] < l) return fail(h, HNSW_ERR_INVALID, "node has links above its level");
            if (d > kAuxWords - 2) return fail(h, HNSW_ERR_INVALID, "degree > 1022 is not supported");
            for (uint64_t e = row_ptr[l][i]; e < row_ptr[l][i + 1]; ++e) {
                if (col[l][e] >= n || col[l][e] == i) return fail(h, HNSW_ERR_INVALID, "neighbour id out of range (or a self link)");
                if (levels[col[l][e]] < l) return fail(h, HNSW_ERR_INVALID, "link to a node that does not reach this layer");
            }
            if (l == 0) md0 = std::max<uint32_t>(md0, (uint32_t)d);
            else mdU = std::max<uint32_t>(mdU, (uint32_t)d);
        }
    }
    uint32_t ns0 = default_stride(h->m_max0, h->m, md0), nsU = default_stride(h->m_max, h->m, mdU);
    hnsw_status s;
    // (re)allocate at the right strides before any data lands
    if (ns0 > h->stride0 || nsU > h->strideU) {
        if ((s = restride(h, std::max(ns0, h->stride0), std::max(nsU, h->strideU))) != HNSW_OK) return s;
    }
    if ((s = ensure_node_cap(h, n)) != HNSW_OK) return s;
    h->h_levels.assign(levels, levels + n);
    h->h_upper_base.assign(n, kNoUpper);
    h->h_dead.assign(n, 0);
    h->n_dead = 0;
    uint32_t used = 0;
    for (uint32_t i = 0; i < n; ++i)
        if (levels[i] > 0) { h->h_upper_base[i] = used; used += levels[i]; }
    if ((s = ensure_upper_cap(h, std::max(used, 1u))) != HNSW_OK) return s;
    h->upper_used = used;
    HIP_TRY(h, hipMemcpyAsync(h->d_vec, vectors, (size_t)n * h->dim * 4, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(h, hipMemcpyAsync(h->d_levels, levels, (size_t)n * 4, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(h, hipMemcpyAsync(h->d_upper_base, h->h_upper_base.data(), (size_t)n * 4, hipMemcpyHostToDevice, h->stream));
    for (uint32_t l = 0; l < n_layers; ++l) {
        uint64_t nnz = row_ptr[l][n];
        DevScratch<uint64_t> s_rp;
        DevScratch<uint32_t> s_col;
        HIP_TRY(h, s_rp.alloc((size_t)n + 1));
        HIP_TRY(h, s_col.alloc((size_t)nnz));
        uint64_t *d_rp = s_rp.p;
        uint32_t *d_col = s_col.p;
        HIP_TRY(h, hipMemcpyAsync(d_rp, row_ptr[l], (size_t)(n + 1) * 8, hipMemcpyHostToDevice, h->stream));
        if (nnz) HIP_TRY(h, hipMemcpyAsync(d_col, col[l], (size_t)nnz * 4, hipMemcpyHostToDevice, h->stream));
        uint32_t blocks = (uint32_t)(((uint64_t)n * 64 + 255) / 256);
        if (l == 0)
            hipLaunchKernelGGL(k_import_rows, dim3(blocks), dim3(256), 0, h->stream, h->d_adj0, h->stride0,
                               (const uint32_t *)nullptr, 0u, h->d_levels, 0u, d_rp, d_col, n);
        else
            hipLaunchKernelGGL(k_import_rows, dim3(blocks), dim3(256), 0, h->stream, h->d_adjU, h->strideU,
                               h->d_upper_base, l - 1, h->d_levels, l, d_rp, d_col, n);
        HIP_TRY(h, hipStreamSynchronize(h->stream));
    }
    h->n = n;
    h->enterpoint = enterpoint;
    h->max_layer = levels[enterpoint];           // the enterpoint is the top node (core.rs:587-593)
    h->max_deg0 = md0;
    h->max_degU = mdU;
    // Graphs the reference builds are symmetric (core.rs:770-772, 793-795); a fast-built one need not be.
    // The exact insert / delete kernels must know (a missing back link is an error only on symmetric graphs).
    {
        DevScratch<unsigned long long> bad;
        HIP_TRY(h, bad.alloc(1));
        HIP_TRY(h, hipMemsetAsync(bad.p, 0, 8, h->stream));
        const uint32_t blocks = (uint32_t)(((uint64_t)n * 64 + 255) / 256);
        for (uint32_t l = 0; l < n_layers; ++l) {
            if (l == 0)
                hipLaunchKernelGGL(k_count_asymmetric, dim3(blocks), dim3(256), 0, h->stream, h->d_adj0, h->stride0,
                                   (const uint32_t *)nullptr, 0u, h->d_levels, 0u, n, bad.p);
            else
                hipLaunchKernelGGL(k_count_asymmetric, dim3(blocks), dim3(256), 0, h->stream, h->d_adjU, h->strideU,
                                   h->d_upper_base, l - 1, h->d_levels, l, n, bad.p);
        }
        HIP_TRY(h, hipGetLastError());
        unsigned long long nbad = 0;
        HIP_TRY(h, hipMemcpyAsync(&nbad, bad.p, 8, hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(h, hipStreamSynchronize(h->stream));
        h->asymmetric = nbad != 0;
    }
    return push_header(h);
}

// ---- replicas: the index tables as device pointers (one-time distribution over RCCL / peer copies) ----------
hnsw_status hnsw_replica_view(hnsw_index *h, hnsw_replica *out)
{
    if (!h || !out) return HNSW_ERR_INVALID;
    ON_DEVICE(h);
    HIP_TRY(h, hipStreamSynchronize(h->stream));          // everything the engine enqueued has landed
    std::memset(out, 0, sizeof *out);
    out->n = h->n; out->dim = h->dim; out->upper_used = h->upper_used;
    out->stride0 = h->stride0; out->stride_upper = h->strideU;
    out->max_layer = h->max_layer; out->max_degree0 = h->max_deg0; out->max_degree_upper = h->max_degU;
    out->n_dead = h->n_dead; out->asymmetric = h->asymmetric ? 1u : 0u; out->format = (uint32_t)h->fmt;
    out->enterpoint = h->enterpoint;
    out->vec_bytes = (uint64_t)h->n * h->dim * (h->fmt == FMT_F32 ? 4 : (h->fmt == FMT_BF16 ? 2 : 1));
    out->adj0_bytes = (uint64_t)h->n * h->stride0 * 4;
    out->adj_upper_bytes = (uint64_t)h->upper_used * h->strideU * 4;
    out->vec = h->d_vec; out->adj0 = h->d_adj0; out->adj_upper = h->d_adjU;
    out->upper_base = h->d_upper_base; out->levels = h->d_levels;
    return HNSW_OK;
}

hnsw_status hnsw_replica_prepare(hnsw_index *h, hnsw_replica *r)
{
    if (!h || !r) return HNSW_ERR_INVALID;
    if (h->n != 0 || h->fmt) return fail(h, HNSW_ERR_INVALID, "hnsw_replica_prepare needs an empty index");
    if (r->dim != h->dim) return fail(h, HNSW_ERR_DIM_MISMATCH, "replica: data dimension does not match Index");
    if (r->n == 0 || r->enterpoint >= (int64_t)r->n || r->max_layer >= kMaxLayers || r->n_dead > r->n ||
        r->stride0 < h->stride0 || r->stride_upper < h->strideU || r->stride0 > kAuxWords || r->stride_upper > kAuxWords ||
        (r->stride0 & 15) || (r->stride_upper & 15) || r->max_degree0 >= r->stride0 || r->max_degree_upper >= r->stride_upper)
        return fail(h, HNSW_ERR_INVALID, "replica: inconsistent header (the source must have the same M)");
    ON_DEVICE(h);
    hnsw_status s;
    if ((s = restride(h, r->stride0, r->stride_upper)) != HNSW_OK) return s;      // the source's row layout
    if ((s = ensure_node_cap(h, r->n)) != HNSW_OK) return s;
    if ((s = ensure_upper_cap(h, std::max(r->upper_used, 1u))) != HNSW_OK) return s;
    if (r->format > (uint32_t)FMT_FP8 || (r->format && h->mode != MODE_AVX))
        return fail(h, HNSW_ERR_INVALID, "replica: unknown storage format (or a compressed source with dim % 32 != 0)");
    const uint32_t esz = r->format == FMT_F32 ? 4 : (r->format == FMT_BF16 ? 2 : 1);
    if (r->format) {
        // the vector matrix of a compressed replica is smaller: swap the f32 allocation for one of the right size
        void *dnew = nullptr;
        const size_t nel = (size_t)h->cap * h->dim;
        HIP_TRY(h, hipMalloc(&dnew, std::max<size_t>(nel, 1) * esz));
        HIP_TRY(h, hipDeviceSynchronize());
        (void)hipFree(h->d_vec);
        h->hbm_bytes -= std::min<uint64_t>(h->hbm_bytes, nel * (4 - esz));
        h->d_vec = reinterpret_cast<float *>(dnew);
    }
    HIP_TRY(h, hipStreamSynchronize(h->stream));          // the allocations' fills are done before anyone writes
    r->vec_bytes = (uint64_t)r->n * h->dim * esz;
    r->adj0_bytes = (uint64_t)r->n * h->stride0 * 4;
    r->adj_upper_bytes = (uint64_t)r->upper_used * h->strideU * 4;
    r->vec = h->d_vec; r->adj0 = h->d_adj0; r->adj_upper = h->d_adjU;
    r->upper_base = h->d_upper_base; r->levels = h->d_levels;
    h->fmt = (int)r->format;                              // the tables are being filled: not searchable until commit (n == 0)
    h->bf16 = h->fmt == FMT_BF16;
    if (h->fmt) h->T = 0;
    return HNSW_OK;
}

hnsw_status hnsw_replica_commit(hnsw_index *h, const hnsw_replica *r, const uint8_t *dead)
{
    if (!h || !r) return HNSW_ERR_INVALID;
    if (h->n != 0) return fail(h, HNSW_ERR_INVALID, "hnsw_replica_commit: the index is not empty");
    if (r->vec != h->d_vec || r->adj0 != h->d_adj0 || r->levels != h->d_levels || r->stride0 != h->stride0 ||
        r->stride_upper != h->strideU || r->n == 0 || r->n > h->cap || r->upper_used > h->upper_cap)
        return fail(h, HNSW_ERR_INVALID, "hnsw_replica_commit: not the block hnsw_replica_prepare returned");
    if (r->n_dead && !dead) return fail(h, HNSW_ERR_INVALID, "hnsw_replica_commit: tombstones missing");
    ON_DEVICE(h);
    HIP_TRY(h, hipDeviceSynchronize());                   // whoever filled the tables (a collective's stream) is done
    // the host mirrors the engine keeps: levels, upper slots, tombstones
    h->h_levels.assign(h->cap, 0);
    h->h_upper_base.assign(h->cap, kNoUpper);
    h->h_dead.assign(h->cap, 0);
    HIP_TRY(h, hipMemcpy(h->h_levels.data(), h->d_levels, (size_t)r->n * 4, hipMemcpyDeviceToHost));
    HIP_TRY(h, hipMemcpy(h->h_upper_base.data(), h->d_upper_base, (size_t)r->n * 4, hipMemcpyDeviceToHost));
    uint32_t used = 0, nd = 0;
    for (uint32_t i = 0; i < r->n; ++i) {
        if (h->h_levels[i] >= kMaxLayers) return fail(h, HNSW_ERR_INVALID, "replica: level too large");
        if (h->h_levels[i] > 0) {
            if (h->h_upper_base[i] != used) return fail(h, HNSW_ERR_INVALID, "replica: upper slots are not in id order");
            used += h->h_levels[i];
        }
        if (dead) { h->h_dead[i] = dead[i] ? 1 : 0; nd += h->h_dead[i]; }
    }
    if (used != r->upper_used || nd != r->n_dead) return fail(h, HNSW_ERR_INVALID, "replica: slot / tombstone counts do not match the header");
    if (r->enterpoint >= 0 && (h->h_dead[r->enterpoint] || h->h_levels[r->enterpoint] != r->max_layer))
        return fail(h, HNSW_ERR_INVALID, "replica: the enterpoint must be a live node of the top layer");
    h->n = r->n; h->n_dead = r->n_dead; h->upper_used = r->upper_used;
    h->enterpoint = r->enterpoint; h->max_layer = r->max_layer;
    h->max_deg0 = r->max_degree0; h->max_degU = r->max_degree_upper;
    h->asymmetric = r->asymmetric != 0;
    return push_header(h);
}

hnsw_status hnsw_get_tombstones(hnsw_index *h, uint8_t *dead)
{
    if (!h || !dead) return HNSW_ERR_INVALID;
    std::copy(h->h_dead.begin(), h->h_dead.begin() + h->n, dead);
    return HNSW_OK;
}

hnsw_status hnsw_get_info(hnsw_index *h, hnsw_info *info)
{
    if (!h || !info) return HNSW_ERR_INVALID;
    info->dim = h->dim; info->m = h->m; info->m_max = h->m_max; info->m_max0 = h->m_max0;
    info->ef_construction = h->efc;
    info->node_count = h->n - h->n_dead;        // the reference's node_count: live nodes
    info->allocated_ids = h->n;
    info->max_layer = h->max_layer; info->enterpoint = h->enterpoint;
    info->stride0 = h->stride0; info->stride_upper = h->strideU;
    info->max_degree0 = h->max_deg0; info->max_degree_upper = h->max_degU;
    info->hbm_bytes = h->hbm_bytes;
    return HNSW_OK;
}

hnsw_status hnsw_get_levels(hnsw_index *h, uint32_t *levels)
{
    if (!h || !levels) return HNSW_ERR_INVALID;
    std::copy(h->h_levels.begin(), h->h_levels.begin() + h->n, levels);
    return HNSW_OK;
}

hnsw_status hnsw_get_level(hnsw_index *h, uint32_t id, uint32_t *level)
{
    if (!h || !level) return HNSW_ERR_INVALID;
    if (id >= h->n) return fail(h, HNSW_ERR_NOT_FOUND, "node id out of range");
    *level = h->h_levels[id];
    return HNSW_OK;
}

hnsw_status hnsw_get_vector(hnsw_index *h, uint32_t id, float *out)
{
    if (!h || !out) return HNSW_ERR_INVALID;
    if (id >= h->n) return fail(h, HNSW_ERR_NOT_FOUND, "node id out of range");
    ON_DEVICE(h);
    if (h->fmt) {                                    // the stored (rounded) values, widened by the search kernels' own code
        DevScratch<float> row;
        HIP_TRY(h, row.alloc(h->dim));
        const uint32_t blocks = (h->dim / 4 + 63) / 64;
        if (h->fmt == FMT_BF16)
            hipLaunchKernelGGL(k_decode_row<FMT_BF16>, dim3(blocks), dim3(64), 0, h->stream, reinterpret_cast<const float4 *>(h->d_vec), (size_t)id, h->dim, row.p);
        else
            hipLaunchKernelGGL(k_decode_row<FMT_FP8>, dim3(blocks), dim3(64), 0, h->stream, reinterpret_cast<const float4 *>(h->d_vec), (size_t)id, h->dim, row.p);
        HIP_TRY(h, hipGetLastError());
        HIP_TRY(h, hipMemcpyAsync(out, row.p, (size_t)h->dim * 4, hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(h, hipStreamSynchronize(h->stream));
        return HNSW_OK;
    }
    HIP_TRY(h, hipMemcpyAsync(out, h->d_vec + (size_t)id * h->dim, (size_t)h->dim * 4, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return HNSW_OK;
}

hnsw_status hnsw_get_neighbors(hnsw_index *h, uint32_t id, uint32_t layer, uint32_t *out, uint32_t cap,
                               uint32_t *n)
{
    if (!h || !n) return HNSW_ERR_INVALID;
    if (id >= h->n) return fail(h, HNSW_ERR_NOT_FOUND, "node id out of range");
    *n = 0;
    if (layer > h->h_levels[id]) return HNSW_OK; // push_levels: rows above the level are empty
    ON_DEVICE(h);
    const uint32_t stride = layer ? h->strideU : h->stride0;
    const uint32_t *row = layer ? h->d_adjU + (size_t)(h->h_upper_base[id] + layer - 1) * stride
                                : h->d_adj0 + (size_t)id * stride;
    std::vector<uint32_t> tmp(stride);
    HIP_TRY(h, hipMemcpyAsync(tmp.data(), row, (size_t)stride * 4, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    uint32_t cnt = std::min(tmp[0], stride - 1);
    *n = cnt;
    for (uint32_t i = 0; i < cnt && i < cap; ++i) out[i] = tmp[1 + i];
    return HNSW_OK;
}

static hnsw_status layer_degrees(hnsw_index *h, uint32_t layer, std::vector<uint32_t> &deg)
{
    deg.assign(h->n, 0);
    if (h->n == 0) return HNSW_OK;
    DevScratch<uint32_t> s_deg;
    HIP_TRY(h, s_deg.alloc(h->n));
    uint32_t *d_deg = s_deg.p;
    uint32_t blocks = (h->n + 255) / 256;
    if (layer == 0)
        hipLaunchKernelGGL(k_degrees, dim3(blocks), dim3(256), 0, h->stream, h->d_adj0, h->stride0,
                           (const uint32_t *)nullptr, 0u, h->d_levels, 0u, h->n, d_deg);
    else
        hipLaunchKernelGGL(k_degrees, dim3(blocks), dim3(256), 0, h->stream, h->d_adjU, h->strideU,
                           h->d_upper_base, layer - 1, h->d_levels, layer, h->n, d_deg);
    HIP_TRY(h, hipMemcpyAsync(deg.data(), d_deg, (size_t)h->n * 4, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return HNSW_OK;
}

hnsw_status hnsw_layer_nnz(hnsw_index *h, uint32_t layer, uint64_t *nnz)
{
    if (!h || !nnz) return HNSW_ERR_INVALID;
    ON_DEVICE(h);
    std::vector<uint32_t> deg;
    hnsw_status s = layer_degrees(h, layer, deg);
    if (s != HNSW_OK) return s;
    uint64_t t = 0;
    for (uint32_t d : deg) t += d;
    *nnz = t;
    return HNSW_OK;
}

hnsw_status hnsw_export_layer(hnsw_index *h, uint32_t layer, uint64_t *row_ptr, uint32_t *col)
{
    if (!h || !row_ptr) return HNSW_ERR_INVALID;
    ON_DEVICE(h);
    std::vector<uint32_t> deg;
    hnsw_status s = layer_degrees(h, layer, deg);
    if (s != HNSW_OK) return s;
    uint64_t t = 0;
    for (uint32_t i = 0; i < h->n; ++i) { row_ptr[i] = t; t += deg[i]; }
    row_ptr[h->n] = t;
    if (t == 0 || h->n == 0) return HNSW_OK;
    DevScratch<uint64_t> s_rp;
    DevScratch<uint32_t> s_col;
    HIP_TRY(h, s_rp.alloc((size_t)h->n + 1));
    HIP_TRY(h, s_col.alloc((size_t)t));
    uint64_t *d_rp = s_rp.p;
    uint32_t *d_col = s_col.p;
    HIP_TRY(h, hipMemcpyAsync(d_rp, row_ptr, (size_t)(h->n + 1) * 8, hipMemcpyHostToDevice, h->stream));
    uint32_t blocks = (uint32_t)(((uint64_t)h->n * 64 + 255) / 256);
    if (layer == 0)
        hipLaunchKernelGGL(k_export_rows, dim3(blocks), dim3(256), 0, h->stream, h->d_adj0, h->stride0,
                           (const uint32_t *)nullptr, 0u, h->d_levels, 0u, h->n, d_rp, d_col);
    else
        hipLaunchKernelGGL(k_export_rows, dim3(blocks), dim3(256), 0, h->stream, h->d_adjU, h->strideU,
                           h->d_upper_base, layer - 1, h->d_levels, layer, h->n, d_rp, d_col);
    HIP_TRY(h, hipMemcpyAsync(col, d_col, (size_t)t * 4, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return HNSW_OK;
}

// ---- snapshot (what the RDB save/load callbacks of src/types.rs:176-284 would stream) -----------
namespace {
constexpr char kSnapMagic[8] = {'H', 'N', 'S', 'W', 'M', 'I', '3', '5'};
struct SnapHeader {
    char magic[8];
    uint32_t version, dim, m, efc, n, n_dead, max_layer, n_layers;
    int64_t enterpoint;
    uint64_t rng[4];
};
inline uint64_t pad8(uint64_t b) { return (b + 7) & ~7ull; }   // every section starts 8-byte aligned
}

hnsw_status hnsw_serialize_size(hnsw_index *h, uint64_t *bytes)
{
    if (!h || !bytes) return HNSW_ERR_INVALID;
    uint64_t total = sizeof(SnapHeader) + pad8((uint64_t)h->n * 4) + pad8(h->n) + pad8((uint64_t)h->n * h->dim * 4);
    const uint32_t n_layers = h->n ? h->max_layer + 1 : 0;
    for (uint32_t l = 0; l < n_layers; ++l) {
        uint64_t nnz = 0;
        hnsw_status s = hnsw_layer_nnz(h, l, &nnz);
        if (s != HNSW_OK) return s;
        total += 8 + ((uint64_t)h->n + 1) * 8 + pad8(nnz * 4);
    }
    *bytes = total;
    return HNSW_OK;
}

hnsw_status hnsw_serialize(hnsw_index *h, void *buf, uint64_t cap, uint64_t *written)
{
    if (!h || !buf || !written) return HNSW_ERR_INVALID;
    if (h->fmt) return fail(h, HNSW_ERR_INVALID, "snapshots are taken from the f32 index (compressed storage is a derived serving copy)");
    uint64_t need = 0;
    hnsw_status s = hnsw_serialize_size(h, &need);
    if (s != HNSW_OK) return s;
    if (cap < need) return fail(h, HNSW_ERR_INVALID, "snapshot buffer too small");
    ON_DEVICE(h);
    unsigned char *p = static_cast<unsigned char *>(buf);
    SnapHeader hd;
    std::memset(&hd, 0, sizeof hd);
    std::memcpy(hd.magic, kSnapMagic, 8);
    hd.version = 1; hd.dim = h->dim; hd.m = h->m; hd.efc = h->efc; hd.n = h->n; hd.n_dead = h->n_dead;
    hd.max_layer = h->max_layer; hd.n_layers = h->n ? h->max_layer + 1 : 0; hd.enterpoint = h->enterpoint;
    std::memcpy(hd.rng, h->rng, sizeof hd.rng);
    std::memcpy(p, &hd, sizeof hd); p += sizeof hd;
    std::memset(p, 0, need - sizeof hd);
    if (h->n) std::memcpy(p, h->h_levels.data(), (size_t)h->n * 4);
    p += pad8((uint64_t)h->n * 4);
    if (h->n) std::memcpy(p, h->h_dead.data(), h->n);
    p += pad8(h->n);
    if (h->n) {
        HIP_TRY(h, hipMemcpyAsync(p, h->d_vec, (size_t)h->n * h->dim * 4, hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(h, hipStreamSynchronize(h->stream));
    }
    p += pad8((uint64_t)h->n * h->dim * 4);
    for (uint32_t l = 0; l < hd.n_layers; ++l) {
        uint64_t nnz = 0;
        if ((s = hnsw_layer_nnz(h, l, &nnz)) != HNSW_OK) return s;
        std::memcpy(p, &nnz, 8); p += 8;
        uint64_t *rp = reinterpret_cast<uint64_t *>(p); p += ((size_t)h->n + 1) * 8;
        uint32_t *cl = reinterpret_cast<uint32_t *>(p); p += pad8(nnz * 4);
        if ((s = hnsw_export_layer(h, l, rp, cl)) != HNSW_OK) return s;
    }
    *written = (uint64_t)(p - static_cast<unsigned char *>(buf));
    return HNSW_OK;
}

hnsw_status hnsw_deserialize(const void *buf, uint64_t bytes, uint64_t seed, int device, hnsw_index **out)
{
    if (!buf || !out || bytes < sizeof(SnapHeader)) return HNSW_ERR_INVALID;
    const unsigned char *p = static_cast<const unsigned char *>(buf), *end = p + bytes;
    SnapHeader hd;
    std::memcpy(&hd, p, sizeof hd); p += sizeof hd;
    if (std::memcmp(hd.magic, kSnapMagic, 8) != 0 || hd.version != 1) { *out = nullptr; return HNSW_ERR_INVALID; }
    hnsw_status s = hnsw_create(hd.dim, hd.m, hd.efc, seed, device, out);
    if (s != HNSW_OK) return s;
    hnsw_index *h = *out;
    std::memcpy(h->rng, hd.rng, sizeof hd.rng);
    if (hd.n == 0) return HNSW_OK;
    // the blob is untrusted: every size is checked against what is left before it is used
    const uint64_t left0 = (uint64_t)(end - p);
    const uint64_t need0 = pad8((uint64_t)hd.n * 4) + pad8(hd.n) + pad8((uint64_t)hd.n * hd.dim * 4);
    if (left0 < need0) return fail(h, HNSW_ERR_INVALID, "truncated snapshot");
    if (hd.n_layers == 0 || hd.n_layers > kMaxLayers || hd.max_layer >= hd.n_layers || hd.n_dead > hd.n ||
        hd.enterpoint >= (int64_t)hd.n || hd.enterpoint < -1)
        return fail(h, HNSW_ERR_INVALID, "inconsistent snapshot header");
    const uint32_t *levels = reinterpret_cast<const uint32_t *>(p); p += pad8((uint64_t)hd.n * 4);
    const unsigned char *dead = p; p += pad8(hd.n);
    const float *vectors = reinterpret_cast<const float *>(p); p += pad8((uint64_t)hd.n * hd.dim * 4);
    uint32_t dead_count = 0;
    for (uint32_t i = 0; i < hd.n; ++i) {
        if (dead[i] > 1) return fail(h, HNSW_ERR_INVALID, "bad tombstone byte");
        dead_count += dead[i];
    }
    if (dead_count != hd.n_dead) return fail(h, HNSW_ERR_INVALID, "tombstone count does not match the header");
    if ((hd.enterpoint < 0) != (hd.n_dead == hd.n)) return fail(h, HNSW_ERR_INVALID, "enterpoint / live count mismatch");
    if (hd.enterpoint >= 0 && (dead[hd.enterpoint] || levels[hd.enterpoint] != hd.max_layer))
        return fail(h, HNSW_ERR_INVALID, "the enterpoint must be a live node of the top layer");
    std::vector<const uint64_t *> rps(hd.n_layers);
    std::vector<const uint32_t *> cols(hd.n_layers);
    for (uint32_t l = 0; l < hd.n_layers; ++l) {
        if ((uint64_t)(end - p) < 8 + ((uint64_t)hd.n + 1) * 8) return fail(h, HNSW_ERR_INVALID, "truncated snapshot");
        uint64_t nnz;
        std::memcpy(&nnz, p, 8); p += 8;
        rps[l] = reinterpret_cast<const uint64_t *>(p); p += ((size_t)hd.n + 1) * 8;
        if (nnz > (uint64_t)(end - p) / 4 || rps[l][hd.n] != nnz) return fail(h, HNSW_ERR_INVALID, "truncated snapshot (layer)");
        cols[l] = reinterpret_cast<const uint32_t *>(p); p += pad8(nnz * 4);
        if (p > end) return fail(h, HNSW_ERR_INVALID, "truncated snapshot (padding)");
    }
    // hnsw_import wants the enterpoint on the top layer.  Tombstones keep their level, which can be above the
    // live graph's top (the enterpoint was deleted and a lower node elected, core.rs:449-472): import with the
    // highest node as a placeholder and empty rows for the layers only tombstones reach, then fix up.
    int64_t ep = 0;
    uint32_t top = 0;
    for (uint32_t i = 0; i < hd.n; ++i) {
        if (levels[i] >= kMaxLayers) return fail(h, HNSW_ERR_INVALID, "level too large");
        if (levels[i] > top) { top = levels[i]; ep = i; }
        if (!dead[i] && levels[i] > hd.max_layer) return fail(h, HNSW_ERR_INVALID, "live node above max_layer");
    }
    if (hd.enterpoint >= 0 && levels[hd.enterpoint] == top) ep = hd.enterpoint;
    std::vector<uint64_t> zero_rp;
    if (top + 1 > hd.n_layers) {
        zero_rp.assign((size_t)hd.n + 1, 0);
        rps.resize(top + 1, zero_rp.data());
        cols.resize(top + 1, nullptr);
    }
    if ((s = hnsw_import(h, hd.n, vectors, levels, ep, (uint32_t)rps.size(), rps.data(), cols.data())) != HNSW_OK) return s;
    h->h_dead.assign(dead, dead + hd.n);
    h->n_dead = hd.n_dead;
    h->enterpoint = hd.enterpoint;
    h->max_layer = hd.max_layer;
    return push_header(h);
}

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
    HIP_TRY(h, hipMemset((char *)h->d_hdr + offsetof(DevHeader, ctr_search), 0, sizeof(unsigned long long) * 16));
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
