// hnsw_tu_planlean.hip -- the insert plans that search with the specialised dim-128 routine (hnsw_plan_lean.hpp):
// k_insert_plan_lean (single hnsw_add) and k_occ_plan_lean (windowed exact build), for adjacency rows of at most
// 63 ids (HNSW_VARIANT 0) or 127 ids (HNSW_VARIANT 1), and their launchers.
#define HNSW_SYNC_WAVE_FULL   // the shared code is run by one wavefront per copy and hands over through HBM too (hnsw_wave_sync.hpp)
#include "hnsw_host.hpp"
#include "hnsw_plan_lean.hpp"

namespace hnsw_host {

template <int R, bool WIDE>
static hnsw_status plan_lean_t(hnsw_index *h, const InsertCfg &c, uint32_t first, uint32_t count, uint32_t idbits)
{
    auto kern = k_insert_plan_lean<R, 3, WIDE>;
    const size_t lds = c.lds + plan_lean_bytes<R>();
    HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const uint32_t grid = std::min(count, h->spill_slots);
    hnsw_status ws = wait_inflight_searches(h);
    if (ws != HNSW_OK) return ws;
    const bool one = count == 1 && h->d_spill_one;      // exact insert: the table that cannot overflow
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64), lds, h->stream, view_tag(h, c.tagcfg), first, count, h->efc, h->m, c.lnb, c.lcap,
                       one ? h->d_spill_one : h->d_spill, one ? h->spill_one_gnb : h->spill_gnb, h->d_plan, h->select_shortcut ? 1u : 0u,
                       (uint32_t)c.lds, idbits);
    HIP_TRY(h, hipGetLastError());
    return HNSW_OK;
}

template <int R, bool WIDE>
static hnsw_status occ_plan_lean_t(hnsw_index *h, const InsertCfg &c, const OccBufs &ob, uint32_t head, uint32_t count, uint32_t idbits)
{
    auto kern = k_occ_plan_lean<R, 3, WIDE>;
    const size_t lds = c.lds + plan_lean_bytes<R>();
    {
        static std::mutex mu;
        static bool attr_set[16] = {false};
        std::lock_guard<std::mutex> lock(mu);
        if (!attr_set[h->device & 15]) {
            HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 2048));
            attr_set[h->device & 15] = true;
        }
    }
    hipLaunchKernelGGL(kern, dim3(count + h->occ_far), dim3(64), lds, h->stream, view_tag(h, c.tagcfg), ob, head, count, h->efc, h->m, c.lnb, c.lcap,
                       h->d_spill, h->spill_gnb, h->d_plan, h->select_shortcut ? 1u : 0u, h->occ_log_cap, (uint32_t)c.lds, idbits, h->plan_split_pos,
                       h->occ_far);
    HIP_TRY(h, hipGetLastError());
    return HNSW_OK;
}

template <bool WIDE>
hnsw_status launch_plan_lean_v(hnsw_index *h, const InsertCfg &c, uint32_t first, uint32_t count, uint32_t idbits)
{
    switch (c.R) {
    case 1: return plan_lean_t<1, WIDE>(h, c, first, count, idbits);
    case 4: return plan_lean_t<4, WIDE>(h, c, first, count, idbits);
    case 8: return plan_lean_t<8, WIDE>(h, c, first, count, idbits);
    }
    return fail(h, HNSW_ERR_INVALID, "bad R for the specialised plan");
}

template <bool WIDE>
hnsw_status launch_occ_plan_lean_v(hnsw_index *h, const InsertCfg &c, const OccBufs &ob, uint32_t head, uint32_t count, uint32_t idbits)
{
    switch (c.R) {
    case 1: return occ_plan_lean_t<1, WIDE>(h, c, ob, head, count, idbits);
    case 4: return occ_plan_lean_t<4, WIDE>(h, c, ob, head, count, idbits);
    case 8: return occ_plan_lean_t<8, WIDE>(h, c, ob, head, count, idbits);
    }
    return fail(h, HNSW_ERR_INVALID, "bad R for the specialised plan");
}

#if HNSW_VARIANT == 0
template hnsw_status launch_plan_lean_v<false>(hnsw_index *, const InsertCfg &, uint32_t, uint32_t, uint32_t);
template hnsw_status launch_occ_plan_lean_v<false>(hnsw_index *, const InsertCfg &, const OccBufs &, uint32_t, uint32_t, uint32_t);
#else
template hnsw_status launch_plan_lean_v<true>(hnsw_index *, const InsertCfg &, uint32_t, uint32_t, uint32_t);
template hnsw_status launch_occ_plan_lean_v<true>(hnsw_index *, const InsertCfg &, const OccBufs &, uint32_t, uint32_t, uint32_t);
#endif

} // namespace hnsw_host
