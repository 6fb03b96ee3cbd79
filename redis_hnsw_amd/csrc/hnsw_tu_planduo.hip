// hnsw_tu_planduo.hip -- the insert plans in their TWO-WAVE form (hnsw_plan_lean.hpp with DUO = true): the first
// wavefront is the one-wave plan kernel, the second keeps W for its layer searches (hnsw_search_duo.hpp).
// HNSW_VARIANT 0: adjacency rows of at most 63 ids, 1: 127 ids.
//
// The insert code shared with the one-wave kernels (select_neighbors, the node's own rows, the list of shrinks)
// synchronises "the block" where it means "this wavefront" -- in a 64-thread block the two are the same thing and
// the compiler drops the s_barrier.  Here the block has a second wave that takes no part in any of that, and the
// only s_barriers the first wave may execute are the two of each hand-over with the keeper: in THIS unit a block-level
// synchronisation of the shared code is the wave's own (same fences, a wave barrier instead of s_barrier).
#define HNSW_SYNC_WAVE_FULL   // the shared code is run by one wavefront per copy and hands over through HBM too (hnsw_wave_sync.hpp)
#include "hnsw_host.hpp"
#include "hnsw_plan_lean.hpp"

namespace hnsw_host {

template <int R, bool WIDE>
static hnsw_status plan_duo_t(hnsw_index *h, const InsertCfg &c, uint32_t first, uint32_t count, uint32_t idbits)
{
    auto kern = k_insert_plan_lean<R, 3, WIDE, true>;
    const size_t lds = c.lds + plan_lean_bytes<R>();
    HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const uint32_t grid = std::min(count, h->spill_slots);
    hnsw_status ws = wait_inflight_searches(h);
    if (ws != HNSW_OK) return ws;
    const bool one = count == 1 && h->d_spill_one;      // exact insert: the table that cannot overflow
    hipLaunchKernelGGL(kern, dim3(grid), dim3(128), lds, h->stream, view_tag(h, c.tagcfg), first, count, h->efc, h->m, c.lnb, c.lcap,
                       one ? h->d_spill_one : h->d_spill, one ? h->spill_one_gnb : h->spill_gnb, h->d_plan, h->select_shortcut ? 1u : 0u,
                       (uint32_t)c.lds, idbits);
    HIP_TRY(h, hipGetLastError());
    return HNSW_OK;
}

template <int R, bool WIDE>
static hnsw_status occ_plan_duo_t(hnsw_index *h, const InsertCfg &c, const OccBufs &ob, uint32_t head, uint32_t count, uint32_t idbits)
{
    auto kern = k_occ_plan_lean<R, 3, WIDE, true>;
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
    hipLaunchKernelGGL(kern, dim3(count + h->occ_far), dim3(128), lds, h->stream, view_tag(h, c.tagcfg), ob, head, count, h->efc, h->m, c.lnb, c.lcap,
                       h->d_spill, h->spill_gnb, h->d_plan, h->select_shortcut ? 1u : 0u, h->occ_log_cap, (uint32_t)c.lds, idbits, h->plan_split_pos,
                       h->occ_far);
    HIP_TRY(h, hipGetLastError());
    return HNSW_OK;
}

template <bool WIDE>
hnsw_status launch_plan_duo_v(hnsw_index *h, const InsertCfg &c, uint32_t first, uint32_t count, uint32_t idbits)
{
    switch (c.R) {
    case 1: return plan_duo_t<1, WIDE>(h, c, first, count, idbits);
    case 4: return plan_duo_t<4, WIDE>(h, c, first, count, idbits);
    case 8: return plan_duo_t<8, WIDE>(h, c, first, count, idbits);
    }
    return fail(h, HNSW_ERR_INVALID, "bad R for the specialised plan");
}

template <bool WIDE>
hnsw_status launch_occ_plan_duo_v(hnsw_index *h, const InsertCfg &c, const OccBufs &ob, uint32_t head, uint32_t count, uint32_t idbits)
{
    switch (c.R) {
    case 1: return occ_plan_duo_t<1, WIDE>(h, c, ob, head, count, idbits);
    case 4: return occ_plan_duo_t<4, WIDE>(h, c, ob, head, count, idbits);
    case 8: return occ_plan_duo_t<8, WIDE>(h, c, ob, head, count, idbits);
    }
    return fail(h, HNSW_ERR_INVALID, "bad R for the specialised plan");
}

#if HNSW_VARIANT == 0
template hnsw_status launch_plan_duo_v<false>(hnsw_index *, const InsertCfg &, uint32_t, uint32_t, uint32_t);
template hnsw_status launch_occ_plan_duo_v<false>(hnsw_index *, const InsertCfg &, const OccBufs &, uint32_t, uint32_t, uint32_t);
#else
template hnsw_status launch_plan_duo_v<true>(hnsw_index *, const InsertCfg &, uint32_t, uint32_t, uint32_t);
template hnsw_status launch_occ_plan_duo_v<true>(hnsw_index *, const InsertCfg &, const OccBufs &, uint32_t, uint32_t, uint32_t);
#endif

} // namespace hnsw_host
