// hnsw_tu_insert.hip -- HNSW.NODE.ADD / HNSW.NODE.DEL kernels of one metric variant (HNSW_VARIANT, see
// hnsw_host.hpp): k_insert_plan, k_insert_commit_exact, k_delete_exact, k_shrink_batch, and their launchers.
#define HNSW_SYNC_WAVE_FULL   // the shared code is run by one wavefront per copy and hands over through HBM too (hnsw_wave_sync.hpp)
#include "hnsw_host.hpp"

namespace hnsw_host {

template <int MODE, int T, int R>
static hnsw_status launch_plan_t(hnsw_index *h, const InsertCfg &c, uint32_t first, uint32_t count)
{
    auto kern = k_insert_plan<MODE, T, R>;
    HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)c.lds));
    uint32_t grid = std::min(count, h->spill_slots);
    hnsw_status ws = wait_inflight_searches(h);
    if (ws != HNSW_OK) return ws;
    const bool one = count == 1 && h->d_spill_one;      // exact insert: the table that cannot overflow
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64), c.lds, h->stream, view_tag(h, c.tagcfg), first, count, h->efc, h->m, c.lnb, c.lcap,
                       one ? h->d_spill_one : h->d_spill, one ? h->spill_one_gnb : h->spill_gnb, h->d_plan, h->select_shortcut ? 1u : 0u);
    HIP_TRY(h, hipGetLastError());
    return HNSW_OK;
}
template <int MODE, int T, int R>
static hnsw_status launch_commit_t(hnsw_index *h, const InsertCfg &c, uint32_t id)
{
    auto kern = k_insert_commit_exact<MODE, T, R>;
    HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)c.lds));
    hipLaunchKernelGGL(kern, dim3(1), dim3(64), c.lds, h->stream, view_tag(h, c.tagcfg), id, h->m, c.lnb, c.lcap, h->d_spill_one,
                       h->spill_one_gnb, h->d_plan, h->d_touched, h->touched_cap);
    HIP_TRY(h, hipGetLastError());
    return HNSW_OK;
}

template <int MODE, int T>
hnsw_status launch_insert_r(hnsw_index *h, const InsertCfg &c, bool plan, uint32_t first, uint32_t count)
{
    switch (c.R) {
    case 1: return plan ? launch_plan_t<MODE, T, 1>(h, c, first, count) : launch_commit_t<MODE, T, 1>(h, c, first);
    case 4: return plan ? launch_plan_t<MODE, T, 4>(h, c, first, count) : launch_commit_t<MODE, T, 4>(h, c, first);
    case 8: return plan ? launch_plan_t<MODE, T, 8>(h, c, first, count) : launch_commit_t<MODE, T, 8>(h, c, first);
    case 16: return plan ? launch_plan_t<MODE, T, 16>(h, c, first, count) : launch_commit_t<MODE, T, 16>(h, c, first);
    case 64: return plan ? launch_plan_t<MODE, T, 64>(h, c, first, count) : launch_commit_t<MODE, T, 64>(h, c, first);
    }
    return fail(h, HNSW_ERR_INVALID, "bad R");
}

template <int MODE, int T>
hnsw_status launch_shrink_t(hnsw_index *h, uint32_t *pending0, uint32_t *pendingU, uint32_t *work_n)
{
    const size_t lds = 128 * 8 + 64 * 8 + (T == 0 ? (((size_t)h->dim * 4 + 15) & ~(size_t)15) : 0);
    const uint32_t grid = 2048;
    hipLaunchKernelGGL((k_shrink_batch<MODE, T>), dim3(grid), dim3(64), lds, h->stream, view(h), h->m, h->d_work, work_n, h->work_cap, pending0, pendingU);
    HIP_TRY(h, hipGetLastError());
    return HNSW_OK;
}

// ---- HNSW.NODE.DEL --------------------------------------------------------------------------
template <int MODE, int T>
hnsw_status launch_delete_r(hnsw_index *h, const InsertCfg &c, uint32_t id)
{
#define HNSW_DEL_CASE(RR)                                                                                       \
    case RR: {                                                                                                  \
        auto kern = k_delete_exact<MODE, T, RR>;                                                                \
        HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)c.lds)); \
        hipLaunchKernelGGL(kern, dim3(1), dim3(64), c.lds, h->stream, view_tag(h, c.tagcfg), id, h->m, c.lnb, c.lcap, \
                           h->d_spill_one, h->spill_one_gnb, h->d_touched, h->touched_cap);                    \
        HIP_TRY(h, hipGetLastError());                                                                          \
        return HNSW_OK;                                                                                         \
    }
    switch (c.R) {
        HNSW_DEL_CASE(1)
        HNSW_DEL_CASE(4)
        HNSW_DEL_CASE(8)
        HNSW_DEL_CASE(16)
        HNSW_DEL_CASE(64)                                 // ef_construction > 1024 (W in LDS), like launch_insert_r / occ_delete_r
    }
#undef HNSW_DEL_CASE
    return fail(h, HNSW_ERR_INVALID, "bad R");
}

template hnsw_status launch_insert_r<kVarMode, kVarT>(hnsw_index *, const InsertCfg &, bool, uint32_t, uint32_t);
template hnsw_status launch_shrink_t<kVarMode, kVarT>(hnsw_index *, uint32_t *, uint32_t *, uint32_t *);
template hnsw_status launch_delete_r<kVarMode, kVarT>(hnsw_index *, const InsertCfg &, uint32_t);

} // namespace hnsw_host
