// hnsw_tu_search.hip -- the general search kernel k_search<MODE,T,R,FMT> for one metric variant (HNSW_VARIANT 0..3,
// see hnsw_host.hpp) or one compressed storage format (4: bf16 rows, 5: fp8 rows) and its launcher.
#define HNSW_SYNC_BLOCK   // search / engine unit: 64-thread workgroups handing over through LDS only (hnsw_device.hpp)
#include "hnsw_host.hpp"
#include "hnsw_kernels.hpp"

namespace hnsw_host {

template <int MODE, int T, int R, int FMT>
static hnsw_status launch_search_t(hnsw_index *h, const float *dQ, uint32_t B, uint32_t k, uint32_t *d_ids,
                                   float *d_sims, uint32_t *d_nout, hipStream_t st)
{
    const VisCfg vc = pick_vis(h, R, T, false, B);
    const uint32_t lnb = vc.lnb;
    const size_t lds = lds_fixed_bytes(R, T, h->dim, false) + vc.bytes;
    GraphView gv = view(h);
    gv.tagcfg = vc.tagcfg;
    auto kern = k_search<MODE, T, R, FMT>;
    static size_t lds_set[16] = {0};
    hnsw_status ss = raise_lds_attr(h, kern, lds, lds_set);
    if (ss != HNSW_OK) return ss;
    uint32_t grid = std::min(B, h->spill_slots);
    if (h->grid_override > 0) grid = std::min<uint32_t>(grid, (uint32_t)h->grid_override);
    uint32_t region, *spill;
    if ((ss = spill_acquire(h, st, &region, &spill)) != HNSW_OK) return ss;
    if (h->time_launches) HIP_TRY(h, hipEventRecord(h->ev0, st));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64), lds, st, gv, dQ, B, k, h->efc, lnb, vc.lcap, spill,
                       h->spill_gnb, d_ids, d_sims, d_nout, h->visited_bounded ? 1u : 0u);
    HIP_TRY(h, hipGetLastError());
    if (h->time_launches) {
        HIP_TRY(h, hipEventRecord(h->ev1, st));
        h->ev_valid = true;
    }
    return spill_release(h, st, region);
}

template <int MODE, int T>
hnsw_status launch_search_r(hnsw_index *h, int R, const float *dQ, uint32_t B, uint32_t k, uint32_t *d_ids,
                            float *d_sims, uint32_t *d_nout, hipStream_t st)
{
    switch (R) {
    case 1: return launch_search_t<MODE, T, 1, FMT_F32>(h, dQ, B, k, d_ids, d_sims, d_nout, st);
    case 4: return launch_search_t<MODE, T, 4, FMT_F32>(h, dQ, B, k, d_ids, d_sims, d_nout, st);
    case 8: return launch_search_t<MODE, T, 8, FMT_F32>(h, dQ, B, k, d_ids, d_sims, d_nout, st);
    case 16: return launch_search_t<MODE, T, 16, FMT_F32>(h, dQ, B, k, d_ids, d_sims, d_nout, st);
    case 64: return launch_search_t<MODE, T, 64, FMT_F32>(h, dQ, B, k, d_ids, d_sims, d_nout, st);
    }
    return fail(h, HNSW_ERR_INVALID, "ef_construction > 4096 is not supported");
}

// compressed storage (bf16 / fp8 rows): any dim % 32 == 0, query pieces in LDS (T = 0)
template <int FMT>
hnsw_status launch_search_fmt(hnsw_index *h, int R, const float *dQ, uint32_t B, uint32_t k, uint32_t *d_ids,
                              float *d_sims, uint32_t *d_nout, hipStream_t st)
{
    switch (R) {
    case 1: return launch_search_t<MODE_AVX, 0, 1, FMT>(h, dQ, B, k, d_ids, d_sims, d_nout, st);
    case 4: return launch_search_t<MODE_AVX, 0, 4, FMT>(h, dQ, B, k, d_ids, d_sims, d_nout, st);
    case 8: return launch_search_t<MODE_AVX, 0, 8, FMT>(h, dQ, B, k, d_ids, d_sims, d_nout, st);
    case 16: return launch_search_t<MODE_AVX, 0, 16, FMT>(h, dQ, B, k, d_ids, d_sims, d_nout, st);
    }
    return fail(h, HNSW_ERR_INVALID, "compressed storage serves ef_construction <= 1024");
}

#if HNSW_VARIANT == 4
template hnsw_status launch_search_fmt<FMT_BF16>(hnsw_index *, int, const float *, uint32_t, uint32_t, uint32_t *, float *, uint32_t *, hipStream_t);
#elif HNSW_VARIANT == 5
template hnsw_status launch_search_fmt<FMT_FP8>(hnsw_index *, int, const float *, uint32_t, uint32_t, uint32_t *, float *, uint32_t *, hipStream_t);
#else
template hnsw_status launch_search_r<kVarMode, kVarT>(hnsw_index *, int, const float *, uint32_t, uint32_t, uint32_t *,
                                                       float *, uint32_t *, hipStream_t);
#endif

} // namespace hnsw_host
