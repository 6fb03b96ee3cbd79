// hnsw_tu_duo.hip -- the two-wave form of the specialised dim-128 search kernel (hnsw_search_duo.hpp:
// k_search_duo<VEC,R,BB,DB,WIDE>, a walker and a W-keeper wavefront per query) for f32 rows, HNSW_VARIANT 0: rows
// of <= 63 ids, 1: rows of 64..127 ids, and its launcher.  Chosen by try_launch_lean when few enough queries are
// in flight that every query can have two SIMDs (hnsw_engine.hip).
#define HNSW_SYNC_BLOCK   // search / engine unit: 64-thread workgroups handing over through LDS only (hnsw_device.hpp)
#include "hnsw_host.hpp"
#include "hnsw_search_duo.hpp"

namespace hnsw_host {

template <class VEC, int R, int DB, bool WIDE>
static hnsw_status launch_duo_t(hnsw_index *h, const float *dQ, uint32_t B, uint32_t k, uint32_t idbits, uint32_t *d_ids,
                                float *d_sims, uint32_t *d_nout, hipStream_t st)
{
    constexpr int BB = 11;                                // 2048 buckets: 14 336 ids before the table stops recording
    const size_t lds = LeanW<R>::kBytes + kDuoBoxBytes + ((size_t)16 << BB);
    auto kern = k_search_duo<VEC, R, BB, DB, WIDE>;
    static size_t lds_set[16] = {0};
    hnsw_status ss = raise_lds_attr(h, kern, lds, lds_set);
    if (ss != HNSW_OK) return ss;
    uint32_t grid = B;                                    // one workgroup (two waves) per query
    if (h->grid_override > 0) grid = std::min<uint32_t>(grid, (uint32_t)h->grid_override);
    if (h->time_launches) HIP_TRY(h, hipEventRecord(h->ev0, st));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(128), lds, st, view(h), dQ, B, k, h->efc, (1u << BB) * 6u, idbits, d_ids, d_sims,
                       d_nout);
    HIP_TRY(h, hipGetLastError());
    if (h->time_launches) {
        HIP_TRY(h, hipEventRecord(h->ev1, st));
        h->ev_valid = true;
    }
    return note_search(h, st);                           // inserts wait for searches in flight
}

template <bool WIDE>
hnsw_status launch_duo_v(hnsw_index *h, int R, uint32_t db, const float *dQ, uint32_t B, uint32_t k, uint32_t idbits,
                         uint32_t *d_ids, float *d_sims, uint32_t *d_nout, hipStream_t st, bool *done)
{
    *done = true;
#define DUO_CASE(RR, DDB)                                                                                     \
    if (R == RR && db == DDB)                                                                                \
        return launch_duo_t<VecF32<4>, RR, DDB, WIDE>(h, dQ, B, k, idbits, d_ids, d_sims, d_nout, st);
    DUO_CASE(1, 3) DUO_CASE(4, 3) DUO_CASE(8, 3) DUO_CASE(1, 2) DUO_CASE(4, 2) DUO_CASE(8, 2)
#undef DUO_CASE
    *done = false;
    return HNSW_OK;
}

#if HNSW_VARIANT == 0
template hnsw_status launch_duo_v<false>(hnsw_index *, int, uint32_t, const float *, uint32_t, uint32_t, uint32_t, uint32_t *,
                                         float *, uint32_t *, hipStream_t, bool *);
#else
template hnsw_status launch_duo_v<true>(hnsw_index *, int, uint32_t, const float *, uint32_t, uint32_t, uint32_t, uint32_t *,
                                        float *, uint32_t *, hipStream_t, bool *);
#endif

} // namespace hnsw_host
