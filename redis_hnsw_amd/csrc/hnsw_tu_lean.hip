// hnsw_tu_lean.hip -- the specialised dim-128 search kernel k_search_lean<VEC,R,BB,DB,WIDE> for one vector format
// and row width (HNSW_VARIANT 0: f32 rows, the reference's data; 1: the bf16 serving copy; 2 / 3: the same two
// for adjacency rows of 64..127 ids; 4 / 5: the fp8 serving copy, narrow / wide rows) and its launcher.
#define HNSW_SYNC_BLOCK   // search / engine unit: 64-thread workgroups handing over through LDS only (hnsw_device.hpp)
#include "hnsw_host.hpp"
#include "hnsw_search_lean.hpp"
#include <type_traits>

namespace hnsw_host {

// no HBM spill table involved, so no region bookkeeping either
template <class VEC, int R, int BB, int DB, bool WIDE, bool TIES = false>
static hnsw_status launch_lean_t(hnsw_index *h, const float *dQ, uint32_t B, uint32_t k, uint32_t idbits, uint32_t per_cu,
                                 uint32_t *d_ids, float *d_sims, uint32_t *d_nout, hipStream_t st)
{
    const size_t lds = LeanW<R>::kBytes + ((size_t)16 << BB);
    auto kern = k_search_lean<VEC, R, BB, DB, WIDE, TIES>;
    static size_t lds_set[16] = {0};
    hnsw_status ss = raise_lds_attr(h, kern, lds, lds_set);
    if (ss != HNSW_OK) return ss;
    // one block per query: a batch larger than the chip holds queues in the dispatcher, which hands a new query to
    // whichever slot frees first (queries differ in length by 2x; a grid-stride loop would fix the pairing up front)
    uint32_t grid = h->grid_stride ? std::min(B, 256u * std::max(per_cu, 8u)) : B;
    if (h->grid_override > 0) grid = std::min<uint32_t>(grid, (uint32_t)h->grid_override);
    if (h->time_launches) HIP_TRY(h, hipEventRecord(h->ev0, st));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64), lds, st, view(h), dQ, B, k, h->efc, (1u << BB) * 6u, idbits, d_ids, d_sims,
                       d_nout, TIES && h->tie_mode == 1 ? h->d_tie_flags : (uint32_t *)nullptr);
    HIP_TRY(h, hipGetLastError());
    if (h->time_launches) {
        HIP_TRY(h, hipEventRecord(h->ev1, st));
        h->ev_valid = true;
    }
    return note_search(h, st);                           // inserts wait for searches in flight
}

template <class VEC, bool WIDE>
hnsw_status launch_lean_v(hnsw_index *h, int R, uint32_t bb, uint32_t db, const float *dQ, uint32_t B, uint32_t k,
                          uint32_t idbits, uint32_t per_cu, uint32_t *d_ids, float *d_sims, uint32_t *d_nout,
                          hipStream_t st, bool *done)
{
    *done = true;
    // the tie census (tuning "tie_census", hnsw_get_tie_counters): f32 rows, the shapes the headline configurations use
    if constexpr (std::is_same<VEC, VecF32<4>>::value) {
        if (h->tie_census) {
#define TIE_CASE(RR, BBB, DDB)                                                                                  \
    if (R == RR && bb == BBB && db == DDB)                                                                      \
        return launch_lean_t<VEC, RR, BBB, DDB, WIDE, true>(h, dQ, B, k, idbits, per_cu, d_ids, d_sims, d_nout, st);
            TIE_CASE(1, 10, 3) TIE_CASE(1, 11, 3) TIE_CASE(4, 10, 3) TIE_CASE(4, 11, 3) TIE_CASE(1, 9, 3) TIE_CASE(4, 9, 3)
#undef TIE_CASE
            h->tie_uncounted = true;                     // (another shape: the plain kernel answers, hnsw_get_tie_counters says "unknown")
        }
    } else if (h->tie_census) h->tie_uncounted = true;
#define LEAN_CASE(RR, BBB, DDB)                                                                                  \
    if (R == RR && bb == BBB && db == DDB)                                                                      \
        return launch_lean_t<VEC, RR, BBB, DDB, WIDE>(h, dQ, B, k, idbits, per_cu, d_ids, d_sims, d_nout, st);
    LEAN_CASE(1, 10, 3) LEAN_CASE(1, 11, 3) LEAN_CASE(4, 10, 3) LEAN_CASE(4, 11, 3)
    LEAN_CASE(1, 10, 2) LEAN_CASE(1, 11, 2) LEAN_CASE(4, 10, 2) LEAN_CASE(4, 11, 2)
    LEAN_CASE(1, 9, 3) LEAN_CASE(4, 9, 3)
    // ef_construction 257..512 (W in eight register slices)
    LEAN_CASE(8, 10, 3) LEAN_CASE(8, 11, 3) LEAN_CASE(8, 10, 2) LEAN_CASE(8, 11, 2) LEAN_CASE(8, 9, 3)
#undef LEAN_CASE
    *done = false;
    return HNSW_OK;
}

#define LEAN_INSTANCE(VEC, WIDE)                                                                                          \
    template hnsw_status launch_lean_v<VEC, WIDE>(hnsw_index *, int, uint32_t, uint32_t, const float *, uint32_t, uint32_t, \
                                                  uint32_t, uint32_t, uint32_t *, float *, uint32_t *, hipStream_t, bool *);
#if HNSW_VARIANT == 0
LEAN_INSTANCE(VecF32<4>, false)
#elif HNSW_VARIANT == 1
LEAN_INSTANCE(VecBF16<4>, false)
#elif HNSW_VARIANT == 2
LEAN_INSTANCE(VecF32<4>, true)
#elif HNSW_VARIANT == 3
LEAN_INSTANCE(VecBF16<4>, true)
#elif HNSW_VARIANT == 4
LEAN_INSTANCE(VecFP8<4>, false)
#else
LEAN_INSTANCE(VecFP8<4>, true)
#endif

} // namespace hnsw_host
