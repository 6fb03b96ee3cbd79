// hnsw_tu_occpar.hip -- the round's commits in validated parallel groups (hnsw_occ_par.hpp: k_occ_commit_par) for one
// metric variant (HNSW_VARIANT, see hnsw_host.hpp), and its launcher.
#include "hnsw_wave_sync.hpp"   // one-wave workgroups: "__syncthreads()" of the shared insert code = the wave's own full synchronisation
#include "hnsw_host.hpp"
#include "hnsw_occ_par.hpp"

namespace hnsw_host {

template <int MODE, int T, int R>
static hnsw_status commit_par_t(hnsw_index *h, const InsertCfg &c, const OccBufs &ob, uint32_t count, uint32_t end_node, bool *done)
{
    const size_t lds = kOccScratchBytes + kParLdsBytes + c.lds;
    if (lds > 160 * 1024 - 2048 || count > 64 || !h->d_par) return HNSW_OK;          // (the in-order kernel takes the round)
    auto kc = k_occ_commit_par<MODE, T, R>;
    {
        static std::mutex mu;
        static bool attr_set[16] = {false};
        std::lock_guard<std::mutex> lock(mu);
        if (!attr_set[h->device & 15]) {
            HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void *>(kc), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 2048));
            attr_set[h->device & 15] = true;
        }
    }
    ParBufs pb = {reinterpret_cast<OccPar *>(h->d_par), reinterpret_cast<OccDelta *>(h->d_par_delta), h->d_par_rows, h->par_ovstride};
    hipLaunchKernelGGL(kc, dim3(count), dim3(64), lds, h->stream, view_tag(h, c.tagcfg), ob, pb, end_node, h->m, c.lnb, c.lcap, h->d_spill,
                       h->spill_gnb, h->d_plan, h->m + 2 + h->occ_slack_extra);
    HIP_TRY(h, hipGetLastError());
    *done = true;
    return HNSW_OK;
}

template <int MODE, int T>
hnsw_status occ_commit_par_r(hnsw_index *h, const InsertCfg &c, const OccBufs &ob, uint32_t count, uint32_t end_node, bool *done)
{
    *done = false;
    switch (c.R) {
    case 1: return commit_par_t<MODE, T, 1>(h, c, ob, count, end_node, done);
    case 4: return commit_par_t<MODE, T, 4>(h, c, ob, count, end_node, done);
    case 8: return commit_par_t<MODE, T, 8>(h, c, ob, count, end_node, done);
    case 16: return commit_par_t<MODE, T, 16>(h, c, ob, count, end_node, done);
    }
    return HNSW_OK;                                      // (ef_construction > 1024: the in-order commit)
}

template hnsw_status occ_commit_par_r<kVarMode, kVarT>(hnsw_index *, const InsertCfg &, const OccBufs &, uint32_t, uint32_t, bool *);

} // namespace hnsw_host
