// hnsw_tu_occpar.hip -- the round's commits in validated parallel groups (hnsw_occ_par.hpp: k_occ_commit_par) for one
// metric variant (HNSW_VARIANT, see hnsw_host.hpp), and its launcher.
#define HNSW_SYNC_WAVE_FULL   // the shared code is run by one wavefront per copy and hands over through HBM too (hnsw_wave_sync.hpp)
#include "hnsw_host.hpp"
#include "hnsw_occ_par.hpp"

namespace hnsw_host {

constexpr int kParHelpers = 3;

// the helpers' LDS share and visited table, from what the committing wave leaves of the CU's LDS (as hnsw_tu_occteam.hip)
template <int T>
static bool par_team_cfg(const hnsw_index *h, const InsertCfg &c, TeamCfg *tc, size_t *lds_total)
{
    const size_t cap = 160 * 1024 - 2048;
    const size_t base = kOccScratchBytes + kParLdsBytes + c.lds + sizeof(TeamTask) + (size_t)kTeamCand * 8;
    if (base >= cap) return false;
    const size_t per = ((cap - base) / kParHelpers) & ~(size_t)63;
    const size_t fixed = team_fixed_bytes(T, h->dim);
    if (per < fixed + 2048) return false;
    const size_t tb = per - fixed;
    tc->hbytes = (uint32_t)per;
    tc->lnb = (uint32_t)(tb / 32);
    tc->lcap = tc->lnb * h->lds_fill_x2 / 2;
    tc->tagcfg = 0;
    if (c.tagcfg) {
        const uint32_t idbits = c.tagcfg >> 8;
        uint32_t bb = 0;
        while (((size_t)16 << (bb + 1)) <= tb && bb + 1 <= 12) ++bb;
        if (bb >= 2 && idbits >= bb && idbits - bb <= 13) {
            tc->tagcfg = bb | (idbits << 8);
            tc->lcap = (1u << bb) * 6u;
        }
    }
    tc->gnb = h->spill_gnb;
    *lds_total = base + per * kParHelpers;
    return h->spill_slots >= 64u + 64u * kParHelpers;
}

// The iterations of k_occ_commit_par are separated by a spin barrier over the whole grid (par_barrier): every
// workgroup must be RESIDENT, or the ones that are spin for ever.  A workgroup asks for nearly a CU's whole LDS, so the
// co-resident count is (workgroups per CU at this block size and LDS) x (CUs of this device or partition -- 32 in CPX
// mode).  Asked once per (kernel, LDS size) and handle; a round with more window nodes than that goes to the in-order
// commit kernel.
template <typename Kern>
static hnsw_status par_resident(hnsw_index *h, Kern kern, int block, size_t lds, uint32_t *out)
{
    const void *kp = reinterpret_cast<const void *>(kern);
    if (h->par_res_kernel == kp && h->par_res_lds == lds) { *out = h->par_res_n; return HNSW_OK; }
    int per_cu = 0, cus = 0;
    HIP_TRY(h, hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kp, block, lds));
    HIP_TRY(h, hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, h->device));
    h->par_res_kernel = kp;
    h->par_res_lds = lds;
    h->par_res_n = (uint32_t)std::max(per_cu, 0) * (uint32_t)std::max(cus, 0);
    *out = h->par_res_n;
    return HNSW_OK;
}

template <int MODE, int T, int R>
static hnsw_status commit_par_t(hnsw_index *h, const InsertCfg &c, const OccBufs &ob, uint32_t count, uint32_t end_node, bool *done)
{
    const size_t lds = kOccScratchBytes + kParLdsBytes + c.lds;
    // (else the in-order kernel takes the round.  The overlay's keys are row << 5 | layer: ids must stay below 2^27)
    if (lds > 160 * 1024 - 2048 || count > 64 || !h->d_par || !h->d_par_delta || !h->d_par_rows || h->cap > (1u << 27)) return HNSW_OK;
    ParBufs pb = {reinterpret_cast<OccPar *>(h->d_par), reinterpret_cast<OccDelta *>(h->d_par_delta), h->d_par_rows, h->par_ovstride};
    TeamCfg tc;
    size_t lds_team = 0;
    if (h->commit_team && par_team_cfg<T>(h, c, &tc, &lds_team)) {
        auto kt = k_occ_commit_par<MODE, T, R, kParHelpers>;
        {
            static std::mutex mu;
            static bool attr_set[16] = {false};
            std::lock_guard<std::mutex> lock(mu);
            if (!attr_set[h->device & 15]) {
                HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void *>(kt), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 2048));
                attr_set[h->device & 15] = true;
            }
        }
        uint32_t resident = 0;
        hnsw_status rs = par_resident(h, kt, 64 * (1 + kParHelpers), lds_team, &resident);
        if (rs != HNSW_OK) return rs;
        if (count > std::min(resident, h->par_max_resident)) return HNSW_OK;      // not all workgroups would be resident: the in-order commit
        hipLaunchKernelGGL(kt, dim3(count), dim3(64 * (1 + kParHelpers)), lds_team, h->stream, view_tag(h, c.tagcfg), ob, pb, end_node, h->m, c.lnb,
                           c.lcap, h->d_spill, h->spill_gnb, h->d_plan, occ_slack(h), (uint32_t)c.lds, tc, h->occ_chained ? 1u : 0u,
                           h->occ_want_touched ? h->d_touched : (uint32_t *)nullptr, h->occ_want_touched ? h->touched_cap : 0u);
        HIP_TRY(h, hipGetLastError());
        *done = true;
        return HNSW_OK;
    }
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
    uint32_t resident = 0;
    hnsw_status rs = par_resident(h, kc, 64, lds, &resident);
    if (rs != HNSW_OK) return rs;
    if (count > std::min(resident, h->par_max_resident)) return HNSW_OK;          // not all workgroups would be resident: the in-order commit
    hipLaunchKernelGGL(kc, dim3(count), dim3(64), lds, h->stream, view_tag(h, c.tagcfg), ob, pb, end_node, h->m, c.lnb, c.lcap, h->d_spill,
                       h->spill_gnb, h->d_plan, occ_slack(h), 0u, TeamCfg{}, h->occ_chained ? 1u : 0u,
                       h->occ_want_touched ? h->d_touched : (uint32_t *)nullptr, h->occ_want_touched ? h->touched_cap : 0u);
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
