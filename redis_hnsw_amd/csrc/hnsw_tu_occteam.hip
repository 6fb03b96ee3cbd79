// hnsw_tu_occteam.hip -- the commit kernels of the exact-order parallel insert / delete (hnsw_occ.hpp) as a TEAM: the
// committing wavefront plus three helpers that share every recomputed select_neighbors with it (team_select), for one
// metric variant (HNSW_VARIANT, see hnsw_host.hpp).
//
// As in hnsw_tu_planduo.hip: the shared insert code synchronises "the block" where it means "this wavefront"; here
// the block has four wavefronts that each run that code on their own, with loops of different lengths, so in THIS
// unit a block-level synchronisation of the shared code is the wave's own, and the only s_barriers are the team's
// hand-overs (team_bar).
#define HNSW_SYNC_WAVE_FULL   // the shared code is run by one wavefront per copy and hands over through HBM too (hnsw_wave_sync.hpp)
#include "hnsw_host.hpp"

namespace hnsw_host {

constexpr int kTeamHelpers = 3;

// the helpers' LDS share and visited table, from what the committing wave leaves of the CU's LDS
template <int T>
static bool team_cfg(const hnsw_index *h, const InsertCfg &c, TeamCfg *tc, size_t *lds_total)
{
    const size_t cap = 160 * 1024 - 2048;
    const size_t base = kOccScratchBytes + c.lds + sizeof(TeamTask) + (size_t)kTeamCand * 8;
    if (base >= cap) return false;
    const size_t per = ((cap - base) / kTeamHelpers) & ~(size_t)63;
    const size_t fixed = team_fixed_bytes(T, h->dim);
    if (per < fixed + 2048) return false;
    const size_t tb = per - fixed;
    tc->hbytes = (uint32_t)per;
    tc->lnb = (uint32_t)(tb / 32);
    tc->lcap = tc->lnb * h->lds_fill_x2 / 2;
    tc->tagcfg = 0;
    if (c.tagcfg) {                                      // the committing wave's table is in tag form: so are the helpers'
        const uint32_t idbits = c.tagcfg >> 8;
        uint32_t bb = 0;
        while (((size_t)16 << (bb + 1)) <= tb && bb + 1 <= 12) ++bb;
        if (bb >= 2 && idbits >= bb && idbits - bb <= 13) {
            tc->tagcfg = bb | (idbits << 8);
            tc->lcap = (1u << bb) * 6u;
        }
    }
    tc->gnb = h->spill_gnb;
    *lds_total = base + per * kTeamHelpers;
    return h->spill_slots >= (uint32_t)kTeamHelpers;
}

template <int MODE, int T, int R>
static hnsw_status commit_team_t(hnsw_index *h, const InsertCfg &c, const OccBufs &ob, uint32_t end_node, bool *done)
{
    TeamCfg tc;
    size_t lds = 0;
    if (!team_cfg<T>(h, c, &tc, &lds)) return HNSW_OK;
    auto kc = k_occ_commit<MODE, T, R, kTeamHelpers>;
    {
        static std::mutex mu;
        static bool attr_set[16] = {false};
        std::lock_guard<std::mutex> lock(mu);
        if (!attr_set[h->device & 15]) {
            HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void *>(kc), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 2048));
            attr_set[h->device & 15] = true;
        }
    }
    hipLaunchKernelGGL(kc, dim3(1), dim3(64 * (1 + kTeamHelpers)), lds, h->stream, view_tag(h, c.tagcfg), ob, end_node, h->m, c.lnb, c.lcap,
                       h->d_spill_one, h->spill_one_gnb, h->d_plan, occ_slack(h),
                       h->occ_want_touched ? h->d_touched : nullptr, h->occ_want_touched ? h->touched_cap : 0u, (uint32_t)c.lds, tc, h->d_spill,
                       h->occ_chained ? 1u : 0u);
    HIP_TRY(h, hipGetLastError());
    *done = true;
    return HNSW_OK;
}

template <int MODE, int T, int R>
static hnsw_status del_commit_team_t(hnsw_index *h, const InsertCfg &c, const OccBufs &ob, uint32_t id, bool *done)
{
    TeamCfg tc;
    size_t lds = 0;
    if (!team_cfg<T>(h, c, &tc, &lds)) return HNSW_OK;
    auto kd = k_occ_del_commit<MODE, T, R, kTeamHelpers>;
    {
        static std::mutex mu;
        static bool attr_set[16] = {false};
        std::lock_guard<std::mutex> lock(mu);
        if (!attr_set[h->device & 15]) {
            HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void *>(kd), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 2048));
            attr_set[h->device & 15] = true;
        }
    }
    hipLaunchKernelGGL(kd, dim3(1), dim3(64 * (1 + kTeamHelpers)), lds, h->stream, view_tag(h, c.tagcfg), ob, id, h->m, c.lnb, c.lcap,
                       h->d_spill_one, h->spill_one_gnb, h->d_touched, h->touched_cap, (uint32_t)c.lds, tc, h->d_spill);
    HIP_TRY(h, hipGetLastError());
    *done = true;
    return HNSW_OK;
}

template <int MODE, int T>
hnsw_status occ_commit_team_r(hnsw_index *h, const InsertCfg &c, const OccBufs &ob, uint32_t end_node, bool *done)
{
    *done = false;
    switch (c.R) {
    case 1: return commit_team_t<MODE, T, 1>(h, c, ob, end_node, done);
    case 4: return commit_team_t<MODE, T, 4>(h, c, ob, end_node, done);
    case 8: return commit_team_t<MODE, T, 8>(h, c, ob, end_node, done);
    case 16: return commit_team_t<MODE, T, 16>(h, c, ob, end_node, done);
    }
    return HNSW_OK;                                      // (ef_construction > 1024: the one-wave commit)
}

template <int MODE, int T>
hnsw_status occ_del_commit_team_r(hnsw_index *h, const InsertCfg &c, const OccBufs &ob, uint32_t id, bool *done)
{
    *done = false;
    switch (c.R) {
    case 1: return del_commit_team_t<MODE, T, 1>(h, c, ob, id, done);
    case 4: return del_commit_team_t<MODE, T, 4>(h, c, ob, id, done);
    case 8: return del_commit_team_t<MODE, T, 8>(h, c, ob, id, done);
    case 16: return del_commit_team_t<MODE, T, 16>(h, c, ob, id, done);
    }
    return HNSW_OK;
}

template hnsw_status occ_commit_team_r<kVarMode, kVarT>(hnsw_index *, const InsertCfg &, const OccBufs &, uint32_t, bool *);
template hnsw_status occ_del_commit_team_r<kVarMode, kVarT>(hnsw_index *, const InsertCfg &, const OccBufs &, uint32_t, bool *);

} // namespace hnsw_host
