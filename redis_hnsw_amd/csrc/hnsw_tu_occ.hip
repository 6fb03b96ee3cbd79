// hnsw_tu_occ.hip -- the exact-order parallel insert (hnsw_occ.hpp) for one metric variant (HNSW_VARIANT, see
// hnsw_host.hpp): k_occ_validate, k_occ_plan, k_occ_shrinks, k_occ_commit and the launcher of one round; k_occ_del_list,
// k_occ_del_commit and the launcher of a delete.
#define HNSW_SYNC_WAVE_FULL   // the shared code is run by one wavefront per copy and hands over through HBM too (hnsw_wave_sync.hpp)
#include "hnsw_host.hpp"

namespace hnsw_host {

template <int MODE, int T, int R>
static hnsw_status occ_round_t(hnsw_index *h, const InsertCfg &c, const OccBufs &ob, uint32_t head, uint32_t count, uint32_t end_node)
{
    const size_t lds_val = kOccScratchBytes + 64 * 8 + (T == 0 ? (((size_t)h->dim * 4 + 15) & ~(size_t)15) : 0);
    const size_t lds_commit = kOccScratchBytes + c.lds;
    auto kv = k_occ_validate<MODE, T>;
    auto kp = k_occ_plan<MODE, T, R>;
    auto kc = k_occ_commit<MODE, T, R>;
    auto ks = k_occ_shrinks<MODE, T, R>;
    {
        static std::mutex mu;
        static bool attr_set[16] = {false};
        std::lock_guard<std::mutex> lock(mu);
        if (!attr_set[h->device & 15]) {
            HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void *>(kv), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 2048));
            HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void *>(kp), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 2048));
            HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void *>(kc), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 2048));
            HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void *>(ks), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 2048));
            attr_set[h->device & 15] = true;
        }
    }
    const GraphView gv = view_tag(h, c.tagcfg);
    if (h->occ_chained) head = kEmpty;                    // a round enqueued ahead of the host: the kernels start at ctl->head (occ_round_window)
    if (!h->occ_fresh_slots)                              // a one-node window starts from cleared slots: nothing to validate
        hipLaunchKernelGGL(kv, dim3(count + h->occ_far), dim3(64), lds_val, h->stream, gv, ob, head, count, h->occ_far);
    bool lean_plan = false;
    if constexpr (MODE == MODE_AVX && T == 4) {
        hnsw_status ls = launch_occ_plan_lean(h, c, ob, head, count, &lean_plan);   // the specialised search routine where the index allows it
        if (ls != HNSW_OK) return ls;
    }
    if (!lean_plan)
        hipLaunchKernelGGL(kp, dim3(count), dim3(64), c.lds, h->stream, gv, ob, head, count, h->efc, h->m, c.lnb, c.lcap, h->d_spill,
                           h->spill_gnb, h->d_plan, h->select_shortcut ? 1u : 0u, h->occ_log_cap);
    hipLaunchKernelGGL(ks, dim3(count * kOccInsShr), dim3(64), c.lds, h->stream, gv, ob, head, count, h->m, c.lnb, c.lcap, h->d_spill,
                       h->spill_gnb, kEmpty, kOccInsShr);
    bool team = false;
    // (tie_mode 1: always -- a dry run can be thrown away when its selections meet a tie, an in-order commit cannot)
    // (a window of one node under tie_mode 1 -- a single hnsw_add -- too: the dry run tells whether the commit meets a tie
    // before a row is written, and writes the touched list)
    if (!lean_plan && h->tie_mode == 1)
        return fail(h, HNSW_ERR_INVALID, "tie_mode 1 needs the dim-128 plan kernels (their searches count the ties); use tie_mode 2");
    const bool lone_gated = count == 1 && h->tie_mode == 1;
    if (lone_gated || (count > 1 && !h->occ_want_touched && (h->commit_par == 2 || h->tie_mode == 1 || (h->commit_par == 1 && h->occ_yield * 10.0 >= (double)h->commit_par_min_x10)))) {
        // the window's commits in validated parallel groups, one workgroup per window node (hnsw_occ_par.hpp)
        HIP_TRY(h, hipGetLastError());
        // (the commit's grid is the FRONT; the plans above cover the whole depth of the window, add_exact_window)
        hnsw_status ps = occ_commit_par_r<MODE, T>(h, c, ob, h->occ_front ? std::min(h->occ_front, count) : count, end_node, &team);
        if (ps != HNSW_OK) return ps;
    }
    if (!team && h->tie_mode == 1)
        return fail(h, HNSW_ERR_INVALID, "tie_mode 1 needs the group commit kernel (ef_construction <= 1024, ids < 2^27, a device that holds the window's workgroups)");
    if (!team && h->commit_team) {
        HIP_TRY(h, hipGetLastError());
        hnsw_status ts = occ_commit_team_r<MODE, T>(h, c, ob, end_node, &team);
        if (ts != HNSW_OK) return ts;
    }
    if (!team)
        hipLaunchKernelGGL(kc, dim3(1), dim3(64), lds_commit, h->stream, gv, ob, end_node, h->m, c.lnb, c.lcap, h->d_spill_one,
                           h->spill_one_gnb, h->d_plan, occ_slack(h), h->occ_want_touched ? h->d_touched : nullptr,
                           h->occ_want_touched ? h->touched_cap : 0u, 0u, TeamCfg{}, (uint32_t *)nullptr, h->occ_chained ? 1u : 0u);
    HIP_TRY(h, hipGetLastError());
    return HNSW_OK;
}

// HNSW.NODE.DEL with its re-selections computed speculatively in parallel (k_occ_del_list / k_occ_shrinks / k_occ_del_commit)
template <int MODE, int T, int R>
static hnsw_status occ_delete_t(hnsw_index *h, const InsertCfg &c, const OccBufs &ob, uint32_t id)
{
    const size_t lds_commit = kOccScratchBytes + c.lds;
    auto kl = k_occ_del_list<MODE, T>;
    auto ks = k_occ_shrinks<MODE, T, R>;
    auto kd = k_occ_del_commit<MODE, T, R>;
    {
        static std::mutex mu;
        static bool attr_set[16] = {false};
        std::lock_guard<std::mutex> lock(mu);
        if (!attr_set[h->device & 15]) {
            HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void *>(ks), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 2048));
            HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void *>(kd), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 2048));
            attr_set[h->device & 15] = true;
        }
    }
    const GraphView gv = view_tag(h, c.tagcfg);
    hipLaunchKernelGGL(kl, dim3(1), dim3(64), 0, h->stream, gv, ob, id);
    hipLaunchKernelGGL(ks, dim3(kOccMaxShr), dim3(64), c.lds, h->stream, gv, ob, id, 1u, h->m, c.lnb, c.lcap, h->d_spill, h->spill_gnb, id,
                       kOccMaxShr);
    bool team = false;
    if (h->commit_team) {
        HIP_TRY(h, hipGetLastError());
        hnsw_status ts = occ_del_commit_team_r<MODE, T>(h, c, ob, id, &team);
        if (ts != HNSW_OK) return ts;
    }
    if (!team)
        hipLaunchKernelGGL(kd, dim3(1), dim3(64), lds_commit, h->stream, gv, ob, id, h->m, c.lnb, c.lcap, h->d_spill_one, h->spill_one_gnb,
                           h->d_touched, h->touched_cap, 0u, TeamCfg{}, (uint32_t *)nullptr);
    HIP_TRY(h, hipGetLastError());
    return HNSW_OK;
}

template <int MODE, int T>
hnsw_status occ_delete_r(hnsw_index *h, const InsertCfg &c, const OccBufs &ob, uint32_t id)
{
    switch (c.R) {
    case 1: return occ_delete_t<MODE, T, 1>(h, c, ob, id);
    case 4: return occ_delete_t<MODE, T, 4>(h, c, ob, id);
    case 8: return occ_delete_t<MODE, T, 8>(h, c, ob, id);
    case 16: return occ_delete_t<MODE, T, 16>(h, c, ob, id);
    case 64: return occ_delete_t<MODE, T, 64>(h, c, ob, id);
    }
    return fail(h, HNSW_ERR_INVALID, "bad R");
}

template <int MODE, int T>
hnsw_status occ_round_r(hnsw_index *h, const InsertCfg &c, const OccBufs &ob, uint32_t head, uint32_t count, uint32_t end_node)
{
    switch (c.R) {
    case 1: return occ_round_t<MODE, T, 1>(h, c, ob, head, count, end_node);
    case 4: return occ_round_t<MODE, T, 4>(h, c, ob, head, count, end_node);
    case 8: return occ_round_t<MODE, T, 8>(h, c, ob, head, count, end_node);
    case 16: return occ_round_t<MODE, T, 16>(h, c, ob, head, count, end_node);
    case 64: return occ_round_t<MODE, T, 64>(h, c, ob, head, count, end_node);
    }
    return fail(h, HNSW_ERR_INVALID, "bad R");
}

template hnsw_status occ_round_r<kVarMode, kVarT>(hnsw_index *, const InsertCfg &, const OccBufs &, uint32_t, uint32_t, uint32_t);
template hnsw_status occ_delete_r<kVarMode, kVarT>(hnsw_index *, const InsertCfg &, const OccBufs &, uint32_t);

} // namespace hnsw_host
