// hnsw_tu_std.hip -- HNSW.NODE.ADD / HNSW.SEARCH in the reference binary's own tie order (hnsw_std_heap.hpp: one wavefront per operation,
// std::collections::BinaryHeap restated), and their launchers.  Used by tuning "tie_mode" for the operations the tie
// census flags.
#define HNSW_SYNC_WAVE_FULL
#include "hnsw_host.hpp"
#include "hnsw_std_heap.hpp"
#include <cstring>
#include <mutex>

namespace hnsw_host {

static_assert(sizeof(hnsw_index::std_ctx0) == sizeof(StdScratch), "hnsw_index::std_ctx0 mirrors hnsw::StdScratch");

// scratch contexts: [0] the insert's (heaps as large as the index), [1 ..] the searches' (smaller heaps)
constexpr uint32_t kStdSearchCtx = 32;    // queries answered side by side (one wavefront each; 4 B x capacity of stamps per context)

hnsw_status ensure_std_scratch(hnsw_index *h)
{
    {   // the kernels' heaps take 102 KB of dynamic LDS
        static std::mutex mu;
        static bool attr_set[16] = {false};
        std::lock_guard<std::mutex> lock(mu);
        if (!attr_set[h->device & 15]) {
            HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void *>(k_insert_std_heap), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kStdLdsBytes));
            HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void *>(k_delete_std_heap), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kStdLdsBytes));
            HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void *>(k_search_std_heap), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kStdLdsBytes));
            attr_set[h->device & 15] = true;
        }
    }
    const uint32_t hcap_ins = std::min<uint32_t>(std::max(h->cap, 4096u), 1u << 21) + 1024u;
    const uint32_t hcap_q = std::min<uint32_t>(std::max(h->cap, 4096u), 1u << 17) + 1024u;
    if (h->d_std_stamp && h->std_cap >= h->cap && h->std_hcap >= hcap_ins) return HNSW_OK;
    HIP_TRY(h, hipDeviceSynchronize());
    (void)hipFree(h->d_std_stamp); (void)hipFree(h->d_std_heaps); (void)hipFree(h->d_std_ctx); (void)hipFree(h->d_std_misc);
    h->d_std_stamp = nullptr; h->d_std_heaps = nullptr; h->d_std_ctx = nullptr; h->d_std_misc = nullptr;
    const uint32_t nctx = 1 + kStdSearchCtx;
    const size_t stamp_words = (size_t)nctx * h->cap;
    const size_t heap_pairs = (size_t)10 * hcap_ins + (size_t)kStdSearchCtx * 10 * hcap_q;
    HIP_TRY(h, hipMalloc((void **)&h->d_std_stamp, stamp_words * 4));
    HIP_TRY(h, hipMemset(h->d_std_stamp, 0, stamp_words * 4));
    HIP_TRY(h, hipMalloc((void **)&h->d_std_heaps, heap_pairs * sizeof(StdPair)));
    HIP_TRY(h, hipMalloc((void **)&h->d_std_misc, (size_t)(nctx + 1) * 4));           // epochs [nctx], status [1]
    HIP_TRY(h, hipMemset(h->d_std_misc, 0, (size_t)(nctx + 1) * 4));
    std::vector<StdScratch> ctx(nctx);
    uint32_t *misc = reinterpret_cast<uint32_t *>(h->d_std_misc);
    StdPair *hp = reinterpret_cast<StdPair *>(h->d_std_heaps);
    for (uint32_t i = 0; i < nctx; ++i) {
        ctx[i].stamp = reinterpret_cast<uint32_t *>(h->d_std_stamp) + (size_t)i * h->cap;
        ctx[i].epoch = misc + i;
        ctx[i].status = misc + nctx;
        ctx[i].hcap = i == 0 ? hcap_ins : hcap_q;
        ctx[i].heaps = i == 0 ? hp : hp + (size_t)10 * hcap_ins + (size_t)(i - 1) * 10 * hcap_q;
    }
    HIP_TRY(h, hipMalloc((void **)&h->d_std_ctx, sizeof(StdScratch) * nctx));
    HIP_TRY(h, hipMemcpy(h->d_std_ctx, ctx.data(), sizeof(StdScratch) * nctx, hipMemcpyHostToDevice));
    std::memcpy(&h->std_ctx0, &ctx[0], sizeof(StdScratch));
    h->std_cap = h->cap;
    h->std_hcap = hcap_ins;
    return HNSW_OK;
}

// one insert (the node's vector, level and empty rows are in place), on the handle's stream
hnsw_status launch_insert_std(hnsw_index *h, uint32_t id, bool want_touched)
{
    hnsw_status s = ensure_std_scratch(h);
    if (s != HNSW_OK) return s;
    StdScratch sc;
    std::memcpy(&sc, &h->std_ctx0, sizeof sc);
    hipLaunchKernelGGL(k_insert_std_heap, dim3(1), dim3(64), kStdLdsBytes, h->stream, view(h), sc, id, h->m, h->efc, want_touched ? h->d_touched : nullptr,
                       want_touched ? h->touched_cap : 0u);
    HIP_TRY(h, hipGetLastError());
    return HNSW_OK;
}

// one delete (every neighbour re-selected with the node ignored), on the handle's stream; the caller keeps the tombstone
hnsw_status launch_delete_std(hnsw_index *h, uint32_t id)
{
    hnsw_status s = ensure_std_scratch(h);
    if (s != HNSW_OK) return s;
    StdScratch sc;
    std::memcpy(&sc, &h->std_ctx0, sizeof sc);
    hipLaunchKernelGGL(k_delete_std_heap, dim3(1), dim3(64), kStdLdsBytes, h->stream, view(h), sc, id, h->m, h->d_touched, h->touched_cap);
    HIP_TRY(h, hipGetLastError());
    return HNSW_OK;
}

// tie_mode after a search launch on `st`: the queries whose tie flag the census kernel set (all = every query: tie_mode 2, or
// a shape without a census kernel) are answered again in the reference binary's own tie order, results written over
// d_ids / d_sims / d_nout.  Asynchronous on `st`; launches of this kind share the scratch contexts and are chained by an event.
hnsw_status launch_search_std(hnsw_index *h, const float *dQ, uint32_t B, uint32_t k, uint32_t *d_ids, float *d_sims, uint32_t *d_nout, bool all,
                              hipStream_t st)
{
    hnsw_status s = ensure_std_scratch(h);
    if (s != HNSW_OK) return s;
    if (!h->std_ev) HIP_TRY(h, hipEventCreateWithFlags(&h->std_ev, hipEventDisableTiming));
    if (h->std_ev_valid) HIP_TRY(h, hipStreamWaitEvent(st, h->std_ev, 0));
    uint32_t *count = h->d_tie_flags + h->tie_flags_cap, *which = h->d_tie_flags + h->tie_flags_cap + 4;
    HIP_TRY(h, hipMemsetAsync(count, 0, 4, st));
    HIP_TRY(h, hipMemsetAsync(reinterpret_cast<uint32_t *>(h->d_std_misc) + 1 + kStdSearchCtx, 0, 4, st));    // (an insert reads its own at once)
    hipLaunchKernelGGL(k_tie_compact, dim3((B + 255) / 256), dim3(256), 0, st, h->d_tie_flags, B, which, count, all ? 1u : 0u);
    const StdScratch *ctx = reinterpret_cast<const StdScratch *>(h->d_std_ctx) + 1;
    hipLaunchKernelGGL(k_search_std_heap, dim3(kStdSearchCtx), dim3(64), kStdLdsBytes, st, view(h), ctx, dQ, which, count, k, h->efc, d_ids, d_sims, d_nout);
    HIP_TRY(h, hipGetLastError());
    HIP_TRY(h, hipEventRecord(h->std_ev, st));
    h->std_ev_valid = true;
    return HNSW_OK;
}

// per-query tie flags of a search launch (written by the census kernel), then the count and the list of flagged queries
hnsw_status ensure_tie_flags(hnsw_index *h, uint32_t B)
{
    if (h->d_tie_flags && h->tie_flags_cap >= B) return HNSW_OK;
    HIP_TRY(h, hipDeviceSynchronize());
    (void)hipFree(h->d_tie_flags);
    h->d_tie_flags = nullptr;
    const uint32_t cap = std::max(B, 1024u);
    HIP_TRY(h, hipMalloc((void **)&h->d_tie_flags, ((size_t)2 * cap + 4) * 4));
    h->tie_flags_cap = cap;
    return HNSW_OK;
}

// 1 if a heap of the std-order kernels overflowed since the last call (host-side check after a synchronisation)
hnsw_status std_status(hnsw_index *h, uint32_t *out)
{
    *out = 0;
    if (!h->d_std_misc) return HNSW_OK;
    uint32_t *misc = reinterpret_cast<uint32_t *>(h->d_std_misc);
    HIP_TRY(h, hipMemcpy(out, misc + 1 + kStdSearchCtx, 4, hipMemcpyDeviceToHost));
    if (*out) HIP_TRY(h, hipMemset(misc + 1 + kStdSearchCtx, 0, 4));
    return HNSW_OK;
}

} // namespace hnsw_host
