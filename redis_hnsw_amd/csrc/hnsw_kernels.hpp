// hnsw_kernels.hpp -- __global__ kernels of the MI355X HNSW engine (gfx950).
#pragma once
#include "hnsw_device.hpp"

namespace hnsw {

// ---------------------------------------------------------------------------
// HNSW.SEARCH (core.rs:477-486 -> search_knn_internal :865-892), one wave per
// query, grid-stride over the batch.
// ---------------------------------------------------------------------------
template <int MODE, int T, int R, int FMT = FMT_F32>
__global__ __launch_bounds__(64) void k_search(GraphView g, const float *__restrict__ Q, uint32_t B,
                                               uint32_t k, uint32_t ef, uint32_t lnb, uint32_t lcap,
                                               uint32_t *__restrict__ gspill, uint32_t gnb,
                                               uint32_t *__restrict__ out_ids, float *__restrict__ out_sims,
                                               uint32_t *__restrict__ out_n, uint32_t bounded)
{
    extern __shared__ __align__(16) unsigned char smem[];
    const int lane = threadIdx.x;
    WaveMem m;
    Visited vis;
    carve<R, T, false>(smem, g.dim, lnb, lcap, m, vis, g.tagcfg);
    vis.glob = gspill + (size_t)blockIdx.x * gnb * 8;
    vis.gnb = gnb;
    vis.glob_dirty = false;
    vis.spilled = false;
    vis.count = 0;
    vis.lossy = false;
    // bounded: a full LDS table stops recording instead of moving to HBM (register-resident W only:
    // search_level_v2 drops re-met members of W itself)
    vis.bounded = bounded != 0 && MODE == MODE_AVX;

    WorkCtr ctr = {};
#ifdef HNSW_PHASE_TIMERS
    const unsigned long long wave_t0 = __builtin_readcyclecounter();
    const unsigned long long real_t0 = __builtin_amdgcn_s_memrealtime();
#endif
    const int32_t ep0 = g.hdr->enterpoint;        // core.rs:866
    const uint32_t lmax = g.hdr->max_layer;       // core.rs:867

    for (uint32_t qi = blockIdx.x; qi < B; qi += gridDim.x) {
#ifdef HNSW_PHASE_TIMERS
        const unsigned long long q_t0 = __builtin_readcyclecounter();
        const uint32_t q_e0 = ctr.n_expand, q_d0 = ctr.n_dist;
#endif
        QReg<T> qr;
        load_query<MODE, T>(Q + (size_t)qi * g.dim, g.dim, qr, m.qlds, lane);
        bool fail = false;
        uint32_t ep = (uint32_t)ep0;
        for (uint32_t lc = lmax; lc >= 1 && !fail; --lc) {  // core.rs:870-874
            search_level<MODE, T, 1, FMT>(g, m, vis, qr, ep, 1, lc, ctr, lane, fail);
            ep = key_id(m.W[0]);                          // core.rs:872
            __syncthreads();
        }
        uint32_t nW = 0;
        if (!fail) nW = search_level<MODE, T, R, FMT>(g, m, vis, qr, ep, ef, 0, ctr, lane, fail); // core.rs:876
        if (fail) {
            nW = 0;
            if (lane == 0) atomicOr(&g.hdr->status, ST_VISITED_OVERFLOW);
        }
        // core.rs:878-890: nearest first, min(k, |W|) results; sim = -dist (metrics.rs:75)
        const uint32_t nres = nW < k ? nW : k;
        for (uint32_t i = lane; i < k; i += 64) {
            uint64_t key = i < nres ? m.W[i] : 0;
            out_ids[(size_t)qi * k + i] = i < nres ? key_id(key) : kEmpty;
            out_sims[(size_t)qi * k + i] = i < nres ? -key_dist(key) : -__builtin_inff();
        }
        if (lane == 0) out_n[qi] = fail ? kEmpty : nres;
#ifdef HNSW_PHASE_TIMERS
        // profiling builds only: per-query (kilo-cycles, expansions, distance evals) in the last result slots
        if (lane == 0 && k >= 4) {
            out_ids[(size_t)qi * k + k - 1] = (uint32_t)((__builtin_readcyclecounter() - q_t0) >> 10);
            out_ids[(size_t)qi * k + k - 2] = ctr.n_expand - q_e0;
            out_ids[(size_t)qi * k + k - 3] = ctr.n_dist - q_d0;
            out_ids[(size_t)qi * k + k - 4] = (uint32_t)__builtin_amdgcn_s_getreg(((1 - 1) << 11) | (0 << 6) | 20); // XCC_ID
        }
#endif
        __syncthreads();
    }
    // leave the HBM spill table clean for the next launch
    if (vis.glob_dirty) visited_clear(vis, lane);
    if (lane == 0) {
        atomicAdd(&g.hdr->ctr_search[0], (unsigned long long)ctr.n_dist);
        atomicAdd(&g.hdr->ctr_search[1], (unsigned long long)ctr.n_ids);
        atomicAdd(&g.hdr->ctr_search[2], (unsigned long long)ctr.n_expand);
#ifdef HNSW_PHASE_TIMERS
        for (int i = 0; i < 7; ++i) atomicAdd(&g.hdr->prof[i], ctr.ph[i]);
        atomicMax(&g.hdr->prof[7], __builtin_amdgcn_s_memrealtime() - real_t0); // longest wave (100 MHz ticks)
        (void)wave_t0;
#endif
    }
}

// ---------------------------------------------------------------------------
// the metric on its own (metrics.rs:14-23), for the known-answer tests
// ---------------------------------------------------------------------------
template <int MODE>
__global__ __launch_bounds__(64) void k_metric_pairs(const float *__restrict__ a, const float *__restrict__ b,
                                                     uint32_t n, uint32_t dim, float *__restrict__ sims)
{
    extern __shared__ __align__(16) unsigned char smem[];
    const int lane = threadIdx.x;
    WaveMem m;
    m.qlds = reinterpret_cast<float *>(smem);
    m.fresh = reinterpret_cast<uint32_t *>(smem + (((size_t)dim * 4 + 15) & ~(size_t)15));
    m.dsc = reinterpret_cast<float *>(m.fresh + 64);
    GraphView g = {};
    g.vec = b;
    g.dim = dim;
    QReg<0> qr;
    for (uint32_t i = blockIdx.x; i < n; i += gridDim.x) {
        __syncthreads();
        load_query<MODE, 0>(a + (size_t)i * dim, dim, qr, m.qlds, lane);
        if (lane == 0) m.fresh[0] = i;
        __syncthreads();
        compute_dists<MODE, 0>(g, qr, m, 1, lane);
        __syncthreads();
        if (lane == 0) sims[i] = -m.dsc[0];
    }
}

// same pair metric through the register-resident (T > 0) path the search uses
template <int T>
__global__ __launch_bounds__(64) void k_metric_pairs_reg(const float *__restrict__ a, const float *__restrict__ b,
                                                         uint32_t n, float *__restrict__ sims)
{
    __shared__ uint32_t fresh[64];
    __shared__ float dsc[64];
    const int lane = threadIdx.x;
    WaveMem m = {};
    m.fresh = fresh;
    m.dsc = dsc;
    GraphView g = {};
    g.vec = b;
    g.dim = T * 32;
    for (uint32_t base = blockIdx.x * 64; base < n; base += gridDim.x * 64) {
        // 64 pairs per pass share nothing: recompute the query registers per pair group
        for (uint32_t j = 0; j < 64 && base + j < n; ++j) {
            QReg<T> qr;
            load_query<MODE_AVX, T>(a + (size_t)(base + j) * g.dim, g.dim, qr, nullptr, lane);
            __syncthreads();
            // fill several slots with the same id so every 8-lane group position is exercised
            uint32_t nf = 1 + (j % 19);
            if ((uint32_t)lane < nf) fresh[lane] = base + j;
            __syncthreads();
            compute_dists<MODE_AVX, T>(g, qr, m, nf, lane);
            __syncthreads();
            if (lane == 0) sims[base + j] = -dsc[j % nf];
        }
    }
}

#ifdef HNSW_UTILITY_KERNELS   // plain (non-template) kernels: defined once, in hnsw_engine.hip
// ---------------------------------------------------------------------------
// bulk import: CSR rows -> fixed-stride rows
// ---------------------------------------------------------------------------
__global__ void k_import_rows(uint32_t *adj, uint32_t stride, const uint32_t *__restrict__ slot_of_node,
                              uint32_t layer_off, const uint32_t *__restrict__ levels, uint32_t layer,
                              const uint64_t *__restrict__ rp, const uint32_t *__restrict__ col, uint32_t n)
{
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    if (wave >= n) return;
    if (levels[wave] < layer) return;
    const uint32_t slot = slot_of_node ? slot_of_node[wave] : wave;
    if (slot_of_node && slot == kNoUpper) return;
    const uint64_t b = rp[wave], e = rp[wave + 1];
    uint32_t *row = adj + (size_t)(slot + layer_off) * stride;
    uint32_t cnt = (uint32_t)(e - b);
    if (cnt > stride - 1) cnt = stride - 1;
    if (lane == 0) row[0] = cnt;
    for (uint32_t i = lane; i < cnt; i += 64) row[1 + i] = col[b + i];
}

// re-stride an adjacency table (rows keep their content)
__global__ void k_restride(const uint32_t *__restrict__ src, uint32_t sstride, uint32_t *__restrict__ dst,
                           uint32_t dstride, uint64_t rows)
{
    const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    if (wave >= rows) return;
    const uint32_t *s = src + wave * sstride;
    uint32_t *d = dst + wave * dstride;
    const uint32_t n = sstride < dstride ? sstride : dstride;
    for (uint32_t i = lane; i < dstride; i += 64) d[i] = i < n ? s[i] : 0u;
}

// HNSW.NODE.DEL on a graph with one-directional links (fast build): k_delete_exact walks the deleted
// node's own rows, which finds every neighbour only if links are symmetric (core.rs:770-772 keeps the
// reference's so).  This pass removes what is left of `id` from every row of a table, keeping the stored
// order.  One wave per row.
// The rows it edits are reported (row index | table flag) so that the host can hand their owners to the caller's
// update_fn like every other touched node (the integration writes hnswnodet values through for touched ids only).
__global__ void k_purge_inbound(uint32_t *adj, uint32_t stride, uint64_t rows, uint32_t id, uint32_t *n_removed,
                                uint32_t *edited, uint32_t edited_cap, uint32_t flag)
{
    const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    if (wave >= rows) return;
    uint32_t *row = adj + wave * stride;
    uint32_t cnt = row[0];
    if (cnt > stride - 1) cnt = stride - 1;
    uint32_t kept = 0, removed = 0;
    for (uint32_t base = 0; base < cnt; base += 64) {
        const uint32_t i = base + lane;
        const uint32_t x = i < cnt ? row[1 + i] : kEmpty;
        const bool keep = i < cnt && x != id;
        const uint64_t kb = __ballot(keep);
        removed += (uint32_t)__popcll(__ballot(i < cnt && x == id));
        // in-place compaction is safe: destinations never pass the chunk being read
        if (keep && removed) row[1 + kept + __popcll(kb & ((1ull << lane) - 1ull))] = x;
        kept += (uint32_t)__popcll(kb);
    }
    if (removed && lane == 0) {
        row[0] = kept;
        atomicAdd(n_removed, removed);
        const uint32_t slot = atomicAdd(&edited[0], 1u);             // edited[0] = count, entries follow
        if (slot < edited_cap) edited[1 + slot] = (uint32_t)wave | flag;
    }
}

// number of links i -> j whose reverse j -> i is missing (0 for every graph the reference can produce)
__global__ void k_count_asymmetric(const uint32_t *__restrict__ adj, uint32_t stride,
                                   const uint32_t *__restrict__ slot_of_node, uint32_t layer_off,
                                   const uint32_t *__restrict__ levels, uint32_t layer, uint32_t n,
                                   unsigned long long *__restrict__ out)
{
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    if (wave >= n || levels[wave] < layer) return;
    const uint32_t slot = slot_of_node ? slot_of_node[wave] : wave;
    if (slot_of_node && slot == kNoUpper) return;
    const uint32_t *row = adj + (size_t)(slot + layer_off) * stride;
    uint32_t cnt = row[0];
    if (cnt > stride - 1) cnt = stride - 1;
    uint32_t bad = 0;
    for (uint32_t a = 0; a < cnt; ++a) {
        const uint32_t j = row[1 + a];
        bool found = false;
        if (j < n && levels[j] >= layer) {
            const uint32_t js = slot_of_node ? slot_of_node[j] : j;
            if (!(slot_of_node && js == kNoUpper)) {
                const uint32_t *jr = adj + (size_t)(js + layer_off) * stride;
                uint32_t jc = jr[0];
                if (jc > stride - 1) jc = stride - 1;
                for (uint32_t b = lane; b < jc; b += 64) found |= jr[1 + b] == wave;
            }
        }
        if (!__ballot(found)) ++bad;
    }
    if (bad && lane == 0) atomicAdd(out, (unsigned long long)bad);
}

// f32 -> bf16, round to nearest even (finite inputs: the host entry points refuse anything else)
__global__ void k_f32_to_bf16(const float *__restrict__ src, unsigned short *__restrict__ dst, size_t n)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const uint32_t u = __float_as_uint(src[i]);
        dst[i] = (unsigned short)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);
    }
}

// f32 -> fp8 (e4m3 as this part's conversion instructions define it; round to nearest even, magnitudes beyond the
// format's largest finite value saturate to it), two components per conversion
__global__ void k_f32_to_fp8(const float *__restrict__ src, unsigned char *__restrict__ dst, size_t n)
{
    size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 2;
    for (; i < n; i += (size_t)gridDim.x * blockDim.x * 2) {
        const float a = fminf(fmaxf(src[i], -448.f), 448.f);
        const float b = i + 1 < n ? fminf(fmaxf(src[i + 1], -448.f), 448.f) : 0.f;
        const int pk = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0, false);
        dst[i] = (unsigned char)(pk & 0xFF);
        if (i + 1 < n) dst[i + 1] = (unsigned char)((pk >> 8) & 0xFF);
    }
}

// one stored row widened back to f32 (what hnsw_get_vector reports for a compressed index): the same widening the
// search kernels apply
template <int FMT>
__global__ void k_decode_row(const float4 *__restrict__ vec4, size_t row, uint32_t dim, float *__restrict__ out)
{
    const uint32_t piece = blockIdx.x * blockDim.x + threadIdx.x;         // 4 components each
    if (piece * 4 >= dim) return;
    const float4 v = load_piece<FMT>(vec4, row * (dim >> 2) + piece);
    reinterpret_cast<float4 *>(out)[piece] = v;
}

// -inf fill (similarities of an empty result)
__global__ void k_fill_f32(float *p, size_t n, float v)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v;
}

// per-row degrees of one layer (export)
__global__ void k_degrees(const uint32_t *__restrict__ adj, uint32_t stride, const uint32_t *__restrict__ slot_of_node,
                          uint32_t layer_off, const uint32_t *__restrict__ levels, uint32_t layer, uint32_t n,
                          uint32_t *__restrict__ deg)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t d = 0;
    if (levels[i] >= layer) {
        uint32_t slot = slot_of_node ? slot_of_node[i] : i;
        if (!(slot_of_node && slot == kNoUpper)) d = adj[(size_t)(slot + layer_off) * stride];
    }
    deg[i] = d;
}

__global__ void k_export_rows(const uint32_t *__restrict__ adj, uint32_t stride,
                              const uint32_t *__restrict__ slot_of_node, uint32_t layer_off,
                              const uint32_t *__restrict__ levels, uint32_t layer, uint32_t n,
                              const uint64_t *__restrict__ rp, uint32_t *__restrict__ col)
{
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    if (wave >= n) return;
    if (levels[wave] < layer) return;
    uint32_t slot = slot_of_node ? slot_of_node[wave] : wave;
    if (slot_of_node && slot == kNoUpper) return;
    const uint32_t *row = adj + (size_t)(slot + layer_off) * stride;
    const uint32_t cnt = (uint32_t)(rp[wave + 1] - rp[wave]);
    for (uint32_t i = lane; i < cnt; i += 64) col[rp[wave] + i] = row[1 + i];
}

#endif // HNSW_UTILITY_KERNELS

} // namespace hnsw
