// What HBM delivers for the search kernel's access pattern: random rows of `row_bytes` contiguous
// bytes out of a large array, 8 lanes x 16 B per 128-byte block exactly as dist_rounds reads them,
// with as many loads in flight as registers allow and nothing else to do.
// Build: hipcc --offload-arch=gfx950 -O2 -o gather_bw gather_bw.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <random>

template <int T>   // T x 128 bytes per row
__global__ __launch_bounds__(256) void k_gather(const float4 *__restrict__ vec, const uint32_t *__restrict__ ids,
                                                uint32_t n_ids, float *__restrict__ out)
{
    const int lane = threadIdx.x & 63, grp = lane >> 3, pp = lane & 7;
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint32_t nwaves = (gridDim.x * blockDim.x) >> 6;
    float acc = 0.f;
    // 32 rows per step (4 rounds of 8), like one expansion
    for (uint32_t base = wave * 32; base + 32 <= n_ids; base += nwaves * 32) {
        float4 v[4][T];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const uint32_t id = ids[base + r * 8 + grp];
            const float4 *p = vec + (size_t)id * (T * 8) + pp;
#pragma unroll
            for (int t = 0; t < T; ++t) v[r][t] = p[t * 8];
        }
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int t = 0; t < T; ++t) acc += v[r][t].x + v[r][t].y + v[r][t].z + v[r][t].w;
    }
    if (acc == 12345.678f) out[0] = acc;
}

template <int T>
void run(const char *tag, size_t rows, int blocks, int threads)
{
    const size_t row_f4 = T * 8, n_ids = 1u << 24;
    float4 *vec;
    uint32_t *ids;
    float *out;
    if (hipMalloc(&vec, rows * row_f4 * sizeof(float4)) != hipSuccess) { printf("# %s: allocation failed\n", tag); return; }
    hipMemset(vec, 0, rows * row_f4 * sizeof(float4));
    hipMalloc(&ids, n_ids * 4);
    hipMalloc(&out, 4);
    std::vector<uint32_t> h(n_ids);
    std::mt19937_64 rng(1);
    for (auto &x : h) x = (uint32_t)(rng() % rows);
    hipMemcpy(ids, h.data(), n_ids * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(k_gather<T>, dim3(blocks), dim3(threads), 0, 0, vec, ids, (uint32_t)n_ids, out);
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_gather<T>, dim3(blocks), dim3(threads), 0, 0, vec, ids, (uint32_t)n_ids, out);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    const double bytes = (double)n_ids * T * 128;
    printf("gather row_bytes=%d rows=%zu array_mb=%.0f waves_per_simd=%d tbs=%.3f   # %s\n", T * 128, rows,
           rows * row_f4 * 16 / 1e6, blocks * threads / 64 / 1024, bytes / (best * 1e-3) / 1e12, tag);
    fflush(stdout);
    hipFree(vec); hipFree(ids); hipFree(out);
}

// the guide's "what a streaming read sustains": every wave reads contiguous 1 KB pieces of a large array, read-only
__global__ __launch_bounds__(256) void k_stream(const float4 *__restrict__ vec, size_t n_f4, float *__restrict__ out)
{
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_f4; i += (size_t)gridDim.x * blockDim.x) {
        const float4 v = vec[i];
        acc += v.x + v.y + v.z + v.w;
    }
    if (acc == 12345.678f) out[0] = acc;
}
void run_stream(size_t bytes)
{
    float4 *vec;
    float *out;
    if (hipMalloc(&vec, bytes) != hipSuccess) return;
    hipMemset(vec, 0, bytes);
    hipMalloc(&out, 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(k_stream, dim3(256 * 8), dim3(256), 0, 0, vec, bytes / 16, out);
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_stream, dim3(256 * 8), dim3(256), 0, 0, vec, bytes / 16, out);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    printf("stream_read array_mb=%.0f tbs=%.3f\n", bytes / 1e6, (double)bytes / (best * 1e-3) / 1e12);
    fflush(stdout);
    hipFree(vec); hipFree(out);
}

int main()
{
    printf("# scripts/microbench/gather_bw.hip: read-only random gathers of whole rows (8 lanes x 16 B per 128-byte block, 32 rows\n"
           "# in flight per wave, as the search kernel's distance rounds read them), 2^24 rows per launch, best of 3 launches;\n"
           "# tbs = row bytes x rows / time.  The Infinity Cache is 256 MB: arrays far larger than it are served by HBM.\n");
    run_stream((size_t)8 << 30);
    // 512-byte rows (dim 128): the search kernel's residency is 1 wave per SIMD for a lone launch, 2 with launches overlapped
    for (int wps : {1, 2, 4, 8}) run<4>("8 GB, far beyond the cache", (size_t)16 << 20, 256 * wps, 256);
    for (int wps : {2, 8}) run<4>("256 MB: fits the Infinity Cache", (size_t)500000, 256 * wps, 256);
    for (int wps : {1, 2, 8}) run<4>("C2's matrix: 1 M x 512 B", (size_t)1000000, 256 * wps, 256);
    for (int wps : {2, 8}) run<4>("C4's matrix: 10 M x 512 B", (size_t)10000000, 256 * wps, 256);
    // 3072-byte rows (dim 768)
    for (int wps : {1, 2, 4}) run<24>("8 GB, far beyond the cache", (size_t)2796202, 256 * wps, 256);
    for (int wps : {2, 4}) run<24>("256 MB: fits the Infinity Cache", (size_t)83333, 256 * wps, 256);
    for (int wps : {1, 2, 4}) run<24>("C3's matrix: 1 M x 3072 B", (size_t)1000000, 256 * wps, 256);
    return 0;
}
