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
void run(size_t rows, int blocks, int threads)
{
    const size_t row_f4 = T * 8, n_ids = 1u << 24;
    float4 *vec;
    uint32_t *ids;
    float *out;
    hipMalloc(&vec, rows * row_f4 * sizeof(float4));
    hipMemset(vec, 0, rows * row_f4 * sizeof(float4));
    hipMalloc(&ids, n_ids * 4);
    hipMalloc(&out, 4);
    std::vector<uint32_t> h(n_ids);
    std::mt19937 rng(1);
    for (auto &x : h) x = rng() % rows;
    hipMemcpy(ids, h.data(), n_ids * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(k_gather<T>, dim3(blocks), dim3(threads), 0, 0, vec, ids, (uint32_t)n_ids, out);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_gather<T>, dim3(blocks), dim3(threads), 0, 0, vec, ids, (uint32_t)n_ids, out);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double bytes = (double)n_ids * T * 128;
    printf("row %4d B, %7zu rows (%5.0f MB), %4d blocks x %3d threads: %6.2f TB/s\n", T * 128, rows,
           rows * row_f4 * 16 / 1e6, blocks, threads, bytes / (ms * 1e-3) / 1e12);
    hipFree(vec); hipFree(ids); hipFree(out);
}

int main()
{
    // the search kernel's residency: 1024 waves (one per SIMD); then 2, 4 and 8 per SIMD
    for (int wps : {1, 2, 4, 8}) {
        run<4>(1000000, 256 * wps, 256);      // C2: 1 M x 512 B
    }
    run<4>(10000000, 256 * 8, 256);           // C4: 10 M x 512 B
    run<24>(1000000, 256 * 2, 256);           // C3: 1 M x 3 KB
    run<24>(1000000, 256 * 4, 256);
    return 0;
}
