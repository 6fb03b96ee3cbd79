// Dependent-load latency as the search sees it: every wave walks its own random chain through a
// large array (one 4-byte load per step, the next address depends on the value).  Reported per step,
// for one wave alone and for 1024 / 4096 waves at once, over arrays of the sizes the index uses.
// Build: hipcc --offload-arch=gfx950 -O2 -o chase_latency chase_latency.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <random>
#include <numeric>
#include <algorithm>

__global__ __launch_bounds__(64) void k_chase(const uint32_t *__restrict__ next, uint32_t stride_words, uint32_t steps,
                                              uint32_t n, unsigned long long *__restrict__ cycles, uint32_t *sink, uint32_t seed)
{
    // all lanes of the wave follow the same chain (like a row fetch: one dependent address per step)
    uint32_t at = ((blockIdx.x + seed) * 2654435761u) % n;
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();   // 100 MHz
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (uint32_t i = 0; i < steps; ++i) at = next[(size_t)at * stride_words];
    const unsigned long long t1 = __builtin_readcyclecounter();
    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) {
        atomicAdd(cycles, t1 - t0);
        atomicAdd(cycles + 1, r1 - r0);
        if (at == 0xFFFFFFFFu) *sink = at;
    }
}

void run(size_t n, uint32_t stride_bytes, int waves)
{
    const uint32_t sw = stride_bytes / 4, steps = 2000;
    std::vector<uint32_t> perm(n), host((size_t)n * sw, 0);
    std::iota(perm.begin(), perm.end(), 0u);
    std::mt19937 rng(7);
    std::shuffle(perm.begin(), perm.end(), rng);
    for (size_t i = 0; i < n; ++i) host[(size_t)perm[i] * sw] = perm[(i + 1) % n];   // one big cycle
    uint32_t *d, *sink;
    unsigned long long *cyc, h[2] = {0, 0};
    hipMalloc(&d, host.size() * 4);
    hipMemcpy(d, host.data(), host.size() * 4, hipMemcpyHostToDevice);
    hipMalloc(&cyc, 16);
    hipMalloc(&sink, 4);
    hipMemset(cyc, 0, 16);
    hipLaunchKernelGGL(k_chase, dim3(waves), dim3(64), 0, 0, d, sw, steps, (uint32_t)n, cyc, sink, 0u);
    hipMemset(cyc, 0, 16);
    hipLaunchKernelGGL(k_chase, dim3(waves), dim3(64), 0, 0, d, sw, steps, (uint32_t)n, cyc, sink, 777777u);
    hipMemcpy(h, cyc, 16, hipMemcpyDeviceToHost);
    printf("%8zu entries x %4u B (%6.0f MB), %5d waves: %7.0f counter ticks = %6.0f ns per dependent load\n", n, stride_bytes,
           n * (double)stride_bytes / 1e6, waves, (double)h[0] / waves / steps, (double)h[1] * 10.0 / waves / steps);
    hipFree(d); hipFree(cyc); hipFree(sink);
}

int main()
{
    for (int waves : {1, 1024, 4096}) {
        run(1000000, 256, waves);      // layer-0 rows of C2: 256 MB
        run(1000000, 512, waves);      // vectors of C2: 512 MB
    }
    run(10000000, 512, 1024);          // vectors of C4: 5 GB
    run(100000, 256, 1);               // 25 MB: fits the Infinity Cache
    run(4000, 256, 1);                 // 1 MB: fits L2
    return 0;
}
