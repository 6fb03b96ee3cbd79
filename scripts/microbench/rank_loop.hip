// rank_loop.hip -- what one key of merge_rank costs a LONE wavefront, in shader clocks, for several formulations
// (scripts/phase_lean.py showed the loop at 39 % of an expansion on an otherwise empty chip).
//   hipcc --offload-arch=gfx950 -O2 -o rank_loop rank_loop.hip && ./rank_loop
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

__device__ __forceinline__ uint64_t readlane64(uint64_t v, int lane)
{
    uint32_t lo = __builtin_amdgcn_readlane((int)(uint32_t)v, lane);
    uint32_t hi = __builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), lane);
    return ((uint64_t)hi << 32) | lo;
}

constexpr int R = 4;

// A: the loop as the engine has it
__device__ __forceinline__ void rank_a(const uint64_t (&w)[R], uint64_t nk, bool take, uint32_t (&up)[R], uint32_t &mypos, int lane)
{
    uint32_t stay[R] = {0, 0, 0, 0};
    mypos = 0;
    const uint64_t mm0 = __ballot(take);
    uint64_t mm = mm0;
    while (mm) {
        const int j = __ffsll((unsigned long long)mm) - 1;
        mm &= mm - 1;
        const uint64_t s = readlane64(nk, j);
        uint32_t rank = (uint32_t)__popcll(__ballot(nk < s) & mm0);
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const bool below = w[r] < s;
            rank += __popcll(__ballot(below));
            stay[r] += below ? 1u : 0u;
        }
        mypos = lane == j ? rank : mypos;
    }
    const uint32_t n = (uint32_t)__popcll(mm0);
#pragma unroll
    for (int r = 0; r < R; ++r) up[r] = n - stay[r];
}

// B: compares grouped before the scalar work (a scheduling barrier keeps them apart)
__device__ __forceinline__ void rank_b(const uint64_t (&w)[R], uint64_t nk, bool take, uint32_t (&up)[R], uint32_t &mypos, int lane)
{
    uint32_t stay[R] = {0, 0, 0, 0};
    mypos = 0;
    const uint64_t mm0 = __ballot(take);
    uint64_t mm = mm0;
    while (mm) {
        const int j = __ffsll((unsigned long long)mm) - 1;
        mm &= mm - 1;
        const uint64_t s = readlane64(nk, j);
        uint64_t b[R];
        const uint64_t bn = __ballot(nk < s);
#pragma unroll
        for (int r = 0; r < R; ++r) b[r] = __ballot(w[r] < s);
        __builtin_amdgcn_sched_barrier(0);
        uint32_t rank = (uint32_t)__popcll(bn & mm0);
#pragma unroll
        for (int r = 0; r < R; ++r) {
            rank += __popcll(b[r]);
            stay[r] += (uint32_t)((b[r] >> lane) & 1ull);
        }
        mypos = lane == j ? rank : mypos;
    }
    const uint32_t n = (uint32_t)__popcll(mm0);
#pragma unroll
    for (int r = 0; r < R; ++r) up[r] = n - stay[r];
}

// C: pivots -- slice r holds entries [64r, 64r+63] in order, so only the slice the key falls into needs a lane compare
__device__ __forceinline__ void rank_c(const uint64_t (&w)[R], uint64_t nk, bool take, uint32_t (&up)[R], uint32_t &mypos, int lane)
{
    uint32_t stay[R] = {0, 0, 0, 0};
    uint32_t nfull[R] = {0, 0, 0, 0};
    mypos = 0;
    uint32_t myslice = 0;
#pragma unroll
    for (int r = 0; r < R - 1; ++r) myslice += (readlane64(w[r], 63) < nk) ? 1u : 0u;
    const uint64_t mm0 = __ballot(take);
    uint64_t mm = mm0;
    while (mm) {
        const int j = __ffsll((unsigned long long)mm) - 1;
        mm &= mm - 1;
        const uint64_t s = readlane64(nk, j);
        const uint32_t rs = (uint32_t)__builtin_amdgcn_readlane((int)myslice, j);
        uint32_t rank = rs * 64u + (uint32_t)__popcll(__ballot(nk < s) & mm0);
        switch (rs) {
        case 0: { const bool b = w[0] < s; rank += __popcll(__ballot(b)); stay[0] += b ? 1u : 0u; } break;
        case 1: { const bool b = w[1] < s; rank += __popcll(__ballot(b)); stay[1] += b ? 1u : 0u; } break;
        case 2: { const bool b = w[2] < s; rank += __popcll(__ballot(b)); stay[2] += b ? 1u : 0u; } break;
        default: { const bool b = w[3] < s; rank += __popcll(__ballot(b)); stay[3] += b ? 1u : 0u; } break;
        }
#pragma unroll
        for (int r = 0; r < R; ++r) nfull[r] += (uint32_t)r < rs ? 1u : 0u;
        mypos = lane == j ? rank : mypos;
    }
    const uint32_t n = (uint32_t)__popcll(mm0);
#pragma unroll
    for (int r = 0; r < R; ++r) up[r] = n - stay[r] - nfull[r];
}

// D: A with 32-bit compares (the distance word only; not exact on ties -- to see what the 64-bit compares cost)
__device__ __forceinline__ void rank_d(const uint64_t (&w)[R], uint64_t nk, bool take, uint32_t (&up)[R], uint32_t &mypos, int lane)
{
    uint32_t stay[R] = {0, 0, 0, 0};
    mypos = 0;
    const uint64_t mm0 = __ballot(take);
    uint64_t mm = mm0;
    const uint32_t nkh = (uint32_t)(nk >> 32);
    uint32_t wh[R];
#pragma unroll
    for (int r = 0; r < R; ++r) wh[r] = (uint32_t)(w[r] >> 32);
    while (mm) {
        const int j = __ffsll((unsigned long long)mm) - 1;
        mm &= mm - 1;
        const uint32_t s = (uint32_t)__builtin_amdgcn_readlane((int)nkh, j);
        uint32_t rank = (uint32_t)__popcll(__ballot(nkh < s) & mm0);
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const bool below = wh[r] < s;
            rank += __popcll(__ballot(below));
            stay[r] += below ? 1u : 0u;
        }
        mypos = lane == j ? rank : mypos;
    }
    const uint32_t n = (uint32_t)__popcll(mm0);
#pragma unroll
    for (int r = 0; r < R; ++r) up[r] = n - stay[r];
}

template <int VAR>
__global__ __launch_bounds__(64) void k(const uint64_t *W, const uint64_t *NK, const uint64_t *TAKE, int iters, unsigned long long *clk, uint32_t *sink)
{
    const int lane = threadIdx.x;
    uint64_t w[R];
    for (int r = 0; r < R; ++r) w[r] = W[r * 64 + lane];
    uint32_t acc = 0;
    unsigned long long total = 0;
    for (int it = 0; it < iters; ++it) {
        const uint64_t nk = NK[(it & 63) * 64 + lane];
        const bool take = (TAKE[it & 63] >> lane) & 1ull;
        uint32_t up[R], mypos;
        __builtin_amdgcn_s_waitcnt(0);
        const unsigned long long t0 = __builtin_readcyclecounter();
        if (VAR == 0) rank_a(w, nk, take, up, mypos, lane);
        else if (VAR == 1) rank_b(w, nk, take, up, mypos, lane);
        else if (VAR == 2) rank_c(w, nk, take, up, mypos, lane);
        else rank_d(w, nk, take, up, mypos, lane);
        acc += up[0] + up[1] * 3 + up[2] * 5 + up[3] * 7 + mypos * 11;
        __builtin_amdgcn_s_waitcnt(0);
        total += __builtin_readcyclecounter() - t0;
    }
    if (lane == 0 && blockIdx.x == 0) *clk = total;
    sink[blockIdx.x * 64 + lane] = acc;
}

int main()
{
    const int keys_per_call = 4, iters = 2000, blocks = 1024;
    std::vector<uint64_t> W(256), NK(64 * 64), TAKE(64);
    uint64_t x = 88172645463325252ull;
    auto rnd = [&] { x ^= x << 13; x ^= x >> 7; x ^= x << 17; return x; };
    for (int i = 0; i < 256; ++i) W[i] = i < 200 ? ((uint64_t)(0x3f000000u + i * 40000u) << 32) | (rnd() & 0xFFFFFFFEull) : ~0ull;
    for (int i = 0; i < 64 * 64; ++i) NK[i] = ((uint64_t)(0x3f000000u + (uint32_t)(rnd() % (200u * 40000u))) << 32) | (rnd() & 0xFFFFFFFEull);
    for (int i = 0; i < 64; ++i) { uint64_t m = 0; while (__builtin_popcountll(m) < keys_per_call) m |= 1ull << (rnd() % 40); TAKE[i] = m; }
    uint64_t *dW, *dN, *dT; unsigned long long *dc; uint32_t *ds;
    hipMalloc(&dW, 256 * 8); hipMalloc(&dN, 64 * 64 * 8); hipMalloc(&dT, 64 * 8); hipMalloc(&dc, 8); hipMalloc(&ds, blocks * 64 * 4);
    hipMemcpy(dW, W.data(), 256 * 8, hipMemcpyHostToDevice); hipMemcpy(dN, NK.data(), 64 * 64 * 8, hipMemcpyHostToDevice);
    hipMemcpy(dT, TAKE.data(), 64 * 8, hipMemcpyHostToDevice);
    std::vector<uint32_t> ref, got(blocks * 64);
    const char *names[4] = {"A current", "B compares grouped", "C pivots", "D 32-bit compares (inexact)"};
    for (int v = 0; v < 4; ++v) {
        for (int rep = 0; rep < 2; ++rep) {
            if (v == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(64), 0, 0, dW, dN, dT, iters, dc, ds);
            if (v == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(64), 0, 0, dW, dN, dT, iters, dc, ds);
            if (v == 2) hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(64), 0, 0, dW, dN, dT, iters, dc, ds);
            if (v == 3) hipLaunchKernelGGL(k<3>, dim3(blocks), dim3(64), 0, 0, dW, dN, dT, iters, dc, ds);
            hipDeviceSynchronize();
        }
        unsigned long long c = 0;
        hipMemcpy(&c, dc, 8, hipMemcpyDeviceToHost);
        hipMemcpy(got.data(), ds, blocks * 64 * 4, hipMemcpyDeviceToHost);
        if (v == 0) ref = got;
        printf("%-30s %7.0f clocks per call (%d keys) = %5.0f per key   %s\n", names[v], (double)c / iters, keys_per_call, (double)c / iters / keys_per_call,
               v == 3 ? "" : (got == ref ? "same result as A" : "RESULT DIFFERS"));
    }
    return 0;
}
