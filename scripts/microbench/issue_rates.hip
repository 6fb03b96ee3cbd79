// Issue cost of the instructions the search loop leans on, one wavefront alone on a SIMD (gfx950).
// Build: hipcc --offload-arch=gfx950 -O2 -o issue_rates issue_rates.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define REP16(X) X X X X X X X X X X X X X X X X

template <int KIND>
__global__ __launch_bounds__(1024) void k(uint64_t *out, uint32_t iters, uint64_t seed)
{
    __shared__ uint4 lds[256];
    const int lane = threadIdx.x & 63;
    if (threadIdx.x < 64) { lds[lane] = make_uint4(lane, lane + 1, lane + 2, lane + 3);
    lds[lane + 64] = lds[lane]; lds[lane + 128] = lds[lane]; lds[lane + 192] = lds[lane]; }
    __syncthreads();
    uint64_t a = seed + lane, b = seed * 3 + 7 * lane;
    uint32_t a0 = (uint32_t)a, a1 = (uint32_t)(a >> 32), b0 = (uint32_t)b, b1 = (uint32_t)(b >> 32);
    uint32_t acc = 0, t0 = 0, t1 = 0;
    float f0 = lane, f1 = 1.5f;
    uint32_t addr = (lane * 16) & 4095;
    uint64_t m0 = 0;
    const unsigned long long c0 = __builtin_readcyclecounter();
    for (uint32_t i = 0; i < iters; ++i) {
        if constexpr (KIND == 0) { REP16(asm volatile("v_cmp_lt_u64 %0, %1, %2" : "=s"(m0) : "v"(a), "v"(b));) }
        if constexpr (KIND == 1) { REP16(asm volatile("v_cmp_lt_u32 %0, %1, %2" : "=s"(m0) : "v"(a0), "v"(b0));) }
        if constexpr (KIND == 2) { REP16(asm volatile("v_sub_co_u32 %0, vcc, %2, %3\n v_subb_co_u32 %1, vcc, %4, %5, vcc" : "=v"(t0), "=v"(t1) : "v"(a0), "v"(b0), "v"(a1), "v"(b1) : "vcc");) }
        if constexpr (KIND == 3) { REP16(asm volatile("v_mul_lo_u32 %0, %1, %2" : "=v"(t0) : "v"(a0), "v"(b0));) }
        if constexpr (KIND == 4) { REP16(asm volatile("v_pk_fma_f32 %0, %1, %1, %0" : "+v"(a) : "v"(b));) }
        if constexpr (KIND == 5) { REP16(asm volatile("v_add_f32_dpp %0, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "=v"(f0) : "v"(f1));) }
        if constexpr (KIND == 6) { REP16(asm volatile("v_readlane_b32 %0, %1, 3" : "=s"(t0) : "v"(a0));) }
        if constexpr (KIND == 7) { REP16(asm volatile("s_bcnt1_i32_b64 %0, %1" : "=s"(t0) : "s"(m0) : "scc");) }
        if constexpr (KIND == 8) { REP16(asm volatile("ds_bpermute_b32 %0, %1, %2\n s_waitcnt lgkmcnt(0)" : "=v"(t0) : "v"(addr), "v"(a0));) }
        if constexpr (KIND == 9) { REP16(asm volatile("ds_read_b128 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(*(uint4 *)&lds[0]) : "v"(addr));) }
        if constexpr (KIND == 10) { REP16(asm volatile("v_cmp_eq_u64 %0, %1, %2" : "=s"(m0) : "v"(a), "v"(b));) }
        if constexpr (KIND == 11) { REP16(asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(a) : "v"(a0), "v"(b0) : "vcc");) }
        if constexpr (KIND == 12) { REP16(asm volatile("v_lshl_add_u64 %0, %1, 2, %0" : "+v"(a) : "v"(b));) }
        if constexpr (KIND == 13) { REP16(asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(t0) : "v"(a0), "v"(b0) : "vcc");) }
        if constexpr (KIND == 14) { REP16(asm volatile("s_and_b64 %0, %1, exec" : "=s"(m0) : "s"(m0) : "scc");) }
        if constexpr (KIND == 15) { REP16(asm volatile("v_add_f32 %0, %1, %0" : "+v"(f0) : "v"(f1));) }
        if constexpr (KIND == 16) { REP16(asm volatile("ds_read_b64 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(a) : "v"(addr));) }
        if constexpr (KIND == 17) { REP16(asm volatile("v_pk_add_f32 %0, %1, %0" : "+v"(a) : "v"(b));) }
        acc += t0 + t1 + (uint32_t)m0;
    }
    const unsigned long long c1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) { out[0] = c1 - c0; out[1] = acc + a0 + (uint32_t)a + (uint32_t)f0; }
}

// cycles per instruction of one wave while 1, 2 or 4 waves share its SIMD (blocks of 64/256 = one
// wave per SIMD, 512 = two, 1024 = four)
template <int KIND>
void run(const char *name, uint64_t *d)
{
    const uint32_t iters = 4096;
    printf("%-34s", name);
    for (int threads : {64, 256, 512, 1024}) {
        hipLaunchKernelGGL(k<KIND>, dim3(1), dim3(threads), 0, 0, d, iters, 12345ull);
        hipLaunchKernelGGL(k<KIND>, dim3(1), dim3(threads), 0, 0, d, iters, 12345ull);
        uint64_t h[2];
        hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
        printf(" %6.1f", (double)h[0] / (iters * 16.0));
    }
    printf("   (waves per SIMD: 1, 1, 2, 4)\n");
}

int main()
{
    uint64_t *d;
    hipMalloc(&d, 16);
    run<0>("v_cmp_lt_u64", d);
    run<10>("v_cmp_eq_u64", d);
    run<1>("v_cmp_lt_u32", d);
    run<2>("v_sub_co + v_subb_co (pair)", d);
    run<3>("v_mul_lo_u32", d);
    run<11>("v_mad_u64_u32", d);
    run<12>("v_lshl_add_u64", d);
    run<13>("v_cndmask_b32", d);
    run<4>("v_pk_fma_f32 (dependent)", d);
    run<17>("v_pk_add_f32 (dependent)", d);
    run<15>("v_add_f32 (dependent)", d);
    run<5>("v_add_f32_dpp", d);
    run<6>("v_readlane_b32", d);
    run<7>("s_bcnt1_i32_b64", d);
    run<14>("s_and_b64", d);
    run<8>("ds_bpermute_b32 + wait", d);
    run<9>("ds_read_b128 + wait", d);
    run<16>("ds_read_b64 + wait", d);
    hipFree(d);
    return 0;
}
