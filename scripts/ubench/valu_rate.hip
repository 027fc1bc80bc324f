// Dev tool: wave64 VALU issue cost per instruction class on gfx950 (8 waves/SIMD, independent chains).
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP8(x) x x x x x x x x
template <int OP>
__global__ __launch_bounds__(256) void k(float *out, int iters, float a, float b)
{
    float r0 = threadIdx.x, r1 = r0 + 1, r2 = r0 + 2, r3 = r0 + 3, r4 = r0 + 4, r5 = r0 + 5, r6 = r0 + 6, r7 = r0 + 7;
    float2 p0 = {r0, r1}, p1 = {r2, r3}, p2 = {r4, r5}, p3 = {r6, r7}, pa = {a, b};
    unsigned long long m = __ballot(threadIdx.x & 1);
    for (int i = 0; i < iters; i++) {
        if (OP == 0) { REP8(asm volatile("v_add_f32 %0, %0, %1\n v_add_f32 %2, %2, %1\n v_add_f32 %3, %3, %1\n v_add_f32 %4, %4, %1" : "+v"(r0), "+v"(a), "+v"(r1), "+v"(r2), "+v"(r3));) }
        if (OP == 1) { REP8(asm volatile("v_fma_f32 %0, %0, %1, %1\n v_fma_f32 %2, %2, %1, %1\n v_fma_f32 %3, %3, %1, %1\n v_fma_f32 %4, %4, %1, %1" : "+v"(r0), "+v"(a), "+v"(r1), "+v"(r2), "+v"(r3));) }
        if (OP == 2) { REP8(asm volatile("v_pk_add_f32 %0, %0, %1\n v_pk_add_f32 %2, %2, %1\n v_pk_add_f32 %3, %3, %1\n v_pk_add_f32 %4, %4, %1" : "+v"(p0), "+v"(pa), "+v"(p1), "+v"(p2), "+v"(p3));) }
        if (OP == 3) { REP8(asm volatile("v_pk_fma_f32 %0, %0, %1, %1\n v_pk_fma_f32 %2, %2, %1, %1\n v_pk_fma_f32 %3, %3, %1, %1\n v_pk_fma_f32 %4, %4, %1, %1" : "+v"(p0), "+v"(pa), "+v"(p1), "+v"(p2), "+v"(p3));) }
        if (OP == 4) { REP8(asm volatile("v_cndmask_b32 %0, %0, %1, %5\n v_cndmask_b32 %2, %2, %1, %5\n v_cndmask_b32 %3, %3, %1, %5\n v_cndmask_b32 %4, %4, %1, %5" : "+v"(r0), "+v"(a), "+v"(r1), "+v"(r2), "+v"(r3) : "s"(m));) }
        if (OP == 5) { REP8(asm volatile("v_min_f32 %0, %0, %1\n v_max_f32 %2, %2, %1\n v_min_f32 %3, %3, %1\n v_max_f32 %4, %4, %1" : "+v"(r0), "+v"(a), "+v"(r1), "+v"(r2), "+v"(r3));) }
        if (OP == 6) { REP8(asm volatile("v_mul_f32 %0, %0, %1\n v_mul_f32 %2, %2, %1\n v_mul_f32 %3, %3, %1\n v_mul_f32 %4, %4, %1" : "+v"(r0), "+v"(a), "+v"(r1), "+v"(r2), "+v"(r3));) }
        if (OP == 7) { REP8(asm volatile("v_rcp_f32 %0, %0\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n v_rcp_f32 %4, %4" : "+v"(r0), "+v"(a), "+v"(r1), "+v"(r2), "+v"(r3));) }
        if (OP == 8) { REP8(asm volatile("v_min3_f32 %0, %0, %1, %2\n v_max3_f32 %2, %2, %1, %3\n v_min3_f32 %3, %3, %1, %4\n v_max3_f32 %4, %4, %1, %0" : "+v"(r0), "+v"(a), "+v"(r1), "+v"(r2), "+v"(r3));) }
        if (OP == 9) { REP8(asm volatile("v_mul_lo_u32 %0, %0, %1\n v_mul_lo_u32 %2, %2, %1\n v_mul_lo_u32 %3, %3, %1\n v_mul_lo_u32 %4, %4, %1" : "+v"(r0), "+v"(a), "+v"(r1), "+v"(r2), "+v"(r3));) }
        if (OP == 10) { REP8(asm volatile("v_cmp_lt_f32 vcc, %0, %1\n v_cmp_gt_f32 vcc, %2, %1\n v_cmp_lt_f32 vcc, %3, %1\n v_cmp_gt_f32 vcc, %4, %1" : "+v"(r0), "+v"(a), "+v"(r1), "+v"(r2), "+v"(r3) : : "vcc");) }
        if (OP == 11) { REP8(asm volatile("v_add_u32 %0, %0, %1\n v_xor_b32 %2, %2, %1\n v_lshrrev_b32 %3, 3, %3\n v_and_b32 %4, %4, %1" : "+v"(r0), "+v"(a), "+v"(r1), "+v"(r2), "+v"(r3));) }
    }
    out[blockIdx.x * 256 + threadIdx.x] = r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y + a;
}
template <int OP> void run(const char *name, float *d)
{
    const int iters = 4096, grid = 256 * 8;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    k<OP><<<grid, 256>>>(d, 16, 1.0001f, 0.5f);
    hipDeviceSynchronize();
    hipEventRecord(a);
    k<OP><<<grid, 256>>>(d, iters, 1.0001f, 0.5f);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double insts_per_simd = (double)iters * 32 * 8 /*waves per SIMD*/;   // 32 instrs per iter per wave
    const double cyc = ms * 1e-3 * 2.4e9;
    printf("%-14s %8.3f ms  -> %.2f cycles/wave-instr (at 2.4 GHz nominal)\n", name, ms, cyc / insts_per_simd);
}
int main()
{
    float *d; hipMalloc(&d, 256 * 8 * 256 * 4);
    run<0>("v_add_f32", d); run<1>("v_fma_f32", d); run<6>("v_mul_f32", d); run<2>("v_pk_add_f32", d); run<3>("v_pk_fma_f32", d);
    run<4>("v_cndmask", d); run<5>("v_min/max", d); run<8>("v_min3/max3", d); run<10>("v_cmp_f32", d); run<7>("v_rcp_f32", d);
    run<9>("v_mul_lo_u32", d); run<11>("int add/xor/shift", d);
    return 0;
}
