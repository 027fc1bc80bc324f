// Dev tool: what scalar instructions cost a VALU-bound wave64 kernel on gfx950 (8 waves/SIMD, independent chains).
// Question behind it (round 6): k_fused issues 0.44 scalar instructions per VALU instruction (exec-mask bookkeeping of its divergent blocks).
// Do they take issue slots from the VALU stream -- is the scalar unit one per CU or one per SIMD, and does a scalar instruction between two
// VALU instructions delay the second?
//   hipcc --offload-arch=gfx950 -O3 scripts/ubench/salu_rate.hip -o scripts/ubench/bin/salu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP8(x) x x x x x x x x
// every OP issues 4 VALU (v_add_f32, independent) + NS scalar instructions per group; OP 0: VALU only; OP 9: scalar only (4 per group)
template <int OP>
__global__ __launch_bounds__(256) void k(float *out, int iters, float a)
{
    float r0 = threadIdx.x, r1 = r0 + 1, r2 = r0 + 2, r3 = r0 + 3;
    for (int i = 0; i < iters; i++) {
        if (OP == 0) { REP8(asm volatile("v_add_f32 %0, %0, %1\n v_add_f32 %2, %2, %1\n v_add_f32 %3, %3, %1\n v_add_f32 %4, %4, %1" : "+v"(r0), "+v"(a), "+v"(r1), "+v"(r2), "+v"(r3));) }
        if (OP == 1) { REP8(asm volatile("v_add_f32 %0, %0, %1\n s_add_u32 s10, s10, s11\n v_add_f32 %2, %2, %1\n v_add_f32 %3, %3, %1\n v_add_f32 %4, %4, %1" : "+v"(r0), "+v"(a), "+v"(r1), "+v"(r2), "+v"(r3) : : "s10", "s11", "scc");) }
        if (OP == 2) { REP8(asm volatile("v_add_f32 %0, %0, %1\n s_add_u32 s10, s10, s11\n v_add_f32 %2, %2, %1\n s_and_b64 s[12:13], s[12:13], s[14:15]\n v_add_f32 %3, %3, %1\n v_add_f32 %4, %4, %1" : "+v"(r0), "+v"(a), "+v"(r1), "+v"(r2), "+v"(r3) : : "s10", "s11", "s12", "s13", "s14", "s15", "scc");) }
        if (OP == 4) { REP8(asm volatile("v_add_f32 %0, %0, %1\n s_add_u32 s10, s10, s11\n v_add_f32 %2, %2, %1\n s_and_b64 s[12:13], s[12:13], s[14:15]\n v_add_f32 %3, %3, %1\n s_or_b64 s[16:17], s[16:17], s[14:15]\n v_add_f32 %4, %4, %1\n s_andn2_b64 s[18:19], s[18:19], s[14:15]" : "+v"(r0), "+v"(a), "+v"(r1), "+v"(r2), "+v"(r3) : : "s10", "s11", "s12", "s13", "s14", "s15", "s16", "s17", "s18", "s19", "scc");) }
        if (OP == 8) { REP8(asm volatile("v_add_f32 %0, %0, %1\n s_add_u32 s10, s10, s11\n s_and_b64 s[12:13], s[12:13], s[14:15]\n v_add_f32 %2, %2, %1\n s_or_b64 s[16:17], s[16:17], s[14:15]\n s_andn2_b64 s[18:19], s[18:19], s[14:15]\n v_add_f32 %3, %3, %1\n s_add_u32 s10, s10, s11\n s_and_b64 s[12:13], s[12:13], s[14:15]\n v_add_f32 %4, %4, %1\n s_or_b64 s[16:17], s[16:17], s[14:15]\n s_andn2_b64 s[18:19], s[18:19], s[14:15]" : "+v"(r0), "+v"(a), "+v"(r1), "+v"(r2), "+v"(r3) : : "s10", "s11", "s12", "s13", "s14", "s15", "s16", "s17", "s18", "s19", "scc");) }
        if (OP == 9) { REP8(asm volatile("s_add_u32 s10, s10, s11\n s_and_b64 s[12:13], s[12:13], s[14:15]\n s_or_b64 s[16:17], s[16:17], s[14:15]\n s_andn2_b64 s[18:19], s[18:19], s[14:15]" : : : "s10", "s11", "s12", "s13", "s14", "s15", "s16", "s17", "s18", "s19", "scc");) }
        // the exec-mask idiom of a divergent block: save + and, (4 VALU), restore
        if (OP == 10) { REP8(asm volatile("s_and_saveexec_b64 s[12:13], s[14:15]\n v_add_f32 %0, %0, %1\n v_add_f32 %2, %2, %1\n v_add_f32 %3, %3, %1\n v_add_f32 %4, %4, %1\n s_or_b64 exec, exec, s[12:13]" : "+v"(r0), "+v"(a), "+v"(r1), "+v"(r2), "+v"(r3) : : "s12", "s13", "s14", "s15", "scc");) }
        // a compare that writes a lane mask, the mask combined by the scalar unit, a select that reads the result (the dependent chain VALU -> SALU -> VALU)
        if (OP == 11) { REP8(asm volatile("v_cmp_lt_f32 s[12:13], %0, %1\n s_and_b64 s[16:17], s[12:13], s[14:15]\n v_cndmask_b32 %2, %2, %1, s[16:17]\n v_add_f32 %3, %3, %1\n v_add_f32 %4, %4, %1" : "+v"(r0), "+v"(a), "+v"(r1), "+v"(r2), "+v"(r3) : : "s12", "s13", "s14", "s15", "s16", "s17", "scc");) }
        if (OP == 12) { REP8(asm volatile("v_cmp_lt_f32 s[12:13], %0, %1\n v_cndmask_b32 %2, %2, %1, s[12:13]\n v_add_f32 %3, %3, %1\n v_add_f32 %4, %4, %1" : "+v"(r0), "+v"(a), "+v"(r1), "+v"(r2), "+v"(r3) : : "s12", "s13", "s14", "s15", "s16", "s17", "scc");) }
        // a taken / not-taken scalar branch per group
        if (OP == 13) { REP8(asm volatile("v_add_f32 %0, %0, %1\n v_add_f32 %2, %2, %1\n s_cbranch_execz 1f\n v_add_f32 %3, %3, %1\n v_add_f32 %4, %4, %1\n1:" : "+v"(r0), "+v"(a), "+v"(r1), "+v"(r2), "+v"(r3));) }
        if (OP == 14) { REP8(asm volatile("v_add_f32 %0, %0, %1\n v_add_f32 %2, %2, %1\n s_cbranch_execnz 1f\n s_nop 0\n1:\n v_add_f32 %3, %3, %1\n v_add_f32 %4, %4, %1" : "+v"(r0), "+v"(a), "+v"(r1), "+v"(r2), "+v"(r3));) }
    }
    out[blockIdx.x * 256 + threadIdx.x] = r0 + r1 + r2 + r3 + a;
}
template <int OP> void run(const char *name, float *d, int n_valu, int n_other)
{
    const int iters = 4096, grid = 256 * 8;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    k<OP><<<grid, 256>>>(d, 16, 1.0001f);
    hipDeviceSynchronize();
    hipEventRecord(a);
    k<OP><<<grid, 256>>>(d, iters, 1.0001f);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double groups_per_simd = (double)iters * 8 * 8 /*waves per SIMD*/;
    const double cyc = ms * 1e-3 * 2.4e9;
    printf("%-58s %8.3f ms  -> %6.2f cycles per group of %d VALU + %d other per SIMD (at 2.4 GHz nominal)\n", name, ms, cyc / groups_per_simd, n_valu, n_other);
}
int main()
{
    float *d; hipMalloc(&d, 256 * 8 * 256 * 4);
    run<0>("4 v_add_f32", d, 4, 0);
    run<1>("4 v_add_f32 + 1 s_add", d, 4, 1);
    run<2>("4 v_add_f32 + 2 scalar", d, 4, 2);
    run<4>("4 v_add_f32 + 4 scalar (alternating)", d, 4, 4);
    run<8>("4 v_add_f32 + 8 scalar (two between)", d, 4, 8);
    run<9>("4 scalar only", d, 0, 4);
    run<10>("s_and_saveexec, 4 v_add_f32, s_or exec", d, 4, 2);
    run<11>("v_cmp -> s_and -> v_cndmask, 2 v_add_f32", d, 4, 1);
    run<12>("v_cmp -> v_cndmask, 2 v_add_f32", d, 4, 0);
    run<13>("4 v_add_f32 with a not-taken s_cbranch_execz in the middle", d, 4, 1);
    run<14>("4 v_add_f32 with a taken s_cbranch_execnz in the middle", d, 4, 1);
    return 0;
}
