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
        if (OP == 12) { REP8(asm volatile("v_fmac_f32 %0, %1, %1\n v_fmac_f32 %2, %1, %1\n v_fmac_f32 %3, %1, %1\n v_fmac_f32 %4, %1, %1" : "+v"(r0), "+v"(a), "+v"(r1), "+v"(r2), "+v"(r3));) }
        if (OP == 13) { REP8(asm volatile("v_min_u32 %0, %0, %1\n v_max_u32 %2, %2, %1\n v_min_u32 %3, %3, %1\n v_max_u32 %4, %4, %1" : "+v"(r0), "+v"(a), "+v"(r1), "+v"(r2), "+v"(r3));) }
        if (OP == 14) { REP8(asm volatile("v_and_or_b32 %0, %0, %1, %1\n v_and_or_b32 %2, %2, %1, %1\n v_and_or_b32 %3, %3, %1, %1\n v_and_or_b32 %4, %4, %1, %1" : "+v"(r0), "+v"(a), "+v"(r1), "+v"(r2), "+v"(r3));) }
        if (OP == 15) { REP8(asm volatile("v_lshl_add_u32 %0, %0, 2, %1\n v_lshl_add_u32 %2, %2, 2, %1\n v_lshl_add_u32 %3, %3, 2, %1\n v_lshl_add_u32 %4, %4, 2, %1" : "+v"(r0), "+v"(a), "+v"(r1), "+v"(r2), "+v"(r3));) }
        if (OP == 16) { REP8(asm volatile("v_med3_f32 %0, %0, %1, %2\n v_med3_f32 %2, %2, %1, %3\n v_med3_f32 %3, %3, %1, %4\n v_med3_f32 %4, %4, %1, %0" : "+v"(r0), "+v"(a), "+v"(r1), "+v"(r2), "+v"(r3));) }
        if (OP == 17) { REP8(asm volatile("v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %2, %2, %1, vcc\n v_cndmask_b32 %3, %3, %1, vcc\n v_cndmask_b32 %4, %4, %1, vcc" : "+v"(r0), "+v"(a), "+v"(r1), "+v"(r2), "+v"(r3) : : "vcc");) }
        if (OP == 18) { REP8(asm volatile("v_mov_b32 %0, %1\n v_mov_b32 %2, %1\n v_mov_b32 %3, %1\n v_mov_b32 %4, %1" : "+v"(r0), "+v"(a), "+v"(r1), "+v"(r2), "+v"(r3));) }
        if (OP == 19) { REP8(asm volatile("v_bfi_b32 %0, %0, %1, %1\n v_bfi_b32 %2, %2, %1, %1\n v_bfi_b32 %3, %3, %1, %1\n v_bfi_b32 %4, %4, %1, %1" : "+v"(r0), "+v"(a), "+v"(r1), "+v"(r2), "+v"(r3));) }
        if (OP == 20) { REP8(asm volatile("v_mul_hi_u32 %0, %0, %1\n v_mul_hi_u32 %2, %2, %1\n v_mul_hi_u32 %3, %3, %1\n v_mul_hi_u32 %4, %4, %1" : "+v"(r0), "+v"(a), "+v"(r1), "+v"(r2), "+v"(r3));) }
        if (OP == 21) { REP8(asm volatile("v_sqrt_f32 %0, %0\n v_sqrt_f32 %2, %2\n v_sqrt_f32 %3, %3\n v_sqrt_f32 %4, %4" : "+v"(r0), "+v"(a), "+v"(r1), "+v"(r2), "+v"(r3));) }
        if (OP == 22) { REP8(asm volatile("v_div_fixup_f32 %0, %0, %1, %1\n v_div_fixup_f32 %2, %2, %1, %1\n v_div_fixup_f32 %3, %3, %1, %1\n v_div_fixup_f32 %4, %4, %1, %1" : "+v"(r0), "+v"(a), "+v"(r1), "+v"(r2), "+v"(r3));) }
        if (OP == 23) { REP8(asm volatile("v_sub_f32 %0, %0, %1\n v_sub_f32 %2, %2, %1\n v_sub_f32 %3, %3, %1\n v_sub_f32 %4, %4, %1" : "+v"(r0), "+v"(a), "+v"(r1), "+v"(r2), "+v"(r3));) }
        if (OP == 24) { REP8(asm volatile("v_add3_u32 %0, %0, %1, %1\n v_add3_u32 %2, %2, %1, %1\n v_add3_u32 %3, %3, %1, %1\n v_add3_u32 %4, %4, %1, %1" : "+v"(r0), "+v"(a), "+v"(r1), "+v"(r2), "+v"(r3));) }
        if (OP == 25) { REP8(asm volatile("v_cmp_lt_f32 s[10:11], %0, %1\n v_cmp_gt_f32 s[12:13], %2, %1\n v_cmp_lt_f32 s[14:15], %3, %1\n v_cmp_gt_f32 s[16:17], %4, %1" : "+v"(r0), "+v"(a), "+v"(r1), "+v"(r2), "+v"(r3) : : "s10","s11","s12","s13","s14","s15","s16","s17");) }
        if (OP == 26) { REP8(asm volatile("v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %2, %2, %2, %2\n v_fma_f32 %3, %3, %3, %3\n v_fma_f32 %4, %4, %4, %4" : "+v"(r0), "+v"(a), "+v"(r1), "+v"(r2), "+v"(r3));) }
        if (OP == 27) { REP8(asm volatile("v_mul_f32_e64 %0, %0, %1\n v_mul_f32_e64 %2, %2, %1\n v_mul_f32_e64 %3, %3, %1\n v_mul_f32_e64 %4, %4, %1" : "+v"(r0), "+v"(a), "+v"(r1), "+v"(r2), "+v"(r3));) }
        if (OP == 28) { REP8(asm volatile("v_add_u32 %0, %0, %1\n v_add_u32 %2, %2, %1\n v_add_u32 %3, %3, %1\n v_add_u32 %4, %4, %1" : "+v"(r0), "+v"(a), "+v"(r1), "+v"(r2), "+v"(r3));) }
        if (OP == 29) { REP8(asm volatile("v_perm_b32 %0, %0, %1, %1\n v_perm_b32 %2, %2, %1, %1\n v_perm_b32 %3, %3, %1, %1\n v_perm_b32 %4, %4, %1, %1" : "+v"(r0), "+v"(a), "+v"(r1), "+v"(r2), "+v"(r3));) }
        if (OP == 30) { REP8(asm volatile("v_cmp_lt_f32 vcc, %0, %1\n v_cndmask_b32 %2, %2, %1, vcc\n v_cmp_gt_f32 vcc, %3, %1\n v_cndmask_b32 %4, %4, %1, vcc" : "+v"(r0), "+v"(a), "+v"(r1), "+v"(r2), "+v"(r3) : : "vcc");) }
        if (OP == 31) { REP8(asm volatile("v_cmp_lt_f32 s[10:11], %0, %1\n v_cndmask_b32 %2, %2, %1, s[10:11]\n v_cmp_gt_f32 s[12:13], %3, %1\n v_cndmask_b32 %4, %4, %1, s[12:13]" : "+v"(r0), "+v"(a), "+v"(r1), "+v"(r2), "+v"(r3) : : "s10","s11","s12","s13");) }
        if (OP == 32) { REP8(asm volatile("v_cndmask_b32_e64 %0, %0, %1, vcc\n v_cndmask_b32_e64 %2, %2, %1, vcc\n v_cndmask_b32_e64 %3, %3, %1, vcc\n v_cndmask_b32_e64 %4, %4, %1, vcc" : "+v"(r0), "+v"(a), "+v"(r1), "+v"(r2), "+v"(r3) : : "vcc");) }
        if (OP == 33) { REP8(asm volatile("v_cmp_lt_f32 vcc, %0, %1\n v_cndmask_b32 %2, %2, %1, vcc\n v_cndmask_b32 %3, %3, %1, vcc\n v_cndmask_b32 %4, %4, %1, vcc" : "+v"(r0), "+v"(a), "+v"(r1), "+v"(r2), "+v"(r3) : : "vcc");) }
        if (OP == 34) { REP8(asm volatile("v_cmp_lt_f32 vcc, %0, %2\n v_cndmask_b32 %0, %0, %2, vcc\n v_cndmask_b32 %2, %2, %1, vcc\n v_cndmask_b32 %3, %3, %4, vcc\n v_cndmask_b32 %4, %4, %1, vcc" : "+v"(r0), "+v"(a), "+v"(r1), "+v"(r2), "+v"(r3) : : "vcc");) }
        if (OP == 35) { REP8(asm volatile("v_cmp_lt_f32 s[10:11], %0, %2\n v_cndmask_b32 %0, %0, %2, s[10:11]\n v_cndmask_b32 %2, %2, %1, s[10:11]\n v_cndmask_b32 %3, %3, %4, s[10:11]\n v_cndmask_b32 %4, %4, %1, s[10:11]" : "+v"(r0), "+v"(a), "+v"(r1), "+v"(r2), "+v"(r3) : : "s10","s11");) }
        if (OP == 36) { REP8(asm volatile("v_cndmask_b32 %0, %0, %1, vcc\n v_add_f32 %2, %2, %1\n v_cndmask_b32 %3, %3, %1, vcc\n v_add_f32 %4, %4, %1" : "+v"(r0), "+v"(a), "+v"(r1), "+v"(r2), "+v"(r3) : : "vcc");) }
        if (OP == 37) { REP8(asm volatile("v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %2, %2, %1, s[10:11]\n v_cndmask_b32 %3, %3, %1, vcc\n v_cndmask_b32 %4, %4, %1, s[10:11]" : "+v"(r0), "+v"(a), "+v"(r1), "+v"(r2), "+v"(r3) : : "vcc","s10","s11");) }
        if (OP == 38) { REP8(asm volatile("v_cndmask_b32 %0, %0, %1, vcc\n s_nop 0\n v_cndmask_b32 %3, %3, %1, vcc\n s_nop 0" : "+v"(r0), "+v"(a), "+v"(r1), "+v"(r2), "+v"(r3) : : "vcc");) }
        if (OP == 39) { REP8(asm volatile("v_addc_co_u32 %0, vcc, %0, %1, vcc\n v_addc_co_u32 %2, vcc, %2, %1, vcc\n v_addc_co_u32 %3, vcc, %3, %1, vcc\n v_addc_co_u32 %4, vcc, %4, %1, vcc" : "+v"(r0), "+v"(a), "+v"(r1), "+v"(r2), "+v"(r3) : : "vcc");) }
        if (OP == 40) { REP8(asm volatile("v_cndmask_b32 %0, %2, %3, vcc\n v_cndmask_b32 %2, %3, %4, vcc\n v_cndmask_b32 %3, %4, %0, vcc\n v_cndmask_b32 %4, %0, %2, vcc" : "+v"(r0), "+v"(a), "+v"(r1), "+v"(r2), "+v"(r3) : : "vcc");) }
        if (OP == 41) { REP8(asm volatile("v_cndmask_b32 %0, %2, %3, s[10:11]\n v_cndmask_b32 %2, %3, %4, s[10:11]\n v_cndmask_b32 %3, %4, %0, s[10:11]\n v_cndmask_b32 %4, %0, %2, s[10:11]" : "+v"(r0), "+v"(a), "+v"(r1), "+v"(r2), "+v"(r3) : : "s10","s11");) }
        if (OP == 42) { REP8(asm volatile("v_fma_f32 %0, %5, %1, %6\n v_fma_f32 %2, %6, %1, %5\n v_fma_f32 %3, %5, %1, %6\n v_fma_f32 %4, %6, %1, %5" : "+v"(r0), "+v"(a), "+v"(r1), "+v"(r2), "+v"(r3) : "v"(r6), "v"(r7));) }
        if (OP == 43) { REP8(asm volatile("v_pk_fma_f32 %0, %5, %1, %6 op_sel_hi:[1,0,0]\n v_pk_fma_f32 %2, %6, %1, %5 op_sel:[0,1,1] op_sel_hi:[1,1,1]\n v_pk_fma_f32 %3, %5, %1, %6 op_sel_hi:[1,0,0]\n v_pk_fma_f32 %4, %6, %1, %5 op_sel:[0,1,1] op_sel_hi:[1,1,1]" : "+v"(p0), "+v"(pa), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(p3), "v"(p2));) }
        if (OP == 44) { REP8(asm volatile("v_max3_f32 %0, %5, %1, %6\n v_min3_f32 %2, %6, %1, %5\n v_max3_f32 %3, %5, %1, %6\n v_min3_f32 %4, %6, %1, %5" : "+v"(r0), "+v"(a), "+v"(r1), "+v"(r2), "+v"(r3) : "v"(r6), "v"(r7));) }
        if (OP == 45) { REP8(asm volatile("v_cvt_f32_ubyte0 %0, %1\n v_cvt_f32_ubyte1 %2, %1\n v_cvt_f32_ubyte2 %3, %1\n v_cvt_f32_ubyte3 %4, %1" : "+v"(r0), "+v"(a), "+v"(r1), "+v"(r2), "+v"(r3));) }
        if (OP == 46) { REP8(asm volatile("v_cvt_f32_u32 %0, %0\n v_cvt_f32_u32 %2, %2\n v_cvt_f32_u32 %3, %3\n v_cvt_f32_u32 %4, %4" : "+v"(r0), "+v"(a), "+v"(r1), "+v"(r2), "+v"(r3));) }
        if (OP == 47) { REP8(asm volatile("v_cvt_f32_ubyte0 %0, %1\n v_fma_f32 %2, %0, %1, %1\n v_cvt_f32_ubyte2 %3, %1\n v_fma_f32 %4, %3, %1, %1" : "+v"(r0), "+v"(a), "+v"(r1), "+v"(r2), "+v"(r3));) }
        if (OP == 48) { REP8(asm volatile("v_bfe_u32 %0, %1, 8, 8\n v_bfe_u32 %2, %1, 16, 8\n v_bfe_u32 %3, %1, 0, 8\n v_bfe_u32 %4, %1, 24, 8" : "+v"(r0), "+v"(a), "+v"(r1), "+v"(r2), "+v"(r3));) }
        if (OP == 49) { REP8(asm volatile("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n v_fma_mix_f32 %2, %1, %3, %4 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n v_fma_mix_f32 %3, %1, %4, %0 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n v_fma_mix_f32 %4, %1, %0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(r0), "+v"(a), "+v"(r1), "+v"(r2), "+v"(r3));) }
        // round 5: sub-dword addressing (SDWA) -- a byte of a register zero-extended as an operand of a VOP2 instruction, or the result written to one byte
        if (OP == 50) { REP8(asm volatile("v_or_b32_sdwa %0, %1, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0\n v_or_b32_sdwa %2, %1, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n v_or_b32_sdwa %3, %1, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2\n v_or_b32_sdwa %4, %1, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3" : "+v"(r0), "+v"(a), "+v"(r1), "+v"(r2), "+v"(r3));) }
        if (OP == 51) { REP8(asm volatile("v_mov_b32_sdwa %0, %1 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_0\n v_mov_b32_sdwa %2, %1 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_1\n v_mov_b32_sdwa %3, %1 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_2\n v_mov_b32_sdwa %4, %1 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_3" : "+v"(r0), "+v"(a), "+v"(r1), "+v"(r2), "+v"(r3));) }
        if (OP == 52) { REP8(asm volatile("v_or_b32_sdwa %0, %1, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0\n v_fma_f32 %2, %0, %1, %1\n v_or_b32_sdwa %3, %1, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2\n v_fma_f32 %4, %3, %1, %1" : "+v"(r0), "+v"(a), "+v"(r1), "+v"(r2), "+v"(r3));) }
        if (OP == 53) { REP8(asm volatile("v_cvt_f32_ubyte0_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1\n v_cvt_f32_ubyte0_sdwa %2, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2\n v_cvt_f32_ubyte0_sdwa %3, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3\n v_cvt_f32_ubyte0_sdwa %4, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0" : "+v"(r0), "+v"(a), "+v"(r1), "+v"(r2), "+v"(r3));) }
        if (OP == 54) { REP8(asm volatile("v_mul_f32_sdwa %0, %1, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:DWORD\n v_mul_f32_sdwa %2, %1, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:DWORD\n v_mul_f32_sdwa %3, %1, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:DWORD\n v_mul_f32_sdwa %4, %1, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:DWORD" : "+v"(r0), "+v"(a), "+v"(r1), "+v"(r2), "+v"(r3));) }
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
    run<12>("v_fmac_f32 (VOP2)", d); run<13>("v_min/max_u32", d); run<14>("v_and_or_b32", d); run<15>("v_lshl_add_u32", d); run<16>("v_med3_f32", d); run<17>("v_cndmask e32 vcc", d); run<18>("v_mov_b32", d); run<19>("v_bfi_b32", d); run<20>("v_mul_hi_u32", d); run<21>("v_sqrt_f32", d); run<22>("v_div_fixup_f32", d); run<23>("v_sub_f32", d); run<24>("v_add3_u32", d); run<25>("v_cmp e64 sgpr", d); run<26>("v_fma_f32 2 srcs", d); run<27>("v_mul_f32 e64", d); run<28>("v_add_u32", d); run<29>("v_perm_b32", d);
    run<30>("cmp vcc + cndmask vcc", d); run<31>("cmp sgpr + cndmask sgpr", d); run<32>("cndmask e64 vcc", d); run<33>("cmp vcc, 3x cndmask vcc", d);
    run<34>("cswap: cmp vcc + 4 cndmask e32 (x5/iter)", d); run<35>("cswap: cmp sgpr + 4 cndmask e64 (x5/iter)", d);
  run<36>("P1 cnd32vcc,add,cnd32vcc,add", d); run<37>("P2 cnd32vcc,cnd64sgpr alternating", d); run<38>("P3 cnd32vcc,s_nop alternating (2 valu/grp)", d); run<39>("P4 4x v_addc_co (vcc carry in/out)", d); run<40>("P5 4x cnd e32 vcc, distinct srcs", d); run<41>("P6 4x cnd e64 sgpr, distinct srcs", d);
  run<42>("v_fma_f32 3 distinct srcs", d); run<43>("v_pk_fma_f32 3 srcs, op_sel broadcast", d); run<44>("v_max3/min3 3 distinct srcs", d);
  run<45>("v_cvt_f32_ubyte0..3", d); run<46>("v_cvt_f32_u32", d); run<47>("cvt_ubyte + fma alternating", d); run<48>("v_bfe_u32", d); run<49>("v_fma_mix_f32 (f16 src0)", d);
    run<50>("v_or_b32_sdwa src1 BYTE_k", d); run<51>("v_mov_b32_sdwa dst BYTE_1 preserve", d); run<52>("or_sdwa + fma alternating", d); run<53>("v_cvt_f32_ubyte0_sdwa src BYTE_k", d); run<54>("v_mul_f32_sdwa dwords", d);
    printf("(the two cswap lines issue 5 instructions per group, not 4: multiply their figure by 4/5... i.e. cycles per GROUP = figure x 4)\n");
    return 0;
}
