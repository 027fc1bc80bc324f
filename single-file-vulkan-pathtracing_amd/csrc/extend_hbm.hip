// extend_hbm.hip -- the instantiations of k_extend that walk the scene out of L2 / MALL / HBM (scenes that do not
// fit LDS: BASELINE config C5, the 1M-triangle soup), in their own translation unit because they want a different
// instruction scheduler than the rest of the library.
//
// Built with -mllvm -amdgpu-sched-strategy=max-ilp (csrc/Makefile).  This kernel waits on memory for 40 % of its
// wave-cycles; the max-ILP strategy hoists the independent loads of a step (six plane loads + child words of a
// node, three vertex loads of a triangle) ahead of the arithmetic of the previous ones at the price of 6 VGPRs
// (70 -> 76, still six waves per SIMD, which LDS allows anyway).  Measured on MI355X, same box, A/B:
//     C5  1715 -> 1882 Mrays/s (+9.7 %), extend 829 -> 735 ms per 8 frames
// and the same flag on the LDS-resident Cornell kernel: -4 % (VALU-bound, the longer live ranges cost 4 spills
// at 72 VGPRs) -- hence two translation units rather than one flag.  Other strategies tried: max-memory-clause
// +6 % C5 / -7 % C2, iterative-ilp -1.5 % / -13 %, iterative-maxocc 0 / -6 %.
#define PT_EXTEND_TEMPLATES_ONLY  // (k_extend_lds7 & co. live in extend_launch.hip)
#include "extend_kernel.h"
#include "extend8_kernel.h"

// rec64 (pt_tuning.rec64 != 0, the default): the leaf step reads the 64-B per-triangle records k_shade gathers anyway
// (extend_kernel.h, REC64) instead of the 48-B tri4 records.
const void *ptw_extend_hbm_fn(bool count, bool rec64)
{
    if (rec64)
        return count ? reinterpret_cast<const void *>(k_extend<false, true, true, false, true>)
                     : reinterpret_cast<const void *>(k_extend<false, false, true, false, true>);
    return count ? reinterpret_cast<const void *>(k_extend<false, true, true>)
                 : reinterpret_cast<const void *>(k_extend<false, false, true>);
}

void ptw_launch_extend_hbm(bool count, bool rec64, int grid, size_t smem, hipStream_t st, hipEvent_t ev0, hipEvent_t ev1,
                           const float4 *wide, const uint2 *wide16, const float *norm_c, const float *norm_s,
                           const float *norm_rs, const float4 *tri4, const float4 *rec64_tab, uint32_t n_wide, uint32_t n_tris,
                           const float4 *rayA, const float2 *rayB, float4 *hit, const uint32_t *count_in, uint32_t *count_zero,
                           unsigned long long *stats, uint2 *spill, uint32_t spill_stride, int refill, float tmin,
                           float tmax, int lds_stack, int raw_hit, const uint32_t *perm, const float *ray_tmax)
{
    const NormBox nb = { norm_c[0], norm_c[1], norm_c[2], norm_s[0], norm_s[1], norm_s[2], norm_rs[0], norm_rs[1], norm_rs[2] };
#define PT_LAUNCH_HBM(C, R)                                                                                               \
    hipExtLaunchKernelGGL((k_extend<false, C, true, false, R>), dim3(grid), dim3(TB), (uint32_t)smem, st, ev0, ev1, 0u, wide, wide16, \
                          nb, tri4, n_wide, n_tris, rayA, rayB, hit, count_in, count_zero, stats, spill, spill_stride, refill, tmin, \
                          tmax, lds_stack, raw_hit, perm, ray_tmax, rec64_tab)
    if (rec64 && rec64_tab) {
        if (count) PT_LAUNCH_HBM(true, true); else PT_LAUNCH_HBM(false, true);
    } else {
        if (count) PT_LAUNCH_HBM(true, false); else PT_LAUNCH_HBM(false, false);
    }
#undef PT_LAUNCH_HBM
}

// ---- BVH8 kernel (extend8_kernel.h), same translation unit for the same scheduler --------------------------------
const void *ptw_extend8_fn(bool count, bool spills, bool waves7)
{
    if (spills) return count ? reinterpret_cast<const void *>(k_extend8<true, true, 6>) : reinterpret_cast<const void *>(k_extend8<false, true, 6>);
    if (count) return reinterpret_cast<const void *>(k_extend8<true, false, 6>);
    return waves7 ? reinterpret_cast<const void *>(k_extend8<false, false, 7>) : reinterpret_cast<const void *>(k_extend8<false, false, 6>);
}

void ptw_launch_extend8(bool count, bool spills, bool waves7, int grid, size_t smem, hipStream_t st, hipEvent_t ev0, hipEvent_t ev1, const uint4 *nodes8,
                        const float *norm_c, const float *norm_s, const float *norm_rs, const float4 *tri4, const float4 *rec64, const float4 *rayA,
                        const float2 *rayB, float4 *hit, const uint32_t *count_in, uint32_t *count_zero, unsigned long long *stats,
                        uint2 *spill, uint32_t spill_stride, int refill, float tmin, float tmax, int lds_stack, int raw_hit,
                        const uint32_t *perm, const float *ray_tmax)
{
    const NormBox nb = { norm_c[0], norm_c[1], norm_c[2], norm_s[0], norm_s[1], norm_s[2], norm_rs[0], norm_rs[1], norm_rs[2] };
#define PT_LAUNCH8(C, S, W)                                                                                                           \
    hipExtLaunchKernelGGL((k_extend8<C, S, W>), dim3(grid), dim3(TB), (uint32_t)smem, st, ev0, ev1, 0u, nodes8, nb, tri4, rec64, rayA, rayB, hit, \
                          count_in, count_zero, stats, spill, spill_stride, refill, tmin, tmax, lds_stack, raw_hit, perm, ray_tmax)
    if (spills) { if (count) PT_LAUNCH8(true, true, 6); else PT_LAUNCH8(false, true, 6); }
    else if (count) PT_LAUNCH8(true, false, 6);
    else if (waves7) PT_LAUNCH8(false, false, 7);
    else PT_LAUNCH8(false, false, 6);
#undef PT_LAUNCH8
}
