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
#include "extend_kernel.h"
#include "extend8_kernel.h"

#include <cstdlib>

// PT_TUNE_UNIFIED=1 selects the unified-fetch step (extend_kernel.h: every lane fetches what its `cur` points to, the
// wave waits once per iteration) instead of the vote-scheduled one.  Measured on MI355X, C5: 26 % fewer iterations and
// node-step lane occupancy 24 -> 28 of 64, extend 290 -> 311 ms and the overlapped shade 105 -> 83 ms per 4 frames: the
// same total (beyond L2 the kernel is bound by distinct 128-B lines per second, scripts/ubench/gather_rate.hip, not by
// the number of waits), so the established kernel stays the default.
static bool unified_step()
{
    static const bool on = getenv("PT_TUNE_UNIFIED") && atoi(getenv("PT_TUNE_UNIFIED")) == 1;
    return on;
}

// PT_TUNE_REC64=0 keeps the 48-B tri4 records in the leaf step (extend_kernel.h, REC64): the A/B switch of the 64-B records.
static bool rec64()
{
    static const bool on = !(getenv("PT_TUNE_REC64") && atoi(getenv("PT_TUNE_REC64")) == 0);
    return on;
}

const void *ptw_extend_hbm_fn(bool count)
{
    if (unified_step())
        return count ? reinterpret_cast<const void *>(k_extend<false, true, true, false, true>)
                     : reinterpret_cast<const void *>(k_extend<false, false, true, false, true>);
    if (rec64())
        return count ? reinterpret_cast<const void *>(k_extend<false, true, true, false, false, true>)
                     : reinterpret_cast<const void *>(k_extend<false, false, true, false, false, true>);
    return count ? reinterpret_cast<const void *>(k_extend<false, true, true>)
                 : reinterpret_cast<const void *>(k_extend<false, false, true>);
}

void ptw_launch_extend_hbm(bool count, int grid, size_t smem, hipStream_t st, hipEvent_t ev0, hipEvent_t ev1,
                           const float4 *wide, const uint2 *wide16, const float *norm_c, const float *norm_s,
                           const float *norm_rs, const float4 *tri4, const float4 *rec64_tab, uint32_t n_wide, uint32_t n_tris,
                           const float4 *rayA, const float2 *rayB, float4 *hit, const uint32_t *count_in, uint32_t *count_zero,
                           unsigned long long *stats, uint2 *spill, uint32_t spill_stride, int refill, float tmin,
                           float tmax, int lds_stack, int raw_hit, const uint32_t *perm, const float *ray_tmax)
{
    const NormBox nb = { norm_c[0], norm_c[1], norm_c[2], norm_s[0], norm_s[1], norm_s[2], norm_rs[0], norm_rs[1], norm_rs[2] };
#define PT_LAUNCH_HBM(C, U, R)                                                                                               \
    hipExtLaunchKernelGGL((k_extend<false, C, true, false, U, R>), dim3(grid), dim3(TB), (uint32_t)smem, st, ev0, ev1, 0u, wide, wide16, \
                          nb, tri4, n_wide, n_tris, rayA, rayB, hit, count_in, count_zero, stats, spill, spill_stride, refill, tmin, \
                          tmax, lds_stack, raw_hit, perm, ray_tmax, rec64_tab)
    if (unified_step()) {
        if (count) PT_LAUNCH_HBM(true, true, false); else PT_LAUNCH_HBM(false, true, false);
    } else if (rec64() && rec64_tab) {
        if (count) PT_LAUNCH_HBM(true, false, true); else PT_LAUNCH_HBM(false, false, true);
    } else {
        if (count) PT_LAUNCH_HBM(true, false, false); else PT_LAUNCH_HBM(false, false, false);
    }
#undef PT_LAUNCH_HBM
}

// ---- BVH8 kernel (extend8_kernel.h), same translation unit for the same scheduler --------------------------------
const void *ptw_extend8_fn(bool count)
{
    return count ? reinterpret_cast<const void *>(k_extend8<true>) : reinterpret_cast<const void *>(k_extend8<false>);
}

void ptw_launch_extend8(bool count, int grid, size_t smem, hipStream_t st, hipEvent_t ev0, hipEvent_t ev1, const uint4 *nodes8,
                        const float *norm_c, const float *norm_s, const float *norm_rs, const float4 *tri4, const float4 *rayA,
                        const float2 *rayB, float4 *hit, const uint32_t *count_in, uint32_t *count_zero, unsigned long long *stats,
                        uint2 *spill, uint32_t spill_stride, int refill, float tmin, float tmax, int lds_stack, int raw_hit,
                        const uint32_t *perm, const float *ray_tmax)
{
    const NormBox nb = { norm_c[0], norm_c[1], norm_c[2], norm_s[0], norm_s[1], norm_s[2], norm_rs[0], norm_rs[1], norm_rs[2] };
    if (count)
        hipExtLaunchKernelGGL((k_extend8<true>), dim3(grid), dim3(TB), (uint32_t)smem, st, ev0, ev1, 0u, nodes8, nb, tri4, rayA, rayB, hit,
                              count_in, count_zero, stats, spill, spill_stride, refill, tmin, tmax, lds_stack, raw_hit, perm, ray_tmax);
    else
        hipExtLaunchKernelGGL((k_extend8<false>), dim3(grid), dim3(TB), (uint32_t)smem, st, ev0, ev1, 0u, nodes8, nb, tri4, rayA, rayB, hit,
                              count_in, count_zero, stats, spill, spill_stride, refill, tmin, tmax, lds_stack, raw_hit, perm, ray_tmax);
}
