// fused.hip -- PT_PIPELINE_FUSED: the kernel (fused_kernel.h), the scenes it takes, its launch.
#include "wavefront_host.h"

#include <algorithm>

#define PT_EXTEND_TEMPLATES_ONLY
#include "extend_kernel.h"  // the LDS node / stack helpers the fused kernel shares with k_extend_lds7p

namespace {
using namespace ptw;
#include "fused_kernel.h"
}  // namespace

pt_status ptw_plan_fused(pt_scene *s, const ExtendPlan &pl, float tmin, FusedPlan &fp)
{
    pt_ctx *ctx = s->ctx;
    const size_t tables = sizeof(float4) * 5 * (size_t)s->n_tris;  // shade4 + tangent frames (the vertices are the kz = 2 triangle copy)
    if (s->n_inst || pl.variant != PT_EXTEND_LDS || pl.spill || !pl.pairs || !(tmin > 0.f) || tables > 16 * 1024) {
        ctx->err = "PT_PIPELINE_FUSED is for single-level scenes whose BVH4, triangles and shading tables fit LDS (the compact pair-leaf "
                   "kernel's class: <= 2047 triangles in <= 24 KB, stack bound <= 16, tmin > 0)";
        return PT_ERR_UNSUPPORTED;
    }
    fp.lds_stack = pl.lds_stack;
    fp.smem = (size_t)pl.lds_stack * FTB * sizeof(uint32_t) + (pl.smem - (size_t)pl.lds_stack * TB * sizeof(uint32_t)) + tables +
              sizeof(uint32_t) * FS_FIELDS * FTB + sizeof(uint32_t) * (FTB / 64) * PT_FUSED_WTILES;
    for (const void *fn : { reinterpret_cast<const void *>(k_fused<false>), reinterpret_cast<const void *>(k_fused<true>) })
        if (fp.smem > 48 * 1024) PT_HIP(ctx, hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)fp.smem));
    int per_cu = 0;
    PT_HIP(ctx, hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void *>(k_fused<false>), FTB, fp.smem));
    per_cu = std::max(1, std::min(per_cu, 8));
    per_cu = pt_tuned(ctx->tune.extend_blocks, per_cu, 1, per_cu);
    fp.grid = ctx->num_cus * per_cu;
    fp.block = FTB;
    // of 64: the share of a wave's live lanes that must wait with a finished ray before the shade block runs for them.  The block
    // is ~4x a node step, so it pays to run it fuller than k_extend's refill (16): 8 / 16 / 24 / 32 / 40 -> 33.9 / 34.1 / 34.6 /
    // 35.7 / 36.3 Grays/s on the Cornell box at 1080p (profiles/r04b_fused_refill_sweep.txt)
    fp.refill = pt_tuned(ctx->tune.refill, 40, 1, 64);
    return PT_OK;
}

void ptw_launch_fused(const FusedPlan &fp, bool grouped, const ptw::RenderConst &rc, const uint32_t *tiles, const ptw::Radiance &rad,
                      const pt_scene *s, uint32_t n_slots, uint32_t *next_slot, unsigned long long *stats, float tmin, float tmax,
                      hipStream_t st, hipEvent_t ev0, hipEvent_t ev1)
{
    if (grouped)
        hipExtLaunchKernelGGL((k_fused<true>), dim3(fp.grid), dim3(FTB), (uint32_t)fp.smem, st, ev0, ev1, 0u, rc, tiles, rad, s->d_wide, s->d_tri4,
                              s->d_shade4, s->d_frame4, s->n_wide, s->n_tris, 0u, n_slots, next_slot, stats, fp.refill, tmin, tmax, fp.lds_stack);
    else
        hipExtLaunchKernelGGL((k_fused<false>), dim3(fp.grid), dim3(FTB), (uint32_t)fp.smem, st, ev0, ev1, 0u, rc, tiles, rad, s->d_wide, s->d_tri4,
                              s->d_shade4, s->d_frame4, s->n_wide, s->n_tris, 0u, n_slots, next_slot, stats, fp.refill, tmin, tmax, fp.lds_stack);
}
