// fused.hip -- PT_PIPELINE_FUSED: the kernel (fused_kernel.h), the scenes it takes, its launch.
#include "wavefront_host.h"

#include <algorithm>

#define PT_EXTEND_TEMPLATES_ONLY
#include "extend_kernel.h"  // the LDS node / stack helpers the fused kernel shares with k_extend_lds7p
#include "extend_inst16.h"  // the two-level walk's node codes and register barrier (k_extend_inst16 itself is a template: not instantiated here)

#ifdef PT_FUSED_TIMELINE
__device__ unsigned long long *g_fused_timeline = nullptr;
extern "C" int pt_debug_fused_timeline(void *device_u64x4_per_wave)
{
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_fused_timeline), &device_u64x4_per_wave, sizeof(void *));
}
#endif

#ifdef PT_FUSED_HIST
__device__ uint32_t *g_fused_hist = nullptr;
extern "C" int pt_debug_fused_hist(void *device_u32_2x8192x16)
{
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_fused_hist), &device_u32_2x8192x16, sizeof(void *));
}
#endif

#include "fused_cull.h"

namespace {
using namespace ptw;
#include "fused_dev.h"
#include "fused_kernel.h"
#include "fused_inst_kernel.h"
constexpr size_t FUSED_COUNT_LDS = sizeof(uint32_t) * 2 * FB_N * (FTB / 64);  // the instrumented twin's per-wave block counters, behind the product plan
static_assert((int)FB_N == (int)PT_FB_COUNT && FB_N <= PT_N_BLOCKS, "fused_kernel.h FusedBlock mirrors include/pt_api.h pt_fused_block");

// two-level scenes: k_extend_inst16's class (extend_launch.hip: both levels in 15-bit child codes, BLAS in LDS, pair leaves)
pt_status plan_fused_inst(pt_scene *s, const ExtendPlan &pl, float tmin, FusedPlan &fp)
{
    pt_ctx *ctx = s->ctx;
    const size_t tables = sizeof(float4) * 3 * (size_t)s->n_tris;  // shade4 (the vertices are the kz = 2 triangle copy)
    if (!pl.inst16 || !(tmin > 0.f) || tables > 16 * 1024) {
        ctx->err = "PT_PIPELINE_FUSED takes instanced scenes of the fp16 two-level kernel's class: 2 .. 32767 instances, a BLAS of <= 2047 "
                   "triangles that fits LDS, tmin > 0";
        return PT_ERR_UNSUPPORTED;
    }
    fp.inst = true;
    fp.lds_stack = pl.lds_stack;
    // TLAS nodes staged in LDS: the wavefront kernel keeps 8 KB of them because the other pipeline's k_shade needs LDS beside it
    // (extend_launch.hip); this kernel has the CU to itself
    const size_t tlas_lds_bytes = (size_t)pt_tuned(ctx->tune.tlas_lds_kb, PT_FUSEDI_TLAS_KB, 0, 96) * 1024;
    fp.n_tlas_lds = (uint32_t)std::min<size_t>(s->n_tlas16, tlas_lds_bytes / (sizeof(uint32_t) * I16_NODE_DW));
    fp.smem = (size_t)fp.lds_stack * FITB * sizeof(uint32_t) + sizeof(uint32_t) * I16_NODE_DW * ((size_t)s->n_wide + fp.n_tlas_lds) +
              sizeof(float4) * 9 * (size_t)s->n_tris + tables + sizeof(uint32_t) * FS_FIELDS * FITB + sizeof(uint32_t) * (FITB / 64) * PT_FUSED_WTILES;
    if (fp.smem > 160 * 1024) { ctx->err = "PT_PIPELINE_FUSED: the two-level kernel's LDS plan exceeds 160 KB (pt_tuning lds_stack / tlas_lds_kb)"; return PT_ERR_UNSUPPORTED; }
    int per_cu = ctx->fused_per_cu[1];
    const size_t key1 = (fp.smem << 1) | (s->pair_leaves ? 1u : 0u);
    if (ctx->fused_smem[1] != key1 || per_cu <= 0) {
        for (const void *fn : { reinterpret_cast<const void *>(k_fused_inst<false, true>), reinterpret_cast<const void *>(k_fused_inst<true, true>),
                                 reinterpret_cast<const void *>(k_fused_inst<false, false>), reinterpret_cast<const void *>(k_fused_inst<true, false>) })
            if (fp.smem > 48 * 1024) PT_HIP(ctx, hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)fp.smem));
        PT_HIP(ctx, hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, s->pair_leaves ? reinterpret_cast<const void *>(k_fused_inst<false, true>) : reinterpret_cast<const void *>(k_fused_inst<false, false>), FITB, fp.smem));
        ctx->fused_smem[1] = key1;
        ctx->fused_per_cu[1] = per_cu;
    }
    per_cu = std::max(1, std::min(per_cu, 8));
    per_cu = pt_tuned(ctx->tune.extend_blocks, per_cu, 1, per_cu);
    fp.grid = ctx->num_cus * per_cu;
    fp.block = FITB;
    fp.refill = pt_tuned(ctx->tune.refill, 48, 1, 64);
    // stack entries beyond the LDS ones: one dword each, [level][thread], in the context's spill area (sized by ptw_plan_extend for
    // the wavefront kernels' grids; grown here if this grid asks for more)
    const size_t need = (size_t)std::max(pl.spill_levels, 1u) * (size_t)fp.grid * FITB * sizeof(uint32_t);
    if (need > ctx->spill_bytes) {
        (void)hipFree(ctx->d_spill);
        ctx->d_spill = nullptr;
        ctx->spill_bytes = 0;
        PT_HIP(ctx, hipMalloc((void **)&ctx->d_spill, need));
        ctx->spill_bytes = need;
    }
    fp.spill = reinterpret_cast<uint32_t *>(ctx->d_spill);
    if (ctx->tune.inst_frames != 0) {
        const pt_status rcf = ptb_ensure_inst_frames(s);
        if (rcf != PT_OK) return rcf;
        fp.inst_frame = s->d_inst_frame;
    }
    return PT_OK;
}
}  // namespace

pt_status ptw_plan_fused(pt_scene *s, const ExtendPlan &pl, float tmin, FusedPlan &fp)
{
    pt_ctx *ctx = s->ctx;
    if (s->n_inst) return plan_fused_inst(s, pl, tmin, fp);
    const size_t tables = sizeof(float4) * 5 * (size_t)s->n_tris;  // shade4 + tangent frames (the vertices are the kz = 2 triangle copy)
    if (pl.variant != PT_EXTEND_LDS || pl.spill || !(tmin > 0.f) || tables > 16 * 1024) {
        ctx->err = "PT_PIPELINE_FUSED is for single-level scenes whose BVH4, triangles and shading tables fit LDS (the compact "
                   "kernels' class: <= 2047 triangles in <= 24 KB, stack bound <= 16, tmin > 0)";
        return PT_ERR_UNSUPPORTED;
    }
    // (one stack level more than the walk needs: level -1, never written, is what the node step's read of the stack's top entry lands on when the stack is
    // empty -- fused_kernel.h)
    fp.lds_stack = pl.lds_stack + 1;
    fp.smem = (size_t)fp.lds_stack * FTB * sizeof(uint32_t) + (pl.smem - (size_t)pl.lds_stack * TB * sizeof(uint32_t)) + tables +
              sizeof(uint32_t) * FS_FIELDS * FTB + sizeof(uint32_t) * (FTB / 64) * PT_FUSED_WTILES;
    fp.pairs = pl.pairs;
    int per_cu = ctx->fused_per_cu[0];
    const size_t key0 = (fp.smem << 1) | (pl.pairs ? 1u : 0u);
    if (ctx->fused_smem[0] != key0 || per_cu <= 0) {
        for (const void *fn : { reinterpret_cast<const void *>(k_fused_count<0, true>), reinterpret_cast<const void *>(k_fused_count<1, true>),
                                 reinterpret_cast<const void *>(k_fused_count<2, true>) })
            PT_HIP(ctx, hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(fp.smem + FUSED_COUNT_LDS)));
        for (const void *fn : { reinterpret_cast<const void *>(k_fused<0, true>), reinterpret_cast<const void *>(k_fused<1, true>),
                                 reinterpret_cast<const void *>(k_fused<0, false>), reinterpret_cast<const void *>(k_fused<1, false>),
                                 reinterpret_cast<const void *>(k_fused<2, true>), reinterpret_cast<const void *>(k_fused<2, false>) })
            if (fp.smem > 48 * 1024) PT_HIP(ctx, hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)fp.smem));
        PT_HIP(ctx, hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, pl.pairs ? reinterpret_cast<const void *>(k_fused<0, true>) : reinterpret_cast<const void *>(k_fused<0, false>), FTB, fp.smem));
        ctx->fused_smem[0] = key0;
        ctx->fused_per_cu[0] = per_cu;
    }
    per_cu = std::max(1, std::min(per_cu, 8));
    per_cu = pt_tuned(ctx->tune.extend_blocks, per_cu, 1, per_cu);
    fp.grid = ctx->num_cus * per_cu;
    fp.block = FTB;
    // of 64: the share of a wave's live lanes that must wait with a finished ray before the shade block runs for them.  The block
    // is ~4x a node step, so it pays to run it fuller than k_extend's refill (16): 8 / 16 / 24 / 32 / 40 -> 33.9 / 34.1 / 34.6 /
    // 35.7 / 36.3 Grays/s on the Cornell box at 1080p (profiles/r04b_fused_refill_sweep.txt).  Round 6, after the shade block lost a fifth of
    // its instructions (the shared spawn steps) and the tree a node: 32 / 36 / 40 / 44 / 48 -> 73.1 / 73.1 / 73.8 / 75.1 / 76.1 ms per 16 frames,
    // one blocking frame 5.48 / 5.45 / 5.43 / 5.47 / 5.52 (profiles/r06l_refill_exit_resweep.log): 36
    fp.refill = pt_tuned(ctx->tune.refill, 36, 1, 64);
    return PT_OK;
}

void ptw_launch_fused(const FusedPlan &fp, bool grouped, const ptw::RenderConst &rc, const uint32_t *tiles, const ptw::Radiance &rad,
                      const pt_scene *s, uint32_t n_slots, uint32_t *next_slot, unsigned long long *stats, float tmin, float tmax,
                      hipStream_t st, hipEvent_t ev0, hipEvent_t ev1)
{
    if (fp.inst) {
        const NormBox nbt = { s->tlas_norm_c[0], s->tlas_norm_c[1], s->tlas_norm_c[2], s->tlas_norm_s[0], s->tlas_norm_s[1], s->tlas_norm_s[2],
                              s->tlas_norm_rs[0], s->tlas_norm_rs[1], s->tlas_norm_rs[2] };
        const NormBox nbb = { s->norm_c[0], s->norm_c[1], s->norm_c[2], s->norm_s[0], s->norm_s[1], s->norm_s[2], s->norm_rs[0], s->norm_rs[1], s->norm_rs[2] };
        // the waiting rules of k_extend_inst16 (extend_launch.hip has the measurements), re-swept for this kernel in round 6 on the tree with the
        // least-area cut (leaves are reached a node earlier): a leaf step waits for 14 lanes (8 / 10 / 12 / 14 / 20 / 24: 16.42 / 16.60 / 16.72 / 16.76 /
        // 16.52 / 16.32 Grays/s on the 10 000-instance grid at 16 frames), an instance entry for 12 (profiles/r06m_c4_fused_knobs.log)
        const int enter_min = pt_tuned(s->ctx->tune.enter_min, 12, 1, 64), leaf_min = pt_tuned(s->ctx->tune.leaf_min, 14, 1, 64);
        const int node_yield = pt_tuned(s->ctx->tune.node_yield, 6, 0, 64);
#define PT_LAUNCH_FUSED_INST(G, P)                                                                                                     \
    hipExtLaunchKernelGGL((k_fused_inst<G, P>), dim3(fp.grid), dim3(FITB), (uint32_t)fp.smem, st, ev0, ev1, 0u, rc, tiles, rad, s->d_tlas16, nbt, \
                          reinterpret_cast<const uint4 *>(s->d_wide16), nbb, s->d_tri4, s->d_shade4, s->n_wide, s->n_tris, s->d_inst6,       \
                          s->d_tlas_prim_of, fp.inst_frame, 0u, n_slots, next_slot, stats, fp.spill, (uint32_t)fp.grid * FITB, fp.refill, \
                          tmin, tmax, fp.lds_stack, enter_min, leaf_min, node_yield, fp.n_tlas_lds)
        if (s->pair_leaves) { if (grouped) PT_LAUNCH_FUSED_INST(true, true); else PT_LAUNCH_FUSED_INST(false, true); }
        else { if (grouped) PT_LAUNCH_FUSED_INST(true, false); else PT_LAUNCH_FUSED_INST(false, false); }
#undef PT_LAUNCH_FUSED_INST
        return;
    }
    FastDiv div_frames;  // one group: the hand-out order is tile-major (fused_kernel.h), chunk -> (tile, frame) by this
    div_frames.init(std::max(rc.lanes_active, 1u));
#define PT_LAUNCH_FUSED_K(K, G, P)                                                                                                         \
    hipExtLaunchKernelGGL((K<G, P>), dim3(fp.grid), dim3(FTB), (uint32_t)(fp.smem + (fp.count ? FUSED_COUNT_LDS : 0)), st, ev0, ev1, 0u, rc, tiles, rad, s->d_wide, s->d_tri4,       \
                          s->d_shade4, s->d_frame4, s->n_wide, s->n_tris, 0u, n_slots, next_slot, stats, fp.refill, tmin, tmax, fp.lds_stack, div_frames)
#define PT_LAUNCH_FUSED(G, P) PT_LAUNCH_FUSED_K(k_fused, G, P)
    const int mode = rc.tail ? 2 : (grouped ? 1 : 0);
    if (fp.count) {  // the instrumented twins (pair-leaf trees: what the compact class gets by default)
        if (mode == 2) PT_LAUNCH_FUSED_K(k_fused_count, 2, true); else if (mode == 1) PT_LAUNCH_FUSED_K(k_fused_count, 1, true); else PT_LAUNCH_FUSED_K(k_fused_count, 0, true);
        return;
    }
    if (fp.pairs) { if (mode == 2) PT_LAUNCH_FUSED(2, true); else if (mode == 1) PT_LAUNCH_FUSED(1, true); else PT_LAUNCH_FUSED(0, true); }
    else { if (mode == 2) PT_LAUNCH_FUSED(2, false); else if (mode == 1) PT_LAUNCH_FUSED(1, false); else PT_LAUNCH_FUSED(0, false); }
#undef PT_LAUNCH_FUSED
#undef PT_LAUNCH_FUSED_K
}
