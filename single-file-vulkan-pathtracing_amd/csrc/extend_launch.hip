// extend_launch.hip -- which closest-hit kernel stands in for traceRayEXT (raygen.rgen:63-75) on a given scene, and its launch.
//
//   scene                                   kernel (AUTO)                          where the scene lives
//   single level, <= 24 KB                  k_extend_lds7p / k_extend_lds7         LDS (BVH4 + 3 permuted triangle copies)
//   single level, up to ~11 000 triangles   k_extend<hbm> (extend_hbm.hip)         L2: 64-B fp16 BVH4 nodes, top-down layout
//   single level, beyond                    k_extend8 (extend_hbm.hip)             L2 / MALL / HBM: 64-B byte-plane 8-wide nodes
//   instanced                               k_extend_inst16                        TLAS in L2 (top levels in LDS), BLAS in LDS
//   instanced, outside that kernel's limits k_extend_inst                          fp32 nodes, both levels through L1 / L2
// PT_EXTEND_* asks for a variant explicitly (tests, A/B); every variant returns the same hit records.
#include "wavefront_host.h"

#include <algorithm>

#include "extend_kernel.h"  // k_extend<>, k_extend_lds7 / _lds7p (+ their shadow-ray twins), the slab / stack helpers
#include "extend_inst16.h"  // k_extend_inst16
#include "extend_inst.h"    // k_extend_inst

namespace {

}  // namespace

pt_status ptw_plan_extend(pt_scene *s, uint32_t want, ExtendPlan &pl)
{
    pt_ctx *ctx = s->ctx;
    if (want > PT_EXTEND_HBM8) { ctx->err = "unknown extend variant"; return PT_ERR_INVALID_ARG; }
    if (want == PT_EXTEND_FLAT_REMOVED) {
        ctx->err = "PT_EXTEND_FLAT (the brute-force loop, never AUTO) was removed in API version 5: every tree walk returns the brute-force closest hit";
        return PT_ERR_UNSUPPORTED;
    }
    if (s->broken) {  // an earlier rebuild of the tree ran out of memory (lbvh_build.hip): never launch on null tables
        const pt_status rcb = ptb_repair(s);
        if (rcb != PT_OK) return rcb;
    }
    // AUTO walks scenes beyond L2 (at first; now nearly every scene beyond LDS, below) through the 8-wide tree (64-B nodes with byte planes): fewer distinct lines per ray -- measured on
    // MI355X, same box, three rounds: C5 2 465 -> 2 547 Mrays/s (+3.4 %), C5x 2 405 -> 2 546 (+5.9 %), 36.2 -> 27.7 and 29.5 -> 24.2
    // node visits per ray (profiles/r03_ab_c5_c5x_hbm8_64B_nodes.log); pt_tuning.hbm8 = 0 keeps the BVH4, 1 takes the 8-wide tree
    // for every scene that does not fit LDS
    const uint64_t ws4 = 64ull * (s->n_wide16t ? s->n_wide16t : s->n_wide) + 64ull * s->n_tris;
    // (round 3, last session: with the 8-wide kernel's new vote, refill threshold and spill-free instantiation the crossover fell from
    // 32 MiB of BVH4 nodes + records to ~1 MiB, i.e. ~11 000 triangles -- soups of 2 500 / 5 000 / 12 000 / 20 000 / 50 000 / 100 000 /
    // 200 000 / 400 000 triangles, 8-wide against BVH4 kernel: -4 / -1.7 / +1.7 / +3.8 / +11 / +11 / +18 / +21 %,
    // profiles/r03ca_bvh4_vs_8wide_midsize.log, r03cb_bvh4_vs_8wide_small.log)
    const bool auto8_big = want == PT_EXTEND_AUTO && ctx->tune.hbm8 != 0 && ws4 > (1ull << 20) && s->n_tris > PT_SAH_MAX_TRIS;
    if ((want == PT_EXTEND_HBM8 || (want == PT_EXTEND_AUTO && ctx->tune.hbm8 == 1) || auto8_big) && !s->n_inst && !s->d_wide8) {
        const pt_status rc8 = ptb_ensure_wide8(s);   // built on first request (260 B per triangle nobody else needs)
        if (rc8 != PT_OK) return rc8;
    }
    if (want == PT_EXTEND_HBM8 && (s->n_inst || !s->d_wide8)) { ctx->err = "no 8-wide nodes for this scene (instanced, or <= 2048 triangles)"; return PT_ERR_UNSUPPORTED; }
    if (s->n_inst) {  // two-level scenes: one kernel variant (BVH4s read through L1/L2)
        if (want == PT_EXTEND_LDS) { ctx->err = "instanced scenes only have the HBM extend variant"; return PT_ERR_UNSUPPORTED; }
        pl.variant = PT_EXTEND_HBM;
        const size_t blas_bytes = 16 * LDS_NODE_F4 * (size_t)s->n_wide + sizeof(float4) * 9 * (size_t)s->n_tris;
        pl.lds_scene = blas_bytes <= 24 * 1024;  // here: the BLAS (shared by all instances) is staged in LDS
        pl.smem = (size_t)LDS_STACK * TB * sizeof(uint2) + (pl.lds_scene ? blas_bytes : 0);
        int per_cu_i = 0;
        PT_HIP(ctx, hipOccupancyMaxActiveBlocksPerMultiprocessor(
                        &per_cu_i, pl.lds_scene ? reinterpret_cast<const void *>(k_extend_inst<false, true>)
                                                : reinterpret_cast<const void *>(k_extend_inst<false, false>), TB, pl.smem));
        per_cu_i = std::max(1, std::min(per_cu_i, 8));
        pl.refill = 64;  // a wave takes new rays only when all its lanes are done: entering an instance (ray transform, three
                         // divides) and the TLAS root are too expensive to run for a few refilled lanes.  C4: 16: 8.2, 32: 8.85,
                         // 48: 9.26, 56: 9.2, 64: 9.38 Grays/s
        pl.refill = pt_tuned(ctx->tune.refill, pl.refill, 1, 64);
        pl.grid = ctx->num_cus * per_cu_i;
        pl.smem_inst_fallback = pl.smem; pl.grid_inst_fallback = pl.grid;
        // the round-2 kernel when both levels fit its 15-bit child codes and the BLAS fits LDS
        const size_t smem16_scene = sizeof(uint32_t) * I16_NODE_DW * (size_t)s->n_wide + sizeof(float4) * 9 * (size_t)s->n_tris;
        const int lds16 = pt_tuned(ctx->tune.lds_stack, 16, 1, 32);
        pl.inst16 = s->d_tlas16 && s->d_wide16 && s->n_inst < 32768u && s->n_tlas16 < 32767u && s->n_wide < 32767u && s->n_tris <= 2047u &&
                    smem16_scene <= 24 * 1024 && ctx->tune.inst16 != 0;
        if (pl.inst16) {
            pl.lds_stack = lds16;
            // top levels of the TLAS staged in LDS next to the BLAS.  8 KB (102 nodes: the top four levels) measured best on
            // C4: 0 / 4 / 8 / 16 / 24 KB -> 11.95 / 12.16 / 12.31 / 10.7 / 11.1 Grays/s (profiles/r02i_ab_c4_tlas_lds.log) --
            // from 16 KB on the four resident blocks leave the other pipeline's k_shade no LDS to run beside them
            const size_t tlas_lds_bytes = (size_t)pt_tuned(ctx->tune.tlas_lds_kb, 8, 0, 96) * 1024;
            pl.n_tlas_lds = (uint32_t)std::min<size_t>(s->n_tlas16, tlas_lds_bytes / (sizeof(uint32_t) * I16_NODE_DW));
            pl.smem = (size_t)lds16 * TB * sizeof(uint32_t) + smem16_scene + sizeof(uint32_t) * I16_NODE_DW * (size_t)pl.n_tlas_lds;
            const void *fn16 = s->pair_leaves ? reinterpret_cast<const void *>(k_extend_inst16<false, true>)
                                              : reinterpret_cast<const void *>(k_extend_inst16<false, false>);
            if (pl.smem > 48 * 1024)
                for (const void *f : { reinterpret_cast<const void *>(k_extend_inst16<false, true>), reinterpret_cast<const void *>(k_extend_inst16<false, false>),
                                       reinterpret_cast<const void *>(k_extend_inst16<true, true>), reinterpret_cast<const void *>(k_extend_inst16<true, false>),
                                       // (the shadow-ray twins of the NEE pipeline: the same launch shape)
                                       reinterpret_cast<const void *>(k_extend_inst16<false, true, true>), reinterpret_cast<const void *>(k_extend_inst16<false, false, true>) })
                    PT_HIP(ctx, hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)pl.smem));
            int per16 = 0;
            PT_HIP(ctx, hipOccupancyMaxActiveBlocksPerMultiprocessor(&per16, fn16, TB, pl.smem));
            // four blocks per CU and refill at 48 idle lanes measured best on the 10 000-instance grid (C4: 4/48 10.73,
            // 5/48 10.33, 4/40 10.66, 4/56 10.35, 4/64 9.74 Grays/s; the fp32 kernel at its best, 4/64: 9.33)
            per16 = pt_tuned(ctx->tune.inst16_blocks, std::min(per16, 4), 1, 8);
            pl.grid = ctx->num_cus * std::max(1, std::min(per16, 8));
            pl.refill = pt_tuned(ctx->tune.refill, 48, 1, 64);
        }
        // TLAS pushes <= 3 per level + 3 extra instances of a leaf, + EXIT, + the BLAS walk: the exact bound of the
        // BVH4 that is TRAVERSED when the builder gave one (the surface-area BVH4 of a small scene can be deeper
        // than the balanced LBVH whose height s->height is), else 3 per level of the collapsed LBVH
        const uint32_t blas_bound = s->stack_need != 0xFFFFFFFFu ? s->stack_need + 1u : 3u * (s->height_tree / 2u + 1u);
        const uint32_t bound_i = 3u * (std::max(s->tlas_height / 2u + 1u, s->tlas16_levels)) + 4u + blas_bound + 2u;
        pl.spill_levels = bound_i > (uint32_t)LDS_STACK ? bound_i - (uint32_t)LDS_STACK : 0u;  // (sized for the 8-entry fallback kernel)
        const size_t need_i = PT_MAX_PIPES * (size_t)std::max(pl.spill_levels, 1u) * (size_t)std::max(pl.grid, pl.grid_inst_fallback) * TB * sizeof(uint2);
        if (need_i > ctx->spill_bytes) {
            (void)hipFree(ctx->d_spill);
            ctx->d_spill = nullptr;
            ctx->spill_bytes = 0;
            PT_HIP(ctx, hipMalloc((void **)&ctx->d_spill, need_i));
            ctx->spill_bytes = need_i;
        }
        return PT_OK;
    }
    const size_t scene_bytes = 16 * LDS_NODE_F4 * (size_t)s->n_wide + sizeof(float4) * 9 * (size_t)s->n_tris;  // 3 permuted triangle copies
    // (round 2's 128-B eight-wide node with fp16 planes visited 26 % fewer nodes and fetched as many 128-B LINES -- two 64-B
    // BVH4 siblings share one -- and lost: 458 vs 395 ms of kernel time per 4 frames of C5; the 64-B node above is its successor)
    const bool auto8 = want == PT_EXTEND_AUTO && scene_bytes > 24 * 1024 && s->d_wide8 && (ctx->tune.hbm8 == 1 || auto8_big);
    if (want == PT_EXTEND_HBM8 || auto8) {
        pl.variant = PT_EXTEND_HBM8;
        pl.bvh8 = true;
        pl.lds_scene = false;
        // one stack entry per visited node: at most one per level of the 8-wide tree.  The LDS stack is sized to exactly that (8 M
        // triangles: 10 entries) -- then the kernel is instantiated without the spill column's address arithmetic (C5 +1.9 %, C5x
        // +2.5 %) and the LDS it does not take is there for the co-resident k_shade (9 ... 11 entries instead of 12: C5 +1 %; one
        // entry too few, i.e. the spill kernel: -1.7 %; profiles/r03bk_*, r03bl_*)
        const uint32_t bound8 = s->levels8 + 1u;
        pl.lds_stack = pt_tuned(ctx->tune.lds_stack, (int)std::min(std::max(bound8, 4u), 12u), 1, 32);
        pl.smem = (size_t)pl.lds_stack * TB * sizeof(uint2);
        const bool spills8 = bound8 > (uint32_t)pl.lds_stack;
        // 7 waves per SIMD where the walk waits on HBM (same rule as AUTO ray sorting: nodes + records beyond the Infinity Cache);
        // pt_tuning.extend_blocks = 6 / 7 forces either
        pl.waves7 = !spills8 && pl.lds_stack <= 10 &&
                    (ctx->tune.extend_blocks == 7 || (ctx->tune.extend_blocks < 0 && 64ull * s->n_wide8 + 64ull * s->n_tris > (256ull << 20)));
        int per_cu8 = 0;
        PT_HIP(ctx, hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu8, ptw_extend8_fn(false, spills8, pl.waves7), TB, pl.smem));
        per_cu8 = std::max(1, std::min(per_cu8, 8));
        // with the triangle vote at 16 lanes (launch_extend) the refill optimum moved from 32 idle lanes to 12: C5 2 907 ->
        // 3 240 Mrays/s, C5x 2 715 -> 2 960 for both together (profiles/r03bb_*, r03bc_*: 32: 3 025, 24: 3 140, 16: 3 230, 8: 3 235, 4: 3 170)
        pl.refill = pt_tuned(ctx->tune.refill, 12, 1, 64);
        pl.grid = ctx->num_cus * per_cu8;
        pl.spill_levels = bound8 > (uint32_t)pl.lds_stack ? bound8 - (uint32_t)pl.lds_stack : 0u;
        const size_t need8 = PT_MAX_PIPES * (size_t)std::max(pl.spill_levels, 1u) * (size_t)pl.grid * TB * sizeof(uint2);
        if (need8 > ctx->spill_bytes) {
            (void)hipFree(ctx->d_spill);
            ctx->d_spill = nullptr;
            ctx->spill_bytes = 0;
            PT_HIP(ctx, hipMalloc((void **)&ctx->d_spill, need8));
            ctx->spill_bytes = need8;
        }
        return PT_OK;
    }
    if (want == PT_EXTEND_LDS && scene_bytes > 96 * 1024) { ctx->err = "scene does not fit LDS"; return PT_ERR_UNSUPPORTED; }
    pl.lds_scene = want == PT_EXTEND_LDS || (want == PT_EXTEND_AUTO && scene_bytes <= 24 * 1024);
    pl.variant = pl.lds_scene ? PT_EXTEND_LDS : PT_EXTEND_HBM;
    // deep trees of big scenes: 12 LDS entries measured best on the 1M-triangle soup (4: -15 %, 8: -3 %,
    // 16: -5 %, 24: -16 %: beyond 12 the extra LDS costs occupancy; 9/10/11, which would admit a 7th block per CU: -2.4 %)
    pl.lds_stack = pl.lds_scene ? LDS_STACK : 12;
    // LDS-resident scenes are small enough for an exact stack bound (lbvh_build.hip: wide_stack_need):
    // if it fits 16 LDS entries the kernel is instantiated without the spill path (Cornell: 9)
    // (the no-spill kernel packs child words into 14 bits: <= 2047 triangles, <= 8191 nodes, leaves of <= 4)
    pl.spill = !(pl.lds_scene && s->stack_need <= 16u && s->n_tris <= 2047u && s->n_wide <= 8191u);
    if (!pl.spill) pl.lds_stack = (int)std::max(s->stack_need, 1u);
    pl.pairs = !pl.spill && s->pair_leaves && ctx->tune.pair_kernel != 0;
    pl.smem_wide_entries = (size_t)pl.lds_stack * TB * sizeof(uint2) + (pl.lds_scene ? scene_bytes : 0);
    pl.smem = pl.spill ? pl.smem_wide_entries : (size_t)pl.lds_stack * TB * sizeof(uint32_t) + scene_bytes;
    if (!pl.spill && pl.smem_wide_entries > 48 * 1024) {
        PT_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(k_extend<true, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)pl.smem_wide_entries));
        PT_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(k_extend<true, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)pl.smem_wide_entries));
    }
    const void *fn = pl.pairs ? reinterpret_cast<const void *>(k_extend_lds7p)
                     : !pl.spill ? reinterpret_cast<const void *>(k_extend_lds7)
                     : pl.lds_scene ? reinterpret_cast<const void *>(k_extend<true, false, true>)
                                    : ptw_extend_hbm_fn(false, ctx->tune.rec64 != 0);
    const void *fn_count = pl.pairs ? reinterpret_cast<const void *>(k_extend<true, true, false, true>)
                           : !pl.spill ? reinterpret_cast<const void *>(k_extend<true, true, false>)
                           : pl.lds_scene ? reinterpret_cast<const void *>(k_extend<true, true, true>)
                                          : ptw_extend_hbm_fn(true, ctx->tune.rec64 != 0);
    if (pl.smem > 48 * 1024)
        PT_HIP(ctx, hipFuncSetAttribute(fn_count, hipFuncAttributeMaxDynamicSharedMemorySize, (int)pl.smem));
    if (pl.smem > 48 * 1024) PT_HIP(ctx, hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)pl.smem));
    if (pl.smem > 48 * 1024 && !pl.spill)  // the shadow-ray twins of the two compact kernels (NEE pipeline)
        PT_HIP(ctx, hipFuncSetAttribute(pl.pairs ? reinterpret_cast<const void *>(k_extend_lds7p_sh) : reinterpret_cast<const void *>(k_extend_lds7_sh),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)pl.smem));
    int per_cu = 0;
    PT_HIP(ctx, hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, TB, pl.smem));
    per_cu = std::max(1, std::min(per_cu, 8));
    per_cu = pt_tuned(ctx->tune.extend_blocks, per_cu, 1, per_cu);
    // big scenes (vote-scheduled steps): 32 idle lanes measured best on C5 (16: -2.5 %, 48: -3 %)
    pl.refill = pt_tuned(ctx->tune.refill, pl.lds_scene ? REFILL_MIN_IDLE : 32, 1, 64);
    pl.grid = ctx->num_cus * per_cu;
    // the HBM variant of a scene whose traversed BVH4 is the collapsed LBVH walks the top-down layout of it
    pl.topdown4 = !pl.lds_scene && s->bvh4_builder != 1 && s->d_wide16t && ctx->tune.topdown4 != 0;
    // stack bound: the exact one of the BVH4 that is traversed when its builder computed it (small scenes; the
    // surface-area BVH4 is not bounded by the LBVH's height), else a BVH4 node pushes <= 3 entries per level and the
    // collapsed LBVH's wide height is <= binary height/2 + 1
    const uint32_t bound = pl.topdown4 ? 3u * s->levels4t + 1u
                           : s->stack_need != 0xFFFFFFFFu ? s->stack_need + 1u : 3u * (s->height_tree / 2u + 1u) + 1u;
    pl.spill_levels = bound > (uint32_t)pl.lds_stack ? bound - (uint32_t)pl.lds_stack : 0u;
    const size_t need = PT_MAX_PIPES * (size_t)std::max(pl.spill_levels, 1u) * (size_t)pl.grid * TB * sizeof(uint2);
    if (need > ctx->spill_bytes) {
        (void)hipFree(ctx->d_spill);
        ctx->d_spill = nullptr;
        ctx->spill_bytes = 0;
        PT_HIP(ctx, hipMalloc((void **)&ctx->d_spill, need));
        ctx->spill_bytes = need;
    }
    return PT_OK;
}

void ptw_launch_extend(const ExtendPlan &pl, pt_scene *s, const float4 *rayA, const float2 *rayB, float4 *hit, uint32_t *hit_inst,
                       const uint32_t *count_in, uint32_t *count_zero, unsigned long long *stats, float tmin, float tmax, bool count,
                       bool raw_hit, hipStream_t st, int pipe, hipEvent_t ev0, hipEvent_t ev1, const uint32_t *perm, const float *ray_tmax)
{
    const int raw = raw_hit ? 1 : 0;  // hit records as (pos, V, W, det) for k_shade instead of (pos, t, u, v)
    // hipExtLaunchKernelGGL stamps THIS kernel's start/stop into ev0/ev1 (null = plain launch): under
    // two overlapping pipelines an event recorded between kernels would also count queueing time
    // each concurrently running extend kernel owns its own [spill_levels][grid*TB] region
    const size_t spill_off = (size_t)pipe * std::max(pl.spill_levels, 1u) * (size_t)pl.grid * TB;
    if (s->n_inst && pl.inst16 && tmin > 0.f) {
        // own region of the spill buffer, counted in dwords (the buffer is sized in 8-byte entries for the larger grid)
        uint32_t *sp32 = reinterpret_cast<uint32_t *>(reinterpret_cast<uint2 *>(s->ctx->d_spill) + (size_t)pipe * std::max(pl.spill_levels, 1u) * (size_t)std::max(pl.grid, pl.grid_inst_fallback) * TB);
        const uint32_t str = (uint32_t)pl.grid * TB;
        const NormBox nbt = { s->tlas_norm_c[0], s->tlas_norm_c[1], s->tlas_norm_c[2], s->tlas_norm_s[0], s->tlas_norm_s[1], s->tlas_norm_s[2],
                              s->tlas_norm_rs[0], s->tlas_norm_rs[1], s->tlas_norm_rs[2] };
        const NormBox nbb = { s->norm_c[0], s->norm_c[1], s->norm_c[2], s->norm_s[0], s->norm_s[1], s->norm_s[2], s->norm_rs[0], s->norm_rs[1], s->norm_rs[2] };
        const int enter_min = pt_tuned(s->ctx->tune.enter_min, 16, 1, 64);  // lanes that wait to enter an instance together (8, 16, 24 measured alike within 1 %)
        // ... and lanes that wait with a triangle leaf (extend_inst16.h): 8 until round 6; on the BLAS with the least-area cut 8 / 12 / 14 / 20 -> 16.12 / 16.33 /
        // 16.34 / 16.19 Grays/s on the 10 000-instance grid (profiles/r06o_c4_wavefront_knobs.log)
        const int leaf_min = pt_tuned(s->ctx->tune.leaf_min, 14, 1, 64);
        // the node loop yields to the lanes waiting with a leaf once fewer than 1/6 of the wave's rays still descend
        // (C4 11.7 -> 12.2 Grays/s; 2, 3, 4, 8 measured within 1 % of it, 0 = never: profiles/r02i_ab_c4_node_yield.log)
        const int node_yield = pt_tuned(s->ctx->tune.node_yield, 6, 0, 64);
#define PT_LAUNCH_INST16(C, P, S)                                                                                           \
    hipExtLaunchKernelGGL((k_extend_inst16<C, P, S>), dim3(pl.grid), dim3(TB), (uint32_t)pl.smem, st, ev0, ev1, 0u, s->d_tlas16, nbt, \
                          reinterpret_cast<const uint4 *>(s->d_wide16), nbb, s->d_tri4, s->n_wide, s->n_tris, s->d_inst6,     \
                          s->d_tlas_prim_of, rayA, rayB, hit, hit_inst, count_in, count_zero, stats, sp32, str, pl.refill, tmin, \
                          tmax, raw, pl.lds_stack, enter_min, leaf_min, node_yield, pl.n_tlas_lds, ray_tmax)
        if (ray_tmax) { if (s->pair_leaves) PT_LAUNCH_INST16(false, true, true); else PT_LAUNCH_INST16(false, false, true); }  // shadow rays (NEE)
        else if (s->pair_leaves) { if (count) PT_LAUNCH_INST16(true, true, false); else PT_LAUNCH_INST16(false, true, false); }
        else { if (count) PT_LAUNCH_INST16(true, false, false); else PT_LAUNCH_INST16(false, false, false); }
#undef PT_LAUNCH_INST16
        return;
    }
    if (s->n_inst) {
        const int grid_i = pl.inst16 ? pl.grid_inst_fallback : pl.grid;
        const size_t smem_i = pl.inst16 ? pl.smem_inst_fallback : pl.smem;
        uint2 *sp = reinterpret_cast<uint2 *>(s->ctx->d_spill) + (size_t)pipe * std::max(pl.spill_levels, 1u) * (size_t)std::max(pl.grid, pl.grid_inst_fallback) * TB;
        const uint32_t str = (uint32_t)grid_i * TB;
#define PT_LAUNCH_INST(C, L, S)                                                                                          \
    hipExtLaunchKernelGGL((k_extend_inst<C, L, S>), dim3(grid_i), dim3(TB), (uint32_t)smem_i, st, ev0, ev1, 0u, s->d_tlas_wide, \
                          s->d_wide, s->d_tri4, s->n_wide, s->n_tris, s->d_inst6, s->d_tlas_prim_of, rayA, rayB, hit,           \
                          hit_inst, count_in, count_zero, stats, sp, str, pl.refill, tmin, tmax, raw, ray_tmax)
        if (ray_tmax) { if (pl.lds_scene) PT_LAUNCH_INST(false, true, true); else PT_LAUNCH_INST(false, false, true); }  // shadow rays (NEE)
        else if (pl.lds_scene) {
            if (count) PT_LAUNCH_INST(true, true, false); else PT_LAUNCH_INST(false, true, false);
        } else {
            if (count) PT_LAUNCH_INST(true, false, false); else PT_LAUNCH_INST(false, false, false);
        }
#undef PT_LAUNCH_INST
        return;
    }
    uint2 *spill = reinterpret_cast<uint2 *>(s->ctx->d_spill) + spill_off;
    const uint32_t stride = (uint32_t)pl.grid * TB;
    if (pl.bvh8) {
        // the vote of the 8-wide kernel (extend8_kernel.h): a triangle step runs once tri_enter lanes wait with leaf triangles
        // (or more than descend), and repeats while tri_stay lanes still hold one.  Majority voting (64) parks ~25 lanes
        // behind every node step -- a node step is 240 instructions, a triangle step 110: 16 measured best (8: -2 %, 12: -0.5 %,
        // 20: equal on C5x, 24: -3 %; repeating triangle steps changes nothing: profiles/r03ba_ab_c5_vote.log)
        const int tri_enter = pt_tuned(s->ctx->tune.tri_enter, 16, 1, 64), tri_stay = pt_tuned(s->ctx->tune.tri_stay, 65, 1, 65);
        ptw_launch_extend8(count, pl.spill_levels > 0u, pl.waves7, pl.grid, pl.smem, st, ev0, ev1, s->d_wide8, s->norm_c, s->norm_s, s->norm_rs, s->d_tri4_8, s->d_shade64_8, rayA, rayB, hit,
                           count_in, count_zero, stats, spill, stride, pl.refill | (tri_enter << 8) | (tri_stay << 16), tmin, tmax, pl.lds_stack, raw, perm, ray_tmax);
        return;
    }
    const NormBox nbox = { s->norm_c[0], s->norm_c[1], s->norm_c[2], s->norm_s[0], s->norm_s[1], s->norm_s[2],
                           s->norm_rs[0], s->norm_rs[1], s->norm_rs[2] };
    // one-dword stack entries truncate the entry distance toward zero, which is only conservative for t >= 0, and the
    // sort of the one-dword keys takes entry distances for positive floats (tmin = 0 could make one -0): a tmin <= 0
    // (not valid in Vulkan, accepted here) runs the same plan through the 8-byte-entry kernel
    const bool no_spill = !pl.spill && tmin > 0.f;
    const size_t smem = (!pl.spill && !no_spill) ? pl.smem_wide_entries : pl.smem;
#define PT_LAUNCH_EXTEND(L, C, S)                                                                                     \
    hipExtLaunchKernelGGL((k_extend<L, C, S>), dim3(pl.grid), dim3(TB), (uint32_t)smem, st, ev0, ev1, 0u, s->d_wide, \
                          s->d_wide16, nbox,                                                                          \
                          s->d_tri4, s->n_wide, s->n_tris, rayA, rayB, hit, count_in, count_zero, stats, spill, stride, \
                          pl.refill, tmin, tmax, pl.lds_stack, raw, nullptr, ray_tmax, nullptr)
    if (no_spill && pl.pairs) {
        if (count)
            hipExtLaunchKernelGGL((k_extend<true, true, false, true>), dim3(pl.grid), dim3(TB), (uint32_t)smem, st, ev0, ev1, 0u, s->d_wide,
                                  s->d_wide16, nbox, s->d_tri4, s->n_wide, s->n_tris, rayA, rayB, hit, count_in, count_zero, stats, spill,
                                  stride, pl.refill, tmin, tmax, pl.lds_stack, raw, nullptr, ray_tmax, nullptr);
        else
            hipExtLaunchKernelGGL(ray_tmax ? k_extend_lds7p_sh : k_extend_lds7p, dim3(pl.grid), dim3(TB), (uint32_t)smem, st, ev0, ev1, 0u, s->d_wide, s->d_wide16, nbox,
                                  s->d_tri4, s->n_wide, s->n_tris, rayA, rayB, hit, count_in, count_zero, stats, spill, stride,
                                  pl.refill, tmin, tmax, pl.lds_stack, raw, nullptr, ray_tmax, nullptr);
    } else if (no_spill) {
        if (count) PT_LAUNCH_EXTEND(true, true, false);
        else
            hipExtLaunchKernelGGL(ray_tmax ? k_extend_lds7_sh : k_extend_lds7, dim3(pl.grid), dim3(TB), (uint32_t)smem, st, ev0, ev1, 0u, s->d_wide, s->d_wide16, nbox,
                                  s->d_tri4, s->n_wide, s->n_tris, rayA, rayB, hit, count_in, count_zero, stats, spill, stride,
                                  pl.refill, tmin, tmax, pl.lds_stack, raw, nullptr, ray_tmax, nullptr);
    } else if (pl.lds_scene) {
        if (count) PT_LAUNCH_EXTEND(true, true, true); else PT_LAUNCH_EXTEND(true, false, true);
    } else {
        ptw_launch_extend_hbm(count, s->ctx->tune.rec64 != 0, pl.grid, smem, st, ev0, ev1, s->d_wide, pl.topdown4 ? reinterpret_cast<const uint2 *>(s->d_wide16t) : s->d_wide16, s->norm_c, s->norm_s, s->norm_rs, s->d_tri4,
                              s->d_shade64, s->n_wide, s->n_tris, rayA, rayB, hit, count_in, count_zero, stats, spill, stride,
                              pl.refill | (pt_tuned(s->ctx->tune.tri_enter, 0, 0, 64) << 8), tmin, tmax, pl.lds_stack, raw, perm, ray_tmax);
    }
#undef PT_LAUNCH_EXTEND
}
