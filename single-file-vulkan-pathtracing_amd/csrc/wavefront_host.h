// wavefront_host.h -- the host-side pieces of pt_render and how they fit together (one translation unit each):
//
//   extend_launch.hip   which closest-hit kernel walks a scene and with what launch shape (ExtendPlan), and its launch
//   shade_kernels.hip   k_generate / k_shade / k_shadow_add / k_resolve / k_hits_to_api and their launchers
//   fused.hip           PT_PIPELINE_FUSED: k_fused (fused_kernel.h), its plan and its launcher
//   film_work.hip       the film's workspace: how many slots a render gets (RenderShape) and the buffers behind them
//   render.hip          pt_render / pt_render_prepare / pt_trace: batches, pipelines on streams, rounds, polls, redo
//
// Kernels are launched from the translation unit that defines them, so each unit exports small launchers; what crosses
// the units is plain data (wavefront_types.h) and the structs below.
#pragma once
#include "wavefront_types.h"

#include <hip/hip_ext.h>

// ---- extend_launch.hip ---------------------------------------------------------------------------------------------
struct ExtendPlan {
    uint32_t variant = PT_EXTEND_LDS;  // PT_EXTEND_LDS / _HBM / _HBM8 that will run
    bool lds_scene = false;
    size_t smem = 0;
    int grid = 0;
    uint32_t spill_levels = 0;
    int refill = 16;
    int lds_stack = 8;          // stack entries per lane kept in LDS (single-level kernel)
    bool spill = true;          // false: the scene's exact stack bound fits lds_stack, kernel without spill path
                                // (and with one-dword stack entries: COMPACT in k_extend)
    bool pairs = false;         // ... and every leaf of the BVH4 is one triangle or one fan pair: the PAIRS kernel
    bool waves7 = false;        // 8-wide kernel: the 72-VGPR instantiation, 7 blocks per CU (scenes beyond the Infinity Cache)
    bool bvh8 = false;          // PT_EXTEND_HBM8: the 8-wide tree and ITS triangle order (s->d_tri4_8, d_shade64_8, d_ke4_8)
    bool topdown4 = false;      // HBM variant over the top-down BVH4 with contiguous children (s->d_wide16t)
    uint32_t n_tlas_lds = 0;    // k_extend_inst16: TLAS nodes staged in LDS (its top levels)
    bool inst16 = false;        // two-level scenes: k_extend_inst16 (64-B fp16 nodes on both levels, one-dword stack entries)
    size_t smem_inst_fallback = 0; int grid_inst_fallback = 0;  // k_extend_inst's launch shape (tmin <= 0 takes it)
    size_t smem_wide_entries = 0;  // LDS bytes of the same plan run by the 8-byte-entry kernel (negative tmin)
};
// Picks the kernel (`want`: PT_EXTEND_*, AUTO by scene size), sizes its launch and the context's stack-spill area; builds the
// 8-wide nodes of a big scene on first use; repairs a scene whose last rebuild failed.
pt_status ptw_plan_extend(pt_scene *s, uint32_t want, ExtendPlan &pl);
// One closest-hit launch over the rays [0, *count_in) of a queue.  ev0 / ev1 (nullable): the kernel's own start / stop
// events; perm: ray order of ray_sort.hip; ray_tmax: per-ray bound = the any-hit form for shadow rays.
void ptw_launch_extend(const ExtendPlan &pl, pt_scene *s, const float4 *rayA, const float2 *rayB, float4 *hit, uint32_t *hit_inst,
                       const uint32_t *count_in, uint32_t *count_zero, unsigned long long *stats, float tmin, float tmax, bool count,
                       bool raw_hit, hipStream_t st, int pipe = 0, hipEvent_t ev0 = nullptr, hipEvent_t ev1 = nullptr,
                       const uint32_t *perm = nullptr, const float *ray_tmax = nullptr);

// ---- shade_kernels.hip -----------------------------------------------------------------------------------------------
// everything a k_shade launch takes; `in` / `out` / the counts change from round to round, the rest per batch
struct ShadeLaunch {
    ptw::RenderConst rc;
    const uint32_t *tiles = nullptr;
    const pt_scene *scene = nullptr;   // per-triangle tables (LDS-sized or the 64-B records), instances
    bool bvh8 = false;                 // ... in the 8-wide tree's triangle order
    bool lds_tables = false, nee = false;
    int grid = 0;
    size_t smem = 0;
    ptw::Radiance rad{};
    const float4 *hit = nullptr;
    const uint32_t *hit_inst = nullptr;
    ptw::QueueView in{}, out{};
    const uint32_t *count_in = nullptr;
    uint32_t *count_out = nullptr;
    const float4 *inst_frame = nullptr;  // world-space normal + tangent per (instance, triangle), or null
    const float4 *lights = nullptr;      // NEE
    uint32_t n_lights = 0;
    float light_area = 0.f;
    ptw::ShadowQueue sq{};
    uint32_t *sq_count = nullptr;
};
void ptw_launch_shade(const ShadeLaunch &a, hipStream_t st, hipEvent_t ev0, hipEvent_t ev1);
// (stats: the context's counters -- k_generate counts the camera rays of the slots it finishes without a walk, RenderConst::cull)
void ptw_launch_generate(const ptw::RenderConst &rc, const uint32_t *tiles, uint32_t slot_base, uint32_t n_slots, const ptw::Radiance &rad,
                         const ptw::QueueView &out, uint32_t *count_out, unsigned long long *stats, int num_cus, hipStream_t st);
void ptw_launch_shadow_add(const ptw::RenderConst &rc, const ptw::Radiance &rad, const float4 *sq_hit, const float4 *contrib,
                           const uint32_t *slot, const uint32_t *count, int grid, hipStream_t st);
// (skip_if_set: a device word; the kernel leaves the film alone when it is non-zero -- the fused pipeline's overflow flag, read by the host afterwards)
void ptw_launch_resolve(const ptw::RenderConst &rc, const uint32_t *tiles, const ptw::Radiance &rad, float *film, uint8_t *bgra, hipStream_t st,
                        const unsigned long long *skip_if_set = nullptr);
void ptw_launch_hits_to_api(const float4 *hit, const float4 *tri4, const uint32_t *hit_inst, const uint32_t *inst_id, uint32_t n, pt_hit *out,
                            hipStream_t st);

constexpr size_t PTW_COUNT_WORDS = 16 * 32;  // pt_film::Work::d_count: the wavefront's queue sizes (2 per pipeline) | eight slot counters of the fused kernel, one 128-B line each (sixteen with head + tail slots)

// ---- fused.hip ---------------------------------------------------------------------------------------------------------
struct FusedPlan {
    size_t smem = 0;
    int grid = 0, block = 0, lds_stack = 0, refill = 40;
    bool pairs = true;                   // single-level: the pair-leaf instantiation (ExtendPlan::pairs)
    bool count = false;                  // PT_FLAG_COUNT_VISITS: the instrumented twin (single-level scenes; wave-level block counts)
    bool inst = false;                   // two-level scene: k_fused_inst (fused_inst_kernel.h) around k_extend_inst16's walk
    uint32_t n_tlas_lds = 0;             // ... its TLAS nodes staged in LDS
    uint32_t *spill = nullptr;           // ... its stack entries beyond lds_stack: [levels][grid * block] dwords of the context's spill area
    const float4 *inst_frame = nullptr;  // ... the (instance, triangle) normal + tangent table, or null
};
pt_status ptw_plan_fused(pt_scene *s, const ExtendPlan &pl, float tmin, FusedPlan &fp);
void ptw_launch_fused(const FusedPlan &fp, bool grouped, const ptw::RenderConst &rc, const uint32_t *tiles, const ptw::Radiance &rad,
                      const pt_scene *s, uint32_t n_slots, uint32_t *next_slot, unsigned long long *stats, float tmin, float tmax,
                      hipStream_t st, hipEvent_t ev0, hipEvent_t ev1);

// ---- film_work.hip -----------------------------------------------------------------------------------------------------
struct RenderShape {
    uint32_t lanes = 1, groups = 1, group_size = 1, term_cap = 0, term_pcap = 0;
    bool bounded = false;  // term_cap < group_size * max_depth: a full log is detected and the batch redone ungrouped
    uint32_t tail = 0;     // fused pipeline: one-sample tail slots per (frame, pixel) behind a head slot of spp - tail samples (groups == 1 then)
};
// launch_class: 0 instanced scenes, 1 scenes walked out of L2 / MALL / HBM, 2 single-level scenes in LDS, 3 the fused pipeline
RenderShape ptw_choose_shape(const pt_film *f, const pt_params *p, int launch_class, int shrink = 0);
pt_status ptw_ensure_work(pt_film *f, uint32_t rank, uint32_t world, uint32_t lanes, uint32_t groups, uint32_t term_cap, uint32_t term_pcap,
                          bool queues = true, uint32_t tail = 0);
pt_status ptw_shape_and_work(pt_film *f, const pt_params *p, RenderShape &sh, int launch_class, bool queues = true);
pt_status ptw_tiles_subject_first(pt_film *f, const int32_t rect[4], hipStream_t st);  // fused pipeline: tiles that can see the scene first (film_work.hip)
uint64_t ptw_workspace_bytes(const pt_film *f);
ptw::RenderConst ptw_render_const(const pt_params *p, const pt_film::Work &w, const RenderShape &sh);
uint64_t ptw_valid_local_pixels(const pt_film *f, const pt_params *p);

// ---- extend_hbm.hip / ray_sort.hip (launchers of the kernels compiled with the max-ILP scheduler; the ray sorter) --------
const void *ptw_extend_hbm_fn(bool count, bool rec64);
void ptw_launch_extend_hbm(bool count, bool rec64, int grid, size_t smem, hipStream_t st, hipEvent_t ev0, hipEvent_t ev1,
                           const float4 *wide, const uint2 *wide16, const float *norm_c, const float *norm_s,
                           const float *norm_rs, const float4 *tri4, const float4 *rec64_tab, uint32_t n_wide, uint32_t n_tris,
                           const float4 *rayA, const float2 *rayB, float4 *hit, const uint32_t *count_in, uint32_t *count_zero,
                           unsigned long long *stats, uint2 *spill, uint32_t spill_stride, int refill, float tmin,
                           float tmax, int lds_stack, int raw_hit, const uint32_t *perm, const float *ray_tmax);
const void *ptw_extend8_fn(bool count, bool spills, bool waves7);
void ptw_launch_extend8(bool count, bool spills, bool waves7, int grid, size_t smem, hipStream_t st, hipEvent_t ev0, hipEvent_t ev1, const uint4 *nodes8,
                        const float *norm_c, const float *norm_s, const float *norm_rs, const float4 *tri4, const float4 *rec64, const float4 *rayA,
                        const float2 *rayB, float4 *hit, const uint32_t *count_in, uint32_t *count_zero, unsigned long long *stats,
                        uint2 *spill, uint32_t spill_stride, int refill, float tmin, float tmax, int lds_stack, int raw_hit,
                        const uint32_t *perm, const float *ray_tmax);
size_t ptw_ray_sort_bytes(size_t cap);
const uint32_t *ptw_sort_rays(hipStream_t st, const float4 *rayA, const float2 *rayB, const uint32_t *count, size_t cap,
                              const float *bmin, const float *bmax, int bits, int num_cus, void *scratch);
