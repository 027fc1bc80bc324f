// pt_internal.h -- host-side objects behind the opaque C-ABI handles, and the HBM data layout.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>
#include <vector>

#include "../../include/pt_api.h"

// ---- HBM layout of a scene --------------------------------------------
// All per-triangle arrays are in LBVH leaf order (sorted Morton position), so neighbouring
// leaves are neighbouring memory.
//   tri4   : 3 x float4 per triangle  {v0.xyz, bits(prim id)} {v1.xyz, 0} {v2.xyz, 0}       48 B
//   shade4 : 3 x float4 per triangle  {n.xyz, brdf.r} {brdf.gb, emission.rg} {emission.b,0,0,0} 48 B
//   nodes  : binary LBVH, 4 x float4 per internal node {lmin.xyz,lmax.x} {lmax.yz,rmin.xy}
//            {rmin.z,rmax.xyz} {bits(left), bits(right), 0, 0}; child bit31 = leaf position       64 B
//   wide16 : the same BVH4 with fp16 boxes, 16 dwords per node: lo.x[4] lo.y[4] lo.z[4] hi.x[4] hi.y[4] hi.z[4]
//            as halves (2 dwords each), child[4]                                                      64 B
//   wide   : BVH4 collapsed from it, 8 x float4 per node: lo.x[4] lo.y[4] lo.z[4] hi.x[4] hi.y[4]
//            hi.z[4] child[4] spare; leaf child = LEAF | (count-1)<<28 | first sorted position     128 B
constexpr int PT_N_STATS = 24;  // u64 slots of pt_ctx::d_stats that pt_get_stats reads ...
constexpr int PT_N_BLOCKS = 32;  // ... followed by {wave executions, lanes} of up to this many kernel blocks (pt_get_block_counts; fused_kernel.h FusedBlock)
constexpr int PT_N_STATS_ALL = PT_N_STATS + 2 * PT_N_BLOCKS;
constexpr int PT_MAX_PIPES = 4;  // concurrent wavefront pipelines (streams) per pt_render

struct pt_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    int num_cus = 256;
    std::string err;
    // statistics block in device memory (u64 x 8): [0] rays, [1] unused, [2] BVH4 nodes visited, [3] triangles tested, [4] wave steps of the node code, [5] of the triangle code, [6] term-log overflow flag, [7] term pool fill, [8..12] wave executions of refill / pop iteration / hit block / finish / outer iteration, [13] lanes in leaf steps, [14] lanes in pop iterations, [15] lanes in divide blocks, [16] wave executions of the instance entry, [17] lanes in them
    unsigned long long *d_stats = nullptr;
    void *d_rad = nullptr;                 // ptw::Radiance of the fused launch in device memory (wavefront_types.h: Radiance::dev) ...
    unsigned char h_rad[128] = {};         // ... and what it holds (uploaded when it changes: the film's buffers only move when they grow)
    pt_stats stats{};
    hipEvent_t ev_a = nullptr, ev_b = nullptr;
    hipStream_t pipe_stream[PT_MAX_PIPES] = {};  // extra pipelines of pt_render ([0] unused: that is `stream`)
    hipEvent_t ev_fork = nullptr, ev_join[PT_MAX_PIPES] = {};
    // live-count polls of the round loop (render.hip): per pipeline two pinned words and two events, used alternately,
    // so the host reads the count of eight rounds ago while the stream still holds eight rounds of work
    uint32_t *h_poll = nullptr;                      // [PT_MAX_PIPES][2], hipHostMalloc
    hipEvent_t ev_poll[PT_MAX_PIPES][2] = {};
    hipEvent_t ev_shade[PT_MAX_PIPES] = {};          // end of each pipeline's most recent shade launch (the shade rule of three pipelines)
    std::vector<hipEvent_t> ev_pool;  // PT_FLAG_PROFILE start/stop events, reused across pt_render calls
    void *d_spill = nullptr;   // HBM overflow of the traversal short stack: [level][thread] uint2
    size_t spill_bytes = 0;
    // Upper bound on the workspace of a film (0 = what hipMemGetInfo reports as free).  pt_tuning.mem_budget_mb / the environment variable
    // PT_MEM_BUDGET_MB at pt_ctx_create; left alone (-1) the library plans within PT_DEFAULT_MEM_BUDGET_MB: a scene of 400 MB should not
    // take 25 GB of the device without being asked (what each GB buys: profiles/r06e_mem_budget.log).
    size_t mem_budget = 0;
    pt_tuning tune;            // include/pt_api.h: defaults (-1) + PT_TUNE, filled once by pt_ctx_create
    // fused.hip: the launch attributes / occupancy of the fused kernels for the last LDS size planned ([0] single-level, [1] two-level):
    // a blocking call per frame plans twice (PT_PIPELINE_AUTO's look, then the render) and should not pay five runtime calls each time
    size_t fused_smem[2] = { 0, 0 };
    int fused_per_cu[2] = { 0, 0 };
};
// The workspace budget a context plans within when the caller names none: 8 GB.  Round 6, one MI355X, Grays/s at 2 / 4 / 8 / 16 / 32 GB / no
// bound: the 10 000-instance grid through the queues 13.6 / 15.0 / 15.0 / 15.1 / 15.4 / 15.3 (the fused kernel 15.5 in 0.6 GB at every budget);
// the 1 M-triangle soup 2.30 / 2.30 / 3.10 / 3.31 / 3.44 / 3.43; the 8 M-triangle soup 1.89 / 2.96 / 3.18 / 3.29 / 3.27 / 3.28
// (profiles/r06e_mem_budget.log).  A caller who wants the last 3 .. 10 % on big scenes says so (mem_budget_mb = 32768, or 0 for no bound).
constexpr int PT_DEFAULT_MEM_BUDGET_MB = 8192;
inline size_t pt_budget_bytes(int32_t mb) { return mb > 0 ? (size_t)mb << 20 : mb == 0 ? 0 : (size_t)PT_DEFAULT_MEM_BUDGET_MB << 20; }
// a tuning field with its built-in choice and its valid range
inline int pt_tuned(int32_t v, int dflt, int lo, int hi) { return v < 0 ? dflt : (v < lo ? lo : (v > hi ? hi : v)); }

struct pt_scene {
    pt_ctx *ctx = nullptr;
    uint32_t n_tris = 0, n_nodes = 0, height = 0;   // height: of the binary LBVH (read-back)
    uint32_t height_tree = 0;                       // height of the binary tree the traversed BVH4 was collapsed from (LBVH or PLOC)
    float bmin[3]{}, bmax[3]{};
    float build_ms = 0.f;
    float4 *d_tri4 = nullptr;
    float4 *d_shade4 = nullptr;
    float4 *d_shade64 = nullptr;  // 4 x float4 per leaf position {v0, n.x} {v1, n.y} {v2, n.z} {brdf, emits ? 1 : 0}: what k_shade gathers when the tables are not in LDS
    float4 *d_ke4 = nullptr;      // float4 per leaf position {Ke, 0}: read for emitters only
    float4 *d_frame4 = nullptr;   // 2 x float4 per leaf position {T.xyz, B.x} {B.yz, 0, 0}: the tangent frame of the normal (raygen.rgen:14-21), for k_shade's LDS tables
    float4 *d_nodes = nullptr;            // binary LBVH (parity read-back; the collapse reads it)
    float4 *d_wide = nullptr;             // BVH4, 8 x float4 = 128 B per node: what traversal walks
    uint32_t n_wide = 0;
    uint32_t stack_need = 0xFFFFFFFFu;    // exact bound of pending traversal-stack entries (small scenes), else unknown
    // BVH quality (main.cpp:419 asks the driver for ePreferFastTrace).  d_wide / n_wide / stack_need above and
    // the order of tri4/shade4 describe the BVH4 that is TRAVERSED: the collapsed LBVH (builder 0) or, for
    // scenes of <= PT_SAH_MAX_TRIS triangles, a surface-area sweep built on the device (builder 1, bvh4_sah_device.hip).
    // d_wide aliases one of the two owned arrays below.
    uint32_t bvh4_builder = 0;            // 0 collapsed LBVH, 1 surface-area sweep (small scenes), 2 PLOC rebuild of the binary tree (big scenes)
    uint32_t quality = 0;                 // pt_bvh_quality the products were built for
    bool broken = false;                  // a rebuild of the tree products failed (lbvh_build.hip build_tree_products): nothing to traverse
    double area_lbvh = 0.0, area_ploc = 0.0;  // big scenes: sum of the internal nodes' surface areas of the two binary trees (0 = not built)
    // 64-B copy of the traversed BVH4 for scenes that are walked in HBM/L2 (lbvh_build.hip make_wide16):
    // boxes as fp16 of coordinates normalised to the scene box, rounded outwards; halves the bytes per node
    uint2 *d_wide16 = nullptr;
    float norm_c[3]{}, norm_s[3]{1.f, 1.f, 1.f}, norm_rs[3]{1.f, 1.f, 1.f};  // x' = (x - c) * rs,  s = 1/rs
    // BVH8 of the same LBVH for scenes walked out of L2 / MALL / HBM (lbvh_build.hip k_w8_*): 128-B nodes, and the
    // per-triangle tables in ITS triangle order (a node's leaf triangles are contiguous), used instead of
    // d_tri4 / d_shade64 / d_ke4 whenever the BVH8 kernel traverses
    uint4 *d_wide8 = nullptr;
    uint32_t n_wide8 = 0, levels8 = 0;
    // BVH4 in the 64-B format built top-down (area-guided collapse, the children of a node contiguous): what
    // k_extend<hbm> walks for scenes whose traversed BVH4 is the collapsed LBVH (lbvh_build.hip k_w4_emit)
    uint4 *d_wide16t = nullptr;
    uint32_t n_wide16t = 0, levels4t = 0;
    uint32_t *d_prim_of8 = nullptr;
    float4 *d_tri4_8 = nullptr, *d_shade64_8 = nullptr, *d_ke4_8 = nullptr;
    uint64_t device_bytes8 = 0;
    float4 *d_wide_lbvh = nullptr; uint32_t n_wide_lbvh = 0, stack_need_lbvh = 0xFFFFFFFFu;
    float4 *d_wide_sah = nullptr;  uint32_t n_wide_sah = 0, stack_need_sah = 0xFFFFFFFFu;
    uint32_t *d_prim_of_sah = nullptr;      // leaf order of the SAH BVH4 (position -> prim id)
    float4 *d_tri_orig = nullptr;           // small scenes keep the unsorted triangles + materials so the
    float *d_faces = nullptr;               // per-triangle tables can be re-packed in another leaf order
    std::vector<float> h_tlo, h_thi;        // small scenes: unpadded triangle boxes (3 floats each)
    std::vector<uint8_t> h_pair;            // small scenes: [n] triangle i+1 = (v0, v2, v3) of the quad whose (v0, v1, v2) is triangle i
    bool sah_pair_leaves = false;           // ... of the surface-area BVH4 (d_wide_sah)
    bool pair_leaves = false;               // the traversed BVH4 has exactly one primitive (triangle or such a pair) per leaf
    unsigned long long *d_keys = nullptr;  // sorted Morton keys (kept for parity read-back)
    uint32_t *d_prim_of = nullptr;         // sorted position -> prim id
    uint64_t device_bytes = 0;
    // emitters (Ke != 0) in primitive order for PT_PIPELINE_WAVEFRONT_NEE: 5 float4 each {A, cdf} {B, 0} {C, 0} {N, 0} {Ke, 0}
    // (cdf = running float sum of the triangle areas, light_area its total)
    float4 *d_lights = nullptr;
    uint32_t n_lights = 0;
    float light_area = 0.f;
    std::vector<float4> h_lights;   // host copy of d_lights: instancing makes its world-space copies from it
    // instanced scenes: every instance's emitters in world space, gl_InstanceID-major, same 5-float4 layout and running cdf
    std::vector<float> h_xforms;    // the instance set's object->world matrices as given (gl_InstanceID order): ptb_ensure_inst_lights
    float4 *d_lights_inst = nullptr;  // built on the first NEE render of the instance set
    uint32_t n_lights_inst = 0;
    float light_area_inst = 0.f;
    // two-level scenes: instances in TLAS leaf order, 6 float4 each {object->world rows, world->object rows}
    uint32_t n_inst = 0, n_tlas_wide = 0, tlas_height = 0;
    double tlas_area_lbvh = 0.0, tlas_area_ploc = 0.0;  // area sums of the TLAS's binary trees (ploc: 0 = not built)
    float4 *d_inst6 = nullptr;
    float4 *d_inst_frame = nullptr;  // [n_inst][n_tris][2]: world-space normal + tangent of every instanced triangle (ptb_ensure_inst_frames)
    float4 *d_tlas_wide = nullptr;        // BVH4 over the instances' world boxes
    uint32_t *d_tlas_prim_of = nullptr;   // sorted position -> instance id (gl_InstanceID)
    // the TLAS k_extend_inst16 walks: 64-B fp16 nodes (normalised to the TLAS box), built top-down, 16-bit child codes
    uint4 *d_tlas16 = nullptr;
    uint32_t n_tlas16 = 0, tlas16_levels = 0;
    float tlas_norm_c[3]{}, tlas_norm_s[3]{1.f, 1.f, 1.f}, tlas_norm_rs[3]{1.f, 1.f, 1.f};
    float tlas_bmin[3]{}, tlas_bmax[3]{};  // the union of the instances' world boxes (k_inst_boxes): what no ray outside of can hit
};

struct pt_film {
    pt_ctx *ctx = nullptr;
    uint32_t w = 0, h = 0;
    float *d_rgb = nullptr;   // w*h*3 running mean (canonical float film)
    bool own_rgb = true;
    uint8_t *d_bgra = nullptr;  // w*h*4 reference-display image
    // wavefront workspace, (re)allocated by pt_render for (rank, world, frames_in_flight)
    struct Work {
        uint32_t rank = 0, world = 0, lanes = 0;  // lanes = frames in flight
        uint32_t tile_order = 0;                  // 0: the rank's tiles row by row, 1: centre first (fused pipeline; film_work.hip)
        uint32_t groups = 0, term_cap = 0;        // sample groups per pixel, radiance-term log capacity per slot
        uint32_t tail = 0;                        // one-sample tail slots per (frame, pixel) of the last shape (fused head + tail form)
        uint32_t n_tiles = 0;                     // local 8x8 tiles
        uint32_t n_slots = 0;                     // lanes * n_tiles * 64
        uint32_t *d_tiles = nullptr;              // local tile -> global tile id
        std::vector<uint32_t> h_tiles;            // tile_order 1: the centre-first list as built (ptw_tiles_subject_first re-orders d_tiles from it)
        int32_t tile_rect[4] = { 0, 0, -1, -1 };  // ... the pixel rectangle {x0, y0, x1, y1} whose tiles currently go first in d_tiles (x1 < x0: none)
        float4 *d_color = nullptr;                // per slot: frame colour accumulator rgb + pad (groups == 1)
        float4 *d_terms = nullptr;                // per slot: ordered radiance terms, dense primary log   (groups > 1)
        float4 *d_terms_over = nullptr;           // per slot: overflow of the primary log (worst-case sized)
        uint32_t *d_nterm = nullptr;              // per slot: number of logged terms               (groups > 1)
        uint32_t *d_spill_head = nullptr;         // per slot: last entry of the slot in the spill pool (groups > 1)
        float4 *d_spill = nullptr;                // shared pool {r, g, b, previous entry of the slot}: terms beyond term_cap
        // double-buffered dense queues (index = queue position)
        uint2 *d_qid[2] = { nullptr, nullptr };       // {slot, sample | depth<<16}
        float4 *d_qstate[2] = { nullptr, nullptr };   // {bits(seed), weight.rgb}
        float4 *d_qrayA[2] = { nullptr, nullptr };    // {org.xyz, dir.x}
        float2 *d_qrayB[2] = { nullptr, nullptr };    // {dir.y, dir.z}
        float4 *d_hit = nullptr;                      // {bits(pos), t, u, v}
        uint32_t *d_hit_inst = nullptr;               // instance (TLAS sorted position); only for two-level scenes
        uint32_t *d_count = nullptr;                  // [2] queue sizes
        size_t cap_slots = 0, cap_meta = 0, cap_color = 0, cap_terms = 0, cap_terms_over = 0;  // allocated capacities (buffers only grow); cap_slots: queues + hit records, cap_meta: d_nterm / d_spill_head
        size_t bytes = 0;                             // device bytes held by the buffers above
        // shadow queue of the NEE pipeline (one entry per hit whose light sample faces it)
        float4 *d_sq_rayA = nullptr; float2 *d_sq_rayB = nullptr; float4 *d_sq_contrib = nullptr;  // {r, g, b, tmax}
        uint32_t *d_sq_slot = nullptr; float *d_sq_tmax = nullptr; float4 *d_sq_hit = nullptr; uint32_t *d_sq_count = nullptr;
        size_t cap_sq = 0;
        void *d_sort = nullptr;                       // ray_sort.hip scratch for all pipelines (ptw_ray_sort_bytes per slot range)
        size_t sort_bytes = 0;
    } work;
};

#define PT_HIP(ctx, call)                                                                         \
    do {                                                                                          \
        hipError_t e__ = (call);                                                                  \
        if (e__ != hipSuccess) {                                                                  \
            (ctx)->err = std::string(#call) + ": " + hipGetErrorString(e__);                      \
            return PT_ERR_HIP;                                                                    \
        }                                                                                         \
    } while (0)

#define PT_BROKEN_SCENE_MSG "the scene lost its acceleration structure in a failed rebuild (out of memory?): call pt_scene_set_bvh_quality again, or recreate it"
// lbvh_build.hip
pt_status ptb_build_scene(pt_scene *s, const float *h_vertices, uint32_t n_verts, const uint32_t *h_indices,
                          uint32_t n_tris, const float *h_faces);
pt_status ptb_set_instances(pt_scene *s, const float *xforms3x4, uint32_t n);
pt_status ptb_set_bvh_quality(pt_scene *s, uint32_t quality);
void ptb_free_scene_buffers(pt_scene *s);
pt_status ptb_ensure_inst_lights(pt_scene *s);  // world-space emitter copies of an instanced scene (NEE pipeline only)
constexpr uint64_t PT_SOURCE_BYTES_PER_TRI = 72;  // d_tri_orig + d_faces, kept for rebuilds: part of pt_scene_info.device_bytes
pt_status ptb_ensure_inst_frames(pt_scene *s);  // the table k_shade reads instead of transforming the normal per hit (instanced scenes)
pt_status ptb_repair(pt_scene *s);        // no-op unless a rebuild of the tree products failed earlier: then one more try
pt_status ptb_ensure_wide8(pt_scene *s);  // builds the 8-wide nodes of a scene that was created without them
constexpr uint32_t PT_SAH_MAX_TRIS = 2048;
// bvh4_sah_device.hip: surface-area sweep on the device (one workgroup) -> BVH4 rows (32 dwords each) + leaf order
// pair_with_next (nullable): [n] flags, triangle i and i+1 are the two halves (v0,v1,v2),(v0,v2,v3) of a quad and form ONE primitive
uint32_t pt_wide_stack_need(const std::vector<uint32_t> &rows32);
pt_status pt_sah_build_bvh4_device(pt_ctx *ctx, const float *tlo, const float *thi, uint32_t n, const uint8_t *pair_with_next, float pad,
                                   uint32_t leaf_max_prims, std::vector<uint32_t> &rows32, std::vector<uint32_t> &order);
void ptb_free_instances(pt_scene *s);
// render.hip / film_work.hip
pt_status ptw_render(pt_scene *s, pt_film *f, const pt_params *p);
pt_status ptw_prepare(pt_scene *s, pt_film *f, const pt_params *p);
pt_status ptw_trace(pt_scene *s, const float *rays6, uint32_t n, float tmin, float tmax, uint32_t extend, pt_hit *hits);
void ptw_free_work(pt_film *f);
