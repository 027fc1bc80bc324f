/*
 * pt_api.h -- C-ABI of the MI355X-native wavefront path tracer (libpt_amd.so).
 *
 * This is the drop-in boundary for the reference's per-pixel radiance loop.  The reference
 * (yknishidate/single-file-vulkan-pathtracing) has no FFI layer: its operator boundary is the
 * Vulkan dispatch itself, so every entry point below names the Vulkan-side interface it
 * replaces (file:line in the reference).  Plain pointers and sizes only; no C++/torch types;
 * no exceptions or aborts cross this boundary -- every call returns a pt_status and
 * pt_last_error() gives the text (the reference throws std::runtime_error, main.cpp:35, 118,
 * 151, 221, 594, 612, 681).
 *
 * Threading: calls on one context are serialised by the caller, as in the reference (single
 * host thread, single queue, main.cpp:224-238, 683).  One context per GPU; multi-GPU runs use
 * one process (or thread) per GPU, each rendering its interleaved pixel tiles.
 */
#ifndef PT_API_H
#define PT_API_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PT_API_VERSION 6  /* 6: pt_get_block_counts (PT_FLAG_COUNT_VISITS on PT_PIPELINE_FUSED: the instrumented single-level fused kernel).
                           * 5: PT_PIPELINE_AUTO (what pt_params_default returns); pt_stats.pipeline / .tail_samples / .rays_culled (appended); pt_tuning.fused_tail /
                           *    .fused_subject / .cull (from the reserved words); pt_device_write.  4: PT_PIPELINE_FUSED; pt_tuning.tlas_ploc / ploc_adopt_pct / fail_rebuild */

typedef enum pt_status {
    PT_OK = 0,
    PT_ERR_INVALID_ARG = 1,
    PT_ERR_NO_DEVICE = 2,   /* no HIP device / HIP runtime unusable (replaces main.cpp:105,118) */
    PT_ERR_HIP = 3,         /* a HIP call failed; see pt_last_error                            */
    PT_ERR_OOM = 4,
    PT_ERR_UNSUPPORTED = 5
} pt_status;

typedef struct pt_ctx pt_ctx;     /* replaces Context (main.cpp:74-267): device + queue      */
typedef struct pt_scene pt_scene; /* replaces vertex/index/face Buffers + BLAS + TLAS        */
typedef struct pt_film pt_film;   /* replaces the storage Image (main.cpp:481-484)            */

/* ---- context ------------------------------------------------------------------------- */
/* device: HIP ordinal (the reference takes physical device 0, main.cpp:105).
 * stream: a hipStream_t to launch on, or NULL for the library's own stream.  The caller
 *         keeps ownership of a stream it passes in.                                        */
pt_status pt_ctx_create(int device, void *stream, pt_ctx **out);
void pt_ctx_destroy(pt_ctx *ctx);
/* Tuning knobs of a context: launch shapes and kernel choices that change SPEED, never results (every combination is
 * covered by the bit-exact parity tests).  -1 = the built-in choice, which is what was measured best on MI355X
 * (DESIGN.md section 6 has the numbers).  pt_ctx_create fills the defaults and then applies the environment variable
 * PT_TUNE once ("name=value,name=value", names as below) -- the library reads no other tuning from the environment,
 * and nothing at all inside pt_render.  Set before pt_scene_create (pair_leaves is read when a scene's BVH4 is built). */
typedef struct pt_tuning {
    int32_t refill;         /* idle lanes of a wave before it takes new rays (1..64)                                   */
    int32_t lds_stack;      /* traversal-stack entries per lane kept in LDS (kernels with a spill path)                */
    int32_t extend_blocks;  /* cap on persistent extend blocks per CU                                                  */
    int32_t pipes;          /* concurrent wavefront pipelines (streams) per pt_render, 1..4                            */
    int32_t stagger;        /* 0: free-running pipelines; 1: started half a round apart (2 pipelines); 2: shade rule (3)  */
    int32_t sort_bits;      /* ray sorting: Morton bits per axis of the origin cell (1..9)                             */
    int32_t pair_leaves;    /* 0: small scenes get leaves of <= 4 independent triangles instead of one primitive each  */
    int32_t pair_kernel;    /* 0: the per-triangle leaf loop over a pair-leaf tree instead of the pair test            */
    int32_t topdown4;       /* 0: the HBM kernel walks the collapsed LBVH instead of the top-down BVH4                 */
    int32_t rec64;          /* 0: the HBM kernel reads leaves from the 48-B tri4 records instead of the 64-B ones      */
    int32_t inst16;         /* 0: instanced scenes always take the general two-level kernel (32-bit child words)       */
    int32_t inst16_blocks;  /* compact two-level kernel: persistent blocks per CU                                      */
    int32_t enter_min;      /* ... lanes that wait to enter an instance together (1..64)                               */
    int32_t node_yield;     /* ... the node loop yields below 1/N descending lanes (0 = never)                         */
    int32_t tlas_lds_kb;    /* ... KB of TLAS top levels staged in LDS                                                 */
    int32_t term_ocap;      /* tests: cap on the per-slot overflow term log (entries)                                  */
    int32_t term_spill;     /* tests: cap on the shared term pool (entries)                                            */
    int32_t mem_budget_mb;  /* upper bound on a film's workspace, MB (also env PT_MEM_BUDGET_MB); 0 = none; -1 = 8192  */
    int32_t hbm8;           /* 1: AUTO walks big scenes through the 8-wide compressed nodes (PT_EXTEND_HBM8)           */
    int32_t ploc_radius;    /* PLOC rebuild of big scenes' binary tree: neighbours searched on either side (1..32, 8)  */
    int32_t leaf_min;       /* compact two-level kernel: lanes that wait with a triangle leaf before the leaf step runs */
    int32_t tri_enter;      /* 8-wide tree kernel: lanes that wait with leaf triangles before a triangle step runs     */
    int32_t tri_stay;       /* ... and triangle steps repeat while at least this many lanes still hold one (65 = never) */
    int32_t inst_frames;    /* 0: instanced scenes transform the normal and build its tangent frame per hit instead of reading
                               the per-(instance, triangle) table                                                         */
    int32_t tlas_ploc;      /* 1: the TLAS of an instanced scene is rebuilt by PLOC like a big scene's binary tree (0: LBVH)   */
    int32_t ploc_adopt_pct; /* a PLOC tree is kept when its area sum is below this percentage of the LBVH's (90; 1000 = always) */
    int32_t fail_rebuild;   /* NOT a speed knob -- failure injection for the tests: > 0 makes the next rebuilds of a scene's tree products fail
                               after the old ones were freed.  Only pt_ctx_set_tuning sets it; PT_TUNE refuses the name.                      */
    int32_t fused_tail;     /* fused pipeline, sample_groups left at 0, single-level scenes: S of a pixel's spp samples are traced as one-sample
                               tail slots handed out after every head slot (spp - S samples) -- a launch with few slots per lane ends with short
                               work.  0 = never; -1: by the launch's slots per lane (render.hip fused_tail_samples); clamped to spp - 1.      */
    int32_t fused_subject;  /* fused pipeline: 0 = hand the tiles out centre first only; -1 / 1: the tiles the scene's box projects to first (render.hip) */
    int32_t cull;           /* 0 = walk every camera ray; -1 / 1: the slots of pixels outside the projection of the scene's box (two-level: of the instances'
                               boxes) are finished without a walk -- each of their samples is one counted ray that misses (pt_stats.rays_culled).
                               Every pipeline; with PT_FLAG_COUNT_VISITS (the walk of every ray is measured) only when set to 1             */
    int32_t reserved[2];
} pt_tuning;
pt_status pt_ctx_get_tuning(const pt_ctx *ctx, pt_tuning *out);
pt_status pt_ctx_set_tuning(pt_ctx *ctx, const pt_tuning *in);
/* Last error text of this context (ctx may be NULL: text of the last failed pt_ctx_create). */
const char *pt_last_error(const pt_ctx *ctx);
/* Blocks until everything queued on the context's stream is done (queue.waitIdle, main.cpp:683). */
pt_status pt_sync(pt_ctx *ctx);

/* ---- scene: descriptor bindings 0,2,3,4 (main.cpp:561-567, 628-641) -------------------- */
/* Takes the exact arrays the reference uploads (main.cpp:492-494):
 *   vertices f32[3*n_verts]  (closesthit.rchit:6, stride 3, closesthit.rchit:24-31)
 *   indices  u32[3*n_tris]   (closesthit.rchit:7)
 *   faces    f32[6*n_tris]   (closesthit.rchit:8: Kd.rgb, Ke.rgb, stride 6, :33-41)
 * Inputs are copied (main.cpp:321-325); the caller keeps its arrays.  Builds the LBVH on the
 * device (Morton keys + radix sort + Karras hierarchy + refit): the replacement of the BLAS/
 * TLAS builds at main.cpp:496-538 (one identity instance, opaque, no culling).              */
pt_status pt_scene_create(pt_ctx *ctx, const float *vertices, uint32_t n_verts,
                          const uint32_t *indices, uint32_t n_tris, const float *faces,
                          pt_scene **out);
void pt_scene_destroy(pt_scene *scene);

/* Two-level scenes (BASELINE config C4).  The reference builds exactly ONE identity instance
 * (main.cpp:515-538: VkAccelerationStructureInstanceKHR with an identity 3x4, mask 0xFF,
 * TriangleFacingCullDisable); this sets n instances of the scene's geometry instead.
 * xforms3x4: n object->world matrices, 3x4 row major (VkTransformMatrixKHR layout), copied.
 * n = 0 restores the single-level scene.  A TLAS (BVH4 over the instances' world boxes) is built
 * on the device.  Hits report gl_InstanceID in pt_hit.inst; ties in t go to the lowest
 * (instance, primitive).  Shading transforms the hit position by the matrix and the normal by the
 * inverse transpose (renormalised); materials are per primitive, shared by all instances.      */
pt_status pt_scene_set_instances(pt_scene *scene, const float *xforms3x4, uint32_t n);

typedef struct pt_scene_info {
    uint32_t n_tris, n_nodes /* binary LBVH */, bvh_height /* of the binary LBVH */;
    uint32_t n_wide_nodes;    /* BVH4 nodes (128 B each) the traversal kernels walk               */
    uint32_t n_instances;     /* 0 = single-level scene                                           */
    uint32_t n_tlas_nodes;    /* BVH4 nodes of the TLAS                                           */
    uint32_t leaf_max;        /* triangles per BVH4 leaf of the collapse rule (bvh4 read-back)    */
    uint32_t bvh4_builder;    /* which BVH4 is traversed: 0 collapsed LBVH, 1 surface-area sweep (small scenes), 2 PLOC tree */
    float    bbox_min[3], bbox_max[3];
    float    build_ms;        /* device time of the LBVH build (reported apart from rendering) */
    uint64_t device_bytes;    /* resident scene + BVH bytes of the BVH4 path, incl. the source arrays kept for rebuilds (72 B per triangle) */
    uint32_t n_wide8_nodes;   /* 8-wide nodes (64 B each: byte planes) of the PT_EXTEND_HBM8 path, levels of that tree */
    uint32_t wide8_levels;
    uint64_t device_bytes8;   /* resident triangle tables + BVH8 bytes of that path             */
    /* big scenes (> 2048 triangles): sum of the surface areas of the binary tree's internal nodes over the root's, for
     * the Morton-median LBVH and for its PLOC rebuild (0: not built -- FAST_BUILD, or small scene).  FAST_TRACE keeps
     * the rebuild when its sum is below 0.9 of the LBVH's (bvh4_builder says which tree is traversed).                                                      */
    float    tree_area_lbvh, tree_area_ploc;
} pt_scene_info;
pt_status pt_scene_get_info(const pt_scene *scene, pt_scene_info *info);

/* Build quality, the counterpart of vk::BuildAccelerationStructureFlagBitsKHR (main.cpp:419 passes
 * ePreferFastTrace, which is the default here too).  FAST_TRACE: scenes of <= 2048 triangles get their
 * BVH4 from an exhaustive surface-area sweep (one workgroup on the device) -- about 1/6 less traversal work
 * on the Cornell box; larger scenes get the LBVH's binary tree rebuilt bottom-up by parallel locally-ordered
 * clustering (PLOC, radius 8) before the wide nodes are collapsed from it -- what makes a finely tessellated
 * object in a large room cheap to walk.  FAST_BUILD: always the collapsed LBVH (Morton median splits).  Hit
 * records and images do not depend on the choice (closest t, lowest primitive id).  Changing the quality of
 * a big scene rebuilds its tree.  Call before pt_scene_set_instances.                                       */
typedef enum pt_bvh_quality { PT_BVH_PREFER_FAST_TRACE = 0, PT_BVH_PREFER_FAST_BUILD = 1 } pt_bvh_quality;
pt_status pt_scene_set_bvh_quality(pt_scene *scene, uint32_t quality);

/* Debug/parity read-back of the device-built LBVH.  keys/prim_of_pos: n_tris entries each;
 * nodes16: n_nodes x 16 dwords {lmin[3] lmax[3] rmin[3] rmax[3] left right 0 0}, child bit31 =
 * leaf (sorted position).  Any pointer may be NULL.                                         */
pt_status pt_scene_read_bvh(const pt_scene *scene, uint64_t *keys, uint32_t *prim_of_pos,
                            uint32_t *nodes16);
/* The BVH4 that is traversed (collapsed from that LBVH, or the surface-area one): n_wide_nodes x 32 dwords {lo.x[4] lo.y[4] lo.z[4] hi.x[4] hi.y[4]
 * hi.z[4] child[4] 0[4]}; child = 0xFFFFFFFF empty | node index | bit31: leaf,
 * (count-1)<<28 | first sorted position.                                                     */
pt_status pt_scene_read_bvh4(const pt_scene *scene, uint32_t *nodes32);
/* The 8-wide tree (PT_EXTEND_HBM8; big scenes build it on first request): n_wide8_nodes x 16 dwords = 64 B per node --
 * dwords 0..11: the rows lo.x lo.y lo.z hi.x hi.y hi.z of the eight children's boxes, one BYTE per child (two dwords per
 * row, child k in byte k & 3 of dword k >> 2); dword 12: origin.x | origin.y << 16; dword 13: origin.z | ex << 16 |
 * ey << 21 | ez << 26; dword 14: child_base | imask << 24; dword 15: tri_base | lmask << 24.  A plane is
 * origin16 * 2^-14 - 2 + byte * 2^-e in coordinates (x - c) / s normalised to the scene box (c, s: its centre and half
 * extent); lower planes are rounded down and upper planes up; an empty slot is lo = 255, hi = 0.  Internal children are
 * the nodes child_base + (rank of the slot in imask), leaf children the triangle positions tri_base + (rank in lmask).
 * prim_of_pos8: n_tris entries, 8-wide triangle position -> gl_PrimitiveID.  Either pointer may be NULL.               */
pt_status pt_scene_read_bvh8(const pt_scene *scene, uint32_t *nodes32, uint32_t *prim_of_pos8);

/* ---- film: descriptor binding 1 (raygen.rgen:7, main.cpp:481-484) ---------------------- */
/* float32 running-mean radiance (the canonical result) plus the reference's rgba8 display
 * image (B,G,R,A bytes, clamped + quantised on every frame like raygen.rgen:88-90).         */
pt_status pt_film_create(pt_ctx *ctx, uint32_t width, uint32_t height, pt_film **out);
/* Same, but the float film lives in caller-owned DEVICE memory (width*height*3 floats), e.g.
 * a torch tensor's data_ptr(), so a collective can reduce it in place.                      */
pt_status pt_film_create_external(pt_ctx *ctx, uint32_t width, uint32_t height,
                                  void *device_rgb_f32, pt_film **out);
pt_status pt_film_clear(pt_film *film);
/* rgb: width*height*3 floats, row-major, linear radiance mean over all frames so far.       */
pt_status pt_film_read_f32(pt_film *film, float *rgb);
/* bgra: width*height*4 bytes = what main.cpp:661-667 copies to the swapchain.               */
pt_status pt_film_read_bgra8(pt_film *film, uint8_t *bgra);
void pt_film_destroy(pt_film *film);

/* ---- dispatch: pushConstants + traceRaysKHR (main.cpp:656-659) ------------------------- */
enum {
    PT_PIPELINE_WAVEFRONT = 0,     /* generate / extend / shade queues: the reference's estimator, bit for bit        */
    /* NOT the reference's estimator (opt-in, no parity with the reference's images at equal sample counts, only in
     * expectation): next-event estimation.  At every hit one point on one emitter (chosen by area) is sampled and a
     * SHADOW ray queued -- a third queue, compacted like the others and traced by the same extend kernels as an
     * any-hit query; the emission of a surface the path runs into counts for camera rays only.  Same random stream
     * otherwise (three more numbers per hit).  Fully specified arithmetic like the reference path's (the tests' CPU checker restates it bit for bit).  Instanced scenes sample every instance's copy of the emitters (world space, one cdf). */
    PT_PIPELINE_WAVEFRONT_NEE = 1,
    /* The reference's estimator, bit for bit, as ONE persistent kernel -- the shape of the reference's own raygen shader
     * (raygen.rgen:41-91: one invocation owns its path): traversal and shading in the same lane, path state in LDS, no
     * queues in HBM; the workspace is the per-slot radiance only (16 B per slot instead of ~150).  For scenes whose
     * triangles fit LDS: single-level ones (the Cornell-box class) and instanced ones of 2 .. 32767 instances over such a
     * BLAS (the TLAS stays in L2); PT_ERR_UNSUPPORTED otherwise, tmin > 0, blocking calls only.  Same films, same ray
     * counts as PT_PIPELINE_WAVEFRONT.                                                                                   */
    PT_PIPELINE_FUSED = 2,
    /* What pt_params_default returns: the fastest pipeline that renders the reference's estimator bit for bit for THIS scene and call --
     * PT_PIPELINE_FUSED where it applies (scenes that live in LDS, see above; blocking calls without PT_FLAG_COUNT_VISITS), else
     * PT_PIPELINE_WAVEFRONT.  Films, rgba8 images and ray counts do not depend on the choice; pt_stats.pipeline says which one ran.  */
    PT_PIPELINE_AUTO = 3
};
enum {
    PT_FLAG_PROFILE = 1u,      /* hipEvent-time every extend/shade launch (adds events to the stream)       */
    PT_FLAG_COUNT_VISITS = 2u, /* instrumented traversal: count BVH4 nodes / triangles visited (slower)      */
    PT_FLAG_ASYNC = 4u,        /* pt_render only queues the work (no waitIdle, main.cpp:683); pt_sync and the */
                               /* film read-backs wait for it.  No timing statistics; not with PT_FLAG_PROFILE */
    /* Ray sorting (scenes walked out of HBM): before every extend pass after the first the queue is put in (origin cell,
     * direction octant) order by a device radix sort of a permutation.  AUTO (neither flag): on when the traversal
     * working set (BVH4 nodes + triangles) exceeds the 256 MiB Infinity Cache.  Results do not depend on it.        */
    PT_FLAG_SORT_RAYS = 8u, PT_FLAG_NO_SORT_RAYS = 16u
};
/* Which closest-hit kernel runs.  All variants implement the same closest-hit definition and
 * return identical bits; AUTO picks by scene size. */
enum {
    PT_EXTEND_AUTO = 0,
    PT_EXTEND_FLAT = 1, /* deprecated (until API version 4: a brute-force loop over <= 1024 triangles, never AUTO): the name still compiles, pt_render /
                         * pt_trace return PT_ERR_UNSUPPORTED for it.  The CPU oracle (oracle/, tests only) is the brute-force reference since. */
    PT_EXTEND_FLAT_REMOVED = PT_EXTEND_FLAT,
    PT_EXTEND_LDS = 2,  /* BVH4 + triangles staged in LDS (scenes <= 24 KB by AUTO), lane refill             */
    PT_EXTEND_HBM = 3,  /* BVH4 + triangles read through L1/L2/MALL from HBM, LDS short stack + HBM spill    */
    PT_EXTEND_HBM8 = 4  /* 8-wide tree: 64-B nodes with byte planes, one stack entry per node (AUTO only with pt_tuning.hbm8) */
};

typedef struct pt_params {
    int32_t  frame;            /* push constant `frame` (main.cpp:658, raygen.rgen:8-10): first frame */
    uint32_t frame_count;      /* consecutive frames rendered by this call (reference: 1 per dispatch) */
    uint32_t width, height;    /* launch size (main.cpp:659); must equal the film's                   */
    uint32_t spp_per_frame;    /* maxSamples, 32 (raygen.rgen:43)                                      */
    uint32_t max_depth;        /* 8 (raygen.rgen:62)                                                   */
    float    tmin, tmax;       /* 0.001, 10000 (raygen.rgen:71,73)                                     */
    float    cam_origin[3];    /* (0,-1,5) (raygen.rgen:55)                                            */
    float    cam_target[3];    /* target = (d.x+tx, d.y+ty, tz); (0,-1,2) (raygen.rgen:56)             */
    float    env[3];           /* (0.7,0.6,0.5) (miss.rmiss:10)                                        */
    uint32_t rank, world;      /* this call renders the 8x8 pixel tiles (tx+ty) % world == rank        */
    uint32_t pipeline;         /* PT_PIPELINE_*                                                        */
    uint32_t frames_in_flight; /* frames traced concurrently (0 = auto); results do not depend on it   */
    uint32_t flags;            /* PT_FLAG_*                                                            */
    uint32_t extend;           /* PT_EXTEND_*                                                          */
    uint32_t sample_groups;    /* slots per (frame, pixel) tracing disjoint sample ranges concurrently  */
                               /* (0 = auto); results do not depend on it                              */
} pt_params;
void pt_params_default(pt_params *p); /* the reference's compile-time constants, 1024x1024, world 1, PT_PIPELINE_AUTO */

/* Renders frames [frame, frame+frame_count) into the film: each frame is one reference launch
 * (spp_per_frame samples/pixel, <= max_depth rays each) blended by raygen.rgen:88-90.
 * Blocking (returns after the device is done), like submit + waitIdle (main.cpp:672-683),
 * unless PT_FLAG_ASYNC is set.                                                               */
pt_status pt_render(pt_scene *scene, pt_film *film, const pt_params *params);
/* Allocates (or grows) the film's wavefront workspace for exactly the shape pt_render would pick
 * for these params, without rendering -- so the first timed pt_render does not pay for hipMalloc
 * (the pipeline / descriptor set-up of main.cpp:540-641 plays this role in the reference).
 * pt_get_stats afterwards reports the chosen frames_in_flight / sample_groups.              */
pt_status pt_render_prepare(pt_scene *scene, pt_film *film, const pt_params *params);

/* ---- closest-hit query alone: traceRayEXT (raygen.rgen:63-75) -------------------------- */
typedef struct pt_hit {
    uint32_t prim;  /* gl_PrimitiveID, 0xFFFFFFFF = miss                                   */
    float    t;     /* hit distance (0 on miss)                                             */
    float    u, v;  /* hitAttributeEXT attribs.xy: weights of v1, v2 (closesthit.rchit:56)  */
    uint32_t inst;  /* gl_InstanceID (0 without instances), 0xFFFFFFFF = miss              */
} pt_hit;
/* rays6: host array n x {origin.xyz, direction.xyz}; hits: host array of n.  Runs the same
 * extend kernel the renderer uses (opaque, no culling, tmin < t < tmax).                   */
pt_status pt_trace(pt_scene *scene, const float *rays6, uint32_t n, float tmin, float tmax,
                   uint32_t extend /* PT_EXTEND_* */, pt_hit *hits);

/* ---- multi-GPU: assembling the presented image of a tile-sharded render ---------- */
/* The reference renders on physical device 0 (main.cpp:105) and copies its storage image to the swapchain
 * (main.cpp:661-667).  Here N ranks -- processes or host threads, one context and one GPU each -- render the
 * interleaved 8x8 tiles of one image (pt_params.rank / .world); pt_film_present stands in for that copy: ONE RCCL
 * gather of the packed tiles to the root per presented image (W*H*12/N bytes per rank over xGMI), written to a
 * separate image, so every rank's accumulation film stays valid for further progressive frames.
 * RCCL is loaded at run time (dlopen) by the first pt_comm_* call: PT_ERR_UNSUPPORTED when it is not installed. */
typedef struct pt_comm pt_comm;
typedef struct pt_unique_id { char internal[128]; } pt_unique_id; /* = ncclUniqueId */
/* One rank makes the id (ncclGetUniqueId) and hands the 128 bytes to the others by any means (the launcher's
 * store, MPI, a file); then every rank creates its communicator (ncclCommInitRank; collective: returns when all
 * `world` ranks have called it).  One communicator per context.                                                   */
pt_status pt_comm_unique_id(pt_unique_id *id);
pt_status pt_comm_create(pt_ctx *ctx, const pt_unique_id *id, uint32_t world, uint32_t rank, pt_comm **out);
pt_status pt_comm_ranks(const pt_comm *comm, uint32_t *n); /* ncclCommCount: the ranks RCCL actually connected */
void pt_comm_destroy(pt_comm *comm);
/* Called by every rank once per presented image, with the film it rendered as (rank, world) of the communicator.
 * d_image: DEVICE memory for width*height*3 floats on the root (ignored elsewhere).  Blocking.               */
pt_status pt_film_present(pt_film *film, pt_comm *comm, uint32_t root, float *d_image);
/* The two kernels of that path alone (no communicator): a rank's tiles <-> a dense device buffer
 * [tile][64 pixels][rgb], tiles in row-major order of the tile grid.  pt_film_tile_count gives its length / 192.  */
pt_status pt_film_tile_count(const pt_film *film, uint32_t rank, uint32_t world, uint32_t *n_tiles);
pt_status pt_film_pack_tiles(pt_film *film, uint32_t rank, uint32_t world, float *d_packed);
pt_status pt_film_unpack_tiles(pt_film *film, uint32_t rank, uint32_t world, const float *d_packed, float *d_image);

/* Device memory for the buffers a caller hands to the library (pt_film_create_external, pt_film_present), for hosts
 * that do not link HIP themselves (host/pt_main.cpp is plain g++): hipMalloc / hipFree / blocking copies either way,
 * ordered after the context's stream.                                                                            */
pt_status pt_device_alloc(pt_ctx *ctx, size_t bytes, void **out);
pt_status pt_device_free(pt_ctx *ctx, void *device_ptr);
pt_status pt_device_read(pt_ctx *ctx, const void *device_src, void *host_dst, size_t bytes);
pt_status pt_device_write(pt_ctx *ctx, void *device_dst, const void *host_src, size_t bytes); /* (API version 5) blocking host->device copy */

/* ---- statistics ------------------------------------------------------------------------ */
typedef struct pt_stats {
    uint64_t rays;             /* closest-hit queries (= traceRayEXT calls) since last reset, exact */
    uint64_t paths;            /* samples started                                                   */
    uint32_t rounds;           /* wavefront rounds executed                                         */
    uint32_t launches_extend, launches_shade, launches_other;
    float    ms_total;         /* device time of pt_render calls (first kernel .. last kernel)      */
    float    ms_extend, ms_shade; /* summed kernel times; only with PT_FLAG_PROFILE                 */
    uint32_t extend_variant;   /* PT_EXTEND_* that actually ran                                     */
    uint64_t nodes_visited;    /* BVH4 nodes fetched (128 B each) -- only with PT_FLAG_COUNT_VISITS  */
    uint64_t tris_tested;      /* triangles tested                -- only with PT_FLAG_COUNT_VISITS  */
    uint32_t frames_in_flight; /* shape of the last pt_render / pt_render_prepare                    */
    uint32_t sample_groups;
    /* wave64 steps of the single-level extend kernel (PT_FLAG_COUNT_VISITS): how often a WAVE ran the node
     * code / the triangle code.  nodes_visited / (64 * node_steps) is the lane occupancy of the node phase,
     * tris_tested / (64 * tri_steps) that of the triangle tests.                                        */
    uint64_t node_steps, tri_steps;
    /* batches whose sample-group term log filled up and that were rendered again with one group (exact either way) */
    uint32_t redone_batches;
    uint32_t pipelines;        /* concurrent wavefront pipelines (streams) of the last pt_render */
    /* PT_FLAG_COUNT_VISITS, single-level extend kernel: how often a WAVE executed the other blocks of the kernel --
     * the refill block, one iteration of the stack-pop loop, the hit block of the triangle test (the true divide),
     * the block that writes a hit record, one iteration of the outer loop.  With the per-block instruction counts
     * of the shipped ISA (bench.py) these give the VALU instructions a launch issues without a PMC run.     */
    uint64_t wave_refills, wave_pops, wave_hit_blocks, wave_finishes, wave_iterations;
    /* (API version 3) PT_FLAG_COUNT_VISITS: LANES inside those wave-level steps -- leaf_lanes: lanes that ran a leaf step (one
     * triangle, or a fan pair tested together: tris_tested counts two for those, so tris_tested / tri_steps can exceed 64
     * and is not a lane count); pop_lanes / hit_lanes: lanes in the pop iterations / divide blocks; the two-level kernel's
     * instance-entry block: enter_steps wave executions with enter_lanes lanes.  lanes / (64 * steps) is the occupancy
     * of a block; weighted by the blocks' instruction counts it gives the active lanes per VALU instruction.        */
    uint64_t leaf_lanes, pop_lanes, hit_lanes, enter_steps, enter_lanes;
    /* device bytes the last pt_render / pt_render_prepare holds for its wavefront workspace: the film's queues, hit
     * records, radiance accumulators or term logs, sort scratch and shadow queue, plus the context's traversal-stack
     * spill area (the scene and the film images themselves are not included: pt_scene_info.device_bytes, W*H*16)   */
    uint64_t workspace_bytes;
    uint32_t pipeline;         /* (API version 5) PT_PIPELINE_* the last pt_render / pt_render_prepare ran (never PT_PIPELINE_AUTO) */
    uint32_t tail_samples;     /* fused pipeline: samples per pixel and frame traced as one-sample tail slots behind a head slot (0: none) */
    uint64_t rays_culled;      /* of `rays`: camera rays of pixels outside the projection of the scene's box, which are resolved as the misses they
                                * are without a walk (pt_tuning.cull); the reference traces them (raygen.rgen:62), so they count */
} pt_stats;
pt_status pt_get_stats(pt_ctx *ctx, pt_stats *stats);
pt_status pt_reset_stats(pt_ctx *ctx);
/* (API version 6) Where the fused kernel's instructions go.  A render with pipeline = PT_PIPELINE_FUSED and PT_FLAG_COUNT_VISITS (single-level
 * scenes with pair leaves; never timed) runs the instrumented twin of the kernel: every block of its loop -- the restatement of raygen.rgen:41-91 --
 * counts how often a WAVE executed it and how many LANES were inside.  waves_lanes receives 2 * n_blocks words {wave executions, lanes} in the
 * order of pt_fused_block, summed since pt_reset_stats.  With the blocks' instruction counts in the shipped ISA (scripts/isa_regions.py,
 * profiles/isa_valu_model.json) these give the kernel's VALU wave-instructions and its active lanes per instruction block by block.          */
enum pt_fused_block {
    PT_FB_ITER = 0,  /* one pass of the outer loop                                              */
    PT_FB_SHADE,     /* the shade block ran (lanes: with a finished ray, or asking for a slot)  */
    PT_FB_HIT,       /* ... state loads of the lanes with a hit record                          */
    PT_FB_MISS,      /* ... miss.rmiss:8-12                                                     */
    PT_FB_SURFACE,   /* ... closesthit.rchit:33-41, 50-53: material, emission                   */
    PT_FB_ADD,       /* ... raygen.rgen:76 color += weight * emission                           */
    PT_FB_BOUNCE,    /* ... closesthit.rchit:56-57 + raygen.rgen:77-80                          */
    PT_FB_NEXT,      /* ... path ended: next sample or slot complete (raygen.rgen:43, 62, 81)   */
    PT_FB_DONE,      /* ... the slot's radiance goes to memory                                  */
    PT_FB_HANDOUT,   /* slot hand-out ran (lanes: asking for a slot)                            */
    PT_FB_DRAW,      /* ... the wave drew a batch of slots                                      */
    PT_FB_TAKE,      /* ... lanes that took a slot                                              */
    PT_FB_CULLED,    /* ... of them: finished here, the pixel cannot see the scene              */
    PT_FB_PRIMARY,   /* camera ray: the sample's seed (raygen.rgen:47-48)                       */
    PT_FB_SETUP,     /* state to LDS, ray set-up for the walk                                   */
    PT_FB_NODE,      /* one BVH4 node step                                                      */
    PT_FB_POP,       /* one iteration of the stack-pop loop                                     */
    PT_FB_LEAF,      /* one leaf step (a triangle or a fan pair)                                */
    PT_FB_DIV,       /* ... its divide block                                                    */
    PT_FB_FINISH,    /* a walk ended                                                            */
    PT_FB_TRACE,     /* (no code of its own) once per pass with a walk: the lanes that trace    */
    PT_FB_SPAWN,     /* the steps a bounce and a camera ray share: two rand, one square root    */
    PT_FB_PTARGET,   /* camera ray: pixel + jitter -> target - origin (raygen.rgen:51-56)       */
    PT_FB_PDIR,      /* camera ray: normalize (raygen.rgen:57)                                  */
    PT_FB_POPTOP,    /* a node step without a hit child takes the stack's top entry from a register */
    PT_FB_COUNT
};
pt_status pt_get_block_counts(pt_ctx *ctx, uint64_t *waves_lanes, uint32_t n_blocks /* <= 32 */);

#ifdef __cplusplus
}
#endif
#endif
