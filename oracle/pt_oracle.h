/*
 * pt_oracle.h -- CPU ORACLE for the per-pixel radiance loop.  TEST INFRASTRUCTURE ONLY.
 *
 * This is a plain-C restatement of the reference's hot path
 *   shaders/common.glsl, shaders/raygen.rgen, shaders/closesthit.rchit, shaders/miss.rmiss
 * plus a software stand-in for the Vulkan driver's traceRayEXT (raygen.rgen:63-75).
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may call it.
 * The product (single-file-vulkan-pathtracing_amd/) never links, imports or executes it.
 *
 * PARITY STATUS: pinned against outputs of the reference itself, run here.  The reference has
 * no tests or golden vectors and its host program cannot be built
 * (Vulkan SDK + RT driver + GLFW + tinyobjloader, none present), but its compiled shaders --
 * shaders/{raygen.rgen,closesthit.rchit,miss.rmiss}.spv, the whole radiance loop -- are
 * committed, and oracle/spirv_vm.py executes them instruction by instruction.  Pins:
 *   (1) tests/golden/spirv_pixels.npz (generator tests/golden/make_spirv_goldens.py): texels and
 *       traceRayEXT counts those binaries produce for 275 pixels x 3 progressive frames of the
 *       1920x1080 launch, 46 pixels x 4 frames through the rgba8 storage image, every
 *       invocation of a 120x68 launch x 2 frames and a 96x64 crop of BASELINE config 2:
 *       this oracle reproduces all of it bit for bit (tests/test_spirv_pin.py);
 *   (2) the known-answer vectors of SURVEY.md section 8c (tests/golden/kats.json);
 *   (3) self-consistency (brute force == LBVH, libm vs polynomial sincos).
 * NOT pinned, because Vulkan leaves it to the driver and no driver exists here: the
 * ray/triangle intersection behind traceRayEXT, the ulp-level results of sin/cos/sqrt/
 * normalize, and the unorm8 rounding of the storage image.  For these the VM is given this
 * project's canonical definitions (below), so (1) checks the shader-level restatement
 * (seeds, order of rand() calls, camera, loops, shading, throughput, blend), not those.
 *
 * Canonical arithmetic (what "bit-exact" means for this project):
 * every float operation is IEEE-754 binary32, round-to-nearest-even, never contracted
 * into FMA, denormals kept; division and sqrt correctly rounded; sin/cos are the
 * fixed polynomial orc_sincos() below; the ray/triangle test is the watertight test
 * of Woop, Benthin, Wald (JCGT 2013) without back-face culling.
 */
#ifndef PT_ORACLE_H
#define PT_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- RNG (shaders/common.glsl:13-37) ---------------------------------------------- */
uint32_t orc_pcg(uint32_t *state);                          /* common.glsl:13-19 */
void     orc_pcg2d(uint32_t vx, uint32_t vy, uint32_t out[2]); /* common.glsl:21-31 */
float    orc_rand(uint32_t *seed);                          /* common.glsl:33-37 */
uint32_t orc_seed(uint32_t px, uint32_t py, uint32_t sample, int32_t frame,
                  uint32_t spp_per_frame);                  /* raygen.rgen:47-48 */

/* ---- canonical transcendental ------------------------------------------------------ */
/* sin/cos of a in [0, 2*pi]; fixed Cody-Waite + cephes-style minimax polynomial.      */
void orc_sincos(float a, float *s, float *c);

/* ---- render parameters (defaults = the reference's compile-time constants) --------- */
typedef struct orc_params {
    int32_t  frame;            /* push constant, main.cpp:658 / raygen.rgen:8-10        */
    uint32_t width, height;    /* launch size, main.cpp:16-17, 659                      */
    uint32_t spp_per_frame;    /* maxSamples = 32, raygen.rgen:43                       */
    uint32_t max_depth;        /* 8, raygen.rgen:62                                     */
    float    tmin, tmax;       /* 0.001, 10000.0, raygen.rgen:71,73                     */
    float    cam_origin[3];    /* (0,-1,5), raygen.rgen:55                              */
    float    cam_target[3];    /* target = (d.x+tx, d.y+ty, tz), (0,-1,2), raygen.rgen:56 */
    float    env[3];           /* (0.7,0.6,0.5), miss.rmiss:10                          */
    uint32_t libm_sincos;      /* 0 = canonical polynomial, 1 = libm sinf/cosf (tolerance study) */
    uint32_t nee;              /* 1 = the product's opt-in PT_PIPELINE_WAVEFRONT_NEE estimator (NOT the reference's):
                                  next-event estimation, one light sample + one shadow ray per hit; emission of a hit
                                  counts for camera rays only.  Same expectation, other variance and other random numbers. */
} orc_params;
void orc_params_default(orc_params *p);

/* ---- scene ------------------------------------------------------------------------- */
typedef struct orc_scene orc_scene;

/* Same three arrays the reference uploads (main.cpp:492-494):
 * vertices f32[3*n_verts], indices u32[3*n_tris], faces f32[6*n_tris] = {Kd, Ke}.     */
orc_scene *orc_scene_create(const float *vertices, uint32_t n_verts,
                            const uint32_t *indices, uint32_t n_tris, const float *faces);
void orc_scene_destroy(orc_scene *s);
/* Two-level scenes (BASELINE config C4; the reference builds ONE identity instance,
 * main.cpp:515-538): n object->world matrices, 3x4 row major like VkTransformMatrixKHR.
 * n = 0 returns to the single-level scene.  Semantics: world->object
 * matrix = adjugate/determinant in binary64 rounded to float; the ray goes to object space
 * un-normalised (t is shared by both spaces); closest t, ties -> lowest (instance, primitive);
 * hit position by the matrix, normal by the inverse transpose, renormalised.              */
int orc_scene_set_instances(orc_scene *s, const float *xforms3x4, uint32_t n);

/* LBVH facts (Morton 63-bit keys of triangle-AABB centres, stable sort, Karras 2012).  */
typedef struct orc_bvh_info {
    uint32_t n_tris, n_nodes, height;
    float bbox_min[3], bbox_max[3];
} orc_bvh_info;
void orc_scene_bvh_info(const orc_scene *s, orc_bvh_info *info);
/* Sorted Morton keys and leaf order (prim id at each sorted position). n_tris each.    */
void orc_scene_bvh_keys(const orc_scene *s, uint64_t *keys, uint32_t *prim_of_pos);
/* Internal nodes, n_tris-1 of them (1 if n_tris==1); 16 dwords each:
 * lmin[3] lmax[3] rmin[3] rmax[3] left right pad pad; child bit31 set = leaf(position) */
void orc_scene_bvh_nodes(const orc_scene *s, uint32_t *nodes16);

/* ---- closest hit (stands in for traceRayEXT, raygen.rgen:63-75) --------------------- */
typedef struct orc_hit {
    uint32_t prim;   /* 0xFFFFFFFF = miss */
    float    t, u, v; /* u -> weight of v1 (attribs.x), v -> weight of v2 (attribs.y) */
    uint32_t inst;   /* gl_InstanceID: 0 without instances, 0xFFFFFFFF on a miss */
} orc_hit;
typedef struct orc_counters {
    uint64_t rays, nodes_visited /* child boxes tested */, tris_tested;
} orc_counters;
/* mode 0 = brute force over all triangles, 1 = LBVH traversal. Both must agree bit-for-bit. */
void orc_trace(const orc_scene *s, int mode, const float org[3], const float dir[3],
               float tmin, float tmax, orc_hit *hit, orc_counters *cnt);
void orc_trace_batch(const orc_scene *s, int mode, uint32_t n, const float *rays6,
                     float tmin, float tmax, orc_hit *hits, orc_counters *cnt);

/* ---- shading pieces (closesthit.rchit:50-65, raygen.rgen:14-39) -------------------- */
void orc_primary_ray(const orc_params *p, uint32_t px, uint32_t py, uint32_t *seed,
                     float org[3], float dir[3]);          /* raygen.rgen:51-57 */
void orc_shade_hit(const orc_scene *s, const orc_hit *h, float position[3], float normal[3],
                   float brdf[3], float emission[3]);     /* closesthit.rchit:50-65 */
void orc_sample_direction(float r1, float r2, const float n[3], int libm, float out[3]);

/* ---- the frame (raygen.rgen:41-91) -------------------------------------------------- */
/* Renders one launch: frame_color[3*(y*W+x)] = (sum over spp_per_frame samples)/spp,
 * i.e. `color` after raygen.rgen:86, as float (no 8-bit clamp). Returns exact ray count
 * (= number of traceRayEXT calls) and, if cnt != NULL, traversal counters.
 * mode: 0 brute, 1 LBVH. nthreads >= 1 (interleaved rows).
 * If first_hits != NULL (W*H entries) the hit record of sample 0's primary ray is stored. */
uint64_t orc_render_frame(const orc_scene *s, const orc_params *p, int mode, int nthreads,
                          float *frame_color, orc_hit *first_hits, orc_counters *cnt);

/* The same for the pixel rectangle [x0, x0+rw) x [y0, y0+rh) of the width x height launch only
 * (output rw x rh, row-major): full-size configs are checked through crops.                  */
uint64_t orc_render_rect(const orc_scene *s, const orc_params *p, int mode, int nthreads, uint32_t x0,
                         uint32_t y0, uint32_t rw, uint32_t rh, float *frame_color, orc_hit *first_hits,
                         orc_counters *cnt);

/* ... and, if ray_map != NULL (rw*rh entries), the number of traceRayEXT calls every pixel made: what a rank of a
 * tile-sharded render (BASELINE config C3) must count is the sum over its own pixels.                              */
uint64_t orc_render_rect_map(const orc_scene *s, const orc_params *p, int mode, int nthreads, uint32_t x0,
                             uint32_t y0, uint32_t rw, uint32_t rh, float *frame_color, orc_hit *first_hits,
                             orc_counters *cnt, uint32_t *ray_map);

/* Only the 16x16-pixel tiles t (row-major over the whole image) with t % tile_stride == tile_offset; the other pixels of
 * frame_color (W*H*3) are left as they are.  bench.py's cpu_baseline times one thread on every 16th tile -- the same ray
 * population as the all-core run of the whole image (border and centre in the image's own proportion).                */
uint64_t orc_render_tile_subset(const orc_scene *s, const orc_params *p, int mode, int nthreads, uint32_t tile_stride,
                                uint32_t tile_offset, float *frame_color, orc_counters *cnt);

/* raygen.rgen:88-90 in float32 (canonical film): film = (color + film*frame)/(frame+1) */
void orc_accumulate_f32(float *film_rgb, const float *frame_color, int32_t frame, uint64_t n_pixels);
/* raygen.rgen:88-90 as the reference displays it: rgba8 image in B,G,R,A byte order
 * (main.cpp:483), every store clamps to [0,1] and rounds to 1/255.                     */
void orc_accumulate_bgra8(uint8_t *bgra, const float *frame_color, int32_t frame, uint64_t n_pixels);

#ifdef __cplusplus
}
#endif
#endif
