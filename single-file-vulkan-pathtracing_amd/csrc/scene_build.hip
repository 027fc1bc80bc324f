// scene_build.hip -- pt_scene_create / pt_scene_set_bvh_quality / pt_scene_set_instances: what main.cpp:492-538 does with the
// three scene arrays.  De-indexes the triangles (k_gather), has the tree built (lbvh_build.hip; the surface-area BVH4 of small
// scenes: bvh4_sah_device.hip), packs the per-triangle tables in the traversed leaf order (k_pack: vertices, the geometric
// normal of closesthit.rchit:43-48, brdf = Kd / pi, the tangent frame of raygen.rgen:14-21 -- evaluated once with the
// per-hit operations), and for instanced scenes the TLAS, the per-instance matrices and the world-space emitter / frame tables.
#include "bvh_build.h"

#include <hip/hip_fp16.h>

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

namespace {

// 1. gather: de-indexed triangles + their boxes.  tri_orig: 3 float4 per triangle in prim-id order.
__global__ __launch_bounds__(TB) void k_gather(const float *__restrict__ vertices, const uint32_t *__restrict__ indices,
                                               uint32_t n_tris, float4 *__restrict__ tri_orig,
                                               float4 *__restrict__ tlo, float4 *__restrict__ thi)
{
    const uint32_t t = blockIdx.x * TB + threadIdx.x;
    float mn[3] = { INFINITY, INFINITY, INFINITY }, mx[3] = { -INFINITY, -INFINITY, -INFINITY };
    if (t < n_tris) {
        float v[3][3];
        for (int c = 0; c < 3; c++) {
            const uint32_t vi = indices[3 * (size_t)t + c];
            for (int k = 0; k < 3; k++) v[c][k] = vertices[3 * (size_t)vi + k];
        }
        for (int k = 0; k < 3; k++) {
            mn[k] = fminf(fminf(v[0][k], v[1][k]), v[2][k]);
            mx[k] = fmaxf(fmaxf(v[0][k], v[1][k]), v[2][k]);
        }
        tri_orig[3 * (size_t)t + 0] = make_float4(v[0][0], v[0][1], v[0][2], __uint_as_float(t));
        tri_orig[3 * (size_t)t + 1] = make_float4(v[1][0], v[1][1], v[1][2], 0.f);
        tri_orig[3 * (size_t)t + 2] = make_float4(v[2][0], v[2][1], v[2][2], 0.f);
        tlo[t] = make_float4(mn[0], mn[1], mn[2], 0.f);
        thi[t] = make_float4(mx[0], mx[1], mx[2], 0.f);
    }
}


// 6. leaf-ordered records
__global__ __launch_bounds__(TB) void k_pack(const float4 *__restrict__ tri_orig, const float *__restrict__ faces,
                                             const uint32_t *__restrict__ prim_of, uint32_t n,
                                             float4 *__restrict__ tri4, float4 *__restrict__ shade4,
                                             float4 *__restrict__ shade64, float4 *__restrict__ ke4,
                                             float4 *__restrict__ frame4 = nullptr)
{
    const uint32_t pos = blockIdx.x * TB + threadIdx.x;
    if (pos >= n) return;
    const uint32_t prim = prim_of[pos];
    const float4 a = tri_orig[3 * (size_t)prim + 0], b = tri_orig[3 * (size_t)prim + 1],
                 c = tri_orig[3 * (size_t)prim + 2];
    tri4[3 * (size_t)pos + 0] = a;  // .w = bits(prim)
    tri4[3 * (size_t)pos + 1] = b;
    tri4[3 * (size_t)pos + 2] = make_float4(c.x, c.y, c.z, a.w);  // .w = bits(prim) again: the pair-leaf test of k_extend
                                                                  // reads only this vertex of a quad's second triangle
    // closesthit.rchit:43-48 normal (never flipped), :60 brdf = Kd / pi (true divide), :61 emission
    const ptm::f3 nrm = ptm::tri_normal({ a.x, a.y, a.z }, { b.x, b.y, b.z }, { c.x, c.y, c.z });
    const float *f = faces + 6 * (size_t)prim;
    const float br = ptm::fdiv(f[0], 3.1415927410125732f), bg = ptm::fdiv(f[1], 3.1415927410125732f),
                bb = ptm::fdiv(f[2], 3.1415927410125732f);
    shade4[3 * (size_t)pos + 0] = make_float4(nrm.x, nrm.y, nrm.z, br);
    shade4[3 * (size_t)pos + 1] = make_float4(bg, bb, f[3], f[4]);
    shade4[3 * (size_t)pos + 2] = make_float4(f[5], 0.f, 0.f, 0.f);
    if (frame4) {  // raygen.rgen:14-21 for this triangle's normal: {T.xyz, B.x} {B.yz, 0, 0}
        ptm::f3 T, B;
        ptm::tangent_frame(nrm, T, B);
        frame4[2 * (size_t)pos + 0] = make_float4(T.x, T.y, T.z, B.x);
        frame4[2 * (size_t)pos + 1] = make_float4(B.y, B.z, 0.f, 0.f);
    }
    // the same values regrouped for scenes whose tables stay in HBM (k_shade<.., false>): one 64-B record instead of
    // two 48-B ones (4 divergent 16-B loads per hit instead of 6, 1 instead of 3 for a path that ends at this hit),
    // the emission apart because almost no triangle has one
    const bool emits = !(f[3] == 0.f && f[4] == 0.f && f[5] == 0.f);
    shade64[4 * (size_t)pos + 0] = make_float4(a.x, a.y, a.z, nrm.x);
    shade64[4 * (size_t)pos + 1] = make_float4(b.x, b.y, b.z, nrm.y);
    shade64[4 * (size_t)pos + 2] = make_float4(c.x, c.y, c.z, nrm.z);
    shade64[4 * (size_t)pos + 3] = make_float4(br, bg, bb, emits ? 1.f : 0.f);
    ke4[pos] = make_float4(f[3], f[4], f[5], 0.f);
}


__global__ __launch_bounds__(TB) void k_compose(const uint32_t *__restrict__ order8, const uint32_t *__restrict__ prim_of, uint32_t n,
                                                uint32_t *__restrict__ out)
{
    const uint32_t i = blockIdx.x * TB + threadIdx.x;
    if (i < n) out[i] = prim_of[order8[i]];
}


}  // namespace

__global__ __launch_bounds__(TB) void k_wide_half(const float4 *__restrict__ wide, uint32_t n_wide, float cx, float cy,
                                                  float cz, float rsx, float rsy, float rsz, uint4 *__restrict__ out)
{
    const uint32_t i = blockIdx.x * TB + threadIdx.x;
    if (i >= n_wide) return;
    const float4 *nd = wide + 8 * (size_t)i;
    const float c[3] = { cx, cy, cz }, rs[3] = { rsx, rsy, rsz };
    uint32_t d[12];
    for (int ax = 0; ax < 3; ax++) {
        const float4 lo = nd[ax], hi = nd[3 + ax];
        const float l[4] = { lo.x, lo.y, lo.z, lo.w }, h[4] = { hi.x, hi.y, hi.z, hi.w };
        uint32_t hl[4], hh[4];
        for (int k = 0; k < 4; k++) {
            hl[k] = __half_as_ushort(__float2half_rd((l[k] - c[ax]) * rs[ax] - 3.814697265625e-06f));
            hh[k] = __half_as_ushort(__float2half_ru((h[k] - c[ax]) * rs[ax] + 3.814697265625e-06f));
        }
        d[2 * ax + 0] = hl[0] | (hl[1] << 16); d[2 * ax + 1] = hl[2] | (hl[3] << 16);
        d[6 + 2 * ax + 0] = hh[0] | (hh[1] << 16); d[6 + 2 * ax + 1] = hh[2] | (hh[3] << 16);
    }
    const float4 cw = nd[6];
    uint4 *o = out + 4 * (size_t)i;
    o[0] = make_uint4(d[0], d[1], d[2], d[3]);
    o[1] = make_uint4(d[4], d[5], d[6], d[7]);
    o[2] = make_uint4(d[8], d[9], d[10], d[11]);
    o[3] = make_uint4(__float_as_uint(cw.x), __float_as_uint(cw.y), __float_as_uint(cw.z), __float_as_uint(cw.w));
}

// (re)builds s->d_wide16 from the BVH4 that is currently traversed
static pt_status make_wide16(pt_scene *s)
{
    pt_ctx *ctx = s->ctx;
    hipStream_t st = ctx->stream;
    ptb_norm_box(s->bmin, s->bmax, s->norm_c, s->norm_s, s->norm_rs);
    (void)hipFree(s->d_wide16);
    s->d_wide16 = nullptr;
    PT_HIP(ctx, hipMalloc((void **)&s->d_wide16, 64 * (size_t)s->n_wide));
    k_wide_half<<<(s->n_wide + TB - 1) / TB, TB, 0, st>>>(s->d_wide, s->n_wide, s->norm_c[0], s->norm_c[1], s->norm_c[2], s->norm_rs[0],
                                                         s->norm_rs[1], s->norm_rs[2], reinterpret_cast<uint4 *>(s->d_wide16));
    PT_HIP(ctx, hipStreamSynchronize(st));
    PT_HIP(ctx, hipGetLastError());
    return PT_OK;
}

// triangle boxes from the de-indexed triangles (the same float operations as k_gather)
__global__ __launch_bounds__(TB) void k_tri_boxes(const float4 *__restrict__ tri_orig, uint32_t n, float4 *__restrict__ tlo,
                                                  float4 *__restrict__ thi)
{
    const uint32_t t = blockIdx.x * TB + threadIdx.x;
    if (t >= n) return;
    const float4 a = tri_orig[3 * (size_t)t + 0], b = tri_orig[3 * (size_t)t + 1], c = tri_orig[3 * (size_t)t + 2];
    tlo[t] = make_float4(fminf(fminf(a.x, b.x), c.x), fminf(fminf(a.y, b.y), c.y), fminf(fminf(a.z, b.z), c.z), 0.f);
    thi[t] = make_float4(fmaxf(fmaxf(a.x, b.x), c.x), fmaxf(fmaxf(a.y, b.y), c.y), fmaxf(fmaxf(a.z, b.z), c.z), 0.f);
}

// Everything that hangs off the binary tree -- the tree itself (LBVH, or its PLOC rebuild for big scenes under
// ePreferFastTrace), the BVH4 in both node formats, optionally the 8-wide nodes, the per-triangle tables in the traversed
// leaf order -- built (or rebuilt: quality change, first request for the 8-wide nodes) from the kept triangles.
static void free_tree_products(pt_scene *s)
{
    (void)hipFree(s->d_nodes); (void)hipFree(s->d_keys); (void)hipFree(s->d_prim_of); (void)hipFree(s->d_prim_of_sah);
    (void)hipFree(s->d_wide_lbvh ? s->d_wide_lbvh : s->d_wide); (void)hipFree(s->d_wide_sah);
    (void)hipFree(s->d_wide16); (void)hipFree(s->d_wide16t);
    (void)hipFree(s->d_wide8); (void)hipFree(s->d_prim_of8); (void)hipFree(s->d_tri4_8); (void)hipFree(s->d_shade64_8); (void)hipFree(s->d_ke4_8);
    s->d_nodes = nullptr; s->d_keys = nullptr; s->d_prim_of = s->d_prim_of_sah = nullptr;
    s->d_wide = s->d_wide_lbvh = s->d_wide_sah = nullptr; s->d_wide16 = nullptr; s->d_wide16t = nullptr;
    s->d_wide8 = nullptr; s->d_prim_of8 = nullptr; s->d_tri4_8 = s->d_shade64_8 = s->d_ke4_8 = nullptr;
    s->n_wide8 = s->levels8 = 0; s->n_wide16t = s->levels4t = 0;
}

static pt_status build_tree_products_unguarded(pt_scene *s, uint32_t quality, bool want8);

// A rebuild frees the old products first (peak memory = one set, and a scene of 8 M triangles holds 2.6 GB of them), so a
// rebuild that fails part-way -- out of memory beside a 70-100 GB film workspace is the plausible case -- leaves the scene
// WITHOUT a tree.  It is then marked broken: every product pointer null, every count zero, and plan_extend / pt_scene_read_* /
// pt_scene_set_instances answer PT_ERR_UNSUPPORTED instead of launching kernels on null tables.  The triangles and materials
// (d_tri_orig, d_faces) are untouched, so a later pt_scene_set_bvh_quality -- or the next render's request for the 8-wide
// nodes -- can build again; success clears the mark.
static pt_status build_tree_products(pt_scene *s, uint32_t quality, bool want8)
{
    const pt_status rc = build_tree_products_unguarded(s, quality, want8);
    if (rc != PT_OK) {
        (void)hipGetLastError();  // an out-of-memory error is sticky until read
        free_tree_products(s);
        s->n_nodes = s->n_wide = s->n_wide_lbvh = 0;
        s->stack_need = s->stack_need_lbvh = 0xFFFFFFFFu;
        s->device_bytes = s->device_bytes8 = 0;
        s->quality = quality;  // what the scene is meant to have: ptb_repair / the next pt_scene_set_bvh_quality build exactly that
        s->broken = true;
    } else {
        s->broken = false;
    }
    return rc;
}

static pt_status build_tree_products_unguarded(pt_scene *s, uint32_t quality, bool want8)
{
    pt_ctx *ctx = s->ctx;
    hipStream_t st = ctx->stream;
    const uint32_t n = s->n_tris, gt = (n + TB - 1) / TB;
    free_tree_products(s);
    if (ctx->tune.fail_rebuild > 0 && (s->broken || s->n_nodes)) {  // tests: an out-of-memory rebuild, without the memory
        ctx->tune.fail_rebuild--;
        ctx->err = "hipMalloc: out of memory (pt_tuning.fail_rebuild)";
        return PT_ERR_OOM;
    }
    DevBuf<float4> d_tlo, d_thi;
    PT_HIP(ctx, d_tlo.alloc(n));
    PT_HIP(ctx, d_thi.alloc(n));
    PT_HIP(ctx, hipEventRecord(ctx->ev_a, st));
    k_tri_boxes<<<gt, TB, 0, st>>>(s->d_tri_orig, n, d_tlo.p, d_thi.p);
    const bool ploc = quality == PT_BVH_PREFER_FAST_TRACE && n > PT_SAH_MAX_TRIS;
    BvhOut o;
    pt_status rc = ptb_build_bvh(ctx, d_tlo.p, d_thi.p, n, PT_BLAS_LEAF_MAX, o, 2 | (want8 ? 1 : 0), ploc);
    s->d_keys = o.d_keys; s->d_prim_of = o.d_prim_of; s->d_nodes = o.d_nodes; s->d_wide = o.d_wide;  // freed by pt_scene_destroy
    s->d_prim_of_sah = o.d_prim_q;
    s->d_wide8 = o.d_wide8; s->n_wide8 = o.n_wide8; s->levels8 = o.levels8;
    s->d_wide16t = o.d_wide16t; s->n_wide16t = o.n_wide16t; s->levels4t = o.levels4t;
    DevBuf<uint32_t> d_order8;
    d_order8.p = o.d_order8;
    s->d_wide_lbvh = s->d_wide;
    if (rc != PT_OK) return rc;
    s->n_nodes = o.n_nodes; s->n_wide = o.n_wide; s->height = o.height; s->height_tree = o.height_tree; s->stack_need = o.stack_need;
    s->n_wide_lbvh = s->n_wide; s->stack_need_lbvh = s->stack_need;
    for (int k = 0; k < 3; k++) { s->bmin[k] = o.bmin[k]; s->bmax[k] = o.bmax[k]; }
    s->bvh4_builder = o.d_prim_q ? 2u : 0u;
    s->area_lbvh = o.area_lbvh; s->area_ploc = o.area_ploc;
    s->pair_leaves = PT_BLAS_LEAF_MAX == 1u;
    const uint32_t *order = o.d_prim_q ? o.d_prim_q : s->d_prim_of;   // the traversed leaf order
    k_pack<<<gt, TB, 0, st>>>(s->d_tri_orig, s->d_faces, order, n, s->d_tri4, s->d_shade4, s->d_shade64, s->d_ke4, s->d_frame4);
    if (s->d_wide8) {  // the 8-wide tree's own triangle order: its per-triangle tables (the LDS-sized shade4 is never used with it)
        PT_HIP(ctx, hipMalloc((void **)&s->d_prim_of8, sizeof(uint32_t) * (size_t)n));
        PT_HIP(ctx, hipMalloc((void **)&s->d_tri4_8, sizeof(float4) * 3 * (size_t)n));
        PT_HIP(ctx, hipMalloc((void **)&s->d_shade64_8, sizeof(float4) * 4 * (size_t)n));
        PT_HIP(ctx, hipMalloc((void **)&s->d_ke4_8, sizeof(float4) * (size_t)n));
        DevBuf<float4> d_shade4_scratch;
        PT_HIP(ctx, d_shade4_scratch.alloc(3 * (size_t)n));
        k_compose<<<gt, TB, 0, st>>>(d_order8.p, order, n, s->d_prim_of8);
        k_pack<<<gt, TB, 0, st>>>(s->d_tri_orig, s->d_faces, s->d_prim_of8, n, s->d_tri4_8, d_shade4_scratch.p, s->d_shade64_8, s->d_ke4_8);
        PT_HIP(ctx, hipStreamSynchronize(st));
    }
    PT_HIP(ctx, hipEventRecord(ctx->ev_b, st));
    PT_HIP(ctx, hipStreamSynchronize(st));
    PT_HIP(ctx, hipGetLastError());
    PT_HIP(ctx, hipEventElapsedTime(&s->build_ms, ctx->ev_a, ctx->ev_b));
    // resident bytes of the BVH4 path: triangle tables (tri4 48 + shade4 48 + shade64 64 + ke4 16 + frames 32 B each), the kept
    // source arrays a rebuild re-packs from (d_tri_orig 48 + d_faces 24) + the 128-B and the two 64-B node arrays; of the
    // 8-wide path: its tables + nodes
    s->device_bytes = (uint64_t)n * (48 + 48 + 64 + 16 + 32 + PT_SOURCE_BYTES_PER_TRI) + 128ull * s->n_wide + 64ull * s->n_wide + 64ull * s->n_wide16t;
    s->device_bytes8 = s->d_wide8 ? (uint64_t)n * (48 + 64 + 16 + 4) + 64ull * s->n_wide8 : 0ull;
    s->quality = quality;
    return make_wide16(s);
}

// PT_EXTEND_HBM8 / pt_tuning.hbm8 on a scene that was built without the 8-wide nodes: build them now (extend_launch.hip asks)
pt_status ptb_ensure_wide8(pt_scene *s)
{
    if (s->d_wide8 || s->n_tris < 2 || s->n_inst) return PT_OK;
    PT_HIP(s->ctx, hipStreamSynchronize(s->ctx->stream));
    return build_tree_products(s, s->quality, true);
}

// a scene whose last rebuild failed (above) gets one more try per render / trace / read-back: the film whose workspace
// crowded it out may be gone by now
pt_status ptb_repair(pt_scene *s)
{
    if (!s->broken) return PT_OK;
    if (s->n_inst) { s->ctx->err = PT_BROKEN_SCENE_MSG; return PT_ERR_UNSUPPORTED; }  // (cannot happen: rebuilds are refused on instanced scenes)
    PT_HIP(s->ctx, hipStreamSynchronize(s->ctx->stream));
    const pt_status rc = build_tree_products(s, s->quality, s->ctx->tune.hbm8 != 0);
    if (rc != PT_OK) s->ctx->err = std::string(PT_BROKEN_SCENE_MSG) + " [" + s->ctx->err + "]";
    return rc;
}

pt_status ptb_build_scene(pt_scene *s, const float *h_vertices, uint32_t n_verts, const uint32_t *h_indices,
                          uint32_t n_tris, const float *h_faces)
{
    pt_ctx *ctx = s->ctx;
    hipStream_t st = ctx->stream;
    const uint32_t n = n_tris;
    const uint32_t gt = (n + TB - 1) / TB;
    DevBuf<float> d_vert;
    DevBuf<uint32_t> d_idx;
    DevBuf<float4> d_tlo, d_thi;
    PT_HIP(ctx, d_vert.alloc(3 * (size_t)n_verts));
    PT_HIP(ctx, d_idx.alloc(3 * (size_t)n));
    PT_HIP(ctx, d_tlo.alloc(n));
    PT_HIP(ctx, d_thi.alloc(n));
    s->n_tris = n;
    // the de-indexed triangles and the per-face materials stay resident (72 B per triangle): a change of the BVH quality,
    // or the first request for the 8-wide nodes, re-packs the tables from them in another leaf order
    PT_HIP(ctx, hipMalloc((void **)&s->d_tri_orig, sizeof(float4) * 3 * (size_t)n));
    PT_HIP(ctx, hipMalloc((void **)&s->d_faces, sizeof(float) * 6 * (size_t)n));
    PT_HIP(ctx, hipMalloc((void **)&s->d_tri4, sizeof(float4) * 3 * (size_t)n));
    PT_HIP(ctx, hipMalloc((void **)&s->d_shade4, sizeof(float4) * 3 * (size_t)n));
    PT_HIP(ctx, hipMalloc((void **)&s->d_shade64, sizeof(float4) * 4 * (size_t)n));
    PT_HIP(ctx, hipMalloc((void **)&s->d_ke4, sizeof(float4) * (size_t)n));
    PT_HIP(ctx, hipMalloc((void **)&s->d_frame4, sizeof(float4) * 2 * (size_t)n));
    PT_HIP(ctx, hipMemcpyAsync(d_vert.p, h_vertices, sizeof(float) * 3 * (size_t)n_verts, hipMemcpyHostToDevice, st));
    PT_HIP(ctx, hipMemcpyAsync(d_idx.p, h_indices, sizeof(uint32_t) * 3 * (size_t)n, hipMemcpyHostToDevice, st));
    PT_HIP(ctx, hipMemcpyAsync(s->d_faces, h_faces, sizeof(float) * 6 * (size_t)n, hipMemcpyHostToDevice, st));
    PT_HIP(ctx, hipStreamSynchronize(st));  // pageable host sources are done with
    k_gather<<<gt, TB, 0, st>>>(d_vert.p, d_idx.p, n, s->d_tri_orig, d_tlo.p, d_thi.p);
    // the tree of the default quality (ePreferFastTrace, main.cpp:419): PLOC for big scenes; small scenes get the LBVH
    // here and the exact surface-area BVH4 below.  The 8-wide nodes only when the context asks AUTO to use them.
    // (small scenes get the 8-wide nodes at once -- a few KB; big ones on first request: 260 B per triangle nobody else needs)
    // ... and scenes AUTO walks through them: more than 1 MiB of BVH4 nodes + records, ~96 B per triangle (extend_launch.hip ptw_plan_extend)
    pt_status rc = build_tree_products(s, PT_BVH_PREFER_FAST_TRACE, ctx->tune.hbm8 == 1 || n <= PT_SAH_MAX_TRIS || (ctx->tune.hbm8 != 0 && 96ull * n > (1ull << 20)));
    if (rc != PT_OK) return rc;
    {   // emitters for the NEE pipeline: normal as closesthit.rchit:43-48, area = |cross| / 2, cdf = running float sum of the
        // areas in primitive order (this file is compiled with -ffp-contract=off on the host side too)
        std::vector<float4> lights;
        float run = 0.f;
        for (uint32_t t = 0; t < n; t++) {
            const float *f = h_faces + 6 * (size_t)t;
            if (!(f[3] != 0.f || f[4] != 0.f || f[5] != 0.f)) continue;
            const float *a = h_vertices + 3 * (size_t)h_indices[3 * (size_t)t + 0], *b = h_vertices + 3 * (size_t)h_indices[3 * (size_t)t + 1],
                        *c = h_vertices + 3 * (size_t)h_indices[3 * (size_t)t + 2];
            const float e1[3] = { b[0] - a[0], b[1] - a[1], b[2] - a[2] }, e2[3] = { c[0] - a[0], c[1] - a[1], c[2] - a[2] };
            const float cx = e1[1] * e2[2] - e1[2] * e2[1], cy = e1[2] * e2[0] - e1[0] * e2[2], cz = e1[0] * e2[1] - e1[1] * e2[0];
            const float len = sqrtf((cx * cx + cy * cy) + cz * cz);
            run = run + 0.5f * len;
            lights.push_back(make_float4(a[0], a[1], a[2], run));
            lights.push_back(make_float4(b[0], b[1], b[2], 0.f));
            lights.push_back(make_float4(c[0], c[1], c[2], 0.f));
            lights.push_back(make_float4(-(cx / len), -(cy / len), -(cz / len), 0.f));
            lights.push_back(make_float4(f[3], f[4], f[5], 0.f));
        }
        s->n_lights = (uint32_t)(lights.size() / 5);
        s->light_area = run;
        s->h_lights = lights;
        if (s->n_lights) {
            PT_HIP(ctx, hipMalloc((void **)&s->d_lights, sizeof(float4) * lights.size()));
            PT_HIP(ctx, hipMemcpy(s->d_lights, lights.data(), sizeof(float4) * lights.size(), hipMemcpyHostToDevice));
        }
    }
    if (n <= PT_SAH_MAX_TRIS) {
        // small scene: keep what a rebuild of the BVH4 in another leaf order needs, then apply the default
        // quality (ePreferFastTrace, main.cpp:419)
        std::vector<float4> lo(n), hi(n);
        PT_HIP(ctx, hipMemcpy(lo.data(), d_tlo.p, sizeof(float4) * n, hipMemcpyDeviceToHost));
        PT_HIP(ctx, hipMemcpy(hi.data(), d_thi.p, sizeof(float4) * n, hipMemcpyDeviceToHost));
        s->h_tlo.resize(3 * (size_t)n);
        s->h_thi.resize(3 * (size_t)n);
        for (uint32_t i = 0; i < n; i++) {
            s->h_tlo[3 * i + 0] = lo[i].x; s->h_tlo[3 * i + 1] = lo[i].y; s->h_tlo[3 * i + 2] = lo[i].z;
            s->h_thi[3 * i + 0] = hi[i].x; s->h_thi[3 * i + 1] = hi[i].y; s->h_thi[3 * i + 2] = hi[i].z;
        }
        // fan pairs as a loader emits them for quads: the next triangle starts at the same vertex and continues from
        // this one's third (bitwise equal coordinates); greedy, non-overlapping
        s->h_pair.assign(n, 0);
        auto vtx = [&](uint32_t tri, int k) { return h_vertices + 3 * (size_t)h_indices[3 * (size_t)tri + k]; };
        for (uint32_t i = 0; i + 1 < n; i++) {
            const bool same = std::memcmp(vtx(i, 0), vtx(i + 1, 0), 12) == 0 && std::memcmp(vtx(i, 2), vtx(i + 1, 1), 12) == 0;
            if (same) { s->h_pair[i] = 1; i++; }
        }
        const float lbvh_ms = s->build_ms;
        PT_HIP(ctx, hipEventRecord(ctx->ev_a, st));
        const pt_status q = ptb_set_bvh_quality(s, PT_BVH_PREFER_FAST_TRACE);
        if (q != PT_OK) return q;
        PT_HIP(ctx, hipEventRecord(ctx->ev_b, st));
        PT_HIP(ctx, hipStreamSynchronize(st));
        float sah_ms = 0.f;
        PT_HIP(ctx, hipEventElapsedTime(&sah_ms, ctx->ev_a, ctx->ev_b));
        s->build_ms = lbvh_ms + sah_ms;
    }
    return PT_OK;
}

// Chooses the BVH4 that is traversed (pt_internal.h).  Re-packs the per-triangle tables in its leaf order.
pt_status ptb_set_bvh_quality(pt_scene *s, uint32_t quality)
{
    pt_ctx *ctx = s->ctx;
    if (quality > PT_BVH_PREFER_FAST_BUILD) { ctx->err = "unknown BVH quality"; return PT_ERR_INVALID_ARG; }
    if (s->n_inst) { ctx->err = "set the BVH quality before the instances"; return PT_ERR_UNSUPPORTED; }
    if (s->n_tris > PT_SAH_MAX_TRIS) {  // big scene: PLOC tree <-> LBVH, everything that hangs off the tree is rebuilt
        if (quality == s->quality && !s->broken) return PT_OK;
        PT_HIP(ctx, hipStreamSynchronize(ctx->stream));
        return build_tree_products(s, quality, s->d_wide8 != nullptr);
    }
    const bool want_sah = quality == PT_BVH_PREFER_FAST_TRACE && s->n_tris <= PT_SAH_MAX_TRIS && s->d_tri_orig;
    if (want_sah == (s->bvh4_builder == 1u)) return PT_OK;
    hipStream_t st = ctx->stream;
    const uint32_t n = s->n_tris;
    if (want_sah && !s->d_wide_sah) {
        float scale = 0.f;  // leaf_pad() of the device build, same float operations
        for (int k = 0; k < 3; k++) scale = fmaxf(scale, fmaxf(fabsf(s->bmin[k]), fabsf(s->bmax[k])));
        const float pad = scale * 3.814697265625e-06f;
        std::vector<uint32_t> rows, order;
        // one primitive per leaf, a primitive being a triangle or a quad's two halves (pt_tuning.pair_leaves = 0: the former
        // rule, up to PT_SAH_LEAF_MAX independent triangles per leaf where splitting does not pay); built on the device
        const bool pairs = ctx->tune.pair_leaves != 0;
        const pt_status rc8 = pt_sah_build_bvh4_device(ctx, s->h_tlo.data(), s->h_thi.data(), n, pairs ? s->h_pair.data() : nullptr, pad,
                                                       pairs ? 1u : PT_SAH_LEAF_MAX, rows, order);
        if (rc8 != PT_OK) return rc8;
        s->sah_pair_leaves = pairs;
        if (order.size() != n || rows.empty()) { ctx->err = "internal: SAH build lost triangles"; return PT_ERR_HIP; }
        s->n_wide_sah = (uint32_t)(rows.size() / 32);
        s->stack_need_sah = pt_wide_stack_need(rows);
        PT_HIP(ctx, hipMalloc((void **)&s->d_wide_sah, rows.size() * sizeof(uint32_t)));
        PT_HIP(ctx, hipMalloc((void **)&s->d_prim_of_sah, sizeof(uint32_t) * n));
        PT_HIP(ctx, hipMemcpy(s->d_wide_sah, rows.data(), rows.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
        PT_HIP(ctx, hipMemcpy(s->d_prim_of_sah, order.data(), sizeof(uint32_t) * n, hipMemcpyHostToDevice));
    }
    PT_HIP(ctx, hipStreamSynchronize(st));  // nothing may still be traversing the old tables
    if (want_sah) {
        s->d_wide = s->d_wide_sah; s->n_wide = s->n_wide_sah; s->stack_need = s->stack_need_sah; s->bvh4_builder = 1;
        s->pair_leaves = s->sah_pair_leaves;
    } else {
        s->d_wide = s->d_wide_lbvh; s->n_wide = s->n_wide_lbvh; s->stack_need = s->stack_need_lbvh; s->bvh4_builder = 0;
        s->pair_leaves = PT_BLAS_LEAF_MAX == 1u;  // 1-triangle leaves are the degenerate case of the pair kernel
    }
    k_pack<<<(n + TB - 1) / TB, TB, 0, st>>>(s->d_tri_orig, s->d_faces, want_sah ? s->d_prim_of_sah : s->d_prim_of, n, s->d_tri4,
                                           s->d_shade4, s->d_shade64, s->d_ke4, s->d_frame4);
    PT_HIP(ctx, hipStreamSynchronize(st));
    PT_HIP(ctx, hipGetLastError());
    (void)hipFree(s->d_inst_frame);  // (the triangle order changed: ptb_ensure_inst_frames builds it again on the next render)
    s->d_inst_frame = nullptr;
    s->device_bytes = (uint64_t)n * (48 + 48 + 64 + 16 + 32 + PT_SOURCE_BYTES_PER_TRI) + 128ull * s->n_wide + 64ull * s->n_wide + 64ull * s->n_wide16t;
    s->quality = quality;
    return make_wide16(s);
}

void ptb_free_scene_buffers(pt_scene *s)
{
    (void)hipFree(s->d_tri4); (void)hipFree(s->d_shade4); (void)hipFree(s->d_nodes);
    (void)hipFree(s->d_shade64); (void)hipFree(s->d_ke4); (void)hipFree(s->d_frame4);
    s->d_shade64 = s->d_ke4 = s->d_frame4 = nullptr;
    (void)hipFree(s->d_wide_lbvh ? s->d_wide_lbvh : s->d_wide);  // d_wide aliases d_wide_lbvh or d_wide_sah
    (void)hipFree(s->d_wide_sah); (void)hipFree(s->d_prim_of_sah);
    (void)hipFree(s->d_keys); (void)hipFree(s->d_prim_of);
    (void)hipFree(s->d_tri_orig); (void)hipFree(s->d_faces); (void)hipFree(s->d_wide16);
    s->d_wide16 = nullptr;
    (void)hipFree(s->d_wide16t);
    s->d_wide16t = nullptr;
    (void)hipFree(s->d_lights);
    s->d_lights = nullptr; s->n_lights = 0;
    (void)hipFree(s->d_wide8); (void)hipFree(s->d_prim_of8); (void)hipFree(s->d_tri4_8); (void)hipFree(s->d_shade64_8); (void)hipFree(s->d_ke4_8);
    s->d_wide8 = nullptr; s->d_prim_of8 = nullptr; s->d_tri4_8 = s->d_shade64_8 = s->d_ke4_8 = nullptr;
    s->d_tri4 = s->d_shade4 = s->d_nodes = s->d_wide = s->d_wide_lbvh = s->d_wide_sah = s->d_tri_orig = nullptr;
    s->d_prim_of_sah = s->d_prim_of = nullptr; s->d_keys = nullptr; s->d_faces = nullptr;
}

// ---- instances: TLAS over world boxes of the transformed BLAS root box --------------------------
// (beyond the reference, which builds ONE identity instance, main.cpp:515-538)
__global__ __launch_bounds__(TB) void k_inst_boxes(const float4 *__restrict__ blas_wide, const float4 *__restrict__ inst6,
                                                   uint32_t n, float4 *__restrict__ tlo, float4 *__restrict__ thi)
{
    const uint32_t i = blockIdx.x * TB + threadIdx.x;
    if (i >= n) return;
    // object box = union of the (padded) child boxes of the BLAS root
    float omin[3] = { INFINITY, INFINITY, INFINITY }, omax[3] = { -INFINITY, -INFINITY, -INFINITY };
    const float4 lx = blas_wide[0], ly = blas_wide[1], lz = blas_wide[2], hx = blas_wide[3], hy = blas_wide[4], hz = blas_wide[5];
    const float4 cw = blas_wide[6];
    const float l[3][4] = { { lx.x, lx.y, lx.z, lx.w }, { ly.x, ly.y, ly.z, ly.w }, { lz.x, lz.y, lz.z, lz.w } };
    const float h[3][4] = { { hx.x, hx.y, hx.z, hx.w }, { hy.x, hy.y, hy.z, hy.w }, { hz.x, hz.y, hz.z, hz.w } };
    const uint32_t w[4] = { __float_as_uint(cw.x), __float_as_uint(cw.y), __float_as_uint(cw.z), __float_as_uint(cw.w) };
    for (int c = 0; c < 4; c++)
        if (w[c] != PT_MISS)
            for (int k = 0; k < 3; k++) { omin[k] = fminf(omin[k], l[k][c]); omax[k] = fmaxf(omax[k], h[k][c]); }
    const float4 m0 = inst6[6 * (size_t)i + 0], m1 = inst6[6 * (size_t)i + 1], m2 = inst6[6 * (size_t)i + 2];
    float mn[3] = { INFINITY, INFINITY, INFINITY }, mx[3] = { -INFINITY, -INFINITY, -INFINITY };
    for (int c = 0; c < 8; c++) {
        const float px = (c & 1) ? omax[0] : omin[0], py = (c & 2) ? omax[1] : omin[1], pz = (c & 4) ? omax[2] : omin[2];
        const float wx = ((m0.x * px + m0.y * py) + m0.z * pz) + m0.w;
        const float wy = ((m1.x * px + m1.y * py) + m1.z * pz) + m1.w;
        const float wz = ((m2.x * px + m2.y * py) + m2.z * pz) + m2.w;
        mn[0] = fminf(mn[0], wx); mn[1] = fminf(mn[1], wy); mn[2] = fminf(mn[2], wz);
        mx[0] = fmaxf(mx[0], wx); mx[1] = fmaxf(mx[1], wy); mx[2] = fmaxf(mx[2], wz);
    }
    tlo[i] = make_float4(mn[0], mn[1], mn[2], 0.f);
    thi[i] = make_float4(mx[0], mx[1], mx[2], 0.f);
}

__global__ __launch_bounds__(TB) void k_inst_sort(const float4 *__restrict__ inst6, const uint32_t *__restrict__ prim_of,
                                                  uint32_t n, float4 *__restrict__ sorted6)
{
    const uint32_t pos = blockIdx.x * TB + threadIdx.x;
    if (pos >= n) return;
    const uint32_t id = prim_of[pos];
    for (int k = 0; k < 6; k++) sorted6[6 * (size_t)pos + k] = inst6[6 * (size_t)id + k];
}

// world -> object matrix: adjugate / determinant in binary64, rounded once to float
static void invert_3x4(const float m[12], float inv[12])
{
    const double a00 = m[0], a01 = m[1], a02 = m[2], a10 = m[4], a11 = m[5], a12 = m[6], a20 = m[8], a21 = m[9], a22 = m[10];
    const double c00 = a11 * a22 - a12 * a21, c01 = a12 * a20 - a10 * a22, c02 = a10 * a21 - a11 * a20;
    const double det = (a00 * c00 + a01 * c01) + a02 * c02;
    const double i00 = c00 / det, i01 = (a02 * a21 - a01 * a22) / det, i02 = (a01 * a12 - a02 * a11) / det;
    const double i10 = c01 / det, i11 = (a00 * a22 - a02 * a20) / det, i12 = (a02 * a10 - a00 * a12) / det;
    const double i20 = c02 / det, i21 = (a01 * a20 - a00 * a21) / det, i22 = (a00 * a11 - a01 * a10) / det;
    const double tx = m[3], ty = m[7], tz = m[11];
    inv[0] = (float)i00; inv[1] = (float)i01; inv[2] = (float)i02;  inv[3] = (float)(-((i00 * tx + i01 * ty) + i02 * tz));
    inv[4] = (float)i10; inv[5] = (float)i11; inv[6] = (float)i12;  inv[7] = (float)(-((i10 * tx + i11 * ty) + i12 * tz));
    inv[8] = (float)i20; inv[9] = (float)i21; inv[10] = (float)i22; inv[11] = (float)(-((i20 * tx + i21 * ty) + i22 * tz));
}

// World-space normal and tangent of every (instance, triangle): what k_shade's instanced branch used to evaluate per hit -- the
// normal by the inverse transpose, renormalised (a square root and three true divides), and createCoordinateSystem on it (another
// square root and two divides) -- evaluated ONCE with exactly those operations (shade_kernels.hip k_shade; pt_math.h tangent_frame),
// so the bits are the same.  32 B per entry: {n.xyz, T.x} {T.yz, -, -}; the bitangent is the cross product k_shade forms anyway.
// Instance order = d_inst6's (TLAS leaf order), triangle order = d_shade4's (BVH4 leaf order): rebuilt when either changes.
__global__ __launch_bounds__(TB) void k_inst_frames(const float4 *__restrict__ inst6, const float4 *__restrict__ shade4, uint32_t n_inst,
                                                    uint32_t n_tris, float4 *__restrict__ out)
{
    const size_t idx = (size_t)blockIdx.x * TB + threadIdx.x;
    if (idx >= (size_t)n_inst * n_tris) return;
    const uint32_t ip = (uint32_t)(idx / n_tris), pos = (uint32_t)(idx - (size_t)ip * n_tris);
    const float4 s0 = shade4[3 * (size_t)pos];
    const float4 i0 = inst6[6 * (size_t)ip + 3], i1 = inst6[6 * (size_t)ip + 4], i2 = inst6[6 * (size_t)ip + 5];
    const float nx = (i0.x * s0.x + i1.x * s0.y) + i2.x * s0.z;
    const float ny = (i0.y * s0.x + i1.y * s0.y) + i2.y * s0.z;
    const float nz = (i0.z * s0.x + i1.z * s0.y) + i2.z * s0.z;
    const float l = ptm::fsqrt((nx * nx + ny * ny) + nz * nz);
    const ptm::f3 n = { ptm::fdiv(nx, l), ptm::fdiv(ny, l), ptm::fdiv(nz, l) };
    ptm::f3 T, B;
    ptm::tangent_frame(n, T, B);
    out[2 * idx + 0] = make_float4(n.x, n.y, n.z, T.x);
    out[2 * idx + 1] = make_float4(T.y, T.z, 0.f, 0.f);
}

pt_status ptb_ensure_inst_frames(pt_scene *s)
{
    pt_ctx *ctx = s->ctx;
    if (!s->n_inst || s->d_inst_frame) return PT_OK;
    const size_t entries = (size_t)s->n_inst * s->n_tris;
    if (entries * 32 > (512ull << 20)) return PT_OK;  // (a table beyond the caches would cost more than it saves: the per-hit transform stays)
    PT_HIP(ctx, hipMalloc((void **)&s->d_inst_frame, 32 * entries));
    k_inst_frames<<<(unsigned)((entries + TB - 1) / TB), TB, 0, ctx->stream>>>(s->d_inst6, s->d_shade4, s->n_inst, s->n_tris, s->d_inst_frame);
    PT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    PT_HIP(ctx, hipGetLastError());
    return PT_OK;
}

// Emitters of the NEE pipeline for an instanced scene: every instance's copy, in gl_InstanceID order, vertices taken to world
// space by the instance's matrix with the operation order of the shading transform; normal and area from the world-space
// triangle; one running cdf over all of them (the tests' CPU checker restates this loop).  Built on the first NEE render of
// the instance set -- 80 B per (instance, emitter) on host and device, nothing a scene that never samples lights should pay --
// and refused beyond 2^24 copies (1.3 GB; the float running sum of the areas stops resolving small emitters well before).
pt_status ptb_ensure_inst_lights(pt_scene *s)
{
    pt_ctx *ctx = s->ctx;
    if (!s->n_inst || !s->n_lights || s->d_lights_inst) return PT_OK;
    const uint64_t copies = (uint64_t)s->n_inst * s->n_lights;
    if (copies > (1ull << 24) || s->h_xforms.size() != 12 * (size_t)s->n_inst) {
        ctx->err = "the NEE pipeline would need " + std::to_string(copies) + " world-space emitter copies (instances x emitters); the limit is 16 777 216";
        return PT_ERR_UNSUPPORTED;
    }
    const uint32_t n = s->n_inst;
    const float *xforms3x4 = s->h_xforms.data();
    std::vector<float4> wl;
    wl.reserve(5 * (size_t)copies);
    float run = 0.f;
    for (uint32_t i = 0; i < n; i++) {
        const float *m = xforms3x4 + 12 * (size_t)i;
        for (uint32_t k = 0; k < s->n_lights; k++) {
            float w[3][3];
            for (int c = 0; c < 3; c++) {
                const float4 v = s->h_lights[5 * (size_t)k + c];
                for (int r = 0; r < 3; r++) w[c][r] = ((m[4 * r] * v.x + m[4 * r + 1] * v.y) + m[4 * r + 2] * v.z) + m[4 * r + 3];
            }
            const float e1[3] = { w[1][0] - w[0][0], w[1][1] - w[0][1], w[1][2] - w[0][2] }, e2[3] = { w[2][0] - w[0][0], w[2][1] - w[0][1], w[2][2] - w[0][2] };
            const float cx = e1[1] * e2[2] - e1[2] * e2[1], cy = e1[2] * e2[0] - e1[0] * e2[2], cz = e1[0] * e2[1] - e1[1] * e2[0];
            const float len = sqrtf((cx * cx + cy * cy) + cz * cz);
            run = run + 0.5f * len;
            wl.push_back(make_float4(w[0][0], w[0][1], w[0][2], run));
            wl.push_back(make_float4(w[1][0], w[1][1], w[1][2], 0.f));
            wl.push_back(make_float4(w[2][0], w[2][1], w[2][2], 0.f));
            wl.push_back(make_float4(-(cx / len), -(cy / len), -(cz / len), 0.f));
            wl.push_back(s->h_lights[5 * (size_t)k + 4]);
        }
    }
    const hipError_t e = hipMalloc((void **)&s->d_lights_inst, sizeof(float4) * wl.size());
    if (e != hipSuccess) {
        (void)hipGetLastError();
        s->d_lights_inst = nullptr;
        ctx->err = std::string("hipMalloc of the instanced emitter table: ") + hipGetErrorString(e);
        return e == hipErrorOutOfMemory ? PT_ERR_OOM : PT_ERR_HIP;
    }
    PT_HIP(ctx, hipMemcpy(s->d_lights_inst, wl.data(), sizeof(float4) * wl.size(), hipMemcpyHostToDevice));
    s->n_lights_inst = (uint32_t)copies;
    s->light_area_inst = run;
    return PT_OK;
}

void ptb_free_instances(pt_scene *s)
{
    s->h_xforms.clear();
    s->h_xforms.shrink_to_fit();
    (void)hipFree(s->d_inst_frame);
    s->d_inst_frame = nullptr;
    (void)hipFree(s->d_lights_inst);
    s->d_lights_inst = nullptr; s->n_lights_inst = 0; s->light_area_inst = 0.f;
    (void)hipFree(s->d_inst6); (void)hipFree(s->d_tlas_wide); (void)hipFree(s->d_tlas_prim_of); (void)hipFree(s->d_tlas16);
    s->d_inst6 = nullptr; s->d_tlas_wide = nullptr; s->d_tlas_prim_of = nullptr; s->d_tlas16 = nullptr; s->n_tlas16 = 0;
    s->n_inst = 0; s->n_tlas_wide = 0; s->tlas_height = 0;
}

pt_status ptb_set_instances(pt_scene *s, const float *xforms3x4, uint32_t n)
{
    pt_ctx *ctx = s->ctx;
    hipStream_t st = ctx->stream;
    PT_HIP(ctx, hipStreamSynchronize(st));
    ptb_free_instances(s);
    if (n == 0) return PT_OK;
    if (s->broken) { ctx->err = PT_BROKEN_SCENE_MSG; return PT_ERR_UNSUPPORTED; }
    std::vector<float> rec(24 * (size_t)n);
    for (uint32_t i = 0; i < n; i++) {
        const float *m = xforms3x4 + 12 * (size_t)i;
        for (int k = 0; k < 12; k++) {
            if (!(m[k] == m[k]) || __builtin_isinf(m[k])) { ctx->err = "instance matrix has a NaN/Inf"; return PT_ERR_INVALID_ARG; }
            rec[24 * (size_t)i + k] = m[k];
        }
        invert_3x4(m, &rec[24 * (size_t)i + 12]);
        for (int k = 12; k < 24; k++)
            if (!(rec[24 * (size_t)i + k] == rec[24 * (size_t)i + k]) || __builtin_isinf(rec[24 * (size_t)i + k])) {
                ctx->err = "instance matrix is singular";
                return PT_ERR_INVALID_ARG;
            }
    }
    DevBuf<float4> d_in, d_tlo, d_thi;
    PT_HIP(ctx, d_in.alloc(6 * (size_t)n));
    PT_HIP(ctx, d_tlo.alloc(n));
    PT_HIP(ctx, d_thi.alloc(n));
    PT_HIP(ctx, hipMemcpy(d_in.p, rec.data(), sizeof(float) * 24 * (size_t)n, hipMemcpyHostToDevice));
    const uint32_t g = (n + TB - 1) / TB;
    k_inst_boxes<<<g, TB, 0, st>>>(s->d_wide, d_in.p, n, d_tlo.p, d_thi.p);
    BvhOut o;
    // (n < 32768: also the top-down 64-B TLAS with 16-bit child codes that k_extend_inst16 walks)
    // (pt_tuning.tlas_ploc = 1: the TLAS's binary tree by PLOC, as for big single-level scenes; kept under the same area rule)
    pt_status rc = ptb_build_bvh(ctx, d_tlo.p, d_thi.p, n, PT_TLAS_LEAF_MAX, o, (n > 1 && n < 32768u && PT_TLAS_LEAF_MAX == 1u) ? 6 : 0,
                                 ctx->tune.tlas_ploc == 1 && n > 2);
    (void)hipFree(o.d_keys);
    (void)hipFree(o.d_nodes);
    s->d_tlas_wide = o.d_wide;
    if (o.d_prim_q) { (void)hipFree(o.d_prim_of); s->d_tlas_prim_of = o.d_prim_q; }   // the leaf order of the tree that is walked
    else s->d_tlas_prim_of = o.d_prim_of;
    s->tlas_area_lbvh = o.area_lbvh; s->tlas_area_ploc = o.area_ploc;
    s->d_tlas16 = o.d_wide16t; s->n_tlas16 = o.n_wide16t; s->tlas16_levels = o.levels4t;
    for (int k = 0; k < 3; k++) { s->tlas_norm_c[k] = o.norm_c[k]; s->tlas_norm_s[k] = o.norm_s[k]; s->tlas_norm_rs[k] = o.norm_rs[k]; }
    for (int k = 0; k < 3; k++) { s->tlas_bmin[k] = o.bmin[k]; s->tlas_bmax[k] = o.bmax[k]; }
    if (rc != PT_OK) { ptb_free_instances(s); return rc; }
    PT_HIP(ctx, hipMalloc((void **)&s->d_inst6, sizeof(float4) * 6 * (size_t)n));
    // (the emitters' world-space copies for the NEE pipeline are made on that pipeline's first render: ptb_ensure_inst_lights)
    s->h_xforms.assign(xforms3x4, xforms3x4 + 12 * (size_t)n);
    k_inst_sort<<<g, TB, 0, st>>>(d_in.p, s->d_tlas_prim_of, n, s->d_inst6);
    PT_HIP(ctx, hipStreamSynchronize(st));
    PT_HIP(ctx, hipGetLastError());
    s->n_inst = n;
    s->n_tlas_wide = o.n_wide;
    s->tlas_height = std::max(o.height, o.height_tree);  // (of the tree the TLAS was collapsed from: the stack bound)
    return PT_OK;
}
