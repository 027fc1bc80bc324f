// pt_api.hip -- the C-ABI (include/pt_api.h) over the HIP kernels.  No exception leaves this file.
#include "pt_internal.h"
#include "pt_math.h"

#include <algorithm>
#include <climits>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <new>

static thread_local std::string g_create_err;

// The layers below use std::vector / std::string: a failed host allocation must come back as a status, never as
// an exception through the C boundary.
template <class F>
static pt_status guarded(pt_ctx *ctx, F &&body)
{
    try {
        return body();
    } catch (const std::bad_alloc &) {
        try { if (ctx) ctx->err = "out of host memory"; } catch (...) {}
        return PT_ERR_OOM;
    } catch (const std::exception &e) {
        try { if (ctx) ctx->err = std::string("internal error: ") + e.what(); } catch (...) {}
        return PT_ERR_HIP;
    } catch (...) {
        return PT_ERR_HIP;
    }
}

// ---- tuning (include/pt_api.h pt_tuning): the names PT_TUNE and the Python mirror use, in field order
static const char *const k_tune_names[] = { "refill", "lds_stack", "extend_blocks", "pipes", "stagger", "sort_bits", "pair_leaves",
                                            "pair_kernel", "topdown4", "rec64", "inst16", "inst16_blocks", "enter_min", "node_yield",
                                            "tlas_lds_kb", "term_ocap", "term_spill", "mem_budget_mb", "hbm8", "ploc_radius", "leaf_min",
                                            "tri_enter", "tri_stay", "inst_frames", "tlas_ploc", "ploc_adopt_pct", "fail_rebuild", "fused_tail",
                                            "fused_subject", "cull" };
constexpr int k_tune_count = (int)(sizeof(k_tune_names) / sizeof(k_tune_names[0]));
static_assert(sizeof(pt_tuning) == sizeof(int32_t) * (k_tune_count + 2), "pt_tuning: names and fields out of step");

static void tuning_defaults(pt_tuning *t)
{
    int32_t *f = reinterpret_cast<int32_t *>(t);
    for (size_t i = 0; i < sizeof(pt_tuning) / sizeof(int32_t); i++) f[i] = -1;
}

// "name=value,name=value" (also ';' or blanks between pairs); unknown names and malformed pairs are reported, not ignored
static bool tuning_parse(const char *text, pt_tuning *t, std::string &err)
{
    int32_t *f = reinterpret_cast<int32_t *>(t);
    std::string s(text ? text : "");
    size_t i = 0;
    while (i < s.size()) {
        while (i < s.size() && (s[i] == ',' || s[i] == ';' || s[i] == ' ')) i++;
        if (i >= s.size()) break;
        size_t e = i;
        while (e < s.size() && s[e] != ',' && s[e] != ';' && s[e] != ' ') e++;
        const std::string pair = s.substr(i, e - i);
        i = e;
        const size_t eq = pair.find('=');
        int idx = -1;
        if (eq != std::string::npos)
            for (int k = 0; k < k_tune_count; k++)
                if (pair.compare(0, eq, k_tune_names[k]) == 0) idx = k;
        char *end = nullptr;
        const long v = eq != std::string::npos ? std::strtol(pair.c_str() + eq + 1, &end, 10) : 0;
        if (idx >= 0 && std::string(k_tune_names[idx]) == "fail_rebuild") idx = -1;  // fail_rebuild is failure injection for the tests, not a knob: never from the environment
        if (idx < 0 || !end || *end != 0 || end == pair.c_str() + eq + 1 || v < INT32_MIN || v > INT32_MAX) { err = "PT_TUNE: cannot use '" + pair + "'"; return false; }
        f[idx] = (int32_t)v;
    }
    return true;
}

extern "C" {

pt_status pt_ctx_get_tuning(const pt_ctx *ctx, pt_tuning *out)
{
    if (!ctx || !out) return PT_ERR_INVALID_ARG;
    *out = ctx->tune;
    return PT_OK;
}

pt_status pt_ctx_set_tuning(pt_ctx *ctx, const pt_tuning *in)
{
    if (!ctx || !in) return PT_ERR_INVALID_ARG;
    ctx->tune = *in;
    // mem_budget_mb: > 0 a budget, 0 none, -1 the built-in choice (PT_DEFAULT_MEM_BUDGET_MB) -- so writing back what pt_ctx_get_tuning returned
    // restores the state it described (pt_ctx_create applies the same rule)
    ctx->mem_budget = pt_budget_bytes(in->mem_budget_mb);
    return PT_OK;
}

pt_status pt_ctx_create(int device, void *stream, pt_ctx **out)
{
    if (!out) return PT_ERR_INVALID_ARG;
    *out = nullptr;
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0) {
        g_create_err = std::string("no HIP device: ") + (e != hipSuccess ? hipGetErrorString(e) : "device count is 0");
        (void)hipGetLastError();
        return PT_ERR_NO_DEVICE;
    }
    if (device < 0 || device >= count) {
        g_create_err = "device ordinal out of range";
        return PT_ERR_INVALID_ARG;
    }
    pt_ctx *ctx = new (std::nothrow) pt_ctx();
    if (!ctx) return PT_ERR_OOM;
    ctx->device = device;
    auto fail = [&](const char *what, hipError_t err) {
        g_create_err = std::string(what) + ": " + hipGetErrorString(err);
        pt_ctx_destroy(ctx);
        return PT_ERR_HIP;
    };
    if ((e = hipSetDevice(device)) != hipSuccess) return fail("hipSetDevice", e);
    hipDeviceProp_t prop;
    if ((e = hipGetDeviceProperties(&prop, device)) != hipSuccess) return fail("hipGetDeviceProperties", e);
    ctx->num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    // the only two environment variables the library reads, both here and only here
    tuning_defaults(&ctx->tune);
    const char *const env[2] = { "PT_TUNE", "PT_MEM_BUDGET_MB" };
    const char *val[2];
    for (int k = 0; k < 2; k++) val[k] = getenv(env[k]);
    if (val[0] && !tuning_parse(val[0], &ctx->tune, g_create_err)) { pt_ctx_destroy(ctx); return PT_ERR_INVALID_ARG; }
    if (val[1]) ctx->tune.mem_budget_mb = (int32_t)std::max(0ll, std::min(atoll(val[1]), 1ll << 30));  // (0: no bound)
    ctx->mem_budget = pt_budget_bytes(ctx->tune.mem_budget_mb);
    if (stream) {
        ctx->stream = reinterpret_cast<hipStream_t>(stream);
    } else {
        if ((e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking)) != hipSuccess) return fail("hipStreamCreate", e);
        ctx->own_stream = true;
    }
    if ((e = hipEventCreate(&ctx->ev_a)) != hipSuccess) return fail("hipEventCreate", e);
    if ((e = hipEventCreate(&ctx->ev_b)) != hipSuccess) return fail("hipEventCreate", e);
    if ((e = hipMalloc((void **)&ctx->d_stats, sizeof(unsigned long long) * PT_N_STATS_ALL)) != hipSuccess) return fail("hipMalloc", e);
    if ((e = hipMemset(ctx->d_stats, 0, sizeof(unsigned long long) * PT_N_STATS_ALL)) != hipSuccess) return fail("hipMemset", e);
    if ((e = hipMalloc(&ctx->d_rad, sizeof(ctx->h_rad))) != hipSuccess) return fail("hipMalloc", e);
    if ((e = hipMemset(ctx->d_rad, 0, sizeof(ctx->h_rad))) != hipSuccess) return fail("hipMemset", e);
    *out = ctx;
    return PT_OK;
}

void pt_ctx_destroy(pt_ctx *ctx)
{
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    if (ctx->d_stats) (void)hipFree(ctx->d_stats);
    if (ctx->d_rad) (void)hipFree(ctx->d_rad);
    for (hipEvent_t e : ctx->ev_pool) (void)hipEventDestroy(e);
    if (ctx->d_spill) (void)hipFree(ctx->d_spill);
    if (ctx->ev_a) (void)hipEventDestroy(ctx->ev_a);
    if (ctx->ev_b) (void)hipEventDestroy(ctx->ev_b);
    if (ctx->ev_fork) (void)hipEventDestroy(ctx->ev_fork);
    if (ctx->h_poll) (void)hipHostFree(ctx->h_poll);
    for (int k = 0; k < PT_MAX_PIPES; k++)
        for (int j = 0; j < 2; j++)
            if (ctx->ev_poll[k][j]) (void)hipEventDestroy(ctx->ev_poll[k][j]);
    for (int k = 0; k < PT_MAX_PIPES; k++)
        if (ctx->ev_shade[k]) (void)hipEventDestroy(ctx->ev_shade[k]);
    for (int k = 0; k < PT_MAX_PIPES; k++) {
        if (ctx->ev_join[k]) (void)hipEventDestroy(ctx->ev_join[k]);
        if (ctx->pipe_stream[k]) (void)hipStreamDestroy(ctx->pipe_stream[k]);
    }
    if (ctx->own_stream && ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

const char *pt_last_error(const pt_ctx *ctx) { return ctx ? ctx->err.c_str() : g_create_err.c_str(); }

pt_status pt_sync(pt_ctx *ctx)
{
    if (!ctx) return PT_ERR_INVALID_ARG;
    PT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return PT_OK;
}

pt_status pt_scene_create(pt_ctx *ctx, const float *vertices, uint32_t n_verts, const uint32_t *indices, uint32_t n_tris,
                          const float *faces, pt_scene **out)
{
    if (!ctx) return PT_ERR_INVALID_ARG;
    if (!out || !vertices || !indices || !faces) { ctx->err = "null argument"; return PT_ERR_INVALID_ARG; }
    *out = nullptr;
    if (n_tris == 0 || n_verts == 0) { ctx->err = "empty scene"; return PT_ERR_INVALID_ARG; }
    // BVH4 leaf words keep the first sorted position in bits 0..27 and (count - 1) in bits 28..30
    if (n_tris >= (1u << 28)) { ctx->err = "too many triangles (the BVH4 leaf encoding holds 2^28 - 1)"; return PT_ERR_INVALID_ARG; }
    for (size_t i = 0; i < 3 * (size_t)n_tris; i++)
        if (indices[i] >= n_verts) { ctx->err = "vertex index out of range"; return PT_ERR_INVALID_ARG; }
    PT_HIP(ctx, hipSetDevice(ctx->device));
    pt_scene *s = new (std::nothrow) pt_scene();
    if (!s) return PT_ERR_OOM;
    s->ctx = ctx;
    pt_status rc = guarded(ctx, [&] { return ptb_build_scene(s, vertices, n_verts, indices, n_tris, faces); });
    if (rc != PT_OK) { pt_scene_destroy(s); return rc; }
    *out = s;
    return PT_OK;
}

pt_status pt_scene_set_instances(pt_scene *s, const float *xforms3x4, uint32_t n)
{
    if (!s) return PT_ERR_INVALID_ARG;
    if (n && !xforms3x4) { s->ctx->err = "null argument"; return PT_ERR_INVALID_ARG; }
    if (n >= (1u << 28)) { s->ctx->err = "too many instances"; return PT_ERR_INVALID_ARG; }
    PT_HIP(s->ctx, hipSetDevice(s->ctx->device));
    return guarded(s->ctx, [&] { return ptb_set_instances(s, xforms3x4, n); });
}

void pt_scene_destroy(pt_scene *s)
{
    if (!s) return;
    ptb_free_instances(s);
    ptb_free_scene_buffers(s);
    delete s;
}

pt_status pt_scene_set_bvh_quality(pt_scene *s, uint32_t quality)
{
    if (!s) return PT_ERR_INVALID_ARG;
    PT_HIP(s->ctx, hipSetDevice(s->ctx->device));
    return guarded(s->ctx, [&] { return ptb_set_bvh_quality(s, quality); });
}

pt_status pt_scene_get_info(const pt_scene *s, pt_scene_info *info)
{
    if (!s || !info) return PT_ERR_INVALID_ARG;
    info->n_tris = s->n_tris; info->n_nodes = s->n_nodes; info->bvh_height = s->height;
    info->n_wide_nodes = s->n_wide;
    info->n_instances = s->n_inst;
    info->n_tlas_nodes = s->n_tlas_wide;
    info->leaf_max = PT_BLAS_LEAF_MAX;
    info->n_wide8_nodes = s->n_wide8; info->wide8_levels = s->levels8; info->device_bytes8 = s->device_bytes8;  // (0 until built: big scenes, first request)
    info->bvh4_builder = s->bvh4_builder;
    for (int k = 0; k < 3; k++) { info->bbox_min[k] = s->bmin[k]; info->bbox_max[k] = s->bmax[k]; }
    info->build_ms = s->build_ms;
    info->device_bytes = s->device_bytes;
    // (normalised by the root's area = the scene box's: the same for both trees)
    const double ex = (double)s->bmax[0] - s->bmin[0], ey = (double)s->bmax[1] - s->bmin[1], ez = (double)s->bmax[2] - s->bmin[2];
    const double root = (ex * ey + ey * ez) + ez * ex;
    info->tree_area_lbvh = root > 0.0 ? (float)(s->area_lbvh / root) : 0.f;
    info->tree_area_ploc = root > 0.0 ? (float)(s->area_ploc / root) : 0.f;
    return PT_OK;
}

pt_status pt_scene_read_bvh(const pt_scene *s, uint64_t *keys, uint32_t *prim_of_pos, uint32_t *nodes16)
{
    if (!s) return PT_ERR_INVALID_ARG;
    pt_ctx *ctx = s->ctx;
    if (s->broken) { const pt_status rb = guarded(ctx, [&] { return ptb_repair(const_cast<pt_scene *>(s)); }); if (rb != PT_OK) return rb; }
    PT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (keys) PT_HIP(ctx, hipMemcpy(keys, s->d_keys, sizeof(uint64_t) * (size_t)s->n_tris, hipMemcpyDeviceToHost));
    if (prim_of_pos) PT_HIP(ctx, hipMemcpy(prim_of_pos, s->d_prim_of, sizeof(uint32_t) * (size_t)s->n_tris, hipMemcpyDeviceToHost));
    if (nodes16) PT_HIP(ctx, hipMemcpy(nodes16, s->d_nodes, 64 * (size_t)s->n_nodes, hipMemcpyDeviceToHost));
    return PT_OK;
}

pt_status pt_scene_read_bvh4(const pt_scene *s, uint32_t *nodes32)
{
    if (!s || !nodes32) return PT_ERR_INVALID_ARG;
    pt_ctx *ctx = s->ctx;
    if (s->broken) { const pt_status rb = guarded(ctx, [&] { return ptb_repair(const_cast<pt_scene *>(s)); }); if (rb != PT_OK) return rb; }
    PT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    PT_HIP(ctx, hipMemcpy(nodes32, s->d_wide, 128 * (size_t)s->n_wide, hipMemcpyDeviceToHost));
    return PT_OK;
}

pt_status pt_scene_read_bvh8(const pt_scene *s, uint32_t *nodes32, uint32_t *prim_of_pos8)
{
    if (!s) return PT_ERR_INVALID_ARG;
    pt_ctx *ctx = s->ctx;
    if (s->broken) { const pt_status rb = guarded(ctx, [&] { return ptb_repair(const_cast<pt_scene *>(s)); }); if (rb != PT_OK) return rb; }
    if (!s->d_wide8) {   // big scenes build their 8-wide nodes on first request
        const pt_status rc = guarded(ctx, [&] { return ptb_ensure_wide8(const_cast<pt_scene *>(s)); });
        if (rc != PT_OK) return rc;
    }
    if (!s->d_wide8) { ctx->err = "the scene has no BVH8 (a single triangle, or instanced)"; return PT_ERR_UNSUPPORTED; }
    PT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (nodes32) PT_HIP(ctx, hipMemcpy(nodes32, s->d_wide8, 64 * (size_t)s->n_wide8, hipMemcpyDeviceToHost));
    if (prim_of_pos8) PT_HIP(ctx, hipMemcpy(prim_of_pos8, s->d_prim_of8, sizeof(uint32_t) * (size_t)s->n_tris, hipMemcpyDeviceToHost));
    return PT_OK;
}

static pt_status film_create(pt_ctx *ctx, uint32_t w, uint32_t h, void *ext, pt_film **out)
{
    if (!ctx) return PT_ERR_INVALID_ARG;
    if (!out || w == 0 || h == 0 || w > 524280u || h > 524280u || (uint64_t)w * h >= (1ull << 28)) { ctx->err = "bad film size"; return PT_ERR_INVALID_ARG; }
    *out = nullptr;
    PT_HIP(ctx, hipSetDevice(ctx->device));
    pt_film *f = new (std::nothrow) pt_film();
    if (!f) return PT_ERR_OOM;
    f->ctx = ctx; f->w = w; f->h = h;
    hipError_t e = hipSuccess;
    if (ext) { f->d_rgb = static_cast<float *>(ext); f->own_rgb = false; }
    else e = hipMalloc((void **)&f->d_rgb, sizeof(float) * 3 * (size_t)w * h);
    if (e == hipSuccess) e = hipMalloc((void **)&f->d_bgra, 4 * (size_t)w * h);
    if (e != hipSuccess) { ctx->err = std::string("hipMalloc: ") + hipGetErrorString(e); pt_film_destroy(f); return PT_ERR_OOM; }
    pt_status rc = pt_film_clear(f);
    if (rc != PT_OK) { pt_film_destroy(f); return rc; }
    *out = f;
    return PT_OK;
}

pt_status pt_film_create(pt_ctx *ctx, uint32_t w, uint32_t h, pt_film **out) { return film_create(ctx, w, h, nullptr, out); }

pt_status pt_film_create_external(pt_ctx *ctx, uint32_t w, uint32_t h, void *device_rgb_f32, pt_film **out)
{
    if (ctx && !device_rgb_f32) { ctx->err = "null device buffer"; return PT_ERR_INVALID_ARG; }
    return film_create(ctx, w, h, device_rgb_f32, out);
}

pt_status pt_film_clear(pt_film *f)
{
    if (!f) return PT_ERR_INVALID_ARG;
    pt_ctx *ctx = f->ctx;
    PT_HIP(ctx, hipMemsetAsync(f->d_rgb, 0, sizeof(float) * 3 * (size_t)f->w * f->h, ctx->stream));
    PT_HIP(ctx, hipMemsetAsync(f->d_bgra, 0, 4 * (size_t)f->w * f->h, ctx->stream));
    PT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return PT_OK;
}

pt_status pt_film_read_f32(pt_film *f, float *rgb)
{
    if (!f || !rgb) return PT_ERR_INVALID_ARG;
    pt_ctx *ctx = f->ctx;
    PT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    PT_HIP(ctx, hipMemcpy(rgb, f->d_rgb, sizeof(float) * 3 * (size_t)f->w * f->h, hipMemcpyDeviceToHost));
    return PT_OK;
}

pt_status pt_film_read_bgra8(pt_film *f, uint8_t *bgra)
{
    if (!f || !bgra) return PT_ERR_INVALID_ARG;
    pt_ctx *ctx = f->ctx;
    PT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    PT_HIP(ctx, hipMemcpy(bgra, f->d_bgra, 4 * (size_t)f->w * f->h, hipMemcpyDeviceToHost));
    return PT_OK;
}

void pt_film_destroy(pt_film *f)
{
    if (!f) return;
    ptw_free_work(f);
    if (f->own_rgb) (void)hipFree(f->d_rgb);
    (void)hipFree(f->d_bgra);
    delete f;
}

void pt_params_default(pt_params *p)
{
    if (!p) return;
    std::memset(p, 0, sizeof(*p));
    p->frame = 0; p->frame_count = 1;
    p->width = 1024; p->height = 1024;          // main.cpp:16-17
    p->spp_per_frame = 32; p->max_depth = 8;    // raygen.rgen:43, 62
    p->tmin = 0.001f; p->tmax = 10000.0f;       // raygen.rgen:71, 73
    p->cam_origin[0] = 0.f; p->cam_origin[1] = -1.f; p->cam_origin[2] = 5.f;  // raygen.rgen:55
    p->cam_target[0] = 0.f; p->cam_target[1] = -1.f; p->cam_target[2] = 2.f;  // raygen.rgen:56
    p->env[0] = 0.7f; p->env[1] = 0.6f; p->env[2] = 0.5f;                     // miss.rmiss:10
    p->rank = 0; p->world = 1;
    p->pipeline = PT_PIPELINE_AUTO;
    p->frames_in_flight = 0;
    p->flags = 0;
    p->extend = PT_EXTEND_AUTO;
    p->sample_groups = 0;
}

pt_status pt_render(pt_scene *s, pt_film *f, const pt_params *p)
{
    if (!s || !f || !p) return PT_ERR_INVALID_ARG;
    if (s->ctx != f->ctx) { s->ctx->err = "scene and film belong to different contexts"; return PT_ERR_INVALID_ARG; }
    PT_HIP(s->ctx, hipSetDevice(s->ctx->device));
    return guarded(s->ctx, [&] { return ptw_render(s, f, p); });
}

pt_status pt_render_prepare(pt_scene *s, pt_film *f, const pt_params *p)
{
    if (!s || !f || !p) return PT_ERR_INVALID_ARG;
    if (s->ctx != f->ctx) { s->ctx->err = "scene and film belong to different contexts"; return PT_ERR_INVALID_ARG; }
    PT_HIP(s->ctx, hipSetDevice(s->ctx->device));
    return guarded(s->ctx, [&] { return ptw_prepare(s, f, p); });
}

pt_status pt_trace(pt_scene *s, const float *rays6, uint32_t n, float tmin, float tmax, uint32_t extend, pt_hit *hits)
{
    if (!s) return PT_ERR_INVALID_ARG;
    if (n && (!rays6 || !hits)) { s->ctx->err = "null argument"; return PT_ERR_INVALID_ARG; }
    PT_HIP(s->ctx, hipSetDevice(s->ctx->device));
    return guarded(s->ctx, [&] { return ptw_trace(s, rays6, n, tmin, tmax, extend, hits); });
}

pt_status pt_device_alloc(pt_ctx *ctx, size_t bytes, void **out)
{
    if (!ctx || !out) return PT_ERR_INVALID_ARG;
    *out = nullptr;
    PT_HIP(ctx, hipSetDevice(ctx->device));
    const hipError_t e = hipMalloc(out, bytes ? bytes : 1);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        *out = nullptr;
        ctx->err = std::string("hipMalloc: ") + hipGetErrorString(e);
        return e == hipErrorOutOfMemory ? PT_ERR_OOM : PT_ERR_HIP;
    }
    return PT_OK;
}

pt_status pt_device_free(pt_ctx *ctx, void *p)
{
    if (!ctx) return PT_ERR_INVALID_ARG;
    PT_HIP(ctx, hipSetDevice(ctx->device));
    PT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    PT_HIP(ctx, hipFree(p));
    return PT_OK;
}

pt_status pt_device_read(pt_ctx *ctx, const void *src, void *dst, size_t bytes)
{
    if (!ctx) return PT_ERR_INVALID_ARG;
    if (!src || !dst) { ctx->err = "null argument"; return PT_ERR_INVALID_ARG; }
    PT_HIP(ctx, hipSetDevice(ctx->device));
    PT_HIP(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    PT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return PT_OK;
}

pt_status pt_device_write(pt_ctx *ctx, void *dst, const void *src, size_t bytes)
{
    if (!ctx) return PT_ERR_INVALID_ARG;
    if (!src || !dst) { ctx->err = "null argument"; return PT_ERR_INVALID_ARG; }
    PT_HIP(ctx, hipSetDevice(ctx->device));
    PT_HIP(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
    PT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return PT_OK;
}

pt_status pt_get_stats(pt_ctx *ctx, pt_stats *out)
{
    if (!ctx || !out) return PT_ERR_INVALID_ARG;
    unsigned long long h[PT_N_STATS];
    PT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    PT_HIP(ctx, hipMemcpy(h, ctx->d_stats, sizeof(h), hipMemcpyDeviceToHost));
    ctx->stats.rays = h[0];
    ctx->stats.nodes_visited = h[2];
    ctx->stats.tris_tested = h[3];
    ctx->stats.node_steps = h[4];
    ctx->stats.tri_steps = h[5];
    ctx->stats.wave_refills = h[8]; ctx->stats.wave_pops = h[9]; ctx->stats.wave_hit_blocks = h[10];
    ctx->stats.wave_finishes = h[11]; ctx->stats.wave_iterations = h[12];
    ctx->stats.leaf_lanes = h[13]; ctx->stats.pop_lanes = h[14]; ctx->stats.hit_lanes = h[15];
    ctx->stats.enter_steps = h[16]; ctx->stats.enter_lanes = h[17];
    ctx->stats.rays_culled = h[19];  // (h[18], h[20]: the ray counters before a launch whose term logs may overflow, render.hip)
    *out = ctx->stats;
    return PT_OK;
}

pt_status pt_get_block_counts(pt_ctx *ctx, uint64_t *waves_lanes, uint32_t n_blocks)
{
    if (!ctx) return PT_ERR_INVALID_ARG;
    if (!waves_lanes || n_blocks > (uint32_t)PT_N_BLOCKS) { ctx->err = "pt_get_block_counts: null array or more than 32 blocks"; return PT_ERR_INVALID_ARG; }
    PT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    PT_HIP(ctx, hipMemcpy(waves_lanes, ctx->d_stats + PT_N_STATS, sizeof(unsigned long long) * 2 * n_blocks, hipMemcpyDeviceToHost));
    return PT_OK;
}

pt_status pt_reset_stats(pt_ctx *ctx)
{
    if (!ctx) return PT_ERR_INVALID_ARG;
    PT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    PT_HIP(ctx, hipMemset(ctx->d_stats, 0, sizeof(unsigned long long) * PT_N_STATS_ALL));
    ctx->stats = pt_stats{};
    return PT_OK;
}

}  // extern "C"
