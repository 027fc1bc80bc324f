// render.hip -- pt_render / pt_render_prepare / pt_trace: what runs behind vkCmdTraceRaysKHR (main.cpp:659) on the host side.
//
// A call renders its frames in batches of `frames in flight`; a batch is
//     set-up      its slot lanes (frame x sample group) are split over 1-3 PIPELINES, each with its own stream, queue pair and
//                 counters; k_generate fills queue 0 of each
//     rounds      per pipeline and round: [ray sort] -> closest hit -> k_shade [-> shadow rays -> k_shadow_add]; no stream is
//                 drained inside a batch: queue sizes live on the device, the live counts are polled one poll behind
//     finish      the pipelines join, a batch whose term log overflowed is redone with one sample group, k_resolve blends the
//                 batch's frames into the film in frame order (raygen.rgen:86-90)
// The scheduling rules (stagger of two pipelines, the shade rule of three, the lagged poll) are measured choices; each is
// documented where it is applied.  PT_PIPELINE_FUSED replaces rounds by one persistent kernel per batch (fused.hip).
#include "wavefront_host.h"

#include <cstring>
#include <algorithm>
#include <string>
#include <vector>

namespace {
using namespace ptw;

pt_status check_params(pt_scene *s, pt_film *f, const pt_params *p)
{
    pt_ctx *ctx = s->ctx;
    if (p->width != f->w || p->height != f->h) { ctx->err = "params width/height differ from the film's"; return PT_ERR_INVALID_ARG; }
    if (p->world == 0 || p->rank >= p->world) { ctx->err = "rank/world invalid"; return PT_ERR_INVALID_ARG; }
    if (p->spp_per_frame == 0 || p->spp_per_frame > 0xFFFFu || p->max_depth == 0 || p->max_depth > 0xFFFFu) {
        ctx->err = "spp_per_frame and max_depth must be in 1..65535";
        return PT_ERR_INVALID_ARG;
    }
    if (p->frame < 0 || p->frame_count == 0) { ctx->err = "frame must be >= 0 and frame_count >= 1"; return PT_ERR_INVALID_ARG; }
    if (p->pipeline > PT_PIPELINE_AUTO) { ctx->err = "unknown pipeline"; return PT_ERR_UNSUPPORTED; }
    if (p->pipeline == PT_PIPELINE_FUSED && (p->flags & PT_FLAG_ASYNC)) {
        ctx->err = "the fused pipeline has no asynchronous form";
        return PT_ERR_UNSUPPORTED;
    }
    if (p->pipeline == PT_PIPELINE_WAVEFRONT_NEE) {
        if (p->sample_groups > 1) { ctx->err = "the NEE pipeline runs one sample group per pixel"; return PT_ERR_UNSUPPORTED; }
    }
    return PT_OK;
}


// 0 instanced scenes, 1 scenes walked out of L2 / MALL / HBM, 2 single-level scenes in LDS (film_work.hip ptw_choose_shape)
int launch_class(const pt_scene *s, const ExtendPlan &pl) { return s->n_inst ? 0 : pl.lds_scene ? 2 : 1; }

// The one-time objects of a context's renders: side streams, fork / join / poll / shade events, the pinned poll words.
pt_status ensure_schedule_objects(pt_ctx *ctx, int n_pipes)
{
    for (int k = 1; k < n_pipes; k++)
        if (!ctx->pipe_stream[k]) {
            PT_HIP(ctx, hipStreamCreateWithFlags(&ctx->pipe_stream[k], hipStreamNonBlocking));
            PT_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_join[k], hipEventDisableTiming));
        }
    if (!ctx->ev_fork) PT_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming));
    if (!ctx->h_poll) PT_HIP(ctx, hipHostMalloc((void **)&ctx->h_poll, sizeof(uint32_t) * 2 * PT_MAX_PIPES, hipHostMallocDefault));
    for (int k = 0; k < n_pipes; k++) {
        for (int j = 0; j < 2; j++)
            if (!ctx->ev_poll[k][j]) PT_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_poll[k][j], hipEventDisableTiming));
        if (!ctx->ev_shade[k]) PT_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_shade[k], hipEventDisableTiming));
    }
    return PT_OK;
}

pt_status grow_event_pool(pt_ctx *ctx, size_t want)
{
    while (ctx->ev_pool.size() < want) {
        hipEvent_t e = nullptr;
        PT_HIP(ctx, hipEventCreate(&e));
        ctx->ev_pool.push_back(e);
    }
    return PT_OK;
}

// One wavefront pipeline of a batch: a contiguous range of slot lanes with its own stream, queue pair and counters.
struct Pipe {
    hipStream_t st;
    uint32_t slot_begin, n_slots;
    QueueView qv[2];
    float4 *hit;
    uint32_t *hit_inst;
    uint32_t *count;  // [2] queue sizes of this pipeline
    int cur;
    bool done;
    int polls;        // live-count polls queued on this pipeline's stream in the current batch
};

// What one pt_render call (wavefront pipelines) decides once and every batch uses.
struct Job {
    pt_scene *s; pt_film *f; const pt_params *p; pt_ctx *ctx;
    ExtendPlan pl;
    RenderShape sh;
    RenderConst rc;
    Radiance rad;
    bool nested = false, profile = false, count_visits = false, async = false, nee = false;
    bool shade_lds = false, sort_rays = false, stagger = false;
    int n_pipes = 1, sort_bits = 4, shade_grid = 0;
    size_t shade_smem = 0;
    uint32_t spill_cap = 0;
    const float4 *inst_frame = nullptr, *nee_lights = nullptr;
    uint32_t nee_n_lights = 0;
    float nee_light_area = 0.f;
    unsigned long long *d_overflow = nullptr, *d_spill_count = nullptr;
    std::vector<hipEvent_t> ev_extend, ev_shade;  // (start, stop) pairs filled in by the launches (PT_FLAG_PROFILE)
    size_t ev_used = 0;
};

Radiance job_radiance(const Job &j)
{
    const pt_film::Work &w = j.f->work;
    return { w.d_color, w.d_terms, w.d_terms_over, w.d_nterm, w.d_spill, w.d_spill_head, j.d_spill_count, j.spill_cap, j.d_overflow, nullptr };
}

// The pixel rectangle the scene's box projects to (film_work.hip ptw_tiles_subject_first: its tiles are handed out first).  The camera of
// raygen.rgen:51-57 shoots from cam_origin through (target.x + dx, target.y + dy, target.z), dx, dy in [-1, 1] across the image: a point P in
// front of the origin lands at dx = o.x + (P.x - o.x) (t.z - o.z) / (P.z - o.z) - t.x.  A box is convex, so its image lies inside the bounding
// rectangle of its corners' images.  No rectangle (x1 < x0) when a corner is not in front of the origin (the camera is inside or beside the
// box).  Used twice: as a guess about cost (the hand-out order) and as a proof (the cull below):
// the pixel of slack on every side is far above the rounding of this projection and of raygen's own.
void subject_rect(const pt_scene *s, const pt_params *p, int32_t rect[4])
{
    rect[0] = rect[1] = 0; rect[2] = rect[3] = -1;
    const float *bmin = s->n_inst ? s->tlas_bmin : s->bmin, *bmax = s->n_inst ? s->tlas_bmax : s->bmax;  // (two-level: the union of the instances' world boxes)
    const float den = p->cam_target[2] - p->cam_origin[2];
    if (!(std::fabs(den) > 0.f)) return;
    float lo[2] = { 3.0e38f, 3.0e38f }, hi[2] = { -3.0e38f, -3.0e38f };
    for (int c = 0; c < 8; c++) {
        const float P[3] = { (c & 1) ? bmax[0] : bmin[0], (c & 2) ? bmax[1] : bmin[1], (c & 4) ? bmax[2] : bmin[2] };
        const float a = (P[2] - p->cam_origin[2]) / den;  // how far along the view axis the corner is, in units of the image plane's distance
        if (!(a > 1.0e-4f)) return;
        for (int k = 0; k < 2; k++) {
            const float d = p->cam_origin[k] + (P[k] - p->cam_origin[k]) / a - p->cam_target[k];
            lo[k] = std::min(lo[k], d); hi[k] = std::max(hi[k], d);
        }
    }
    const float size[2] = { (float)p->width, (float)p->height };
    int32_t r[4];
    for (int k = 0; k < 2; k++) {  // dx -> pixel (raygen.rgen:52-53 inverted), one pixel of slack, clamped to the image
        const float a = (lo[k] + 1.0f) * 0.5f * size[k] - 1.0f, b = (hi[k] + 1.0f) * 0.5f * size[k] + 1.0f;
        if (!(a == a) || !(b == b) || b < 0.f || a > size[k]) return;  // (NaN, or the box is off the image: no subject to put first)
        r[k] = (int32_t)std::max(a, 0.f);
        r[k + 2] = (int32_t)std::min(b, size[k] - 1.0f);
    }
    std::copy(r, r + 4, rect);
}

// Pixels outside the rectangle cannot see the scene: their slots are finished where they are handed out / generated (wavefront_types.h
// RenderConst::cull, fused_cull.h; pt_tuning.cull = 0: every camera ray is walked; an instrumented render measures the walk of every ray unless
// pt_tuning.cull = 1 asks for the walked ones only).  The sum a slot without a log stores is the reference's sequence of adds (raygen.rgen:76 with weight 1 and miss.rmiss:10), done here.
void apply_cull(const pt_ctx *ctx, const pt_params *p, const int32_t rect[4], RenderConst &rc)
{
    rc.cull_on = 0u;
    if (rect[2] < rect[0] || rect[3] < rect[1] || ctx->tune.cull == 0 || ((p->flags & PT_FLAG_COUNT_VISITS) && ctx->tune.cull != 1)) return;
    if (!std::isfinite(p->env[0]) || !std::isfinite(p->env[1]) || !std::isfinite(p->env[2])) return;
    rc.cull_on = 1u;
    std::copy(rect, rect + 4, rc.cull);
    for (int k = 0; k < 3; k++) {
        volatile float c = 0.0f;  // (volatile: one rounded float add per sample, nothing folded)
        for (uint32_t i = 0; i < p->spp_per_frame; i++) c = c + 1.0f * p->env[k];
        rc.cull_sum[k] = c;
    }
}

// ---- once per call: kernel plan, shape + workspace, number of pipelines, ray-sort scratch, shadow queue -----------------
pt_status job_setup(Job &j)
{
    pt_scene *s = j.s; pt_film *f = j.f; const pt_params *p = j.p; pt_ctx *ctx = j.ctx;
    pt_film::Work &w = f->work;
    const uint32_t lanes = j.sh.lanes, groups = j.sh.groups;
    if (!j.nested) {
        ctx->stats.frames_in_flight = lanes;
        ctx->stats.sample_groups = groups;
    }
    j.d_overflow = ctx->d_stats + 6; j.d_spill_count = ctx->d_stats + 7;
    j.spill_cap = j.sh.bounded ? SPILL_POOL_ENTRIES : 0u;  // worst-case logs never reach the pool
    if (ctx->tune.term_spill >= 0) j.spill_cap = std::min<uint32_t>(j.spill_cap, (uint32_t)ctx->tune.term_spill);  // tests
    j.rad = job_radiance(j);
    j.rc = ptw_render_const(p, w, j.sh);
    {
        int32_t rect[4];
        subject_rect(s, p, rect);
        apply_cull(ctx, p, rect, j.rc);
    }
    j.profile = !j.nested && (p->flags & PT_FLAG_PROFILE) != 0;  // (a redo would re-record the pooled events of its caller)
    j.count_visits = (p->flags & PT_FLAG_COUNT_VISITS) != 0;
    j.async = (p->flags & PT_FLAG_ASYNC) != 0;
    if (j.async && j.profile) { ctx->err = "PT_FLAG_ASYNC and PT_FLAG_PROFILE exclude each other"; return PT_ERR_INVALID_ARG; }

    // k_shade: 2 paths per thread (one queue-tail atomic per 512 paths), 8 blocks per CU (flat between 4 and 16)
    j.shade_grid = ctx->num_cus * 8;
    j.shade_smem = sizeof(float4) * 8 * (size_t)s->n_tris;  // tri4 + shade4 + the tangent frames
    j.shade_lds = j.shade_smem <= 16 * 1024 && !j.pl.bvh8;  // per-triangle tables of small scenes are staged in LDS (in the BVH4's order)
    // instanced scenes: world-space normal + tangent per (instance, triangle), built once (lbvh_build.hip); pt_tuning.inst_frames = 0
    // keeps the per-hit transform
    if (s->n_inst && !j.pl.bvh8 && ctx->tune.inst_frames != 0) {
        const pt_status rcf = ptb_ensure_inst_frames(s);
        if (rcf != PT_OK) return rcf;
        j.inst_frame = s->d_inst_frame;
    }
    // Pipelines: the slot lanes of a batch are split into parts that run their rounds independently on separate streams, so
    // the VALU-bound traversal of one overlaps the memory-bound shading of another.  Two for most shapes (a third: C4 -1 ... -3 %,
    // C5 -3 ... -6 %; a fourth loses everywhere).  THREE where both kernels keep their tables in LDS, the scene has one level and
    // the batch holds >= 24 M slots: free-running, two pipelines can settle with traversal beside traversal and shade beside
    // shade for a whole process (C2: 22.3-23.0 Grays/s in one pass of a box, 24.0-24.6 in the next); three, held in rotation by
    // the shade rule (run_rounds), cannot: C2 at K = 16 25.8-27.1 against 23.2-24.6, K = 8 +6 %, K = 4 +2.3 %, K = 2 +1.7 %
    // (profiles/r03v_c2_pipes.log, r03w_pipes_by_shape.log, r03aj_shade_rule_other_shapes.log).  Small batches keep one pipeline.
    const bool three = j.shade_lds && !s->n_inst && (uint64_t)w.n_slots >= (24ull << 20);
    int n_pipes = three ? 3 : ((uint64_t)w.n_slots >= (4ull << 20) ? 2 : 1);
    n_pipes = pt_tuned(ctx->tune.pipes, n_pipes, 1, PT_MAX_PIPES);
    j.n_pipes = std::max(1, std::min(n_pipes, std::min<int>(PT_MAX_PIPES, (int)(lanes * groups))));
    ctx->stats.pipelines = (uint32_t)j.n_pipes;
    const pt_status rce = ensure_schedule_objects(ctx, j.n_pipes);
    if (rce != PT_OK) return rce;
    // (two pipelines start half a round apart; three start together and keep the shade rule: run_rounds)
    j.stagger = ctx->tune.stagger < 0 ? j.n_pipes == 2 : ctx->tune.stagger == 1;

    // ray sorting (ray_sort.hip): the HBM kernels only; AUTO when the traversal working set does not fit the Infinity Cache
    if ((j.pl.variant == PT_EXTEND_HBM || j.pl.variant == PT_EXTEND_HBM8) && !s->n_inst) {
        const uint64_t working_set = j.pl.bvh8 ? 64ull * s->n_wide8 + 64ull * s->n_tris
                                               : 64ull * (j.pl.topdown4 ? s->n_wide16t : s->n_wide) + 48ull * s->n_tris;
        j.sort_rays = working_set > (256ull << 20);
        if (p->flags & PT_FLAG_SORT_RAYS) j.sort_rays = true;
        if (p->flags & PT_FLAG_NO_SORT_RAYS) j.sort_rays = false;
    }
    // 4 bits per axis + octant = 15-bit keys = two 8-bit passes (C5x: 6 bits, three passes: +0 %, 4 bits: +3.5 %)
    j.sort_bits = pt_tuned(ctx->tune.sort_bits, 4, 1, 9);
    if (j.sort_rays) {
        // one scratch area per pipeline, sized for that pipeline's share of the slots (+ slack for the uneven split)
        const size_t per_pipe = ptw_ray_sort_bytes((size_t)w.n_slots / (size_t)j.n_pipes + (size_t)j.rc.slots_per_lane + 1);
        const size_t need = per_pipe * (size_t)j.n_pipes;
        if (need > w.sort_bytes) {
            (void)hipFree(w.d_sort);
            w.d_sort = nullptr;
            w.bytes -= w.sort_bytes;
            w.sort_bytes = 0;
            const hipError_t e = hipMalloc(&w.d_sort, need);
            if (e != hipSuccess) { (void)hipGetLastError(); j.sort_rays = false; }  // no room: render unsorted
            else { w.sort_bytes = need; w.bytes += need; }
        }
    }
    j.nee = p->pipeline == PT_PIPELINE_WAVEFRONT_NEE;
    // the emitters the NEE pipeline samples: the scene's, or -- instanced -- every instance's copy of them in world space
    if (j.nee && s->n_inst) {
        const pt_status rcl = ptb_ensure_inst_lights(s);
        if (rcl != PT_OK) return rcl;
    }
    j.nee_lights = s->n_inst ? s->d_lights_inst : s->d_lights;
    j.nee_n_lights = s->n_inst ? s->n_lights_inst : s->n_lights;
    j.nee_light_area = s->n_inst ? s->light_area_inst : s->light_area;
    if (j.nee && (size_t)w.n_slots > w.cap_sq) {  // the shadow queue: at most one entry per live path and round
        (void)hipFree(w.d_sq_rayA); (void)hipFree(w.d_sq_rayB); (void)hipFree(w.d_sq_contrib); (void)hipFree(w.d_sq_slot);
        (void)hipFree(w.d_sq_tmax); (void)hipFree(w.d_sq_hit);
        w.d_sq_rayA = w.d_sq_contrib = w.d_sq_hit = nullptr; w.d_sq_rayB = nullptr; w.d_sq_slot = nullptr; w.d_sq_tmax = nullptr;
        w.cap_sq = 0;
        const size_t ns = w.n_slots;
        PT_HIP(ctx, hipMalloc((void **)&w.d_sq_rayA, sizeof(float4) * ns));
        PT_HIP(ctx, hipMalloc((void **)&w.d_sq_rayB, sizeof(float2) * ns));
        PT_HIP(ctx, hipMalloc((void **)&w.d_sq_contrib, sizeof(float4) * ns));
        PT_HIP(ctx, hipMalloc((void **)&w.d_sq_slot, sizeof(uint32_t) * ns));
        PT_HIP(ctx, hipMalloc((void **)&w.d_sq_tmax, sizeof(float) * ns));
        PT_HIP(ctx, hipMalloc((void **)&w.d_sq_hit, sizeof(float4) * ns));
        if (!w.d_sq_count) PT_HIP(ctx, hipMalloc((void **)&w.d_sq_count, sizeof(uint32_t) * PT_MAX_PIPES));
        w.cap_sq = ns;
    }
    ctx->stats.extend_variant = j.pl.variant;
    return PT_OK;
}

// ---- per batch: the slot lanes split over the pipelines, queue 0 of each filled ---------------------------------------------
pt_status batch_begin(Job &j, Pipe *pipe, int &pipes_now, unsigned long long &rays_before)
{
    pt_ctx *ctx = j.ctx;
    pt_film::Work &w = j.f->work;
    hipStream_t st = ctx->stream;
    rays_before = 0;
    if (j.sh.bounded) {  // the exact ray counter as it is before this batch, should the batch have to be redone
        PT_HIP(ctx, hipStreamSynchronize(st));
        PT_HIP(ctx, hipMemcpy(&rays_before, ctx->d_stats, sizeof(rays_before), hipMemcpyDeviceToHost));
        // (rays_culled before the batch: ON the stream, ahead of the k_generate launches that add to word 19 -- a plain device-to-device copy runs
        // on the null stream, which the library's non-blocking streams are not ordered against)
        PT_HIP(ctx, hipMemcpyAsync(ctx->d_stats + 20, ctx->d_stats + 19, sizeof(unsigned long long), hipMemcpyDeviceToDevice, st));
    }
    PT_HIP(ctx, hipMemsetAsync(w.d_count, 0, sizeof(uint32_t) * 2 * PT_MAX_PIPES, st));
    if (j.sh.bounded) PT_HIP(ctx, hipMemsetAsync(j.d_spill_count, 0, sizeof(unsigned long long), st));
    const uint32_t slot_lanes = j.rc.lanes_active * j.sh.groups;
    pipes_now = std::min<int>(j.n_pipes, (int)slot_lanes);
    for (int k = 0; k < pipes_now; k++) {
        const uint32_t l0 = (uint32_t)((uint64_t)slot_lanes * k / pipes_now);
        const uint32_t l1 = (uint32_t)((uint64_t)slot_lanes * (k + 1) / pipes_now);
        Pipe &pp = pipe[k];
        pp.st = k == 0 ? st : ctx->pipe_stream[k];
        pp.slot_begin = l0 * j.rc.slots_per_lane;
        pp.n_slots = (l1 - l0) * j.rc.slots_per_lane;
        for (int i = 0; i < 2; i++)
            pp.qv[i] = { w.d_qid[i] + pp.slot_begin, w.d_qstate[i] + pp.slot_begin, w.d_qrayA[i] + pp.slot_begin, w.d_qrayB[i] + pp.slot_begin };
        pp.hit = w.d_hit + pp.slot_begin;
        pp.hit_inst = w.d_hit_inst + pp.slot_begin;
        pp.count = w.d_count + 2 * k;
        pp.cur = 0;
        pp.done = false;
        pp.polls = 0;
    }
    if (pipes_now > 1) {  // the other streams start after the counters are cleared
        PT_HIP(ctx, hipEventRecord(ctx->ev_fork, st));
        for (int k = 1; k < pipes_now; k++) PT_HIP(ctx, hipStreamWaitEvent(ctx->pipe_stream[k], ctx->ev_fork, 0));
    }
    for (int k = 0; k < pipes_now; k++) {
        Pipe &pp = pipe[k];
        ptw_launch_generate(j.rc, w.d_tiles, pp.slot_begin, pp.n_slots, j.rad, pp.qv[0], &pp.count[0], ctx->d_stats, ctx->num_cus, pp.st);
        ctx->stats.launches_other++;
    }
    return PT_OK;
}

// ---- one round of one pipeline: [sort] -> closest hit -> shade [-> shadow rays -> add] ---------------------------------
pt_status pipe_round(Job &j, Pipe *pipe, int k, int pipes_now, uint32_t round, bool shade_rule, bool *shade_recorded)
{
    pt_ctx *ctx = j.ctx; pt_scene *s = j.s; const pt_params *p = j.p;
    pt_film::Work &w = j.f->work;
    Pipe &pp = pipe[k];
    const int cur = pp.cur;
    hipEvent_t x0 = nullptr, x1 = nullptr, h0 = nullptr, h1 = nullptr;
    if (j.profile) {  // from the context's pool: creating ~2000 events per call showed in the wall time
        const pt_status rce = grow_event_pool(ctx, j.ev_used + 4);
        if (rce != PT_OK) return rce;
        x0 = ctx->ev_pool[j.ev_used++]; x1 = ctx->ev_pool[j.ev_used++]; h0 = ctx->ev_pool[j.ev_used++]; h1 = ctx->ev_pool[j.ev_used++];
    }
    const uint32_t *perm = nullptr;
    if (j.sort_rays && round > 0) {  // (round 0 is the primary rays: one origin, generated tile by tile)
        const size_t per_pipe = w.sort_bytes / (size_t)j.n_pipes;
        perm = ptw_sort_rays(pp.st, pp.qv[cur].rayA, pp.qv[cur].rayB, &pp.count[cur], pp.n_slots, s->bmin, s->bmax, j.sort_bits, ctx->num_cus,
                             static_cast<char *>(w.d_sort) + per_pipe * (size_t)k);
        ctx->stats.launches_other += 1 + 5 * (uint32_t)((3 * j.sort_bits + 3 + 7) / 8);
    }
    // Two pipelines start half a round apart: pipeline k > 0 begins its first traversal launch when pipeline k-1's first one
    // has finished.  Started together they can lock in phase -- traversal beside traversal, shade beside shade -- and whether
    // they do depended on the box: same-box A/B 22.45 -> 23.99 Grays/s where all plain runs were slow, no change where they
    // were fast; C4 unchanged (profiles/r02i_ab_stagger.log).  Only where the two kernels are of similar length (tables in
    // LDS) and more than one frame is in flight.  A strict token (one traversal launch at a time) is 20 % slower.
    const bool stag = j.stagger && j.shade_lds && j.rc.lanes_active > 1 && round == 0;
    if (stag && k > 0) PT_HIP(ctx, hipStreamWaitEvent(pp.st, ctx->ev_fork, 0));
    ptw_launch_extend(j.pl, s, pp.qv[cur].rayA, pp.qv[cur].rayB, pp.hit, pp.hit_inst, &pp.count[cur], &pp.count[cur ^ 1], ctx->d_stats, p->tmin,
                      p->tmax, j.count_visits, true, pp.st, k, x0, x1, perm);
    if (stag && k + 1 < pipes_now) PT_HIP(ctx, hipEventRecord(ctx->ev_fork, pp.st));
    // The shade rule (three pipelines): never all three in their shade launch at once.  Pipeline k's shade launch waits for the
    // end of the most recent shade launch of pipeline k + 1 -- the one a third of a rotation ahead, long over when the three are
    // evenly spread, so in the pattern the rule aims at nobody waits, and out of it the laggard is held back until the rotation
    // is restored.  Free-running, the three spend 11-19 % of a frame shade beside shade beside shade (latency-bound, VALUs
    // idle) and which pattern a process falls into is chance: 24.3-25.7 Grays/s over eight processes of one box; with the rule
    // 25.8-27.1, mean +6.0 % (profiles/r03ag_c2_rules_distribution.log).  The same rule on the traversal launches, on both, one
    // position further ahead, or with two / four pipelines: all slower (r03af_*, r03ah_*).
    const int ahead = (k + 1) % pipes_now;
    if (shade_rule && shade_recorded[ahead]) PT_HIP(ctx, hipStreamWaitEvent(pp.st, ctx->ev_shade[ahead], 0));
    ShadeLaunch sl;
    sl.rc = j.rc; sl.tiles = w.d_tiles; sl.scene = s; sl.bvh8 = j.pl.bvh8; sl.lds_tables = j.shade_lds; sl.nee = j.nee;
    sl.grid = j.shade_grid; sl.smem = j.shade_smem; sl.rad = j.rad; sl.hit = pp.hit; sl.hit_inst = pp.hit_inst;
    sl.in = pp.qv[cur]; sl.out = pp.qv[cur ^ 1]; sl.count_in = &pp.count[cur]; sl.count_out = &pp.count[cur ^ 1];
    sl.inst_frame = j.inst_frame; sl.lights = j.nee_lights; sl.n_lights = j.nee_n_lights; sl.light_area = j.nee_light_area;
    if (j.nee) {
        sl.sq = { w.d_sq_rayA + pp.slot_begin, w.d_sq_rayB + pp.slot_begin, w.d_sq_contrib + pp.slot_begin, w.d_sq_tmax + pp.slot_begin,
                  w.d_sq_slot + pp.slot_begin };
        sl.sq_count = w.d_sq_count + k;
        PT_HIP(ctx, hipMemsetAsync(sl.sq_count, 0, sizeof(uint32_t), pp.st));
    }
    ptw_launch_shade(sl, pp.st, h0, h1);
    if (j.nee && j.nee_n_lights) {
        // the shadow rays of this round: any-hit queries with their own tmax, then the unoccluded terms
        ptw_launch_extend(j.pl, s, sl.sq.rayA, sl.sq.rayB, w.d_sq_hit + pp.slot_begin, nullptr, sl.sq_count, nullptr, ctx->d_stats, p->tmin, p->tmax,
                          false, true, pp.st, k, nullptr, nullptr, nullptr, sl.sq.tmax);
        ptw_launch_shadow_add(j.rc, j.rad, w.d_sq_hit + pp.slot_begin, sl.sq.contrib, sl.sq.slot, sl.sq_count, j.shade_grid, pp.st);
        ctx->stats.launches_extend++;
        ctx->stats.launches_other++;
    }
    PT_HIP(ctx, hipGetLastError());  // a launch the runtime refused (LDS size, grid) must not pass for a rendered round
    if (shade_rule) { PT_HIP(ctx, hipEventRecord(ctx->ev_shade[k], pp.st)); shade_recorded[k] = true; }
    if (j.profile) {
        j.ev_extend.push_back(x0); j.ev_extend.push_back(x1);
        j.ev_shade.push_back(h0); j.ev_shade.push_back(h1);
    }
    ctx->stats.launches_extend++;
    ctx->stats.launches_shade++;
    pp.cur ^= 1;
    return PT_OK;
}

// The live counts, one poll behind.  Every slot needs >= group_size rounds; after that the counts are polled every eighth
// round so that a batch whose paths have all ended stops early.  The host reads the count of the PREVIOUS poll -- eight rounds
// back -- while each stream still holds eight rounds of launches: waiting for the newest count drained the streams, restarted
// the pipelines in phase and left one of them idle until the other had caught up: 5.3 ms of the 147 ms of 16 C2 frames
// (profiles/r03r_c2_timeline_before.txt).  A pipeline found empty has at most sixteen empty rounds queued behind it.
pt_status poll_live_counts(Job &j, Pipe *pipe, int pipes_now, bool &all_done)
{
    pt_ctx *ctx = j.ctx;
    all_done = true;
    for (int k = 0; k < pipes_now; k++) {
        Pipe &pp = pipe[k];
        if (pp.done) continue;
        const int jj = pp.polls & 1;
        PT_HIP(ctx, hipMemcpyAsync(ctx->h_poll + 2 * k + jj, &pp.count[pp.cur], sizeof(uint32_t), hipMemcpyDeviceToHost, pp.st));
        PT_HIP(ctx, hipEventRecord(ctx->ev_poll[k][jj], pp.st));
        if (pp.polls > 0) {
            PT_HIP(ctx, hipEventSynchronize(ctx->ev_poll[k][jj ^ 1]));
            if (ctx->h_poll[2 * k + (jj ^ 1)] == 0u) pp.done = true;
        }
        pp.polls++;
        if (!pp.done) all_done = false;
    }
    return PT_OK;
}

pt_status run_rounds(Job &j, Pipe *pipe, int pipes_now)
{
    pt_ctx *ctx = j.ctx;
    const uint32_t group_size = j.sh.group_size;
    const uint32_t max_rounds = group_size * j.p->max_depth;  // every sample of a slot at full depth
    bool shade_recorded[PT_MAX_PIPES] = {};
    const bool shade_rule = pipes_now == 3 && ((ctx->tune.stagger < 0 && j.shade_lds) || ctx->tune.stagger == 2);
    for (uint32_t round = 0; round < max_rounds; round++) {
        for (int k = 0; k < pipes_now; k++) {
            if (pipe[k].done) continue;
            const pt_status rc = pipe_round(j, pipe, k, pipes_now, round, shade_rule, shade_recorded);
            if (rc != PT_OK) return rc;
        }
        ctx->stats.rounds++;
        if (!j.async && round + 1 >= group_size && ((round + 1) & 7u) == 0u && round + 1 < max_rounds) {
            bool all_done = false;
            const pt_status rc = poll_live_counts(j, pipe, pipes_now, all_done);
            if (rc != PT_OK) return rc;
            if (all_done) break;
        }
    }
    return PT_OK;
}

// A failure between batch_begin and the join leaves launches queued on the side streams that read and write the film's
// workspace: nothing may free or reuse it before they have drained.
void drain_side_streams(pt_ctx *ctx, int pipes_now)
{
    for (int k = 1; k < pipes_now; k++)
        if (ctx->pipe_stream[k]) (void)hipStreamSynchronize(ctx->pipe_stream[k]);
    (void)hipStreamSynchronize(ctx->stream);
    (void)hipGetLastError();
}

pt_status render_wavefront(pt_scene *s, pt_film *f, const pt_params *p, const ExtendPlan &pl, bool nested);

// ---- per batch: join, term-log overflow check, resolve (or the same frames once more with one sample group) -----------------
pt_status batch_finish(Job &j, int pipes_now, unsigned long long rays_before)
{
    pt_ctx *ctx = j.ctx; pt_film *f = j.f; const pt_params *p = j.p;
    hipStream_t st = ctx->stream;
    for (int k = 1; k < pipes_now; k++) {  // join before the resolve reads every pipeline's slots
        PT_HIP(ctx, hipEventRecord(ctx->ev_join[k], ctx->pipe_stream[k]));
        PT_HIP(ctx, hipStreamWaitEvent(st, ctx->ev_join[k], 0));
    }
    bool redo = false;
    if (j.sh.bounded) {  // did a slot fill its term log?
        unsigned long long flag = 0;
        PT_HIP(ctx, hipStreamSynchronize(st));
        PT_HIP(ctx, hipMemcpy(&flag, j.d_overflow, sizeof(flag), hipMemcpyDeviceToHost));
        redo = flag != 0ull;
    }
    if (!redo) {
        ptw_launch_resolve(j.rc, f->work.d_tiles, j.rad, f->d_rgb, f->d_bgra, st);
        ctx->stats.launches_other++;
        return PT_OK;
    }
    // Rare (scenes where most surfaces emit): nothing of this batch has touched the film yet.  Put the ray counter back, clear
    // the flag and render the same frames with one slot per (frame, pixel) -- the plain accumulator needs no log -- then return
    // to this call's workspace shape.
    PT_HIP(ctx, hipMemcpy(ctx->d_stats, &rays_before, sizeof(rays_before), hipMemcpyHostToDevice));
    PT_HIP(ctx, hipMemcpy(ctx->d_stats + 19, ctx->d_stats + 20, sizeof(unsigned long long), hipMemcpyDeviceToDevice));
    PT_HIP(ctx, hipMemset(j.d_overflow, 0, sizeof(unsigned long long)));
    ctx->stats.redone_batches++;
    pt_params q = *p;
    q.frame = j.rc.frame_base;
    q.frame_count = j.rc.lanes_active;
    q.frames_in_flight = j.rc.lanes_active;
    q.sample_groups = 1;
    pt_status rc = render_wavefront(j.s, f, &q, j.pl, true);
    if (rc != PT_OK) return rc;
    rc = ptw_ensure_work(f, p->rank, p->world, j.sh.lanes, j.sh.groups, j.sh.term_cap, j.sh.term_pcap);
    if (rc != PT_OK) return rc;
    j.rad = job_radiance(j);
    return PT_OK;
}

// ---- the wavefront pipelines (PT_PIPELINE_WAVEFRONT, _NEE) ---------------------------------------------------------------
pt_status render_wavefront(pt_scene *s, pt_film *f, const pt_params *p, const ExtendPlan &pl, bool nested)
{
    pt_ctx *ctx = s->ctx;
    hipStream_t st = ctx->stream;
    Job j{};
    j.s = s; j.f = f; j.p = p; j.ctx = ctx; j.pl = pl; j.nested = nested;
    pt_status rc = ptw_shape_and_work(f, p, j.sh, launch_class(s, pl));
    if (rc != PT_OK) return rc;
    rc = job_setup(j);
    if (rc != PT_OK) return rc;
    if (!nested) PT_HIP(ctx, hipEventRecord(ctx->ev_a, st));
    for (uint32_t done = 0; f->work.n_slots > 0 && done < p->frame_count; done += j.sh.lanes) {
        j.rc.frame_base = p->frame + (int32_t)done;
        j.rc.lanes_active = std::min(j.sh.lanes, p->frame_count - done);
        Pipe pipe[PT_MAX_PIPES];
        int pipes_now = 0;
        unsigned long long rays_before = 0;
        rc = batch_begin(j, pipe, pipes_now, rays_before);
        if (rc == PT_OK) rc = run_rounds(j, pipe, pipes_now);
        if (rc == PT_OK) rc = batch_finish(j, pipes_now, rays_before);
        if (rc != PT_OK) {
            drain_side_streams(ctx, std::max(pipes_now, j.n_pipes));
            return rc;
        }
    }
    if (!nested) PT_HIP(ctx, hipEventRecord(ctx->ev_b, st));
    if (!j.async) {
        PT_HIP(ctx, hipStreamSynchronize(st));
        PT_HIP(ctx, hipGetLastError());
        if (!nested) {
            float ms = 0.f;
            PT_HIP(ctx, hipEventElapsedTime(&ms, ctx->ev_a, ctx->ev_b));
            ctx->stats.ms_total += ms;
        }
    }
    if (!nested) {
        ctx->stats.workspace_bytes = ptw_workspace_bytes(f);
        ctx->stats.paths += ptw_valid_local_pixels(f, p) * p->spp_per_frame * p->frame_count;  // samples started
    }
    if (j.profile) {
        for (size_t i = 0; i + 1 < j.ev_extend.size(); i += 2) {
            float a = 0.f, b = 0.f;
            if (hipEventElapsedTime(&a, j.ev_extend[i], j.ev_extend[i + 1]) == hipSuccess) ctx->stats.ms_extend += a;
            if (hipEventElapsedTime(&b, j.ev_shade[i], j.ev_shade[i + 1]) == hipSuccess) ctx->stats.ms_shade += b;
        }
    }
    return PT_OK;
}

// ---- PT_PIPELINE_FUSED (fused.hip, fused_kernel.h) ---------------------------------------------------------------------------
// The shape of a fused render: frames in flight as the wavefront pipeline batches them (<= 32, equal batches), and ONE sample group from
// two frames per launch on; a single frame is cut into 32 groups (one-sample slots at the reference's 32 spp).  What decides is the drain of a launch -- the slots
// handed out last run alone, and a slot of 32 samples is up to 256 rays = ~2.5 ms of a lane's time against ~5.5 ms for a frame -- against
// what groups cost (a 16-B store per radiance term and a log k_resolve replays: ~10 % of a frame).  1080p Cornell box, ms per frame, with
// one-tile batches (fused_kernel.h PT_FUSED_BATCH1): 1 frame 7.2 / 6.9 / 6.4 with 1 / 8 / 32 groups; 2 frames 6.21 / 6.27 with 1 / 16;
// 4 frames 5.78 / 6.15 with 1 / 8 (profiles/r05c_fused_batch1.log, r05d_grouped_cost.log; with the 256-slot batches of round 4 groups
// paid up to 8 frames: r04k_fused_groups_by_frames.log).  Explicit frames_in_flight / sample_groups are taken as given.
// 32 x the head slots that are WALKED (frames x owned pixels that can see the scene: the tiles of the cull rectangle, or all of them where there
// is none) per lane of the fused grid: what the shape rules below go by.
uint32_t fused_slots_x32(const pt_ctx *ctx, const pt_film *f, const pt_params *p, uint32_t frames, const int32_t rect[4])
{
    uint64_t tiles = (uint64_t)((f->w + 7) / 8) * ((f->h + 7) / 8);
    if (rect[2] >= rect[0] && rect[3] >= rect[1] && ctx->tune.cull != 0)
        tiles = (uint64_t)(rect[2] / 8 - rect[0] / 8 + 1) * (uint64_t)(rect[3] / 8 - rect[1] / 8 + 1);
    const uint64_t world = std::max(p->world, 1u);
    const uint64_t heads = (uint64_t)frames * 64ull * ((tiles + world - 1) / world);
    const uint64_t grid_lanes = (uint64_t)std::max(ctx->num_cus, 1) * 6ull * 256ull;  // (six 256-thread workgroups per CU: fused.hip)
    return (uint32_t)std::min<uint64_t>(heads * 32ull / grid_lanes, 1u << 24);
}

void fused_shape_defaults(const pt_film *f, const pt_params *p, const FusedPlan &fp, const int32_t rect[4], pt_params &q)
{
    q = *p;
    if (q.frames_in_flight == 0) {
        const uint32_t cap = 32;
        const uint32_t batches = (p->frame_count + cap - 1) / cap;
        q.frames_in_flight = (p->frame_count + batches - 1) / batches;
    }
    q.frames_in_flight = std::max(1u, std::min(q.frames_in_flight, p->frame_count));
    if (q.sample_groups == 0) {
        // every sample its own slot (the smallest divisor of spp that is >= 32, or spp itself) where the launch has little more than one walked
        // slot per lane of the grid -- small films, whatever the frames: 256 x 256 x 4 frames 1.23 ms against 2.50 with one group -- and for a
        // single frame up to 10 per lane (1080p, 3 per lane: 6.43 against 7.24; 2160p, 12 per lane: 24.5 against 23.2, so not there); else one
        // group.  profiles/r05z4_tail_rule.log.  (Where fused_tail_samples below gives S > 0 the head + tail shape replaces either.)
        const uint32_t x32 = fused_slots_x32(f->ctx, f, p, q.frames_in_flight, rect);
        uint32_t g = 1;
        if (x32 < 36u || (q.frames_in_flight < 2u && x32 <= 324u))
            while (g < p->spp_per_frame && (g < 32u || p->spp_per_frame % g)) g++;
        // two-level scenes have no head + tail form: up to four walked slots per lane their slots are cut in eight (the smallest divisor of spp
        // that is >= 8).  The 10 000-instance grid, ms per frame with 1 / 8 / 32 groups: 2 frames 12.90 / 11.57 / 11.76, 4 frames 11.18 / 11.28 /
        // 11.60 (profiles/r05zk_c4_shapes.log)
        else if (fp.inst && x32 <= 130u)
            while (g < p->spp_per_frame && (g < 8u || p->spp_per_frame % g)) g++;
        q.sample_groups = g;
    }
}

// Tail samples per pixel of a fused launch (0: the one-group or all-groups shape above).  A launch with few slots per lane of the grid ends
// with whole 32-sample slots still running; with S of a pixel's samples as one-sample tail slots handed out after every head the launch ends with
// short work, and only those S samples' radiance terms go through the log.  Measured on one MI355X, the Cornell box (58 % of the tiles can see
// it: 3.05 walked slots per lane and 1080p frame), spp 32, depth 8, ms per call as all groups / one group / best S
// (profiles/r05z4_tail_rule.log, r05z_fused_tail_samples.log, r05z2_..., r05z3_... before the cull; r05zf_cull.log with it):
//   walked slots per lane  1.3 (720p x 1)  1.5 (540p x 2)  2.7 (720p x 2)  3.0 (1080p x 1)  6.1 (1080p x 2)  9.1 (1080p x 3)  12.2 (1080p x 4)
//   all / one              3.09 4.19       3.60 4.53       5.80 6.80       6.43 7.24        -    12.40       -    17.64       -     23.0
//   best S                 3.07 (S 24)     3.39 (S 20)     5.72 (S 20)     6.29 (S 16)      11.92 (S 12)     17.35 (S 8)      22.9 (S 4)
//   with the cull          .               .               .               7.11 / 6.15 (16) 12.37 / 11.67 (12) 17.39 / 16.94 (8) 22.70 / 22.32 (4)
// (a rank of world 8 at 16 frames, 6.1: 12.38 -> 12.19; of 4 at 8 frames: 12.38 -> 12.06; 8 and 16 frames: S 2 .. 4 within 0.3 % of none.)
// In single steps with the cull (r05zr_tail_fine.log): one frame S 15 .. 16, two frames S 10 (11.51 against 11.55 at 12), three S 8, four S 6
// (22.11 against 22.22 at 4).
// Round 6, the rule against hand-picked shapes at sizes it was not fitted on (1280 x 720, 1024 x 1024 -- the reference's own launch --, 2560 x 1440,
// 3840 x 2160; K = 1, 2, 20; scripts/probe_shape_rules.py, profiles/r06h_shape_rules.log): within 1 % of the best of twelve shapes everywhere (S 24 against
// the rule's S 20 on one small frame: 2.82 / 2.85 and 2.85 / 2.83 ms in two passes -- noise); nothing changed.
// So by walked slots per lane: under 1.1 -> 0 (all groups); to 2.5 -> 5 spp / 8; to 4.5 -> spp / 2; to 7.9 -> 5 spp / 16; to 10.1 -> spp / 4;
// above -> 0.  (Round 5 had two more steps, 3 spp / 16 up to 13.5 and spp / 8 up to 21.9 per lane: on round 6's kernel -- 17 % faster, its launches end
// sooner -- they cost what they gained or more: 4 / 5 / 6 / 7 frames with S 6 / 4 / 4 / 4 against none 19.43 / 23.89 / 28.51 / 33.10 against 19.44 / 23.62 /
// 28.02 / 32.39 ms, a rank of world 8 at 32 frames 19.69 against 19.15, of world 4 at 16 frames 19.52 against 19.19: profiles/r06u_tail_rule_refit.log.)
// pt_tuning.fused_tail >= 0 overrides.
uint32_t fused_tail_samples(const pt_ctx *ctx, const pt_film *f, const pt_params *p, uint32_t frames, const int32_t rect[4])
{
    const uint32_t spp = p->spp_per_frame;
    if (spp < 2u) return 0u;
    int t = ctx->tune.fused_tail;
    if (t < 0) {
        const uint32_t x = fused_slots_x32(ctx, f, p, frames, rect);
        t = x < 36u ? 0 : x <= 81u ? (int)(spp * 5u / 8u) : x <= 144u ? (int)(spp / 2u) : x <= 252u ? (int)(spp * 5u / 16u) : x <= 324u ? (int)(spp / 4u) : 0;
    }
    return (uint32_t)std::max(0, std::min<int>(t, (int)spp - 1));
}

pt_status render_fused(pt_scene *s, pt_film *f, const pt_params *p_in, const ExtendPlan &pl, bool nested, bool prepare_only)
{
    pt_ctx *ctx = s->ctx;
    hipStream_t st = ctx->stream;
    if (p_in->width > 0xFFFFu || p_in->height > 0xFFFFu) {  // (the kernel keeps a path's pixel as two 16-bit halves of one LDS word)
        ctx->err = "PT_PIPELINE_FUSED renders images up to 65535 x 65535";
        return PT_ERR_UNSUPPORTED;
    }
    FusedPlan fp;
    pt_status rc_ = ptw_plan_fused(s, pl, p_in->tmin, fp);
    if (rc_ != PT_OK) return rc_;
    if (p_in->flags & PT_FLAG_COUNT_VISITS) {  // the instrumented twin of the single-level kernel (wave-level block counts: pt_get_block_counts)
        if (fp.inst || !fp.pairs) {
            ctx->err = "PT_FLAG_COUNT_VISITS on the fused pipeline: single-level scenes with pair leaves only (the two-level kernel has no instrumented form)";
            return PT_ERR_UNSUPPORTED;
        }
        fp.count = true;
    }
    int32_t rect[4];
    subject_rect(s, p_in, rect);
    pt_params q;
    fused_shape_defaults(f, p_in, fp, rect, q);
    const pt_params *p = &q;
    RenderShape sh;
    // HEAD + TAIL slots (fused_kernel.h MODE 2): where the library picks the groups itself and the launch holds few frames, a pixel's frame is one
    // head slot of spp - S samples and S one-sample tail slots -- the launch ends with short work and only the tail's radiance terms go through
    // the log (see fused_tail_samples for S).
    const uint32_t tail = (!fp.inst && !nested && p_in->sample_groups == 0) ? fused_tail_samples(ctx, f, p_in, q.frames_in_flight, rect) : 0u;
    if (tail) {
        q.sample_groups = 1;
        sh = RenderShape{};
        sh.lanes = q.frames_in_flight; sh.groups = 1; sh.group_size = p->spp_per_frame; sh.tail = tail;
        const uint32_t worst = p->max_depth;                       // a tail slot is one sample: at most one term per ray
        sh.term_pcap = std::min(worst, 3u);
        const uint64_t n_tail = (uint64_t)sh.lanes * tail * (uint64_t)((f->w + 7) / 8) * ((f->h + 7) / 8) * 64ull / std::max(p->world, 1u) + 64;
        uint64_t ocap = std::min<uint64_t>(worst - sh.term_pcap, (1ull << 30) / std::max<uint64_t>(n_tail * sizeof(float4), 1));
        if (ctx->tune.term_ocap >= 0) ocap = std::min<uint64_t>(ocap, (uint64_t)ctx->tune.term_ocap);
        sh.term_cap = sh.term_pcap + (uint32_t)ocap;
        sh.bounded = sh.term_cap < worst;
        rc_ = ptw_ensure_work(f, p->rank, p->world, sh.lanes, 1, sh.term_cap, sh.term_pcap, false, tail);
        if (rc_ == PT_ERR_OOM) {  // (no room for the logs: the plain shape)
            fused_shape_defaults(f, p_in, fp, rect, q);
            sh = RenderShape{};
        }
    }
    for (; !sh.tail;) {
        rc_ = ptw_shape_and_work(f, p, sh, 3, false);
        if (rc_ != PT_ERR_OOM) break;
        // a shape this function chose (the caller passed 0) and that does not fit is planned again smaller, like the wavefront's AUTO
        // shapes: first fewer sample groups (the next smaller divisor of spp), then fewer frames in flight
        if (p_in->sample_groups == 0 && q.sample_groups > 1) {
            uint32_t g = q.sample_groups - 1;
            while (g > 1 && p_in->spp_per_frame % g) g--;
            q.sample_groups = g;
        } else if (p_in->frames_in_flight == 0 && q.frames_in_flight > 1) {
            q.frames_in_flight = (q.frames_in_flight + 1) / 2;
        } else {
            break;
        }
    }
    if (!nested) {
        ctx->stats.frames_in_flight = sh.lanes;
        ctx->stats.sample_groups = sh.groups;
        ctx->stats.tail_samples = sh.tail;
    }
    if (rc_ != PT_OK) return rc_;
    ctx->stats.workspace_bytes = ptw_workspace_bytes(f);
    if (prepare_only) return PT_OK;
    {
        const int32_t none[4] = { 0, 0, -1, -1 };
        rc_ = ptw_tiles_subject_first(f, ctx->tune.fused_subject == 0 ? none : rect, st);
        if (rc_ != PT_OK) return rc_;
    }
    pt_film::Work &w = f->work;
    unsigned long long *const d_overflow = ctx->d_stats + 6, *const d_spill_count = ctx->d_stats + 7;
    uint32_t spill_cap = sh.bounded ? SPILL_POOL_ENTRIES : 0u;
    if (ctx->tune.term_spill >= 0) spill_cap = std::min<uint32_t>(spill_cap, (uint32_t)ctx->tune.term_spill);
    static_assert(sizeof(Radiance) <= sizeof(pt_ctx::h_rad), "pt_ctx::h_rad holds a Radiance");
    // (the record the kernel gets by value, and the same in device memory for the ends of the log it rarely takes: Radiance::dev)
    Radiance rad{};
    auto set_rad = [&]() -> pt_status {
        rad = { w.d_color, w.d_terms, w.d_terms_over, w.d_nterm, w.d_spill, w.d_spill_head, d_spill_count, spill_cap, d_overflow, static_cast<const Radiance *>(ctx->d_rad) };
        if (std::memcmp(ctx->h_rad, &rad, sizeof(rad)) != 0) {
            std::memcpy(ctx->h_rad, &rad, sizeof(rad));
            PT_HIP(ctx, hipMemcpyAsync(ctx->d_rad, ctx->h_rad, sizeof(rad), hipMemcpyHostToDevice, st));
        }
        return PT_OK;
    };
    rc_ = set_rad();
    if (rc_ != PT_OK) return rc_;
    RenderConst rc = ptw_render_const(p, w, sh);
    apply_cull(ctx, p, rect, rc);
    const bool profile = !nested && (p->flags & PT_FLAG_PROFILE) != 0;
    ctx->stats.extend_variant = pl.variant;
    ctx->stats.pipelines = 1;
    if (!nested) PT_HIP(ctx, hipEventRecord(ctx->ev_a, st));
    std::vector<hipEvent_t> evs;
    size_t ev_used = 0;
    for (uint32_t done = 0; w.n_slots > 0 && done < p->frame_count; done += sh.lanes) {
        rc.frame_base = p->frame + (int32_t)done;
        rc.lanes_active = std::min(sh.lanes, p->frame_count - done);
        // A launch whose term logs are bounded can overflow them (scenes where most surfaces emit): k_fused raises d_overflow, k_resolve then
        // leaves the film alone, and the host -- which looks at the flag once the stream is idle anyway: after the call's last launch, or
        // before the next launch of a longer call, whose blend must follow this one's -- renders the same frames again with one group.
        // Nothing here waits for the device before the launch: the ray counter to restore is kept in a spare device word.
        unsigned long long *const d_rays_before = ctx->d_stats + 18;
        if (sh.bounded) {
            PT_HIP(ctx, hipMemcpyAsync(d_rays_before, ctx->d_stats, sizeof(unsigned long long), hipMemcpyDeviceToDevice, st));
            PT_HIP(ctx, hipMemcpyAsync(ctx->d_stats + 20, ctx->d_stats + 19, sizeof(unsigned long long), hipMemcpyDeviceToDevice, st));  // (rays_culled)
            PT_HIP(ctx, hipMemsetAsync(d_spill_count, 0, sizeof(unsigned long long), st));
        }
        PT_HIP(ctx, hipMemsetAsync(w.d_count, 0, sizeof(uint32_t) * PTW_COUNT_WORDS, st));  // the slot counters (fused_kernel.h: eight, 128 B apart)
        const uint32_t n_slots = rc.lanes_active * sh.groups * rc.slots_per_lane;  // (head + tail: the launch's head slots; its tail slots follow from rc)
        hipEvent_t e0 = nullptr, e1 = nullptr;
        if (profile) {
            rc_ = grow_event_pool(ctx, ev_used + 2);
            if (rc_ != PT_OK) return rc_;
            e0 = ctx->ev_pool[ev_used++]; e1 = ctx->ev_pool[ev_used++];
            evs.push_back(e0); evs.push_back(e1);
        }
        ptw_launch_fused(fp, sh.groups > 1, rc, w.d_tiles, rad, s, n_slots, w.d_count, ctx->d_stats, p->tmin, p->tmax, st, e0, e1);  // (rc.tail != 0: the head + tail kernel)
        PT_HIP(ctx, hipGetLastError());
        ctx->stats.launches_extend++;
        ctx->stats.rounds++;
        ptw_launch_resolve(rc, w.d_tiles, rad, f->d_rgb, f->d_bgra, st, sh.bounded ? d_overflow : nullptr);
        ctx->stats.launches_other++;
        bool redo = false;
        if (sh.bounded) {
            unsigned long long flag = 0;
            PT_HIP(ctx, hipStreamSynchronize(st));  // (the last launch of a blocking call: the wait the call ends with)
            PT_HIP(ctx, hipMemcpy(&flag, d_overflow, sizeof(flag), hipMemcpyDeviceToHost));
            redo = flag != 0ull;
        }
        if (redo) {  // a slot filled its term log: the same frames once more with one group
            PT_HIP(ctx, hipMemcpy(ctx->d_stats, d_rays_before, sizeof(unsigned long long), hipMemcpyDeviceToDevice));
            PT_HIP(ctx, hipMemcpy(ctx->d_stats + 19, ctx->d_stats + 20, sizeof(unsigned long long), hipMemcpyDeviceToDevice));
            PT_HIP(ctx, hipMemset(d_overflow, 0, sizeof(unsigned long long)));
            ctx->stats.redone_batches++;
            pt_params r = *p;
            r.frame = rc.frame_base; r.frame_count = rc.lanes_active; r.frames_in_flight = rc.lanes_active; r.sample_groups = 1;
            rc_ = render_fused(s, f, &r, pl, true, false);
            if (rc_ != PT_OK) return rc_;
            rc_ = ptw_ensure_work(f, p->rank, p->world, sh.lanes, sh.groups, sh.term_cap, sh.term_pcap, false, sh.tail);
            if (rc_ != PT_OK) return rc_;
            rc_ = set_rad();
            if (rc_ != PT_OK) return rc_;
        }
    }
    if (!nested) PT_HIP(ctx, hipEventRecord(ctx->ev_b, st));
    PT_HIP(ctx, hipStreamSynchronize(st));
    PT_HIP(ctx, hipGetLastError());
    if (!nested) {
        float ms = 0.f;
        PT_HIP(ctx, hipEventElapsedTime(&ms, ctx->ev_a, ctx->ev_b));
        ctx->stats.ms_total += ms;
        ctx->stats.workspace_bytes = ptw_workspace_bytes(f);
        ctx->stats.paths += ptw_valid_local_pixels(f, p) * p->spp_per_frame * p->frame_count;
    }
    for (size_t i = 0; i + 1 < evs.size(); i += 2) {
        float a = 0.f;
        if (hipEventElapsedTime(&a, evs[i], evs[i + 1]) == hipSuccess) ctx->stats.ms_extend += a;
    }
    return PT_OK;
}

// PT_PIPELINE_AUTO (what pt_params_default returns): the reference's raygen shader is one invocation per pixel that owns its path
// (raygen.rgen:41-91) and its host issues one blocking dispatch per frame (main.cpp:656-683) -- the fused kernel is that shape, and
// wherever it applies it is the faster bit-exact pipeline at every call shape measured (1080p Cornell box, K = 16 / 2 / 1 frames per call:
// 41.2 / 36.0 / 34.9 Grays/s against the wavefront's 28 / 27.5 / 26; a rank of world 8 at 16 frames 36 against 25; the 10 000-instance grid
// 14.9 against 13.9: profiles/r05c_fused_batch1.log) in a workspace of 16 B per slot.  So: fused for the scenes fused.hip plans (they live in
// LDS) when the call is one it can serve (blocking, not instrumented, tmin > 0, image <= 65535^2, the closest-hit kernel left to AUTO);
// the wavefront pipeline for everything else -- big scenes need its queues, sorting and long launches.
uint32_t resolve_pipeline(pt_scene *s, const pt_params *p, const ExtendPlan &pl)
{
    if (p->pipeline != PT_PIPELINE_AUTO) return p->pipeline;
    if ((p->flags & (PT_FLAG_ASYNC | PT_FLAG_COUNT_VISITS)) || p->extend != PT_EXTEND_AUTO || p->width > 0xFFFFu || p->height > 0xFFFFu)
        return PT_PIPELINE_WAVEFRONT;
    FusedPlan fp;
    const std::string keep = s->ctx->err;
    const pt_status rc = ptw_plan_fused(s, pl, p->tmin, fp);
    if (rc != PT_OK) s->ctx->err = keep;  // (not an error of this call: the scene is simply not the fused kernel's)
    if (rc != PT_OK) return PT_PIPELINE_WAVEFRONT;
    // Two-level scenes: the fused two-level kernel, in 0.4 .. 5 GB.  With the pixels that look past the instances out of its queues (the cull)
    // the wavefront pipeline is the faster one on launches of few frames -- the 10 000-instance grid at 1080p, ms per frame fused / wavefront:
    // 1 frame 11.95 / 10.40, 2 frames 11.41 / 10.98, 4 frames 11.08 / 10.55, 8 frames 10.49 / 10.39, 16 frames 10.21 / 10.30
    // (profiles/r06d_c4_pipelines.log) -- but it takes 13 .. 37 GB for that, and under the library's own workspace budget (8 GB,
    // pt_internal.h) its shapes shrink until the fused kernel is ahead at every frame count (8 frames: 15.0 against 15.5 Grays/s,
    // profiles/r06e_mem_budget.log).  So the queues only where they are >= 15 % ahead AND the caller has raised the budget to what they
    // take: one frame per launch with >= 16 GB.
    if (fp.inst) {
        const uint32_t per_launch = p->frames_in_flight ? std::min(p->frames_in_flight, p->frame_count) : std::min(p->frame_count, 32u);
        const size_t budget = s->ctx->mem_budget;
        if (per_launch == 1u && (budget == 0 || budget >= ((size_t)16 << 30))) return PT_PIPELINE_WAVEFRONT;
    }
    return PT_PIPELINE_FUSED;
}

}  // namespace

// pt_render_prepare: exactly the shape and workspace pt_render would pick, and the one-time objects of its schedule
pt_status ptw_prepare(pt_scene *s, pt_film *f, const pt_params *p_in)
{
    pt_ctx *ctx = s->ctx;
    pt_status rc_ = check_params(s, f, p_in);
    if (rc_ != PT_OK) return rc_;
    ExtendPlan pl;
    rc_ = ptw_plan_extend(s, p_in->extend, pl);
    if (rc_ != PT_OK) return rc_;
    pt_params p_res = *p_in;
    p_res.pipeline = resolve_pipeline(s, p_in, pl);
    const pt_params *p = &p_res;
    ctx->stats.pipeline = p->pipeline;
    if (p->pipeline == PT_PIPELINE_FUSED) return render_fused(s, f, p, pl, false, true);
    RenderShape sh;
    rc_ = ptw_shape_and_work(f, p, sh, launch_class(s, pl));
    ctx->stats.frames_in_flight = sh.lanes;
    ctx->stats.sample_groups = sh.groups;
    if (rc_ != PT_OK) return rc_;
    ctx->stats.workspace_bytes = ptw_workspace_bytes(f);
    rc_ = ensure_schedule_objects(ctx, 3);
    if (rc_ != PT_OK) return rc_;
    if (p->flags & PT_FLAG_PROFILE) {  // the pooled (start, stop) events of every extend / shade launch of one batch (4 per round and pipeline)
        const size_t batches = ((size_t)p->frame_count + sh.lanes - 1) / sh.lanes;
        rc_ = grow_event_pool(ctx, std::min<size_t>(4ull * sh.group_size * p->max_depth * 3ull * batches, 1u << 16));
        if (rc_ != PT_OK) return rc_;
    }
    return PT_OK;
}

pt_status ptw_render(pt_scene *s, pt_film *f, const pt_params *p_in)
{
    pt_status rc_ = check_params(s, f, p_in);
    if (rc_ != PT_OK) return rc_;
    ExtendPlan pl;
    rc_ = ptw_plan_extend(s, p_in->extend, pl);
    if (rc_ != PT_OK) return rc_;
    pt_params p_res = *p_in;
    p_res.pipeline = resolve_pipeline(s, p_in, pl);
    const pt_params *p = &p_res;
    s->ctx->stats.pipeline = p->pipeline;
    if (p->pipeline == PT_PIPELINE_FUSED) return render_fused(s, f, p, pl, false, false);
    return render_wavefront(s, f, p, pl, false);
}

pt_status ptw_trace(pt_scene *s, const float *rays6, uint32_t n, float tmin, float tmax, uint32_t extend, pt_hit *hits)
{
    pt_ctx *ctx = s->ctx;
    hipStream_t st = ctx->stream;
    if (n == 0) return PT_OK;
    ExtendPlan pl;
    pt_status rc_ = ptw_plan_extend(s, extend, pl);
    if (rc_ != PT_OK) return rc_;
    std::vector<float4> a(n);
    std::vector<float2> b(n);
    for (uint32_t i = 0; i < n; i++) {
        const float *r = rays6 + 6 * (size_t)i;
        a[i] = make_float4(r[0], r[1], r[2], r[3]);
        b[i] = make_float2(r[4], r[5]);
    }
    float4 *d_a = nullptr, *d_hit = nullptr;
    float2 *d_b = nullptr;
    uint32_t *d_cnt = nullptr, *d_hi = nullptr;
    pt_hit *d_out = nullptr;
    pt_status ret = PT_OK;
    auto fail = [&](hipError_t e, const char *what) {
        ctx->err = std::string(what) + ": " + hipGetErrorString(e);
        ret = PT_ERR_HIP;
    };
    hipError_t e;
    if ((e = hipMalloc((void **)&d_a, sizeof(float4) * n)) != hipSuccess) fail(e, "hipMalloc");
    if (ret == PT_OK && (e = hipMalloc((void **)&d_b, sizeof(float2) * n)) != hipSuccess) fail(e, "hipMalloc");
    if (ret == PT_OK && (e = hipMalloc((void **)&d_hit, sizeof(float4) * n)) != hipSuccess) fail(e, "hipMalloc");
    if (ret == PT_OK && (e = hipMalloc((void **)&d_out, sizeof(pt_hit) * n)) != hipSuccess) fail(e, "hipMalloc");
    if (ret == PT_OK && (e = hipMalloc((void **)&d_cnt, sizeof(uint32_t) * 2)) != hipSuccess) fail(e, "hipMalloc");
    if (ret == PT_OK && (e = hipMalloc((void **)&d_hi, sizeof(uint32_t) * n)) != hipSuccess) fail(e, "hipMalloc");
    if (ret == PT_OK) {
        (void)hipMemcpyAsync(d_a, a.data(), sizeof(float4) * n, hipMemcpyHostToDevice, st);
        (void)hipMemcpyAsync(d_b, b.data(), sizeof(float2) * n, hipMemcpyHostToDevice, st);
        const uint32_t cnt_head[2] = { n, 0u };
        (void)hipMemcpyAsync(d_cnt, cnt_head, sizeof(cnt_head), hipMemcpyHostToDevice, st);
        (void)hipEventRecord(ctx->ev_a, st);
        ptw_launch_extend(pl, s, d_a, d_b, d_hit, d_hi, d_cnt, nullptr, ctx->d_stats, tmin, tmax, false, false, st);
        (void)hipEventRecord(ctx->ev_b, st);
        ptw_launch_hits_to_api(d_hit, pl.bvh8 ? s->d_tri4_8 : s->d_tri4, s->n_inst ? d_hi : nullptr, s->d_tlas_prim_of, n, d_out, st);
        (void)hipMemcpyAsync(hits, d_out, sizeof(pt_hit) * n, hipMemcpyDeviceToHost, st);
        if ((e = hipStreamSynchronize(st)) != hipSuccess) fail(e, "pt_trace");
        else if ((e = hipGetLastError()) != hipSuccess) fail(e, "pt_trace");
        float ms = 0.f;
        if (ret == PT_OK && hipEventElapsedTime(&ms, ctx->ev_a, ctx->ev_b) == hipSuccess) ctx->stats.ms_extend += ms;
        ctx->stats.launches_extend++;
    }
    (void)hipFree(d_a); (void)hipFree(d_b); (void)hipFree(d_hit); (void)hipFree(d_out); (void)hipFree(d_cnt); (void)hipFree(d_hi);
    return ret;
}
