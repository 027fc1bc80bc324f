// shade_kernels.hip -- everything of the wavefront pipeline that is not the closest-hit query: the kernels that restate
// raygen.rgen:41-91 around it, closesthit.rchit:50-65 and miss.rmiss:8-12.
//
//     k_generate    sample 0 of every slot of a batch -> queue 0                                  (raygen.rgen:45-60)
//     k_shade       hit[q] -> emission / environment radiance, bounce, or the slot's next sample; survivors compacted into
//                   the other queue with wave ballots + one atomic per 512 paths                  (raygen.rgen:76-83)
//     k_shadow_add  PT_PIPELINE_WAVEFRONT_NEE: the unoccluded light samples of a round
//     k_resolve     replay of the term logs in sample order, /spp, the progressive blend, rgba8     (raygen.rgen:86-90)
//     k_hits_to_api pt_trace: hit records in the public layout
// One round of a pipeline = one closest-hit launch (extend_launch.hip) + one k_shade launch; render.hip schedules them.
#include "wavefront_host.h"

#include <algorithm>

#include "fused_cull.h"

namespace {
using namespace ptw;

// ---- generate: sample 0 of every (frame, pixel) slot of the batch ----------------------------
__global__ __launch_bounds__(TB) void k_generate(RenderConst rc, const uint32_t *__restrict__ tiles, uint32_t slot_base,
                                                 uint32_t n_slots, Radiance rad, QueueView out, uint32_t *count_out, unsigned long long *stats)
{
    uint32_t n_culled = 0;  // camera rays of the slots this thread finished without a walk (RenderConst::cull, fused_cull.h)
    // four slots per thread and ONE queue-tail atomic per 1024 slots, as in k_shade: with one atomic per 256 slots
    // the 133 M slots of 16 frames x 4 sample groups spent 2.9 ms per launch on the ~88 atomics/us a single word takes
    constexpr int GEN_ITEMS = 4;
    constexpr uint32_t CHUNK = TB * GEN_ITEMS;
    __shared__ uint32_t s_wcnt[GEN_ITEMS][4];
    __shared__ uint32_t s_base;
    for (uint32_t base = blockIdx.x * CHUNK; base < n_slots; base += gridDim.x * CHUNK) {
        bool alive[GEN_ITEMS];
        uint32_t o_slot[GEN_ITEMS], o_seed[GEN_ITEMS], o_sample[GEN_ITEMS];
        ptm::f3 o_org[GEN_ITEMS], o_dir[GEN_ITEMS];
#pragma unroll
        for (int it = 0; it < GEN_ITEMS; it++) {
            const uint32_t local = base + it * TB + threadIdx.x;
            const uint32_t slot = slot_base + local;
            alive[it] = false;
            o_slot[it] = slot; o_seed[it] = 0u; o_sample[it] = 0u; o_org[it] = {}; o_dir[it] = {};
            if (local < n_slots) {
                uint32_t f, g, px, py;
                slot_pixel(rc, tiles, slot, f, g, px, py);
                if (rc.groups == 1u) rad.color[slot] = make_float4(0.f, 0.f, 0.f, 0.f);  // raygen.rgen:42
                else {
                    rad.nterm[slot] = 0u;
                    rad.spill_head[slot] = SPILL_NONE;
                }
                const uint32_t sample0 = g * rc.group_size;
                o_sample[it] = sample0;
                if (px < rc.width && py < rc.height && f < rc.lanes_active && sample0 < rc.spp) {
                    if (ptc::pixel_culled(rc, px, py)) {
                        // the pixel cannot see the scene: each of the slot's samples is one camera ray (counted) that misses and adds 1 * env -- the
                        // slot never enters the queues
                        n_culled += rc.groups == 1u ? ptc::finish_plain(rc, rad, slot) : ptc::finish_group(rc, rad, slot, g);
                    } else {
                        alive[it] = true;
                        o_seed[it] = ptm::make_seed(px, py, sample0, rc.frame_base + (int32_t)f, rc.spp);
                        ptm::primary_ray(rc.cam, px, py, o_seed[it], o_org[it], o_dir[it]);
                    }
                }
            }
        }
        uint32_t dst[GEN_ITEMS];
        chunk_offsets<GEN_ITEMS>(alive, dst, count_out, s_wcnt, &s_base);
#pragma unroll
        for (int it = 0; it < GEN_ITEMS; it++) {
            if (alive[it]) {
                ptm::st_stream<PT_NT_GEN>(out.id + dst[it], make_uint2(o_slot[it], o_sample[it]));
                ptm::st_stream<PT_NT_GEN>(out.state + dst[it], make_float4(__uint_as_float(o_seed[it]), 1.f, 1.f, 1.f));  // raygen.rgen:59
                ptm::st_stream<PT_NT_GEN>(out.rayA + dst[it], make_float4(o_org[it].x, o_org[it].y, o_org[it].z, o_dir[it].x));
                ptm::st_stream<PT_NT_GEN>(out.rayB + dst[it], make_float2(o_dir[it].y, o_dir[it].z));
            }
        }
    }
    if (rc.cull_on) {  // (uniform: every thread of the block is here)
        for (int o = 32; o > 0; o >>= 1) n_culled += (uint32_t)__shfl_xor((int)n_culled, o, 64);
        if ((threadIdx.x & 63u) == 0u && n_culled) {
            atomicAdd(stats, (unsigned long long)n_culled);        // pt_stats.rays
            atomicAdd(stats + 19, (unsigned long long)n_culled);   // pt_stats.rays_culled
        }
    }
}

// One light sample for the hit at `pos` (normal n, brdf, path weight w); the operations and their order are part of the
// pipeline's definition (the CPU checker of the tests restates them, and the two agree bit for bit).  Returns false when no shadow ray is needed.
__device__ __forceinline__ bool nee_sample(const float4 *__restrict__ lights, uint32_t n_lights, float light_area, uint32_t &seed,
                                           const ptm::f3 pos, const ptm::f3 n, float br, float bg, float bb, float wr, float wg,
                                           float wb, ptm::f3 &wi, float4 &contrib)
{
    const float rl = ptm::rnd(seed), ru = ptm::rnd(seed), rv = ptm::rnd(seed);
    const float pick = rl * light_area;
    // first emitter whose running area exceeds pick (the last one if none does): binary search of the cdf
    uint32_t li = 0, hi_ = n_lights - 1u;
    while (li < hi_) {
        const uint32_t mid = (li + hi_) >> 1;
        if (lights[5 * (size_t)mid].w > pick) hi_ = mid; else li = mid + 1u;
    }
    const float4 A = lights[5 * (size_t)li + 0], B = lights[5 * (size_t)li + 1], C = lights[5 * (size_t)li + 2],
                 N = lights[5 * (size_t)li + 3], Ke = lights[5 * (size_t)li + 4];
    const float su = ptm::fsqrt(ru);
    const float b0 = 1.0f - su, b1 = su * (1.0f - rv), b2 = su * rv;
    const float dx = ((A.x * b0 + B.x * b1) + C.x * b2) - pos.x, dy = ((A.y * b0 + B.y * b1) + C.y * b2) - pos.y,
                dz = ((A.z * b0 + B.z * b1) + C.z * b2) - pos.z;
    const float d2 = (dx * dx + dy * dy) + dz * dz;
    if (!(d2 > 0.0f)) return false;
    const float dist = ptm::fsqrt(d2);
    ptm::div3_dominant(dx, dy, dz, dist, wi.x, wi.y, wi.z);
    const float cs = (wi.x * n.x + wi.y * n.y) + wi.z * n.z;
    const float cl = fabsf((wi.x * N.x + wi.y * N.y) + wi.z * N.z);
    if (!(cs > 0.0f && cl > 0.0f)) return false;
    const float fgeo = ptm::fdiv(cs * cl, d2) * light_area;
    contrib = make_float4(((wr * br) * Ke.x) * fgeo, ((wg * bg) * Ke.y) * fgeo, ((wb * bb) * Ke.z) * fgeo, dist * 0.999f);
    return true;
}

// after the shadow rays were traced: the contributions of those that reached their light
__global__ __launch_bounds__(TB) void k_shadow_add(RenderConst rc, Radiance rad, const float4 *__restrict__ sq_hit,
                                                   const float4 *__restrict__ contrib, const uint32_t *__restrict__ slot,
                                                   const uint32_t *__restrict__ count)
{
    const uint32_t n = *count;
    for (uint32_t i = blockIdx.x * TB + threadIdx.x; i < n; i += gridDim.x * TB) {
        if (__float_as_uint(sq_hit[i].x) != PT_MISS) continue;  // occluded
        const float4 c = contrib[i];
        add_radiance(rc, rad, slot[i], c.x, c.y, c.z);
    }
}

// ---- shade: closesthit / miss + the bounce logic of raygen.rgen:76-83, regeneration, compaction
// k_shade: paths per thread and waves per SIMD asked of the compiler.  Measured (C2 / C5 Mrays/s, same box): 4 x 4 waves
// (105 VGPRs) 22 050 / 2 266; 4 x 5 (96 VGPRs, 14 spilled since the term-log tiers) 22 060 / 2 258; 3 x 5 22 260 / 2 271;
// 3 x 6 22 280 / 2 270; 2 x 7 (72 VGPRs, no spills) 22 630 / 2 281 -- all within the run-to-run noise, so the one without
// spills and with the most waves in flight is used.  One queue-tail atomic per 512 paths.
#ifndef PT_SHADE_ITEMS
#define PT_SHADE_ITEMS 2
#endif
// (round 3, after the instancing template -- 70 VGPRs, no spills at 7 waves: ten interleaved processes each, 7 waves median 26.8
// Grays/s, 6 waves 25.8; 5 waves with PT_SHADE_PRELOAD 25.2: profiles/r03ck_ab_c2_shade_7_vs_6_waves.log, r03cj_*)
#ifndef PT_SHADE_WAVES
#define PT_SHADE_WAVES 7
#endif
// (the instanced instantiation carries the position transform and the table gather on top: 25 spilled registers at 7 waves = C4
// -8 %, 8 at 6 waves = +2 %, none at 5 waves / 96 VGPRs = +4 % over the kernel before: profiles/r03cf_ab_c4_inst_frames.log)
#ifndef PT_SHADE_WAVES_INST
#define PT_SHADE_WAVES_INST 5
#endif
// PT_SHADE_PRELOAD=1 requests every queue record of a chunk before the first is used (one memory round trip per chunk
// instead of one per item).  Alone on the chip (one pipeline) k_shade gets 13 % faster with it at 5 waves (91 VGPRs, no
// spills: 104 -> 90 ms per 16 C2 frames); next to the other pipeline's traversal kernel, which is how it runs, nothing
// changes (three interleaved repetitions, profiles/r02_shade_preload.txt) -- the frame is bound by the VALU work of both
// kernels, not by k_shade's latency -- so the simpler code stays the default.  Round 4, at 7 waves, all four configs: C5 / C5x / C4
// within 0.3 %, C2 -1.1 % (profiles/r04o_ab_shade_preload_*.log).
#ifndef PT_SHADE_PRELOAD
#define PT_SHADE_PRELOAD 0
#endif
// INST: the scene is instanced (position and normal go to world space per hit; the single-level instantiations carry none of that code)
template <int SH_ITEMS, bool LDS_TABLES, bool NEE = false, bool INST = false>
__global__ __launch_bounds__(TB, NEE ? 4 : INST ? PT_SHADE_WAVES_INST : PT_SHADE_WAVES) void k_shade(RenderConst rc_arg, const uint32_t *__restrict__ tiles,
                                              const float4 *__restrict__ g_tri4, const float4 *__restrict__ g_shade4,
                                              uint32_t n_tris_arg,
                                              const float4 *__restrict__ hit, Radiance rad_arg, QueueView in_arg,
                                              QueueView out_arg, const uint32_t *__restrict__ count_in, uint32_t *count_out_arg,
                                              const float4 *__restrict__ inst6, const uint32_t *__restrict__ hit_inst,
                                              const float4 *__restrict__ shade64, const float4 *__restrict__ ke4,
                                              const float4 *__restrict__ lights, uint32_t n_lights_arg, float light_area_arg,
                                              ShadowQueue sq_arg, uint32_t *sq_count_arg, const float4 *__restrict__ g_frame4,
                                              const float4 *__restrict__ inst_frame)
{
    // (the records and scalars the slot loop reads: scalar registers of their own -- ptm::own_sgprs; with the argument tuples as the compiler
    // fetches them the instanced variants reloaded 212 .. 270 spilled scalars per pass, a v_readlane_b32 each)
    const RenderConst rc = ptm::own_sgprs(rc_arg);
    const Radiance rad = ptm::own_sgprs(rad_arg);
    const QueueView in = ptm::own_sgprs(in_arg), out = ptm::own_sgprs(out_arg);
    const ShadowQueue sq = ptm::own_sgprs(sq_arg);
    const uint32_t n_tris = ptm::own_sgprs(n_tris_arg), n_lights = ptm::own_sgprs(n_lights_arg);
    const float light_area = ptm::own_sgprs(light_area_arg);
    uint32_t *count_out = ptm::own_sgprs(count_out_arg), *sq_count = ptm::own_sgprs(sq_count_arg);
    __shared__ uint32_t s_wcnt[SH_ITEMS][4];
    __shared__ uint32_t s_base;
    // Under two pipelines the shade launches run back to back -- their durations add up to the wall clock -- while the VALU-bound
    // traversal kernel of the other pipeline fits in between with slack: the shade waves are the critical chain and get issue
    // priority over the traversal waves they share a SIMD with.  Same-box A/B, six rounds: C2 23.59 -> 24.26 Grays/s (+2.9 %,
    // shade 153 -> 138 ms, extend 130 -> 138 ms per 16 frames), C4 +4.6 %; priority 1: none, 2: +1.7 %.  With the tables in HBM
    // (C5 +0.6 %, C5x -0.7 %) the kernel waits for its gathers and keeps the default (profiles/r02i_ab_shade_prio.log).
#ifndef PT_SHADE_PRIO
#define PT_SHADE_PRIO 3
#endif
    if (LDS_TABLES && PT_SHADE_PRIO > 0) __builtin_amdgcn_s_setprio(PT_SHADE_PRIO);
#ifndef PT_SHADE_DENSE_REGEN
#define PT_SHADE_DENSE_REGEN 1
#endif
    // (small scenes only: with the tables in HBM the kernel waits for its gathers, and the extra LDS round trip cost C5 1 %)
    constexpr bool DENSE_REGEN = LDS_TABLES && PT_SHADE_DENSE_REGEN != 0;
    // per wave: one 16-B cell per ended path -- first its job {slot, next sample}, then, written by the lane that took the
    // job, the result {dir.xyz, seed} (dir.x = 2: the slot has no further sample)
    __shared__ float4 s_regen[DENSE_REGEN ? 4 : 1][DENSE_REGEN ? 64 * SH_ITEMS : 1];
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const float4 *tri4 = g_tri4;
    const float4 *shade4 = g_shade4;
    const float4 *frame4 = g_frame4;
    if (LDS_TABLES) {  // small scenes: the per-triangle tables live in LDS, no dependent global gathers
        float4 *s_tri = reinterpret_cast<float4 *>(smem);
        float4 *s_shade = s_tri + 3 * (size_t)n_tris;
        float4 *s_frame = s_shade + 3 * (size_t)n_tris;
        for (uint32_t i = threadIdx.x; i < 3 * n_tris; i += TB) {
            s_tri[i] = g_tri4[i];
            s_shade[i] = g_shade4[i];
        }
        for (uint32_t i = threadIdx.x; i < 2 * n_tris; i += TB) s_frame[i] = g_frame4[i];
        __syncthreads();
        tri4 = s_tri;
        shade4 = s_shade;
        frame4 = s_frame;
    }
    const uint32_t n = *count_in;
    constexpr uint32_t CHUNK = TB * SH_ITEMS;
    for (uint32_t base = blockIdx.x * CHUNK; base < n; base += gridDim.x * CHUNK) {
        bool alive[SH_ITEMS];
        uint32_t o_slot[SH_ITEMS], o_ctr[SH_ITEMS];
        float4 o_state[SH_ITEMS], o_rayA[SH_ITEMS];
        float2 o_rayB[SH_ITEMS];
        bool regen[SH_ITEMS];            // DENSE_REGEN: the path ended, the slot's next sample has to be started
        bool s_alive[SH_ITEMS];          // NEE: a shadow ray for this item
        float4 s_rayA[SH_ITEMS], s_contrib[SH_ITEMS];
        float2 s_rayB[SH_ITEMS];
#if PT_SHADE_PRELOAD
        // every queue record of the chunk is requested before the first one is used: the per-item bodies below store and
        // add to the radiance arrays, which the compiler must assume alias the queue, so without this each item's three
        // loads wait behind the previous item's stores -- one memory round trip per item instead of one per chunk
        uint2 in_id[SH_ITEMS];
        float4 in_st[SH_ITEMS], in_hit[SH_ITEMS];
#pragma unroll
        for (int it = 0; it < SH_ITEMS; it++) {
            const uint32_t q = min(base + it * TB + threadIdx.x, n - 1u);
            in_id[it] = in.id[q];
            in_st[it] = in.state[q];
            in_hit[it] = hit[q];
        }
#endif
#pragma unroll
        for (int it = 0; it < SH_ITEMS; it++) {
            const uint32_t q = base + it * TB + threadIdx.x;
            alive[it] = false;
            regen[it] = false;
            if (NEE) { s_alive[it] = false; o_slot[it] = 0u; }
            if (q >= n) continue;
#if PT_SHADE_PRELOAD
            const uint2 id = in_id[it];
            const float4 st = in_st[it];
            const float4 h = in_hit[it];
#else
            const uint2 id = ptm::ld_stream<INST>(in.id + q);
            const float4 st = ptm::ld_stream<INST>(in.state + q);
            const float4 h = ptm::ld_stream<INST>(hit + q);
#endif
            const uint32_t slot = id.x, ctr = id.y;
            uint32_t sample = ctr & 0xFFFFu, depth = ctr >> 16;
            uint32_t seed = __float_as_uint(st.x);
            float wr = st.y, wg = st.z, wb = st.w;
            const uint32_t pos = __float_as_uint(h.x);
            bool terminated;
            ptm::f3 org{}, dir{};
            if (pos == PT_MISS) {
                // miss.rmiss:10-11 then raygen.rgen:76: color += weight * (0.7,0.6,0.5); break
                add_radiance(rc, rad, slot, wr * rc.env[0], wg * rc.env[1], wb * rc.env[2]);
                terminated = true;
            } else {
                // per-triangle record.  Tables in LDS: {n, brdf.r} {brdf.gb, Ke.rg} {Ke.b} + the three vertices.
                // Tables in HBM: every 16-B load of a wave whose lanes hit different triangles is one L1 look-up
                // per lane, so the record is regrouped (k_pack) into {v0, n.x} {v1, n.y} {v2, n.z} {brdf, emits}
                // + Ke apart: 4 look-ups per hit instead of 6, 1 instead of 3 when the path ends here.
                float4 s0, s1, s2, a{}, b{}, c{};
                if (LDS_TABLES) {
                    s0 = shade4[3 * pos + 0]; s1 = shade4[3 * pos + 1]; s2 = shade4[3 * pos + 2];
                } else {
                    // the three vertex rows of the same 64-B record are requested WITH its fourth row, not after the emission test that
                    // waits for it: one gather round trip per hit instead of two (the second one short -- the line is on its way --
                    // but still a full wait of the wave); a path that ends here does not need them and does not ask.  k_shade -5 %,
                    // C5 +0.8 %, C5x +0.7 %, three of three rounds each (profiles/r04p_ab_shade_hoist_*.log)
                    if (depth + 1u < rc.max_depth) {
                        a = shade64[4 * (size_t)pos + 0]; b = shade64[4 * (size_t)pos + 1]; c = shade64[4 * (size_t)pos + 2];
                    }
                    const float4 r3 = shade64[4 * (size_t)pos + 3];
                    const float4 ke = r3.w != 0.f ? ke4[pos] : make_float4(0.f, 0.f, 0.f, 0.f);
                    s0 = make_float4(0.f, 0.f, 0.f, r3.x); s1 = make_float4(r3.y, r3.z, ke.x, ke.y); s2 = make_float4(ke.z, 0.f, 0.f, 0.f);
                }
                // raygen.rgen:76: color += weight * emission.  Adding +0 changes no bit of a
                // non-negative accumulator, so the read-modify-write is skipped for non-emitters
                // (NaN compares false and still takes the add).
                const float er = wr * s1.z, eg = wg * s1.w, eb = wb * s2.x;
                // (NEE: the emitters are sampled explicitly, so running into one counts for camera rays only)
                if ((!NEE || depth == 0u) && !(er == 0.f && eg == 0.f && eb == 0.f)) add_radiance(rc, rad, slot, er, eg, eb);
                depth++;
                terminated = depth >= rc.max_depth;  // raygen.rgen:62 loop bound
                // (NEE samples no light at the path's last hit: that sample stands for the emission the next ray would find,
                // and the reference's sum ends with the hit of ray max_depth - 1, raygen.rgen:62-83)
                if (!terminated) {
                    if (LDS_TABLES) {
                        a = tri4[3 * pos + 0]; b = tri4[3 * pos + 1]; c = tri4[3 * pos + 2];
                    } else {
                        s0.x = a.w; s0.y = b.w; s0.z = c.w;
                    }
                    // closesthit.rchit:56-57: position from barycentrics, (v0*b0 + v1*b1) + v2*b2
                    // the hit record carries (V, W, det) of the watertight test; attribs = (V/det, W/det)
                    float hu, hv;  // (0 <= V/det, W/det <= 1: ptm::div2_dominant's exact short division)
                    ptm::div2_dominant(h.y, h.z, h.w, hu, hv);
                    const float b0 = (1.0f - hu) - hv;
                    org = { (a.x * b0 + b.x * hu) + c.x * hv, (a.y * b0 + b.y * hu) + c.y * hv,
                            (a.z * b0 + b.z * hu) + c.z * hv };
                    ptm::f3 nrm = { s0.x, s0.y, s0.z };
                    ptm::f3 tng{};  // instanced scenes with the (instance, triangle) table: the tangent of createCoordinateSystem
                    if (INST) {
                        // instanced scene: position by the object->world matrix, normal by the inverse
                        // transpose, renormalised (the reference's closesthit has one identity instance)
                        const uint32_t ip = hit_inst[q];
                        const float4 m0 = inst6[6 * (size_t)ip + 0], m1 = inst6[6 * (size_t)ip + 1], m2 = inst6[6 * (size_t)ip + 2];
                        const ptm::f3 pw = { ((m0.x * org.x + m0.y * org.y) + m0.z * org.z) + m0.w,
                                             ((m1.x * org.x + m1.y * org.y) + m1.z * org.z) + m1.w,
                                             ((m2.x * org.x + m2.y * org.y) + m2.z * org.z) + m2.w };
                        org = pw;
                        if (inst_frame) {
                            // ... both evaluated once per (instance, triangle) with these very operations (lbvh_build.hip
                            // k_inst_frames): a 32-B gather instead of two square roots and five true divides per hit
                            const size_t e = 2 * ((size_t)ip * n_tris + pos);
                            const float4 f0 = inst_frame[e], f1 = inst_frame[e + 1];
                            nrm = { f0.x, f0.y, f0.z };
                            tng = { f0.w, f1.x, f1.y };
                        } else {
                            const float4 i0 = inst6[6 * (size_t)ip + 3], i1 = inst6[6 * (size_t)ip + 4], i2 = inst6[6 * (size_t)ip + 5];
                            const float nx = (i0.x * nrm.x + i1.x * nrm.y) + i2.x * nrm.z;
                            const float ny = (i0.y * nrm.x + i1.y * nrm.y) + i2.y * nrm.z;
                            const float nz = (i0.z * nrm.x + i1.z * nrm.y) + i2.z * nrm.z;
                            const float l = ptm::fsqrt((nx * nx + ny * ny) + nz * nz);
                            nrm = { ptm::fdiv(nx, l), ptm::fdiv(ny, l), ptm::fdiv(nz, l) };
                        }
                    }
                    if (NEE && n_lights) {  // one light sample -> shadow queue (three random numbers, drawn before the bounce's)
                        ptm::f3 wi;
                        float4 cb;
                        if (nee_sample(lights, n_lights, light_area, seed, org, nrm, s0.w, s1.x, s1.y, wr, wg, wb, wi, cb)) {
                            s_alive[it] = true;
                            s_rayA[it] = make_float4(org.x, org.y, org.z, wi.x);
                            s_rayB[it] = make_float2(wi.y, wi.z);
                            s_contrib[it] = cb;
                        }
                    }
                    const float r1 = ptm::rnd(seed);  // cos(theta) first, azimuth second
                    const float r2 = ptm::rnd(seed);
                    if (LDS_TABLES && !INST) {  // the triangle's tangent frame was evaluated once, by k_pack, with the same operations
                        const float4 f0 = frame4[2 * pos + 0], f1 = frame4[2 * pos + 1];
                        dir = ptm::sample_direction_frame(r1, r2, nrm, { f0.x, f0.y, f0.z }, { f0.w, f1.x, f1.y });
                    } else if (INST && inst_frame) {  // bitangent = the cross product of tangent_frame, same operands
                        const ptm::f3 btg = { nrm.y * tng.z - nrm.z * tng.y, nrm.z * tng.x - nrm.x * tng.z, nrm.x * tng.y - nrm.y * tng.x };
                        dir = ptm::sample_direction_frame(r1, r2, nrm, tng, btg);
                    } else {
                        dir = ptm::sample_direction(r1, r2, nrm);  // raygen.rgen:78
                    }
                    const float dt = (dir.x * nrm.x + dir.y * nrm.y) + dir.z * nrm.z;
                    // raygen.rgen:79-80: weight *= brdf * dot / pdf, pdf = 1/(2*pi) as a true divide
                    float fr = s0.w * dt, fg = s1.x * dt, fb = s1.y * dt;
                    ptm::div3_by_pdf(fr, fg, fb);
                    wr = wr * fr;
                    wg = wg * fg;
                    wb = wb * fb;
                }
            }
            if (terminated) {
                sample++;
                if (DENSE_REGEN) {
                    regen[it] = true;  // the next sample's primary ray is built after the item loop, by densely packed lanes
                } else {
                    uint32_t f, g, px, py;
                    slot_pixel(rc, tiles, slot, f, g, px, py);
                    if (sample < min(rc.spp, (g + 1u) * rc.group_size)) {  // next sample of this slot: raygen.rgen:45-60
                        seed = ptm::make_seed(px, py, sample, rc.frame_base + (int32_t)f, rc.spp);
                        ptm::primary_ray(rc.cam, px, py, seed, org, dir);
                        wr = wg = wb = 1.0f;
                        depth = 0;
                        alive[it] = true;
                    }
                }
            } else {
                alive[it] = true;
            }
            o_slot[it] = slot;
            o_ctr[it] = sample | (depth << 16);
            o_state[it] = make_float4(__uint_as_float(seed), wr, wg, wb);
            o_rayA[it] = make_float4(org.x, org.y, org.z, dir.x);
            o_rayB[it] = make_float2(dir.y, dir.z);
        }
        if (DENSE_REGEN) {
            // Regeneration (raygen.rgen:45-60 for the slot's next sample: pixel of the slot, seed, jitter, camera ray -- five
            // true divides and a square root) used to sit in the per-item branch above, which a wave enters whenever ANY of
            // its lanes ended a path, i.e. always, at ~30 % lane occupancy.  Here the ended paths of all SH_ITEMS items of a
            // wave are handed, through a wave-private piece of LDS, to consecutive lanes: one pass (two when more than 64
            // ended) at 60 % occupancy instead of SH_ITEMS passes at 30 %.  No block barrier: a wave's LDS operations
            // execute in order.  Same operations per path, same bits.
            const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
            const unsigned long long lt = (1ull << lane) - 1ull;
            float4 *cell = s_regen[wave];
            uint32_t rank[SH_ITEMS], total = 0;
#pragma unroll
            for (int it = 0; it < SH_ITEMS; it++) {
                const unsigned long long m = __ballot(regen[it]);
                rank[it] = total + (uint32_t)__popcll(m & lt);
                total += (uint32_t)__popcll(m);
                if (regen[it]) cell[rank[it]] = make_float4(__uint_as_float(o_slot[it]), __uint_as_float(o_ctr[it] & 0xFFFFu), 0.f, 0.f);
            }
            __builtin_amdgcn_wave_barrier();
            for (uint32_t j = (uint32_t)lane; j < total; j += 64u) {
                const float4 jc = cell[j];
                const uint2 job = make_uint2(__float_as_uint(jc.x), __float_as_uint(jc.y));
                uint32_t f, g, px, py;
                slot_pixel(rc, tiles, job.x, f, g, px, py);
                float4 r = make_float4(2.0f, 0.f, 0.f, 0.f);
                if (job.y < min(rc.spp, (g + 1u) * rc.group_size)) {
                    uint32_t seed = ptm::make_seed(px, py, job.y, rc.frame_base + (int32_t)f, rc.spp);
                    ptm::f3 org, dir;
                    ptm::primary_ray(rc.cam, px, py, seed, org, dir);
                    r = make_float4(dir.x, dir.y, dir.z, __uint_as_float(seed));
                }
                cell[j] = r;
            }
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int it = 0; it < SH_ITEMS; it++) {
                const float4 r = regen[it] ? cell[rank[it]] : make_float4(2.0f, 0.f, 0.f, 0.f);
                if (r.x != 2.0f) {
                    alive[it] = true;
                    o_ctr[it] = o_ctr[it] & 0xFFFFu;  // depth 0
                    o_state[it] = make_float4(r.w, 1.0f, 1.0f, 1.0f);  // raygen.rgen:59
                    o_rayA[it] = make_float4(rc.cam.ox, rc.cam.oy, rc.cam.oz, r.x);
                    o_rayB[it] = make_float2(r.y, r.z);
                }
            }
            __builtin_amdgcn_wave_barrier();  // the area is reused by the next chunk
        }
        uint32_t dst[SH_ITEMS];
        if (NEE) {  // the shadow queue, compacted like the path queue (its entries outlive this path's regeneration: own slot copy)
            uint32_t sdst[SH_ITEMS];
            chunk_offsets<SH_ITEMS>(s_alive, sdst, sq_count, s_wcnt, &s_base);
#pragma unroll
            for (int it = 0; it < SH_ITEMS; it++) {
                if (s_alive[it]) {
                    sq.rayA[sdst[it]] = s_rayA[it];
                    sq.rayB[sdst[it]] = s_rayB[it];
                    sq.contrib[sdst[it]] = s_contrib[it];
                    sq.tmax[sdst[it]] = s_contrib[it].w;
                    sq.slot[sdst[it]] = o_slot[it];
                }
            }
        }
        chunk_offsets<SH_ITEMS>(alive, dst, count_out, s_wcnt, &s_base);
#pragma unroll
        for (int it = 0; it < SH_ITEMS; it++) {
            if (alive[it]) {
                ptm::st_stream<true>(out.id + dst[it], make_uint2(o_slot[it], o_ctr[it]));
                ptm::st_stream<true>(out.state + dst[it], o_state[it]);
                ptm::st_stream<true>(out.rayA + dst[it], o_rayA[it]);
                ptm::st_stream<true>(out.rayB + dst[it], o_rayB[it]);
            }
        }
    }
}

// ---- resolve: raygen.rgen:86-90 for every frame of the batch, in frame order -------------------
__device__ __forceinline__ uint8_t to_unorm8(float c)
{
    if (!(c > 0.0f)) return 0;
    if (c > 1.0f) c = 1.0f;
    return (uint8_t)(c * 255.0f + 0.5f);
}

__global__ __launch_bounds__(TB) void k_resolve(RenderConst rc, const uint32_t *__restrict__ tiles, Radiance rad,
                                                float *__restrict__ film, uint8_t *__restrict__ bgra, const unsigned long long *__restrict__ skip_if_set)
{
    const uint32_t local = blockIdx.x * TB + threadIdx.x;
    if (local >= rc.slots_per_lane) return;
    if (skip_if_set && *skip_if_set != 0ull) return;  // (a term log overflowed in the launch before: the host renders these frames again)
    uint32_t f0, g0, px, py;
    slot_pixel(rc, tiles, local, f0, g0, px, py);
    if (px >= rc.width || py >= rc.height) return;
    const size_t pix = (size_t)py * rc.width + px;
    float fr = film[3 * pix + 0], fg = film[3 * pix + 1], fb = film[3 * pix + 2];
    uchar4 img = reinterpret_cast<uchar4 *>(bgra)[pix];  // bytes B,G,R,A
    const float spp = (float)rc.spp;
    for (uint32_t f = 0; f < rc.lanes_active; f++) {
        float4 c = make_float4(0.f, 0.f, 0.f, 0.f);
        if (rc.groups == 1u) {
            c = rad.color[(size_t)f * rc.slots_per_lane + local];
        }
        // (head + tail with the cull: the head slot of a pixel outside the rectangle holds all its samples, its tail slots were never written)
        const bool no_tails = rc.tail && rc.cull_on && ((int32_t)px < rc.cull[0] || (int32_t)px > rc.cull[2] || (int32_t)py < rc.cull[1] || (int32_t)py > rc.cull[3]);
        if ((rc.groups > 1u || rc.tail) && !no_tails) {  // replay the groups' (or, behind a head slot's accumulator, the one-sample tail slots') term logs in sample order: the reference's sequence of adds
            const uint32_t n_logs = rc.tail ? rc.tail : rc.groups;
            const size_t log_slots = rc.tail ? rc.n_tail : rc.n_slots;
            for (uint32_t g = 0; g < n_logs; g++) {
                const size_t slot = ((size_t)f * n_logs + g) * rc.slots_per_lane + local;
                const uint32_t nt_all = rad.nterm[slot], nt = min(nt_all, rc.term_cap);
                const float4 *to = rad.terms_over + slot * (rc.term_cap - rc.term_pcap);
                for (uint32_t k = 0; k < nt; k++) {
                    const float4 e = k < rc.term_pcap ? rad.terms[(size_t)k * log_slots + slot] : to[k - rc.term_pcap];
                    c.x = c.x + e.x;
                    c.y = c.y + e.y;
                    c.z = c.z + e.z;
                }
                // the few slots with more terms: their pool entries are chained newest-first, so the j-th one in
                // path order is reached by walking m-1-j links (m is small; quadratic in m, rare).  (When the pool
                // itself overflowed the chain is incomplete: the host discards this batch.)
                const uint32_t m = nt_all - nt;
                for (uint32_t j = 0; j < m; j++) {
                    uint32_t idx = rad.spill_head[slot];
                    for (uint32_t w = j + 1; w < m && idx != SPILL_NONE; w++) idx = __float_as_uint(rad.spill[idx].w);
                    if (idx == SPILL_NONE) break;
                    const float4 e = rad.spill[idx];
                    c.x = c.x + e.x;
                    c.y = c.y + e.y;
                    c.z = c.z + e.z;
                }
            }
        }
        const float cr = ptm::fdiv(c.x, spp), cg = ptm::fdiv(c.y, spp), cb = ptm::fdiv(c.z, spp);  // :86
        const int32_t frame = rc.frame_base + (int32_t)f;
        const float ff = (float)frame, f1 = (float)(frame + 1);
        const bool first = frame == 0;  // old * 0: never read the uninitialised image
        // float film (canonical): new = (color + old*frame) / (frame+1)
        fr = ptm::fdiv(cr + (first ? 0.f : fr) * ff, f1);
        fg = ptm::fdiv(cg + (first ? 0.f : fg) * ff, f1);
        fb = ptm::fdiv(cb + (first ? 0.f : fb) * ff, f1);
        // reference display image: rgba8 load -> blend -> clamp + quantise on store
        const float orr = first ? 0.f : ptm::fdiv((float)img.z, 255.0f);
        const float og = first ? 0.f : ptm::fdiv((float)img.y, 255.0f);
        const float ob = first ? 0.f : ptm::fdiv((float)img.x, 255.0f);
        const float oa = first ? 0.f : ptm::fdiv((float)img.w, 255.0f);
        img.z = to_unorm8(ptm::fdiv(cr + orr * ff, f1));
        img.y = to_unorm8(ptm::fdiv(cg + og * ff, f1));
        img.x = to_unorm8(ptm::fdiv(cb + ob * ff, f1));
        img.w = to_unorm8(ptm::fdiv(1.0f + oa * ff, f1));
    }
    film[3 * pix + 0] = fr;
    film[3 * pix + 1] = fg;
    film[3 * pix + 2] = fb;
    reinterpret_cast<uchar4 *>(bgra)[pix] = img;
}

// hit records of the internal layout (sorted position) -> API layout (gl_PrimitiveID)
__global__ __launch_bounds__(TB) void k_hits_to_api(const float4 *__restrict__ hit, const float4 *__restrict__ tri4,
                                                    const uint32_t *__restrict__ hit_inst,
                                                    const uint32_t *__restrict__ inst_id, uint32_t n,
                                                    pt_hit *__restrict__ out)
{
    const uint32_t i = blockIdx.x * TB + threadIdx.x;
    if (i >= n) return;
    const float4 h = hit[i];
    const uint32_t pos = __float_as_uint(h.x);
    pt_hit o;
    o.prim = pos == PT_MISS ? PT_MISS : __float_as_uint(tri4[3 * (size_t)pos].w);
    o.t = h.y; o.u = h.z; o.v = h.w;
    o.inst = pos == PT_MISS ? PT_MISS : (hit_inst ? inst_id[hit_inst[i]] : 0u);
    out[i] = o;
}

}  // namespace

// ---- launchers (the scheduler lives in render.hip) ---------------------------------------------------------------------
void ptw_launch_generate(const ptw::RenderConst &rc, const uint32_t *tiles, uint32_t slot_base, uint32_t n_slots, const ptw::Radiance &rad,
                         const ptw::QueueView &out, uint32_t *count_out, unsigned long long *stats, int num_cus, hipStream_t st)
{
    const int grid = (int)std::min<uint32_t>((n_slots + 4 * TB - 1) / (4 * TB), (uint32_t)num_cus * 16u);
    k_generate<<<grid, TB, 0, st>>>(rc, tiles, slot_base, n_slots, rad, out, count_out, stats);
}

void ptw_launch_shade(const ShadeLaunch &a, hipStream_t st, hipEvent_t ev0, hipEvent_t ev1)
{
    const pt_scene *s = a.scene;
#define PT_LAUNCH_SHADE(L, E, I)                                                                                                        \
    hipExtLaunchKernelGGL((k_shade<PT_SHADE_ITEMS, L, E, I>), dim3(a.grid), dim3(TB), (uint32_t)((L) ? a.smem : 0), st, ev0, ev1, 0u, a.rc,  \
                          a.tiles, s->d_tri4, s->d_shade4, s->n_tris, a.hit, a.rad, a.in, a.out, a.count_in, a.count_out,                 \
                          s->n_inst ? s->d_inst6 : nullptr, a.hit_inst, a.bvh8 ? s->d_shade64_8 : s->d_shade64,                            \
                          a.bvh8 ? s->d_ke4_8 : s->d_ke4, a.lights, a.n_lights, a.light_area, a.sq, a.sq_count, s->d_frame4, a.inst_frame)
    const int sel = (a.lds_tables ? 4 : 0) | (a.nee ? 2 : 0) | (s->n_inst ? 1 : 0);
    switch (sel) {
    case 0: PT_LAUNCH_SHADE(false, false, false); break;
    case 1: PT_LAUNCH_SHADE(false, false, true); break;
    case 2: PT_LAUNCH_SHADE(false, true, false); break;
    case 3: PT_LAUNCH_SHADE(false, true, true); break;
    case 4: PT_LAUNCH_SHADE(true, false, false); break;
    case 5: PT_LAUNCH_SHADE(true, false, true); break;
    case 6: PT_LAUNCH_SHADE(true, true, false); break;
    default: PT_LAUNCH_SHADE(true, true, true); break;
    }
#undef PT_LAUNCH_SHADE
}

void ptw_launch_shadow_add(const ptw::RenderConst &rc, const ptw::Radiance &rad, const float4 *sq_hit, const float4 *contrib,
                           const uint32_t *slot, const uint32_t *count, int grid, hipStream_t st)
{
    k_shadow_add<<<grid, TB, 0, st>>>(rc, rad, sq_hit, contrib, slot, count);
}

void ptw_launch_resolve(const ptw::RenderConst &rc, const uint32_t *tiles, const ptw::Radiance &rad, float *film, uint8_t *bgra, hipStream_t st,
                        const unsigned long long *skip_if_set)
{
    k_resolve<<<(rc.slots_per_lane + TB - 1) / TB, TB, 0, st>>>(rc, tiles, rad, film, bgra, skip_if_set);
}

void ptw_launch_hits_to_api(const float4 *hit, const float4 *tri4, const uint32_t *hit_inst, const uint32_t *inst_id, uint32_t n, pt_hit *out,
                            hipStream_t st)
{
    k_hits_to_api<<<(n + TB - 1) / TB, TB, 0, st>>>(hit, tri4, hit_inst, inst_id, n, out);
}
