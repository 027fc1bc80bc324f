// fused_inst_kernel.h -- PT_PIPELINE_FUSED for two-level scenes (BASELINE config C4): k_fused's radiance loop around the
// two-level walk of k_extend_inst16.
//
// What fused_kernel.h says about k_fused holds here: a lane keeps its path from bounce to bounce, its shading state in LDS,
// and runs the shade block -- closesthit.rchit:50-65 / miss.rmiss:8-12, raygen.rgen:76-83, the next sample's camera ray
// (raygen.rgen:45-60), or the first sample of a new slot -- once enough of the wave's lanes wait with a finished ray; slots
// come from the same eight counters.  What differs is the walk between two shade blocks:
//   * traversal: k_extend_inst16<false, PAIRS> (extend_inst16.h) restated operation for operation -- 64-B fp16 nodes on
//     both levels (TLAS from L2 with its top levels in LDS, BLAS in LDS), one-dword stack entries with the spill area
//     behind them, instance entry / exit with the waiting rules `enter_min` / `leaf_min` / `node_yield`;
//   * shading of a hit: k_shade<INST> -- position to world space by the instance's matrix (48 B from L2 per hit), normal and
//     tangent from the (instance, triangle) table of k_inst_frames (32 B), or by the inverse transpose where that table was
//     not built; the same operations in the same order, so the film is the wavefront pipeline's bit for bit.
// The hit record of the wavefront pipeline {pos, V, W, det} + hit_inst stays in registers (best_pos .. best_ipos).
#pragma once

#ifndef PT_FUSEDI_WAVES
#define PT_FUSEDI_WAVES 4
#endif
#ifndef PT_FUSEDI_TB
#define PT_FUSEDI_TB 512
#endif
#ifndef PT_FUSEDI_TLAS_KB
#define PT_FUSEDI_TLAS_KB 8  // TLAS nodes staged in LDS per workgroup (pt_tuning.tlas_lds_kb overrides)
#endif
constexpr int FITB = PT_FUSEDI_TB;

// PAIRS: every BLAS leaf is one triangle or one fan pair (tested with shared vertex work); else leaves of up to four triangles
template <bool GROUPED, bool PAIRS>
__global__ __launch_bounds__(FITB, PT_FUSEDI_WAVES) void k_fused_inst(
    RenderConst rc_arg, const uint32_t *__restrict__ tiles_arg, Radiance rad_arg, const uint4 *__restrict__ tlas16_arg, NormBox nbt_arg,
    const uint4 *__restrict__ g_blas16, NormBox nbb_arg, const float4 *__restrict__ g_tri4, const float4 *__restrict__ g_shade4,
    uint32_t n_blas_wide, uint32_t n_tris, const float4 *__restrict__ inst6_arg, const uint32_t *__restrict__ inst_id_arg,
    const float4 *__restrict__ inst_frame_arg, uint32_t slot_base_arg, uint32_t n_slots_arg, uint32_t *next_slot_arg, unsigned long long *stats_arg,
    uint32_t *__restrict__ spill_arg, uint32_t spill_stride_arg, int refill_arg, float tmin_arg, float tmax_arg, int lds_stack, int enter_min_arg,
    int leaf_min_arg, int node_yield_arg, uint32_t n_tlas_lds_arg)
{
    // (everything the persistent loop reads: a scalar register of its own -- ptm::own_sgprs)
    const RenderConst rc = ptm::own_sgprs(rc_arg);
    const Radiance rad = ptm::own_sgprs(rad_arg);
    const NormBox nbt = ptm::own_sgprs(nbt_arg), nbb = ptm::own_sgprs(nbb_arg);
    const uint32_t *tiles = ptm::own_sgprs(static_cast<const uint32_t *>(tiles_arg));
    const uint4 *tlas16 = ptm::own_sgprs(static_cast<const uint4 *>(tlas16_arg));
    const float4 *inst6 = ptm::own_sgprs(static_cast<const float4 *>(inst6_arg)), *inst_frame = ptm::own_sgprs(static_cast<const float4 *>(inst_frame_arg));
    const uint32_t *inst_id = ptm::own_sgprs(static_cast<const uint32_t *>(inst_id_arg));
    uint32_t *next_slot = ptm::own_sgprs(next_slot_arg), *spill = ptm::own_sgprs(static_cast<uint32_t *>(spill_arg));
    unsigned long long *stats = ptm::own_sgprs(stats_arg);
    const uint32_t slot_base = ptm::own_sgprs(slot_base_arg), n_slots = ptm::own_sgprs(n_slots_arg), spill_stride = ptm::own_sgprs(spill_stride_arg),
                   n_tlas_lds = ptm::own_sgprs(n_tlas_lds_arg);
    const int refill = ptm::own_sgprs(refill_arg), enter_min = ptm::own_sgprs(enter_min_arg), leaf_min = ptm::own_sgprs(leaf_min_arg),
              node_yield = ptm::own_sgprs(node_yield_arg);
    const float tmin = ptm::own_sgprs(tmin_arg), tmax = ptm::own_sgprs(tmax_arg);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // ---- LDS: stack | BLAS nodes + the TLAS's top levels | three permuted triangle copies | shade4 | path state | tile words
    uint32_t *s_stack = reinterpret_cast<uint32_t *>(smem);  // [lds_stack][FITB]
    uint32_t *s_blas = s_stack + (size_t)lds_stack * FITB;    // [n_blas_wide + n_tlas_lds][I16_NODE_DW]
    float4 *s_tri = reinterpret_cast<float4 *>(s_blas + (size_t)I16_NODE_DW * (n_blas_wide + n_tlas_lds));
    float4 *s_shade = s_tri + 9 * (size_t)n_tris;
    lds_u32 *my_state = (lds_u32 *)reinterpret_cast<uint32_t *>(s_shade + 3 * (size_t)n_tris) + threadIdx.x;
    for (uint32_t i = threadIdx.x; i < 4 * n_tlas_lds; i += FITB)
        *reinterpret_cast<uint4 *>(s_blas + (size_t)(n_blas_wide + (i >> 2)) * I16_NODE_DW + 4 * (i & 3u)) = tlas16[i];
    for (uint32_t i = threadIdx.x; i < 4 * n_blas_wide; i += FITB) {
        uint4 v = g_blas16[i];
        if ((i & 3u) == 3u) {  // the four child words -> 16-bit codes (extend_inst16.h)
            auto code = [](uint32_t w) {
                if (w == SENTINEL) return I16_DONE;
                return (w & PT_LEAF) ? (I16_LEAF | (((w >> 28) & 3u) << 11) | (w & 0x7FFu)) : (w & 0x7FFFu);
            };
            v = make_uint4(code(v.x), code(v.y), code(v.z), code(v.w));
        }
        *reinterpret_cast<uint4 *>(s_blas + (size_t)(i >> 2) * I16_NODE_DW + 4 * (i & 3u)) = v;
    }
    for (uint32_t i = threadIdx.x; i < 3 * n_tris; i += FITB) {
        const float4 v = g_tri4[i];
        s_tri[i] = make_float4(v.y, v.z, v.x, v.w);
        s_tri[3 * n_tris + i] = make_float4(v.z, v.x, v.y, v.w);
        s_tri[6 * n_tris + i] = v;  // kz = 2: (0,1,2) -- also what the shade block reads
        s_shade[i] = g_shade4[i];
    }
    __syncthreads();
    const float4 *verts = s_tri + 6 * (size_t)n_tris;

    lds_u32 *my_stack = (lds_u32 *)s_stack + threadIdx.x;
    uint32_t *my_spill = spill + (size_t)blockIdx.x * FITB + threadIdx.x;
    const float INF = __builtin_inff();
    const int lane = threadIdx.x & 63;

    // (a lane traces a ray <=> cur != I16_DONE: no flag is kept, fused_kernel.h)
    // (... and owns a live path <=> sp >= 0: a lane without one holds sp = -1, fused_kernel.h)
    bool out_of_slots = false, in_blas = false;
    uint32_t n_rays_wave = 0, n_cull_wave = 0;
    uint32_t w_next = 0, w_end = 0, w_base = 0;
    uint32_t w_part = blockIdx.x % (uint32_t)PT_FUSED_PARTS, w_tried = 0;
    const uint32_t part_len = ((n_slots + PT_FUSED_PARTS - 1) / PT_FUSED_PARTS + 63u) & ~63u;
    lds_u32 *s_wtile = (lds_u32 *)reinterpret_cast<uint32_t *>(s_shade + 3 * (size_t)n_tris) + FS_FIELDS * FITB + (threadIdx.x >> 6) * PT_FUSED_WTILES;
    ptm::f3 org_w{}, dir_w{};
    ptm::f3 inv{}, invf{}, on{}, of{}, orgp{};  // the level being walked, in that level's normalised coordinates
    uint32_t mx = 0, my = 0, mz = 0;
    ptm::f3 w_inv{}, w_invf{}, w_on{}, w_of{};  // the TLAS level's constants, kept across instance visits
    uint32_t w_mx = 0, w_my = 0, w_mz = 0;
    uint32_t tri_base = 0;
    ptm::RayPre pre{};
    float best_t = tmax, best_V = 0.f, best_W = 0.f, best_det = 1.f;
    uint32_t best_pos = PT_MISS, best_prim = PT_MISS, best_ipos = PT_MISS, best_iid = PT_MISS;
    uint32_t cur = I16_DONE, cur_ipos = 0, cur_iid = 0;
    int sp = -1, sp_exit = 0;

    auto level_setup = [&](const ptm::f3 o, const ptm::f3 d, const NormBox &nb) {
        const ptm::f3 on_ = { (o.x - nb.cx) * nb.rsx, (o.y - nb.cy) * nb.rsy, (o.z - nb.cz) * nb.rsz };
        inv = { ptm::safe_inv(d.x) * nb.sx, ptm::safe_inv(d.y) * nb.sy, ptm::safe_inv(d.z) * nb.sz };
        slab_setup(on_, inv, invf, on, of);
        mx = inv.x < 0.f ? 0xFFFFFFFFu : 0u; my = inv.y < 0.f ? 0xFFFFFFFFu : 0u; mz = inv.z < 0.f ? 0xFFFFFFFFu : 0u;
    };
    auto push = [&](uint32_t e) {
        if (sp < lds_stack) my_stack[sp * FITB] = e;
        else my_spill[(size_t)(sp - lds_stack) * spill_stride] = e;
        sp++;
    };
    auto pop = [&]() -> uint32_t {  // (a loop on the WAVE's condition, fused_kernel.h)
        constexpr uint32_t PENDING = 0xFFFFFFFFu;
        uint32_t r = PENDING;
        while (__ballot(r == PENDING)) {
            if (r == PENDING) {
                if (sp > 0) {
                    if (in_blas && sp == sp_exit) in_blas = false;  // the instance is done: back to the world-space ray and the TLAS
                    sp--;
                    uint32_t e;
                    if (sp < lds_stack) e = my_stack[sp * FITB];
                    else e = my_spill[(size_t)(sp - lds_stack) * spill_stride];
                    r = __uint_as_float(e & 0xFFFF0000u) <= best_t ? (e & 0xFFFFu) : PENDING;  // (a select, not a branch: fused_kernel.h)
                } else {
                    in_blas = false;
                    r = I16_DONE;
                }
            }
        }
        return r;
    };
    auto pop_and_restore = [&]() -> uint32_t {
        const bool was_in = in_blas;
        const uint32_t c = pop();
        if (was_in && !in_blas) {
            inv = w_inv; invf = w_invf; on = w_on; of = w_of;
            mx = w_mx; my = w_my; mz = w_mz;
        }
        return c;
    };

    for (;;) {
        // ---- shade block (fused_kernel.h, with k_shade<INST>'s hit shading)
        const bool have = cur != I16_DONE, path = sp >= 0;
        const unsigned long long m_have = __ballot(have), m_path = __ballot(path);
        const unsigned long long m_in_blk = ~m_have & (out_of_slots ? m_path : ~0ull);  // (waves are whole: FITB is a multiple of 64)
        const bool in_blk = !have && (path || !out_of_slots);
        const int n_work = __popcll(m_in_blk);
        if (n_work && n_work * 64 >= refill * (n_work + __popcll(m_have))) {  // (refill <= 64: a wave without a tracing lane always passes)
            uint32_t slot = 0, ctr = 0, seed = 0, pxy = 0;
            float wr = 0.f, wg = 0.f, wb = 0.f;
            ptm::f3 org{}, dir{};
            bool got_ray = false, need_primary = false, bounce = false;
            // (1) the hit of the ray that just ended
            if (in_blk && path) {
                slot = my_state[FS_SLOT * FITB]; ctr = my_state[FS_CTR * FITB]; seed = my_state[FS_SEED * FITB];
                wr = __uint_as_float(my_state[FS_WR * FITB]); wg = __uint_as_float(my_state[FS_WG * FITB]); wb = __uint_as_float(my_state[FS_WB * FITB]);
                pxy = my_state[FS_PXY * FITB];
                uint32_t sample = ctr & 0xFFFFu, depth = ctr >> 16;
                float er, eg, eb;
                bool terminated, add;
                const uint32_t pos = best_pos;
                if (pos == PT_MISS) {  // miss.rmiss:10-11 then raygen.rgen:76, 81-83
                    er = wr * rc.env[0]; eg = wg * rc.env[1]; eb = wb * rc.env[2];
                    add = true;
                    terminated = true;
                } else {
                    const float4 s1 = s_shade[3 * pos + 1], s2 = s_shade[3 * pos + 2];
                    er = wr * s1.z; eg = wg * s1.w; eb = wb * s2.x;
                    add = !(er == 0.f && eg == 0.f && eb == 0.f);
                    depth++;
                    terminated = depth >= rc.max_depth;  // raygen.rgen:62
                }
                if (add) {
                    if (!GROUPED) {
                        my_state[FS_A * FITB] = __float_as_uint(__uint_as_float(my_state[FS_A * FITB]) + er);
                        my_state[FS_B * FITB] = __float_as_uint(__uint_as_float(my_state[FS_B * FITB]) + eg);
                        my_state[FS_C * FITB] = __float_as_uint(__uint_as_float(my_state[FS_C * FITB]) + eb);
                    } else {  // the ordered term log of add_radiance (wavefront_types.h), the count kept in LDS
                        const uint32_t k = my_state[FS_A * FITB];
                        if (k < rc.term_pcap) ptm::st_stream<true>(rad.terms + ((size_t)k * rc.n_slots + slot), make_float4(er, eg, eb, 0.f));
                        else if (k < rc.term_cap) ptm::st_stream<true>(rad.dev->terms_over + ((size_t)slot * (rc.term_cap - rc.term_pcap) + (k - rc.term_pcap)), make_float4(er, eg, eb, 0.f));  // (rad.dev: wavefront_types.h)
                        else {
                            const Radiance rr = *rad.dev;
                            const unsigned long long idx = atomicAdd(rr.spill_count, 1ull);
                            if (idx < rr.spill_cap) {
                                // (the slot's first pool entry ends its chain: no per-slot initialisation of the heads)
                                rr.spill[idx] = make_float4(er, eg, eb, __uint_as_float(k == rc.term_cap ? SPILL_NONE : rr.spill_head[slot]));
                                rr.spill_head[slot] = (uint32_t)idx;
                            } else {
                                *rr.overflow = 1ull;
                            }
                        }
                        my_state[FS_A * FITB] = k + 1u;
                    }
                }
                if (!terminated) {
                    bounce = true;  // (the bounce itself: step (3), beside the camera rays -- fused_kernel.h explains)
                } else {
                    sample++;
                    depth = 0;
                    bool more;
                    if (!GROUPED) {
                        more = sample < rc.spp;  // (one group: the slot is the pixel's whole frame)
                    } else {
                        const uint32_t lane_slot = rc.div_spl.div(slot);
                        const uint32_t g = lane_slot - rc.div_groups.div(lane_slot) * rc.groups;
                        more = sample < min(rc.spp, (g + 1u) * rc.group_size);
                    }
                    if (more) {
                        need_primary = true;  // the slot's next sample: raygen.rgen:45-60
                    } else {  // the slot is complete
                        if (!GROUPED) rad.color[slot] = make_float4(__uint_as_float(my_state[FS_A * FITB]), __uint_as_float(my_state[FS_B * FITB]),
                                                                   __uint_as_float(my_state[FS_C * FITB]), 0.f);
                        else rad.nterm[slot] = my_state[FS_A * FITB];
                        sp = -1;  // (no path)
                    }
                }
                ctr = sample | (depth << 16);
            }
            // (2) new slots for the lanes without a path: the batches, counters and tile words of k_fused (fused_kernel.h explains)
            const unsigned long long m_want = __ballot(in_blk && sp < 0);
            if (m_want && !out_of_slots) {
                if (w_next >= w_end) {
                    for (;;) {
                        const uint32_t part_begin = w_part * part_len, part_end = min(part_begin + part_len, n_slots);
                        uint32_t rel = 0, size = 0;
                        if (lane == 0) {
                            uint32_t *cnt = next_slot + w_part * (uint32_t)PT_FUSED_PART_STRIDE;
                            size = (uint32_t)(GROUPED ? PT_FUSED_BATCH : PT_FUSED_BATCH1);
                            rel = atomicAdd(cnt, size);
                        }
                        rel = __builtin_amdgcn_readfirstlane(rel);
                        size = __builtin_amdgcn_readfirstlane(size);
                        if (part_begin < part_end && rel < part_end - part_begin) {
                            w_base = w_next = part_begin + rel;
                            w_end = min(w_next + size, part_end);
                            break;
                        }
                        w_part = (w_part + 1u) % (uint32_t)PT_FUSED_PARTS;
                        if (++w_tried >= (uint32_t)PT_FUSED_PARTS) { out_of_slots = true; w_end = w_next; break; }
                    }
                    if (!out_of_slots && (uint32_t)lane < (w_end - w_base + 63u) / 64u) {
                        const uint32_t c = slot_base + w_base + 64u * (uint32_t)lane;
                        s_wtile[lane] = tiles[(c - rc.div_spl.div(c) * rc.slots_per_lane) >> 6];
                    }
                    __builtin_amdgcn_wave_barrier();
                }
                const uint32_t take = min((uint32_t)__popcll(m_want), w_end - w_next);
                const uint32_t rank = (uint32_t)__popcll(m_want & ((1ull << lane) - 1ull));
                uint32_t cull_n = 0u;  // samples of a slot that is finished here: its pixel cannot see the scene (fused_cull.h)
                if (in_blk && sp < 0 && rank < take) {
                    const uint32_t mine = w_next + rank;
                    slot = slot_base + mine;
                    const uint32_t lane_slot = rc.div_spl.div(slot);
                    const uint32_t f = rc.div_groups.div(lane_slot), g = lane_slot - f * rc.groups;
                    const uint32_t local = slot - lane_slot * rc.slots_per_lane;
                    const uint32_t tw = s_wtile[(mine - w_base) >> 6];
                    const uint32_t px = (tw & 0xFFFFu) * 8u + (local & 7u), py = (tw >> 16) * 8u + ((local >> 3) & 7u);
                    const uint32_t sample0 = g * rc.group_size;
                    const bool in_image = px < rc.width && py < rc.height && f < rc.lanes_active && sample0 < rc.spp;
                    if (in_image && ptc::pixel_culled(rc, px, py)) {
                        cull_n = GROUPED ? ptc::finish_group(rc, rad, slot, g) : ptc::finish_plain(rc, rad, slot);
                    } else if (in_image) {
                        pxy = px | (py << 16);
                        ctr = sample0;
                        my_state[FS_A * FITB] = 0u;
                        if (!GROUPED) { my_state[FS_B * FITB] = 0u; my_state[FS_C * FITB] = 0u; }
                        my_state[FS_MB * FITB] = (uint32_t)((int32_t)rc.spp * (rc.frame_base + (int32_t)f)) + 1u;
                        sp = 0;  // (a path)
                        need_primary = true;
                    } else if (GROUPED) {
                        rad.nterm[slot] = 0u;
                    }
                }
                w_next += take;
                if (rc.cull_on) {
                    const uint32_t n_c = ptc::rays_finished<GROUPED>(rc, cull_n);
                    n_rays_wave += n_c;
                    n_cull_wave += n_c;
                }
            }
            // (3) the new ray: a bounce or the camera ray of a slot's next / first sample; their two rand and their square root -- sqrt(1 - r1^2) of the
            // hemisphere sample, the length of the camera ray's direction -- run once for both kinds of lanes (fused_kernel.h has the measurement)
            if (need_primary) {
                const uint32_t m = (ctr & 0xFFFFu) + my_state[FS_MB * FITB];  // = sample + maxSamples * frame + 1 (ptm::make_seed)
                const uint2 sd = ptm::pcg2d(make_uint2((pxy & 0xFFFFu) * m, (pxy >> 16) * m));
                seed = sd.x + sd.y;
                wr = wg = wb = 1.0f;  // raygen.rgen:59
            }
            if (bounce || need_primary) {
                const float r1 = ptm::rnd(seed);  // bounce: cos(theta) first, azimuth second; camera ray: x jitter first, then y
                const float r2 = ptm::rnd(seed);
                float vx = 0.f, vy = 0.f, vz = 0.f, sq_arg;
                if (need_primary) {
                    ptm::primary_target(rc.cam, pxy & 0xFFFFu, pxy >> 16, r1, r2, vx, vy, vz);
                    sq_arg = (vx * vx + vy * vy) + vz * vz;
                } else {
                    sq_arg = 1.0f - r1 * r1;
                }
                const float sq = ptm::fsqrt(sq_arg);
                // (one set of quotients by a common divisor for both kinds of lanes: direction = (target - origin) / length, barycentrics = (V, W) / det --
                // fused_kernel.h)
                float q1, q2, q3;
                ptm::div3_dominant(need_primary ? vx : best_V, need_primary ? vy : best_W, need_primary ? vz : best_W, need_primary ? sq : best_det, q1, q2, q3);
                if (need_primary) {
                    org = { rc.cam.ox, rc.cam.oy, rc.cam.oz };
                    dir = { q1, q2, q3 };
                } else {
                    // closesthit.rchit:56-57 position from the barycentrics, in object space; then k_shade<INST>: position by the
                    // object->world matrix, normal + tangent of the (instance, triangle) pair; raygen.rgen:77-80 the bounce
                    const uint32_t pos = best_pos;
                    const float4 s0 = s_shade[3 * pos + 0], s1 = s_shade[3 * pos + 1];
                    const float4 a = verts[3 * pos + 0], b = verts[3 * pos + 1], c = verts[3 * pos + 2];
                    const float hu = q1, hv = q2;
                    const float b0 = (1.0f - hu) - hv;
                    org = { (a.x * b0 + b.x * hu) + c.x * hv, (a.y * b0 + b.y * hu) + c.y * hv, (a.z * b0 + b.z * hu) + c.z * hv };
                    ptm::f3 nrm = { s0.x, s0.y, s0.z };
                    ptm::f3 tng{};
                    const uint32_t ip = best_ipos;
                    const float4 m0 = inst6[6 * (size_t)ip + 0], m1 = inst6[6 * (size_t)ip + 1], m2 = inst6[6 * (size_t)ip + 2];
                    const ptm::f3 pw = { ((m0.x * org.x + m0.y * org.y) + m0.z * org.z) + m0.w,
                                         ((m1.x * org.x + m1.y * org.y) + m1.z * org.z) + m1.w,
                                         ((m2.x * org.x + m2.y * org.y) + m2.z * org.z) + m2.w };
                    org = pw;
                    if (inst_frame) {
                        const size_t e = 2 * ((size_t)ip * n_tris + pos);
                        const float4 f0 = inst_frame[e], f1 = inst_frame[e + 1];
                        nrm = { f0.x, f0.y, f0.z };
                        tng = { f0.w, f1.x, f1.y };
                        // bitangent = the cross product of tangent_frame, same operands
                        const ptm::f3 btg = { nrm.y * tng.z - nrm.z * tng.y, nrm.z * tng.x - nrm.x * tng.z, nrm.x * tng.y - nrm.y * tng.x };
                        dir = ptm::sample_direction_frame_sq(r1, r2, sq, nrm, tng, btg);
                    } else {
                        const float4 i0 = inst6[6 * (size_t)ip + 3], i1 = inst6[6 * (size_t)ip + 4], i2 = inst6[6 * (size_t)ip + 5];
                        const float nx = (i0.x * nrm.x + i1.x * nrm.y) + i2.x * nrm.z;
                        const float ny = (i0.y * nrm.x + i1.y * nrm.y) + i2.y * nrm.z;
                        const float nz = (i0.z * nrm.x + i1.z * nrm.y) + i2.z * nrm.z;
                        const float l = ptm::fsqrt((nx * nx + ny * ny) + nz * nz);
                        nrm = { ptm::fdiv(nx, l), ptm::fdiv(ny, l), ptm::fdiv(nz, l) };
                        ptm::f3 T, B;
                        ptm::tangent_frame(nrm, T, B);  // raygen.rgen:14-21 (sample_direction's own first half)
                        dir = ptm::sample_direction_frame_sq(r1, r2, sq, nrm, T, B);  // raygen.rgen:78
                    }
                    const float dt = (dir.x * nrm.x + dir.y * nrm.y) + dir.z * nrm.z;
                    float fr = s0.w * dt, fg = s1.x * dt, fb = s1.y * dt;
                    ptm::div3_by_pdf(fr, fg, fb);
                    wr = wr * fr; wg = wg * fg; wb = wb * fb;
                }
                got_ray = true;
            }
            // (4) state back to LDS, ray set-up (the refill block of k_extend_inst16)
            if (got_ray) {
                my_state[FS_SLOT * FITB] = slot; my_state[FS_CTR * FITB] = ctr; my_state[FS_SEED * FITB] = seed;
                my_state[FS_WR * FITB] = __float_as_uint(wr); my_state[FS_WG * FITB] = __float_as_uint(wg); my_state[FS_WB * FITB] = __float_as_uint(wb);
                my_state[FS_PXY * FITB] = pxy;
                org_w = org;
                dir_w = dir;
                level_setup(org_w, dir_w, nbt);
                w_inv = inv; w_invf = invf; w_on = on; w_of = of; w_mx = mx; w_my = my; w_mz = mz;
                in_blas = false;
                best_t = tmax; best_V = 0.f; best_W = 0.f; best_det = 1.f;
                best_pos = PT_MISS; best_prim = PT_MISS; best_ipos = PT_MISS; best_iid = PT_MISS;
                cur = 0u;  // TLAS root
                sp = 0;
            }
            n_rays_wave += (uint32_t)__popcll(__ballot(got_ray));
        }
        const bool tracing = cur != I16_DONE;
        const unsigned long long m_tracing = __ballot(tracing);
        if (m_tracing == 0ull && __ballot(sp >= 0) == 0ull && out_of_slots) break;  // (no `continue`: fused_kernel.h, one way back to the loop's head)

        // ---- node phase, either level (k_extend_inst16)
        // (a loop on the WAVE's condition, fused_kernel.h; the first step is unconditional here: the leaf phase's waiting rules rely on every lane that
        // holds a node taking a step per pass)
        const int n_have = __popcll(m_tracing);
        bool do_node = cur < I16_LEAF;  // (an inner node: the codes of leaves and I16_DONE carry the leaf bit -- ONE compare, fused_kernel.h)
        if (__ballot(do_node) != 0ull) for (;;) {
            if (do_node) {
            uint4 q0, q1, q2, cw;
            if (in_blas || cur < n_tlas_lds) {
                typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
                typedef __attribute__((address_space(3))) const u32x4 lds_cu4;
                const uint32_t li = in_blas ? cur : cur + n_blas_wide;
                lds_cu4 *nd = (lds_cu4 *)reinterpret_cast<const u32x4 *>(s_blas + (size_t)li * I16_NODE_DW);
                const u32x4 r0 = nd[0], r1 = nd[1], r2 = nd[2], r3 = nd[3];
                q0 = make_uint4(r0.x, r0.y, r0.z, r0.w); q1 = make_uint4(r1.x, r1.y, r1.z, r1.w);
                q2 = make_uint4(r2.x, r2.y, r2.z, r2.w); cw = make_uint4(r3.x, r3.y, r3.z, r3.w);
                PT_REG_BARRIER16(q0, q1, q2, cw)
            } else {
                const uint4 *nd = tlas16 + 4 * (size_t)cur;
                q0 = nd[0]; q1 = nd[1]; q2 = nd[2]; cw = nd[3];
                PT_REG_BARRIER16(q0, q1, q2, cw)
            }
            auto sel = [](uint32_t m, uint32_t a, uint32_t b) { return (m & a) | (~m & b); };
            const uint2 hnx = { sel(mx, q1.z, q0.x), sel(mx, q1.w, q0.y) }, hfx = { sel(mx, q0.x, q1.z), sel(mx, q0.y, q1.w) };
            const uint2 hny = { sel(my, q2.x, q0.z), sel(my, q2.y, q0.w) }, hfy = { sel(my, q0.z, q2.x), sel(my, q0.w, q2.y) };
            const uint2 hnz = { sel(mz, q2.z, q1.x), sel(mz, q2.w, q1.y) }, hfz = { sel(mz, q1.x, q2.z), sel(mz, q1.y, q2.w) };
            float t0, t1, t2, t3;
            PT_SLAB4H(t0, x, 0)
            PT_SLAB4H(t1, x, 1)
            PT_SLAB4H(t2, y, 0)
            PT_SLAB4H(t3, y, 1)
            uint32_t k0 = (__float_as_uint(t0) & 0xFFFF0000u) | cw.x, k1 = (__float_as_uint(t1) & 0xFFFF0000u) | cw.y,
                     k2 = (__float_as_uint(t2) & 0xFFFF0000u) | cw.z, k3 = (__float_as_uint(t3) & 0xFFFF0000u) | cw.w;
#define PT_KSWAP(A, B) { const uint32_t lo_ = min(A, B), hi_ = max(A, B); A = lo_; B = hi_; }
            PT_KSWAP(k0, k1)
            PT_KSWAP(k2, k3)
            PT_KSWAP(k0, k2)
            PT_KSWAP(k1, k3)
            PT_KSWAP(k1, k2)
#undef PT_KSWAP
            constexpr uint32_t KINF = 0x7F800000u;
            if (k3 < KINF) push(k3);  // farthest first
            if (k2 < KINF) push(k2);
            if (k1 < KINF) push(k1);
            cur = k0 < KINF ? (k0 & 0xFFFFu) : pop_and_restore();
            }
            do_node = cur < I16_LEAF;
            const int n_cont = __popcll(__ballot(do_node));
            if (n_cont == 0 || (node_yield > 0 && n_cont * node_yield < n_have)) break;
        }
        // ---- leaf phase: a BLAS leaf (one triangle or one fan pair) or a TLAS leaf (enter the instance)
        const bool at_leaf = (uint32_t)(cur - I16_LEAF) < I16_DONE - I16_LEAF;  // (a leaf that is not I16_DONE, as one compare)
        const int n_enter = __popcll(__ballot(at_leaf && !in_blas));
        const int n_leaf = __popcll(__ballot(at_leaf && in_blas));
        const bool descending = __ballot(cur < I16_LEAF) != 0ull;
        const bool do_leaf = n_leaf >= leaf_min || !(descending || n_enter >= enter_min);
        const bool others = descending || (do_leaf && n_leaf > 0);
        const bool do_enter = n_enter >= enter_min || !others;
        {   // (no `if (tracing)` around it: at_leaf is false for a lane without a ray)
            if (at_leaf && in_blas && do_leaf) {
                const uint32_t first = cur & 0x7FFu, cnt = ((cur >> 11) & 3u) + 1u;
                auto accept = [&](float t, float V, float W, float det, uint32_t pos, uint32_t prim) {
                    ptl::closer_instanced(t, V, W, det, pos, prim, cur_ipos, cur_iid, best_t, best_V, best_W, best_det, best_pos, best_prim, best_ipos, best_iid);
                };
                if (PAIRS) {
                    ptl::pair_leaf_test<true>(s_tri, (size_t)tri_base + 3 * (size_t)first, cnt == 2u, first, pre, orgp, tmin, tmax, accept,  // (<true>: the triangle records are followed by the shade table in LDS)
                                        [] {});
                } else {
                    for (uint32_t k = 0; k < cnt; k++) {
                        const uint32_t pos = first + k;
                        const size_t ti = (size_t)tri_base + 3 * (size_t)pos;
                        const float4 a = s_tri[ti + 0], b = s_tri[ti + 1], c = s_tri[ti + 2];
                        float t, V, W, det;
                        if (ptm::tri_test_perm(pre, orgp, { a.x, a.y, a.z }, { b.x, b.y, b.z }, { c.x, c.y, c.z }, tmin, tmax, t, V, W, det, nullptr))
                            accept(t, V, W, det, pos, __float_as_uint(a.w));
                    }
                }
                cur = pop_and_restore();
            } else if (at_leaf && !in_blas && do_enter) {
                // TLAS leaf: one instance.  The ray goes to object space un-normalised (t is the same parameter)
                const uint32_t first = cur & 0x7FFFu;
                cur_ipos = first;
                cur_iid = inst_id[first];
                const float4 r0 = inst6[6 * (size_t)first + 3], r1 = inst6[6 * (size_t)first + 4], r2 = inst6[6 * (size_t)first + 5];
                const ptm::f3 oo = { ((r0.x * org_w.x + r0.y * org_w.y) + r0.z * org_w.z) + r0.w,
                                     ((r1.x * org_w.x + r1.y * org_w.y) + r1.z * org_w.z) + r1.w,
                                     ((r2.x * org_w.x + r2.y * org_w.y) + r2.z * org_w.z) + r2.w };
                const ptm::f3 od = { (r0.x * dir_w.x + r0.y * dir_w.y) + r0.z * dir_w.z,
                                     (r1.x * dir_w.x + r1.y * dir_w.y) + r1.z * dir_w.z,
                                     (r2.x * dir_w.x + r2.y * dir_w.y) + r2.z * dir_w.z };
                level_setup(oo, od, nbb);
                pre = ptm::ray_setup(oo, od);
                tri_base = (uint32_t)pre.kz * 3u * n_tris;
                orgp = { ptm::sel3(pre.kz, oo.y, oo.z, oo.x), ptm::sel3(pre.kz, oo.z, oo.x, oo.y), ptm::sel3(pre.kz, oo.x, oo.y, oo.z) };
                sp_exit = sp;
                in_blas = true;
                cur = 0u;  // BLAS root
            }
            // (cur == I16_DONE: the hit (best_pos, best_V, best_W, best_det, best_ipos) waits in registers for the shade block
        }
    }
    if (lane == 0 && n_rays_wave) atomicAdd(stats, (unsigned long long)n_rays_wave);
    if (lane == 0 && n_cull_wave) atomicAdd(stats + 19, (unsigned long long)n_cull_wave);  // (pt_stats.rays_culled)
}
