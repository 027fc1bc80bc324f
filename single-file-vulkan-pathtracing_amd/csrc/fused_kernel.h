// fused_kernel.h -- PT_PIPELINE_FUSED: the radiance loop as ONE persistent kernel for scenes that live in LDS.
//
// The reference's raygen shader is itself a megakernel: one invocation owns its path state from the first sample of its
// pixel to the last (raygen.rgen:41-91), and traceRayEXT / the closest-hit and miss shaders run inside it.  The wavefront
// pipeline (render.hip, shade_kernels.hip) splits that into k_extend / k_shade over queues in HBM -- 150 B of queue traffic per ray and a
// workspace of tens of GB -- because big scenes need the sorting, the compaction and the long launches.  A scene of a few
// dozen triangles needs none of it: nodes, triangles and the shading tables fit LDS, so here a lane keeps its path from
// bounce to bounce and HBM sees the radiance of a slot only (16 B per slot, or 16 B per logged term with sample groups).
//
//   * traversal: the compact walk of extend_kernel.h (`extend_body<true, false, false, PAIRS>`; pair leaves: one-dword stack
//     entries in LDS, key-sorted children, fan pairs tested together), restated here operation for operation -- the hot
//     instantiation of that template is register-allocated to the last VGPR and must not grow a second use;
//   * a lane whose ray is finished WAITS with its hit in registers until `refill` / 64 of the wave's live lanes wait too
//     (36 / 64: fused.hip), then all of them run the shade block -- closesthit.rchit:50-65 / miss.rmiss:8-12, the bounce of
//     raygen.rgen:76-83, the next sample's camera ray (raygen.rgen:45-60), or the first sample of a NEW slot -- and set up
//     their next ray; the same operations on the same operands as k_shade, per lane in the same order, so the film is the wavefront pipeline's bit
//     for bit.  The steps a bounce and a camera ray share (two rand, one square root) run once for both kinds of lanes (step (3) below);
//   * slots (frame, sample group, pixel) are handed out in order by device-scope counters -- eight, one per XCD's share of the
//     workgroups, with work stealing between them -- PT_FUSED_BATCH at a time per wave, so a wave that drew cheap border
//     pixels simply takes more of them: no tail beyond the last batch's own length;
//   * path state that only the shade block touches (slot, sample | depth, seed, weight, pixel, the slot's colour or its
//     term count) lives in LDS, [field][thread]: the traversal loop keeps the registers it has in k_extend_lds7p;
//   * control flow is written for the CU's ONE scalar unit (round 6: it was as busy as the vector units -- DESIGN.md section 6): the loops inside
//     the persistent loop (node steps, pops) run on the WAVE's condition, with the lanes that are served behind one exec mask; no `continue`
//     (one way back to the loop's head); selects instead of short-circuit branches where the wave runs both sides anyway; no loop-carried
//     flag where a register says the same (`cur != DONE`); every kernel argument in a scalar register of its own (ptm::own_sgprs).
//
// Radiance goes where the wavefront pipeline puts it (struct Radiance): one accumulator per slot written when the slot is
// complete (one sample group), or the ordered term logs that k_resolve replays (several groups) -- k_resolve is shared.
#pragma once

// Block shape: the scene tables (9.2 KB for the Cornell box) are per block, the stack and the path state (20 dwords) per
// thread, so bigger blocks leave more of the 160 KB to waves: 256 threads = 5 blocks = 5 waves per SIMD; 512 threads =
// 50.2 KB = 3 blocks = 6 waves per SIMD at 80 VGPRs (1 - 2 dwords spilled).  Same box, interleaved: 36.2 -> 38.1 Grays/s
// (profiles/r04d_ab_fused_tb.log); 4 waves: 31.4.  Round 6's kernel: 768 x 6 waves -1 %, 256 -7 %, 384 -14 % (profiles/r06q_fused_block_size.log).
#ifndef PT_FUSED_WAVES
#define PT_FUSED_WAVES 6
#endif

#ifndef PT_FUSED_TB
#define PT_FUSED_TB 512
#endif
constexpr int FTB = PT_FUSED_TB;

#ifndef PT_FUSED_BATCH
#define PT_FUSED_BATCH 256  // most slots a wave draws per atomic (a multiple of 64: 64 consecutive slots are one 8x8 tile)
#endif
// ... with one sample group: ONE tile.  A wave hands a batch to its lanes as they finish their slots, so the last slot of a batch STARTS
// (batch / 64 - 1) slot lengths after the batch was drawn -- with 256 and slots of 32 samples (1.5 - 2.6 ms each) interior slots drawn 6 ms
// before the counters ran dry were still being started after it, whatever the hand-out order (the wave timeline of
// scripts/probe_fused_timeline.py: profiles/r05b_fused_timeline.log -> r05c_fused_timeline64.log).  64 / 128 / 256, 1080p Cornell box, Grays/s:
// K = 16 41.2 / 41.4 / 40.8; K = 4 38.8 / 37.7 / 33.8; K = 2 36.0 / 32.6 / 27.3; a rank of world 8 at 16 frames 36.1 / 33.0 / 27.2
// (profiles/r05c_fused_batch1.log).  The guided self-scheduling the bigger batches needed at the end of a launch is gone with them.
#ifndef PT_FUSED_BATCH1
#define PT_FUSED_BATCH1 64
#endif
#define PT_FUSED_WTILES (PT_FUSED_BATCH / 64)  // tile words a wave keeps in LDS for its current batch
#define PT_FUSED_PARTS 8          // slot counters (one per XCD's share of the workgroups)
#define PT_FUSED_PART_STRIDE 32   // ... in dwords: one 128-B line each

// path state in LDS, [field][thread]
enum : int { FS_SLOT = 0, FS_CTR, FS_SEED, FS_WR, FS_WG, FS_WB, FS_PXY, FS_A, FS_B, FS_C, FS_MB, FS_FIELDS };
// FS_A..C: the slot's colour (one sample group) | FS_A: its term count (several groups)
// FS_MB: maxSamples * frame + 1 of the slot's frame (raygen.rgen:47: the seed's multiplier less the sample number), kept so that a
// sample's camera ray does not decode slot -> frame again (two divisions by run-time constants, at the ten lanes of that block)

// PAIRS: every leaf is one triangle or one fan pair (k_extend_lds7p's trees); else leaves of up to four triangles (k_extend_lds7's)
// MODE 0: one sample group (a slot is a pixel's whole frame; radiance added in LDS); 1: several groups (every slot logs its radiance terms);
// 2: HEAD + TAIL (wavefront_types.h RenderConst::tail) -- head slots like mode 0 for samples [0, head_samples), handed out first, then one-sample tail
// slots like mode 1: what a launch of few frames ends with is short work, and only the tail's terms go through the log
// The node loop of a pass ends once fewer than B / A of the wave's tracing lanes still descend (the others hold a leaf or are done).  1080p Cornell,
// 16 / 4 frames per call, ms: 1/2 85.7 / 22.4, 2/5 85.2 / 22.2, 1/3 84.4 / 22.0, 2/7 84.3 / 22.0, 1/4 84.0 / 21.9, 1/5 84.3 / 22.0, 1/6 (rounds 4 - 5)
// 84.8 / 22.1, 1/8 85.7 / 22.4, 1/12 86.9 / 22.7, never (1/64) 96.1 / 24.9 (profiles/r05zs_node_exit.log)
// Round 6, with the loops on the wave's condition (a pass of the node loop got cheaper than a pass of the outer loop): 1/2 64.4, 2/5 63.95, 1/3 63.85, 1/4 64.1,
// 1/5 64.5, 1/8 66.0 ms per 16 frames (profiles/r06z_ab_node_exit.log): 1/3
#ifndef PT_FUSED_NODE_EXIT
#define PT_FUSED_NODE_EXIT 3
#endif
#ifndef PT_FUSED_NODE_EXIT_B
#define PT_FUSED_NODE_EXIT_B 1
#endif
// COUNT (PT_FLAG_COUNT_VISITS on the fused pipeline; never timed): every block of the loop counts its wave executions and the lanes inside them
// (FusedBlock below) -- with the blocks' instruction counts in the shipped ISA (scripts/isa_regions.py) that is where the kernel's VALU
// instructions go and which block runs at how many of its 64 lanes.  The product instantiations (COUNT = false) carry none of it.
enum FusedBlock : int {
    FB_ITER = 0,   // one pass of the outer loop (its head: the ballots that decide what runs)
    FB_SHADE,      // the shade block ran (lanes: those inside it, with a hit or asking for a slot)
    FB_HIT,        // (1) state loads of the lanes with a hit record
    FB_MISS,       // ... miss.rmiss: weight * env
    FB_SURFACE,    // ... closesthit.rchit: the triangle's shading record, weight * Ke
    FB_ADD,        // ... color += (LDS accumulator or term log)
    FB_BOUNCE,     // ... position, tangent frame, direction, brdf * cos / pdf (raygen.rgen:77-80)
    FB_NEXT,       // ... path ended: next sample of the slot, or the slot is complete
    FB_DONE,       // ... the slot's radiance goes to memory
    FB_HANDOUT,    // (2) the wave-uniform slot hand-out ran (lanes: those asking for a slot)
    FB_DRAW,       // ... the wave drew a batch of slots (the atomic and the tile words)
    FB_TAKE,       // ... lanes that took a slot (decode slot -> frame, pixel)
    FB_CULLED,     // ... of them: the pixel cannot see the scene, the slot is finished here
    FB_PRIMARY,    // (3) camera ray of a sample (raygen.rgen:45-60)
    FB_SETUP,      // (4) state back to LDS, ray set-up for the walk
    FB_NODE,       // one BVH4 node step
    FB_POP,        // one iteration of the stack-pop loop (inside node steps and behind leaf steps)
    FB_LEAF,       // one leaf step (a triangle or a fan pair)
    FB_DIV,        // ... its divide block (a lane is inside a triangle's edges)
    FB_FINISH,     // a walk ended (cur == DONE)
    FB_TRACE,      // (no code of its own) once per pass with a walk: the lanes that trace
    FB_SPAWN,      // the steps a bounce and a camera ray share: two rand, one square root
    FB_PTARGET,    // camera ray: pixel + jitter -> target - origin, its squared length (raygen.rgen:51-56)
    FB_PDIR,       // camera ray: the normalisation's three quotients (raygen.rgen:57)
    FB_POPTOP,     // a node step whose four children all missed takes the stack's top entry, read with the node, from a register
    FB_N
};
template <int MODE, bool PAIRS, bool COUNT>
__device__ __forceinline__ void fused_body(RenderConst rc_arg, const uint32_t *__restrict__ tiles_arg, Radiance rad_arg,
                                           const float4 *__restrict__ g_wide, const float4 *__restrict__ g_tri4,
                                           const float4 *__restrict__ g_shade4, const float4 *__restrict__ g_frame4,
                                           uint32_t n_wide, uint32_t n_tris, uint32_t slot_base_arg, uint32_t n_slots_arg,
                                           uint32_t *next_slot_arg, unsigned long long *stats_arg, int refill_arg, float tmin_arg,
                                           float tmax_arg, int lds_stack, FastDiv div_frames_arg)
{
    constexpr uint32_t LEAF_BIT = 0x2000u, DONE = 0x3FFFu;
    constexpr bool GROUPED = MODE == 1, HYB = MODE == 2;
    // (everything the persistent loop reads: a scalar register of its own -- own_sgprs)
    const RenderConst rc = ptm::own_sgprs(rc_arg);
    const Radiance rad = ptm::own_sgprs(rad_arg);
    const FastDiv div_frames = ptm::own_sgprs(div_frames_arg);
    const uint32_t *tiles = ptm::own_sgprs(static_cast<const uint32_t *>(tiles_arg));
    uint32_t *next_slot = ptm::own_sgprs(next_slot_arg);
    unsigned long long *stats = ptm::own_sgprs(stats_arg);
    const uint32_t slot_base = ptm::own_sgprs(slot_base_arg), n_slots = ptm::own_sgprs(n_slots_arg);
    const int refill = ptm::own_sgprs(refill_arg);
    const float tmin = ptm::own_sgprs(tmin_arg), tmax = ptm::own_sgprs(tmax_arg);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // ---- LDS: stack | BVH4 nodes | three permuted triangle copies | shade4 | tangent frames | path state
    float4 *s_wide = reinterpret_cast<float4 *>(smem + (size_t)lds_stack * FTB * sizeof(uint32_t));
    float4 *s_tri = s_wide + LDS_NODE_F4 * (size_t)n_wide;
    float4 *s_shade = s_tri + 9 * (size_t)n_tris;
    float4 *s_frame = s_shade + 3 * (size_t)n_tris;
    lds_u32 *my_state = (lds_u32 *)reinterpret_cast<uint32_t *>(s_frame + 2 * (size_t)n_tris) + threadIdx.x;
    for (uint32_t i = threadIdx.x; i < 8 * n_wide; i += FTB) {
        float4 v = g_wide[i];
        if ((i & 7u) == 6u) {  // the four child words -> 14-bit codes (extend_kernel.h COMPACT)
            auto cw = [](float f) {
                const uint32_t w = __float_as_uint(f);
                const uint32_t c = (w & PT_LEAF) ? (0x2000u | (((w >> 28) & 3u) << 11) | (w & 0x7FFu)) : (w & 0x1FFFu);
                return __uint_as_float(w == SENTINEL ? 0x3FFFu : c);
            };
            v = make_float4(cw(v.x), cw(v.y), cw(v.z), cw(v.w));
        }
        s_wide[(i >> 3) * LDS_NODE_F4 + (i & 7u)] = v;
    }
    for (uint32_t i = threadIdx.x; i < 3 * n_tris; i += FTB) {
        const float4 v = g_tri4[i];
        s_tri[i] = make_float4(v.y, v.z, v.x, v.w);               // kz = 0: (kx,ky,kz) = (1,2,0)
        s_tri[3 * n_tris + i] = make_float4(v.z, v.x, v.y, v.w);  // kz = 1: (2,0,1)
        s_tri[6 * n_tris + i] = v;                                // kz = 2: (0,1,2) -- also what the shade block reads
        s_shade[i] = g_shade4[i];
    }
    for (uint32_t i = threadIdx.x; i < 2 * n_tris; i += FTB) s_frame[i] = g_frame4[i];
    __syncthreads();
    const float4 *wide = s_wide, *tri4 = s_tri;
    const float4 *verts = s_tri + 6 * (size_t)n_tris;

    // (the stack's level 0 is the SECOND of the lds_stack levels: the first, never written, is where the node step's read of the top entry lands
    // when the stack is empty -- an address without a compare and a select)
    lds_u32 *my_stack32 = (lds_u32 *)reinterpret_cast<uint32_t *>(smem) + FTB + threadIdx.x;
    const int lane = threadIdx.x & 63;

    FusedDev dev;  // (fused_dev.h: empty in the product build)
    dev.kernel_begin();
    // COUNT: {wave executions, lanes inside} of every block, kept per wave in LDS behind the product kernel's plan (no registers: the walk keeps
    // its allocation) and added by the first active lane of the moment
    lds_u32 *s_fb = (lds_u32 *)reinterpret_cast<uint32_t *>(s_frame + 2 * (size_t)n_tris) + FS_FIELDS * FTB + (FTB / 64) * PT_FUSED_WTILES + (threadIdx.x >> 6) * (2 * FB_N);
    if constexpr (COUNT) {
        if (lane < 2 * FB_N) s_fb[lane] = 0u;
    }
#define PT_FB(B)                                                                                        \
    if constexpr (COUNT) {                                                                              \
        const unsigned long long m_fb = __ballot(1);                                                    \
        if (lane == __ffsll((long long)m_fb) - 1) { s_fb[2 * (B)] += 1u; s_fb[2 * (B) + 1] += (uint32_t)__popcll(m_fb); } \
    }
    // (a lane traces a ray <=> cur != DONE: the shade block sets cur = 0 with the new ray, the walk ends with cur == DONE.  No flag is kept: a
    // loop-carried bool lives in a lane mask, and every ballot of one costs a v_cndmask + v_cmp to clear its inactive lanes -- a compare does not)
    // (the lane owns a live path -- its state is in LDS; not tracing: a hit record awaits shading -- <=> sp >= 0: a lane without a path holds sp = -1.
    // No flag for that either: the loop's head asks for three wave masks of it per pass, and a ballot of a flag is two vector instructions, of a
    // compare one; the masks are combined as 64-bit integers on the scalar unit)
    bool out_of_slots = false;  // wave-uniform: the slot counter ran past the end
    uint32_t n_rays_wave = 0;   // wave-uniform: rays this wave started
    uint32_t n_cull_wave = 0;   // ... of them camera rays of pixels that cannot see the scene, resolved without a walk
    uint32_t w_next = 0, w_end = 0, w_base = 0;  // wave-uniform: what is left of the wave's current batch of slots, and where it began
    uint32_t w_part = blockIdx.x % (uint32_t)PT_FUSED_PARTS, w_tried = 0;  // ... the part of the slot range it draws from, parts found empty
    bool w_tails = false;       // wave-uniform, MODE 2: the head slots are all handed out, the wave draws tail slots (a second set of eight counters)
    const uint32_t n_tail_slots = HYB ? rc.lanes_active * rc.tail * rc.slots_per_lane : 0u;  // ... of this launch (n_slots: its head slots)
    // several groups: the slot range is cut into PT_FUSED_PARTS contiguous parts.  One group: part p is every PT_FUSED_PARTS-th 64-slot chunk
    // of the HAND-OUT order -- tile-major, all frames of a tile before the next tile (chunk q = tile q / frames, frame q % frames) -- so every part,
    // like the whole launch, runs from the image's centre to its border (film_work.hip numbers the tiles that way) and ENDS with border pixels of
    // all frames: slots of 32 rays where an interior one has a hundred and more.  What runs alone at the end of a launch is then short: a rank of
    // world 8 is not helped (its loss is elsewhere), 16 frames on one device are: 39.76 -> 40.8 Grays/s, two of two rounds; K = 8 +-0, K = 4 +2 %; the two-level
    // kernel loses 1 % with it and keeps the frame-major order (profiles/r04ag_ab_fused_tilemajor.log).  (Slot numbers, and with them the radiance arrays, are frame-major as before.)
    const uint32_t part_len = ((n_slots + PT_FUSED_PARTS - 1) / PT_FUSED_PARTS + 63u) & ~63u;  // (several groups)
    lds_u32 *s_wtile = (lds_u32 *)reinterpret_cast<uint32_t *>(s_frame + 2 * (size_t)n_tris) + FS_FIELDS * FTB + (threadIdx.x >> 6) * PT_FUSED_WTILES;
    ptm::f3 inv{}, invf{}, on{}, of{}, orgp{};
    ptm::RayPre pre{};
    uint32_t ax = 0, ay = 0, az = 0, tri_base = 0;
    float best_t = tmax, best_V = 0.f, best_W = 0.f, best_det = 1.f;
    uint32_t best_pos = PT_MISS;
    uint32_t cur = DONE;
    int sp = -1;

    // The stack's pop with the cull against best_t: a loop on the WAVE's condition -- the lanes that are served sit out behind one exec mask (see the
    // node loop) -- with selects inside, not branches (two more vector instructions for eight fewer scalar ones per pass).  r: PENDING for the lanes
    // that still look for an entry
    constexpr uint32_t PENDING = 0xFFFFFFFFu;
    auto pop_pending = [&](uint32_t r) -> uint32_t {
        while (__ballot(r == PENDING)) {
            if (r == PENDING) {
                PT_FB(FB_POP)
                sp--;
                const uint32_t e = my_stack32[sp * FTB];
                const uint32_t miss = sp == 0 ? DONE : PENDING;
                r = __uint_as_float(e & 0xFFFFC000u) <= best_t ? (e & 0x3FFFu) : miss;
            }
        }
        return r;
    };
    auto pop = [&]() -> uint32_t { return pop_pending(sp > 0 ? PENDING : DONE); };

    for (;;) {
        PT_FB(FB_ITER)
        // ---- shade block: the lanes that wait with a hit (or with nothing, while slots are left) -- once enough of them do
        const bool have = cur != DONE, path = sp >= 0;
        const unsigned long long m_have = __ballot(have), m_path = __ballot(path);
        const unsigned long long m_in_blk = ~m_have & (out_of_slots ? m_path : ~0ull);  // (waves are whole: FTB is a multiple of 64)
        const bool in_blk = !have && (path || !out_of_slots);
        const int n_work = __popcll(m_in_blk);
        if (n_work && n_work * 64 >= refill * (n_work + __popcll(m_have))) {  // (refill <= 64: a wave without a tracing lane always passes)
            uint32_t slot = 0, ctr = 0, seed = 0, pxy = 0;
            float wr = 0.f, wg = 0.f, wb = 0.f;
            ptm::f3 org{}, dir{};
            bool got_ray = false, need_primary = false, bounce = false;
            if (in_blk) { PT_FB(FB_SHADE) }
            // (1) the hit of the ray that just ended: radiance, then bounce / next sample / slot complete
            if (in_blk && path) {
                PT_FB(FB_HIT)
                slot = my_state[FS_SLOT * FTB]; ctr = my_state[FS_CTR * FTB]; seed = my_state[FS_SEED * FTB];
                wr = __uint_as_float(my_state[FS_WR * FTB]); wg = __uint_as_float(my_state[FS_WG * FTB]); wb = __uint_as_float(my_state[FS_WB * FTB]);
                pxy = my_state[FS_PXY * FTB];
                uint32_t sample = ctr & 0xFFFFu, depth = ctr >> 16;
                // raygen.rgen:76 `color += weight * emission` (adding +0 changes no bit of a non-negative accumulator, so
                // non-emitters are skipped -- as k_shade does; NaN compares false and still adds)
                float er, eg, eb;
                bool terminated, add;
                const uint32_t pos = best_pos;
                if (pos == PT_MISS) {  // miss.rmiss:10-11 then raygen.rgen:76, 81-83
                    PT_FB(FB_MISS)
                    er = wr * rc.env[0]; eg = wg * rc.env[1]; eb = wb * rc.env[2];
                    add = true;
                    terminated = true;
                } else {
                    PT_FB(FB_SURFACE)
                    const float4 s1 = s_shade[3 * pos + 1], s2 = s_shade[3 * pos + 2];
                    er = wr * s1.z; eg = wg * s1.w; eb = wb * s2.x;
                    add = !(er == 0.f && eg == 0.f && eb == 0.f);
                    depth++;
                    terminated = depth >= rc.max_depth;  // raygen.rgen:62
                }
                const bool logs = GROUPED || (HYB && slot >= rc.n_head);  // (a slot whose radiance goes through the term log)
                const uint32_t lslot = HYB ? slot - rc.n_head : slot;     // ... its place in the log arrays
                const uint32_t lstride = HYB ? rc.n_tail : rc.n_slots;
                if (add) {
                    PT_FB(FB_ADD)
                    if (!logs) {
                        my_state[FS_A * FTB] = __float_as_uint(__uint_as_float(my_state[FS_A * FTB]) + er);
                        my_state[FS_B * FTB] = __float_as_uint(__uint_as_float(my_state[FS_B * FTB]) + eg);
                        my_state[FS_C * FTB] = __float_as_uint(__uint_as_float(my_state[FS_C * FTB]) + eb);
                    } else {  // the ordered term log of add_radiance (wavefront_types.h), the count kept in LDS
                        const uint32_t k = my_state[FS_A * FTB];
                        if (dbg_no_terms && k != 0xFFFFFFFFu) {}  // (fused_dev.h: false in the product build)
                        else if (k < rc.term_pcap) ptm::st_stream<true>(rad.terms + ((size_t)k * lstride + lslot), make_float4(er, eg, eb, 0.f));
                        else if (k < rc.term_cap) ptm::st_stream<true>(rad.dev->terms_over + ((size_t)lslot * (rc.term_cap - rc.term_pcap) + (k - rc.term_pcap)), make_float4(er, eg, eb, 0.f));  // (rad.dev: wavefront_types.h)
                        else {
                            const Radiance rr = *rad.dev;
                            const unsigned long long idx = atomicAdd(rr.spill_count, 1ull);
                            if (idx < rr.spill_cap) {
                                // (the slot's first pool entry ends its chain: no per-slot initialisation of the heads)
                                rr.spill[idx] = make_float4(er, eg, eb, __uint_as_float(k == rc.term_cap ? SPILL_NONE : rr.spill_head[lslot]));
                                rr.spill_head[lslot] = (uint32_t)idx;
                            } else {
                                *rr.overflow = 1ull;
                            }
                        }
                        my_state[FS_A * FTB] = k + 1u;
                    }
                }
                if (!terminated) {
                    bounce = true;  // (the bounce itself: step (3), beside the camera rays)
                } else {
                    PT_FB(FB_NEXT)
                    sample++;
                    depth = 0;
                    bool more;
                    if (HYB) {
                        more = !logs && sample < rc.head_samples;  // (a tail slot is one sample)
                    } else if (!GROUPED) {
                        more = sample < rc.spp;  // (one group: the slot is the pixel's whole frame)
                    } else {
                        const uint32_t lane_slot = rc.div_spl.div(slot);  // = frame lane * groups + sample group (slot_pixel)
                        const uint32_t g = lane_slot - rc.div_groups.div(lane_slot) * rc.groups;
                        more = sample < min(rc.spp, (g + 1u) * rc.group_size);
                    }
                    if (more) {
                        need_primary = true;  // the slot's next sample: raygen.rgen:45-60
                    } else {  // the slot is complete
                        PT_FB(FB_DONE)
                        if (!logs) rad.color[slot] = make_float4(__uint_as_float(my_state[FS_A * FTB]), __uint_as_float(my_state[FS_B * FTB]),
                                                                   __uint_as_float(my_state[FS_C * FTB]), 0.f);
                        else if (!dbg_no_nterm) rad.nterm[lslot] = my_state[FS_A * FTB];
                        sp = -1;  // (no path)
                        dev.slot_end(slot);
                    }
                }
                ctr = sample | (depth << 16);
            }
            // (2) new slots for the lanes without a path (wave-uniform part).  A wave draws PT_FUSED_BATCH consecutive slots per
            // atomic and the tile words that give them their pixels with it: the two dependent round trips to memory (~3 us) are
            // paid once per batch -- per lane and shade block, as the first version did, they were 3/4 of the kernel's time.
            const unsigned long long m_want = __ballot(in_blk && sp < 0);
            if (m_want && !out_of_slots) {
                if (in_blk && sp < 0) { PT_FB(FB_HANDOUT) }
                if (w_next >= w_end) {
                    PT_FB(FB_DRAW)
                    // The slots are cut into PT_FUSED_PARTS contiguous parts with a counter each, 128 B apart; a wave starts on part
                    // blockIdx % 8 -- workgroups go to the eight XCDs round-robin, so the waves of one XCD share a word -- and moves
                    // on to the next part when its own is exhausted (work stealing, in ring order) until all eight are.  One word takes
                    // ~88 atomics per microsecond on this chip; shapes with many short slots (several sample groups) ask for more.
                    // One group: PT_FUSED_BATCH1 = one tile per draw (see there).  Several groups: always PT_FUSED_BATCH.
                    for (;;) {
                        // (one group: w_base / w_next / w_end count within the part)
                        const bool grp = GROUPED || (HYB && w_tails);  // (contiguous parts of the slot range, as with several groups)
                        const uint32_t ns = (HYB && w_tails) ? n_tail_slots : n_slots;
                        const uint32_t plen = HYB ? ((ns + PT_FUSED_PARTS - 1) / PT_FUSED_PARTS + 63u) & ~63u : part_len;
                        const uint32_t part_begin = grp ? w_part * plen : 0u,
                                       part_end = grp ? min(part_begin + plen, ns)
                                                      : (((ns >> 6) + (uint32_t)PT_FUSED_PARTS - 1u - w_part) / (uint32_t)PT_FUSED_PARTS) << 6;
                        uint32_t rel = 0, size = 0;
                        if (lane == 0) {
                            uint32_t *cnt = next_slot + (w_part + ((HYB && w_tails) ? (uint32_t)PT_FUSED_PARTS : 0u)) * (uint32_t)PT_FUSED_PART_STRIDE;
                            size = (uint32_t)(grp ? PT_FUSED_BATCH : PT_FUSED_BATCH1);
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
                        if (++w_tried >= (uint32_t)PT_FUSED_PARTS) {
                            if (HYB && !w_tails && n_tail_slots) {  // every head slot is handed out: on to the tail slots
                                w_tails = true;
                                w_tried = 0;
                                w_part = blockIdx.x % (uint32_t)PT_FUSED_PARTS;
                                continue;
                            }
                            out_of_slots = true;
                            w_end = w_next;
                            break;
                        }
                    }
                    if (out_of_slots) dev.out_of_slots();
                    if (!out_of_slots && (uint32_t)lane < (w_end - w_base + 63u) / 64u) {
                        if (GROUPED || (HYB && w_tails)) {
                            const uint32_t c = slot_base + w_base + 64u * (uint32_t)lane;   // a 64-aligned chunk of slots = one 8x8 tile
                            s_wtile[lane] = tiles[(c - rc.div_spl.div(c) * rc.slots_per_lane) >> 6];
                        } else {
                            s_wtile[lane] = tiles[div_frames.div(((w_base >> 6) + (uint32_t)lane) * (uint32_t)PT_FUSED_PARTS + w_part)];
                        }
                    }
                    __builtin_amdgcn_wave_barrier();  // (a wave's LDS operations execute in order: the reads below see these words)
                }
                const uint32_t take = min((uint32_t)__popcll(m_want), w_end - w_next);
                const uint32_t rank = (uint32_t)__popcll(m_want & ((1ull << lane) - 1ull));
                uint32_t cull_n = 0u;  // samples of a slot that is finished here: its pixel cannot see the scene (RenderConst::cull)
                if (in_blk && sp < 0 && rank < take) {
                    PT_FB(FB_TAKE)
                    const uint32_t mine = w_next + rank;
                    uint32_t f, g, local;
                    if (HYB && w_tails) {  // tail slot `mine` of the launch: (frame lane, tail j, pixel); sample head_samples + j
                        const uint32_t lane_slot = rc.div_spl.div(mine);
                        f = rc.div_tail.div(lane_slot); g = lane_slot - f * rc.tail;
                        local = mine - lane_slot * rc.slots_per_lane;
                        slot = rc.n_head + mine;
                    } else if (GROUPED) {
                        slot = slot_base + mine;
                        const uint32_t lane_slot = rc.div_spl.div(slot);
                        f = rc.div_groups.div(lane_slot); g = lane_slot - f * rc.groups;
                        local = slot - lane_slot * rc.slots_per_lane;
                    } else {  // hand-out order -> slot: chunk q of the order is (tile q / frames, frame q % frames)
                        const uint32_t q = (mine >> 6) * (uint32_t)PT_FUSED_PARTS + w_part;
                        const uint32_t t = div_frames.div(q);
                        f = q - t * rc.lanes_active; g = 0u;
                        local = t * 64u + (mine & 63u);
                        slot = slot_base + f * rc.slots_per_lane + local;
                    }
                    const uint32_t tw = s_wtile[(mine - w_base) >> 6];
                    const uint32_t px = (tw & 0xFFFFu) * 8u + (local & 7u), py = (tw >> 16) * 8u + ((local >> 3) & 7u);
                    const uint32_t sample0 = (HYB && w_tails) ? rc.head_samples + g : g * rc.group_size;
                    const bool in_image = px < rc.width && py < rc.height && f < rc.lanes_active && sample0 < rc.spp;
                    if (in_image && ptc::pixel_culled(rc, px, py)) {
                        PT_FB(FB_CULLED)
                        // every sample of the slot: one camera ray, a miss, color += 1 * env -- without the walk that would find nothing (fused_cull.h)
                        if (GROUPED) cull_n = ptc::finish_group(rc, rad, slot, g);
                        else if (!(HYB && w_tails)) cull_n = ptc::finish_plain(rc, rad, slot);
                        // (head + tail: the head slot stands for ALL spp samples of such a pixel -- the same adds in the same order -- and its
                        // tail slots are nothing: k_resolve does not replay the logs of a pixel outside the rectangle)
                    } else if (in_image) {
                        pxy = px | (py << 16);
                        ctr = sample0;
                        my_state[FS_A * FTB] = 0u;  // colour.r = +0.0f | term count = 0
                        if (!GROUPED) { my_state[FS_B * FTB] = 0u; my_state[FS_C * FTB] = 0u; }
                        my_state[FS_MB * FTB] = (uint32_t)((int32_t)rc.spp * (rc.frame_base + (int32_t)f)) + 1u;
                        sp = 0;  // (a path)
                        need_primary = true;
                        dev.slot_begin();
                    } else if (GROUPED) {
                        rad.nterm[slot] = 0u;  // (a slot outside the image or the batch: k_resolve never reads it, kept defined anyway)
                    } else if (HYB && w_tails) {
                        rad.nterm[slot - rc.n_head] = 0u;
                    }
                }
                w_next += take;
                if (rc.cull_on) {  // (wave-uniform: the rays of the slots finished above, counted as the rays they are)
                    const uint32_t n_c = ptc::rays_finished<GROUPED>(rc, cull_n);
                    n_rays_wave += n_c;
                    n_cull_wave += n_c;
                }
            }
            // (3) the new ray: a bounce (closesthit.rchit:56-57, raygen.rgen:77-80) or the camera ray of a slot's next / first sample
            // (raygen.rgen:45-60).  Both draw two rand and take one square root -- sqrt(1 - r1^2) of the hemisphere sample, the length of the
            // camera ray's direction -- and those steps run ONCE for both kinds of lanes: the camera rays are a fifth of the rays, so a block of
            // their own ran in every shade block at ten lanes of 64 (the block table of round 6: 11 % of the kernel's instructions).  Same
            // operations on the same operands in the same order per lane: same bits.
            if (need_primary) {
                PT_FB(FB_PRIMARY)
                const uint32_t m = (ctr & 0xFFFFu) + my_state[FS_MB * FTB];  // = sample + maxSamples * frame + 1 (ptm::make_seed)
                const uint2 sd = ptm::pcg2d(make_uint2((pxy & 0xFFFFu) * m, (pxy >> 16) * m));
                seed = sd.x + sd.y;
                wr = wg = wb = 1.0f;  // raygen.rgen:59
            }
            if (bounce || need_primary) {
                PT_FB(FB_SPAWN)
                const float r1 = ptm::rnd(seed);  // bounce: cos(theta) first, azimuth second; camera ray: x jitter first, then y
                const float r2 = ptm::rnd(seed);
                float vx = 0.f, vy = 0.f, vz = 0.f, sq_arg;
                if (need_primary) {
                    PT_FB(FB_PTARGET)
                    ptm::primary_target(rc.cam, pxy & 0xFFFFu, pxy >> 16, r1, r2, vx, vy, vz);
                    sq_arg = (vx * vx + vy * vy) + vz * vz;
                } else {
                    sq_arg = 1.0f - r1 * r1;
                }
                const float sq = ptm::fsqrt(sq_arg);
                // ... and one set of quotients by a common divisor: the camera ray's direction (target - origin) / length, the bounce's barycentrics
                // (V, W) / det (closesthit.rchit:56 through the hit record's undivided numerators) -- div3_dominant serves both (the bounce's third
                // numerator is its second once more: same guard, same bits).  -1.5 %, profiles/r06w_merged_division.log
                float q1, q2, q3;
                ptm::div3_dominant(need_primary ? vx : best_V, need_primary ? vy : best_W, need_primary ? vz : best_W, need_primary ? sq : best_det, q1, q2, q3);
                if (need_primary) {
                    PT_FB(FB_PDIR)
                    org = { rc.cam.ox, rc.cam.oy, rc.cam.oz };
                    dir = { q1, q2, q3 };
                } else {
                    PT_FB(FB_BOUNCE)
                    const uint32_t pos = best_pos;
                    const float4 s0 = s_shade[3 * pos + 0], s1 = s_shade[3 * pos + 1];
                    const ptm::f3 nrm = { s0.x, s0.y, s0.z };
                    const float4 f0 = s_frame[2 * pos + 0], f1 = s_frame[2 * pos + 1];
                    dir = ptm::sample_direction_frame_sq(r1, r2, sq, nrm, { f0.x, f0.y, f0.z }, { f0.w, f1.x, f1.y });
                    const float dt = (dir.x * nrm.x + dir.y * nrm.y) + dir.z * nrm.z;
                    float fr = s0.w * dt, fg = s1.x * dt, fb = s1.y * dt;
                    ptm::div3_by_pdf(fr, fg, fb);
                    wr = wr * fr; wg = wg * fg; wb = wb * fb;
                    const float4 a = verts[3 * pos + 0], b = verts[3 * pos + 1], c = verts[3 * pos + 2];
                    const float hu = q1, hv = q2;
                    const float b0 = (1.0f - hu) - hv;
                    org = { (a.x * b0 + b.x * hu) + c.x * hv, (a.y * b0 + b.y * hu) + c.y * hv, (a.z * b0 + b.z * hu) + c.z * hv };
                }
                got_ray = true;
            }
            // (4) state back to LDS, ray set-up (the refill block of extend_body<true, false, false, true>)
            if (got_ray) {
                PT_FB(FB_SETUP)
                my_state[FS_SLOT * FTB] = slot; my_state[FS_CTR * FTB] = ctr; my_state[FS_SEED * FTB] = seed;
                my_state[FS_WR * FTB] = __float_as_uint(wr); my_state[FS_WG * FTB] = __float_as_uint(wg); my_state[FS_WB * FTB] = __float_as_uint(wb);
                my_state[FS_PXY * FTB] = pxy;
                pre = ptm::ray_setup<true>(org, dir);
                inv = { ptm::safe_inv(dir.x), ptm::safe_inv(dir.y), ptm::safe_inv(dir.z) };
                slab_setup(org, inv, invf, on, of);
                ax = inv.x < 0.f ? 48u : 0u;
                ay = inv.y < 0.f ? 48u : 0u;
                az = inv.z < 0.f ? 48u : 0u;
                tri_base = (uint32_t)pre.kz * 3u * n_tris;
                orgp = { ptm::sel3(pre.kz, org.y, org.z, org.x), ptm::sel3(pre.kz, org.z, org.x, org.y), ptm::sel3(pre.kz, org.x, org.y, org.z) };
                best_t = tmax; best_V = 0.f; best_W = 0.f; best_det = 1.f;
                best_pos = PT_MISS;
                cur = 0u;
                sp = 0;
            }
            const uint32_t n_started = (uint32_t)__popcll(__ballot(got_ray));  // (wave-uniform control flow here: every lane keeps the same count)
            n_rays_wave += n_started;
            dev.rays_started(n_started, HYB && w_tails, lane);
        }
        // Nobody tracing, but hits pending or slots left: the shade block runs in the next pass.  The pass FALLS THROUGH the two phases (no lane
        // enters either) instead of `continue`: a second way back to the loop's head kept the hit record's old registers alive across the leaf
        // phase, and the compiler copied the five of them at every block boundary of it -- ~30 v_mov per pass (profiles/r06x_one_latch.log)
        const bool tracing = cur != DONE;
        const unsigned long long m_tracing = __ballot(tracing);
        if (m_tracing == 0ull && __ballot(sp >= 0) == 0ull && out_of_slots) break;

        // ---- node phase (extend_body, LDS_SCENE && COMPACT): every lane descends until it holds a leaf
        const int n_have = __popcll(m_tracing);
        if (tracing) { PT_FB(FB_TRACE) }
        // The loop's condition is the WAVE's: lanes that hold a leaf sit out a pass behind one exec mask instead of leaving a divergent loop, whose
        // bookkeeping of the lanes that left cost ~15 scalar instructions per step (the scalar unit is one per CU, and this kernel kept it 70 % busy:
        // scripts/ubench/salu_rate.hip, profiles/r06z_*).  cur of a lane without a ray is DONE, which carries the leaf bit: the mask of the lanes
        // that step is one compare.  The node loop ends once fewer than PT_FUSED_NODE_EXIT_B / PT_FUSED_NODE_EXIT of the tracing lanes still descend;
        // the first step is unconditional (the vote before every step: +0.5 %; as a `for` with a first-pass flag: +4.5 %, the compiler peels it).
        {
            bool dn = cur < LEAF_BIT;  // (an inner node: the codes of leaves and DONE carry the leaf bit)
            if (__ballot(dn) != 0ull) {
                int n_cont;
                do {
                    if (dn) {
                        PT_FB(FB_NODE)
                        // (the stack's top entry, read WITH the node's planes: a step whose four children all miss pushed nothing, so that entry is what
                        // its pop looks at first -- from a register instead of behind an LDS round trip: -1.8 %, profiles/r06r_speculative_pop.log.  Level -1,
                        // never written, is what the read lands on at an empty stack)
                        const uint32_t e_top = my_stack32[(sp - 1) * FTB];
                        cur = compact_node_step<FTB>(wide, cur, inv, invf, on, of, ax, ay, az, tmin, best_t, my_stack32, sp, [&]() -> uint32_t {
                            PT_FB(FB_POPTOP)
                            // (the top entry as the pop's first candidate, by selects: empty stack -> DONE; entry within best_t -> taken; else the loop)
                            const bool any = sp > 0;
                            sp -= any ? 1 : 0;
                            const uint32_t below = sp > 0 ? PENDING : DONE;
                            const uint32_t top = __uint_as_float(e_top & 0xFFFFC000u) <= best_t ? (e_top & 0x3FFFu) : below;
                            return pop_pending(any ? top : DONE);
                        });
                    }
                    dn = cur < LEAF_BIT;
                    n_cont = __popcll(__ballot(dn));
                } while (n_cont * PT_FUSED_NODE_EXIT >= n_have * PT_FUSED_NODE_EXIT_B);  // (n_have >= 1 in here: no lane left ends it too)
            }
        }
        // ---- leaf phase (extend_body, PAIRS): one triangle or one fan pair per leaf
        {   // (no `if (tracing)` around it: a lane without a ray holds DONE, and one compare says "a leaf that is not DONE")
            if ((uint32_t)(cur - LEAF_BIT) < DONE - LEAF_BIT) {
                PT_FB(FB_LEAF)
                if (PAIRS) {
                    const uint32_t first = cur & 0x7FFu;
                    ptl::pair_leaf_test<true>(tri4, (size_t)tri_base + 3 * (size_t)first, ((cur >> 11) & 3u) != 0u, first, pre, orgp, tmin, tmax,
                                        [&](float t, float V, float W, float det, uint32_t pos, uint32_t) {
                                            ptl::closer_single_level(tri4, tri_base, t, V, W, det, pos, best_t, best_V, best_W, best_det, best_pos);
                                        },
                                        [&] { PT_FB(FB_DIV) });
                } else {
                    const uint32_t first = cur & 0x7FFu, cnt = ((cur >> 11) & 3u) + 1u;
                    for (uint32_t k = 0; k < cnt; k++) {
                        const uint32_t pos = first + k;
                        const size_t ti = (size_t)tri_base + 3 * (size_t)pos;
                        const float4 a = tri4[ti + 0], b = tri4[ti + 1], c = tri4[ti + 2];
                        float t, V, W, det;
                        if (ptm::tri_test_perm(pre, orgp, { a.x, a.y, a.z }, { b.x, b.y, b.z }, { c.x, c.y, c.z }, tmin, tmax, t, V, W, det, nullptr)) {
                            bool closer = t < best_t;  // closest t; equal t -> lowest gl_PrimitiveID (the first vertex of a record carries it)
                            if (!closer && t == best_t)
                                closer = best_pos == PT_MISS || __float_as_uint(a.w) < __float_as_uint(tri4[(size_t)tri_base + 3 * (size_t)best_pos].w);
                            if (closer) { best_t = t; best_V = V; best_W = W; best_det = det; best_pos = pos; }
                        }
                    }
                }
                cur = pop();
            }
            if (tracing && cur == DONE) {  // the hit (best_pos, best_V, best_W, best_det) waits in registers for the shade block
                PT_FB(FB_FINISH)
            }
        }
    }
    if (lane == 0 && n_rays_wave) atomicAdd(stats, (unsigned long long)n_rays_wave);
    if (lane == 0 && n_cull_wave) atomicAdd(stats + 19, (unsigned long long)n_cull_wave);  // (pt_stats.rays_culled)
    dev.kernel_end(lane, n_rays_wave, FTB);
    if constexpr (COUNT) {  // (pt_get_block_counts: stats[PT_N_STATS + 2 * block] wave executions, [+ 1] lanes inside them)
        if (lane < 2 * FB_N) atomicAdd(stats + PT_N_STATS + lane, (unsigned long long)s_fb[lane]);
    }
#undef PT_FB
}

template <int MODE, bool PAIRS>
__global__ __launch_bounds__(FTB, PT_FUSED_WAVES) void k_fused(RenderConst rc, const uint32_t *__restrict__ tiles, Radiance rad,
                                                              const float4 *__restrict__ g_wide, const float4 *__restrict__ g_tri4,
                                                              const float4 *__restrict__ g_shade4, const float4 *__restrict__ g_frame4,
                                                              uint32_t n_wide, uint32_t n_tris, uint32_t slot_base, uint32_t n_slots,
                                                              uint32_t *next_slot, unsigned long long *stats, int refill, float tmin,
                                                              float tmax, int lds_stack, FastDiv div_frames)
{
    fused_body<MODE, PAIRS, false>(rc, tiles, rad, g_wide, g_tri4, g_shade4, g_frame4, n_wide, n_tris, slot_base, n_slots, next_slot, stats, refill, tmin,
                                   tmax, lds_stack, div_frames);
}
// the instrumented twin (PT_FLAG_COUNT_VISITS): the same LDS plan and block size, so a wave holds what it holds in the timed kernel; the counters
// cost registers (spills), so it is slower and never timed
template <int MODE, bool PAIRS>
__global__ __launch_bounds__(FTB, PT_FUSED_WAVES) void k_fused_count(RenderConst rc, const uint32_t *__restrict__ tiles, Radiance rad,
                                                                    const float4 *__restrict__ g_wide, const float4 *__restrict__ g_tri4,
                                                                    const float4 *__restrict__ g_shade4, const float4 *__restrict__ g_frame4,
                                                                    uint32_t n_wide, uint32_t n_tris, uint32_t slot_base, uint32_t n_slots,
                                                                    uint32_t *next_slot, unsigned long long *stats, int refill, float tmin,
                                                                    float tmax, int lds_stack, FastDiv div_frames)
{
    fused_body<MODE, PAIRS, true>(rc, tiles, rad, g_wide, g_tri4, g_shade4, g_frame4, n_wide, n_tris, slot_base, n_slots, next_slot, stats, refill, tmin,
                                  tmax, lds_stack, div_frames);
}
