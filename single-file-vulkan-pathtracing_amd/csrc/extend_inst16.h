// extend_inst16.h -- two-level closest hit (BASELINE config C4), round-2 kernel.
//
// Same contract and same hit records as k_extend_inst (extend_inst.h), which stays as the general fallback; this one is
// for scenes whose TLAS and BLAS fit 15-bit child codes (< 32768 instances, BLAS <= 2047 triangles staged in LDS) and
// tmin > 0.  What differs:
//   * BOTH levels are 64-B nodes with fp16 planes (TLAS: normalised to the box of all instances, built top-down with
//     contiguous children, read from L2 with four 16-B loads instead of seven; BLAS: the scene's fp16 BVH4, staged in LDS
//     with its child words re-coded to 16 bits), so one slab routine serves both and the ray is re-normalised at every
//     instance entry / exit;
//   * a stack entry is ONE dword: bf16-truncated entry distance | 16-bit child code; keys are formed before the sort, so
//     the network is min/max on integers (extend_kernel.h explains the trick), and the LDS stack is half the size;
//   * BLAS leaves that are fan pairs are tested with shared vertex work (PAIRS, as k_extend_lds7p).
// Codes: TLAS / BLAS inner node = index; TLAS leaf = 0x8000 | instance position (one instance per leaf); BLAS leaf =
// 0x8000 | (count - 1) << 11 | first; 0x7FFF = the EXIT marker under an instance's entries; 0xFFFF = empty / done.
#pragma once
#include "extend_kernel.h"

namespace {

constexpr uint32_t I16_EXIT = 0x7FFFu, I16_DONE = 0xFFFFu, I16_LEAF = 0x8000u;
constexpr uint32_t I16_NODE_DW = 20;  // dwords per BLAS node in LDS (16 used): 80-B stride spreads the banks

// whole 16-B loads in each branch, then a register barrier: left alone the compiler turns the near / far plane selects
// into address selects and reads the node with thirteen FLAT loads that serve both address spaces
#define PT_REG_BARRIER16(A, B, C, D)                                                                                   \
    asm volatile("" : "+v"(A.x), "+v"(A.y), "+v"(A.z), "+v"(A.w), "+v"(B.x), "+v"(B.y), "+v"(B.z), "+v"(B.w), "+v"(C.x), \
                      "+v"(C.y), "+v"(C.z), "+v"(C.w), "+v"(D.x), "+v"(D.y), "+v"(D.z), "+v"(D.w));
// SHADOW (the NEE pipeline's shadow rays): a per-ray upper bound `ray_tmax` instead of tmax, and ANY hit below it ends the walk
template <bool COUNT, bool PAIRS, bool SHADOW = false>
__global__ __launch_bounds__(TB) void k_extend_inst16(const uint4 *__restrict__ tlas16, NormBox nbt, const uint4 *__restrict__ g_blas16,
                                                      NormBox nbb, const float4 *__restrict__ g_tri4, uint32_t n_blas_wide,
                                                      uint32_t n_tris, const float4 *__restrict__ inst6,
                                                      const uint32_t *__restrict__ inst_id, const float4 *__restrict__ rayA,
                                                      const float2 *__restrict__ rayB, float4 *__restrict__ hit,
                                                      uint32_t *__restrict__ hit_inst, const uint32_t *__restrict__ count_in,
                                                      uint32_t *count_zero, unsigned long long *stats, uint32_t *__restrict__ spill,
                                                      uint32_t spill_stride, int refill_min_idle, float tmin, float tmax, int raw_hit,
                                                      int lds_stack, int enter_min, int leaf_min, int node_yield, uint32_t n_tlas_lds,
                                                      const float *__restrict__ ray_tmax)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    uint32_t *s_stack = reinterpret_cast<uint32_t *>(smem);  // [lds_stack][TB]
    uint32_t *s_blas = s_stack + (size_t)lds_stack * TB;      // [n_blas_wide + n_tlas_lds][I16_NODE_DW]
    // the first n_tlas_lds TLAS nodes -- its top levels: the builder numbers the nodes level by level -- sit behind the BLAS
    // nodes, so a visit to one of them is the same LDS read as a BLAS node instead of four loads from L2
    float4 *s_tri = reinterpret_cast<float4 *>(s_blas + (size_t)I16_NODE_DW * (n_blas_wide + n_tlas_lds));
    for (uint32_t i = threadIdx.x; i < 4 * n_tlas_lds; i += TB)
        *reinterpret_cast<uint4 *>(s_blas + (size_t)(n_blas_wide + (i >> 2)) * I16_NODE_DW + 4 * (i & 3u)) = tlas16[i];
    for (uint32_t i = threadIdx.x; i < 4 * n_blas_wide; i += TB) {
        uint4 v = g_blas16[i];
        if ((i & 3u) == 3u) {  // the four child words -> 16-bit codes
            auto code = [](uint32_t w) {
                if (w == SENTINEL) return I16_DONE;
                return (w & PT_LEAF) ? (I16_LEAF | (((w >> 28) & 3u) << 11) | (w & 0x7FFu)) : (w & 0x7FFFu);
            };
            v = make_uint4(code(v.x), code(v.y), code(v.z), code(v.w));
        }
        *reinterpret_cast<uint4 *>(s_blas + (size_t)(i >> 2) * I16_NODE_DW + 4 * (i & 3u)) = v;
    }
    for (uint32_t i = threadIdx.x; i < 3 * n_tris; i += TB) {  // three axis-permuted copies (ptm::tri_test_perm)
        const float4 v = g_tri4[i];
        s_tri[i] = make_float4(v.y, v.z, v.x, v.w);
        s_tri[3 * n_tris + i] = make_float4(v.z, v.x, v.y, v.w);
        s_tri[6 * n_tris + i] = v;
    }
    __syncthreads();
    const uint32_t n = *count_in;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        if (count_zero) *count_zero = 0u;
        if (stats) atomicAdd(stats, (unsigned long long)n);
    }
    lds_u32 *my_stack = (lds_u32 *)s_stack + threadIdx.x;
    uint32_t *my_spill = spill + (size_t)blockIdx.x * TB + threadIdx.x;
    const float INF = __builtin_inff();
    const int lane = threadIdx.x & 63;
    const unsigned long long lt = (1ull << lane) - 1ull;

    bool have = false, exhausted = false, in_blas = false;
    uint32_t q = 0;
    ptm::f3 org_w{}, dir_w{};
    ptm::f3 inv{}, invf{}, on{}, of{}, orgp{};  // the level being walked, in that level's normalised coordinates
    uint32_t mx = 0, my = 0, mz = 0;  // all ones where the level's direction component is negative
    ptm::f3 w_inv{}, w_invf{}, w_on{}, w_of{};  // the TLAS level's constants, kept across instance visits
    uint32_t w_mx = 0, w_my = 0, w_mz = 0;
    uint32_t tri_base = 0;
    ptm::RayPre pre{};
    float best_t = tmax, best_V = 0.f, best_W = 0.f, best_det = 1.f;
    uint32_t best_pos = PT_MISS, best_prim = PT_MISS, best_ipos = PT_MISS, best_iid = PT_MISS;
    uint32_t cur = I16_DONE, cur_ipos = 0, cur_iid = 0;
    int sp = 0, sp_exit = 0;
    unsigned long long c_nodes = 0, c_tris = 0;
    // PT_FLAG_COUNT_VISITS: wave executions of the kernel's blocks and the lanes inside them (pt_stats, include/pt_api.h)
    unsigned long long c_node_steps = 0, c_tri_steps = 0, c_leaf_lanes = 0, c_enter_steps = 0, c_enter_lanes = 0, c_iters = 0,
                       c_refills = 0, c_finishes = 0, c_pops = 0, c_pop_lanes = 0, c_hit_blocks = 0, c_hit_lanes = 0;
#define PT_COUNT_WAVE(C) \
    if (COUNT && lane == __ffsll((long long)__ballot(1)) - 1) (C)++
    const uint32_t wave_base = (blockIdx.x * (TB / 64) + (threadIdx.x >> 6)) * 64u;
    const uint32_t wave_stride = gridDim.x * TB;
    uint32_t cursor = 0;

    auto level_setup = [&](const ptm::f3 o, const ptm::f3 d, const NormBox &nb) {  // slab constants of a level for ray (o, d)
        const ptm::f3 on_ = { (o.x - nb.cx) * nb.rsx, (o.y - nb.cy) * nb.rsy, (o.z - nb.cz) * nb.rsz };
        inv = { ptm::safe_inv(d.x) * nb.sx, ptm::safe_inv(d.y) * nb.sy, ptm::safe_inv(d.z) * nb.sz };
        slab_setup(on_, inv, invf, on, of);
        mx = inv.x < 0.f ? 0xFFFFFFFFu : 0u; my = inv.y < 0.f ? 0xFFFFFFFFu : 0u; mz = inv.z < 0.f ? 0xFFFFFFFFu : 0u;
    };
    auto push = [&](uint32_t e) {
        if (sp < lds_stack) my_stack[sp * TB] = e;
        else my_spill[(size_t)(sp - lds_stack) * spill_stride] = e;
        sp++;
    };
    // (pushes and pop loop duplicated in a form without the spill test, taken while the whole wave is within the LDS entries:
    // C4 -1.6 %, three of three rounds -- the second copy costs more than the test: profiles/r03bo_ab_c4_roomy.log)
    // (an instance visit ends when the stack is back at the height it had on entry, `sp_exit` -- no marker entry under the
    // instance's own entries, which cost the pop loop one iteration per visit for the one or two lanes that met it)
    // (a loop on the WAVE's condition -- the lanes that are served sit out behind one exec mask -- as in the fused kernels: fused_kernel.h)
    auto pop = [&]() -> uint32_t {
        constexpr uint32_t PENDING = 0xFFFFFFFFu;
        uint32_t r = PENDING;
        while (__ballot(r == PENDING)) {
            if (r == PENDING) {
                if (sp > 0) {
                    PT_COUNT_WAVE(c_pops);
                    if (COUNT) c_pop_lanes++;
                    if (in_blas && sp == sp_exit) in_blas = false;  // the instance is done: back to the world-space ray and the TLAS
                    sp--;
                    uint32_t e;
                    if (sp < lds_stack) e = my_stack[sp * TB];
                    else e = my_spill[(size_t)(sp - lds_stack) * spill_stride];
                    r = __uint_as_float(e & 0xFFFF0000u) <= best_t ? (e & 0xFFFFu) : PENDING;
                } else {
                    in_blas = false;
                    r = I16_DONE;
                }
            }
        }
        return r;
    };
    // ... whose slab constants wait in registers (four blocks per CU leave 128) and come back ONCE after the pop loop, for all
    // lanes that left an instance in it: inside the loop the fifteen moves ran in nearly every one of its 20 iterations per 64
    // rays, each time for the one or two lanes that met their marker in that iteration
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
        PT_COUNT_WAVE(c_iters);
        const unsigned long long idle = __ballot(!have);
        const int n_idle = __popcll(idle);
        if (!exhausted && n_idle >= refill_min_idle) {
            if (!have) {
                const uint32_t v = cursor + (uint32_t)__popcll(idle & lt);
                const uint32_t qq = (v >> 6) * wave_stride + wave_base + (v & 63u);
                if (qq < n) {
                    PT_COUNT_WAVE(c_refills);
                    q = qq;
                    const float4 ra = ptm::ld_stream<true>(rayA + q);
                    const float2 rb = ptm::ld_stream<true>(rayB + q);
                    org_w = { ra.x, ra.y, ra.z };
                    dir_w = { ra.w, rb.x, rb.y };
                    level_setup(org_w, dir_w, nbt);
                    w_inv = inv; w_invf = invf; w_on = on; w_of = of; w_mx = mx; w_my = my; w_mz = mz;
                    in_blas = false;
                    best_t = SHADOW ? ray_tmax[q] : tmax; best_V = 0.f; best_W = 0.f; best_det = 1.f;
                    best_pos = PT_MISS; best_prim = PT_MISS; best_ipos = PT_MISS; best_iid = PT_MISS;
                    cur = 0u;  // TLAS root
                    sp = 0;
                    have = true;
                }
            }
            cursor += (uint32_t)n_idle;
            exhausted = (cursor >> 6) * wave_stride + wave_base >= n;
        }
        if (__ballot(have) == 0ull) break;

        // ---- node phase, either level: one routine, the node comes from L2 (TLAS) or LDS (BLAS)
        // (node_yield > 0: once fewer than 1/node_yield of the wave's rays are still descending, the rest -- waiting with a
        // leaf -- goes first and the stragglers resume in the next outer iteration, as in the Cornell kernel)
        const int n_have = __popcll(__ballot(have));
        bool do_node = have && !(cur & I16_LEAF);
        if (__ballot(do_node) != 0ull) for (;;) {  // (on the wave's condition, the first step unconditional: fused_inst_kernel.h)
            if (do_node) {
            uint4 q0, q1, q2, cw;
            if (in_blas || cur < n_tlas_lds) {
                // (LDS-typed pointer: with generic ones the compiler folds the two branches into FLAT loads)
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
            if (COUNT) c_nodes++;
            PT_COUNT_WAVE(c_node_steps);
            // lo planes: q0.xy q0.zw q1.xy, hi planes: q1.zw q2.xy q2.zw (two children per dword)
            // near / far rows by bit-field insert with per-lane masks (mx = all ones where the direction component is negative):
            // twelve v_cndmask_b32 on VCC in a row issue at ~23 cycles each on this chip and made this
            // kernel stall on issue 3.7x as often as the fp32 one; v_bfi_b32 has no such hazard
            auto sel = [](uint32_t m, uint32_t a, uint32_t b) { return (m & a) | (~m & b); };  // m ? a : b, bitwise
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
            do_node = have && !(cur & I16_LEAF);
            const int n_cont = __popcll(__ballot(do_node));
            if (n_cont == 0 || (node_yield > 0 && n_cont * node_yield < n_have)) break;
        }
        // ---- leaf phase: a BLAS leaf (triangles) or a TLAS leaf (enter the instance)
        // entering costs ~130 VALU: lanes that want to wait until ENTER_MIN of them do, or no lane has triangle work
        const int ENTER_MIN = enter_min, LEAF_MIN = leaf_min;
        const bool at_leaf = have && (cur & I16_LEAF) && cur != I16_DONE;  // (a lane that yielded above still holds a node)
        const int n_enter = __popcll(__ballot(at_leaf && !in_blas));
        const int n_leaf = __popcll(__ballot(at_leaf && in_blas));
        const bool descending = __ballot(have && !(cur & I16_LEAF)) != 0ull;
        // triangle work, too, waits for LEAF_MIN lanes as long as something else moves (a lane still descending, or an entry that
        // runs): leaf steps 20.8 -> 25.2 lanes, node steps 44.8 -> 44.3, C4 +1.3 % (8; 16: +0.9 %, 24 and 32: -1 %,
        // profiles/r03k_ab_c4_leaf_min.log)
        const bool do_leaf = n_leaf >= LEAF_MIN || !(descending || n_enter >= ENTER_MIN);
        const bool others = descending || (do_leaf && n_leaf > 0);
        const bool do_enter = n_enter >= ENTER_MIN || !others;
        if (have) {
            if (at_leaf && in_blas && do_leaf) {
                const uint32_t first = cur & 0x7FFu, cnt = ((cur >> 11) & 3u) + 1u;
                if (COUNT) { c_tris += cnt; c_leaf_lanes += PAIRS ? 1u : cnt; }
                if (PAIRS) { PT_COUNT_WAVE(c_tri_steps); }
                auto accept = [&](float t, float V, float W, float det, uint32_t pos, uint32_t prim) {
                    if (ptl::closer_instanced(t, V, W, det, pos, prim, cur_ipos, cur_iid, best_t, best_V, best_W, best_det, best_pos, best_prim, best_ipos, best_iid) && SHADOW)
                        sp = 0;  // any hit will do: nothing pending any more (the pop below finds the stack empty)
                };
                if (PAIRS) {
                    ptl::pair_leaf_test(s_tri, (size_t)tri_base + 3 * (size_t)first, cnt == 2u, first, pre, orgp, tmin, tmax, accept,
                                        [&] {
                                            PT_COUNT_WAVE(c_hit_blocks);
                                            if (COUNT) c_hit_lanes++;
                                        });
                } else {
                    for (uint32_t k = 0; k < cnt; k++) {
                        PT_COUNT_WAVE(c_tri_steps);
                        const uint32_t pos = first + k;
                        const size_t ti = (size_t)tri_base + 3 * (size_t)pos;
                        const float4 a = s_tri[ti + 0], b = s_tri[ti + 1], c = s_tri[ti + 2];
                        float t, V, W, det;
                        bool divided = false;
                        if (ptm::tri_test_perm(pre, orgp, { a.x, a.y, a.z }, { b.x, b.y, b.z }, { c.x, c.y, c.z }, tmin, tmax, t, V, W, det, COUNT ? &divided : nullptr))
                            accept(t, V, W, det, pos, __float_as_uint(a.w));
                        if (COUNT && divided) { PT_COUNT_WAVE(c_hit_blocks); c_hit_lanes++; }
                    }
                }
                cur = pop_and_restore();
            } else if (at_leaf && !in_blas && do_enter) {
                // TLAS leaf: one instance.  The ray goes to object space un-normalised (t is the same parameter)
                const uint32_t first = cur & 0x7FFFu;
                PT_COUNT_WAVE(c_enter_steps);
                if (COUNT) c_enter_lanes++;
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
            if (cur == I16_DONE) {
                PT_COUNT_WAVE(c_finishes);
                const bool miss = best_pos == PT_MISS;
                ptm::st_stream<true>(hit + q, raw_hit ? make_float4(__uint_as_float(best_pos), best_V, best_W, best_det)
                                 : make_float4(__uint_as_float(best_pos), miss ? 0.f : best_t,
                                               miss ? 0.f : ptm::fdiv(best_V, best_det), miss ? 0.f : ptm::fdiv(best_W, best_det)));
                if (!SHADOW) hit_inst[q] = best_ipos;
                have = false;
            }
        }
    }
    if (COUNT) {
        for (int o = 32; o > 0; o >>= 1) {
            c_nodes += __shfl_xor(c_nodes, o, 64);
            c_tris += __shfl_xor(c_tris, o, 64);
            c_node_steps += __shfl_xor(c_node_steps, o, 64); c_tri_steps += __shfl_xor(c_tri_steps, o, 64);
            c_leaf_lanes += __shfl_xor(c_leaf_lanes, o, 64); c_enter_steps += __shfl_xor(c_enter_steps, o, 64);
            c_enter_lanes += __shfl_xor(c_enter_lanes, o, 64); c_iters += __shfl_xor(c_iters, o, 64);
            c_refills += __shfl_xor(c_refills, o, 64); c_finishes += __shfl_xor(c_finishes, o, 64);
            c_pops += __shfl_xor(c_pops, o, 64); c_pop_lanes += __shfl_xor(c_pop_lanes, o, 64);
            c_hit_blocks += __shfl_xor(c_hit_blocks, o, 64); c_hit_lanes += __shfl_xor(c_hit_lanes, o, 64);
        }
        if (lane == 0 && stats) {
            atomicAdd(stats + 2, c_nodes);
            atomicAdd(stats + 3, c_tris);
            atomicAdd(stats + 4, c_node_steps); atomicAdd(stats + 5, c_tri_steps);
            atomicAdd(stats + 8, c_refills); atomicAdd(stats + 9, c_pops); atomicAdd(stats + 10, c_hit_blocks);
            atomicAdd(stats + 11, c_finishes); atomicAdd(stats + 12, c_iters);
            atomicAdd(stats + 13, c_leaf_lanes); atomicAdd(stats + 14, c_pop_lanes); atomicAdd(stats + 15, c_hit_lanes);
            atomicAdd(stats + 16, c_enter_steps); atomicAdd(stats + 17, c_enter_lanes);
        }
    }
#undef PT_COUNT_WAVE
}

}  // namespace
