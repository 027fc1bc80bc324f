// extend_kernel.h -- the single-level closest-hit kernel (k_extend) and what it is made of.
//
// A header because its instantiations are compiled in two translation units with different instruction
// schedulers: extend_launch.hip (scene staged in LDS: the default scheduler) and extend_hbm.hip (scene walked out of
// L2/MALL/HBM: -mllvm -amdgpu-sched-strategy=max-ilp, +10 % on the 1M-triangle soup, -4 % on the Cornell box).
// Everything sits in an anonymous namespace: each translation unit has its own copy.
#pragma once
#include "pt_internal.h"
#include "pt_math.h"
#include "pair_leaf.h"

#include <hip/hip_ext.h>

#ifndef PT_TB_DEFINED
#define PT_TB_DEFINED
namespace {
constexpr int TB = 256;                     // threads per block of every kernel of the library
constexpr uint32_t SENTINEL = 0xFFFFFFFFu;  // "no child" / "no node" in the BVH4 child words
}  // namespace
#endif

namespace {

// ---- extend: closest hit for every queued ray (traceRayEXT, raygen.rgen:63-75) ---------------
// Persistent grid (gridDim = CUs x resident blocks); each block walks 256-ray chunks of the
// dense queue, one ray per lane, over the BVH4 (128-B nodes = one L2 line per visit).
//  * children are visited nearest-first (4-element sorting network on the entry distances);
//  * stack entries are (child word, entry distance): a popped subtree that now starts behind the
//    best hit is dropped without touching memory (culling uses <= so equal-t candidates survive
//    for the deterministic lowest-primitive-id tie-break);
//  * short stack: the first LDS_STACK entries of every lane live in LDS as stack[level][thread]
//    (conflict-free), deeper entries spill to a per-thread column in HBM -- LDS use is constant
//    whatever the tree height, so occupancy is not capped by the scene;
//  * LDS_SCENE: nodes + triangles are staged into LDS once per persistent block and traversal
//    touches no HBM at all (scenes up to ~24 KB).
constexpr int LDS_STACK = 8;
// Nodes staged in LDS are spaced 144 B instead of 128 B: lanes of a wave sit on DIFFERENT nodes but read
// the SAME field of them, and with a 128-B stride (a multiple of the bank cycle) those 16-B reads all fall
// on the same 4 banks -- an n-way conflict for n distinct nodes.  144 B = 36 banks shifts consecutive
// nodes by 4 banks, so 8 nodes tile the 32 banks exactly (measured: +0.6 % on C2, within noise on C4).
constexpr uint32_t LDS_NODE_F4 = 9;  // float4 per LDS node (8 used)
// Stack entries are one 64-bit word (child word | entry distance << 32) and the LDS part is addressed
// through an LDS-typed pointer: with generic pointers the compiler merges the LDS and the spill
// access into FLAT loads/stores of the two halves (seen in the ISA), which cost VMEM issue and latency.
typedef __attribute__((address_space(3))) unsigned long long lds_u64;
typedef __attribute__((address_space(3))) uint32_t lds_u32;
__device__ __forceinline__ unsigned long long stack_entry(uint32_t w, float t)
{
    return (unsigned long long)w | ((unsigned long long)__float_as_uint(t) << 32);
}

// Slab test of the 4 children of a BVH4 node with the near/far planes picked by the ray's direction
// signs THROUGH THE LOAD ADDRESS (ax/ay/az = 48 bytes when the direction component is negative: the near
// plane of x is then float4 3 instead of 0, its far plane 0 instead of 3 -- one add or sub per load), so no
// per-child min/max of the two plane distances is needed.  Conservative like ptm::box_test.
#define PT_F4(P) (*reinterpret_cast<const float4 *>(P))
#define PT_NODE_LOAD(ND)                                                                                 \
    const char *nb_ = reinterpret_cast<const char *>(ND);                                                \
    const float4 nx = PT_F4(nb_ + ax), fx = PT_F4(nb_ - ax + 48), ny = PT_F4(nb_ + ay + 16),             \
                 fy = PT_F4(nb_ - ay + 64), nz = PT_F4(nb_ + az + 32), fz = PT_F4(nb_ - az + 80),        \
                 cw = PT_F4(nb_ + 96);
// Plane distances are ONE fma each, n * inv + (-org * inv), instead of (n - org) * inv: 6 VALU
// instructions less per child.  The rounding of the folded origin term and of the scaled far-plane
// reciprocal (absolute error <= 2^-22 |org*inv| in total) is covered by moving the origin term
// 2^-21 |org*inv| DOWN for the near planes and UP for the far planes (slab_setup below), so tn stays a lower
// and tf an upper bound.  The relative errors (v_rcp_f32's 1 ulp, the fma's rounding) are covered by a
// factor 1 + 4e-7 on the far distances, folded into the far planes' reciprocal and origin term (invf, of)
// so it costs nothing per node; a negative far distance only gets more negative, and such a box is behind
// the ray anyway.  Box tests are not part of the numerical contract -- they only have to never reject a
// box that holds a hit.
// (max_raw/min_raw: fmaxf/fminf on a kernel argument or a loop-carried value make the compiler re-quiet
// that operand with a v_max x,x in every iteration; the instruction itself already has maxNum semantics)
#define PT_SLAB4(T, C)                                                                                           \
    {                                                                                                            \
        const float tn = fmaxf(fmaxf(__builtin_fmaf(nx.C, inv.x, on.x), __builtin_fmaf(ny.C, inv.y, on.y)),      \
                               max_raw_s(__builtin_fmaf(nz.C, inv.z, on.z), tmin));                              \
        const float tf = fminf(fminf(__builtin_fmaf(fx.C, invf.x, of.x), __builtin_fmaf(fy.C, invf.y, of.y)),    \
                               min_raw(__builtin_fmaf(fz.C, invf.z, of.z), best_t));                             \
        T = tn <= tf ? tn : INF;                                                                                 \
    }
__device__ __forceinline__ float max_raw_s(float a, float uniform_b)
{
    float r;
    asm("v_max_f32_e32 %0, %1, %2" : "=v"(r) : "s"(uniform_b), "v"(a));
    return r;
}
__device__ __forceinline__ float min_raw(float a, float b)
{
    float r;
    asm("v_min_f32_e32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// ONE source of the compact node step of the scenes that live in LDS (k_extend_lds7 / _lds7p and k_fused instantiate it with their block
// sizes): the BVH4 node's planes through the direction-sign offsets (PT_NODE_LOAD), four slab tests (PT_SLAB4), the children ordered as
// one-dword keys -- entry distance truncated to its top 18 bits | 14-bit child word; formed BEFORE the sort they order like the distances
// (non-negative floats compare like their bit patterns; a miss is +inf | word, above every hit), so a compare-exchange is v_min_u32 +
// v_max_u32 instead of a compare and four selects: 14 VALU for the network instead of 25, and the pushes store the key as it is.
// Children closer together than 2^-9 of their distance may swap places -- the visit order is not part of the result (closest t, lowest
// primitive id).  The host runs these kernels for tmin > 0 only (entry distances >= tmin: no -0, whose bit pattern would sort last).
// -> the nearest child's code, or what `pop` returns when the ray misses all four.  STRIDE: threads per block (the stack is [level][thread]).
template <int STRIDE, class Pop>
__device__ __forceinline__ uint32_t compact_node_step(const float4 *wide, uint32_t cur, const ptm::f3 &inv, const ptm::f3 &invf, const ptm::f3 &on,
                                                      const ptm::f3 &of, uint32_t ax, uint32_t ay, uint32_t az, float tmin, float best_t,
                                                      lds_u32 *my_stack32, int &sp, Pop &&pop)
{
    const float INF = __builtin_inff();
    float t0, t1, t2, t3;
    // (a 24-bit multiply-add forms the node's LDS address: the child codes are below 2^14)
    const float4 *nd = reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(wide) + __umul24(cur, 16u * LDS_NODE_F4));
    PT_NODE_LOAD(nd)
    const uint32_t w0 = __float_as_uint(cw.x), w1 = __float_as_uint(cw.y), w2 = __float_as_uint(cw.z), w3 = __float_as_uint(cw.w);
    PT_SLAB4(t0, x)
    PT_SLAB4(t1, y)
    PT_SLAB4(t2, z)
    PT_SLAB4(t3, w)
    uint32_t k0 = (__float_as_uint(t0) & 0xFFFFC000u) | w0, k1 = (__float_as_uint(t1) & 0xFFFFC000u) | w1,
             k2 = (__float_as_uint(t2) & 0xFFFFC000u) | w2, k3 = (__float_as_uint(t3) & 0xFFFFC000u) | w3;
#define PT_KSWAP(A, B) { const uint32_t lo_ = min(A, B), hi_ = max(A, B); A = lo_; B = hi_; }
    PT_KSWAP(k0, k1)
    PT_KSWAP(k2, k3)
    PT_KSWAP(k0, k2)
    PT_KSWAP(k1, k3)
    PT_KSWAP(k1, k2)
#undef PT_KSWAP
    constexpr uint32_t KINF = 0x7F800000u;
    if (k3 < KINF) { my_stack32[sp * STRIDE] = k3; sp++; }  // farthest first, so the nearest pending pops first
    if (k2 < KINF) { my_stack32[sp * STRIDE] = k2; sp++; }
    if (k1 < KINF) { my_stack32[sp * STRIDE] = k1; sp++; }
    return k0 < KINF ? (k0 & 0x3FFFu) : pop();
}
__device__ __forceinline__ void slab_setup(const ptm::f3 org, const ptm::f3 inv, ptm::f3 &invf, ptm::f3 &on, ptm::f3 &of)
{
    const float ox = -(org.x * inv.x), oy = -(org.y * inv.y), oz = -(org.z * inv.z);
    const float px = fabsf(ox) * 0x1p-21f, py = fabsf(oy) * 0x1p-21f, pz = fabsf(oz) * 0x1p-21f;
    on = { ox - px, oy - py, oz - pz };
    of = { (ox + px) * 1.0000004f, (oy + py) * 1.0000004f, (oz + pz) * 1.0000004f };
    invf = { inv.x * 1.0000004f, inv.y * 1.0000004f, inv.z * 1.0000004f };
}
// ---- 64-B nodes (scene walked in HBM/L2): the planes are fp16 of box coordinates normalised to the scene box
// (lbvh_build.hip k_wide_half, rounded outwards).  v_fma_mix_f32 reads the half straight out of the loaded
// dword, so the slab arithmetic costs exactly what it costs with fp32 planes; the ray is normalised the same
// way at refill (org' = (org - c) * rs, inv' = inv * s), which leaves every distance t unchanged.
struct NormBox { float cx, cy, cz, sx, sy, sz, rsx, rsy, rsz; };
#define PT_MIXH(DST, REG, HI, INVC, ONC) \
    asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[" #HI ",0,0] op_sel_hi:[1,0,0]" : "=v"(DST) : "v"(REG), "v"(INVC), "v"(ONC))
#define PT_SLAB4H(T, REGC, HI)                                                                      \
    {                                                                                               \
        float nxv, nyv, nzv, fxv, fyv, fzv;                                                         \
        PT_MIXH(nxv, hnx.REGC, HI, inv.x, on.x); PT_MIXH(nyv, hny.REGC, HI, inv.y, on.y);           \
        PT_MIXH(nzv, hnz.REGC, HI, inv.z, on.z); PT_MIXH(fxv, hfx.REGC, HI, invf.x, of.x);          \
        PT_MIXH(fyv, hfy.REGC, HI, invf.y, of.y); PT_MIXH(fzv, hfz.REGC, HI, invf.z, of.z);         \
        const float tn = fmaxf(fmaxf(nxv, nyv), max_raw_s(nzv, tmin));                              \
        const float tf = fminf(fminf(fxv, fyv), min_raw(fzv, best_t));                              \
        T = tn <= tf ? tn : INF;                                                                    \
    }
// m ? a : b for an all-ones / all-zeros mask, as one v_bfi_b32: a run of v_cndmask_b32 on VCC issues at ~23 cycles each on
// gfx950, which made the two-level fp16 kernel stall on issue 3.7 times as often as its predecessor
#ifndef PT_HBM_SELECT_BFI
#define PT_HBM_SELECT_BFI 1
#endif
#if PT_HBM_SELECT_BFI
#define PT_BFI(M, A, B) (((M) & (A)) | (~(M) & (B)))
#else
#define PT_BFI(M, A, B) ((M) ? (A) : (B))
#endif
#ifndef PT_EXTEND_WAVES
#define PT_EXTEND_WAVES 7  // min waves per SIMD asked of the compiler for the no-spill LDS variant: 72 VGPRs instead of 76, no spills (8: 13 spilled, -13 %)
#endif
constexpr int REFILL_MIN_IDLE = 16;  // default number of idle lanes before the wave pulls new rays

// Persistent threads with dynamic ray fetch (Aila & Laine 2009, re-tiled for wave64): a lane whose
// ray is finished does not wait for the slowest ray of its wave; once >= REFILL_MIN_IDLE lanes are
// idle they take the next rays of the wave's OWN sequence of 64-ray chunks (chunk w, w + #waves,
// w + 2*#waves, ... of the dense queue; ballot + popcount give the per-lane offsets).  The cursor
// is wave-private, so there is no shared dequeue word at all: a single device-scope head saturates
// at ~88 atomics/us on this chip, which short Cornell traversals (3.6 nodes/ray) exceed 3x over.
// Incoherent rays otherwise leave a wave64 at 15-20 % lane utilisation (measured: 6x more VALU
// instructions per wave than per average lane).
template <bool LDS_SCENE, bool COUNT, bool SPILL, bool PAIRS = false, bool REC64 = false>
__device__ __forceinline__ void extend_body(const float4 *__restrict__ g_wide, const uint2 *__restrict__ g_wide16,
                                               NormBox nb, const float4 *__restrict__ g_tri4,
                                               uint32_t n_wide, uint32_t n_tris, const float4 *__restrict__ rayA,
                                               const float2 *__restrict__ rayB, float4 *__restrict__ hit,
                                               const uint32_t *__restrict__ count_in, uint32_t *count_zero,
                                               unsigned long long *stats, uint2 *__restrict__ spill,
                                               uint32_t spill_stride, int refill_vote, float tmin, float tmax,
                                               int lds_stack, int raw_hit, const uint32_t *__restrict__ perm,
                                               const float *__restrict__ ray_tmax, const float4 *__restrict__ g_rec64 = nullptr)
{
    // refill_vote: idle lanes before a refill | (scenes in HBM) tri_enter << 8, the lanes that wait with a leaf before a leaf
    // step runs even against a majority of descending lanes (pt_tuning.tri_enter; 0 = majority only)
    const int refill_min_idle = refill_vote & 0xFF, tri_enter = (refill_vote >> 8) & 0xFF;
    // ray_tmax (shadow rays of the NEE pipeline): a per-ray upper bound instead of `tmax`, and ANY hit below it ends
    // the walk (the record then only says hit or miss)
    // Scenes in HBM (deep trees, incoherent rays): inside the classic while-while loop the node phase ran
    // at 18 % lane occupancy on the 1M-triangle soup (device counters) -- lanes that already hold a leaf wait
    // for the last lane to finish descending.  There the wave instead takes ONE step per iteration, of the
    // kind (node or leaf) that more of its lanes are waiting for (node steps counted double until the leaves went to 1 triangle): node occupancy
    // 36 %, triangle steps 36 %, C5 +11 %.  The LDS-resident Cornell box loses 5 % to the extra votes, so it
    // keeps the inner loop.
    constexpr bool VOTE = !LDS_SCENE;
    constexpr int VOTE_NODE_NUM = 1, VOTE_NODE_DEN = 1;  // plain majority (with 1-triangle leaves: 2:1 -1 %, 3:1 -3 %, 1:2 .. 3:4 equal)
    // COMPACT (scene in LDS and its exact stack bound fits: the Cornell box): child words are re-coded to 14 bits
    // when the nodes are staged (leaf: bit 13 | (count-1) << 11 | first; inner: node index; done: 0x3FFF) and a
    // stack entry is ONE dword, the entry distance truncated to its top 18 bits above the child word
    // (sign, exponent, 9 mantissa bits: rounds a non-negative distance DOWN, so the pop test stays conservative).
    // Half the stack bytes in LDS: 15 KB instead of 25 KB per block for the Cornell box, which lifts the LDS cap
    // on resident blocks from 6 to 10 per CU.  The host only picks it for tmin >= 0.
    // PAIRS (a COMPACT variant): every leaf holds ONE primitive -- a triangle, or the two halves (v0,v1,v2),(v0,v2,v3)
    // of a quad at consecutive positions (bvh4_sah_device.hip, pair_with_next).  The leaf step is then one straight piece of
    // code for all lanes that hold a leaf (no per-lane triangle count to loop over: that loop ran at 31 % lane
    // occupancy on the Cornell box), and the second half re-uses the first one's sheared v0, v2 and the products of
    // their shared edge function: 44 instead of 60 VALU for the two edge tests, bit for bit the per-triangle results.
    constexpr bool COMPACT = LDS_SCENE && !SPILL;
    static_assert(!PAIRS || COMPACT, "pair leaves are implemented for the compact LDS kernel");
    // REC64 (scenes walked out of L2 / MALL / HBM, vote-scheduled step): the three vertices come from the 64-B per-triangle
    // record k_shade gathers anyway ({v0, n.x} {v1, n.y} {v2, n.z} {brdf, emits}, `g_rec64` = pt_scene::d_shade64) instead
    // of the 48-B record of tri4.  Beyond L2 the chip charges a divergent access per distinct 128-B line
    // (scripts/ubench/gather_rate.hip): two of every eight 48-B records straddle a line, a 64-B record never does, the
    // shading pass of the same round finds the line of the winning triangle already fetched, and tri4 drops out of the
    // working set (its primitive ids are read only when two hits have exactly the same t).
    static_assert(!REC64 || (!LDS_SCENE && SPILL), "64-B triangle records are the vote-scheduled HBM kernel's");
    constexpr uint32_t LEAF_BIT = COMPACT ? 0x2000u : PT_LEAF;
    constexpr uint32_t DONE = COMPACT ? 0x3FFFu : SENTINEL;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    uint2 *stack = reinterpret_cast<uint2 *>(smem);  // [LDS_STACK][TB]
    const float4 *wide = g_wide;
    const float4 *tri4 = g_tri4;
    if (LDS_SCENE) {
        float4 *s_wide = reinterpret_cast<float4 *>(smem + (size_t)lds_stack * TB * (COMPACT ? sizeof(uint32_t) : sizeof(uint2)));
        float4 *s_tri = s_wide + LDS_NODE_F4 * (size_t)n_wide;
        for (uint32_t i = threadIdx.x; i < 8 * n_wide; i += TB) {
            float4 v = g_wide[i];
            if (COMPACT && (i & 7u) == 6u) {  // the four child words
                auto cw = [](float f) {
                    const uint32_t w = __float_as_uint(f);
                    const uint32_t c = (w & PT_LEAF) ? (0x2000u | (((w >> 28) & 3u) << 11) | (w & 0x7FFu)) : (w & 0x1FFFu);
                    return __uint_as_float(w == SENTINEL ? 0x3FFFu : c);
                };
                v = make_float4(cw(v.x), cw(v.y), cw(v.z), cw(v.w));
            }
            s_wide[(i >> 3) * LDS_NODE_F4 + (i & 7u)] = v;
        }
        // three copies of the triangles with components permuted to (kx,ky,kz) for kz = 0,1,2:
        // the triangle test then needs no per-lane component selects (ptm::tri_test_perm)
        for (uint32_t i = threadIdx.x; i < 3 * n_tris; i += TB) {
            const float4 v = g_tri4[i];
            s_tri[i] = make_float4(v.y, v.z, v.x, v.w);                    // kz = 0: (kx,ky,kz) = (1,2,0)
            s_tri[3 * n_tris + i] = make_float4(v.z, v.x, v.y, v.w);       // kz = 1: (2,0,1)
            s_tri[6 * n_tris + i] = v;                                     // kz = 2: (0,1,2)
        }
        __syncthreads();
        wide = s_wide;
        tri4 = s_tri;
    }
    const uint32_t n = *count_in;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        if (count_zero) *count_zero = 0u;  // the queue the coming shade pass appends to
        if (stats) atomicAdd(stats, (unsigned long long)n);  // exact ray count
    }
    lds_u64 *my_stack = (lds_u64 *)reinterpret_cast<unsigned long long *>(stack) + threadIdx.x;
    lds_u32 *my_stack32 = (lds_u32 *)reinterpret_cast<uint32_t *>(stack) + threadIdx.x;
    unsigned long long *my_spill = reinterpret_cast<unsigned long long *>(spill) + (size_t)blockIdx.x * TB + threadIdx.x;
    const float INF = __builtin_inff();
    const int lane = threadIdx.x & 63;
    const unsigned long long lt = (1ull << lane) - 1ull;  // (v_mbcnt instead of this mask: C5 -5 %, measured twice)

    bool have = false, exhausted = false;
    uint32_t q = 0;
    // wave-private ray sequence: virtual index v -> queue position (v/64)*wave_stride + wave_base + v%64
    const uint32_t wave_base = (blockIdx.x * (TB / 64) + (threadIdx.x >> 6)) * 64u;
    const uint32_t wave_stride = gridDim.x * TB;
    uint32_t cursor = 0;
    ptm::f3 inv{}, invf{}, on{}, of{}, orgp{};  // slab_setup: near/far reciprocals and folded origin terms
    ptm::RayPre pre{};
    uint32_t ax = 0, ay = 0, az = 0;  // 48 where the direction component is negative (PT_NODE_LOAD)
    uint32_t tri_base = 0;           // LDS_SCENE: start of the triangle copy for this ray's kz
    float best_t = tmax, best_V = 0.f, best_W = 0.f, best_det = 1.f;
    uint32_t best_pos = PT_MISS, best_prim = PT_MISS;
    uint32_t cur = DONE;
    int sp = 0;
    unsigned long long c_nodes = 0, c_tris = 0, c_node_steps = 0, c_tri_steps = 0;
    unsigned long long c_refills = 0, c_pops = 0, c_hit_blocks = 0, c_finishes = 0, c_iters = 0;  // wave executions of the other blocks
    unsigned long long c_leaf_lanes = 0, c_pop_lanes = 0, c_hit_lanes = 0;  // lanes inside the leaf steps / pop iterations / divide blocks
    // one lane per wave counts a block the wave executes (exec mask of the moment)
#define PT_COUNT_WAVE(C) \
    if (COUNT && lane == __ffsll((long long)__ballot(1)) - 1) (C)++

    auto push = [&](uint32_t w, float t) {
        if (COMPACT) {
            my_stack32[sp * TB] = (__float_as_uint(t) & 0xFFFFC000u) | w;
            sp++;
            return;
        }
        const unsigned long long e = stack_entry(w, t);
        if (!SPILL || sp < lds_stack) my_stack[sp * TB] = e;  // !SPILL: the host proved lds_stack entries suffice
        else my_spill[(size_t)(sp - lds_stack) * spill_stride] = e;
        sp++;
    };
    auto pop = [&]() -> uint32_t {  // next subtree that can still contain the closest hit
        if (COMPACT) {
            while (sp > 0) {
                PT_COUNT_WAVE(c_pops);
                if (COUNT) c_pop_lanes++;
                sp--;
                const uint32_t e = my_stack32[sp * TB];
                if (__uint_as_float(e & 0xFFFFC000u) <= best_t) return e & 0x3FFFu;
            }
            return DONE;
        }
        while (sp > 0) {
            PT_COUNT_WAVE(c_pops);
            if (COUNT) c_pop_lanes++;
            sp--;
            unsigned long long e;
            if (!SPILL || sp < lds_stack) e = my_stack[sp * TB];
            else e = my_spill[(size_t)(sp - lds_stack) * spill_stride];
            if (__uint_as_float((uint32_t)(e >> 32)) <= best_t) return (uint32_t)e;
        }
        return DONE;
    };

    for (;;) {
        // ---- refill idle lanes from the queue head
        PT_COUNT_WAVE(c_iters);
        const unsigned long long idle = __ballot(!have);
        const int n_idle = __popcll(idle);
        if (!exhausted && n_idle >= refill_min_idle) {
            if (!have) {
                // rank among the idle lanes: the pair-leaf kernel counts with v_mbcnt (the 64-bit prefix mask `lt` is two more
                // registers live through the whole kernel: 70 instead of 72, same speed); the others keep the mask (v_mbcnt in the
                // HBM kernel: C5 -5 %, measured twice; in k_extend_lds7: two spills -- different schedules, nothing else)
                const uint32_t v = cursor + (PAIRS ? __builtin_amdgcn_mbcnt_hi((uint32_t)(idle >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)idle, 0u))
                                                       : (uint32_t)__popcll(idle & lt));
                const uint32_t qq = (v >> 6) * wave_stride + wave_base + (v & 63u);
                if (qq < n) {
                    PT_COUNT_WAVE(c_refills);
                    q = (!LDS_SCENE && perm) ? perm[qq] : qq;  // ray_sort.hip: the queue is walked in (cell, octant) order
                    const float4 ra = ptm::ld_stream<false>(rayA + q);
                    const float2 rb = ptm::ld_stream<false>(rayB + q);
                    const ptm::f3 org = { ra.x, ra.y, ra.z };
                    const ptm::f3 dir = { ra.w, rb.x, rb.y };
                    pre = ptm::ray_setup<!(LDS_SCENE && !PAIRS)>(org, dir);  // (k_extend_lds7: pt_math.h)
                    inv = { ptm::safe_inv(dir.x), ptm::safe_inv(dir.y), ptm::safe_inv(dir.z) };
                    if (LDS_SCENE) {
                        slab_setup(org, inv, invf, on, of);
                    } else {  // fp16 nodes live in the normalised scene box
                        const ptm::f3 orgn = { (org.x - nb.cx) * nb.rsx, (org.y - nb.cy) * nb.rsy, (org.z - nb.cz) * nb.rsz };
                        inv = { inv.x * nb.sx, inv.y * nb.sy, inv.z * nb.sz };
                        slab_setup(orgn, inv, invf, on, of);
                    }
                    // byte offset of the near planes inside a node: 3 planes of 16 B (fp32 node) or 8 B (fp16 node)
                    ax = inv.x < 0.f ? (LDS_SCENE ? 48u : 24u) : 0u;
                    ay = inv.y < 0.f ? (LDS_SCENE ? 48u : 24u) : 0u;
                    az = inv.z < 0.f ? (LDS_SCENE ? 48u : 24u) : 0u;
                    if (LDS_SCENE) {
                        tri_base = (uint32_t)pre.kz * 3u * n_tris;
                        orgp = { ptm::sel3(pre.kz, org.y, org.z, org.x), ptm::sel3(pre.kz, org.z, org.x, org.y),
                                 ptm::sel3(pre.kz, org.x, org.y, org.z) };
                    }
                    best_t = ray_tmax ? ray_tmax[q] : tmax; best_V = 0.f; best_W = 0.f; best_det = 1.f;
                    best_pos = PT_MISS; best_prim = PT_MISS;
                    cur = 0u;  // wide root
                    sp = 0;
                    have = true;
                }
            }
            cursor += (uint32_t)n_idle;
            exhausted = (cursor >> 6) * wave_stride + wave_base >= n;  // chunk of the next refill starts past the end
        }
        if (__ballot(have) == 0ull) break;

        // ---- node phase: every lane descends until it holds a leaf (or runs out of nodes)
        // VOTE: one step per outer iteration, of the kind (node / leaf) that more lanes are waiting for
        bool do_leaf = true;
        bool do_node = have && !(cur & LEAF_BIT);
        const int n_have = __popcll(__ballot(have));
        if (VOTE) {
            const bool want_leaf = have && (cur & LEAF_BIT) && cur != DONE;
            const int nn = __popcll(__ballot(do_node)), nl = __popcll(__ballot(want_leaf));
            const bool node_turn = nn * VOTE_NODE_NUM >= nl * VOTE_NODE_DEN && !(tri_enter && nl >= tri_enter);
            do_node = do_node && node_turn;
            do_leaf = !node_turn;
        }
        while (do_node) {
            if (COUNT) {
                c_nodes++;
                if (lane == __ffsll((long long)__ballot(1)) - 1) c_node_steps++;  // one lane per wave step
            }
            if constexpr (LDS_SCENE && COMPACT) {
                cur = compact_node_step<TB>(wide, cur, inv, invf, on, of, ax, ay, az, tmin, best_t, my_stack32, sp, pop);
            } else {
            float t0, t1, t2, t3;
            uint32_t w0, w1, w2, w3;
            if (LDS_SCENE) {
                // (a 24-bit multiply-add forms the node's LDS address: the child codes are below 2^14)
                const float4 *nd = reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(wide) + __umul24(cur, 16u * LDS_NODE_F4));
                PT_NODE_LOAD(nd)
                w0 = __float_as_uint(cw.x); w1 = __float_as_uint(cw.y); w2 = __float_as_uint(cw.z); w3 = __float_as_uint(cw.w);
                PT_SLAB4(t0, x)
                PT_SLAB4(t1, y)
                PT_SLAB4(t2, z)
                PT_SLAB4(t3, w)
            } else {
                const char *nb_ = reinterpret_cast<const char *>(g_wide16) + 64 * (size_t)cur;
                // The whole 64-B node as four 16-B loads and the near/far planes picked by twelve selects -- not, as in
                // the LDS variant, six 8-B loads addressed through the direction signs plus the child words: lanes sit
                // on different nodes, so every load instruction is one tag look-up per active lane in the vector L1,
                // and this kernel is bound by memory requests in flight, not by bytes or VALU (C5: 7 -> 4 look-ups
                // per node visit, extend -15 %, 1883 -> 2104 Mrays/s).
                const uint4 q0 = *reinterpret_cast<const uint4 *>(nb_), q1 = *reinterpret_cast<const uint4 *>(nb_ + 16),
                            q2 = *reinterpret_cast<const uint4 *>(nb_ + 32);
                const uint4 cw = *reinterpret_cast<const uint4 *>(nb_ + 48);
                // lo planes: q0.xy q0.zw q1.xy, hi planes: q1.zw q2.xy q2.zw; near / far by bit-field insert (PT_BFI)
                const uint32_t mx = ax ? 0xFFFFFFFFu : 0u, my = ay ? 0xFFFFFFFFu : 0u, mz = az ? 0xFFFFFFFFu : 0u;
                const uint2 hnx = { PT_BFI(mx, q1.z, q0.x), PT_BFI(mx, q1.w, q0.y) }, hfx = { PT_BFI(mx, q0.x, q1.z), PT_BFI(mx, q0.y, q1.w) };
                const uint2 hny = { PT_BFI(my, q2.x, q0.z), PT_BFI(my, q2.y, q0.w) }, hfy = { PT_BFI(my, q0.z, q2.x), PT_BFI(my, q0.w, q2.y) };
                const uint2 hnz = { PT_BFI(mz, q2.z, q1.x), PT_BFI(mz, q2.w, q1.y) }, hfz = { PT_BFI(mz, q1.x, q2.z), PT_BFI(mz, q1.y, q2.w) };
                w0 = cw.x; w1 = cw.y; w2 = cw.z; w3 = cw.w;
                PT_SLAB4H(t0, x, 0)
                PT_SLAB4H(t1, x, 1)
                PT_SLAB4H(t2, y, 0)
                PT_SLAB4H(t3, y, 1)
            }
            {
#define PT_CSWAP(TA, WA, TB_, WB)                            \
    {                                                        \
        const bool sw = TB_ < TA;                            \
        const float ta = sw ? TB_ : TA, tb = sw ? TA : TB_;  \
        const uint32_t wa = sw ? WB : WA, wb = sw ? WA : WB; \
        TA = ta; TB_ = tb; WA = wa; WB = wb;                 \
    }
            PT_CSWAP(t0, w0, t1, w1)
            PT_CSWAP(t2, w2, t3, w3)
            PT_CSWAP(t0, w0, t2, w2)
            PT_CSWAP(t1, w1, t3, w3)
            PT_CSWAP(t1, w1, t2, w2)
#undef PT_CSWAP
            if (t3 < INF) push(w3, t3);  // farthest first, so the nearest pending pops first
            if (t2 < INF) push(w2, t2);
            if (t1 < INF) push(w1, t1);
            cur = t0 < INF ? w0 : pop();
            }
            }
            do_node = !VOTE && !(cur & LEAF_BIT);
            if (!VOTE) {
                // fewer than 1/6 of the wave's rays still descending while the rest waits with a leaf: let the
                // leaves go first, the stragglers resume in the next round of the outer loop (node-step lane
                // occupancy 40 % -> 55 %, triangle steps 34 % -> 31 %, C2 +3 %; 1/4: +2.5 %, 1/12: +2.5 %)
                const int n_cont = __popcll(__ballot(do_node));
                if (n_cont * 6 < n_have) break;
            }
        }
        // ---- leaf phase
        if (have) {
            if (PAIRS) {
                if (cur != DONE && (cur & LEAF_BIT)) {
                    const uint32_t first = cur & 0x7FFu;
                    const bool two = ((cur >> 11) & 3u) != 0u;  // count - 1: a fan pair at positions first, first + 1
                    if (COUNT) { c_tris += two ? 2u : 1u; c_leaf_lanes++; }
                    PT_COUNT_WAVE(c_tri_steps);
                    ptl::pair_leaf_test(tri4, (size_t)tri_base + 3 * (size_t)first, two, first, pre, orgp, tmin, tmax,
                                        [&](float t, float V, float W, float det, uint32_t pos, uint32_t) {
                                            if (ptl::closer_single_level(tri4, tri_base, t, V, W, det, pos, best_t, best_V, best_W, best_det, best_pos) && ray_tmax)
                                                sp = 0;  // any hit will do: nothing pending any more
                                        },
                                        [&] {
                                            PT_COUNT_WAVE(c_hit_blocks);
                                            if (COUNT) c_hit_lanes++;
                                        });
                    cur = pop();
                }
            } else
            if (cur != DONE && (cur & LEAF_BIT) && (!VOTE || do_leaf)) {
                const uint32_t first = COMPACT ? (cur & 0x7FFu) : (cur & 0x0FFFFFFFu);
                const uint32_t cnt = (COMPACT ? ((cur >> 11) & 3u) : ((cur >> 28) & 7u)) + 1u;
                if (COUNT) c_tris += cnt;
                for (uint32_t k = 0; k < cnt; k++) {
                    if (COUNT && lane == __ffsll((long long)__ballot(1)) - 1) c_tri_steps++;
                    if (COUNT) c_leaf_lanes++;
                    const uint32_t pos = first + k;
                    const size_t ti = LDS_SCENE ? (size_t)tri_base + 3 * (size_t)pos : (REC64 ? 4 : 3) * (size_t)pos;
                    const float4 *tp = REC64 ? g_rec64 : tri4;
                    const float4 a = tp[ti + 0], b = tp[ti + 1], c = tp[ti + 2];
                    float t, V, W, det;
                    bool divided = false;
                    const bool th = LDS_SCENE
                        ? ptm::tri_test_perm(pre, orgp, { a.x, a.y, a.z }, { b.x, b.y, b.z }, { c.x, c.y, c.z }, tmin, tmax, t, V, W, det, COUNT ? &divided : nullptr)
                        : ptm::tri_test(pre, { a.x, a.y, a.z }, { b.x, b.y, b.z }, { c.x, c.y, c.z }, tmin, tmax, t, V, W, det, COUNT ? &divided : nullptr);
                    if (COUNT && divided) { PT_COUNT_WAVE(c_hit_blocks); c_hit_lanes++; }
                    if (th) {
                        // closest t; equal t -> lowest gl_PrimitiveID (the OBJ has coincident quads)
                        if constexpr (REC64) {  // .w of a 64-B record is the normal: the ids of the two rivals come from tri4, on ties only
                            bool closer = t < best_t;
                            if (!closer && t == best_t)
                                closer = best_pos == PT_MISS ||
                                         __float_as_uint(g_tri4[3 * (size_t)pos].w) < __float_as_uint(g_tri4[3 * (size_t)best_pos].w);
                            if (closer) {
                                best_t = t; best_V = V; best_W = W; best_det = det; best_pos = pos;
                                if (ray_tmax) sp = 0;
                            }
                        } else {
                        const uint32_t prim = __float_as_uint(a.w);
                        if (t < best_t || (t == best_t && prim < best_prim)) {
                            best_t = t; best_V = V; best_W = W; best_det = det; best_pos = pos; best_prim = prim;
                            if (ray_tmax) sp = 0;
                        }
                        }
                    }
                }
                cur = pop();
            }
            if (cur == DONE) {  // traversal finished: emit the hit record, the lane becomes idle
                PT_COUNT_WAVE(c_finishes);
                const bool miss = best_pos == PT_MISS;
                // raw_hit (render path): (V, W, det) go out undivided and k_shade takes the two quotients at
                // full lane occupancy; here they would run once per finishing lane group
                ptm::st_stream<false>(hit + q, raw_hit ? make_float4(__uint_as_float(best_pos), best_V, best_W, best_det)
                                 : make_float4(__uint_as_float(best_pos), miss ? 0.f : best_t,
                                               miss ? 0.f : ptm::fdiv(best_V, best_det), miss ? 0.f : ptm::fdiv(best_W, best_det)));
                have = false;
            }
        }
    }
#undef PT_COUNT_WAVE
    if (COUNT) {
        for (int o = 32; o > 0; o >>= 1) {
            c_nodes += __shfl_xor(c_nodes, o, 64);
            c_tris += __shfl_xor(c_tris, o, 64);
            c_node_steps += __shfl_xor(c_node_steps, o, 64);
            c_tri_steps += __shfl_xor(c_tri_steps, o, 64);
            c_refills += __shfl_xor(c_refills, o, 64);
            c_pops += __shfl_xor(c_pops, o, 64);
            c_hit_blocks += __shfl_xor(c_hit_blocks, o, 64);
            c_finishes += __shfl_xor(c_finishes, o, 64);
            c_iters += __shfl_xor(c_iters, o, 64);
            c_leaf_lanes += __shfl_xor(c_leaf_lanes, o, 64);
            c_pop_lanes += __shfl_xor(c_pop_lanes, o, 64);
            c_hit_lanes += __shfl_xor(c_hit_lanes, o, 64);
        }
        if (lane == 0 && stats) {
            atomicAdd(stats + 2, c_nodes);
            atomicAdd(stats + 3, c_tris);
            atomicAdd(stats + 4, c_node_steps);
            atomicAdd(stats + 5, c_tri_steps);
            atomicAdd(stats + 8, c_refills);
            atomicAdd(stats + 9, c_pops);
            atomicAdd(stats + 10, c_hit_blocks);
            atomicAdd(stats + 11, c_finishes);
            atomicAdd(stats + 12, c_iters);
            atomicAdd(stats + 13, c_leaf_lanes);
            atomicAdd(stats + 14, c_pop_lanes);
            atomicAdd(stats + 15, c_hit_lanes);
        }
    }
}

#define PT_EXTEND_PARAMS                                                                                    \
    const float4 *__restrict__ g_wide, const uint2 *__restrict__ g_wide16, NormBox nb,                \
        const float4 *__restrict__ g_tri4, uint32_t n_wide, uint32_t n_tris, const float4 *__restrict__ rayA,       \
        const float2 *__restrict__ rayB, float4 *__restrict__ hit, const uint32_t *__restrict__ count_in,           \
        uint32_t *count_zero, unsigned long long *stats, uint2 *__restrict__ spill, uint32_t spill_stride,          \
        int refill_min_idle, float tmin, float tmax, int lds_stack, int raw_hit, const uint32_t *__restrict__ perm,       \
        const float *__restrict__ ray_tmax, const float4 *__restrict__ g_rec64
#define PT_EXTEND_ARGS                                                                                               \
    g_wide, g_wide16, nb, g_tri4, n_wide, n_tris, rayA, rayB, hit, count_in, count_zero, stats, spill, spill_stride, \
        refill_min_idle, tmin, tmax, lds_stack, raw_hit, perm, ray_tmax, g_rec64
template <bool LDS_SCENE, bool COUNT, bool SPILL, bool PAIRS = false, bool REC64 = false>
__global__ __launch_bounds__(TB) void k_extend(PT_EXTEND_PARAMS)
{
    extend_body<LDS_SCENE, COUNT, SPILL, PAIRS, REC64>(PT_EXTEND_ARGS);
}
#ifndef PT_EXTEND_TEMPLATES_ONLY  // (fused.hip and extend_hbm.hip take the helpers and the templates, not these four kernels)
// The instantiation the Cornell box runs (scene in LDS, no spill path, one-dword stack entries) as its own kernel:
// asking for PT_EXTEND_WAVES waves per SIMD makes the compiler fit 72 VGPRs instead of 76; the other instantiations
// keep the plain launch bounds they were tuned with.
// (perm / ray_tmax as literal null pointers: the hot instantiations must not carry the shadow-ray branches -- with them
// as run-time arguments the 72-VGPR kernels spilled two registers and lost 8 %; shadow rays run the _sh twins)
#define PT_EXTEND_ARGS_PLAIN                                                                                         \
    g_wide, g_wide16, nb, g_tri4, n_wide, n_tris, rayA, rayB, hit, count_in, count_zero, stats, spill, spill_stride, \
        refill_min_idle, tmin, tmax, lds_stack, raw_hit, nullptr, nullptr
__global__ __launch_bounds__(TB, PT_EXTEND_WAVES) void k_extend_lds7(PT_EXTEND_PARAMS)
{
    extend_body<true, false, false>(PT_EXTEND_ARGS_PLAIN);
}
__global__ __launch_bounds__(TB, PT_EXTEND_WAVES) void k_extend_lds7_sh(PT_EXTEND_PARAMS)
{
    extend_body<true, false, false>(PT_EXTEND_ARGS);
}
// ... and the same over a BVH4 with one primitive (triangle or fan pair) per leaf: what the Cornell box runs by default
__global__ __launch_bounds__(TB, PT_EXTEND_WAVES) void k_extend_lds7p(PT_EXTEND_PARAMS)
{
    extend_body<true, false, false, true>(PT_EXTEND_ARGS_PLAIN);
}
__global__ __launch_bounds__(TB, PT_EXTEND_WAVES) void k_extend_lds7p_sh(PT_EXTEND_PARAMS)
{
    extend_body<true, false, false, true>(PT_EXTEND_ARGS);
}
#undef PT_EXTEND_ARGS_PLAIN
#endif  // PT_EXTEND_TEMPLATES_ONLY
#undef PT_EXTEND_PARAMS
#undef PT_EXTEND_ARGS

}  // namespace
