// extend8_kernel.h -- closest hit over the 8-wide tree (scenes walked out of L2 / MALL / HBM; lbvh_build.hip k_w8_*).
//
// Same contract as k_extend (extend_kernel.h): persistent threads, lane refill from a wave-private sequence of 64-ray
// chunks, the canonical watertight triangle test, closest t with ties to the lowest primitive id -- the hit records are
// bit-identical.  What differs is what a ray fetches and what it remembers:
//   * a node is 64 B with EIGHT children: the planes are bytes on the node's own grid (16-bit origin per axis on the
//     normalised scene box + q * 2^-e, lbvh_build.hip k_w8_emit), so two nodes share a 128-B line and the eight children of a
//     node are four lines: beyond L2 this chip charges a divergent load per distinct line, not per byte
//     (scripts/ubench/gather_rate.hip), and eight-wide nodes need a quarter fewer visits than four-wide ones.  (Round 2's
//     128-B node with fp16 planes fetched as many lines as the four-wide tree and lost; it is gone.)  The price is the
//     decode: per visit the node's origin and scales, the slab constants relative to that origin, 48 byte -> float
//     conversions;
//   * internal children are contiguous and a node's leaf triangles are contiguous, so what is pending of a node is
//     {child_base, mask of the internal children still to visit}: ONE 8-byte stack entry per visited node (with the
//     smallest entry distance of its hit children in the top 16 bits, so a popped group that lies behind the best hit
//     is dropped without a fetch; the bound of the children actually pending -- the two smallest distances, the slot in the key's low
//     bits -- saves 8 % / 14 % of the node visits of C5 / C5x and costs as many instructions as it saves: C5 -2 %, C5x +0.4 %,
//     profiles/r04an_ab_e8_rest_*.log) instead of one entry per child, and no sorting network: children sit in the slot of
//     their octant, "slot xor ray octant" in ascending order is roughly front to back;
//   * the triangles a node visit found (mask of hit leaf slots) are tested before the walk goes on; steps are
//     vote-scheduled like k_extend's (the wave runs the node code or the triangle code, whichever more lanes wait for).
#pragma once
#include "extend_kernel.h"

#ifndef PT_E8_BOUND6
#define PT_E8_BOUND6 1  // 0: the 6-wave instantiation, too, without the pending-group bound (C5 -1.1 %: see extend8_body)
#endif


namespace {

// SPILL = false: the tree's levels fit the LDS stack entries (every tree up to 8^12 leaves does with the default 12), so the
// address arithmetic of the HBM spill column leaves the push and the pop loop
// BOUND: a pending group carries the smallest entry distance of its node's hits and is dropped on the pop when that lies behind the best hit.
// A model of the walk (scripts/sim_bvh8_policies.py, which reproduces the kernel's node visits per ray to 0.2 %) says this bound NEVER culls
// -- it is nearly always the entry distance of the child visited first, whose subtree cannot hold a hit in front of it -- and the kernel
// without it visits exactly as many nodes (27.65 per ray on C5, 24.19 on C5x) with 22 VALU instructions less per node step.  Measured, same box,
// three rounds each: scenes walked out of HBM (C5x) +3.3 % at 6 waves and +2.8 % at 7; the cache-resident C5 -1.1 % at either
// (profiles/r04ao_*, r04ap_*).  So the instantiation for scenes beyond the Infinity Cache (WAVES = 7) goes without, the other keeps it:
// alone on the chip (one pipeline) the 6-wave kernel without the bound is 3.3 % faster on C5 as well, in the two-pipeline frame it is
// 0.9 % slower, and no refill / vote / stack setting changes that (PT_E8_BOUND6 = 0 builds it: profiles/r04ax_*, r04ay_*).
template <bool COUNT, bool SPILL, bool BOUND>
__device__ __forceinline__ void extend8_body(const uint4 *__restrict__ nodes8, NormBox nb, const float4 *__restrict__ tri4, const float4 *__restrict__ rec64,
                                             const float4 *__restrict__ rayA, const float2 *__restrict__ rayB,
                                             float4 *__restrict__ hit, const uint32_t *__restrict__ count_in,
                                             uint32_t *count_zero, unsigned long long *stats, uint2 *__restrict__ spill,
                                             uint32_t spill_stride, int refill_vote, float tmin, float tmax, int lds_stack,
                                             int raw_hit, const uint32_t *__restrict__ perm, const float *__restrict__ ray_tmax)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // refill_vote: idle lanes before a refill | tri_enter << 8 | tri_stay << 16 (pt_tuning; the vote below)
    const int refill_min_idle = refill_vote & 0xFF, tri_enter = (refill_vote >> 8) & 0xFF, tri_stay = (refill_vote >> 16) & 0xFF;
    __shared__ uint8_t s_perm[8 * 256];  // [ray octant][slot mask] -> priority mask (the node step below)
    for (uint32_t i = threadIdx.x; i < 8u * 256u; i += TB) {
        uint32_t m = i & 0xFFu;
        if (i & 0x100u) m = ((m & 0x55u) << 1) | ((m & 0xAAu) >> 1);
        if (i & 0x200u) m = ((m & 0x33u) << 2) | ((m & 0xCCu) >> 2);
        if (i & 0x400u) m = ((m & 0x0Fu) << 4) | ((m & 0xF0u) >> 4);
        s_perm[i] = (uint8_t)m;
    }
    __syncthreads();
    const uint32_t n = *count_in;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        if (count_zero) *count_zero = 0u;  // the queue the coming shade pass appends to
        if (stats) atomicAdd(stats, (unsigned long long)n);  // exact ray count
    }
    lds_u64 *my_stack = (lds_u64 *)reinterpret_cast<unsigned long long *>(smem) + threadIdx.x;
    unsigned long long *my_spill = reinterpret_cast<unsigned long long *>(spill) + (size_t)blockIdx.x * TB + threadIdx.x;
    const float INF = __builtin_inff();
    const int lane = threadIdx.x & 63;

    bool have = false, exhausted = false;
    uint32_t q = 0;
    const uint32_t wave_base = (blockIdx.x * (TB / 64) + (threadIdx.x >> 6)) * 64u;
    const uint32_t wave_stride = gridDim.x * TB;
    uint32_t cursor = 0;
    ptm::f3 inv{};                    // the ray in the normalised scene box: reciprocal direction ...
    // ... and its origin as the two folded terms of the slab test, -(org * inv) moved down (near planes) and up (far planes) by
    // 2^-21 (|org| + 2) |inv|: a node's planes live on the node's own grid, origin + q * step, so a plane distance is
    // q * (step * inv) + (origin * inv + b) and the second term costs a multiply and an add (near) or a multiply-add (far) per
    // axis and visit instead of the subtraction, the set-up and the margins of a relative origin (33 -> 15 instructions).  The
    // margin covers the rounding of b, of the product and of the sum for any origin in [-2, 2) -- absolute, where the
    // relative one was finer, which widens a box by 2^-20 of the scene per unit of |inv| against grid steps of >= 2^-16.
    // C5 +1.0 %, same hits (profiles/r03ab_ab_c5_node8_variants.log, c0m1)
    ptm::f3 bn{}, bf{};
    ptm::RayPre pre{};
    uint32_t oct = 0;                 // ray octant: bit k set where direction component k is negative
    float best_t = tmax, best_V = 0.f, best_W = 0.f, best_det = 1.f;
    uint32_t best_pos = PT_MISS;
    // node group: internal children of one node still to visit.  meta = priority-ordered hit mask (bit p = slot p ^ oct)
    // | imask << 8 | top 16 bits of the smallest entry distance of the node's hit children
    uint32_t ng_base = 0, ng_meta = 0;
    // triangle group: hit leaf slots of the node just visited
    uint32_t tg_base = 0, tg_hits = 0, tg_lmask = 0;
    int sp = 0;
    unsigned long long c_nodes = 0, c_tris = 0, c_node_steps = 0, c_tri_steps = 0, c_refills = 0, c_pops = 0, c_hit_blocks = 0,
                       c_finishes = 0, c_iters = 0, c_leaf_lanes = 0, c_pop_lanes = 0, c_hit_lanes = 0;
#define PT_COUNT_WAVE(C) \
    if (COUNT && lane == __ffsll((long long)__ballot(1)) - 1) (C)++

    for (;;) {
        PT_COUNT_WAVE(c_iters);
        // ---- refill idle lanes from the wave's own chunk sequence (as k_extend)
        const unsigned long long idle = __ballot(!have);
        const int n_idle = __popcll(idle);
        if (!exhausted && n_idle >= refill_min_idle) {
            if (!have) {
                // (rank by v_mbcnt: the prefix mask would be two more registers live through the kernel, and at 80 one value spills)
                const uint32_t v = cursor + __builtin_amdgcn_mbcnt_hi((uint32_t)(idle >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)idle, 0u));
                const uint32_t qq = (v >> 6) * wave_stride + wave_base + (v & 63u);
                if (qq < n) {
                    PT_COUNT_WAVE(c_refills);
                    q = perm ? perm[qq] : qq;  // ray_sort.hip
                    const float4 ra = ptm::ld_stream<false>(rayA + q);
                    const float2 rb = ptm::ld_stream<false>(rayB + q);
                    const ptm::f3 org = { ra.x, ra.y, ra.z };
                    const ptm::f3 dir = { ra.w, rb.x, rb.y };
                    pre = ptm::ray_setup(org, dir);
                    inv = { ptm::safe_inv(dir.x), ptm::safe_inv(dir.y), ptm::safe_inv(dir.z) };
                    const ptm::f3 orgn = { (org.x - nb.cx) * nb.rsx, (org.y - nb.cy) * nb.rsy, (org.z - nb.cz) * nb.rsz };
                    inv = { inv.x * nb.sx, inv.y * nb.sy, inv.z * nb.sz };
                    {
                        const float bx = -(orgn.x * inv.x), by = -(orgn.y * inv.y), bz = -(orgn.z * inv.z);
                        const float px = (fabsf(orgn.x) + 2.0f) * fabsf(inv.x) * 0x1p-21f, py = (fabsf(orgn.y) + 2.0f) * fabsf(inv.y) * 0x1p-21f,
                                    pz = (fabsf(orgn.z) + 2.0f) * fabsf(inv.z) * 0x1p-21f;
                        bn = { bx - px, by - py, bz - pz };
                        bf = { (bx + px) * 1.0000004f, (by + py) * 1.0000004f, (bz + pz) * 1.0000004f };
                    }
                    oct = (inv.x < 0.f ? 1u : 0u) | (inv.y < 0.f ? 2u : 0u) | (inv.z < 0.f ? 4u : 0u);
                    best_t = ray_tmax ? ray_tmax[q] : tmax; best_V = 0.f; best_W = 0.f; best_det = 1.f;  // (shadow rays: extend_kernel.h)
                    best_pos = PT_MISS;
                    // the root as the only child (slot `oct`, so priority 0) of a virtual parent
                    ng_base = 0u;
                    ng_meta = 1u | ((1u << oct) << 8) | (__float_as_uint(tmin < 0.f ? -INF : 0.f) & 0xFFFF0000u);
                    tg_hits = 0u;
                    sp = 0;
                    have = true;
                }
            }
            cursor += (uint32_t)n_idle;
            exhausted = (cursor >> 6) * wave_stride + wave_base >= n;
        }
        const unsigned long long with_ray = __ballot(have);
        if (with_ray == 0ull) break;

        // every lane with a ray has something to do here: triangles of the node it just visited, or a node to visit
        const bool want_tri = have && tg_hits != 0u;
        const bool want_node = have && tg_hits == 0u;
        const int nl = __popcll(__ballot(want_tri)), nn = __popcll(with_ray) - nl;
        if (!(nl > nn || nl >= tri_enter)) {
            if (want_node) {
                if (COUNT) c_nodes++;
                PT_COUNT_WAVE(c_node_steps);
                const uint32_t ph = ng_meta & 0xFFu, im = (ng_meta >> 8) & 0xFFu;
                const uint32_t slot = (uint32_t)(__ffs((int)ph) - 1) ^ oct;
                const uint32_t rest = ph & (ph - 1u);
                const uint32_t idx = ng_base + (uint32_t)__popc(im & ((1u << slot) - 1u));
                if (rest) {  // the other hit children of that node wait on the stack as ONE entry
                    const unsigned long long e = (unsigned long long)ng_base | ((unsigned long long)((ng_meta & 0xFFFFFF00u) | rest) << 32);
                    if (!SPILL || sp < lds_stack) my_stack[sp * TB] = e;
                    else my_spill[(size_t)(sp - lds_stack) * spill_stride] = e;
                    sp++;
                }
                const uint4 *nd = nodes8 + 4 * (size_t)idx;
                const uint4 q0 = nd[0], q1 = nd[1], q2 = nd[2], hd = nd[3];  // the whole node: four 16-B loads of one 64-B record
                // the node's grid: origin = o16 * 2^-14 - 2 (exact), step = 2^-e; a plane distance is ONE fma,
                // q * (step * inv) + (origin * inv + b), b = the ray's folded origin term with its outward margin (above)
                const ptm::f3 o = { __builtin_fmaf((float)(hd.x & 0xFFFFu), 0x1p-14f, -2.0f), __builtin_fmaf((float)(hd.x >> 16), 0x1p-14f, -2.0f),
                                    __builtin_fmaf((float)(hd.y & 0xFFFFu), 0x1p-14f, -2.0f) };
                const ptm::f3 stp = { __uint_as_float((127u - ((hd.y >> 16) & 31u)) << 23), __uint_as_float((127u - ((hd.y >> 21) & 31u)) << 23),
                                      __uint_as_float((127u - (hd.y >> 26)) << 23) };
                const ptm::f3 oi = { o.x * inv.x, o.y * inv.y, o.z * inv.z };
                const ptm::f3 on = { oi.x + bn.x, oi.y + bn.y, oi.z + bn.z };
                const ptm::f3 of = { __builtin_fmaf(oi.x, 1.0000004f, bf.x), __builtin_fmaf(oi.y, 1.0000004f, bf.y), __builtin_fmaf(oi.z, 1.0000004f, bf.z) };
                const ptm::f3 an = { stp.x * inv.x, stp.y * inv.y, stp.z * inv.z }, af = { an.x * 1.0000004f, an.y * 1.0000004f, an.z * 1.0000004f };
                // near / far rows by the ray's octant (bit-field insert with all-ones / all-zeros masks, as k_extend<hbm>):
                // rows lo.x = q0.xy, lo.y = q0.zw, lo.z = q1.xy, hi.x = q1.zw, hi.y = q2.xy, hi.z = q2.zw (4 children per dword)
                const uint32_t mx = (oct & 1u) ? 0xFFFFFFFFu : 0u, my = (oct & 2u) ? 0xFFFFFFFFu : 0u, mz = (oct & 4u) ? 0xFFFFFFFFu : 0u;
                const uint32_t rnx[2] = { PT_BFI(mx, q1.z, q0.x), PT_BFI(mx, q1.w, q0.y) }, rfx[2] = { PT_BFI(mx, q0.x, q1.z), PT_BFI(mx, q0.y, q1.w) };
                const uint32_t rny[2] = { PT_BFI(my, q2.x, q0.z), PT_BFI(my, q2.y, q0.w) }, rfy[2] = { PT_BFI(my, q0.z, q2.x), PT_BFI(my, q0.w, q2.y) };
                const uint32_t rnz[2] = { PT_BFI(mz, q2.z, q1.x), PT_BFI(mz, q2.w, q1.y) }, rfz[2] = { PT_BFI(mz, q1.x, q2.z), PT_BFI(mz, q1.y, q2.w) };
                // hit mask and the smallest entry distance, child by child (nothing per child stays live)
                uint32_t h = 0;
                float gmin = INF;
#define PT_BYTE(W, B) ((float)(((W) >> (8 * (B))) & 0xFFu))   /* v_cvt_f32_ubyteB */
#define PT_SLAB8(K)                                                                                               \
    {                                                                                                             \
        const float nxv = __builtin_fmaf(PT_BYTE(rnx[(K) >> 2], (K) & 3), an.x, on.x), nyv = __builtin_fmaf(PT_BYTE(rny[(K) >> 2], (K) & 3), an.y, on.y), \
                    nzv = __builtin_fmaf(PT_BYTE(rnz[(K) >> 2], (K) & 3), an.z, on.z), fxv = __builtin_fmaf(PT_BYTE(rfx[(K) >> 2], (K) & 3), af.x, of.x), \
                    fyv = __builtin_fmaf(PT_BYTE(rfy[(K) >> 2], (K) & 3), af.y, of.y), fzv = __builtin_fmaf(PT_BYTE(rfz[(K) >> 2], (K) & 3), af.z, of.z); \
        const float tn = fmaxf(fmaxf(nxv, nyv), max_raw_s(nzv, tmin));                                            \
        const float tf = fminf(fminf(fxv, fyv), min_raw(fzv, best_t));                                            \
        /* (a branch-free form -- the hit bit shifted into h through the carry, v_addc_co_u32, the minimum through a select --  \
           is one 240-instruction block instead of nine and 4 % SLOWER on C5: profiles/r03ab_ab_c5_node8_variants.log;           \
           two children per v_pk_fma_f32 -- 24 packed multiply-adds instead of 48 -- needs the twelve distances of a pair live   \
           at once: 10 spilled registers at 6 waves (-30 %), -6 % at 5 waves without spills, -1.5 % with only the near planes   \
           packed: profiles/r03be_ab_c5_pkfma.log; round 4: the bytes through ONE v_perm_b32 per pair of children into halves     \
           0x6400 | q = 1024 + q that v_fma_mix_f32 reads, the 1024 folded into the constant term with a 2^-12-step margin -- 24  \
           permutes instead of 48 conversions, and v_cvt_f32_ubyte issues at 4.2 cycles like a permute, not at 2.5: the node     \
           step loses 7 instructions of 240, C5 -1 %, C5x -20 % (6 dwords spilled at 72 VGPRs): profiles/r04g_ab_node8_perm_*)  */ \
        const bool hk = tn <= tf;                                                                                 \
        h |= hk ? (1u << (K)) : 0u;                                                                               \
        if constexpr (BOUND) gmin = hk ? min_raw(tn, gmin) : gmin;                                                \
    }
                PT_SLAB8(0) PT_SLAB8(1) PT_SLAB8(2) PT_SLAB8(3) PT_SLAB8(4) PT_SLAB8(5) PT_SLAB8(6) PT_SLAB8(7)
#undef PT_SLAB8
#undef PT_BYTE
                const uint32_t nim = hd.z >> 24, lm = hd.w >> 24;
                h &= nim | lm;   // (empty slots are inverted intervals, never hit; the mask costs one instruction)
                uint32_t hi_ = h & nim;
                // slot mask -> priority mask: bit p = slot p ^ oct (neighbours / pairs / nibbles swapped per octant bit)
                // -- through a 2-KB table in LDS, one ds_read_u8 instead of fourteen instructions (three conditional swaps):
                // C5 3 236 -> 3 278 Mrays/s, three of three interleaved rounds (profiles/r03bg_ab_c5_perm_lut.log)
                hi_ = s_perm[(oct << 8) | hi_];
                // entry distance of the group, rounded DOWN to 16 bits (negative values -- only with a negative tmin --
                // away from zero)
                const uint32_t gb = __float_as_uint(gmin);
                const uint32_t g16 = BOUND ? (gb + ((uint32_t)((int32_t)gb >> 31) & 0xFFFFu)) & 0xFFFF0000u : 0u;
                ng_base = hd.z & 0x00FFFFFFu;
                ng_meta = hi_ | (nim << 8) | g16;
                tg_base = hd.w & 0x00FFFFFFu;
                tg_hits = h & lm;
                tg_lmask = lm;
            }
        } else for (;;) {
          if (have && tg_hits != 0u) {
            if (COUNT) { c_tris++; c_leaf_lanes++; }
            PT_COUNT_WAVE(c_tri_steps);
            const uint32_t slot = (uint32_t)(__ffs((int)tg_hits) - 1);
            tg_hits &= tg_hits - 1u;
            const uint32_t pos = tg_base + (uint32_t)__popc(tg_lmask & ((1u << slot) - 1u));
            // the vertices come from the 64-B record k_shade gathers anyway (extend_kernel.h REC64: never straddles a line);
            // its .w components are the normal, so the ids of two rivals at exactly the same t come from tri4
            const float4 a = rec64[4 * (size_t)pos + 0], b = rec64[4 * (size_t)pos + 1], c = rec64[4 * (size_t)pos + 2];
            float t, V, W, det;
            bool divided = false;
            if (ptm::tri_test(pre, { a.x, a.y, a.z }, { b.x, b.y, b.z }, { c.x, c.y, c.z }, tmin, tmax, t, V, W, det, COUNT ? &divided : nullptr)) {
                // closest t; equal t -> lowest gl_PrimitiveID
                bool closer = t < best_t;
                if (!closer && t == best_t)
                    closer = best_pos == PT_MISS || __float_as_uint(tri4[3 * (size_t)pos].w) < __float_as_uint(tri4[3 * (size_t)best_pos].w);
                if (closer) {
                    best_t = t; best_V = V; best_W = W; best_det = det; best_pos = pos;
                    if (ray_tmax) { sp = 0; tg_hits = 0u; ng_meta &= 0xFFFFFF00u; }  // any hit ends a shadow ray
                }
            }
            if (COUNT && divided) { PT_COUNT_WAVE(c_hit_blocks); c_hit_lanes++; }
          }
          // (tri_stay = 65: never repeats; 4 / 8 measured equal.  The loop stays: without it the compiler lays the vote and the
          // triangle step out as one block and C5 / C5x lose 0.4 / 0.8 %, profiles/r03bh_ab_c5_tri_loop.log)
          if (__popcll(__ballot(have && tg_hits != 0u)) < tri_stay) break;
        }
        // ---- nothing left of the current node: the next pending group that can still hold the closest hit, or done
        if (have && tg_hits == 0u && (ng_meta & 0xFFu) == 0u) {
            bool got = false;
            while (sp > 0) {
                PT_COUNT_WAVE(c_pops);
                if (COUNT) c_pop_lanes++;
                sp--;
                unsigned long long e;
                if (!SPILL || sp < lds_stack) e = my_stack[sp * TB];
                else e = my_spill[(size_t)(sp - lds_stack) * spill_stride];
                const uint32_t meta = (uint32_t)(e >> 32);
                if (!BOUND || __uint_as_float(meta & 0xFFFF0000u) <= best_t) {
                    ng_base = (uint32_t)e;
                    ng_meta = meta;
                    got = true;
                    break;
                }
            }
            if (!got) {
                PT_COUNT_WAVE(c_finishes);
                const bool miss = best_pos == PT_MISS;
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

// WAVES per SIMD asked of the compiler: 6 (<= 80 VGPRs) everywhere, 7 (72 VGPRs, no spills in the no-spill kernel) for scenes
// beyond the Infinity Cache, whose walk waits on HBM: C5x +1.5 % over 6 waves at the same 10 stack entries, C5 (cache-resident)
// -3.5 % (profiles/r03bl_ab_c5_lds_stack.log)
template <bool COUNT, bool SPILL, int WAVES>
__global__ __launch_bounds__(TB, WAVES) void k_extend8(const uint4 *__restrict__ nodes8, NormBox nb, const float4 *__restrict__ tri4, const float4 *__restrict__ rec64,
                                                const float4 *__restrict__ rayA, const float2 *__restrict__ rayB,
                                                float4 *__restrict__ hit, const uint32_t *__restrict__ count_in,
                                                uint32_t *count_zero, unsigned long long *stats, uint2 *__restrict__ spill,
                                                uint32_t spill_stride, int refill_min_idle, float tmin, float tmax, int lds_stack,
                                                int raw_hit, const uint32_t *__restrict__ perm, const float *__restrict__ ray_tmax)
{
    extend8_body<COUNT, SPILL, (WAVES != 7) && PT_E8_BOUND6 != 0>(nodes8, nb, tri4, rec64, rayA, rayB, hit, count_in, count_zero, stats, spill, spill_stride, refill_min_idle, tmin,
                        tmax, lds_stack, raw_hit, perm, ray_tmax);
}

}  // namespace
