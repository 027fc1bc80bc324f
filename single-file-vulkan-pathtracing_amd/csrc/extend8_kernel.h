// extend8_kernel.h -- closest hit over the BVH8 (scenes walked out of L2 / MALL / HBM; lbvh_build.hip k_w8_*).
//
// Same contract as k_extend (extend_kernel.h): persistent threads, lane refill from a wave-private sequence of 64-ray
// chunks, the canonical watertight triangle test, closest t with ties to the lowest primitive id -- the hit records are
// bit-identical.  What differs is what a ray fetches and what it remembers:
//   * a node is one 128-B line with EIGHT children (fp16 planes in the normalised scene box): beyond L2 this chip
//     charges a divergent load per distinct line, not per byte (scripts/ubench/gather_rate.hip), and eight-wide
//     nodes need a third fewer visits than four-wide ones;
//   * internal children are contiguous and a node's leaf triangles are contiguous, so what is pending of a node is
//     {child_base, mask of the internal children still to visit}: ONE 8-byte stack entry per visited node (with the
//     smallest entry distance of its hit children in the top 16 bits, so a popped group that lies behind the best hit
//     is dropped without a fetch) instead of one entry per child, and no sorting network: children sit in the slot of
//     their octant, "slot xor ray octant" in ascending order is roughly front to back;
//   * the triangles a node visit found (mask of hit leaf slots) are tested before the walk goes on; steps are
//     vote-scheduled like k_extend's (the wave runs the node code or the triangle code, whichever more lanes wait for).
#pragma once
#include "extend_kernel.h"

namespace {

template <bool COUNT>
__device__ __forceinline__ void extend8_body(const uint4 *__restrict__ nodes8, NormBox nb, const float4 *__restrict__ tri4,
                                             const float4 *__restrict__ rayA, const float2 *__restrict__ rayB,
                                             float4 *__restrict__ hit, const uint32_t *__restrict__ count_in,
                                             uint32_t *count_zero, unsigned long long *stats, uint2 *__restrict__ spill,
                                             uint32_t spill_stride, int refill_min_idle, float tmin, float tmax, int lds_stack,
                                             int raw_hit, const uint32_t *__restrict__ perm, const float *__restrict__ ray_tmax)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const uint32_t n = *count_in;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        if (count_zero) *count_zero = 0u;  // the queue the coming shade pass appends to
        if (stats) atomicAdd(stats, (unsigned long long)n);  // exact ray count
    }
    lds_u64 *my_stack = (lds_u64 *)reinterpret_cast<unsigned long long *>(smem) + threadIdx.x;
    unsigned long long *my_spill = reinterpret_cast<unsigned long long *>(spill) + (size_t)blockIdx.x * TB + threadIdx.x;
    const float INF = __builtin_inff();
    const int lane = threadIdx.x & 63;
    const unsigned long long lt = (1ull << lane) - 1ull;

    bool have = false, exhausted = false;
    uint32_t q = 0;
    const uint32_t wave_base = (blockIdx.x * (TB / 64) + (threadIdx.x >> 6)) * 64u;
    const uint32_t wave_stride = gridDim.x * TB;
    uint32_t cursor = 0;
    ptm::f3 inv{}, invf{}, on{}, of{};
    ptm::RayPre pre{};
    uint32_t ax = 0, ay = 0, az = 0;  // 48 where the direction component is negative: byte offset of the near-plane row
    uint32_t oct = 0;                 // ray octant: bit k set where direction component k is negative
    float best_t = tmax, best_V = 0.f, best_W = 0.f, best_det = 1.f;
    uint32_t best_pos = PT_MISS, best_prim = PT_MISS;
    // node group: internal children of one node still to visit.  meta = priority-ordered hit mask (bit p = slot p ^ oct)
    // | imask << 8 | top 16 bits of the smallest entry distance of the node's hit children
    uint32_t ng_base = 0, ng_meta = 0;
    // triangle group: hit leaf slots of the node just visited
    uint32_t tg_base = 0, tg_hits = 0, tg_lmask = 0;
    int sp = 0;
    unsigned long long c_nodes = 0, c_tris = 0, c_node_steps = 0, c_tri_steps = 0, c_refills = 0, c_pops = 0, c_hit_blocks = 0,
                       c_finishes = 0, c_iters = 0;
#define PT_COUNT_WAVE(C) \
    if (COUNT && lane == __ffsll((long long)__ballot(1)) - 1) (C)++

    for (;;) {
        PT_COUNT_WAVE(c_iters);
        // ---- refill idle lanes from the wave's own chunk sequence (as k_extend)
        const unsigned long long idle = __ballot(!have);
        const int n_idle = __popcll(idle);
        if (!exhausted && n_idle >= refill_min_idle) {
            if (!have) {
                const uint32_t v = cursor + (uint32_t)__popcll(idle & lt);
                const uint32_t qq = (v >> 6) * wave_stride + wave_base + (v & 63u);
                if (qq < n) {
                    PT_COUNT_WAVE(c_refills);
                    q = perm ? perm[qq] : qq;  // ray_sort.hip
                    const float4 ra = rayA[q];
                    const float2 rb = rayB[q];
                    const ptm::f3 org = { ra.x, ra.y, ra.z };
                    const ptm::f3 dir = { ra.w, rb.x, rb.y };
                    pre = ptm::ray_setup(org, dir);
                    inv = { ptm::safe_inv(dir.x), ptm::safe_inv(dir.y), ptm::safe_inv(dir.z) };
                    const ptm::f3 orgn = { (org.x - nb.cx) * nb.rsx, (org.y - nb.cy) * nb.rsy, (org.z - nb.cz) * nb.rsz };
                    inv = { inv.x * nb.sx, inv.y * nb.sy, inv.z * nb.sz };
                    slab_setup(orgn, inv, invf, on, of);
                    ax = inv.x < 0.f ? 48u : 0u;
                    ay = inv.y < 0.f ? 48u : 0u;
                    az = inv.z < 0.f ? 48u : 0u;
                    oct = (inv.x < 0.f ? 1u : 0u) | (inv.y < 0.f ? 2u : 0u) | (inv.z < 0.f ? 4u : 0u);
                    best_t = ray_tmax ? ray_tmax[q] : tmax; best_V = 0.f; best_W = 0.f; best_det = 1.f;  // (shadow rays: extend_kernel.h)
                    best_pos = PT_MISS; best_prim = PT_MISS;
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
        if (__ballot(have) == 0ull) break;

        // every lane with a ray has something to do here: triangles of the node it just visited, or a node to visit
        const bool want_tri = have && tg_hits != 0u;
        const bool want_node = have && tg_hits == 0u;
        const int nn = __popcll(__ballot(want_node)), nl = __popcll(__ballot(want_tri));
        if (nn >= nl) {
            if (want_node) {
                if (COUNT) c_nodes++;
                PT_COUNT_WAVE(c_node_steps);
                const uint32_t ph = ng_meta & 0xFFu, im = (ng_meta >> 8) & 0xFFu;
                const uint32_t slot = (uint32_t)(__ffs((int)ph) - 1) ^ oct;
                const uint32_t rest = ph & (ph - 1u);
                const uint32_t idx = ng_base + (uint32_t)__popc(im & ((1u << slot) - 1u));
                if (rest) {  // the other hit children of that node wait on the stack as ONE entry
                    const unsigned long long e = (unsigned long long)ng_base | ((unsigned long long)((ng_meta & 0xFFFFFF00u) | rest) << 32);
                    if (sp < lds_stack) my_stack[sp * TB] = e;
                    else my_spill[(size_t)(sp - lds_stack) * spill_stride] = e;
                    sp++;
                }
                const char *nb_ = reinterpret_cast<const char *>(nodes8) + 128 * (size_t)idx;
                // near / far plane rows picked through the load address (row = 8 halves = 16 B; lo rows at 0/16/32, hi at 48/64/80)
                const uint4 rnx = *reinterpret_cast<const uint4 *>(nb_ + ax), rfx = *reinterpret_cast<const uint4 *>(nb_ - ax + 48),
                            rny = *reinterpret_cast<const uint4 *>(nb_ + ay + 16), rfy = *reinterpret_cast<const uint4 *>(nb_ - ay + 64),
                            rnz = *reinterpret_cast<const uint4 *>(nb_ + az + 32), rfz = *reinterpret_cast<const uint4 *>(nb_ - az + 80);
                const uint4 m6 = *reinterpret_cast<const uint4 *>(nb_ + 96);
                // hit mask and the smallest entry distance, child by child (nothing per child stays live: 84 VGPRs, 6 waves)
                uint32_t h = 0;
                float gmin = INF;
#define PT_SLAB8(K, REGC, HI)                                                                             \
    {                                                                                                     \
        float nxv, nyv, nzv, fxv, fyv, fzv;                                                               \
        PT_MIXH(nxv, rnx.REGC, HI, inv.x, on.x); PT_MIXH(nyv, rny.REGC, HI, inv.y, on.y);                 \
        PT_MIXH(nzv, rnz.REGC, HI, inv.z, on.z); PT_MIXH(fxv, rfx.REGC, HI, invf.x, of.x);                \
        PT_MIXH(fyv, rfy.REGC, HI, invf.y, of.y); PT_MIXH(fzv, rfz.REGC, HI, invf.z, of.z);               \
        const float tn = fmaxf(fmaxf(nxv, nyv), max_raw_s(nzv, tmin));                                    \
        const float tf = fminf(fminf(fxv, fyv), min_raw(fzv, best_t));                                    \
        const bool hk = tn <= tf;                                                                         \
        h |= hk ? (1u << K) : 0u;                                                                         \
        gmin = hk ? min_raw(tn, gmin) : gmin;                                                             \
    }
                PT_SLAB8(0, x, 0) PT_SLAB8(1, x, 1) PT_SLAB8(2, y, 0) PT_SLAB8(3, y, 1)
                PT_SLAB8(4, z, 0) PT_SLAB8(5, z, 1) PT_SLAB8(6, w, 0) PT_SLAB8(7, w, 1)
#undef PT_SLAB8
                const uint32_t nim = m6.z & 0xFFu, lm = (m6.z >> 8) & 0xFFu;
                uint32_t hi_ = h & nim;
                // slot mask -> priority mask: bit p = slot p ^ oct (swap neighbours / pairs / nibbles per octant bit)
                if (oct & 1u) hi_ = ((hi_ & 0x55u) << 1) | ((hi_ & 0xAAu) >> 1);
                if (oct & 2u) hi_ = ((hi_ & 0x33u) << 2) | ((hi_ & 0xCCu) >> 2);
                if (oct & 4u) hi_ = ((hi_ & 0x0Fu) << 4) | ((hi_ & 0xF0u) >> 4);
                // entry distance of the group, rounded DOWN to 16 bits (negative values -- only with a negative tmin --
                // away from zero)
                const uint32_t gb = __float_as_uint(gmin);
                const uint32_t g16 = (gb + ((uint32_t)((int32_t)gb >> 31) & 0xFFFFu)) & 0xFFFF0000u;
                ng_base = m6.x;
                ng_meta = hi_ | (nim << 8) | g16;
                tg_base = m6.y;
                tg_hits = h & lm;
                tg_lmask = lm;
            }
        } else if (want_tri) {
            if (COUNT) c_tris++;
            PT_COUNT_WAVE(c_tri_steps);
            const uint32_t slot = (uint32_t)(__ffs((int)tg_hits) - 1);
            tg_hits &= tg_hits - 1u;
            const uint32_t pos = tg_base + (uint32_t)__popc(tg_lmask & ((1u << slot) - 1u));
            const float4 a = tri4[3 * (size_t)pos + 0], b = tri4[3 * (size_t)pos + 1], c = tri4[3 * (size_t)pos + 2];
            float t, V, W, det;
            bool divided = false;
            if (ptm::tri_test(pre, { a.x, a.y, a.z }, { b.x, b.y, b.z }, { c.x, c.y, c.z }, tmin, tmax, t, V, W, det, COUNT ? &divided : nullptr)) {
                const uint32_t prim = __float_as_uint(a.w);
                // closest t; equal t -> lowest gl_PrimitiveID
                if (t < best_t || (t == best_t && prim < best_prim)) {
                    best_t = t; best_V = V; best_W = W; best_det = det; best_pos = pos; best_prim = prim;
                    if (ray_tmax) { sp = 0; tg_hits = 0u; ng_meta &= 0xFFFFFF00u; }  // any hit ends a shadow ray
                }
            }
            if (COUNT && divided) { PT_COUNT_WAVE(c_hit_blocks); }
        }
        // ---- nothing left of the current node: the next pending group that can still hold the closest hit, or done
        if (have && tg_hits == 0u && (ng_meta & 0xFFu) == 0u) {
            bool got = false;
            while (sp > 0) {
                PT_COUNT_WAVE(c_pops);
                sp--;
                unsigned long long e;
                if (sp < lds_stack) e = my_stack[sp * TB];
                else e = my_spill[(size_t)(sp - lds_stack) * spill_stride];
                const uint32_t meta = (uint32_t)(e >> 32);
                if (__uint_as_float(meta & 0xFFFF0000u) <= best_t) {
                    ng_base = (uint32_t)e;
                    ng_meta = meta;
                    got = true;
                    break;
                }
            }
            if (!got) {
                PT_COUNT_WAVE(c_finishes);
                const bool miss = best_pos == PT_MISS;
                hit[q] = raw_hit ? make_float4(__uint_as_float(best_pos), best_V, best_W, best_det)
                                 : make_float4(__uint_as_float(best_pos), miss ? 0.f : best_t,
                                               miss ? 0.f : ptm::fdiv(best_V, best_det), miss ? 0.f : ptm::fdiv(best_W, best_det));
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
        }
    }
}

#ifndef PT_EXTEND8_WAVES
#define PT_EXTEND8_WAVES 6
#endif
template <bool COUNT>
__global__ __launch_bounds__(TB, PT_EXTEND8_WAVES) void k_extend8(const uint4 *__restrict__ nodes8, NormBox nb, const float4 *__restrict__ tri4,
                                                const float4 *__restrict__ rayA, const float2 *__restrict__ rayB,
                                                float4 *__restrict__ hit, const uint32_t *__restrict__ count_in,
                                                uint32_t *count_zero, unsigned long long *stats, uint2 *__restrict__ spill,
                                                uint32_t spill_stride, int refill_min_idle, float tmin, float tmax, int lds_stack,
                                                int raw_hit, const uint32_t *__restrict__ perm, const float *__restrict__ ray_tmax)
{
    extend8_body<COUNT>(nodes8, nb, tri4, rayA, rayB, hit, count_in, count_zero, stats, spill, spill_stride, refill_min_idle, tmin,
                        tmax, lds_stack, raw_hit, perm, ray_tmax);
}

}  // namespace
