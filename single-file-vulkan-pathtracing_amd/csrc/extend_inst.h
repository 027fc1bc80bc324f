// extend_inst.h -- k_extend_inst: the general two-level closest-hit kernel (TLAS over instance boxes, BLAS in object space) with
// 128-B fp32 nodes and 8-byte stack entries.  The fallback of instanced scenes: AUTO walks them with k_extend_inst16
// (extend_inst16.h) whenever both levels fit that kernel's 15-bit child codes and the BLAS fits LDS, and with this one
// otherwise (big BLAS, >= 32768 instances, tmin <= 0).
#pragma once
#include "extend_kernel.h"

namespace {

// ---- extend, two-level variant (BASELINE config C4: instanced scenes) ----------------------------
// TLAS = BVH4 over the instances' world boxes, BLAS = the scene's BVH4 in object space.  Same
// persistent-thread structure; one stack serves both levels: entering an instance pushes an EXIT
// marker, everything above it belongs to the BLAS walk, popping it restores the world-space ray.
// The ray goes to object space un-normalised (Vulkan semantics: t is the same parameter in both
// spaces), so entry distances and the best hit compare across levels.  Not in the reference
// (one identity instance, main.cpp:515-538); semantics in DESIGN.md section 3.
constexpr uint32_t EXIT_MARK = 0x7FFFFFFFu;

template <bool COUNT, bool LDS_BLAS, bool SHADOW = false>
__global__ __launch_bounds__(TB) void k_extend_inst(const float4 *__restrict__ tlas, const float4 *__restrict__ g_blas,
                                                    const float4 *__restrict__ g_tri4, uint32_t n_blas_wide,
                                                    uint32_t n_tris, const float4 *__restrict__ inst6,
                                                    const uint32_t *__restrict__ inst_id, const float4 *__restrict__ rayA,
                                                    const float2 *__restrict__ rayB, float4 *__restrict__ hit,
                                                    uint32_t *__restrict__ hit_inst, const uint32_t *__restrict__ count_in,
                                                    uint32_t *count_zero, unsigned long long *stats,
                                                    uint2 *__restrict__ spill, uint32_t spill_stride, int refill_min_idle,
                                                    float tmin, float tmax, int raw_hit, const float *__restrict__ ray_tmax)
{
    // SHADOW (the NEE pipeline's shadow rays): a per-ray upper bound instead of tmax, any hit below it ends the walk
    extern __shared__ __attribute__((aligned(16))) char smem[];
    uint2 *stack = reinterpret_cast<uint2 *>(smem);  // [LDS_STACK][TB]
    const float4 *blas = g_blas;
    const float4 *tri4 = g_tri4;
    if (LDS_BLAS) {  // the BLAS is shared by every instance: keep it (and 3 permuted triangle copies) in LDS
        float4 *s_blas = reinterpret_cast<float4 *>(smem + (size_t)LDS_STACK * TB * sizeof(uint2));
        float4 *s_tri = s_blas + LDS_NODE_F4 * (size_t)n_blas_wide;
        for (uint32_t i = threadIdx.x; i < 8 * n_blas_wide; i += TB) s_blas[(i >> 3) * LDS_NODE_F4 + (i & 7u)] = g_blas[i];
        for (uint32_t i = threadIdx.x; i < 3 * n_tris; i += TB) {
            const float4 v = g_tri4[i];
            s_tri[i] = make_float4(v.y, v.z, v.x, v.w);
            s_tri[3 * n_tris + i] = make_float4(v.z, v.x, v.y, v.w);
            s_tri[6 * n_tris + i] = v;
        }
        __syncthreads();
        blas = s_blas;
        tri4 = s_tri;
    }
    const uint32_t n = *count_in;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        if (count_zero) *count_zero = 0u;
        if (stats) atomicAdd(stats, (unsigned long long)n);
    }
    lds_u64 *my_stack = (lds_u64 *)reinterpret_cast<unsigned long long *>(stack) + threadIdx.x;
    unsigned long long *my_spill = reinterpret_cast<unsigned long long *>(spill) + (size_t)blockIdx.x * TB + threadIdx.x;
    const float INF = __builtin_inff();
    const int lane = threadIdx.x & 63;
    const unsigned long long lt = (1ull << lane) - 1ull;

    bool have = false, exhausted = false, in_blas = false;
    uint32_t q = 0;
    ptm::f3 org_w{}, dir_w{}, inv_w{};   // world-space ray
    ptm::f3 inv{}, invf{}, on{}, of{}, orgp{};  // ray of the level being walked (on/of: folded slab origins; orgp: origin permuted to kx,ky,kz)
    uint32_t tri_base = 0;
    uint32_t ax = 0, ay = 0, az = 0;      // 48 where the walked level's direction component is negative
    ptm::RayPre pre{};
    float best_t = tmax, best_V = 0.f, best_W = 0.f, best_det = 1.f;
    uint32_t best_pos = PT_MISS, best_prim = PT_MISS, best_ipos = PT_MISS, best_iid = PT_MISS;
    uint32_t cur = SENTINEL, cur_ipos = 0, cur_iid = 0;
    float cur_t = 0.f;
    int sp = 0;
    unsigned long long c_nodes = 0, c_tris = 0;
    const uint32_t wave_base = (blockIdx.x * (TB / 64) + (threadIdx.x >> 6)) * 64u;
    const uint32_t wave_stride = gridDim.x * TB;
    uint32_t cursor = 0;

    auto push = [&](uint32_t w, float t) {
        const unsigned long long e = stack_entry(w, t);
        if (sp < LDS_STACK) my_stack[sp * TB] = e;
        else my_spill[(size_t)(sp - LDS_STACK) * spill_stride] = e;
        sp++;
    };
    auto pop = [&]() -> uint32_t {
        while (sp > 0) {
            sp--;
            unsigned long long e64;
            if (sp < LDS_STACK) e64 = my_stack[sp * TB];
            else e64 = my_spill[(size_t)(sp - LDS_STACK) * spill_stride];
            const uint2 e = make_uint2((uint32_t)e64, (uint32_t)(e64 >> 32));
            if (e.x == EXIT_MARK) {  // the instance is done: back to the world-space ray and the TLAS
                inv = inv_w;
                slab_setup(org_w, inv_w, invf, on, of);
                ax = inv.x < 0.f ? 48u : 0u;
                ay = inv.y < 0.f ? 48u : 0u;
                az = inv.z < 0.f ? 48u : 0u;
                in_blas = false;
                continue;
            }
            if (__uint_as_float(e.y) <= best_t) {
                cur_t = __uint_as_float(e.y);
                return e.x;
            }
        }
        return SENTINEL;
    };

    for (;;) {
        const unsigned long long idle = __ballot(!have);
        const int n_idle = __popcll(idle);
        if (!exhausted && n_idle >= refill_min_idle) {
            if (!have) {
                const uint32_t v = cursor + (uint32_t)__popcll(idle & lt);
                const uint32_t qq = (v >> 6) * wave_stride + wave_base + (v & 63u);
                if (qq < n) {
                    q = qq;
                    const float4 ra = rayA[q];
                    const float2 rb = rayB[q];
                    org_w = { ra.x, ra.y, ra.z };
                    dir_w = { ra.w, rb.x, rb.y };
                    inv_w = { ptm::safe_inv(dir_w.x), ptm::safe_inv(dir_w.y), ptm::safe_inv(dir_w.z) };
                    inv = inv_w;
                    slab_setup(org_w, inv_w, invf, on, of);
                    ax = inv.x < 0.f ? 48u : 0u;
                    ay = inv.y < 0.f ? 48u : 0u;
                    az = inv.z < 0.f ? 48u : 0u;
                    in_blas = false;
                    best_t = SHADOW ? ray_tmax[q] : tmax; best_V = 0.f; best_W = 0.f; best_det = 1.f;
                    best_pos = PT_MISS; best_prim = PT_MISS; best_ipos = PT_MISS; best_iid = PT_MISS;
                    cur = 0u;  // TLAS root
                    cur_t = tmin;
                    sp = 0;
                    have = true;
                }
            }
            cursor += (uint32_t)n_idle;
            exhausted = (cursor >> 6) * wave_stride + wave_base >= n;
        }
        if (__ballot(have) == 0ull) break;

        // ---- node phase (either level).  (Vote-scheduled single steps as in k_extend measured -4 % here, leaving
        // the node loop early when few lanes still descend +-0.)
        while (have && !(cur & PT_LEAF)) {
            float4 nx, fx, ny, fy, nz, fz, cw;
            if (LDS_BLAS && in_blas) {
                const char *nd = reinterpret_cast<const char *>(blas + LDS_NODE_F4 * (size_t)cur);
                nx = PT_F4(nd + ax); fx = PT_F4(nd - ax + 48); ny = PT_F4(nd + ay + 16); fy = PT_F4(nd - ay + 64);
                nz = PT_F4(nd + az + 32); fz = PT_F4(nd - az + 80); cw = PT_F4(nd + 96);
            } else {
                const char *nd = reinterpret_cast<const char *>((LDS_BLAS ? tlas : (in_blas ? g_blas : tlas)) + 8 * (size_t)cur);
                nx = PT_F4(nd + ax); fx = PT_F4(nd - ax + 48); ny = PT_F4(nd + ay + 16); fy = PT_F4(nd - ay + 64);
                nz = PT_F4(nd + az + 32); fz = PT_F4(nd - az + 80); cw = PT_F4(nd + 96);
            }
            if (COUNT) c_nodes++;
            float t0, t1, t2, t3;
            uint32_t w0 = __float_as_uint(cw.x), w1 = __float_as_uint(cw.y), w2 = __float_as_uint(cw.z),
                     w3 = __float_as_uint(cw.w);
            PT_SLAB4(t0, x)
            PT_SLAB4(t1, y)
            PT_SLAB4(t2, z)
            PT_SLAB4(t3, w)
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
            if (t3 < INF) push(w3, t3);
            if (t2 < INF) push(w2, t2);
            if (t1 < INF) push(w1, t1);
            if (t0 < INF) { cur = w0; cur_t = t0; }
            else cur = pop();
        }
        // ---- leaf phase
        // entering an instance costs ~130 VALU instructions (ray transform, three true divides, slab set-up): lanes that
        // want to wait until ENTER_MIN of them do, or until no other lane of the wave has triangle work left
        // (extend -6 %, C4 +1.5 %; 8, 16 and 32 measured alike)
        constexpr int ENTER_MIN = 16;
        const int n_enter = __popcll(__ballot(have && cur != SENTINEL && !in_blas));
        const bool others = __ballot(have && cur != SENTINEL && in_blas) != 0ull;
        const bool do_enter = n_enter >= ENTER_MIN || !others;
        if (have) {
            if (cur != SENTINEL && (in_blas || do_enter)) {
                const uint32_t first = cur & 0x0FFFFFFFu, cnt = ((cur >> 28) & 7u) + 1u;
                if (in_blas) {
                    if (COUNT) c_tris += cnt;
                    for (uint32_t k = 0; k < cnt; k++) {
                        const uint32_t pos = first + k;
                        const size_t ti = LDS_BLAS ? (size_t)tri_base + 3 * (size_t)pos : 3 * (size_t)pos;
                        const float4 a = tri4[ti + 0], b = tri4[ti + 1], c = tri4[ti + 2];
                        float t, V, W, det;
                        const bool th = LDS_BLAS
                            ? ptm::tri_test_perm(pre, orgp, { a.x, a.y, a.z }, { b.x, b.y, b.z }, { c.x, c.y, c.z }, tmin, tmax, t, V, W, det)
                            : ptm::tri_test(pre, { a.x, a.y, a.z }, { b.x, b.y, b.z }, { c.x, c.y, c.z }, tmin, tmax, t, V, W, det);
                        if (th) {
                            const uint32_t prim = __float_as_uint(a.w);
                            // closest t; equal t -> lowest (gl_InstanceID, gl_PrimitiveID)
                            if (t < best_t || (t == best_t && (cur_iid < best_iid || (cur_iid == best_iid && prim < best_prim)))) {
                                best_t = t; best_V = V; best_W = W; best_det = det; best_pos = pos; best_prim = prim;
                                best_ipos = cur_ipos; best_iid = cur_iid;
                                if (SHADOW) sp = 0;  // any hit will do
                            }
                        }
                    }
                    cur = pop();
                } else {
                    // TLAS leaf: up to 4 instances; all but the first go back on the stack as
                    // single-instance leaves, the first is entered now
                    for (uint32_t k = cnt - 1u; k >= 1u; k--) push(PT_LEAF | (first + k), cur_t);
                    cur_ipos = first;
                    cur_iid = inst_id[first];
                    const float4 r0 = inst6[6 * (size_t)first + 3], r1 = inst6[6 * (size_t)first + 4],
                                 r2 = inst6[6 * (size_t)first + 5];
                    const ptm::f3 oo = { ((r0.x * org_w.x + r0.y * org_w.y) + r0.z * org_w.z) + r0.w,
                                         ((r1.x * org_w.x + r1.y * org_w.y) + r1.z * org_w.z) + r1.w,
                                         ((r2.x * org_w.x + r2.y * org_w.y) + r2.z * org_w.z) + r2.w };
                    const ptm::f3 od = { (r0.x * dir_w.x + r0.y * dir_w.y) + r0.z * dir_w.z,
                                         (r1.x * dir_w.x + r1.y * dir_w.y) + r1.z * dir_w.z,
                                         (r2.x * dir_w.x + r2.y * dir_w.y) + r2.z * dir_w.z };
                    inv = { ptm::safe_inv(od.x), ptm::safe_inv(od.y), ptm::safe_inv(od.z) };
                    slab_setup(oo, inv, invf, on, of);
                    ax = inv.x < 0.f ? 48u : 0u;
                    ay = inv.y < 0.f ? 48u : 0u;
                    az = inv.z < 0.f ? 48u : 0u;
                    pre = ptm::ray_setup(oo, od);
                    if (LDS_BLAS) {
                        tri_base = (uint32_t)pre.kz * 3u * n_tris;
                        orgp = { ptm::sel3(pre.kz, oo.y, oo.z, oo.x), ptm::sel3(pre.kz, oo.z, oo.x, oo.y),
                                 ptm::sel3(pre.kz, oo.x, oo.y, oo.z) };
                    }
                    push(EXIT_MARK, 0.f);
                    in_blas = true;
                    cur = 0u;  // BLAS root
                    cur_t = tmin;
                }
            }
            if (cur == SENTINEL) {
                const bool miss = best_pos == PT_MISS;
                // raw_hit (render path): (V, W, det) go out undivided and k_shade takes the two quotients at
                // full lane occupancy; here they would run once per finishing lane group
                hit[q] = raw_hit ? make_float4(__uint_as_float(best_pos), best_V, best_W, best_det)
                                 : make_float4(__uint_as_float(best_pos), miss ? 0.f : best_t,
                                               miss ? 0.f : ptm::fdiv(best_V, best_det), miss ? 0.f : ptm::fdiv(best_W, best_det));
                if (!SHADOW) hit_inst[q] = best_ipos;
                have = false;
            }
        }
    }
    if (COUNT) {
        for (int o = 32; o > 0; o >>= 1) {
            c_nodes += __shfl_xor(c_nodes, o, 64);
            c_tris += __shfl_xor(c_tris, o, 64);
        }
        if (lane == 0 && stats) {
            atomicAdd(stats + 2, c_nodes);
            atomicAdd(stats + 3, c_tris);
        }
    }
}

}  // namespace
