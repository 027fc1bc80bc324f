// pair_leaf.h -- ONE source of the fan-pair leaf step of the scenes that live in LDS: the watertight test (Woop, Benthin, Wald, JCGT 2013)
// of a leaf that is one triangle (v0, v1, v2) or a fan pair (v0, v1, v2) + (v0, v2, v3), with the sheared vertices and the products of the
// shared edge v0-v2 computed once, the true divide, the range check tmin < t < tmax -- and the two closest-hit rules that go with it.
// Stands in for what the driver does behind traceRayEXT (raygen.rgen:63-75) for opaque, un-culled triangles (main.cpp:508, 525, 536).
//
// Four kernels instantiate it: k_extend (PAIRS: k_extend_lds7p) and k_fused for single-level scenes, k_extend_inst16 and k_fused_inst for
// two-level ones.  Until round 5 each of them carried its own copy of these lines, kept equal by the parity tests only.
#pragma once
#include "pt_math.h"

namespace ptl {

// tri4: the permuted triangle records in LDS -- {v0, id} {v1} {v2} per triangle, a pair's second triangle behind the first, so the pair's
// fourth vertex is record 5 (.w = the second triangle's gl_PrimitiveID); ti: index of the first record; first: the leaf's first position.
// accept(t, V, W, det, pos, prim_bits): a hit inside the range, in primitive order (first half first); hit_block(): once per divide block.
template <bool LOAD_D_FIRST = false, class Accept, class HitBlock>
__device__ __forceinline__ void pair_leaf_test(const float4 *tri4, size_t ti, bool two, uint32_t first, const ptm::RayPre &pre,
                                               const ptm::f3 &orgp, float tmin, float tmax, Accept &&accept, HitBlock &&hit_block)
{
    const float4 a = tri4[ti + 0], b = tri4[ti + 1], c = tri4[ti + 2];
    // LOAD_D_FIRST: the pair's fourth vertex is read WITH the first three instead of behind `two` -- one LDS round trip per leaf step instead of
    // two (k_fused: -0.5 %, profiles/r06y_leaf_fourth_vertex_first.log).  A single triangle's step then reads 16 bytes of the record behind it:
    // only for callers whose triangle records are followed by more of their own LDS (the fused kernel: its shade table).
    float4 d = make_float4(0.f, 0.f, 0.f, 0.f);
    if (LOAD_D_FIRST) {
        d = tri4[ti + 5];
        asm volatile("" : "+v"(d.x), "+v"(d.y), "+v"(d.z), "+v"(d.w));  // (keeps the load here: the compiler sinks it back behind the branch)
    }
    // the sheared vertices and the products of the edge v0-v2 serve both halves (ptm::tri_test_perm, same operands in the same order:
    // bit-identical numerators)
    const float Az_ = a.z - orgp.z, Bz_ = b.z - orgp.z, Cz_ = c.z - orgp.z;
    const float Ax = (a.x - orgp.x) - pre.Sx * Az_, Ay = (a.y - orgp.y) - pre.Sy * Az_;
    const float Bx = (b.x - orgp.x) - pre.Sx * Bz_, By = (b.y - orgp.y) - pre.Sy * Bz_;
    const float Cx = (c.x - orgp.x) - pre.Sx * Cz_, Cy = (c.y - orgp.y) - pre.Sy * Cz_;
    const float pAC = Ax * Cy, qAC = Ay * Cx;
    // edge test of one half: inside (no strictly negative AND strictly positive edge function) and not edge-on
    // (no short-circuit evaluation: each `&&` was a branch -- exec-mask bookkeeping on the scalar unit for instructions the wave runs anyway)
    auto inside = [](float U, float V, float W) {
        const bool neg = fminf(fminf(U, V), W) < 0.0f, pos = fmaxf(fmaxf(U, V), W) > 0.0f, nz = ((U + V) + W) != 0.0f;
        return bool(!(neg & pos) & nz);
    };
    auto finish = [&](float U, float V, float W, float z0, float z1, float z2, uint32_t pos, uint32_t prim) {
        const float det = (U + V) + W;
        hit_block();
        const float T = (U * (pre.Sz * z0) + V * (pre.Sz * z1)) + W * (pre.Sz * z2);
        const float t = ptm::fdiv(T, det);
        // outside the range: a NaN goes on, which is neither closer than nor equal to anything (one select instead of two branches)
        accept(((t > tmin) & (t < tmax)) ? t : __builtin_nanf(""), V, W, det, pos, prim);
    };
    const float UA = Cx * By - Cy * Bx, VA = pAC - qAC, WA = Bx * Ay - By * Ax;
    const bool inA = inside(UA, VA, WA);
    float UB = 0.f, VB = 0.f, WB = 0.f, Dz_ = 0.f;
    uint32_t primB = 0u;
    bool inB = false;
    if (LOAD_D_FIRST || two) {  // (LOAD_D_FIRST: the second half is computed for every lane -- nearly every wave holds a pair, so the branch only cost its bookkeeping)
        if (!LOAD_D_FIRST) d = tri4[ti + 5];  // third vertex of the second half; .w = its primitive id (k_pack)
        Dz_ = d.z - orgp.z;
        const float Dx = (d.x - orgp.x) - pre.Sx * Dz_, Dy = (d.y - orgp.y) - pre.Sy * Dz_;
        // (v0, v2, v3): U = Dx*Cy - Dy*Cx, V = Ax*Dy - Ay*Dx, W = Cx*Ay - Cy*Ax = qAC - pAC
        UB = Dx * Cy - Dy * Cx; VB = Ax * Dy - Ay * Dx; WB = qAC - pAC;
        primB = __float_as_uint(d.w);
        inB = bool(two & inside(UB, VB, WB));
    }
    // A wave nearly always holds lanes inside the first half AND lanes inside the second, so two separate divide blocks both ran in 95 %
    // of the steps, each for a handful of lanes.  One block serves both: a lane inside the second half only brings that half's operands;
    // the lane inside BOTH (a ray through the shared diagonal, a folded quad) takes the first half here and the second in a block of its
    // own, in primitive order.  Same operations on the same operands: same bits.
    if (inA || inB) {
        const bool sb = !inA;
        finish(sb ? UB : UA, sb ? VB : VA, sb ? WB : WA, Az_, sb ? Cz_ : Bz_, sb ? Dz_ : Cz_, sb ? first + 1u : first, sb ? primB : __float_as_uint(a.w));
    }
    if (inA && inB) finish(UB, VB, WB, Az_, Cz_, Dz_, first + 1u, primB);
}

// Single-level rule: closest t; equal t -> lowest gl_PrimitiveID (the OBJ has coincident quads).  The ids of the two rivals are read when it
// happens (the third vertex of every record carries its triangle's id, k_pack), not kept.  -> true: the hit replaced the best one.
__device__ __forceinline__ bool closer_single_level(const float4 *tri4, uint32_t tri_base, float t, float V, float W, float det, uint32_t pos,
                                                    float &best_t, float &best_V, float &best_W, float &best_det, uint32_t &best_pos)
{
    bool closer = t < best_t;
    if (closer) {
        best_t = t; best_V = V; best_W = W; best_det = det; best_pos = pos;
    } else if (t == best_t) {  // (rare, and apart from the common path: merged into `closer` it cost every divide block seven scalar instructions)
        if (best_pos == PT_MISS || __float_as_uint(tri4[(size_t)tri_base + 3 * (size_t)pos + 2].w) <
                                       __float_as_uint(tri4[(size_t)tri_base + 3 * (size_t)best_pos + 2].w)) {
            best_t = t; best_V = V; best_W = W; best_det = det; best_pos = pos;
            closer = true;
        }
    }
    return closer;
}

// Two-level rule: closest t; equal t -> lowest (gl_InstanceID, gl_PrimitiveID).
__device__ __forceinline__ bool closer_instanced(float t, float V, float W, float det, uint32_t pos, uint32_t prim, uint32_t cur_ipos, uint32_t cur_iid,
                                                 float &best_t, float &best_V, float &best_W, float &best_det, uint32_t &best_pos, uint32_t &best_prim,
                                                 uint32_t &best_ipos, uint32_t &best_iid)
{
    if (t < best_t || (t == best_t && (cur_iid < best_iid || (cur_iid == best_iid && prim < best_prim)))) {
        best_t = t; best_V = V; best_W = W; best_det = det; best_pos = pos; best_prim = prim;
        best_ipos = cur_ipos; best_iid = cur_iid;
        return true;
    }
    return false;
}

}  // namespace ptl
