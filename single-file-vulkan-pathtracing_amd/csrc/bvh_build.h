// bvh_build.h -- what the acceleration-structure builders share (lbvh_build.hip, ploc_build.hip, scene_build.hip).
#pragma once
#include "pt_internal.h"
#include "pt_math.h"

namespace {

constexpr int TB = 256;

// ---- float <-> order-preserving uint (for atomicMin/Max on floats) --------------------------
__device__ __forceinline__ uint32_t f2ord(float f)
{
    const uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__host__ __device__ __forceinline__ float ord2f(uint32_t u)
{
    const uint32_t b = (u & 0x80000000u) ? (u & 0x7FFFFFFFu) : ~u;
#ifdef __HIP_DEVICE_COMPILE__
    return __uint_as_float(b);
#else
    float f;
    __builtin_memcpy(&f, &b, 4);
    return f;
#endif
}

__device__ __forceinline__ float wave_min(float v)
{
    for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ float wave_max(float v)
{
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

__device__ __forceinline__ float box_area(const float4 lo, const float4 hi)
{
    const float x = hi.x - lo.x, y = hi.y - lo.y, z = hi.z - lo.z;
    return (x * y + y * z) + z * x;
}


constexpr int PLOC_R_MAX = 32;  // PLOC's search radius is a run-time choice (pt_tuning.ploc_radius, default 8) up to this

template <typename T>
struct DevBuf {
    T *p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    hipError_t alloc(size_t n) { return hipMalloc((void **)&p, sizeof(T) * (n ? n : 1)); }
    T *release() { T *q = p; p = nullptr; return q; }
};

}  // namespace

// ---- generic part: n boxes (tlo/thi on the device) -> sorted order, binary LBVH, BVH4 ----------
struct BvhOut {
    unsigned long long *d_keys = nullptr;  // sorted Morton keys           (caller owns)
    uint32_t *d_prim_of = nullptr;         // sorted position -> box id
    uint32_t *d_prim_q = nullptr;          // PLOC: leaf position of the rebuilt tree -> box id (null: the LBVH is the tree)
    float4 *d_nodes = nullptr;             // binary nodes, 64 B
    float4 *d_wide = nullptr;              // BVH4 nodes, 128 B
    uint32_t n_nodes = 0, n_wide = 0, height = 0, height_tree = 0;  // height: of the LBVH; height_tree: of the tree the collapses ran on
    uint32_t stack_need = 0;               // most entries a depth-first walk of the BVH4 can have pending
    float bmin[3]{}, bmax[3]{};
    // BVH8 (want8): 128-B nodes, the triangle order that goes with them (position -> sorted position), levels
    uint4 *d_wide8 = nullptr;
    uint32_t *d_order8 = nullptr;
    uint32_t n_wide8 = 0, levels8 = 0;
    uint4 *d_wide16t = nullptr;            // BVH4, 64-B nodes, built top-down with contiguous children (k_w4_emit)
    uint32_t n_wide16t = 0, levels4t = 0;
    float norm_c[3]{}, norm_s[3]{1.f, 1.f, 1.f}, norm_rs[3]{1.f, 1.f, 1.f};
    double area_lbvh = 0.0, area_ploc = 0.0, area_tree = 0.0;  // sums of the internal nodes' surface areas: LBVH, PLOC rebuild (0: not built), the tree kept
};


// lbvh_build.hip: n boxes -> sorted order, binary tree (LBVH, or its PLOC rebuild), BVH4 in both node formats, optionally the
// 8-wide nodes.  top_down: bit 0 = also the 8-wide tree (+ its triangle order), bit 1 = also the top-down BVH4 in the 64-B format,
// bit 2 = that BVH4 with 16-bit child codes (the TLAS of k_extend_inst16; needs n < 32768).  ploc: the binary tree is rebuilt by
// PLOC before the collapses (out.d_prim_q = its leaf order; out.d_prim_of, d_keys and d_nodes stay the LBVH's, for the read-back).
pt_status ptb_build_bvh(pt_ctx *ctx, const float4 *d_tlo, const float4 *d_thi, uint32_t n, uint32_t leaf_max, BvhOut &out, int top_down = 0,
                        bool ploc = false);
void ptb_norm_box(const float *bmin, const float *bmax, float *c, float *sv, float *rs);
// ploc_build.hip
pt_status ptb_tree_area(pt_ctx *ctx, uint32_t n, const float4 *d_blo, const float4 *d_bhi, double *out);
pt_status ptb_ploc_refine(pt_ctx *ctx, uint32_t n, int radius, double area_lbvh, double *area_ploc, uint2 *d_topo, uint2 *d_range,
                          uint32_t *d_pint, uint32_t *d_pleaf, float4 *d_blo, float4 *d_bhi, const uint32_t *d_prim_of, uint32_t *d_prim_q,
                          uint32_t *d_sums, uint32_t *h_height);
