// lbvh_build.hip -- on-device LBVH construction (gfx950).
//
// Replaces the driver's acceleration-structure build of the reference (Accel::Accel,
// main.cpp:414-455, called for the BLAS at main.cpp:497-512 and the one-instance TLAS at
// main.cpp:515-538).  Pipeline, all on the GPU:
//   1. k_gather    de-index triangles (closesthit.rchit:52-54 semantics), per-triangle AABB,
//                  scene AABB by wave/block reduction + ordered-int atomics
//   2. k_morton    63-bit Morton key of the AABB centre (21 bits/axis, x most significant)
//   3. radix sort  LSD, 8 passes x 8-bit digits, stable (ties keep prim-id order)
//   4. k_karras    Karras 2012 hierarchy (duplicate keys disambiguated by position)
//   5. k_refit     bottom-up boxes with one arrival counter per node
//   6. k_pack      leaf-ordered triangle + shading records for the traversal / shade kernels
// The tree is fully determined by the input, so a CPU builder following the same rules yields
// bit-identical keys, order, topology and boxes (tests compare them).
#include "pt_internal.h"
#include "pt_math.h"

#include <hip/hip_fp16.h>

#include "device_scan.h"  // block_exclusive_scan, k_scan_*, exclusive_scan

#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

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

// 1. gather: de-indexed triangles + their boxes.  tri_orig: 3 float4 per triangle in prim-id order.
__global__ __launch_bounds__(TB) void k_gather(const float *__restrict__ vertices, const uint32_t *__restrict__ indices,
                                               uint32_t n_tris, float4 *__restrict__ tri_orig,
                                               float4 *__restrict__ tlo, float4 *__restrict__ thi)
{
    const uint32_t t = blockIdx.x * TB + threadIdx.x;
    float mn[3] = { INFINITY, INFINITY, INFINITY }, mx[3] = { -INFINITY, -INFINITY, -INFINITY };
    if (t < n_tris) {
        float v[3][3];
        for (int c = 0; c < 3; c++) {
            const uint32_t vi = indices[3 * (size_t)t + c];
            for (int k = 0; k < 3; k++) v[c][k] = vertices[3 * (size_t)vi + k];
        }
        for (int k = 0; k < 3; k++) {
            mn[k] = fminf(fminf(v[0][k], v[1][k]), v[2][k]);
            mx[k] = fmaxf(fmaxf(v[0][k], v[1][k]), v[2][k]);
        }
        tri_orig[3 * (size_t)t + 0] = make_float4(v[0][0], v[0][1], v[0][2], __uint_as_float(t));
        tri_orig[3 * (size_t)t + 1] = make_float4(v[1][0], v[1][1], v[1][2], 0.f);
        tri_orig[3 * (size_t)t + 2] = make_float4(v[2][0], v[2][1], v[2][2], 0.f);
        tlo[t] = make_float4(mn[0], mn[1], mn[2], 0.f);
        thi[t] = make_float4(mx[0], mx[1], mx[2], 0.f);
    }
}

__device__ __forceinline__ unsigned long long expand21(uint32_t v)
{
    unsigned long long x = v & 0x1FFFFFu;
    x = (x | x << 32) & 0x1F00000000FFFFull;
    x = (x | x << 16) & 0x1F0000FF0000FFull;
    x = (x | x << 8) & 0x100F00F00F00F00Full;
    x = (x | x << 4) & 0x10C30C30C30C30C3ull;
    x = (x | x << 2) & 0x1249249249249249ull;
    return x;
}

__device__ __forceinline__ uint32_t quant21(float c, float lo, float ext)
{
    const float n = ext > 0.0f ? ptm::fdiv(c - lo, ext) : 0.0f;
    float q = n * 2097152.0f;
    if (!(q >= 0.0f)) q = 0.0f;
    if (q > 2097151.0f) q = 2097151.0f;
    return (uint32_t)q;
}

// 2. Morton keys
__global__ __launch_bounds__(TB) void k_morton(const float4 *__restrict__ tlo, const float4 *__restrict__ thi,
                                               uint32_t n_tris, const uint32_t *__restrict__ scene_ord,
                                               unsigned long long *__restrict__ keys, uint32_t *__restrict__ vals)
{
    const uint32_t t = blockIdx.x * TB + threadIdx.x;
    if (t >= n_tris) return;
    const float lo[3] = { ord2f(scene_ord[0]), ord2f(scene_ord[1]), ord2f(scene_ord[2]) };
    const float hi[3] = { ord2f(scene_ord[3]), ord2f(scene_ord[4]), ord2f(scene_ord[5]) };
    const float4 a = tlo[t], b = thi[t];
    const uint32_t qx = quant21((a.x + b.x) * 0.5f, lo[0], hi[0] - lo[0]);
    const uint32_t qy = quant21((a.y + b.y) * 0.5f, lo[1], hi[1] - lo[1]);
    const uint32_t qz = quant21((a.z + b.z) * 0.5f, lo[2], hi[2] - lo[2]);
    keys[t] = (expand21(qx) << 2) | (expand21(qy) << 1) | expand21(qz);
    vals[t] = t;
}

// 3. radix sort: one pass = hist -> scan -> scatter.  A block owns a tile of 2048 keys, each of
// its 4 waves a contiguous 512-key run (so (wave, item, lane) order == index order == stable).
constexpr int RS_KPT = 8;
constexpr int RS_TILE = TB * RS_KPT;

__global__ __launch_bounds__(TB) void k_rs_hist(const unsigned long long *__restrict__ keys, uint32_t n, int shift,
                                                uint32_t *__restrict__ hist, uint32_t nblocks)
{
    __shared__ uint32_t h[256];
    h[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t base = blockIdx.x * RS_TILE;
    for (int k = 0; k < RS_KPT; k++) {
        const uint32_t i = base + k * TB + threadIdx.x;
        if (i < n) atomicAdd(&h[(uint32_t)(keys[i] >> shift) & 255u], 1u);
    }
    __syncthreads();
    hist[(size_t)threadIdx.x * nblocks + blockIdx.x] = h[threadIdx.x];
}

__global__ __launch_bounds__(TB) void k_rs_scatter(const unsigned long long *__restrict__ kin,
                                                   const uint32_t *__restrict__ vin,
                                                   unsigned long long *__restrict__ kout, uint32_t *__restrict__ vout,
                                                   uint32_t n, int shift, const uint32_t *__restrict__ offs,
                                                   uint32_t nblocks)
{
    __shared__ uint32_t wh[4][256];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int w = 0; w < 4; w++) wh[w][threadIdx.x] = 0;
    __syncthreads();
    const uint32_t base = blockIdx.x * RS_TILE + wave * (RS_TILE / 4);
    unsigned long long key[RS_KPT];
    uint32_t val[RS_KPT];
    for (int k = 0; k < RS_KPT; k++) {
        const uint32_t i = base + k * 64 + lane;
        const bool ok = i < n;
        key[k] = ok ? kin[i] : 0ull;
        val[k] = ok ? vin[i] : 0u;
        if (ok) atomicAdd(&wh[wave][(uint32_t)(key[k] >> shift) & 255u], 1u);
    }
    __syncthreads();
    {
        const int d = threadIdx.x;
        uint32_t run = offs[(size_t)d * nblocks + blockIdx.x];
        for (int w = 0; w < 4; w++) {
            const uint32_t c = wh[w][d];
            wh[w][d] = run;
            run += c;
        }
    }
    __syncthreads();
    const unsigned long long lt = (1ull << lane) - 1ull;
    for (int k = 0; k < RS_KPT; k++) {
        const uint32_t i = base + k * 64 + lane;
        const bool ok = i < n;
        const uint32_t d = (uint32_t)(key[k] >> shift) & 255u;
        unsigned long long m = __ballot(ok);
        for (int b = 0; b < 8; b++) {
            const bool bit = (d >> b) & 1u;
            const unsigned long long bb = __ballot(bit);
            m &= bit ? bb : ~bb;
        }
        const uint32_t rank = __popcll(m & lt);
        uint32_t pos = 0;
        if (ok) pos = wh[wave][d] + rank;
        __syncthreads();
        if (ok && rank == 0) wh[wave][d] += __popcll(m);
        __syncthreads();
        if (ok) {
            kout[pos] = key[k];
            vout[pos] = val[k];
        }
    }
}

// 4. Karras 2012
__device__ __forceinline__ int kdelta(const unsigned long long *__restrict__ keys, int n, int i, int j)
{
    if (j < 0 || j >= n) return -1;
    const unsigned long long a = keys[i], b = keys[j];
    if (a == b) return 64 + __clz((uint32_t)i ^ (uint32_t)j);
    return __clzll(a ^ b);
}

__global__ __launch_bounds__(TB) void k_karras(const unsigned long long *__restrict__ keys, int n,
                                               uint2 *__restrict__ topo, uint32_t *__restrict__ parent_int,
                                               uint32_t *__restrict__ parent_leaf, uint2 *__restrict__ range)
{
    const int i = blockIdx.x * TB + threadIdx.x;
    if (i >= n - 1) return;
    const int d = (kdelta(keys, n, i, i + 1) - kdelta(keys, n, i, i - 1)) < 0 ? -1 : 1;
    const int dmin = kdelta(keys, n, i, i - d);
    int lmax = 2;
    while (kdelta(keys, n, i, i + lmax * d) > dmin) lmax *= 2;
    int l = 0;
    for (int t = lmax / 2; t >= 1; t /= 2)
        if (kdelta(keys, n, i, i + (l + t) * d) > dmin) l += t;
    const int j = i + l * d;
    const int dnode = kdelta(keys, n, i, j);
    int sp = 0, t = l;
    do {
        t = (t + 1) >> 1;
        if (kdelta(keys, n, i, i + (sp + t) * d) > dnode) sp += t;
    } while (t > 1);
    const int gamma = i + sp * d + (d < 0 ? -1 : 0);
    const int lo = min(i, j), hi = max(i, j);
    uint32_t left, right;
    if (lo == gamma) { left = PT_LEAF | (uint32_t)gamma; parent_leaf[gamma] = (uint32_t)i; }
    else { left = (uint32_t)gamma; parent_int[gamma] = (uint32_t)i; }
    if (hi == gamma + 1) { right = PT_LEAF | (uint32_t)(gamma + 1); parent_leaf[gamma + 1] = (uint32_t)i; }
    else { right = (uint32_t)(gamma + 1); parent_int[gamma + 1] = (uint32_t)i; }
    topo[i] = make_uint2(left, right);
    range[i] = make_uint2((uint32_t)lo, (uint32_t)hi);  // sorted positions covered by this node
}

// leaf pad = 2^-18 of the scene scale: keeps the slab test conservative w.r.t. the rounded
// watertight triangle test
__device__ __forceinline__ float leaf_pad(const uint32_t *__restrict__ scene_ord)
{
    float scale = 0.f;
    for (int k = 0; k < 6; k++) scale = fmaxf(scale, fabsf(ord2f(scene_ord[k])));
    return scale * 3.814697265625e-06f;
}

// 5. refit.  box arrays: index pos for leaves, n + node for internal nodes; .w of lo = height bits
__global__ __launch_bounds__(TB) void k_refit(const float4 *__restrict__ tlo, const float4 *__restrict__ thi,
                                              const uint32_t *__restrict__ prim_of, int n,
                                              const uint2 *__restrict__ topo, const uint32_t *__restrict__ parent_int,
                                              const uint32_t *__restrict__ parent_leaf, float4 *box_lo, float4 *box_hi,
                                              uint32_t *flags, const uint32_t *__restrict__ scene_ord,
                                              float4 *__restrict__ nodes, uint32_t *__restrict__ height_out)
{
    const int pos = blockIdx.x * TB + threadIdx.x;
    if (pos >= n) return;
    {
        const float pad = leaf_pad(scene_ord);
        const uint32_t prim = prim_of[pos];
        const float4 a = tlo[prim], b = thi[prim];
        box_lo[pos] = make_float4(a.x - pad, a.y - pad, a.z - pad, __uint_as_float(0u));
        box_hi[pos] = make_float4(b.x + pad, b.y + pad, b.z + pad, 0.f);
    }
    uint32_t node = parent_leaf[pos];
    for (;;) {
        // publish my subtree's box, then arrive (agent-scope release; the explicit vmcnt wait
        // keeps the arrival from overtaking the write-back)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const uint32_t old = atomicAdd(&flags[node], 1u);
        if (old == 0u) return;  // sibling subtree not finished: its last thread continues
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");  // acquire the sibling's box
        const uint2 ch = topo[node];
        const size_t li = (ch.x & PT_LEAF) ? (size_t)(ch.x & ~PT_LEAF) : (size_t)n + ch.x;
        const size_t ri = (ch.y & PT_LEAF) ? (size_t)(ch.y & ~PT_LEAF) : (size_t)n + ch.y;
        const float4 llo = box_lo[li], lhi = box_hi[li], rlo = box_lo[ri], rhi = box_hi[ri];
        const uint32_t h = 1u + max(__float_as_uint(llo.w), __float_as_uint(rlo.w));
        nodes[4 * (size_t)node + 0] = make_float4(llo.x, llo.y, llo.z, lhi.x);
        nodes[4 * (size_t)node + 1] = make_float4(lhi.y, lhi.z, rlo.x, rlo.y);
        nodes[4 * (size_t)node + 2] = make_float4(rlo.z, rhi.x, rhi.y, rhi.z);
        nodes[4 * (size_t)node + 3] = make_float4(__uint_as_float(ch.x), __uint_as_float(ch.y), 0.f, 0.f);
        box_lo[(size_t)n + node] = make_float4(fminf(llo.x, rlo.x), fminf(llo.y, rlo.y), fminf(llo.z, rlo.z),
                                               __uint_as_float(h));
        box_hi[(size_t)n + node] = make_float4(fmaxf(lhi.x, rhi.x), fmaxf(lhi.y, rhi.y), fmaxf(lhi.z, rhi.z), 0.f);
        if (node == 0u) { *height_out = h; return; }
        node = parent_int[node];
    }
}

// n == 1: a root whose two children are the same leaf (tested twice, same result)
__global__ void k_single(const float4 *__restrict__ tlo, const float4 *__restrict__ thi,
                         const uint32_t *__restrict__ scene_ord, float4 *__restrict__ nodes,
                         uint32_t *__restrict__ height_out)
{
    const float pad = leaf_pad(scene_ord);
    const float4 a = tlo[0], b = thi[0];
    const float lx = a.x - pad, ly = a.y - pad, lz = a.z - pad, hx = b.x + pad, hy = b.y + pad, hz = b.z + pad;
    nodes[0] = make_float4(lx, ly, lz, hx);
    nodes[1] = make_float4(hy, hz, lx, ly);
    nodes[2] = make_float4(lz, hx, hy, hz);
    nodes[3] = make_float4(__uint_as_float(PT_LEAF), __uint_as_float(PT_LEAF), 0.f, 0.f);
    *height_out = 1u;
}


// 7. BVH4 collapse.  Binary nodes at even depth whose subtree holds more than leaf_max primitives
// become wide nodes; their internal children (odd depth) are absorbed, so a wide node holds the
// up-to-4 grandchildren.  Any binary subtree with <= leaf_max primitives becomes ONE leaf child
// (its triangles are contiguous in sorted order): child word = LEAF | (count-1)<<28 | first.
// Wide node = 128 B = one gfx950 L2 line: 6 float4 {lo.x[4]} {lo.y[4]} {lo.z[4]} {hi.x[4]} {hi.y[4]}
// {hi.z[4]}, 1 uint4 children (0xFFFFFFFF = empty slot, box lo = hi = +inf: never hit), 1 spare.
__device__ __forceinline__ bool leaf_like(uint32_t ref, const uint2 *__restrict__ range, uint32_t leaf_max)
{
    if (ref & PT_LEAF) return true;
    const uint2 r = range[ref];
    return r.y - r.x + 1u <= leaf_max;
}

__global__ __launch_bounds__(TB) void k_wide_flag(int n_int, const uint32_t *__restrict__ parent_int,
                                                  const uint2 *__restrict__ range, uint32_t *__restrict__ flag,
                                                  uint32_t leaf_max)
{
    const int i = blockIdx.x * TB + threadIdx.x;
    if (i >= n_int) return;
    uint32_t depth = 0;
    for (uint32_t a = (uint32_t)i; a != 0u; a = parent_int[a]) depth++;
    const uint2 r = range[i];
    const bool big = r.y - r.x + 1u > leaf_max;
    flag[i] = (i == 0 || (big && (depth & 1u) == 0u)) ? 1u : 0u;
}

__device__ __forceinline__ void wide_child(uint32_t ref, int n, const uint2 *__restrict__ range,
                                           const uint32_t *__restrict__ widx, const float4 *__restrict__ box_lo,
                                           const float4 *__restrict__ box_hi, uint32_t leaf_max, uint32_t &word,
                                           float4 &lo, float4 &hi)
{
    if (ref & PT_LEAF) {
        const uint32_t pos = ref & ~PT_LEAF;
        word = PT_LEAF | pos;  // count 1
        lo = box_lo[pos];
        hi = box_hi[pos];
        return;
    }
    const uint2 r = range[ref];
    const uint32_t cnt = r.y - r.x + 1u;
    word = cnt <= leaf_max ? (PT_LEAF | ((cnt - 1u) << 28) | r.x) : widx[ref];
    lo = box_lo[(size_t)n + ref];
    hi = box_hi[(size_t)n + ref];
}

__global__ __launch_bounds__(TB) void k_wide_emit(int n, int n_int, const uint2 *__restrict__ topo,
                                                  const uint2 *__restrict__ range, const uint32_t *__restrict__ flag,
                                                  const uint32_t *__restrict__ widx, const float4 *__restrict__ box_lo,
                                                  const float4 *__restrict__ box_hi, float4 *__restrict__ wide,
                                                  uint32_t leaf_max)
{
    const int i = blockIdx.x * TB + threadIdx.x;
    if (i >= n_int || !flag[i]) return;
    uint32_t word[4] = { PT_MISS, PT_MISS, PT_MISS, PT_MISS };
    float4 lo[4], hi[4];
    for (int k = 0; k < 4; k++) {
        lo[k] = make_float4(INFINITY, INFINITY, INFINITY, 0.f);  // empty slot: lo = hi = +inf, every slab
        hi[k] = make_float4(INFINITY, INFINITY, INFINITY, 0.f);  // test sees an empty interval
    }
    int m = 0;
    if (leaf_like((uint32_t)i, range, leaf_max)) {  // tiny scene: the root itself is one leaf
        wide_child((uint32_t)i, n, range, widx, box_lo, box_hi, leaf_max, word[0], lo[0], hi[0]);
        m = 1;
    } else {
        const uint2 ch = topo[i];
        const uint32_t c2[2] = { ch.x, ch.y };
        for (int a = 0; a < 2; a++) {
            if (leaf_like(c2[a], range, leaf_max)) {
                wide_child(c2[a], n, range, widx, box_lo, box_hi, leaf_max, word[m], lo[m], hi[m]);
                m++;
            } else {  // absorbed odd-depth node: its two children move up
                const uint2 g = topo[c2[a]];
                wide_child(g.x, n, range, widx, box_lo, box_hi, leaf_max, word[m], lo[m], hi[m]);
                m++;
                wide_child(g.y, n, range, widx, box_lo, box_hi, leaf_max, word[m], lo[m], hi[m]);
                m++;
            }
        }
    }
    float4 *dst = wide + 8 * (size_t)widx[i];
    dst[0] = make_float4(lo[0].x, lo[1].x, lo[2].x, lo[3].x);
    dst[1] = make_float4(lo[0].y, lo[1].y, lo[2].y, lo[3].y);
    dst[2] = make_float4(lo[0].z, lo[1].z, lo[2].z, lo[3].z);
    dst[3] = make_float4(hi[0].x, hi[1].x, hi[2].x, hi[3].x);
    dst[4] = make_float4(hi[0].y, hi[1].y, hi[2].y, hi[3].y);
    dst[5] = make_float4(hi[0].z, hi[1].z, hi[2].z, hi[3].z);
    dst[6] = make_float4(__uint_as_float(word[0]), __uint_as_float(word[1]), __uint_as_float(word[2]),
                         __uint_as_float(word[3]));
    dst[7] = make_float4(0.f, 0.f, 0.f, 0.f);
}

// n == 1: one wide node with a single one-triangle leaf
__global__ void k_wide_single(const float4 *__restrict__ nodes, float4 *__restrict__ wide)
{
    const float4 n0 = nodes[0], n1 = nodes[1];
    const float inf = INFINITY;
    wide[0] = make_float4(n0.x, inf, inf, inf);
    wide[1] = make_float4(n0.y, inf, inf, inf);
    wide[2] = make_float4(n0.z, inf, inf, inf);
    wide[3] = make_float4(n0.w, inf, inf, inf);
    wide[4] = make_float4(n1.x, inf, inf, inf);
    wide[5] = make_float4(n1.y, inf, inf, inf);
    wide[6] = make_float4(__uint_as_float(PT_LEAF | 0u), __uint_as_float(PT_MISS), __uint_as_float(PT_MISS),
                          __uint_as_float(PT_MISS));
    wide[7] = make_float4(0.f, 0.f, 0.f, 0.f);
}

// 6. leaf-ordered records
__global__ __launch_bounds__(TB) void k_pack(const float4 *__restrict__ tri_orig, const float *__restrict__ faces,
                                             const uint32_t *__restrict__ prim_of, uint32_t n,
                                             float4 *__restrict__ tri4, float4 *__restrict__ shade4,
                                             float4 *__restrict__ shade64, float4 *__restrict__ ke4,
                                             float4 *__restrict__ frame4 = nullptr)
{
    const uint32_t pos = blockIdx.x * TB + threadIdx.x;
    if (pos >= n) return;
    const uint32_t prim = prim_of[pos];
    const float4 a = tri_orig[3 * (size_t)prim + 0], b = tri_orig[3 * (size_t)prim + 1],
                 c = tri_orig[3 * (size_t)prim + 2];
    tri4[3 * (size_t)pos + 0] = a;  // .w = bits(prim)
    tri4[3 * (size_t)pos + 1] = b;
    tri4[3 * (size_t)pos + 2] = make_float4(c.x, c.y, c.z, a.w);  // .w = bits(prim) again: the pair-leaf test of k_extend
                                                                  // reads only this vertex of a quad's second triangle
    // closesthit.rchit:43-48 normal (never flipped), :60 brdf = Kd / pi (true divide), :61 emission
    const ptm::f3 nrm = ptm::tri_normal({ a.x, a.y, a.z }, { b.x, b.y, b.z }, { c.x, c.y, c.z });
    const float *f = faces + 6 * (size_t)prim;
    const float br = ptm::fdiv(f[0], 3.1415927410125732f), bg = ptm::fdiv(f[1], 3.1415927410125732f),
                bb = ptm::fdiv(f[2], 3.1415927410125732f);
    shade4[3 * (size_t)pos + 0] = make_float4(nrm.x, nrm.y, nrm.z, br);
    shade4[3 * (size_t)pos + 1] = make_float4(bg, bb, f[3], f[4]);
    shade4[3 * (size_t)pos + 2] = make_float4(f[5], 0.f, 0.f, 0.f);
    if (frame4) {  // raygen.rgen:14-21 for this triangle's normal: {T.xyz, B.x} {B.yz, 0, 0}
        ptm::f3 T, B;
        ptm::tangent_frame(nrm, T, B);
        frame4[2 * (size_t)pos + 0] = make_float4(T.x, T.y, T.z, B.x);
        frame4[2 * (size_t)pos + 1] = make_float4(B.y, B.z, 0.f, 0.f);
    }
    // the same values regrouped for scenes whose tables stay in HBM (k_shade<.., false>): one 64-B record instead of
    // two 48-B ones (4 divergent 16-B loads per hit instead of 6, 1 instead of 3 for a path that ends at this hit),
    // the emission apart because almost no triangle has one
    const bool emits = !(f[3] == 0.f && f[4] == 0.f && f[5] == 0.f);
    shade64[4 * (size_t)pos + 0] = make_float4(a.x, a.y, a.z, nrm.x);
    shade64[4 * (size_t)pos + 1] = make_float4(b.x, b.y, b.z, nrm.y);
    shade64[4 * (size_t)pos + 2] = make_float4(c.x, c.y, c.z, nrm.z);
    shade64[4 * (size_t)pos + 3] = make_float4(br, bg, bb, emits ? 1.f : 0.f);
    ke4[pos] = make_float4(f[3], f[4], f[5], 0.f);
}


// ---- PLOC: the surface-area-class binary tree of big scenes (ePreferFastTrace, main.cpp:419) --------------------------
// The LBVH above splits by Morton-code bits, i.e. at spatial medians: near-optimal for uniformly distributed, equally sized
// triangles and poor for everything else (a finely tessellated object in a large room: the "teapot in a stadium").  For
// scenes beyond the one-workgroup surface-area sweep (bvh4_sah_device.hip, <= PT_SAH_MAX_TRIS triangles) the binary tree is
// therefore rebuilt BOTTOM-UP from the Morton order by parallel locally-ordered clustering (Meister & Bittner 2018): every
// cluster looks at its PLOC_R neighbours on either side in the current cluster array, picks the one whose union with it
// has the smallest surface area, and mutual choices merge -- all clusters at once, ~log n rounds, each one a nearest-
// neighbour kernel, two scans and a merge kernel.  Small triangles cluster with small triangles before anything large
// touches them, which is what the spatial median cannot do.  Output: the same arrays the LBVH stage produces (topo, range,
// parents, boxes at [pos] / [n + node], root = node 0) over a NEW leaf order -- the depth-first order of the new tree, so
// a subtree is again a contiguous range of positions -- and everything downstream (BVH4 collapse, top-down BVH4, 8-wide
// nodes, triangle tables) runs unchanged.  Deterministic: ties go to the lowest index, node numbers come from scans.
__device__ __forceinline__ float box_area(const float4 lo, const float4 hi)
{
    const float x = hi.x - lo.x, y = hi.y - lo.y, z = hi.z - lo.z;
    return (x * y + y * z) + z * x;
}

constexpr int PLOC_R_MAX = 32;  // the search radius is a run-time choice (pt_tuning.ploc_radius, default 8) up to this

__device__ __forceinline__ float union_area(const float4 alo, const float4 ahi, const float4 blo, const float4 bhi)
{
    const float x = fmaxf(ahi.x, bhi.x) - fminf(alo.x, blo.x), y = fmaxf(ahi.y, bhi.y) - fminf(alo.y, blo.y),
                z = fmaxf(ahi.z, bhi.z) - fminf(alo.z, blo.z);
    return (x * y + y * z) + z * x;
}

__global__ __launch_bounds__(TB) void k_ploc_init(uint32_t n, const float4 *__restrict__ box_lo, const float4 *__restrict__ box_hi,
                                                  uint32_t *__restrict__ cl_ref, float4 *__restrict__ cl_lo, float4 *__restrict__ cl_hi)
{
    const uint32_t i = blockIdx.x * TB + threadIdx.x;
    if (i >= n) return;
    cl_ref[i] = PT_LEAF | i;
    cl_lo[i] = box_lo[i];
    cl_hi[i] = box_hi[i];
}

// nearest neighbour of every cluster within `radius` positions: the partner j minimising the PAIR key
// (union area, parity of the pair's lower index, lower index, upper index).  The key is a function of the unordered pair,
// so the pair that is minimal among all candidate pairs chooses each other and every round merges at least one; the parity
// term is what keeps regular geometry moving: in a row of equal tiles every union area ties, "lowest index wins" would make
// everybody point left (one merge per round), "even lower index first" pairs them all up at once.
__global__ __launch_bounds__(TB) void k_ploc_nn(uint32_t m, int radius, const float4 *__restrict__ cl_lo, const float4 *__restrict__ cl_hi,
                                                uint32_t *__restrict__ nn)
{
    __shared__ float4 s_lo[TB + 2 * PLOC_R_MAX], s_hi[TB + 2 * PLOC_R_MAX];
    const int base = (int)(blockIdx.x * TB) - radius;
    for (int t = threadIdx.x; t < TB + 2 * radius; t += TB) {
        const int j = base + t;
        if (j >= 0 && j < (int)m) { s_lo[t] = cl_lo[j]; s_hi[t] = cl_hi[j]; }
    }
    __syncthreads();
    const int i = (int)(blockIdx.x * TB + threadIdx.x);
    if (i >= (int)m) return;
    const float4 alo = s_lo[threadIdx.x + radius], ahi = s_hi[threadIdx.x + radius];
    float best = INFINITY;
    int bj = -1, bpar = 0;
    for (int d = -radius; d <= radius; d++) {  // ascending j: among equal (area, parity) the lowest partner, i.e. the lowest pair
        const int j = i + d;
        if (d == 0 || j < 0 || j >= (int)m) continue;
        const float a = union_area(alo, ahi, s_lo[threadIdx.x + radius + d], s_hi[threadIdx.x + radius + d]);
        const int par = (j < i ? j : i) & 1;
        if (bj < 0 || a < best || (a == best && par < bpar)) { best = a; bj = j; bpar = par; }
    }
    nn[i] = (uint32_t)bj;
}

// sum of the surface areas of the internal nodes' boxes, per block (the host adds the partial sums in order): what a
// surface-area cost compares between two binary trees over the same leaves
__global__ __launch_bounds__(TB) void k_tree_area(uint32_t n_int, uint32_t n, const float4 *__restrict__ box_lo, const float4 *__restrict__ box_hi,
                                                  double *__restrict__ partial)
{
    __shared__ double s[TB];
    const uint32_t i = blockIdx.x * TB + threadIdx.x;
    double a = 0.0;
    if (i < n_int) a = (double)box_area(box_lo[(size_t)n + i], box_hi[(size_t)n + i]);
    s[threadIdx.x] = a;
    __syncthreads();
    for (int o = TB / 2; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) s[threadIdx.x] += s[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[blockIdx.x] = s[0];
}

// keep[i] = the cluster stays in the array (itself, or as the node it merges into); lower[i] = it is the lower half of a
// merging pair and creates the node
__global__ __launch_bounds__(TB) void k_ploc_mark(uint32_t m, const uint32_t *__restrict__ nn, uint32_t *__restrict__ keep,
                                                  uint32_t *__restrict__ lower)
{
    const uint32_t i = blockIdx.x * TB + threadIdx.x;
    if (i >= m) return;
    const uint32_t j = nn[i];
    const bool mutual = nn[j] == i;
    keep[i] = (mutual && j < i) ? 0u : 1u;
    lower[i] = (mutual && i < j) ? 1u : 0u;
}

// node ids are handed out downwards from id_hi (the ids still free are [0, id_hi)), so that the last merge is node 0
__global__ __launch_bounds__(TB) void k_ploc_merge(uint32_t m, uint32_t n, uint32_t id_hi, const uint32_t *__restrict__ nn,
                                                   const uint32_t *__restrict__ oidx, const uint32_t *__restrict__ mrank,
                                                   const uint32_t *__restrict__ ref_in, const float4 *__restrict__ lo_in,
                                                   const float4 *__restrict__ hi_in, uint32_t *__restrict__ ref_out,
                                                   float4 *__restrict__ lo_out, float4 *__restrict__ hi_out, uint2 *__restrict__ topo,
                                                   uint32_t *__restrict__ parent_int, uint32_t *__restrict__ parent_leaf,
                                                   uint32_t *__restrict__ isz, float4 *__restrict__ box_lo, float4 *__restrict__ box_hi)
{
    const uint32_t i = blockIdx.x * TB + threadIdx.x;
    if (i >= m) return;
    const uint32_t j = nn[i];
    const bool mutual = nn[j] == i;
    if (mutual && j < i) return;  // the upper half: its partner writes the node
    uint32_t ref = ref_in[i];
    float4 lo = lo_in[i], hi = hi_in[i];
    if (mutual) {
        const uint32_t id = id_hi - 1u - mrank[i];
        const uint32_t rj = ref_in[j];
        const float4 jlo = lo_in[j], jhi = hi_in[j];
        topo[id] = make_uint2(ref, rj);
        const uint32_t sa = (ref & PT_LEAF) ? 1u : isz[ref], sb = (rj & PT_LEAF) ? 1u : isz[rj];
        isz[id] = sa + sb;
        if (ref & PT_LEAF) parent_leaf[ref & ~PT_LEAF] = id; else parent_int[ref] = id;
        if (rj & PT_LEAF) parent_leaf[rj & ~PT_LEAF] = id; else parent_int[rj] = id;
        lo = make_float4(fminf(lo.x, jlo.x), fminf(lo.y, jlo.y), fminf(lo.z, jlo.z), 0.f);
        hi = make_float4(fmaxf(hi.x, jhi.x), fmaxf(hi.y, jhi.y), fmaxf(hi.z, jhi.z), 0.f);
        box_lo[(size_t)n + id] = lo;
        box_hi[(size_t)n + id] = hi;
        ref = id;
    }
    const uint32_t o = oidx[i];
    ref_out[o] = ref;
    lo_out[o] = lo;
    hi_out[o] = hi;
}

// position of a subtree's first leaf in the depth-first leaf order: the sizes of all left siblings on the way to the root
__device__ __forceinline__ uint32_t ploc_first(uint32_t ref, uint32_t node, const uint2 *__restrict__ topo,
                                               const uint32_t *__restrict__ parent_int, const uint32_t *__restrict__ isz, uint32_t &depth)
{
    uint32_t off = 0;
    depth = 1;
    for (;;) {
        const uint2 ch = topo[node];
        if (ch.y == ref) off += (ch.x & PT_LEAF) ? 1u : isz[ch.x];
        if (node == 0u) break;
        ref = node;
        node = parent_int[node];
        depth++;
    }
    return off;
}

__global__ __launch_bounds__(TB) void k_ploc_leaf_order(uint32_t n, const uint2 *__restrict__ topo, const uint32_t *__restrict__ parent_int,
                                                        const uint32_t *__restrict__ parent_leaf, const uint32_t *__restrict__ isz,
                                                        uint32_t *__restrict__ newpos, uint32_t *__restrict__ height)
{
    const uint32_t i = blockIdx.x * TB + threadIdx.x;
    if (i >= n) return;
    uint32_t depth;
    newpos[i] = ploc_first(PT_LEAF | i, parent_leaf[i], topo, parent_int, isz, depth);
    atomicMax(height, depth);
}

__global__ __launch_bounds__(TB) void k_ploc_ranges(uint32_t n_int, const uint2 *__restrict__ topo, const uint32_t *__restrict__ parent_int,
                                                    const uint32_t *__restrict__ isz, uint2 *__restrict__ range)
{
    const uint32_t i = blockIdx.x * TB + threadIdx.x;
    if (i >= n_int) return;
    uint32_t depth, first = 0;
    if (i != 0u) first = ploc_first(i, parent_int[i], topo, parent_int, isz, depth);
    range[i] = make_uint2(first, first + isz[i] - 1u);
}

// leaves move to their new positions: boxes, parents, primitive ids
__global__ __launch_bounds__(TB) void k_ploc_move_leaves(uint32_t n, const uint32_t *__restrict__ newpos, const float4 *__restrict__ lo_in,
                                                         const float4 *__restrict__ hi_in, const uint32_t *__restrict__ pleaf_in,
                                                         const uint32_t *__restrict__ prim_in, float4 *__restrict__ box_lo,
                                                         float4 *__restrict__ box_hi, uint32_t *__restrict__ pleaf_out,
                                                         uint32_t *__restrict__ prim_out)
{
    const uint32_t i = blockIdx.x * TB + threadIdx.x;
    if (i >= n) return;
    const uint32_t p = newpos[i];
    box_lo[p] = lo_in[i];
    box_hi[p] = hi_in[i];
    pleaf_out[p] = pleaf_in[i];
    prim_out[p] = prim_in[i];
}

__global__ __launch_bounds__(TB) void k_ploc_retarget(uint32_t n_int, const uint32_t *__restrict__ newpos, uint2 *__restrict__ topo)
{
    const uint32_t i = blockIdx.x * TB + threadIdx.x;
    if (i >= n_int) return;
    uint2 ch = topo[i];
    if (ch.x & PT_LEAF) ch.x = PT_LEAF | newpos[ch.x & ~PT_LEAF];
    if (ch.y & PT_LEAF) ch.y = PT_LEAF | newpos[ch.y & ~PT_LEAF];
    topo[i] = ch;
}

// ---- BVH8 (scenes walked out of L2 / MALL / HBM) ------------------------------------------------------------------
// Measured on MI355X (scripts/ubench/gather_rate.hip): beyond L2 a wave's divergent loads cost per distinct 128-B LINE
// (~56 G lines/s for the chip), not per byte or per load instruction -- four 16-B loads of one line cost what one does.
// So the wide node of big scenes is one whole line holding EIGHT children, and a ray fetches a third fewer lines than
// with four children per 64-B node (22 instead of 35 node visits on a 100k-triangle soup of C5's density).
//   node = 8 x uint4:  lo.x[8] lo.y[8] lo.z[8] hi.x[8] hi.y[8] hi.z[8] as fp16 of coordinates normalised to the scene
//          box (rounded outwards, empty slot = +inf), then {child_base, tri_base, imask | lmask << 8, 0}, then a spare.
//   Internal children are CONTIGUOUS nodes (child of slot s = child_base + popcount(imask below s)) and the triangles
//   of a node's leaf children are contiguous positions of the BVH8's own triangle order (tri_base + popcount(lmask
//   below s)), so a traversal stack entry is {child_base, pending-children mask} for a whole node -- one push per visit
//   instead of one per child, and no child words to keep.
//   Children sit in the slot of their OCTANT about the node's centre where it is free, so "slot xor ray octant" visits
//   them roughly front to back without sorting (Ylitie, Karras, Laine 2017).
// Built top-down, one level per pass, from the binary LBVH: a wide node starts with the two children of its binary
// node and keeps opening the internal one of LARGEST SURFACE AREA until it has eight (the area-guided collapse that
// stands in for ePreferFastTrace, main.cpp:419, on big scenes).  Scans give every level's nodes and triangles their
// places, so the result is deterministic.

// one thread per wide node of this level: its up-to-W children as binary references in slot order (PT_MISS = empty)
template <int W>
__global__ __launch_bounds__(TB) void k_w8_expand(uint32_t count, const uint32_t *__restrict__ front, int n,
                                                  const uint2 *__restrict__ topo, const float4 *__restrict__ box_lo,
                                                  const float4 *__restrict__ box_hi, uint32_t *__restrict__ kids,
                                                  uint32_t *__restrict__ n_int, uint32_t *__restrict__ n_leaf)
{
    const uint32_t j = blockIdx.x * TB + threadIdx.x;
    if (j >= count) return;
    const uint32_t b = front[j];
    uint32_t ref[8];
    int m = 2;
    { const uint2 ch = topo[b]; ref[0] = ch.x; ref[1] = ch.y; }
    while (m < W) {
        int pick = -1;
        float pa = -1.f;
        for (int k = 0; k < m; k++) {
            if (ref[k] & PT_LEAF) continue;
            const float a = box_area(box_lo[(size_t)n + ref[k]], box_hi[(size_t)n + ref[k]]);
            if (a > pa) { pa = a; pick = k; }  // first maximum wins
        }
        if (pick < 0) break;
        const uint2 ch = topo[ref[pick]];
        ref[pick] = ch.x;
        ref[m++] = ch.y;
    }
    // slots by octant of the child's centre about the centre of this node's box; taken -> the free slot with the
    // fewest differing octant bits (lowest index among equals)
    const float4 nlo = box_lo[(size_t)n + b], nhi = box_hi[(size_t)n + b];
    const float cx = 0.5f * (nlo.x + nhi.x), cy = 0.5f * (nlo.y + nhi.y), cz = 0.5f * (nlo.z + nhi.z);
    uint32_t slot_ref[8];
    for (int k = 0; k < 8; k++) slot_ref[k] = PT_MISS;
    uint32_t used = 0, ni = 0, nl = 0;
    for (int k = 0; k < m; k++) {
        const bool leaf = (ref[k] & PT_LEAF) != 0u;
        if (W < 8) {  // four-wide nodes are sorted by entry distance at traversal time: slots in the order found
            slot_ref[k] = ref[k];
            if (leaf) nl++; else ni++;
            continue;
        }
        const size_t bi = leaf ? (size_t)(ref[k] & ~PT_LEAF) : (size_t)n + ref[k];
        const float4 lo = box_lo[bi], hi = box_hi[bi];
        const uint32_t want = (0.5f * (lo.x + hi.x) > cx ? 1u : 0u) | (0.5f * (lo.y + hi.y) > cy ? 2u : 0u) | (0.5f * (lo.z + hi.z) > cz ? 4u : 0u);
        uint32_t best = 8, bd = 9;
        for (uint32_t sl = 0; sl < 8; sl++) {
            if (used & (1u << sl)) continue;
            const uint32_t d = (uint32_t)__popc(sl ^ want);
            if (d < bd) { bd = d; best = sl; }
        }
        used |= 1u << best;
        slot_ref[best] = ref[k];
        if (leaf) nl++; else ni++;
    }
    for (int k = 0; k < 8; k++) kids[8 * (size_t)j + k] = slot_ref[k];
    n_int[j] = ni;
    n_leaf[j] = nl;
}

// writes the level's nodes, the next level's frontier and the triangle order
// Triangle order: a wide node's subtree covers the same contiguous range of positions as its binary node does in the
// sorted (Morton) order; inside it come first the node's own leaf triangles, then the subtrees of its internal children
// in slot order -- so the order stays spatially coherent at every scale (k_shade and the triangle fetches gather from it)
// AND a node's leaf triangles are contiguous.  tri_start = first position of the node's range.
__global__ __launch_bounds__(TB) void k_w8_emit(uint32_t count, uint32_t level_base, uint32_t next_base, int n,
                                                const uint32_t *__restrict__ kids, const uint32_t *__restrict__ int_off,
                                                const uint32_t *__restrict__ tri_start, const uint2 *__restrict__ range,
                                                const float4 *__restrict__ box_lo,
                                                const float4 *__restrict__ box_hi, float cx, float cy, float cz, float rsx,
                                                float rsy, float rsz, uint4 *__restrict__ wide8, uint32_t *__restrict__ next_front,
                                                uint32_t *__restrict__ next_start, uint32_t *__restrict__ order8)
{
    const uint32_t j = blockIdx.x * TB + threadIdx.x;
    if (j >= count) return;
    const float c[3] = { cx, cy, cz }, rs[3] = { rsx, rsy, rsz };
    float bl[3][8], bh[3][8];          // child boxes in normalised scene coordinates, rounded outwards (as k_wide_half)
    bool present[8];
    uint32_t imask = 0, lmask = 0, ni = 0, nl = 0;
    const uint32_t child_base = next_base + int_off[j], tri_base = tri_start[j];
    uint32_t sub_start = tri_base;  // where the next internal child's range begins: behind this node's own leaves
    for (int k = 0; k < 8; k++) {
        const uint32_t r = kids[8 * (size_t)j + k];
        if (r != PT_MISS && (r & PT_LEAF)) sub_start++;
    }
    float nlo[3] = { INFINITY, INFINITY, INFINITY }, nhi[3] = { -INFINITY, -INFINITY, -INFINITY };
    for (int k = 0; k < 8; k++) {
        const uint32_t r = kids[8 * (size_t)j + k];
        present[k] = r != PT_MISS;
        if (!present[k]) continue;
        const bool leaf = (r & PT_LEAF) != 0u;
        const size_t bi = leaf ? (size_t)(r & ~PT_LEAF) : (size_t)n + r;
        const float4 lo = box_lo[bi], hi = box_hi[bi];
        const float l[3] = { lo.x, lo.y, lo.z }, h[3] = { hi.x, hi.y, hi.z };
        for (int ax = 0; ax < 3; ax++) {
            bl[ax][k] = (l[ax] - c[ax]) * rs[ax] - 3.814697265625e-06f;
            bh[ax][k] = (h[ax] - c[ax]) * rs[ax] + 3.814697265625e-06f;
            nlo[ax] = fminf(nlo[ax], bl[ax][k]);
            nhi[ax] = fmaxf(nhi[ax], bh[ax][k]);
        }
        if (leaf) {
            lmask |= 1u << k;
            order8[tri_base + nl] = r & ~PT_LEAF;  // 8-wide triangle position -> position of the binary tree's leaf order
            nl++;
        } else {
            imask |= 1u << k;
            next_front[int_off[j] + ni] = r;
            next_start[int_off[j] + ni] = sub_start;
            const uint2 rg = range[r];
            sub_start += rg.y - rg.x + 1u;
            ni++;
        }
    }
    // 64-B node: the six planes of the eight children as BYTES on the node's own grid -- origin (16 bits per axis on the
    // 2^-14 grid of [-2, 2), at or below the node's lower corner) + q * 2^-e with a per-axis exponent (5 bits) just large enough
    // for 255 steps to reach the node's upper corner; lower planes rounded down, upper planes up, so every decoded box contains
    // the box it stands for.  Empty slots: lo = 255, hi = 0 (an inverted interval on every axis: no ray hits it; the
    // traversal also masks them out).  Then child_base | imask << 24 and tri_base | lmask << 24.
    uint32_t o16[3], ecode[3], ql[3][8], qh[3][8];
    for (int ax = 0; ax < 3; ax++) {
        const double og = floor(((double)nlo[ax] + 2.0) * 16384.0);
        o16[ax] = (uint32_t)fmin(fmax(og, 0.0), 65535.0);
        const double origin = (double)o16[ax] * (1.0 / 16384.0) - 2.0;
        const double ext = (double)nhi[ax] - origin;
        int e = -31;
        while (e < 0 && 255.0 * ldexp(1.0, e) < ext) e++;
        ecode[ax] = (uint32_t)(-e);
        const double inv_step = ldexp(1.0, -e);
        for (int k = 0; k < 8; k++) {
            if (!present[k]) { ql[ax][k] = 255u; qh[ax][k] = 0u; continue; }
            ql[ax][k] = (uint32_t)fmin(fmax(floor(((double)bl[ax][k] - origin) * inv_step), 0.0), 255.0);
            qh[ax][k] = (uint32_t)fmin(fmax(ceil(((double)bh[ax][k] - origin) * inv_step), 0.0), 255.0);
        }
    }
    auto pack4 = [](const uint32_t *q) { return q[0] | (q[1] << 8) | (q[2] << 16) | (q[3] << 24); };
    uint4 *o = wide8 + 4 * (size_t)(level_base + j);
    o[0] = make_uint4(pack4(ql[0]), pack4(ql[0] + 4), pack4(ql[1]), pack4(ql[1] + 4));
    o[1] = make_uint4(pack4(ql[2]), pack4(ql[2] + 4), pack4(qh[0]), pack4(qh[0] + 4));
    o[2] = make_uint4(pack4(qh[1]), pack4(qh[1] + 4), pack4(qh[2]), pack4(qh[2] + 4));
    o[3] = make_uint4(o16[0] | (o16[1] << 16), o16[2] | (ecode[0] << 16) | (ecode[1] << 21) | (ecode[2] << 26), child_base | (imask << 24),
                      tri_base | (lmask << 24));
}

// The same level-by-level build with FOUR children per node, in the 64-B format k_extend<hbm> walks (fp16 planes +
// child words; leaves point into the usual sorted triangle order): area-guided collapse, and the children of a node
// are contiguous -- 4 x 64 B = two 128-B lines, so the siblings a ray visits after one another share lines and the
// levels of the tree are dense in memory (the chip charges per line beyond L2, not per node).
__global__ __launch_bounds__(TB) void k_w4_emit(uint32_t count, uint32_t level_base, uint32_t next_base, int n,
                                                const uint32_t *__restrict__ kids, const uint32_t *__restrict__ int_off,
                                                const float4 *__restrict__ box_lo, const float4 *__restrict__ box_hi, float cx,
                                                float cy, float cz, float rsx, float rsy, float rsz, uint4 *__restrict__ wide16,
                                                uint32_t *__restrict__ next_front, int compact16)
{
    // compact16 (the TLAS of k_extend_inst16): child words as 16-bit codes -- node index, 0x8000 | leaf position, 0xFFFF empty
    const uint32_t j = blockIdx.x * TB + threadIdx.x;
    if (j >= count) return;
    const float c[3] = { cx, cy, cz }, rs[3] = { rsx, rsy, rsz };
    uint32_t hl[3][4], hh[3][4], word[4];
    uint32_t ni = 0;
    for (int k = 0; k < 4; k++) {
        const uint32_t r = kids[8 * (size_t)j + k];
        if (r == PT_MISS) {
            for (int ax = 0; ax < 3; ax++) hl[ax][k] = hh[ax][k] = 0x7C00u;
            word[k] = compact16 ? 0xFFFFu : PT_MISS;
            continue;
        }
        const bool leaf = (r & PT_LEAF) != 0u;
        const size_t bi = leaf ? (size_t)(r & ~PT_LEAF) : (size_t)n + r;
        const float4 lo = box_lo[bi], hi = box_hi[bi];
        const float l[3] = { lo.x, lo.y, lo.z }, h[3] = { hi.x, hi.y, hi.z };
        for (int ax = 0; ax < 3; ax++) {
            hl[ax][k] = __half_as_ushort(__float2half_rd((l[ax] - c[ax]) * rs[ax] - 3.814697265625e-06f));
            hh[ax][k] = __half_as_ushort(__float2half_ru((h[ax] - c[ax]) * rs[ax] + 3.814697265625e-06f));
        }
        if (leaf) {
            word[k] = compact16 ? (0x8000u | (r & 0x7FFFu)) : r;  // PT_LEAF | sorted position, count 1
        } else {
            word[k] = next_base + int_off[j] + ni;
            next_front[int_off[j] + ni] = r;
            ni++;
        }
    }
    uint4 *o = wide16 + 4 * (size_t)(level_base + j);
    o[0] = make_uint4(hl[0][0] | (hl[0][1] << 16), hl[0][2] | (hl[0][3] << 16), hl[1][0] | (hl[1][1] << 16), hl[1][2] | (hl[1][3] << 16));
    o[1] = make_uint4(hl[2][0] | (hl[2][1] << 16), hl[2][2] | (hl[2][3] << 16), hh[0][0] | (hh[0][1] << 16), hh[0][2] | (hh[0][3] << 16));
    o[2] = make_uint4(hh[1][0] | (hh[1][1] << 16), hh[1][2] | (hh[1][3] << 16), hh[2][0] | (hh[2][1] << 16), hh[2][2] | (hh[2][3] << 16));
    o[3] = make_uint4(word[0], word[1], word[2], word[3]);
}

// order8 o sorted order: BVH8 position -> primitive id
__global__ __launch_bounds__(TB) void k_compose(const uint32_t *__restrict__ order8, const uint32_t *__restrict__ prim_of, uint32_t n,
                                                uint32_t *__restrict__ out)
{
    const uint32_t i = blockIdx.x * TB + threadIdx.x;
    if (i < n) out[i] = prim_of[order8[i]];
}

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

// the normalisation of the fp16 node formats: x' = (x - c) * rs with c the centre and 1/rs the half extent of the scene box
static void norm_box(const float *bmin, const float *bmax, float *c, float *sv, float *rs)
{
    float ext = 0.f;
    for (int k = 0; k < 3; k++) ext = fmaxf(ext, bmax[k] - bmin[k]);
    for (int k = 0; k < 3; k++) {
        c[k] = 0.5f * (bmin[k] + bmax[k]);
        // half extent, never degenerate (flat scenes) and never so small that the padded boxes leave fp16's range
        sv[k] = fmaxf(0.5f * (bmax[k] - bmin[k]), fmaxf(ext * 0x1p-10f, 1e-30f));
        rs[k] = 1.0f / sv[k];
    }
}

__global__ __launch_bounds__(TB) void k_bounds(const float4 *__restrict__ tlo, const float4 *__restrict__ thi, uint32_t n,
                                               uint32_t *__restrict__ scene_ord)
{
    const uint32_t t = blockIdx.x * TB + threadIdx.x;
    float mn[3] = { INFINITY, INFINITY, INFINITY }, mx[3] = { -INFINITY, -INFINITY, -INFINITY };
    if (t < n) {
        const float4 a = tlo[t], b = thi[t];
        mn[0] = a.x; mn[1] = a.y; mn[2] = a.z;
        mx[0] = b.x; mx[1] = b.y; mx[2] = b.z;
    }
    __shared__ float s_mn[4][3], s_mx[4][3];
    for (int k = 0; k < 3; k++) {
        const float a = wave_min(mn[k]), b = wave_max(mx[k]);
        if ((threadIdx.x & 63) == 0) {
            s_mn[threadIdx.x >> 6][k] = a;
            s_mx[threadIdx.x >> 6][k] = b;
        }
    }
    __syncthreads();
    if (threadIdx.x < 3) {  // one atomic pair per block and axis (6 words shared by the whole grid)
        const int k = threadIdx.x;
        atomicMin(&scene_ord[k], f2ord(fminf(fminf(s_mn[0][k], s_mn[1][k]), fminf(s_mn[2][k], s_mn[3][k]))));
        atomicMax(&scene_ord[3 + k], f2ord(fmaxf(fmaxf(s_mx[0][k], s_mx[1][k]), fmaxf(s_mx[2][k], s_mx[3][k]))));
    }
}


// Rebuilds the binary tree over the Morton-ordered leaves by PLOC (kernels above), in place of the LBVH's arrays.
// In: leaf boxes box_lo/box_hi[0, n) and prim_of in Morton order.  Out: topo / range / parent_int / parent_leaf, boxes of
// leaves [0, n) and internal nodes [n, 2n - 1) in the NEW leaf order, d_prim_q (new position -> primitive id), height.
// Returns PT_ERR_UNSUPPORTED (and leaves the LBVH arrays untouched as far as the caller's later stages are concerned: they
// are only overwritten at the very end) if the clustering stalls, which the caller answers by keeping the LBVH.
// sum of the internal nodes' surface areas of a tree in the [pos] / [n + node] box layout (deterministic: partial sums added in order)
static pt_status tree_area(pt_ctx *ctx, uint32_t n, const float4 *d_blo, const float4 *d_bhi, double *out)
{
    const uint32_t n_int = n - 1u, g = (n_int + TB - 1) / TB;
    DevBuf<double> part;
    PT_HIP(ctx, part.alloc(g));
    k_tree_area<<<g, TB, 0, ctx->stream>>>(n_int, n, d_blo, d_bhi, part.p);
    std::vector<double> h(g);
    PT_HIP(ctx, hipMemcpyAsync(h.data(), part.p, sizeof(double) * g, hipMemcpyDeviceToHost, ctx->stream));
    PT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    double a = 0.0;
    for (double x : h) a += x;
    *out = a;
    return PT_OK;
}

// area_lbvh: the LBVH's sum of internal surface areas; *area_ploc gets the rebuilt tree's.  The rebuilt tree is adopted
// (PT_OK, arrays replaced) only if its sum is below 0.9 of the LBVH's -- ePreferFastTrace means the cheaper tree, whichever
// builder made it; otherwise PT_ERR_UNSUPPORTED and the LBVH stands.
static pt_status ploc_refine(pt_ctx *ctx, uint32_t n, int radius, double area_lbvh, double *area_ploc, uint2 *d_topo, uint2 *d_range,
                             uint32_t *d_pint, uint32_t *d_pleaf, float4 *d_blo, float4 *d_bhi, const uint32_t *d_prim_of,
                             uint32_t *d_prim_q, uint32_t *d_sums, uint32_t *h_height)
{
    hipStream_t st = ctx->stream;
    DevBuf<uint32_t> ref[2], nn, keep, lower, isz, newpos, pint, pleaf, pleaf2, height;
    DevBuf<float4> lo[2], hi[2], nblo, nbhi;
    DevBuf<uint2> topo;
    for (int k = 0; k < 2; k++) {
        PT_HIP(ctx, ref[k].alloc(n));
        PT_HIP(ctx, lo[k].alloc(n));
        PT_HIP(ctx, hi[k].alloc(n));
    }
    PT_HIP(ctx, nn.alloc(n));
    PT_HIP(ctx, keep.alloc(n));
    PT_HIP(ctx, lower.alloc(n));
    PT_HIP(ctx, isz.alloc(n));
    PT_HIP(ctx, newpos.alloc(n));
    PT_HIP(ctx, pint.alloc(n));
    PT_HIP(ctx, pleaf.alloc(n));
    PT_HIP(ctx, pleaf2.alloc(n));
    PT_HIP(ctx, height.alloc(1));
    PT_HIP(ctx, topo.alloc(n));
    PT_HIP(ctx, nblo.alloc(2 * (size_t)n));
    PT_HIP(ctx, nbhi.alloc(2 * (size_t)n));
    k_ploc_init<<<(n + TB - 1) / TB, TB, 0, st>>>(n, d_blo, d_bhi, ref[0].p, lo[0].p, hi[0].p);
    uint32_t m = n, id_hi = n - 1u;
    int cur = 0;
    for (int round = 0; m > 1u; round++) {
        const uint32_t g = (m + TB - 1) / TB;
        k_ploc_nn<<<g, TB, 0, st>>>(m, radius, lo[cur].p, hi[cur].p, nn.p);
        k_ploc_mark<<<g, TB, 0, st>>>(m, nn.p, keep.p, lower.p);
        uint32_t last[2] = { 0, 0 }, tot[2] = { 0, 0 };
        PT_HIP(ctx, hipMemcpyAsync(&last[0], keep.p + (m - 1), sizeof(uint32_t), hipMemcpyDeviceToHost, st));
        PT_HIP(ctx, hipMemcpyAsync(&last[1], lower.p + (m - 1), sizeof(uint32_t), hipMemcpyDeviceToHost, st));
        exclusive_scan(keep.p, m, d_sums, st);
        exclusive_scan(lower.p, m, d_sums, st);
        PT_HIP(ctx, hipMemcpyAsync(&tot[0], keep.p + (m - 1), sizeof(uint32_t), hipMemcpyDeviceToHost, st));
        PT_HIP(ctx, hipMemcpyAsync(&tot[1], lower.p + (m - 1), sizeof(uint32_t), hipMemcpyDeviceToHost, st));
        PT_HIP(ctx, hipStreamSynchronize(st));
        const uint32_t m_new = tot[0] + last[0], merges = tot[1] + last[1];
        if (merges == 0u || m_new + merges != m || merges > id_hi) { ctx->err = "internal: PLOC round made no progress"; return PT_ERR_HIP; }
        // (typical: a fifth to two fifths of the clusters merge per round, 60-90 rounds for a million triangles.)  A scene
        // whose clusters merge a handful at a time -- pathological chains -- would need ~n rounds: keep the LBVH
        if ((round >= 64 && m > 1024u && merges * 256u < m) || round >= 2000) return PT_ERR_UNSUPPORTED;
        k_ploc_merge<<<g, TB, 0, st>>>(m, n, id_hi, nn.p, keep.p, lower.p, ref[cur].p, lo[cur].p, hi[cur].p, ref[cur ^ 1].p, lo[cur ^ 1].p,
                                       hi[cur ^ 1].p, topo.p, pint.p, pleaf.p, isz.p, nblo.p, nbhi.p);
        id_hi -= merges;
        m = m_new;
        cur ^= 1;
    }
    if (id_hi != 0u) { ctx->err = "internal: PLOC did not use every node id"; return PT_ERR_HIP; }
    {
        const pt_status arc = tree_area(ctx, n, nblo.p, nbhi.p, area_ploc);   // (internal boxes do not depend on the leaf order)
        if (arc != PT_OK) return arc;
        // adopted only when clearly cheaper: on uniformly distributed, equally sized triangles (the soup of config C5) the
        // two sums are within 1 % of each other and the LBVH's balanced tree collapses into the better BVH4 (measured:
        // 36.2 against 38.9 node visits per ray, profiles/r03_probe_stress_scene.txt)
        if (!(*area_ploc < 0.9 * area_lbvh)) return PT_ERR_UNSUPPORTED;
    }
    const uint32_t n_int = n - 1u, gi = (n_int + TB - 1) / TB, gl = (n + TB - 1) / TB;
    PT_HIP(ctx, hipMemsetAsync(height.p, 0, sizeof(uint32_t), st));
    k_ploc_leaf_order<<<gl, TB, 0, st>>>(n, topo.p, pint.p, pleaf.p, isz.p, newpos.p, height.p);
    k_ploc_ranges<<<gi, TB, 0, st>>>(n_int, topo.p, pint.p, isz.p, d_range);
    k_ploc_move_leaves<<<gl, TB, 0, st>>>(n, newpos.p, d_blo, d_bhi, pleaf.p, d_prim_of, nblo.p, nbhi.p, pleaf2.p, d_prim_q);
    k_ploc_retarget<<<gi, TB, 0, st>>>(n_int, newpos.p, topo.p);
    // the new tree replaces the LBVH's working arrays
    PT_HIP(ctx, hipMemcpyAsync(d_topo, topo.p, sizeof(uint2) * (size_t)n_int, hipMemcpyDeviceToDevice, st));
    PT_HIP(ctx, hipMemcpyAsync(d_pint, pint.p, sizeof(uint32_t) * (size_t)n_int, hipMemcpyDeviceToDevice, st));
    PT_HIP(ctx, hipMemcpyAsync(d_pleaf, pleaf2.p, sizeof(uint32_t) * (size_t)n, hipMemcpyDeviceToDevice, st));
    PT_HIP(ctx, hipMemcpyAsync(d_blo, nblo.p, sizeof(float4) * (2 * (size_t)n - 1), hipMemcpyDeviceToDevice, st));
    PT_HIP(ctx, hipMemcpyAsync(d_bhi, nbhi.p, sizeof(float4) * (2 * (size_t)n - 1), hipMemcpyDeviceToDevice, st));
    PT_HIP(ctx, hipMemcpyAsync(h_height, height.p, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    PT_HIP(ctx, hipStreamSynchronize(st));
    PT_HIP(ctx, hipGetLastError());
    return PT_OK;
}

// top_down: bit 0 = also the BVH8 (+ its triangle order), bit 1 = also the top-down BVH4 in the 64-B format, bit 2 = that BVH4
// with 16-bit child codes (k_w4_emit compact16: the TLAS of k_extend_inst16; needs n < 32768)
// ploc: the binary tree is rebuilt by PLOC before the collapses (out.d_prim_q = its leaf order; out.d_prim_of, d_keys and
// d_nodes stay the LBVH's, for the parity read-back)
static pt_status build_bvh(pt_ctx *ctx, const float4 *d_tlo, const float4 *d_thi, uint32_t n, uint32_t leaf_max, BvhOut &out,
                          int top_down = 0, bool ploc = false)
{
    const bool want8 = (top_down & 1) != 0, want4t = (top_down & 2) != 0;
    hipStream_t st = ctx->stream;
    const uint32_t gt = (n + TB - 1) / TB;
    DevBuf<uint32_t> d_scene, d_vals[2], d_hist, d_pint, d_pleaf, d_flags, d_height, d_wflag, d_widx, d_sums;
    DevBuf<float4> d_blo, d_bhi;
    DevBuf<unsigned long long> d_keys[2];
    DevBuf<uint2> d_topo, d_range;
    const uint32_t nblocks = (n + RS_TILE - 1) / RS_TILE;
    PT_HIP(ctx, d_scene.alloc(6));
    for (int i = 0; i < 2; i++) {
        PT_HIP(ctx, d_keys[i].alloc(n));
        PT_HIP(ctx, d_vals[i].alloc(n));
    }
    PT_HIP(ctx, d_hist.alloc(256 * (size_t)nblocks));
    PT_HIP(ctx, d_sums.alloc(std::max<size_t>(256 * (size_t)nblocks, n) / SC_TILE + 2));
    PT_HIP(ctx, d_topo.alloc(n));
    PT_HIP(ctx, d_range.alloc(n));
    PT_HIP(ctx, d_wflag.alloc(n));
    PT_HIP(ctx, d_widx.alloc(n));
    PT_HIP(ctx, d_pint.alloc(n));
    PT_HIP(ctx, d_pleaf.alloc(n));
    PT_HIP(ctx, d_flags.alloc(n));
    PT_HIP(ctx, d_height.alloc(1));
    PT_HIP(ctx, d_blo.alloc(2 * (size_t)n));
    PT_HIP(ctx, d_bhi.alloc(2 * (size_t)n));
    out.n_nodes = n > 1 ? n - 1 : 1;
    PT_HIP(ctx, hipMalloc((void **)&out.d_nodes, sizeof(float4) * 4 * (size_t)out.n_nodes));
    PT_HIP(ctx, hipMalloc((void **)&out.d_keys, sizeof(unsigned long long) * (size_t)n));
    PT_HIP(ctx, hipMalloc((void **)&out.d_prim_of, sizeof(uint32_t) * (size_t)n));

    const uint32_t ord_init[6] = { 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0u, 0u, 0u };
    PT_HIP(ctx, hipMemcpyAsync(d_scene.p, ord_init, sizeof(ord_init), hipMemcpyHostToDevice, st));
    PT_HIP(ctx, hipMemsetAsync(d_flags.p, 0, sizeof(uint32_t) * (size_t)n, st));
    PT_HIP(ctx, hipStreamSynchronize(st));  // ord_init is a stack array

    k_bounds<<<gt, TB, 0, st>>>(d_tlo, d_thi, n, d_scene.p);
    k_morton<<<gt, TB, 0, st>>>(d_tlo, d_thi, n, d_scene.p, d_keys[0].p, d_vals[0].p);
    int cur = 0;
    for (int pass = 0; pass < 8; pass++) {
        const int shift = 8 * pass;
        k_rs_hist<<<nblocks, TB, 0, st>>>(d_keys[cur].p, n, shift, d_hist.p, nblocks);
        exclusive_scan(d_hist.p, 256u * nblocks, d_sums.p, st);
        k_rs_scatter<<<nblocks, TB, 0, st>>>(d_keys[cur].p, d_vals[cur].p, d_keys[cur ^ 1].p, d_vals[cur ^ 1].p, n, shift,
                                             d_hist.p, nblocks);
        cur ^= 1;
    }
    PT_HIP(ctx, hipMemcpyAsync(out.d_keys, d_keys[cur].p, sizeof(unsigned long long) * (size_t)n, hipMemcpyDeviceToDevice, st));
    PT_HIP(ctx, hipMemcpyAsync(out.d_prim_of, d_vals[cur].p, sizeof(uint32_t) * (size_t)n, hipMemcpyDeviceToDevice, st));

    if (n > 1) {
        k_karras<<<(n - 1 + TB - 1) / TB, TB, 0, st>>>(out.d_keys, (int)n, d_topo.p, d_pint.p, d_pleaf.p, d_range.p);
        k_refit<<<gt, TB, 0, st>>>(d_tlo, d_thi, out.d_prim_of, (int)n, d_topo.p, d_pint.p, d_pleaf.p, d_blo.p, d_bhi.p,
                                   d_flags.p, d_scene.p, out.d_nodes, d_height.p);
    } else {
        k_single<<<1, 1, 0, st>>>(d_tlo, d_thi, d_scene.p, out.d_nodes, d_height.p);
    }
    uint32_t ploc_height = 0;
    if (n > 2) {
        const pt_status arc = tree_area(ctx, n, d_blo.p, d_bhi.p, &out.area_lbvh);
        if (arc != PT_OK) return arc;
        out.area_tree = out.area_lbvh;
    }
    if (ploc && n > 2) {
        PT_HIP(ctx, hipMalloc((void **)&out.d_prim_q, sizeof(uint32_t) * (size_t)n));
        double area_ploc = 0.0;
        const pt_status prc = ploc_refine(ctx, n, pt_tuned(ctx->tune.ploc_radius, 8, 1, PLOC_R_MAX), out.area_lbvh, &area_ploc, d_topo.p, d_range.p,
                                          d_pint.p, d_pleaf.p, d_blo.p, d_bhi.p, out.d_prim_of, out.d_prim_q, d_sums.p, &ploc_height);
        out.area_ploc = area_ploc;
        if (prc == PT_ERR_UNSUPPORTED) { (void)hipFree(out.d_prim_q); out.d_prim_q = nullptr; }   // stalled, or no cheaper: the LBVH stands
        else if (prc != PT_OK) return prc;
        else out.area_tree = area_ploc;
    }
    // BVH4 collapse: flag wide roots, number them (exclusive scan), emit 128-B nodes
    uint32_t n_wide = 1;
    if (n > 1) {
        const int n_int = (int)n - 1;
        const uint32_t gi = (uint32_t)(n_int + TB - 1) / TB;
        k_wide_flag<<<gi, TB, 0, st>>>(n_int, d_pint.p, d_range.p, d_wflag.p, leaf_max);
        PT_HIP(ctx, hipMemcpyAsync(d_widx.p, d_wflag.p, sizeof(uint32_t) * (size_t)n_int, hipMemcpyDeviceToDevice, st));
        exclusive_scan(d_widx.p, (uint32_t)n_int, d_sums.p, st);
        uint32_t last_idx = 0, last_flag = 0;
        PT_HIP(ctx, hipMemcpyAsync(&last_idx, d_widx.p + (n_int - 1), sizeof(uint32_t), hipMemcpyDeviceToHost, st));
        PT_HIP(ctx, hipMemcpyAsync(&last_flag, d_wflag.p + (n_int - 1), sizeof(uint32_t), hipMemcpyDeviceToHost, st));
        PT_HIP(ctx, hipStreamSynchronize(st));
        n_wide = last_idx + last_flag;
        PT_HIP(ctx, hipMalloc((void **)&out.d_wide, 128 * (size_t)n_wide));
        k_wide_emit<<<gi, TB, 0, st>>>((int)n, n_int, d_topo.p, d_range.p, d_wflag.p, d_widx.p, d_blo.p, d_bhi.p, out.d_wide, leaf_max);
    } else {
        PT_HIP(ctx, hipMalloc((void **)&out.d_wide, 128));
        k_wide_single<<<1, 1, 0, st>>>(out.d_nodes, out.d_wide);
    }
    out.n_wide = n_wide;
    out.stack_need = 0xFFFFFFFFu;  // unknown: callers fall back to the height bound
    if (n_wide <= 1024) {
        std::vector<uint32_t> h_wide(32 * (size_t)n_wide);
        PT_HIP(ctx, hipMemcpyAsync(h_wide.data(), out.d_wide, 128 * (size_t)n_wide, hipMemcpyDeviceToHost, st));
        PT_HIP(ctx, hipStreamSynchronize(st));
        out.stack_need = pt_wide_stack_need(h_wide);
    }
    uint32_t ord[6];
    PT_HIP(ctx, hipMemcpyAsync(ord, d_scene.p, sizeof(ord), hipMemcpyDeviceToHost, st));
    PT_HIP(ctx, hipMemcpyAsync(&out.height, d_height.p, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    PT_HIP(ctx, hipStreamSynchronize(st));
    out.height_tree = out.d_prim_q ? ploc_height : out.height;   // (out.height stays the LBVH's: the parity read-back)
    for (int k = 0; k < 3; k++) {
        out.bmin[k] = ord2f(ord[k]);
        out.bmax[k] = ord2f(ord[3 + k]);
    }
    PT_HIP(ctx, hipGetLastError());
    norm_box(out.bmin, out.bmax, out.norm_c, out.norm_s, out.norm_rs);
    if ((want8 || want4t) && n > 1) {
        // BVH8, level by level from the root (see k_w8_expand).  Worst case every wide node has two children: n - 1 nodes.
        const uint32_t n_int = n - 1;
        DevBuf<uint32_t> d_front[2], d_start[2], d_kids, d_ni, d_nl;
        for (int k = 0; k < 2; k++) {
            PT_HIP(ctx, d_front[k].alloc(n_int));
            PT_HIP(ctx, d_start[k].alloc(n_int));
        }
        PT_HIP(ctx, d_kids.alloc(8 * (size_t)n_int));
        PT_HIP(ctx, d_ni.alloc(n_int));
        PT_HIP(ctx, d_nl.alloc(n_int));
        uint32_t count = 1, level_base = 0, tri_run = 0, levels = 0;
        int cf = 0;
        if (want8) {
        PT_HIP(ctx, hipMalloc((void **)&out.d_wide8, 64 * (size_t)n_int));
        PT_HIP(ctx, hipMalloc((void **)&out.d_order8, sizeof(uint32_t) * (size_t)n));
        PT_HIP(ctx, hipMemsetAsync(d_front[0].p, 0, sizeof(uint32_t), st));  // level 0: the binary root, whose range starts at 0
        PT_HIP(ctx, hipMemsetAsync(d_start[0].p, 0, sizeof(uint32_t), st));
        while (count > 0) {
            const uint32_t g = (count + TB - 1) / TB;
            k_w8_expand<8><<<g, TB, 0, st>>>(count, d_front[cf].p, (int)n, d_topo.p, d_blo.p, d_bhi.p, d_kids.p, d_ni.p, d_nl.p);
            uint32_t last[2] = { 0, 0 }, tot[2] = { 0, 0 };
            PT_HIP(ctx, hipMemcpyAsync(&last[0], d_ni.p + (count - 1), sizeof(uint32_t), hipMemcpyDeviceToHost, st));
            PT_HIP(ctx, hipMemcpyAsync(&last[1], d_nl.p + (count - 1), sizeof(uint32_t), hipMemcpyDeviceToHost, st));
            exclusive_scan(d_ni.p, count, d_sums.p, st);
            exclusive_scan(d_nl.p, count, d_sums.p, st);  // (only its total is used: the leaf triangles this level places)
            PT_HIP(ctx, hipMemcpyAsync(&tot[0], d_ni.p + (count - 1), sizeof(uint32_t), hipMemcpyDeviceToHost, st));
            PT_HIP(ctx, hipMemcpyAsync(&tot[1], d_nl.p + (count - 1), sizeof(uint32_t), hipMemcpyDeviceToHost, st));
            PT_HIP(ctx, hipStreamSynchronize(st));
            const uint32_t next_count = tot[0] + last[0], leaves = tot[1] + last[1];
            const uint32_t next_base = level_base + count;
            if ((uint64_t)next_base + next_count > n_int || (uint64_t)tri_run + leaves > n) { ctx->err = "internal: BVH8 build overran its bounds"; return PT_ERR_HIP; }
            if ((uint64_t)next_base + next_count >= (1u << 24) || n >= (1u << 24)) { ctx->err = "the 8-wide nodes hold 24-bit child and triangle bases (scenes up to 16.7 M triangles)"; return PT_ERR_UNSUPPORTED; }
            k_w8_emit<<<g, TB, 0, st>>>(count, level_base, next_base, (int)n, d_kids.p, d_ni.p, d_start[cf].p, d_range.p, d_blo.p, d_bhi.p,
                                        out.norm_c[0], out.norm_c[1], out.norm_c[2], out.norm_rs[0], out.norm_rs[1], out.norm_rs[2],
                                        out.d_wide8, d_front[cf ^ 1].p, d_start[cf ^ 1].p, out.d_order8);
            level_base = next_base;
            tri_run += leaves;
            count = next_count;
            cf ^= 1;
            levels++;
        }
        if (tri_run != n) { ctx->err = "internal: BVH8 build lost triangles"; return PT_ERR_HIP; }
        out.n_wide8 = level_base;
        out.levels8 = levels;
        }
        if (want4t) {
        // ... and the four-wide tree in the 64-B format, same passes
        PT_HIP(ctx, hipMalloc((void **)&out.d_wide16t, 64 * (size_t)n_int));
        PT_HIP(ctx, hipMemsetAsync(d_front[0].p, 0, sizeof(uint32_t), st));
        count = 1; level_base = 0; levels = 0; cf = 0;
        while (count > 0) {
            const uint32_t g = (count + TB - 1) / TB;
            k_w8_expand<4><<<g, TB, 0, st>>>(count, d_front[cf].p, (int)n, d_topo.p, d_blo.p, d_bhi.p, d_kids.p, d_ni.p, d_nl.p);
            uint32_t last = 0, tot = 0;
            PT_HIP(ctx, hipMemcpyAsync(&last, d_ni.p + (count - 1), sizeof(uint32_t), hipMemcpyDeviceToHost, st));
            exclusive_scan(d_ni.p, count, d_sums.p, st);
            PT_HIP(ctx, hipMemcpyAsync(&tot, d_ni.p + (count - 1), sizeof(uint32_t), hipMemcpyDeviceToHost, st));
            PT_HIP(ctx, hipStreamSynchronize(st));
            const uint32_t next_count = tot + last, next_base = level_base + count;
            if ((uint64_t)next_base + next_count > n_int) { ctx->err = "internal: BVH4 (top-down) build overran its bounds"; return PT_ERR_HIP; }
            k_w4_emit<<<g, TB, 0, st>>>(count, level_base, next_base, (int)n, d_kids.p, d_ni.p, d_blo.p, d_bhi.p, out.norm_c[0], out.norm_c[1],
                                        out.norm_c[2], out.norm_rs[0], out.norm_rs[1], out.norm_rs[2], out.d_wide16t, d_front[cf ^ 1].p,
                                        (top_down & 4) ? 1 : 0);
            level_base = next_base;
            count = next_count;
            cf ^= 1;
            levels++;
        }
        out.n_wide16t = level_base;
        out.levels4t = levels;
        }
        PT_HIP(ctx, hipStreamSynchronize(st));
        PT_HIP(ctx, hipGetLastError());
    }
    return PT_OK;
}

// fp16 copy of a BVH4 for traversal out of HBM/L2: coordinates normalised to the scene box,
// x' = (x - c) * rs, lower bounds rounded down and upper bounds up (after a 2^-18 allowance for the float
// rounding of the normalisation itself), so every fp16 box contains its fp32 box.  Empty slots stay +inf.
__global__ __launch_bounds__(TB) void k_wide_half(const float4 *__restrict__ wide, uint32_t n_wide, float cx, float cy,
                                                  float cz, float rsx, float rsy, float rsz, uint4 *__restrict__ out)
{
    const uint32_t i = blockIdx.x * TB + threadIdx.x;
    if (i >= n_wide) return;
    const float4 *nd = wide + 8 * (size_t)i;
    const float c[3] = { cx, cy, cz }, rs[3] = { rsx, rsy, rsz };
    uint32_t d[12];
    for (int ax = 0; ax < 3; ax++) {
        const float4 lo = nd[ax], hi = nd[3 + ax];
        const float l[4] = { lo.x, lo.y, lo.z, lo.w }, h[4] = { hi.x, hi.y, hi.z, hi.w };
        uint32_t hl[4], hh[4];
        for (int k = 0; k < 4; k++) {
            hl[k] = __half_as_ushort(__float2half_rd((l[k] - c[ax]) * rs[ax] - 3.814697265625e-06f));
            hh[k] = __half_as_ushort(__float2half_ru((h[k] - c[ax]) * rs[ax] + 3.814697265625e-06f));
        }
        d[2 * ax + 0] = hl[0] | (hl[1] << 16); d[2 * ax + 1] = hl[2] | (hl[3] << 16);
        d[6 + 2 * ax + 0] = hh[0] | (hh[1] << 16); d[6 + 2 * ax + 1] = hh[2] | (hh[3] << 16);
    }
    const float4 cw = nd[6];
    uint4 *o = out + 4 * (size_t)i;
    o[0] = make_uint4(d[0], d[1], d[2], d[3]);
    o[1] = make_uint4(d[4], d[5], d[6], d[7]);
    o[2] = make_uint4(d[8], d[9], d[10], d[11]);
    o[3] = make_uint4(__float_as_uint(cw.x), __float_as_uint(cw.y), __float_as_uint(cw.z), __float_as_uint(cw.w));
}

// (re)builds s->d_wide16 from the BVH4 that is currently traversed
static pt_status make_wide16(pt_scene *s)
{
    pt_ctx *ctx = s->ctx;
    hipStream_t st = ctx->stream;
    norm_box(s->bmin, s->bmax, s->norm_c, s->norm_s, s->norm_rs);
    (void)hipFree(s->d_wide16);
    s->d_wide16 = nullptr;
    PT_HIP(ctx, hipMalloc((void **)&s->d_wide16, 64 * (size_t)s->n_wide));
    k_wide_half<<<(s->n_wide + TB - 1) / TB, TB, 0, st>>>(s->d_wide, s->n_wide, s->norm_c[0], s->norm_c[1], s->norm_c[2], s->norm_rs[0],
                                                         s->norm_rs[1], s->norm_rs[2], reinterpret_cast<uint4 *>(s->d_wide16));
    PT_HIP(ctx, hipStreamSynchronize(st));
    PT_HIP(ctx, hipGetLastError());
    return PT_OK;
}

// triangle boxes from the de-indexed triangles (the same float operations as k_gather)
__global__ __launch_bounds__(TB) void k_tri_boxes(const float4 *__restrict__ tri_orig, uint32_t n, float4 *__restrict__ tlo,
                                                  float4 *__restrict__ thi)
{
    const uint32_t t = blockIdx.x * TB + threadIdx.x;
    if (t >= n) return;
    const float4 a = tri_orig[3 * (size_t)t + 0], b = tri_orig[3 * (size_t)t + 1], c = tri_orig[3 * (size_t)t + 2];
    tlo[t] = make_float4(fminf(fminf(a.x, b.x), c.x), fminf(fminf(a.y, b.y), c.y), fminf(fminf(a.z, b.z), c.z), 0.f);
    thi[t] = make_float4(fmaxf(fmaxf(a.x, b.x), c.x), fmaxf(fmaxf(a.y, b.y), c.y), fmaxf(fmaxf(a.z, b.z), c.z), 0.f);
}

// Everything that hangs off the binary tree -- the tree itself (LBVH, or its PLOC rebuild for big scenes under
// ePreferFastTrace), the BVH4 in both node formats, optionally the 8-wide nodes, the per-triangle tables in the traversed
// leaf order -- built (or rebuilt: quality change, first request for the 8-wide nodes) from the kept triangles.
static void free_tree_products(pt_scene *s)
{
    (void)hipFree(s->d_nodes); (void)hipFree(s->d_keys); (void)hipFree(s->d_prim_of); (void)hipFree(s->d_prim_of_sah);
    (void)hipFree(s->d_wide_lbvh ? s->d_wide_lbvh : s->d_wide); (void)hipFree(s->d_wide_sah);
    (void)hipFree(s->d_wide16); (void)hipFree(s->d_wide16t);
    (void)hipFree(s->d_wide8); (void)hipFree(s->d_prim_of8); (void)hipFree(s->d_tri4_8); (void)hipFree(s->d_shade64_8); (void)hipFree(s->d_ke4_8);
    s->d_nodes = nullptr; s->d_keys = nullptr; s->d_prim_of = s->d_prim_of_sah = nullptr;
    s->d_wide = s->d_wide_lbvh = s->d_wide_sah = nullptr; s->d_wide16 = nullptr; s->d_wide16t = nullptr;
    s->d_wide8 = nullptr; s->d_prim_of8 = nullptr; s->d_tri4_8 = s->d_shade64_8 = s->d_ke4_8 = nullptr;
    s->n_wide8 = s->levels8 = 0; s->n_wide16t = s->levels4t = 0;
}

static pt_status build_tree_products_unguarded(pt_scene *s, uint32_t quality, bool want8);

// A rebuild frees the old products first (peak memory = one set, and a scene of 8 M triangles holds 2.6 GB of them), so a
// rebuild that fails part-way -- out of memory beside a 70-100 GB film workspace is the plausible case -- leaves the scene
// WITHOUT a tree.  It is then marked broken: every product pointer null, every count zero, and plan_extend / pt_scene_read_* /
// pt_scene_set_instances answer PT_ERR_UNSUPPORTED instead of launching kernels on null tables.  The triangles and materials
// (d_tri_orig, d_faces) are untouched, so a later pt_scene_set_bvh_quality -- or the next render's request for the 8-wide
// nodes -- can build again; success clears the mark.
static pt_status build_tree_products(pt_scene *s, uint32_t quality, bool want8)
{
    const pt_status rc = build_tree_products_unguarded(s, quality, want8);
    if (rc != PT_OK) {
        (void)hipGetLastError();  // an out-of-memory error is sticky until read
        free_tree_products(s);
        s->n_nodes = s->n_wide = s->n_wide_lbvh = 0;
        s->stack_need = s->stack_need_lbvh = 0xFFFFFFFFu;
        s->device_bytes = s->device_bytes8 = 0;
        s->quality = quality;  // what the scene is meant to have: ptb_repair / the next pt_scene_set_bvh_quality build exactly that
        s->broken = true;
    } else {
        s->broken = false;
    }
    return rc;
}

static pt_status build_tree_products_unguarded(pt_scene *s, uint32_t quality, bool want8)
{
    pt_ctx *ctx = s->ctx;
    hipStream_t st = ctx->stream;
    const uint32_t n = s->n_tris, gt = (n + TB - 1) / TB;
    free_tree_products(s);
    DevBuf<float4> d_tlo, d_thi;
    PT_HIP(ctx, d_tlo.alloc(n));
    PT_HIP(ctx, d_thi.alloc(n));
    PT_HIP(ctx, hipEventRecord(ctx->ev_a, st));
    k_tri_boxes<<<gt, TB, 0, st>>>(s->d_tri_orig, n, d_tlo.p, d_thi.p);
    const bool ploc = quality == PT_BVH_PREFER_FAST_TRACE && n > PT_SAH_MAX_TRIS;
    BvhOut o;
    pt_status rc = build_bvh(ctx, d_tlo.p, d_thi.p, n, PT_BLAS_LEAF_MAX, o, 2 | (want8 ? 1 : 0), ploc);
    s->d_keys = o.d_keys; s->d_prim_of = o.d_prim_of; s->d_nodes = o.d_nodes; s->d_wide = o.d_wide;  // freed by pt_scene_destroy
    s->d_prim_of_sah = o.d_prim_q;
    s->d_wide8 = o.d_wide8; s->n_wide8 = o.n_wide8; s->levels8 = o.levels8;
    s->d_wide16t = o.d_wide16t; s->n_wide16t = o.n_wide16t; s->levels4t = o.levels4t;
    DevBuf<uint32_t> d_order8;
    d_order8.p = o.d_order8;
    s->d_wide_lbvh = s->d_wide;
    if (rc != PT_OK) return rc;
    s->n_nodes = o.n_nodes; s->n_wide = o.n_wide; s->height = o.height; s->height_tree = o.height_tree; s->stack_need = o.stack_need;
    s->n_wide_lbvh = s->n_wide; s->stack_need_lbvh = s->stack_need;
    for (int k = 0; k < 3; k++) { s->bmin[k] = o.bmin[k]; s->bmax[k] = o.bmax[k]; }
    s->bvh4_builder = o.d_prim_q ? 2u : 0u;
    s->area_lbvh = o.area_lbvh; s->area_ploc = o.area_ploc;
    s->pair_leaves = PT_BLAS_LEAF_MAX == 1u;
    const uint32_t *order = o.d_prim_q ? o.d_prim_q : s->d_prim_of;   // the traversed leaf order
    k_pack<<<gt, TB, 0, st>>>(s->d_tri_orig, s->d_faces, order, n, s->d_tri4, s->d_shade4, s->d_shade64, s->d_ke4, s->d_frame4);
    if (s->d_wide8) {  // the 8-wide tree's own triangle order: its per-triangle tables (the LDS-sized shade4 is never used with it)
        PT_HIP(ctx, hipMalloc((void **)&s->d_prim_of8, sizeof(uint32_t) * (size_t)n));
        PT_HIP(ctx, hipMalloc((void **)&s->d_tri4_8, sizeof(float4) * 3 * (size_t)n));
        PT_HIP(ctx, hipMalloc((void **)&s->d_shade64_8, sizeof(float4) * 4 * (size_t)n));
        PT_HIP(ctx, hipMalloc((void **)&s->d_ke4_8, sizeof(float4) * (size_t)n));
        DevBuf<float4> d_shade4_scratch;
        PT_HIP(ctx, d_shade4_scratch.alloc(3 * (size_t)n));
        k_compose<<<gt, TB, 0, st>>>(d_order8.p, order, n, s->d_prim_of8);
        k_pack<<<gt, TB, 0, st>>>(s->d_tri_orig, s->d_faces, s->d_prim_of8, n, s->d_tri4_8, d_shade4_scratch.p, s->d_shade64_8, s->d_ke4_8);
        PT_HIP(ctx, hipStreamSynchronize(st));
    }
    PT_HIP(ctx, hipEventRecord(ctx->ev_b, st));
    PT_HIP(ctx, hipStreamSynchronize(st));
    PT_HIP(ctx, hipGetLastError());
    PT_HIP(ctx, hipEventElapsedTime(&s->build_ms, ctx->ev_a, ctx->ev_b));
    // resident bytes of the BVH4 path: triangle tables (tri4 48 + shade4 48 + shade64 64 + ke4 16 + frames 32 B each), the kept
    // source arrays a rebuild re-packs from (d_tri_orig 48 + d_faces 24) + the 128-B and the two 64-B node arrays; of the
    // 8-wide path: its tables + nodes
    s->device_bytes = (uint64_t)n * (48 + 48 + 64 + 16 + 32 + PT_SOURCE_BYTES_PER_TRI) + 128ull * s->n_wide + 64ull * s->n_wide + 64ull * s->n_wide16t;
    s->device_bytes8 = s->d_wide8 ? (uint64_t)n * (48 + 64 + 16 + 4) + 64ull * s->n_wide8 : 0ull;
    s->quality = quality;
    return make_wide16(s);
}

// PT_EXTEND_HBM8 / pt_tuning.hbm8 on a scene that was built without the 8-wide nodes: build them now (extend_launch.hip asks)
pt_status ptb_ensure_wide8(pt_scene *s)
{
    if (s->d_wide8 || s->n_tris < 2 || s->n_inst) return PT_OK;
    PT_HIP(s->ctx, hipStreamSynchronize(s->ctx->stream));
    return build_tree_products(s, s->quality, true);
}

// a scene whose last rebuild failed (above) gets one more try per render / trace / read-back: the film whose workspace
// crowded it out may be gone by now
pt_status ptb_repair(pt_scene *s)
{
    if (!s->broken) return PT_OK;
    if (s->n_inst) { s->ctx->err = PT_BROKEN_SCENE_MSG; return PT_ERR_UNSUPPORTED; }  // (cannot happen: rebuilds are refused on instanced scenes)
    PT_HIP(s->ctx, hipStreamSynchronize(s->ctx->stream));
    const pt_status rc = build_tree_products(s, s->quality, s->ctx->tune.hbm8 != 0);
    if (rc != PT_OK) s->ctx->err = std::string(PT_BROKEN_SCENE_MSG) + " [" + s->ctx->err + "]";
    return rc;
}

pt_status ptb_build_scene(pt_scene *s, const float *h_vertices, uint32_t n_verts, const uint32_t *h_indices,
                          uint32_t n_tris, const float *h_faces)
{
    pt_ctx *ctx = s->ctx;
    hipStream_t st = ctx->stream;
    const uint32_t n = n_tris;
    const uint32_t gt = (n + TB - 1) / TB;
    DevBuf<float> d_vert;
    DevBuf<uint32_t> d_idx;
    DevBuf<float4> d_tlo, d_thi;
    PT_HIP(ctx, d_vert.alloc(3 * (size_t)n_verts));
    PT_HIP(ctx, d_idx.alloc(3 * (size_t)n));
    PT_HIP(ctx, d_tlo.alloc(n));
    PT_HIP(ctx, d_thi.alloc(n));
    s->n_tris = n;
    // the de-indexed triangles and the per-face materials stay resident (72 B per triangle): a change of the BVH quality,
    // or the first request for the 8-wide nodes, re-packs the tables from them in another leaf order
    PT_HIP(ctx, hipMalloc((void **)&s->d_tri_orig, sizeof(float4) * 3 * (size_t)n));
    PT_HIP(ctx, hipMalloc((void **)&s->d_faces, sizeof(float) * 6 * (size_t)n));
    PT_HIP(ctx, hipMalloc((void **)&s->d_tri4, sizeof(float4) * 3 * (size_t)n));
    PT_HIP(ctx, hipMalloc((void **)&s->d_shade4, sizeof(float4) * 3 * (size_t)n));
    PT_HIP(ctx, hipMalloc((void **)&s->d_shade64, sizeof(float4) * 4 * (size_t)n));
    PT_HIP(ctx, hipMalloc((void **)&s->d_ke4, sizeof(float4) * (size_t)n));
    PT_HIP(ctx, hipMalloc((void **)&s->d_frame4, sizeof(float4) * 2 * (size_t)n));
    PT_HIP(ctx, hipMemcpyAsync(d_vert.p, h_vertices, sizeof(float) * 3 * (size_t)n_verts, hipMemcpyHostToDevice, st));
    PT_HIP(ctx, hipMemcpyAsync(d_idx.p, h_indices, sizeof(uint32_t) * 3 * (size_t)n, hipMemcpyHostToDevice, st));
    PT_HIP(ctx, hipMemcpyAsync(s->d_faces, h_faces, sizeof(float) * 6 * (size_t)n, hipMemcpyHostToDevice, st));
    PT_HIP(ctx, hipStreamSynchronize(st));  // pageable host sources are done with
    k_gather<<<gt, TB, 0, st>>>(d_vert.p, d_idx.p, n, s->d_tri_orig, d_tlo.p, d_thi.p);
    // the tree of the default quality (ePreferFastTrace, main.cpp:419): PLOC for big scenes; small scenes get the LBVH
    // here and the exact surface-area BVH4 below.  The 8-wide nodes only when the context asks AUTO to use them.
    // (small scenes get the 8-wide nodes at once -- a few KB; big ones on first request: 260 B per triangle nobody else needs)
    // ... and scenes AUTO walks through them: more than 1 MiB of BVH4 nodes + records, ~96 B per triangle (extend_launch.hip ptw_plan_extend)
    pt_status rc = build_tree_products(s, PT_BVH_PREFER_FAST_TRACE, ctx->tune.hbm8 == 1 || n <= PT_SAH_MAX_TRIS || (ctx->tune.hbm8 != 0 && 96ull * n > (1ull << 20)));
    if (rc != PT_OK) return rc;
    {   // emitters for the NEE pipeline: normal as closesthit.rchit:43-48, area = |cross| / 2, cdf = running float sum of the
        // areas in primitive order (this file is compiled with -ffp-contract=off on the host side too)
        std::vector<float4> lights;
        float run = 0.f;
        for (uint32_t t = 0; t < n; t++) {
            const float *f = h_faces + 6 * (size_t)t;
            if (!(f[3] != 0.f || f[4] != 0.f || f[5] != 0.f)) continue;
            const float *a = h_vertices + 3 * (size_t)h_indices[3 * (size_t)t + 0], *b = h_vertices + 3 * (size_t)h_indices[3 * (size_t)t + 1],
                        *c = h_vertices + 3 * (size_t)h_indices[3 * (size_t)t + 2];
            const float e1[3] = { b[0] - a[0], b[1] - a[1], b[2] - a[2] }, e2[3] = { c[0] - a[0], c[1] - a[1], c[2] - a[2] };
            const float cx = e1[1] * e2[2] - e1[2] * e2[1], cy = e1[2] * e2[0] - e1[0] * e2[2], cz = e1[0] * e2[1] - e1[1] * e2[0];
            const float len = sqrtf((cx * cx + cy * cy) + cz * cz);
            run = run + 0.5f * len;
            lights.push_back(make_float4(a[0], a[1], a[2], run));
            lights.push_back(make_float4(b[0], b[1], b[2], 0.f));
            lights.push_back(make_float4(c[0], c[1], c[2], 0.f));
            lights.push_back(make_float4(-(cx / len), -(cy / len), -(cz / len), 0.f));
            lights.push_back(make_float4(f[3], f[4], f[5], 0.f));
        }
        s->n_lights = (uint32_t)(lights.size() / 5);
        s->light_area = run;
        s->h_lights = lights;
        if (s->n_lights) {
            PT_HIP(ctx, hipMalloc((void **)&s->d_lights, sizeof(float4) * lights.size()));
            PT_HIP(ctx, hipMemcpy(s->d_lights, lights.data(), sizeof(float4) * lights.size(), hipMemcpyHostToDevice));
        }
    }
    if (n <= PT_SAH_MAX_TRIS) {
        // small scene: keep what a rebuild of the BVH4 in another leaf order needs, then apply the default
        // quality (ePreferFastTrace, main.cpp:419)
        std::vector<float4> lo(n), hi(n);
        PT_HIP(ctx, hipMemcpy(lo.data(), d_tlo.p, sizeof(float4) * n, hipMemcpyDeviceToHost));
        PT_HIP(ctx, hipMemcpy(hi.data(), d_thi.p, sizeof(float4) * n, hipMemcpyDeviceToHost));
        s->h_tlo.resize(3 * (size_t)n);
        s->h_thi.resize(3 * (size_t)n);
        for (uint32_t i = 0; i < n; i++) {
            s->h_tlo[3 * i + 0] = lo[i].x; s->h_tlo[3 * i + 1] = lo[i].y; s->h_tlo[3 * i + 2] = lo[i].z;
            s->h_thi[3 * i + 0] = hi[i].x; s->h_thi[3 * i + 1] = hi[i].y; s->h_thi[3 * i + 2] = hi[i].z;
        }
        // fan pairs as a loader emits them for quads: the next triangle starts at the same vertex and continues from
        // this one's third (bitwise equal coordinates); greedy, non-overlapping
        s->h_pair.assign(n, 0);
        auto vtx = [&](uint32_t tri, int k) { return h_vertices + 3 * (size_t)h_indices[3 * (size_t)tri + k]; };
        for (uint32_t i = 0; i + 1 < n; i++) {
            const bool same = std::memcmp(vtx(i, 0), vtx(i + 1, 0), 12) == 0 && std::memcmp(vtx(i, 2), vtx(i + 1, 1), 12) == 0;
            if (same) { s->h_pair[i] = 1; i++; }
        }
        const float lbvh_ms = s->build_ms;
        PT_HIP(ctx, hipEventRecord(ctx->ev_a, st));
        const pt_status q = ptb_set_bvh_quality(s, PT_BVH_PREFER_FAST_TRACE);
        if (q != PT_OK) return q;
        PT_HIP(ctx, hipEventRecord(ctx->ev_b, st));
        PT_HIP(ctx, hipStreamSynchronize(st));
        float sah_ms = 0.f;
        PT_HIP(ctx, hipEventElapsedTime(&sah_ms, ctx->ev_a, ctx->ev_b));
        s->build_ms = lbvh_ms + sah_ms;
    }
    return PT_OK;
}

// Chooses the BVH4 that is traversed (pt_internal.h).  Re-packs the per-triangle tables in its leaf order.
pt_status ptb_set_bvh_quality(pt_scene *s, uint32_t quality)
{
    pt_ctx *ctx = s->ctx;
    if (quality > PT_BVH_PREFER_FAST_BUILD) { ctx->err = "unknown BVH quality"; return PT_ERR_INVALID_ARG; }
    if (s->n_inst) { ctx->err = "set the BVH quality before the instances"; return PT_ERR_UNSUPPORTED; }
    if (s->n_tris > PT_SAH_MAX_TRIS) {  // big scene: PLOC tree <-> LBVH, everything that hangs off the tree is rebuilt
        if (quality == s->quality && !s->broken) return PT_OK;
        PT_HIP(ctx, hipStreamSynchronize(ctx->stream));
        return build_tree_products(s, quality, s->d_wide8 != nullptr);
    }
    const bool want_sah = quality == PT_BVH_PREFER_FAST_TRACE && s->n_tris <= PT_SAH_MAX_TRIS && s->d_tri_orig;
    if (want_sah == (s->bvh4_builder == 1u)) return PT_OK;
    hipStream_t st = ctx->stream;
    const uint32_t n = s->n_tris;
    if (want_sah && !s->d_wide_sah) {
        float scale = 0.f;  // leaf_pad() of the device build, same float operations
        for (int k = 0; k < 3; k++) scale = fmaxf(scale, fmaxf(fabsf(s->bmin[k]), fabsf(s->bmax[k])));
        const float pad = scale * 3.814697265625e-06f;
        std::vector<uint32_t> rows, order;
        // one primitive per leaf, a primitive being a triangle or a quad's two halves (pt_tuning.pair_leaves = 0: the former
        // rule, up to PT_SAH_LEAF_MAX independent triangles per leaf where splitting does not pay); built on the device
        const bool pairs = ctx->tune.pair_leaves != 0;
        const pt_status rc8 = pt_sah_build_bvh4_device(ctx, s->h_tlo.data(), s->h_thi.data(), n, pairs ? s->h_pair.data() : nullptr, pad,
                                                       pairs ? 1u : PT_SAH_LEAF_MAX, rows, order);
        if (rc8 != PT_OK) return rc8;
        s->sah_pair_leaves = pairs;
        if (order.size() != n || rows.empty()) { ctx->err = "internal: SAH build lost triangles"; return PT_ERR_HIP; }
        s->n_wide_sah = (uint32_t)(rows.size() / 32);
        s->stack_need_sah = pt_wide_stack_need(rows);
        PT_HIP(ctx, hipMalloc((void **)&s->d_wide_sah, rows.size() * sizeof(uint32_t)));
        PT_HIP(ctx, hipMalloc((void **)&s->d_prim_of_sah, sizeof(uint32_t) * n));
        PT_HIP(ctx, hipMemcpy(s->d_wide_sah, rows.data(), rows.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
        PT_HIP(ctx, hipMemcpy(s->d_prim_of_sah, order.data(), sizeof(uint32_t) * n, hipMemcpyHostToDevice));
    }
    PT_HIP(ctx, hipStreamSynchronize(st));  // nothing may still be traversing the old tables
    if (want_sah) {
        s->d_wide = s->d_wide_sah; s->n_wide = s->n_wide_sah; s->stack_need = s->stack_need_sah; s->bvh4_builder = 1;
        s->pair_leaves = s->sah_pair_leaves;
    } else {
        s->d_wide = s->d_wide_lbvh; s->n_wide = s->n_wide_lbvh; s->stack_need = s->stack_need_lbvh; s->bvh4_builder = 0;
        s->pair_leaves = PT_BLAS_LEAF_MAX == 1u;  // 1-triangle leaves are the degenerate case of the pair kernel
    }
    k_pack<<<(n + TB - 1) / TB, TB, 0, st>>>(s->d_tri_orig, s->d_faces, want_sah ? s->d_prim_of_sah : s->d_prim_of, n, s->d_tri4,
                                           s->d_shade4, s->d_shade64, s->d_ke4, s->d_frame4);
    PT_HIP(ctx, hipStreamSynchronize(st));
    PT_HIP(ctx, hipGetLastError());
    (void)hipFree(s->d_inst_frame);  // (the triangle order changed: ptb_ensure_inst_frames builds it again on the next render)
    s->d_inst_frame = nullptr;
    s->device_bytes = (uint64_t)n * (48 + 48 + 64 + 16 + 32 + PT_SOURCE_BYTES_PER_TRI) + 128ull * s->n_wide + 64ull * s->n_wide + 64ull * s->n_wide16t;
    s->quality = quality;
    return make_wide16(s);
}

void ptb_free_scene_buffers(pt_scene *s)
{
    (void)hipFree(s->d_tri4); (void)hipFree(s->d_shade4); (void)hipFree(s->d_nodes);
    (void)hipFree(s->d_shade64); (void)hipFree(s->d_ke4); (void)hipFree(s->d_frame4);
    s->d_shade64 = s->d_ke4 = s->d_frame4 = nullptr;
    (void)hipFree(s->d_wide_lbvh ? s->d_wide_lbvh : s->d_wide);  // d_wide aliases d_wide_lbvh or d_wide_sah
    (void)hipFree(s->d_wide_sah); (void)hipFree(s->d_prim_of_sah);
    (void)hipFree(s->d_keys); (void)hipFree(s->d_prim_of);
    (void)hipFree(s->d_tri_orig); (void)hipFree(s->d_faces); (void)hipFree(s->d_wide16);
    s->d_wide16 = nullptr;
    (void)hipFree(s->d_wide16t);
    s->d_wide16t = nullptr;
    (void)hipFree(s->d_lights);
    s->d_lights = nullptr; s->n_lights = 0;
    (void)hipFree(s->d_wide8); (void)hipFree(s->d_prim_of8); (void)hipFree(s->d_tri4_8); (void)hipFree(s->d_shade64_8); (void)hipFree(s->d_ke4_8);
    s->d_wide8 = nullptr; s->d_prim_of8 = nullptr; s->d_tri4_8 = s->d_shade64_8 = s->d_ke4_8 = nullptr;
    s->d_tri4 = s->d_shade4 = s->d_nodes = s->d_wide = s->d_wide_lbvh = s->d_wide_sah = s->d_tri_orig = nullptr;
    s->d_prim_of_sah = s->d_prim_of = nullptr; s->d_keys = nullptr; s->d_faces = nullptr;
}

// ---- instances: TLAS over world boxes of the transformed BLAS root box --------------------------
// (beyond the reference, which builds ONE identity instance, main.cpp:515-538)
__global__ __launch_bounds__(TB) void k_inst_boxes(const float4 *__restrict__ blas_wide, const float4 *__restrict__ inst6,
                                                   uint32_t n, float4 *__restrict__ tlo, float4 *__restrict__ thi)
{
    const uint32_t i = blockIdx.x * TB + threadIdx.x;
    if (i >= n) return;
    // object box = union of the (padded) child boxes of the BLAS root
    float omin[3] = { INFINITY, INFINITY, INFINITY }, omax[3] = { -INFINITY, -INFINITY, -INFINITY };
    const float4 lx = blas_wide[0], ly = blas_wide[1], lz = blas_wide[2], hx = blas_wide[3], hy = blas_wide[4], hz = blas_wide[5];
    const float4 cw = blas_wide[6];
    const float l[3][4] = { { lx.x, lx.y, lx.z, lx.w }, { ly.x, ly.y, ly.z, ly.w }, { lz.x, lz.y, lz.z, lz.w } };
    const float h[3][4] = { { hx.x, hx.y, hx.z, hx.w }, { hy.x, hy.y, hy.z, hy.w }, { hz.x, hz.y, hz.z, hz.w } };
    const uint32_t w[4] = { __float_as_uint(cw.x), __float_as_uint(cw.y), __float_as_uint(cw.z), __float_as_uint(cw.w) };
    for (int c = 0; c < 4; c++)
        if (w[c] != PT_MISS)
            for (int k = 0; k < 3; k++) { omin[k] = fminf(omin[k], l[k][c]); omax[k] = fmaxf(omax[k], h[k][c]); }
    const float4 m0 = inst6[6 * (size_t)i + 0], m1 = inst6[6 * (size_t)i + 1], m2 = inst6[6 * (size_t)i + 2];
    float mn[3] = { INFINITY, INFINITY, INFINITY }, mx[3] = { -INFINITY, -INFINITY, -INFINITY };
    for (int c = 0; c < 8; c++) {
        const float px = (c & 1) ? omax[0] : omin[0], py = (c & 2) ? omax[1] : omin[1], pz = (c & 4) ? omax[2] : omin[2];
        const float wx = ((m0.x * px + m0.y * py) + m0.z * pz) + m0.w;
        const float wy = ((m1.x * px + m1.y * py) + m1.z * pz) + m1.w;
        const float wz = ((m2.x * px + m2.y * py) + m2.z * pz) + m2.w;
        mn[0] = fminf(mn[0], wx); mn[1] = fminf(mn[1], wy); mn[2] = fminf(mn[2], wz);
        mx[0] = fmaxf(mx[0], wx); mx[1] = fmaxf(mx[1], wy); mx[2] = fmaxf(mx[2], wz);
    }
    tlo[i] = make_float4(mn[0], mn[1], mn[2], 0.f);
    thi[i] = make_float4(mx[0], mx[1], mx[2], 0.f);
}

__global__ __launch_bounds__(TB) void k_inst_sort(const float4 *__restrict__ inst6, const uint32_t *__restrict__ prim_of,
                                                  uint32_t n, float4 *__restrict__ sorted6)
{
    const uint32_t pos = blockIdx.x * TB + threadIdx.x;
    if (pos >= n) return;
    const uint32_t id = prim_of[pos];
    for (int k = 0; k < 6; k++) sorted6[6 * (size_t)pos + k] = inst6[6 * (size_t)id + k];
}

// world -> object matrix: adjugate / determinant in binary64, rounded once to float
static void invert_3x4(const float m[12], float inv[12])
{
    const double a00 = m[0], a01 = m[1], a02 = m[2], a10 = m[4], a11 = m[5], a12 = m[6], a20 = m[8], a21 = m[9], a22 = m[10];
    const double c00 = a11 * a22 - a12 * a21, c01 = a12 * a20 - a10 * a22, c02 = a10 * a21 - a11 * a20;
    const double det = (a00 * c00 + a01 * c01) + a02 * c02;
    const double i00 = c00 / det, i01 = (a02 * a21 - a01 * a22) / det, i02 = (a01 * a12 - a02 * a11) / det;
    const double i10 = c01 / det, i11 = (a00 * a22 - a02 * a20) / det, i12 = (a02 * a10 - a00 * a12) / det;
    const double i20 = c02 / det, i21 = (a01 * a20 - a00 * a21) / det, i22 = (a00 * a11 - a01 * a10) / det;
    const double tx = m[3], ty = m[7], tz = m[11];
    inv[0] = (float)i00; inv[1] = (float)i01; inv[2] = (float)i02;  inv[3] = (float)(-((i00 * tx + i01 * ty) + i02 * tz));
    inv[4] = (float)i10; inv[5] = (float)i11; inv[6] = (float)i12;  inv[7] = (float)(-((i10 * tx + i11 * ty) + i12 * tz));
    inv[8] = (float)i20; inv[9] = (float)i21; inv[10] = (float)i22; inv[11] = (float)(-((i20 * tx + i21 * ty) + i22 * tz));
}

// World-space normal and tangent of every (instance, triangle): what k_shade's instanced branch used to evaluate per hit -- the
// normal by the inverse transpose, renormalised (a square root and three true divides), and createCoordinateSystem on it (another
// square root and two divides) -- evaluated ONCE with exactly those operations (shade_kernels.hip k_shade; pt_math.h tangent_frame),
// so the bits are the same.  32 B per entry: {n.xyz, T.x} {T.yz, -, -}; the bitangent is the cross product k_shade forms anyway.
// Instance order = d_inst6's (TLAS leaf order), triangle order = d_shade4's (BVH4 leaf order): rebuilt when either changes.
__global__ __launch_bounds__(TB) void k_inst_frames(const float4 *__restrict__ inst6, const float4 *__restrict__ shade4, uint32_t n_inst,
                                                    uint32_t n_tris, float4 *__restrict__ out)
{
    const size_t idx = (size_t)blockIdx.x * TB + threadIdx.x;
    if (idx >= (size_t)n_inst * n_tris) return;
    const uint32_t ip = (uint32_t)(idx / n_tris), pos = (uint32_t)(idx - (size_t)ip * n_tris);
    const float4 s0 = shade4[3 * (size_t)pos];
    const float4 i0 = inst6[6 * (size_t)ip + 3], i1 = inst6[6 * (size_t)ip + 4], i2 = inst6[6 * (size_t)ip + 5];
    const float nx = (i0.x * s0.x + i1.x * s0.y) + i2.x * s0.z;
    const float ny = (i0.y * s0.x + i1.y * s0.y) + i2.y * s0.z;
    const float nz = (i0.z * s0.x + i1.z * s0.y) + i2.z * s0.z;
    const float l = ptm::fsqrt((nx * nx + ny * ny) + nz * nz);
    const ptm::f3 n = { ptm::fdiv(nx, l), ptm::fdiv(ny, l), ptm::fdiv(nz, l) };
    ptm::f3 T, B;
    ptm::tangent_frame(n, T, B);
    out[2 * idx + 0] = make_float4(n.x, n.y, n.z, T.x);
    out[2 * idx + 1] = make_float4(T.y, T.z, 0.f, 0.f);
}

pt_status ptb_ensure_inst_frames(pt_scene *s)
{
    pt_ctx *ctx = s->ctx;
    if (!s->n_inst || s->d_inst_frame) return PT_OK;
    const size_t entries = (size_t)s->n_inst * s->n_tris;
    if (entries * 32 > (512ull << 20)) return PT_OK;  // (a table beyond the caches would cost more than it saves: the per-hit transform stays)
    PT_HIP(ctx, hipMalloc((void **)&s->d_inst_frame, 32 * entries));
    k_inst_frames<<<(unsigned)((entries + TB - 1) / TB), TB, 0, ctx->stream>>>(s->d_inst6, s->d_shade4, s->n_inst, s->n_tris, s->d_inst_frame);
    PT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    PT_HIP(ctx, hipGetLastError());
    return PT_OK;
}

// Emitters of the NEE pipeline for an instanced scene: every instance's copy, in gl_InstanceID order, vertices taken to world
// space by the instance's matrix with the operation order of the shading transform; normal and area from the world-space
// triangle; one running cdf over all of them (the tests' CPU checker restates this loop).  Built on the first NEE render of
// the instance set -- 80 B per (instance, emitter) on host and device, nothing a scene that never samples lights should pay --
// and refused beyond 2^24 copies (1.3 GB; the float running sum of the areas stops resolving small emitters well before).
pt_status ptb_ensure_inst_lights(pt_scene *s)
{
    pt_ctx *ctx = s->ctx;
    if (!s->n_inst || !s->n_lights || s->d_lights_inst) return PT_OK;
    const uint64_t copies = (uint64_t)s->n_inst * s->n_lights;
    if (copies > (1ull << 24) || s->h_xforms.size() != 12 * (size_t)s->n_inst) {
        ctx->err = "the NEE pipeline would need " + std::to_string(copies) + " world-space emitter copies (instances x emitters); the limit is 16 777 216";
        return PT_ERR_UNSUPPORTED;
    }
    const uint32_t n = s->n_inst;
    const float *xforms3x4 = s->h_xforms.data();
    std::vector<float4> wl;
    wl.reserve(5 * (size_t)copies);
    float run = 0.f;
    for (uint32_t i = 0; i < n; i++) {
        const float *m = xforms3x4 + 12 * (size_t)i;
        for (uint32_t k = 0; k < s->n_lights; k++) {
            float w[3][3];
            for (int c = 0; c < 3; c++) {
                const float4 v = s->h_lights[5 * (size_t)k + c];
                for (int r = 0; r < 3; r++) w[c][r] = ((m[4 * r] * v.x + m[4 * r + 1] * v.y) + m[4 * r + 2] * v.z) + m[4 * r + 3];
            }
            const float e1[3] = { w[1][0] - w[0][0], w[1][1] - w[0][1], w[1][2] - w[0][2] }, e2[3] = { w[2][0] - w[0][0], w[2][1] - w[0][1], w[2][2] - w[0][2] };
            const float cx = e1[1] * e2[2] - e1[2] * e2[1], cy = e1[2] * e2[0] - e1[0] * e2[2], cz = e1[0] * e2[1] - e1[1] * e2[0];
            const float len = sqrtf((cx * cx + cy * cy) + cz * cz);
            run = run + 0.5f * len;
            wl.push_back(make_float4(w[0][0], w[0][1], w[0][2], run));
            wl.push_back(make_float4(w[1][0], w[1][1], w[1][2], 0.f));
            wl.push_back(make_float4(w[2][0], w[2][1], w[2][2], 0.f));
            wl.push_back(make_float4(-(cx / len), -(cy / len), -(cz / len), 0.f));
            wl.push_back(s->h_lights[5 * (size_t)k + 4]);
        }
    }
    const hipError_t e = hipMalloc((void **)&s->d_lights_inst, sizeof(float4) * wl.size());
    if (e != hipSuccess) {
        (void)hipGetLastError();
        s->d_lights_inst = nullptr;
        ctx->err = std::string("hipMalloc of the instanced emitter table: ") + hipGetErrorString(e);
        return e == hipErrorOutOfMemory ? PT_ERR_OOM : PT_ERR_HIP;
    }
    PT_HIP(ctx, hipMemcpy(s->d_lights_inst, wl.data(), sizeof(float4) * wl.size(), hipMemcpyHostToDevice));
    s->n_lights_inst = (uint32_t)copies;
    s->light_area_inst = run;
    return PT_OK;
}

void ptb_free_instances(pt_scene *s)
{
    s->h_xforms.clear();
    s->h_xforms.shrink_to_fit();
    (void)hipFree(s->d_inst_frame);
    s->d_inst_frame = nullptr;
    (void)hipFree(s->d_lights_inst);
    s->d_lights_inst = nullptr; s->n_lights_inst = 0; s->light_area_inst = 0.f;
    (void)hipFree(s->d_inst6); (void)hipFree(s->d_tlas_wide); (void)hipFree(s->d_tlas_prim_of); (void)hipFree(s->d_tlas16);
    s->d_inst6 = nullptr; s->d_tlas_wide = nullptr; s->d_tlas_prim_of = nullptr; s->d_tlas16 = nullptr; s->n_tlas16 = 0;
    s->n_inst = 0; s->n_tlas_wide = 0; s->tlas_height = 0;
}

pt_status ptb_set_instances(pt_scene *s, const float *xforms3x4, uint32_t n)
{
    pt_ctx *ctx = s->ctx;
    hipStream_t st = ctx->stream;
    PT_HIP(ctx, hipStreamSynchronize(st));
    ptb_free_instances(s);
    if (n == 0) return PT_OK;
    if (s->broken) { ctx->err = PT_BROKEN_SCENE_MSG; return PT_ERR_UNSUPPORTED; }
    std::vector<float> rec(24 * (size_t)n);
    for (uint32_t i = 0; i < n; i++) {
        const float *m = xforms3x4 + 12 * (size_t)i;
        for (int k = 0; k < 12; k++) {
            if (!(m[k] == m[k]) || __builtin_isinf(m[k])) { ctx->err = "instance matrix has a NaN/Inf"; return PT_ERR_INVALID_ARG; }
            rec[24 * (size_t)i + k] = m[k];
        }
        invert_3x4(m, &rec[24 * (size_t)i + 12]);
        for (int k = 12; k < 24; k++)
            if (!(rec[24 * (size_t)i + k] == rec[24 * (size_t)i + k]) || __builtin_isinf(rec[24 * (size_t)i + k])) {
                ctx->err = "instance matrix is singular";
                return PT_ERR_INVALID_ARG;
            }
    }
    DevBuf<float4> d_in, d_tlo, d_thi;
    PT_HIP(ctx, d_in.alloc(6 * (size_t)n));
    PT_HIP(ctx, d_tlo.alloc(n));
    PT_HIP(ctx, d_thi.alloc(n));
    PT_HIP(ctx, hipMemcpy(d_in.p, rec.data(), sizeof(float) * 24 * (size_t)n, hipMemcpyHostToDevice));
    const uint32_t g = (n + TB - 1) / TB;
    k_inst_boxes<<<g, TB, 0, st>>>(s->d_wide, d_in.p, n, d_tlo.p, d_thi.p);
    BvhOut o;
    // (n < 32768: also the top-down 64-B TLAS with 16-bit child codes that k_extend_inst16 walks)
    pt_status rc = build_bvh(ctx, d_tlo.p, d_thi.p, n, PT_TLAS_LEAF_MAX, o, (n > 1 && n < 32768u && PT_TLAS_LEAF_MAX == 1u) ? 6 : 0);
    (void)hipFree(o.d_keys);
    (void)hipFree(o.d_nodes);
    s->d_tlas_wide = o.d_wide;
    s->d_tlas_prim_of = o.d_prim_of;
    s->d_tlas16 = o.d_wide16t; s->n_tlas16 = o.n_wide16t; s->tlas16_levels = o.levels4t;
    for (int k = 0; k < 3; k++) { s->tlas_norm_c[k] = o.norm_c[k]; s->tlas_norm_s[k] = o.norm_s[k]; s->tlas_norm_rs[k] = o.norm_rs[k]; }
    if (rc != PT_OK) { ptb_free_instances(s); return rc; }
    PT_HIP(ctx, hipMalloc((void **)&s->d_inst6, sizeof(float4) * 6 * (size_t)n));
    // (the emitters' world-space copies for the NEE pipeline are made on that pipeline's first render: ptb_ensure_inst_lights)
    s->h_xforms.assign(xforms3x4, xforms3x4 + 12 * (size_t)n);
    k_inst_sort<<<g, TB, 0, st>>>(d_in.p, s->d_tlas_prim_of, n, s->d_inst6);
    PT_HIP(ctx, hipStreamSynchronize(st));
    PT_HIP(ctx, hipGetLastError());
    s->n_inst = n;
    s->n_tlas_wide = o.n_wide;
    s->tlas_height = o.height;
    return PT_OK;
}
