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
//   6. collapses   BVH4 (k_wide_*), and level by level from the same binary tree: the top-down BVH4 and the 8-wide tree (k_w4 / k_w8_*)
// (k_gather and the leaf-ordered records, k_pack, are scene_build.hip's; the PLOC rebuild of the binary tree is ploc_build.hip's.)
// The tree is fully determined by the input, so a CPU builder following the same rules yields
// bit-identical keys, order, topology and boxes (tests compare them).
#include "bvh_build.h"

#include <hip/hip_fp16.h>

#include "device_scan.h"  // block_exclusive_scan, k_scan_*, exclusive_scan

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace {

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



}  // namespace

// the normalisation of the fp16 node formats: x' = (x - c) * rs with c the centre and 1/rs the half extent of the scene box
void ptb_norm_box(const float *bmin, const float *bmax, float *c, float *sv, float *rs)
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

// top_down: bit 0 = also the BVH8 (+ its triangle order), bit 1 = also the top-down BVH4 in the 64-B format, bit 2 = that BVH4
// with 16-bit child codes (k_w4_emit compact16: the TLAS of k_extend_inst16; needs n < 32768)
// ploc: the binary tree is rebuilt by PLOC before the collapses (out.d_prim_q = its leaf order; out.d_prim_of, d_keys and
// d_nodes stay the LBVH's, for the parity read-back)
pt_status ptb_build_bvh(pt_ctx *ctx, const float4 *d_tlo, const float4 *d_thi, uint32_t n, uint32_t leaf_max, BvhOut &out, int top_down,
                        bool ploc)
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
        const pt_status arc = ptb_tree_area(ctx, n, d_blo.p, d_bhi.p, &out.area_lbvh);
        if (arc != PT_OK) return arc;
        out.area_tree = out.area_lbvh;
    }
    if (ploc && n > 2) {
        PT_HIP(ctx, hipMalloc((void **)&out.d_prim_q, sizeof(uint32_t) * (size_t)n));
        double area_ploc = 0.0;
        const pt_status prc = ptb_ploc_refine(ctx, n, pt_tuned(ctx->tune.ploc_radius, 8, 1, PLOC_R_MAX), out.area_lbvh, &area_ploc, d_topo.p, d_range.p,
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
    ptb_norm_box(out.bmin, out.bmax, out.norm_c, out.norm_s, out.norm_rs);
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
