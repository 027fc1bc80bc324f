// ploc_build.hip -- the binary tree of big scenes rebuilt bottom-up by parallel locally-ordered clustering.
#include "bvh_build.h"

#include "device_scan.h"  // exclusive_scan

#include <vector>

namespace {

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


}  // namespace

// Rebuilds the binary tree over the Morton-ordered leaves by PLOC (kernels above), in place of the LBVH's arrays.
// In: leaf boxes box_lo/box_hi[0, n) and prim_of in Morton order.  Out: topo / range / parent_int / parent_leaf, boxes of
// leaves [0, n) and internal nodes [n, 2n - 1) in the NEW leaf order, d_prim_q (new position -> primitive id), height.
// Returns PT_ERR_UNSUPPORTED (and leaves the LBVH arrays untouched as far as the caller's later stages are concerned: they
// are only overwritten at the very end) if the clustering stalls, which the caller answers by keeping the LBVH.
// sum of the internal nodes' surface areas of a tree in the [pos] / [n + node] box layout (deterministic: partial sums added in order)
pt_status ptb_tree_area(pt_ctx *ctx, uint32_t n, const float4 *d_blo, const float4 *d_bhi, double *out)
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
pt_status ptb_ploc_refine(pt_ctx *ctx, uint32_t n, int radius, double area_lbvh, double *area_ploc, uint2 *d_topo, uint2 *d_range,
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
        const pt_status arc = ptb_tree_area(ctx, n, nblo.p, nbhi.p, area_ploc);   // (internal boxes do not depend on the leaf order)
        if (arc != PT_OK) return arc;
        // adopted only when clearly cheaper: on uniformly distributed, equally sized triangles (the soup of config C5) the
        // two sums are within 1 % of each other and the LBVH's balanced tree collapses into the better BVH4 (measured:
        // 36.2 against 38.9 node visits per ray, profiles/r03_probe_stress_scene.txt)
        if (!(*area_ploc < 0.01 * pt_tuned(ctx->tune.ploc_adopt_pct, 90, 0, 1000) * area_lbvh)) return PT_ERR_UNSUPPORTED;
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

