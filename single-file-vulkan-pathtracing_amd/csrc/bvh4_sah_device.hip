// bvh4_sah_device.hip -- the "prefer fast trace" BVH4 of small scenes, built ON THE DEVICE.
//
// (Round 1 built the same tree on the host; that builder is gone -- tests/test_gpu_parity.py checks this one's trees
// structurally and through the hit records.)  Primitives = triangles or fan pairs, binary surface-area sweep over every split position of all three centroid
// orders, cost area(L) n(L) + area(R) n(R) in binary64, first minimum in (cost, axis, position) order, median split below
// depth 24; then the BVH4 as the cut of that tree with the least sum of internal-node areas (k_sah_emit), rows in the same format and order.
//
// Scenes of <= 2048 triangles; the Cornell box has 18 primitives.  No sorting: for a node
// of m primitives each of the 3m candidates (axis a, primitive j) is "everything whose (centroid_a, id) key is <= j's
// goes left" -- exactly the split positions of the sorted sweep -- and a thread evaluates it by one pass over the
// node's primitives that grows the two boxes and counts the left side; that count minus one IS j's position in the sorted
// order along a, so the winning axis' positions re-order the node's primitives for its children for free.  O(m^2) per
// node: one big workgroup for the nodes at the top, one small workgroup per subtree below (k_sah_top / k_sah_sub).
#include "pt_internal.h"
#include "pt_math.h"

#include <algorithm>
#include <vector>

namespace {

constexpr int TBD = 256;
constexpr int SAH_MAX_DEPTH = 24;

struct SahNode {
    int left, right;          // -1: leaf
    uint32_t first, count;    // range of `ids` (primitives)
    double lo[3], hi[3];
};

__device__ __forceinline__ double box_area_d(const double *lo, const double *hi)
{
    const double x = fmax(hi[0] - lo[0], 0.0), y = fmax(hi[1] - lo[1], 0.0), z = fmax(hi[2] - lo[2], 0.0);
    return 2.0 * (x * y + y * z + z * x);
}

struct Best { double cost; int axis; uint32_t k; };
__device__ __forceinline__ bool better(const Best &a, const Best &b)  // a before b in (cost, axis, position) order
{
    return a.cost < b.cost || (a.cost == b.cost && (a.axis < b.axis || (a.axis == b.axis && a.k < b.k)));
}

// prim boxes (binary64) from the triangle boxes; a primitive is one triangle or two consecutive ones
__global__ void k_sah_prims(const float *__restrict__ tlo, const float *__restrict__ thi, const uint32_t *__restrict__ prim_first,
                            const uint8_t *__restrict__ prim_tris, uint32_t np, double *__restrict__ plo, double *__restrict__ phi,
                            uint32_t *__restrict__ ids)
{
    const uint32_t p = blockIdx.x * TBD + threadIdx.x;
    if (p >= np) return;
    const uint32_t t = prim_first[p];
    for (int k = 0; k < 3; k++) {
        double lo = tlo[3 * (size_t)t + k], hi = thi[3 * (size_t)t + k];
        if (prim_tris[p] == 2) { lo = fmin(lo, (double)tlo[3 * (size_t)(t + 1) + k]); hi = fmax(hi, (double)thi[3 * (size_t)(t + 1) + k]); }
        plo[3 * (size_t)p + k] = lo;
        phi[3 * (size_t)p + k] = hi;
    }
    ids[p] = p;
}

// The binary tree, in two launches.  k_sah_top: ONE workgroup of 1024 threads walks the nodes of more than SAH_SUB primitives depth
// first -- the candidate passes of those nodes were what the build's time was made of; it ranks the candidates and scans boxes instead
// (below) -- and hands every child of <= SAH_SUB primitives to a list; k_sah_sub: one workgroup of 256 threads PER
// listed subtree builds it out of its own LDS, all of them at once.  Node records are numbered by a global counter (the BVH4
// emission follows the left / right links, not the numbers).  A node step reads LDS only -- every primitive's box (24 B), the order
// array, the stack of pending nodes; range, depth and box travel in the stack entry / in registers -- and leaves (one
// primitive) are written by their parent.  Scene build, before -> after (one 256-thread workgroup for everything, its working set in
// global memory): Cornell box 1.9 -> 1.25 ms, 1024 triangles 25.5 -> 5.6, 2047 triangles 46.7 -> 8.6 (16.0 while the top nodes still
// ran the quadratic pass with its boxes, 10.2 while the BVH4 emission read global memory: profiles/r04z_build_small.log,
// r04aa_sah_build_ms.log, r04ai_times.txt).
struct SahJob { uint32_t node, first, count, depth; };
constexpr int SAH_STACK = 160;  // pending nodes: <= 1 per level of a depth-first walk + 1; the tree is <= 24 + log2(2048) + 1 levels deep
constexpr uint32_t SAH_SUB = 128;  // subtrees of <= this many primitives are built by k_sah_sub

// NT threads build the subtree of `root` (a range of the order array) with the quadratic candidate pass.  Primitives are addressed by SLOT: the
// position the subtree's range had when it was loaded (cap = SAH_SUB), gid[slot] = primitive id, which ties between equal centroids are broken by.  (k_sah_top has its own loop for nodes above SAH_SUB.)
template <int NT>
__device__ __forceinline__ void sah_build(const SahJob root, uint32_t cap, uint32_t leaf_max, const float *g_box, const uint32_t *s_gid, uint32_t *s_ids,
                                          uint32_t *s_id, uint32_t *s_idg, float *s_box, uint32_t *s_pos, SahNode *__restrict__ nodes,
                                          uint32_t *__restrict__ n_nodes)
{
    __shared__ Best s_best[NT / 64];
    __shared__ double s_lo[NT / 64][3], s_hi[NT / 64][3];
    __shared__ uint32_t s_sp;
    __shared__ int s_axis;
    __shared__ uint32_t s_k;
    __shared__ int s_split;
    __shared__ SahJob s_todo[SAH_STACK];
    const int tid = threadIdx.x;
    if (tid == 0) { s_sp = 1; s_todo[0] = root; }
    const double INF = __builtin_inf();
    for (;;) {
        __syncthreads();  // the stack as thread 0 left it; everybody is done with the previous node's LDS
        if (s_sp == 0) break;
        const SahJob job = s_todo[s_sp - 1];
        __syncthreads();
        if (tid == 0) s_sp--;
        const uint32_t me = job.node, first = job.first, m = job.count, depth = job.depth;
        // the node's primitives side by side (slots, ids, boxes), and its box on the way
        double lo[3] = { INF, INF, INF }, hi[3] = { -INF, -INF, -INF };
        for (uint32_t i = tid; i < m; i += NT) {
            const uint32_t sl = s_ids[first - root.first + i];
            s_id[i] = sl;
            s_idg[i] = s_gid[sl];
            for (int k = 0; k < 3; k++) {
                const float a = g_box[6 * sl + k], b = g_box[6 * sl + 3 + k];
                s_box[6 * i + k] = a; s_box[6 * i + 3 + k] = b;
                lo[k] = fmin(lo[k], (double)a); hi[k] = fmax(hi[k], (double)b);
            }
        }
        for (int o = 32; o > 0; o >>= 1)
            for (int k = 0; k < 3; k++) { lo[k] = fmin(lo[k], __shfl_xor(lo[k], o, 64)); hi[k] = fmax(hi[k], __shfl_xor(hi[k], o, 64)); }
        if ((tid & 63) == 0)
            for (int k = 0; k < 3; k++) { s_lo[tid >> 6][k] = lo[k]; s_hi[tid >> 6][k] = hi[k]; }
        __syncthreads();
        for (int k = 0; k < 3; k++) { lo[k] = s_lo[0][k]; hi[k] = s_hi[0][k]; }
        for (int t = 1; t < NT / 64; t++)
            for (int k = 0; k < 3; k++) { lo[k] = fmin(lo[k], s_lo[t][k]); hi[k] = fmax(hi[k], s_hi[t][k]); }
        if (tid == 0)
            for (int k = 0; k < 3; k++) { nodes[me].lo[k] = lo[k]; nodes[me].hi[k] = hi[k]; }
        if (m <= 1) continue;  // (only a one-primitive scene's root comes here as a leaf)
        // every candidate (axis, primitive j): left = keys <= j's key
        Best best = { INF, 0, 0u };
        for (uint32_t c = tid; c < 3u * m; c += NT) {
            const int ax = (int)(c / m);
            const uint32_t jj = c - (uint32_t)ax * m;
            const uint32_t j = s_idg[jj];
            const double cj = (double)s_box[6 * jj + ax] + (double)s_box[6 * jj + 3 + ax];
            float llo[3] = { __builtin_inff(), __builtin_inff(), __builtin_inff() }, lhi[3] = { -__builtin_inff(), -__builtin_inff(), -__builtin_inff() };
            float rlo[3] = { __builtin_inff(), __builtin_inff(), __builtin_inff() }, rhi[3] = { -__builtin_inff(), -__builtin_inff(), -__builtin_inff() };
            uint32_t nl = 0;
            for (uint32_t i = 0; i < m; i++) {
                const uint32_t p = s_idg[i];
                const double cp = (double)s_box[6 * i + ax] + (double)s_box[6 * i + 3 + ax];
                const bool left = cp < cj || (cp == cj && p <= j);
                if (left) {
                    nl++;
                    for (int k = 0; k < 3; k++) { llo[k] = fminf(llo[k], s_box[6 * i + k]); lhi[k] = fmaxf(lhi[k], s_box[6 * i + 3 + k]); }
                } else {
                    for (int k = 0; k < 3; k++) { rlo[k] = fminf(rlo[k], s_box[6 * i + k]); rhi[k] = fmaxf(rhi[k], s_box[6 * i + 3 + k]); }
                }
            }
            s_pos[(size_t)ax * cap + jj] = nl - 1u;  // j's position in the sorted order along ax
            if (nl < m) {
                const double dl[3] = { llo[0], llo[1], llo[2] }, dh[3] = { lhi[0], lhi[1], lhi[2] };
                const double el[3] = { rlo[0], rlo[1], rlo[2] }, eh[3] = { rhi[0], rhi[1], rhi[2] };
                const Best b = { box_area_d(dl, dh) * (double)nl + box_area_d(el, eh) * (double)(m - nl), ax, nl - 1u };
                if (better(b, best)) best = b;
            }
        }
        for (int o = 32; o > 0; o >>= 1) {
            const Best other = { __shfl_xor(best.cost, o, 64), __shfl_xor(best.axis, o, 64), __shfl_xor(best.k, o, 64) };
            if (better(other, best)) best = other;
        }
        if ((tid & 63) == 0) s_best[tid >> 6] = best;
        __syncthreads();
        if (tid == 0) {
            for (int t = 1; t < NT / 64; t++)
                if (better(s_best[t], best)) best = s_best[t];
            const double a_node = box_area_d(lo, hi);
            // a node of <= leaf_max primitives stays a leaf when splitting does not pay (node step : primitive = 1 : 0.6)
            const bool leaf = m <= leaf_max && 0.6 * (double)m * a_node <= 1.0 * a_node + 0.6 * best.cost;
            s_split = leaf ? 0 : 1;
            s_axis = best.axis;
            s_k = depth >= (uint32_t)SAH_MAX_DEPTH ? m / 2u - 1u : best.k;
        }
        __syncthreads();
        if (!s_split) continue;
        const int ax = s_axis;
        for (uint32_t i = tid; i < m; i += NT) s_ids[first - root.first + s_pos[(size_t)ax * cap + i]] = s_id[i];
        __syncthreads();
        if (tid == 0) {
            const uint32_t l = atomicAdd(n_nodes, 2u), r = l + 1u;
            nodes[me].left = (int)l; nodes[me].right = (int)r;
            const uint32_t nl = s_k + 1u;
            auto child = [&](uint32_t c, uint32_t c_first, uint32_t c_count) {
                nodes[c].first = c_first; nodes[c].count = c_count; nodes[c].left = nodes[c].right = -1;
                if (c_count == 1u) {  // a leaf: its box is its primitive's
                    const uint32_t sl = s_ids[c_first - root.first];
                    for (int k = 0; k < 3; k++) { nodes[c].lo[k] = (double)g_box[6 * sl + k]; nodes[c].hi[k] = (double)g_box[6 * sl + 3 + k]; }
                } else {
                    s_todo[s_sp++] = { c, c_first, c_count, depth + 1u };
                }
            };
            child(r, first + nl, m - nl);  // left first
            child(l, first, nl);
        }
    }
}

constexpr int TBT = 1024;

// ---- the top of the tree: nodes of more than SAH_SUB primitives -------------------------------------------------------------
// Same candidates, same costs, same winner as sah_build, found differently: per axis the candidates are RANKED (a pass over the
// node per candidate that only compares keys -- the part of the old pass that is quadratic, now a few instructions per step), the
// boxes are put in that order and two scans over it -- suffix, then prefix -- give every split position its right and left box:
// min and max are exact, so the boxes, their binary64 areas and the costs are the old pass's bit for bit.  The node's primitives
// come from global memory each time (a top node is visited once; there are a few dozen of them).
__device__ __forceinline__ void box_join(float *a, const float *b)
{
    for (int k = 0; k < 3; k++) { a[k] = fminf(a[k], b[k]); a[3 + k] = fmaxf(a[3 + k], b[3 + k]); }
}

// in-place inclusive scan (FWD: prefix, else suffix) of S[m][6] under box_join; every thread of the block calls it
template <bool FWD>
__device__ __forceinline__ void scan_boxes(float *S, uint32_t m, float (*s_wt)[6])
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t C = (m + TBT - 1) / TBT;  // elements per thread (<= 2 for 2048 primitives)
    const float inf = __builtin_inff();
    float tot[6] = { inf, inf, inf, -inf, -inf, -inf };
    const uint32_t e0 = (uint32_t)tid * C, e1 = min(m, e0 + C);
    if (FWD) {
        for (uint32_t e = e0; e < e1; e++) { box_join(tot, S + 6 * e); for (int k = 0; k < 6; k++) S[6 * e + k] = tot[k]; }
    } else {
        for (uint32_t e = e1; e > e0; e--) { box_join(tot, S + 6 * (e - 1)); for (int k = 0; k < 6; k++) S[6 * (e - 1) + k] = tot[k]; }
    }
    // the threads' totals: inclusive scan inside the wave, then the waves' totals
    float inc[6];
    for (int k = 0; k < 6; k++) inc[k] = tot[k];
    for (int o = 1; o < 64; o <<= 1) {
        float t[6];
        for (int k = 0; k < 6; k++) t[k] = FWD ? __shfl_up(inc[k], o, 64) : __shfl_down(inc[k], o, 64);
        if (FWD ? lane >= o : lane + o < 64) box_join(inc, t);
    }
    float exc[6];  // what lies before (FWD) / behind this thread's elements inside its wave
    for (int k = 0; k < 6; k++) {
        const float t = FWD ? __shfl_up(inc[k], 1, 64) : __shfl_down(inc[k], 1, 64);
        exc[k] = (FWD ? lane == 0 : lane == 63) ? (k < 3 ? inf : -inf) : t;
    }
    if (FWD ? lane == 63 : lane == 0)
        for (int k = 0; k < 6; k++) s_wt[wave][k] = inc[k];
    __syncthreads();
    if (FWD) { for (int w = 0; w < wave; w++) box_join(exc, s_wt[w]); }
    else { for (int w = wave + 1; w < TBT / 64; w++) box_join(exc, s_wt[w]); }
    for (uint32_t e = e0; e < e1; e++) box_join(S + 6 * e, exc);
    __syncthreads();  // S is complete, s_wt free again
}

__global__ __launch_bounds__(TBT) void k_sah_top(uint32_t np, uint32_t leaf_max, const double *__restrict__ plo, const double *__restrict__ phi,
                                                 uint32_t *__restrict__ ids, SahNode *__restrict__ nodes, uint32_t *__restrict__ n_nodes,
                                                 SahJob *__restrict__ roots, uint32_t *__restrict__ n_roots)
{
    extern __shared__ __attribute__((aligned(16))) char sah_smem[];
    double *s_key = reinterpret_cast<double *>(sah_smem);                     // [m]: the axis' centroid keys, then the right-hand areas by split position
    float *s_box = reinterpret_cast<float *>(s_key + np);                     // [m][6]: the node's boxes ...
    float *S = s_box + 6 * (size_t)np;                                        // [m][6]: ... in the axis' sorted order, scanned in place
    uint32_t *s_ids = reinterpret_cast<uint32_t *>(S + 6 * (size_t)np);       // [np]: the order array (a node = a range of it)
    uint32_t *s_id = s_ids + np;                                              // [m]: the node's primitive ids
    uint32_t *s_pos = s_id + np;                                              // [3][np]: a candidate's position in its axis' sorted order
    __shared__ Best s_best[TBT / 64];
    __shared__ double s_lo[TBT / 64][3], s_hi[TBT / 64][3];
    __shared__ float s_wt[TBT / 64][6];
    __shared__ uint32_t s_sp;
    __shared__ int s_axis;
    __shared__ uint32_t s_k;
    __shared__ SahJob s_todo[SAH_STACK];
    const int tid = threadIdx.x;
    for (uint32_t i = tid; i < np; i += TBT) s_ids[i] = i;
    const SahJob root = { 0u, 0u, np, 0u };
    if (tid == 0) { nodes[0].first = 0; nodes[0].count = np; nodes[0].left = nodes[0].right = -1; s_sp = 1; s_todo[0] = root; }
    if (np <= SAH_SUB) {  // the whole scene is one subtree: k_sah_sub's work (uniform)
        if (tid == 0) { roots[0] = root; *n_roots = 1u; }
        return;
    }
    const double INF = __builtin_inf();
    for (;;) {
        __syncthreads();
        if (s_sp == 0) break;
        const SahJob job = s_todo[s_sp - 1];
        __syncthreads();
        if (tid == 0) s_sp--;
        const uint32_t me = job.node, first = job.first, m = job.count, depth = job.depth;  // m > SAH_SUB
        double lo[3] = { INF, INF, INF }, hi[3] = { -INF, -INF, -INF };
        for (uint32_t i = tid; i < m; i += TBT) {
            const uint32_t p = s_ids[first + i];
            s_id[i] = p;
            for (int k = 0; k < 3; k++) {
                const double a = plo[3 * (size_t)p + k], b = phi[3 * (size_t)p + k];
                s_box[6 * i + k] = (float)a; s_box[6 * i + 3 + k] = (float)b;  // (exact: made from floats)
                lo[k] = fmin(lo[k], a); hi[k] = fmax(hi[k], b);
            }
        }
        for (int o = 32; o > 0; o >>= 1)
            for (int k = 0; k < 3; k++) { lo[k] = fmin(lo[k], __shfl_xor(lo[k], o, 64)); hi[k] = fmax(hi[k], __shfl_xor(hi[k], o, 64)); }
        if ((tid & 63) == 0)
            for (int k = 0; k < 3; k++) { s_lo[tid >> 6][k] = lo[k]; s_hi[tid >> 6][k] = hi[k]; }
        __syncthreads();
        for (int k = 0; k < 3; k++) { lo[k] = s_lo[0][k]; hi[k] = s_hi[0][k]; }
        for (int t = 1; t < TBT / 64; t++)
            for (int k = 0; k < 3; k++) { lo[k] = fmin(lo[k], s_lo[t][k]); hi[k] = fmax(hi[k], s_hi[t][k]); }
        if (tid == 0)
            for (int k = 0; k < 3; k++) { nodes[me].lo[k] = lo[k]; nodes[me].hi[k] = hi[k]; }
        Best best = { INF, 0, 0u };
        for (int ax = 0; ax < 3; ax++) {
            for (uint32_t i = tid; i < m; i += TBT) s_key[i] = (double)s_box[6 * i + ax] + (double)s_box[6 * i + 3 + ax];
            __syncthreads();
            // rank: j's position in the (centroid, id) order = how many keys are <= j's, minus one
            for (uint32_t jj = tid; jj < m; jj += TBT) {
                const double cj = s_key[jj];
                const uint32_t j = s_id[jj];
                uint32_t nl = 0;
                for (uint32_t i = 0; i < m; i++) {
                    const double cp = s_key[i];
                    nl += (cp < cj || (cp == cj && s_id[i] <= j)) ? 1u : 0u;
                }
                s_pos[(size_t)ax * np + jj] = nl - 1u;
            }
            __syncthreads();
            // right-hand boxes: suffix scan of the sorted boxes; the area right of split position k (left = positions 0 .. k) is that of k + 1
            for (uint32_t i = tid; i < m; i += TBT) {
                const uint32_t q = s_pos[(size_t)ax * np + i];
                for (int k = 0; k < 6; k++) S[6 * q + k] = s_box[6 * i + k];
            }
            __syncthreads();
            scan_boxes<false>(S, m, s_wt);
            for (uint32_t k = tid; k + 1 < m; k += TBT) {
                const float *b = S + 6 * (k + 1);
                const double el[3] = { b[0], b[1], b[2] }, eh[3] = { b[3], b[4], b[5] };
                s_key[k] = box_area_d(el, eh);
            }
            __syncthreads();
            // left-hand boxes: prefix scan; cost of every split position
            for (uint32_t i = tid; i < m; i += TBT) {
                const uint32_t q = s_pos[(size_t)ax * np + i];
                for (int k = 0; k < 6; k++) S[6 * q + k] = s_box[6 * i + k];
            }
            __syncthreads();
            scan_boxes<true>(S, m, s_wt);
            for (uint32_t k = tid; k + 1 < m; k += TBT) {
                const float *b = S + 6 * k;
                const double dl[3] = { b[0], b[1], b[2] }, dh[3] = { b[3], b[4], b[5] };
                const uint32_t nl = k + 1u;
                const Best c = { box_area_d(dl, dh) * (double)nl + s_key[k] * (double)(m - nl), ax, k };
                if (better(c, best)) best = c;
            }
            __syncthreads();  // s_key, S are rewritten for the next axis
        }
        for (int o = 32; o > 0; o >>= 1) {
            const Best other = { __shfl_xor(best.cost, o, 64), __shfl_xor(best.axis, o, 64), __shfl_xor(best.k, o, 64) };
            if (better(other, best)) best = other;
        }
        if ((tid & 63) == 0) s_best[tid >> 6] = best;
        __syncthreads();
        if (tid == 0) {
            for (int t = 1; t < TBT / 64; t++)
                if (better(s_best[t], best)) best = s_best[t];
            // (a node of m > SAH_SUB >= leaf_max primitives is always split)
            s_axis = best.axis;
            s_k = depth >= (uint32_t)SAH_MAX_DEPTH ? m / 2u - 1u : best.k;
        }
        __syncthreads();
        const int ax = s_axis;
        for (uint32_t i = tid; i < m; i += TBT) s_ids[first + s_pos[(size_t)ax * np + i]] = s_id[i];
        __syncthreads();
        if (tid == 0) {
            const uint32_t l = atomicAdd(n_nodes, 2u), r = l + 1u;
            nodes[me].left = (int)l; nodes[me].right = (int)r;
            const uint32_t nl = s_k + 1u;
            auto child = [&](uint32_t c, uint32_t c_first, uint32_t c_count) {
                nodes[c].first = c_first; nodes[c].count = c_count; nodes[c].left = nodes[c].right = -1;
                if (c_count == 1u) {  // a leaf: its box is its primitive's
                    const uint32_t p = s_ids[c_first];
                    for (int k = 0; k < 3; k++) { nodes[c].lo[k] = plo[3 * (size_t)p + k]; nodes[c].hi[k] = phi[3 * (size_t)p + k]; }
                } else if (c_count <= SAH_SUB) {
                    roots[atomicAdd(n_roots, 1u)] = { c, c_first, c_count, depth + 1u };
                } else {
                    s_todo[s_sp++] = { c, c_first, c_count, depth + 1u };
                }
            };
            child(r, first + nl, m - nl);  // left first
            child(l, first, nl);
        }
    }
    for (uint32_t i = tid; i < np; i += TBT) ids[i] = s_ids[i];  // (the subtree kernels read their ranges from here)
}

__global__ __launch_bounds__(TBD) void k_sah_sub(uint32_t leaf_max, const double *__restrict__ plo, const double *__restrict__ phi,
                                                 uint32_t *__restrict__ ids, SahNode *__restrict__ nodes, uint32_t *__restrict__ n_nodes,
                                                 const SahJob *__restrict__ roots, const uint32_t *__restrict__ n_roots)
{
    if (blockIdx.x >= *n_roots) return;
    __shared__ float g_box[6 * SAH_SUB], s_box[6 * SAH_SUB];
    __shared__ uint32_t s_gid[SAH_SUB], s_ids[SAH_SUB], s_id[SAH_SUB], s_idg[SAH_SUB], s_pos[3 * SAH_SUB];
    const SahJob root = roots[blockIdx.x];
    for (uint32_t i = threadIdx.x; i < root.count; i += TBD) {
        const uint32_t p = ids[root.first + i];
        for (int k = 0; k < 3; k++) { g_box[6 * i + k] = (float)plo[3 * (size_t)p + k]; g_box[6 * i + 3 + k] = (float)phi[3 * (size_t)p + k]; }
        s_gid[i] = p;
        s_ids[i] = i;
    }
    sah_build<TBD>(root, SAH_SUB, leaf_max, g_box, s_gid, s_ids, s_id, s_idg, s_box, s_pos, nodes, n_nodes);
    for (uint32_t i = threadIdx.x; i < root.count; i += TBD) ids[root.first + i] = s_gid[s_ids[i]];
}

// per binary node, in parallel: the padded float box of its triangles (what a BVH4 slot holds) and its area
__global__ __launch_bounds__(TBD) void k_sah_node_boxes(const SahNode *__restrict__ nodes, const uint32_t *__restrict__ n_nodes,
                                                        const uint32_t *__restrict__ ids, const uint32_t *__restrict__ prim_first,
                                                        const uint8_t *__restrict__ prim_tris, const float *__restrict__ tlo,
                                                        const float *__restrict__ thi, float pad, float *__restrict__ nbox /* [6] */,
                                                        double *__restrict__ narea, uint32_t *__restrict__ ntris)
{
    const uint32_t i = blockIdx.x * TBD + threadIdx.x;
    if (i >= *n_nodes) return;
    const SahNode &k = nodes[i];
    const float inf = __builtin_inff();
    float lo[3] = { inf, inf, inf }, hi[3] = { -inf, -inf, -inf };
    uint32_t n_tri = 0;
    for (uint32_t t = 0; t < k.count; t++) {
        const uint32_t prim = ids[k.first + t];
        for (uint32_t h = 0; h < prim_tris[prim]; h++, n_tri++) {
            const uint32_t tri = prim_first[prim] + h;
            for (int c = 0; c < 3; c++) {
                lo[c] = fminf(lo[c], tlo[3 * (size_t)tri + c] - pad);  // float, like k_refit
                hi[c] = fmaxf(hi[c], thi[3 * (size_t)tri + c] + pad);
            }
        }
    }
    for (int c = 0; c < 3; c++) { nbox[6 * (size_t)i + c] = lo[c]; nbox[6 * (size_t)i + 3 + c] = hi[c]; }
    narea[i] = box_area_d(k.lo, k.hi);
    ntris[i] = n_tri;
}

// BVH4 rows + leaf order from the binary tree.  Which binary nodes become BVH4 nodes is a CUT of the binary tree, and the cut is chosen to
// minimise what a walk pays for: the sum of the areas of the BVH4's internal nodes (the surface-area estimate of node visits per ray; the
// leaves are the binary tree's and cost the same under every cut).  Dynamic programming over the binary tree (Ylitie, Karras, Laine 2017,
// section 3): F(n, k) = the least area sum that covers subtree n with at most k slots of its parent = min(area(n) + S(n, 4) -- n becomes a
// node --, S(n, k) -- n is opened --) with S(n, k) = min over j of F(left, j) + F(right, k - j); a leaf costs 0.  Until round 6 the cut was
// greedy (open the internal child of largest area until the row is full), which left the Cornell box with a child of the root that spans
// the whole room -- three walls -- and is visited by every ray: 8 nodes, 3.46 expected node visits per ray (3.41 measured) against the
// optimum's 7 nodes and 2.97 (2.87 in a simulation of the walk over path-traced rays, leaf visits 1.39 -> 1.40).  One thread does both passes;
// everything it BRANCHES on -- links, ranges, areas, triangle counts, the primitives, the tables -- is staged in LDS by the whole
// workgroup first (<= 143 KB for 2048 primitives), and all it writes per row is which binary node sits in which slot and the slot's child
// word: k_sah_rows then fills the 128-B rows (boxes, words, padding) in parallel.
constexpr uint32_t ROW_EMPTY = 0xFFFFFFFFu;
__global__ __launch_bounds__(TBD) void k_sah_emit(const SahNode *__restrict__ nodes, const uint32_t *__restrict__ n_nodes, const uint32_t *__restrict__ ids,
                                                  const uint32_t *__restrict__ prim_first, const uint8_t *__restrict__ prim_tris, uint32_t np,
                                                  const double *__restrict__ narea, const uint32_t *__restrict__ ntris, uint32_t *__restrict__ row_kid,
                                                  uint32_t *__restrict__ row_word, uint32_t *__restrict__ order, uint32_t *__restrict__ counts /* {n_rows, n_order} */)
{
    extern __shared__ __attribute__((aligned(16))) char emit_smem[];
    const uint32_t nn = *n_nodes;
    double *s_area = reinterpret_cast<double *>(emit_smem);                 // [nn]
    uint32_t *s_lr = reinterpret_cast<uint32_t *>(s_area + nn);             // [nn]: left | right << 16 (0xFFFF: leaf)
    uint32_t *s_fc = s_lr + nn;                                             // [nn]: first | count << 16
    uint32_t *s_nt = s_fc + nn;                                             // [nn]: triangles below the node
    uint32_t *s_prim = s_nt + nn;                                           // [np]: first triangle | triangles << 16 of the primitive at that place of the order
    float *s_f = reinterpret_cast<float *>(s_prim + np);                    // [nn][3]: F(n, 1 .. 3)
    uint8_t *s_ch = reinterpret_cast<uint8_t *>(s_f + 3 * (size_t)nn);      // [nn]: bit 0 F(n,2) opens n, bit 1 F(n,3) opens n, bit 2 j of S(n,3) - 1, bits 3-4 j of S(n,4) - 1
    __shared__ uint2 s_todo[SAH_STACK * 3];                                 // {binary node, row}: <= 3 pushed per level
    for (uint32_t i = threadIdx.x; i < nn; i += TBD) {
        const SahNode &k = nodes[i];
        s_area[i] = narea[i];
        s_lr[i] = k.left < 0 ? 0xFFFFu : ((uint32_t)k.left | ((uint32_t)k.right << 16));
        s_fc[i] = k.first | (k.count << 16);
        s_nt[i] = ntris[i];
    }
    for (uint32_t i = threadIdx.x; i < np; i += TBD) {
        const uint32_t prim = ids[i];
        s_prim[i] = prim_first[prim] | ((uint32_t)prim_tris[prim] << 16);
    }
    __syncthreads();
    if (threadIdx.x != 0) return;
    uint32_t n_rows = 0, n_order = 0, sp = 0;
    auto is_leaf = [&](uint32_t n) { return (s_lr[n] & 0xFFFFu) == 0xFFFFu; };
    // ---- pass 1: the tables, children before parents (an explicit post-order walk; s_todo.y: 0 = first visit, 1 = children done)
    s_todo[sp++] = make_uint2(0u, 0u);
    while (sp > 0) {
        const uint2 it = s_todo[sp - 1];
        const uint32_t n = it.x;
        if (is_leaf(n)) {
            s_f[3 * n + 0] = s_f[3 * n + 1] = s_f[3 * n + 2] = 0.f;
            s_ch[n] = 0u;
            sp--;
            continue;
        }
        const uint32_t l = s_lr[n] & 0xFFFFu, r = s_lr[n] >> 16;
        if (it.y == 0u) {
            s_todo[sp - 1].y = 1u;
            s_todo[sp++] = make_uint2(l, 0u);
            s_todo[sp++] = make_uint2(r, 0u);
            continue;
        }
        sp--;
        auto F = [&](uint32_t m, uint32_t k) { return s_f[3 * m + (k - 1u)]; };   // k = 1 .. 3
        // S(n, k): the best split of k slots between the two children (first minimum in j)
        const float s2 = F(l, 1) + F(r, 1);
        float s3 = F(l, 1) + F(r, 2);
        uint32_t j3 = 1u;
        if (F(l, 2) + F(r, 1) < s3) { s3 = F(l, 2) + F(r, 1); j3 = 2u; }
        float s4 = F(l, 1) + F(r, 3);
        uint32_t j4 = 1u;
        if (F(l, 2) + F(r, 2) < s4) { s4 = F(l, 2) + F(r, 2); j4 = 2u; }
        if (F(l, 3) + F(r, 1) < s4) { s4 = F(l, 3) + F(r, 1); j4 = 3u; }
        const float as_node = (float)s_area[n] + s4;
        s_f[3 * n + 0] = as_node;
        // (equal cost: open the node -- the same estimate with a row less)
        const bool open2 = s2 <= as_node, open3 = s3 <= as_node;
        s_f[3 * n + 1] = open2 ? s2 : as_node;
        s_f[3 * n + 2] = open3 ? s3 : as_node;
        s_ch[n] = (uint8_t)((open2 ? 1u : 0u) | (open3 ? 2u : 0u) | ((j3 - 1u) << 2) | ((j4 - 1u) << 3));
    }
    // ---- pass 2: the rows.  A row's slots = the cut below its binary node: S(n, 4) at the top, then the tables' choices, left before right
    sp = 0;
    s_todo[sp++] = make_uint2(0u, n_rows++);
    while (sp > 0) {
        const uint2 it = s_todo[--sp];
        uint32_t kids[4];
        int m = 0;
        if (is_leaf(it.x)) {
            kids[m++] = it.x;  // the whole scene is one leaf
        } else {
            uint2 ex[8];       // {binary node, slots it may take}; popped left first
            int xs = 0;
            {
                const uint32_t j4 = ((s_ch[it.x] >> 3) & 3u) + 1u;
                ex[xs++] = make_uint2(s_lr[it.x] >> 16, 4u - j4);
                ex[xs++] = make_uint2(s_lr[it.x] & 0xFFFFu, j4);
            }
            while (xs > 0) {
                const uint2 e = ex[--xs];
                const uint32_t n = e.x, k = e.y;
                const bool open = !is_leaf(n) && k >= 2u && ((s_ch[n] >> (k - 2u)) & 1u);
                if (!open) { kids[m++] = n; continue; }
                const uint32_t j = k == 2u ? 1u : ((s_ch[n] >> 2) & 1u) + 1u;   // (k = 3; k = 4 only at the top)
                ex[xs++] = make_uint2(s_lr[n] >> 16, k - j);
                ex[xs++] = make_uint2(s_lr[n] & 0xFFFFu, j);
            }
        }
        for (int j = 0; j < 4; j++) {
            uint32_t kid = ROW_EMPTY, word = ROW_EMPTY;
            if (j < m) {
                kid = kids[j];
                if (is_leaf(kid)) {
                    word = PT_LEAF | ((s_nt[kid] - 1u) << 28) | n_order;
                    const uint32_t first = s_fc[kid] & 0xFFFFu, count = s_fc[kid] >> 16;
                    for (uint32_t t = 0; t < count; t++) {
                        const uint32_t pr = s_prim[first + t];
                        for (uint32_t h = 0; h < (pr >> 16); h++) order[n_order++] = (pr & 0xFFFFu) + h;
                    }
                } else {
                    word = n_rows++;
                    s_todo[sp++] = make_uint2(kid, word);
                }
            }
            row_kid[4 * (size_t)it.y + j] = kid;
            row_word[4 * (size_t)it.y + j] = word;
        }
    }
    counts[0] = n_rows;
    counts[1] = n_order;
}

// one thread per (row, slot): the slot's padded box (+inf where empty), its child word, the row's padding
__global__ __launch_bounds__(TBD) void k_sah_rows(const uint32_t *__restrict__ counts, const uint32_t *__restrict__ row_kid, const uint32_t *__restrict__ row_word,
                                                  const float *__restrict__ nbox, uint32_t *__restrict__ rows)
{
    const uint32_t i = blockIdx.x * TBD + threadIdx.x;
    if (i >= 4u * counts[0]) return;
    const uint32_t r = i >> 2, j = i & 3u, kid = row_kid[i];
    float *f = reinterpret_cast<float *>(rows + 32 * (size_t)r);
    const float inf = __builtin_inff();
    for (int c = 0; c < 3; c++) {
        f[4 * c + j] = kid == ROW_EMPTY ? inf : nbox[6 * (size_t)kid + c];
        f[12 + 4 * c + j] = kid == ROW_EMPTY ? inf : nbox[6 * (size_t)kid + 3 + c];  // (an empty slot: lo = hi = +inf)
    }
    rows[32 * (size_t)r + 24 + j] = row_word[i];
    rows[32 * (size_t)r + 28 + j] = 0u;
}

template <typename T>
struct Buf {
    T *p = nullptr;
    ~Buf() { if (p) (void)hipFree(p); }
    hipError_t alloc(size_t n) { return hipMalloc((void **)&p, sizeof(T) * (n ? n : 1)); }
};

}  // namespace

// tlo / thi: HOST arrays of the unpadded triangle boxes (3 floats each), as pt_sah_build_bvh4 takes them.  The rows
// (32 dwords per BVH4 node) and the leaf order are produced on the device and returned to the host vectors as well (the
// caller keeps the device copies it uploads from them; a scene of <= 2048 triangles is a few hundred kilobytes).
pt_status pt_sah_build_bvh4_device(pt_ctx *ctx, const float *tlo, const float *thi, uint32_t n, const uint8_t *pair_with_next, float pad,
                                   uint32_t leaf_max, std::vector<uint32_t> &rows, std::vector<uint32_t> &order)
{
    hipStream_t st = ctx->stream;
    std::vector<uint32_t> prim_first;
    std::vector<uint8_t> prim_tris;
    for (uint32_t t = 0; t < n;) {
        const uint32_t cnt = (pair_with_next && t + 1 < n && pair_with_next[t]) ? 2u : 1u;
        prim_first.push_back(t);
        prim_tris.push_back((uint8_t)cnt);
        t += cnt;
    }
    const uint32_t np = (uint32_t)prim_first.size();
    Buf<float> d_tlo, d_thi;
    Buf<uint32_t> d_first, d_ids, d_nn, d_rows, d_order, d_counts;
    Buf<uint8_t> d_tris;
    Buf<double> d_plo, d_phi;
    Buf<SahNode> d_nodes;
    PT_HIP(ctx, d_tlo.alloc(3 * (size_t)n)); PT_HIP(ctx, d_thi.alloc(3 * (size_t)n));
    PT_HIP(ctx, d_first.alloc(np)); PT_HIP(ctx, d_tris.alloc(np)); PT_HIP(ctx, d_ids.alloc(np));
    PT_HIP(ctx, d_nn.alloc(1)); PT_HIP(ctx, d_plo.alloc(3 * (size_t)np)); PT_HIP(ctx, d_phi.alloc(3 * (size_t)np));
    PT_HIP(ctx, d_nodes.alloc(2 * (size_t)np + 1));
    PT_HIP(ctx, d_rows.alloc(32 * (size_t)(2 * np + 1))); PT_HIP(ctx, d_order.alloc(n)); PT_HIP(ctx, d_counts.alloc(2));
    PT_HIP(ctx, hipMemcpyAsync(d_tlo.p, tlo, sizeof(float) * 3 * (size_t)n, hipMemcpyHostToDevice, st));
    PT_HIP(ctx, hipMemcpyAsync(d_thi.p, thi, sizeof(float) * 3 * (size_t)n, hipMemcpyHostToDevice, st));
    PT_HIP(ctx, hipMemcpyAsync(d_first.p, prim_first.data(), sizeof(uint32_t) * np, hipMemcpyHostToDevice, st));
    PT_HIP(ctx, hipMemcpyAsync(d_tris.p, prim_tris.data(), np, hipMemcpyHostToDevice, st));
    k_sah_prims<<<(np + TBD - 1) / TBD, TBD, 0, st>>>(d_tlo.p, d_thi.p, d_first.p, d_tris.p, np, d_plo.p, d_phi.p, d_ids.p);
    // (d_nn: the node counter, one node -- the root -- taken; d_nroots: subtrees handed to k_sah_sub)
    Buf<SahJob> d_roots;
    Buf<uint32_t> d_nroots;
    PT_HIP(ctx, d_roots.alloc(np + 1)); PT_HIP(ctx, d_nroots.alloc(1));
    const uint32_t one = 1u;
    PT_HIP(ctx, hipMemcpyAsync(d_nn.p, &one, sizeof(one), hipMemcpyHostToDevice, st));
    PT_HIP(ctx, hipMemsetAsync(d_nroots.p, 0, sizeof(uint32_t), st));
    const size_t top_smem = sizeof(uint32_t) * 19 * (size_t)np;  // keys / areas 2 + the node's boxes 6 + their sorted copy 6 + order 1 + ids 1 + positions 3: <= 152 KB for 2048 primitives
    if (top_smem > 48 * 1024)
        PT_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(k_sah_top), hipFuncAttributeMaxDynamicSharedMemorySize, (int)top_smem));
    k_sah_top<<<1, TBT, top_smem, st>>>(np, leaf_max > 0 ? leaf_max : 1u, d_plo.p, d_phi.p, d_ids.p, d_nodes.p, d_nn.p, d_roots.p, d_nroots.p);
    // (a child handed over has >= 2 primitives and the ranges are disjoint: <= np / 2 subtrees)
    k_sah_sub<<<std::max(1u, np / 2u), TBD, 0, st>>>(leaf_max > 0 ? leaf_max : 1u, d_plo.p, d_phi.p, d_ids.p, d_nodes.p, d_nn.p, d_roots.p, d_nroots.p);
    Buf<float> d_nbox;
    Buf<double> d_narea;
    Buf<uint32_t> d_ntris;
    PT_HIP(ctx, d_nbox.alloc(6 * (2 * (size_t)np + 1))); PT_HIP(ctx, d_narea.alloc(2 * (size_t)np + 1)); PT_HIP(ctx, d_ntris.alloc(2 * (size_t)np + 1));
    k_sah_node_boxes<<<(2 * np + 1 + TBD - 1) / TBD, TBD, 0, st>>>(d_nodes.p, d_nn.p, d_ids.p, d_first.p, d_tris.p, d_tlo.p, d_thi.p, pad, d_nbox.p,
                                                                   d_narea.p, d_ntris.p);
    Buf<uint32_t> d_row_kid, d_row_word;
    PT_HIP(ctx, d_row_kid.alloc(4 * (2 * (size_t)np + 1))); PT_HIP(ctx, d_row_word.alloc(4 * (2 * (size_t)np + 1)));
    const size_t emit_smem = (sizeof(double) + 3 * sizeof(uint32_t) + 3 * sizeof(float) + 1) * (2 * (size_t)np + 1) + sizeof(uint32_t) * (size_t)np + 16;  // <= 143 KB for 2048 primitives
    if (emit_smem > 48 * 1024)
        PT_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(k_sah_emit), hipFuncAttributeMaxDynamicSharedMemorySize, (int)emit_smem));
    k_sah_emit<<<1, TBD, emit_smem, st>>>(d_nodes.p, d_nn.p, d_ids.p, d_first.p, d_tris.p, np, d_narea.p, d_ntris.p, d_row_kid.p, d_row_word.p, d_order.p, d_counts.p);
    k_sah_rows<<<(4 * (2 * np + 1) + TBD - 1) / TBD, TBD, 0, st>>>(d_counts.p, d_row_kid.p, d_row_word.p, d_nbox.p, d_rows.p);
    uint32_t counts[2] = { 0, 0 };
    PT_HIP(ctx, hipMemcpyAsync(counts, d_counts.p, sizeof(counts), hipMemcpyDeviceToHost, st));
    PT_HIP(ctx, hipStreamSynchronize(st));
    PT_HIP(ctx, hipGetLastError());
    if (counts[1] != n || counts[0] == 0 || counts[0] > 2 * np + 1) { ctx->err = "internal: device SAH build lost triangles"; return PT_ERR_HIP; }
    rows.resize(32 * (size_t)counts[0]);
    order.resize(n);
    PT_HIP(ctx, hipMemcpy(rows.data(), d_rows.p, sizeof(uint32_t) * rows.size(), hipMemcpyDeviceToHost));
    PT_HIP(ctx, hipMemcpy(order.data(), d_order.p, sizeof(uint32_t) * n, hipMemcpyDeviceToHost));
    return PT_OK;
}

// Most entries a depth-first walk of a BVH4 (rows of 32 dwords, child words at 24..27) can have pending: what the LDS-only
// traversal kernels size their stacks by (ptw_plan_extend, extend_launch.hip).
uint32_t pt_wide_stack_need(const std::vector<uint32_t> &w)
{
    // a node with k children pushes at most k-1 of them before descending:
    // need(node) = k-1 + max over internal children (iterative, children always have larger indices or not -- use DFS)
    struct F { uint32_t node; uint32_t depth; };
    const size_t n = w.size() / 32;
    std::vector<uint32_t> need(n, 0);
    std::vector<int> state(n, 0);
    std::vector<uint32_t> stack{ 0u };
    while (!stack.empty()) {
        const uint32_t nd = stack.back();
        if (nd >= n || stack.size() > 4096) return 1u << 20;  // malformed: forces the spilling variant
        if (state[nd] == 0) {
            state[nd] = 1;
            for (int c = 0; c < 4; c++) {
                const uint32_t word = w[32 * (size_t)nd + 24 + c];
                if (word != 0xFFFFFFFFu && !(word & PT_LEAF)) stack.push_back(word);
            }
        } else {
            stack.pop_back();
            uint32_t k = 0, deepest = 0;
            for (int c = 0; c < 4; c++) {
                const uint32_t word = w[32 * (size_t)nd + 24 + c];
                if (word == 0xFFFFFFFFu) continue;
                k++;
                if (!(word & PT_LEAF) && word < n) deepest = std::max(deepest, need[word]);
            }
            need[nd] = (k ? k - 1 : 0) + deepest;
        }
    }
    return need[0];
}
