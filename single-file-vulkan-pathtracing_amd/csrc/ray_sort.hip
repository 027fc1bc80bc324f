// ray_sort.hip -- per-round ordering of the extend queue by (origin cell, direction octant) for scenes whose traversal
// working set exceeds the 256 MiB Infinity Cache (north star: "ray-sorted for coalesced HBM reads of triangle / BVH-node
// data"; the reference leaves ray coherence to the driver behind traceRayEXT, raygen.rgen:63-75).
//
// Evidence (scripts/probe_ray_sort.py, 4 M incoherent rays, extend kernel only, MI355X): a free perfect sort by
// 6 bits/axis + octant buys +36 % on the 8 M-triangle soup (3.02 -> 2.22 ms) and +13 % on the 1 M-triangle one, whose
// 118 MB stay in the Infinity Cache -- so AUTO sorts only past that size, and PT_FLAG_SORT_RAYS / PT_FLAG_NO_SORT_RAYS
// override it.
//
// What is sorted is a PERMUTATION, not the queue: key[i] = Morton(cell of origin i, SORT_BITS per axis) << 3 | octant,
// an LSD radix sort of (key, i) pairs in 8-bit passes (hist -> scan -> scatter; the live count is read on the device,
// the host never waits), and k_extend<hbm> then takes its rays as rayA[perm[j]] and writes hit[perm[j]]: the shade
// kernel and the queue layout do not change, and the hit records cannot depend on the order.
#include "pt_internal.h"

#include "device_scan.h"

#include <algorithm>
#include <cmath>

namespace {

constexpr int TBS = 256;
constexpr int RS32_KPT = 8;
constexpr int RS32_TILE = TBS * RS32_KPT;

__device__ __forceinline__ uint32_t spread3(uint32_t x)  // 10 bits -> every third bit
{
    x &= 0x3FFu;
    x = (x | (x << 16)) & 0x030000FFu;
    x = (x | (x << 8)) & 0x0300F00Fu;
    x = (x | (x << 4)) & 0x030C30C3u;
    x = (x | (x << 2)) & 0x09249249u;
    return x;
}

__global__ __launch_bounds__(TBS) void k_ray_keys(const float4 *__restrict__ rayA, const float2 *__restrict__ rayB,
                                                  const uint32_t *__restrict__ count, float lox, float loy, float loz, float sx,
                                                  float sy, float sz, uint32_t cells, uint32_t *__restrict__ keys,
                                                  uint32_t *__restrict__ vals)
{
    const uint32_t n = *count;
    for (uint32_t i = blockIdx.x * TBS + threadIdx.x; i < n; i += gridDim.x * TBS) {
        const float4 a = rayA[i];
        const float2 b = rayB[i];
        const float fx = fminf(fmaxf((a.x - lox) * sx, 0.f), (float)(cells - 1u));
        const float fy = fminf(fmaxf((a.y - loy) * sy, 0.f), (float)(cells - 1u));
        const float fz = fminf(fmaxf((a.z - loz) * sz, 0.f), (float)(cells - 1u));
        const uint32_t m = (spread3((uint32_t)fx) << 2) | (spread3((uint32_t)fy) << 1) | spread3((uint32_t)fz);
        const uint32_t oct = (a.w < 0.f ? 4u : 0u) | (b.x < 0.f ? 2u : 0u) | (b.y < 0.f ? 1u : 0u);
        keys[i] = (m << 3) | oct;
        vals[i] = i;
    }
}

__global__ __launch_bounds__(TBS) void k_rs32_hist(const uint32_t *__restrict__ keys, const uint32_t *__restrict__ count, int shift,
                                                   uint32_t *__restrict__ hist, uint32_t nblocks)
{
    __shared__ uint32_t h[256];
    h[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t n = *count;
    const uint32_t base = blockIdx.x * RS32_TILE;
    if (base < n)
        for (int k = 0; k < RS32_KPT; k++) {
            const uint32_t i = base + k * TBS + threadIdx.x;
            if (i < n) atomicAdd(&h[(keys[i] >> shift) & 255u], 1u);
        }
    __syncthreads();
    hist[(size_t)threadIdx.x * nblocks + blockIdx.x] = h[threadIdx.x];
}

// (not stable inside a tile: the order of equal keys does not matter here)
__global__ __launch_bounds__(TBS) void k_rs32_scatter(const uint32_t *__restrict__ kin, const uint32_t *__restrict__ vin,
                                                      uint32_t *__restrict__ kout, uint32_t *__restrict__ vout,
                                                      const uint32_t *__restrict__ count, int shift,
                                                      const uint32_t *__restrict__ offs, uint32_t nblocks)
{
    __shared__ uint32_t cur[256];
    cur[threadIdx.x] = offs[(size_t)threadIdx.x * nblocks + blockIdx.x];
    __syncthreads();
    const uint32_t n = *count;
    const uint32_t base = blockIdx.x * RS32_TILE;
    if (base >= n) return;
    for (int k = 0; k < RS32_KPT; k++) {
        const uint32_t i = base + k * TBS + threadIdx.x;
        if (i < n) {
            const uint32_t key = kin[i];
            const uint32_t dst = atomicAdd(&cur[(key >> shift) & 255u], 1u);
            kout[dst] = key;
            vout[dst] = vin[i];
        }
    }
}

}  // namespace

size_t ptw_ray_sort_bytes(size_t cap)
{
    const size_t nblocks = (cap + RS32_TILE - 1) / RS32_TILE;
    return sizeof(uint32_t) * (4 * cap + 256 * nblocks + 256 * nblocks / SC_TILE + 2);
}

// Sorts the first *count rays of (rayA, rayB); returns the permutation (device pointer inside `scratch`).
// scratch: ptw_ray_sort_bytes(cap) bytes; cap >= *count.
const uint32_t *ptw_sort_rays(hipStream_t st, const float4 *rayA, const float2 *rayB, const uint32_t *count, size_t cap,
                              const float *bmin, const float *bmax, int bits, int num_cus, void *scratch)
{
    const uint32_t nblocks = (uint32_t)((cap + RS32_TILE - 1) / RS32_TILE);
    uint32_t *keys[2], *vals[2];
    uint32_t *p = static_cast<uint32_t *>(scratch);
    keys[0] = p; keys[1] = p + cap; vals[0] = p + 2 * cap; vals[1] = p + 3 * cap;
    uint32_t *hist = p + 4 * cap, *sums = hist + 256 * (size_t)nblocks;
    const uint32_t cells = 1u << bits;
    float s[3];
    for (int k = 0; k < 3; k++) s[k] = (float)cells / fmaxf(bmax[k] - bmin[k], 1e-30f);
    const int grid = (int)std::min<size_t>((cap + TBS - 1) / TBS, (size_t)num_cus * 16);
    k_ray_keys<<<grid, TBS, 0, st>>>(rayA, rayB, count, bmin[0], bmin[1], bmin[2], s[0], s[1], s[2], cells, keys[0], vals[0]);
    const int key_bits = 3 * bits + 3;
    int cur = 0;
    for (int shift = 0; shift < key_bits; shift += 8) {
        k_rs32_hist<<<nblocks, TBS, 0, st>>>(keys[cur], count, shift, hist, nblocks);
        exclusive_scan(hist, 256u * nblocks, sums, st);
        k_rs32_scatter<<<nblocks, TBS, 0, st>>>(keys[cur], vals[cur], keys[cur ^ 1], vals[cur ^ 1], count, shift, hist, nblocks);
        cur ^= 1;
    }
    return vals[cur];
}
