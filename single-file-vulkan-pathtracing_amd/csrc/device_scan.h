// device_scan.h -- in-place exclusive scan of 32-bit counters over many blocks (radix-sort histograms, wide-node
// numbering, ray-sort histograms).  Included by the translation units that need it; everything is in an anonymous
// namespace, so each has its own copy.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace {

constexpr int SCAN_TB = 256;
// In-place exclusive scan of `total` counters over many blocks: per-tile sums -> scan of the sums
// (one block) -> per-tile scan + offset.  A tile is 2048 counters (256 threads x 8).
constexpr int SC_PER = 8;
constexpr int SC_TILE = SCAN_TB * SC_PER;

__device__ __forceinline__ uint32_t block_exclusive_scan(uint32_t v, uint32_t *s_wave /*[4]*/, uint32_t &block_total)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t inc = v;
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t t = __shfl_up(inc, o, 64);
        if (lane >= o) inc += t;
    }
    if (lane == 63) s_wave[wave] = inc;
    __syncthreads();
    uint32_t off = 0, tot = 0;
    for (int w = 0; w < 4; w++) {
        const uint32_t t = s_wave[w];
        if (w < wave) off += t;
        tot += t;
    }
    block_total = tot;
    __syncthreads();
    return off + inc - v;
}

__global__ __launch_bounds__(SCAN_TB) void k_scan_sums(const uint32_t *__restrict__ data, uint32_t total,
                                                  uint32_t *__restrict__ sums)
{
    __shared__ uint32_t s_wave[4];
    const uint32_t base = blockIdx.x * SC_TILE + threadIdx.x * SC_PER;
    uint32_t v = 0;
    for (int k = 0; k < SC_PER; k++)
        if (base + k < total) v += data[base + k];
    uint32_t tot;
    (void)block_exclusive_scan(v, s_wave, tot);
    if (threadIdx.x == 0) sums[blockIdx.x] = tot;
}

// exclusive scan of the tile sums in place (one block; a few thousand entries at most)
__global__ __launch_bounds__(SCAN_TB) void k_scan_top(uint32_t *__restrict__ sums, uint32_t n)
{
    __shared__ uint32_t s_wave[4];
    uint32_t carry = 0;
    for (uint32_t base = 0; base < n; base += SCAN_TB) {
        const uint32_t i = base + threadIdx.x;
        const uint32_t v = i < n ? sums[i] : 0u;
        uint32_t tot;
        const uint32_t ex = block_exclusive_scan(v, s_wave, tot);
        if (i < n) sums[i] = carry + ex;
        carry += tot;
    }
}

__global__ __launch_bounds__(SCAN_TB) void k_scan_apply(uint32_t *__restrict__ data, uint32_t total,
                                                   const uint32_t *__restrict__ sums)
{
    __shared__ uint32_t s_wave[4];
    const uint32_t base = blockIdx.x * SC_TILE + threadIdx.x * SC_PER;
    uint32_t x[SC_PER], v = 0;
    for (int k = 0; k < SC_PER; k++) {
        x[k] = base + k < total ? data[base + k] : 0u;
        v += x[k];
    }
    uint32_t tot;
    uint32_t run = sums[blockIdx.x] + block_exclusive_scan(v, s_wave, tot);
    for (int k = 0; k < SC_PER; k++) {
        if (base + k < total) data[base + k] = run;
        run += x[k];
    }
}


inline void exclusive_scan(uint32_t *d_data, uint32_t total, uint32_t *d_sums, hipStream_t st)
{
    const uint32_t tiles = (total + SC_TILE - 1) / SC_TILE;
    k_scan_sums<<<tiles, SCAN_TB, 0, st>>>(d_data, total, d_sums);
    k_scan_top<<<1, SCAN_TB, 0, st>>>(d_sums, tiles);
    k_scan_apply<<<tiles, SCAN_TB, 0, st>>>(d_data, total, d_sums);
}


}  // namespace
