// Dev tool: how many DIVERGENT 16-byte loads per second a MI355X sustains -- the pattern of a BVH walk whose lanes sit
// on different nodes (every load instruction = one vector-L1 look-up per active lane).  Per lane and step: LOADS
// consecutive 16-B loads at a pseudo-random 64-B record of a buffer of `mb` megabytes; DEP = the next record index
// comes from the loaded data (a pointer chase, what traversal does) or from a hash (pure throughput).
// Usage: gather_rate            prints a table: footprint x loads-per-record x dependent/independent x active lanes
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

template <int LOADS, bool DEP>
__global__ __launch_bounds__(256) void k_gather(const uint4 *__restrict__ buf, uint32_t mask, int iters, int active, uint32_t *out)
{
    const int lane = threadIdx.x & 63;
    uint32_t idx = (blockIdx.x * 256u + threadIdx.x) * 2654435761u;
    uint32_t acc = 0;
    if (lane < active) {
        for (int i = 0; i < iters; i++) {
            const uint4 *p = buf + 4 * (size_t)(idx & mask);
            uint4 a = p[0], b = {}, c = {}, d = {};
            if (LOADS > 1) b = p[1];
            if (LOADS > 2) c = p[2];
            if (LOADS > 3) d = p[3];
            const uint32_t v = a.x ^ b.y ^ c.z ^ d.w;
            acc += v;
            idx = DEP ? (v + idx * 747796405u + 2891336453u) : (idx * 747796405u + 2891336453u);
        }
    }
    if (acc == 0x12345678u) out[0] = acc;
}

template <int LOADS, bool DEP>
double run(const uint4 *buf, uint32_t mask, int active, uint32_t *out, int blocks)
{
    const int iters = 2000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k_gather<LOADS, DEP><<<blocks, 256>>>(buf, mask, 100, active, out);
    hipEventRecord(e0);
    k_gather<LOADS, DEP><<<blocks, 256>>>(buf, mask, iters, active, out);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return (double)blocks * 4 * active * iters * LOADS / (ms * 1e-3) / 1e9;  // G lane-loads / s
}

int main()
{
    uint32_t *out;
    hipMalloc(&out, 4);
    const int blocks = 256 * 6;  // 6 blocks of 4 waves per CU = 6 waves per SIMD, like k_extend<hbm>
    printf("G lane-loads/s (16 B each); 6 waves/SIMD; record = 64 B\n");
    printf("%8s %6s %5s %10s %10s %10s %10s\n", "MB", "lanes", "dep", "1 load", "2 loads", "3 loads", "4 loads");
    for (size_t mb : { 16, 128, 1024, 8192 }) {
        const size_t n = mb * (1u << 20) / 64;
        uint4 *buf;
        if (hipMalloc(&buf, n * 64) != hipSuccess) continue;
        std::vector<uint32_t> h(n * 16);
        uint32_t s = 12345;
        for (auto &x : h) { s = s * 1664525u + 1013904223u; x = s; }
        hipMemcpy(buf, h.data(), n * 64, hipMemcpyHostToDevice);
        const uint32_t mask = (uint32_t)n - 1;
        for (int active : { 64, 32, 16 })
            for (int dep = 0; dep < 2; dep++) {
                double r[4];
                if (dep) { r[0] = run<1, true>(buf, mask, active, out, blocks); r[1] = run<2, true>(buf, mask, active, out, blocks);
                           r[2] = run<3, true>(buf, mask, active, out, blocks); r[3] = run<4, true>(buf, mask, active, out, blocks); }
                else { r[0] = run<1, false>(buf, mask, active, out, blocks); r[1] = run<2, false>(buf, mask, active, out, blocks);
                       r[2] = run<3, false>(buf, mask, active, out, blocks); r[3] = run<4, false>(buf, mask, active, out, blocks); }
                printf("%8zu %6d %5d %10.1f %10.1f %10.1f %10.1f\n", mb, active, dep, r[0], r[1], r[2], r[3]);
            }
        hipFree(buf);
    }
    return 0;
}
