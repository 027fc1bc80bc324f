// Dev tool: what rocprofv3's FETCH_SIZE / WRITE_SIZE report on gfx950 for access patterns whose byte count is KNOWN
// (MI355X_MICROARCH.md section HBM states the x2 only for wide coalesced streaming reads and calls
// every other access width uncalibrated).  One kernel NAME per pattern and footprint so that the counter CSV separates
// them; each kernel is launched REPS times and prints the bytes it asked for per launch:
//   k_stream_read<MB>     coalesced 16 B / lane, every byte of the footprint once
//   k_stream_write<MB>    coalesced 16 B / lane stores
//   k_gather_line<MB>     both 64-B halves of a pseudo-random 128-B line: does the L2 fetch lines or halves?
//   k_gather<MB, LOADS>   the BVH walk's pattern: every lane reads LOADS consecutive 16-B pieces of a pseudo-random
//                         64-B record (LOADS = 1: 16 B of the record, 4: the whole record)
// Footprints: 128 MB (beyond the 32 MiB of L2, inside the 256 MiB Infinity Cache) and 8192 MB (beyond both).
// Usage:  rocprofv3 --pmc FETCH_SIZE --kernel-trace -f csv -d out -o calib -- fetch_calib     (then scripts/calib_fetch.py)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

constexpr int REPS = 3;

template <int MB>
__global__ __launch_bounds__(256) void k_stream_read(const uint4 *__restrict__ buf, size_t n16, uint32_t *out)
{
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) {
        const uint4 v = buf[i];
        acc += v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345678u) out[0] = acc;
}

template <int MB>
__global__ __launch_bounds__(256) void k_stream_write(uint4 *__restrict__ buf, size_t n16)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256)
        buf[i] = make_uint4((uint32_t)i, 1u, 2u, 3u);
}

template <int MB, int LOADS>
__global__ __launch_bounds__(256) void k_gather(const uint4 *__restrict__ buf, uint32_t mask, int iters, uint32_t *out)
{
    // a full-period walk over the records per thread: idx -> idx * a + c (mod 2^32), the low log2(n) bits of which
    // visit every record once per 2^log2(n) steps; threads start at hashed offsets
    uint32_t idx = (blockIdx.x * 256u + threadIdx.x) * 2654435761u + 12345u;
    uint32_t acc = 0;
    for (int i = 0; i < iters; i++) {
        const uint4 *p = buf + 4 * (size_t)(idx & mask);
        uint4 a = p[0], b = {}, c = {}, d = {};
        if (LOADS > 1) b = p[1];
        if (LOADS > 2) c = p[2];
        if (LOADS > 3) d = p[3];
        acc += a.x ^ b.y ^ c.z ^ d.w;
        idx = idx * 747796405u + 2891336453u;
    }
    if (acc == 0x12345678u) out[0] = acc;
}

// both 64-B halves of a pseudo-random 128-B line (16 B at offset 0 and 16 B at offset 64): one request per line if the
// L2 fetches whole 128-B lines from the fabric, two if it fetches 64-B halves
template <int MB>
__global__ __launch_bounds__(256) void k_gather_line(const uint4 *__restrict__ buf, uint32_t mask, int iters, uint32_t *out)
{
    uint32_t idx = (blockIdx.x * 256u + threadIdx.x) * 2654435761u + 12345u;
    uint32_t acc = 0;
    for (int i = 0; i < iters; i++) {
        const uint4 *p = buf + 8 * (size_t)(idx & mask);
        const uint4 a = p[0], b = p[4];
        acc += a.x ^ b.y;
        idx = idx * 747796405u + 2891336453u;
    }
    if (acc == 0x12345678u) out[0] = acc;
}

template <typename F>
static float timed(F launch)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    for (int r = 0; r < REPS; r++) launch();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    hipEventDestroy(e0); hipEventDestroy(e1);
    return ms / REPS;
}

template <int MB>
static void run_footprint(uint32_t *out)
{
    const size_t bytes = (size_t)MB << 20;
    uint4 *buf = nullptr;
    if (hipMalloc(&buf, bytes) != hipSuccess) { printf("CALIB skip %d MB: hipMalloc failed\n", MB); return; }
    const size_t n16 = bytes / 16;
    const int blocks = 256 * 8;
    // fill (also the write pattern under test)
    float ms = timed([&] { k_stream_write<MB><<<blocks, 256>>>(buf, n16); });
    printf("CALIB kernel=k_stream_write<%d> launches=%d bytes_per_launch=%zu records_per_launch=0 ms=%.4f\n", MB, REPS, bytes, ms);
    ms = timed([&] { k_stream_read<MB><<<blocks, 256>>>(buf, n16, out); });
    printf("CALIB kernel=k_stream_read<%d> launches=%d bytes_per_launch=%zu records_per_launch=0 ms=%.4f\n", MB, REPS, bytes, ms);
    const uint32_t mask = (uint32_t)(bytes / 64) - 1u;
    const int gblocks = 256 * 6;  // 6 waves per SIMD, like k_extend<hbm>
    const int iters = 2000;
    const size_t records = (size_t)gblocks * 256 * iters;
    ms = timed([&] { k_gather<MB, 1><<<gblocks, 256>>>(buf, mask, iters, out); });
    printf("CALIB kernel=k_gather<%d,1> launches=%d bytes_per_launch=%zu records_per_launch=%zu ms=%.4f\n", MB, REPS, records * 16, records, ms);
    ms = timed([&] { k_gather<MB, 4><<<gblocks, 256>>>(buf, mask, iters, out); });
    printf("CALIB kernel=k_gather<%d,4> launches=%d bytes_per_launch=%zu records_per_launch=%zu ms=%.4f\n", MB, REPS, records * 64, records, ms);
    ms = timed([&] { k_gather_line<MB><<<gblocks, 256>>>(buf, mask >> 1, iters, out); });
    printf("CALIB kernel=k_gather_line<%d> launches=%d bytes_per_launch=%zu records_per_launch=%zu ms=%.4f\n", MB, REPS, records * 128, records, ms);
    hipFree(buf);
}

int main()
{
    uint32_t *out;
    hipMalloc(&out, 4);
    run_footprint<128>(out);
    run_footprint<8192>(out);
    hipDeviceSynchronize();
    return 0;
}
