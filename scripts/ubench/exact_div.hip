// Dev tool / proof: the fma division sequences of csrc/pt_math.h against the IEEE divide, EXHAUSTIVELY.
//
//   recip(b):      y0 = v_rcp_f32(b); e = fma(-b, y0, 1); y = fma(e, y0, y0)           claimed: y == RN(1/b)
//   quot(a, b, y): q0 = a * y;  r = fma(-b, q0, a);  q = fma(r, y, q0)                 claimed: q == RN(a/b)
//
// Both are exact scalings by powers of two away from the case 1 <= a, b < 2 as long as nothing overflows, underflows or
// turns denormal (the guards in pt_math.h keep the operands where that holds), so enumerating every significand of b
// (2^23, and every exponent for the reciprocal) and every PAIR of significands (2^46) for the quotient is a proof.
//   exact_div recip            all normal b with 2^-126 <= |b| < 2^126, both signs
//   exact_div quot [b0 b1]     significand pairs, b's significand in [b0, b1) (default: all 2^23), a's: all 2^23
//   exact_div selfcheck [b0 b1] the same enumeration of q0 alone: must report mismatches (about a quarter of the pairs)
// Prints the number of mismatches (expected: 0) and the first few.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cstdint>

__device__ __forceinline__ float recip_fast(float b)
{
    const float y0 = __builtin_amdgcn_rcpf(b);
    const float e = __builtin_fmaf(-b, y0, 1.0f);
    return __builtin_fmaf(e, y0, y0);
}
__device__ __forceinline__ float quot_fast(float a, float b, float y)
{
    const float q0 = a * y;
    const float r = __builtin_fmaf(-b, q0, a);
    return __builtin_fmaf(r, y, q0);
}

__global__ void k_recip(unsigned long long *bad, uint32_t *examples)
{
    // exponents 1 .. 252 (2^-126 .. 2^125), 2^23 significands, 2 signs
    const unsigned long long total = 252ull << 24;
    for (unsigned long long i = blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x; i < total; i += (unsigned long long)gridDim.x * blockDim.x) {
        const uint32_t sign = (uint32_t)(i & 1ull), mant = (uint32_t)((i >> 1) & 0x7FFFFFu), ex = (uint32_t)(i >> 24) + 1u;
        const float b = __uint_as_float((sign << 31) | (ex << 23) | mant);
        const float y = recip_fast(b), want = __fdiv_rn(1.0f, b);
        if (__float_as_uint(y) != __float_as_uint(want)) {
            const unsigned long long k = atomicAdd(bad, 1ull);
            if (k < 16) examples[k] = __float_as_uint(b);
        }
    }
}

template <bool BROKEN>  // BROKEN: q0 alone, the harness's self-check (a one-ulp quotient must be caught)
__global__ void k_quot(uint32_t b0, uint32_t b1, unsigned long long *bad, uint32_t *examples)
{
    // one b per block iteration (its reciprocal is wave-uniform work), a's significands across the threads
    for (uint32_t mb = b0 + blockIdx.x; mb < b1; mb += gridDim.x) {
        const float b = __uint_as_float(0x3F800000u | mb);
        const float y = recip_fast(b);
        unsigned long long local = 0;
        for (uint32_t ma = threadIdx.x; ma < (1u << 23); ma += blockDim.x) {
            const float a = __uint_as_float(0x3F800000u | ma);
            const float q = BROKEN ? a * y : quot_fast(a, b, y), want = __fdiv_rn(a, b);
            if (__float_as_uint(q) != __float_as_uint(want)) {
                local++;
                const unsigned long long k = atomicAdd(bad + 1, 1ull);
                if (k < 8) { examples[2 * k] = __float_as_uint(a); examples[2 * k + 1] = __float_as_uint(b); }
            }
        }
        if (local) atomicAdd(bad, local);
    }
}

int main(int argc, char **argv)
{
    const char *mode = argc > 1 ? argv[1] : "recip";
    unsigned long long *d_bad; uint32_t *d_ex;
    hipMalloc(&d_bad, 16); hipMalloc(&d_ex, 64 * 4);
    hipMemset(d_bad, 0, 16); hipMemset(d_ex, 0, 64 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    unsigned long long cases = 0;
    if (!strcmp(mode, "recip")) {
        k_recip<<<256 * 16, 256>>>(d_bad, d_ex);
        cases = 252ull << 24;
    } else {
        const uint32_t b0 = argc > 2 ? (uint32_t)strtoul(argv[2], 0, 0) : 0u, b1 = argc > 3 ? (uint32_t)strtoul(argv[3], 0, 0) : (1u << 23);
        if (!strcmp(mode, "selfcheck")) k_quot<true><<<256 * 8, 256>>>(b0, b1, d_bad, d_ex);
        else k_quot<false><<<256 * 8, 256>>>(b0, b1, d_bad, d_ex);
        cases = (unsigned long long)(b1 - b0) << 23;
    }
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long bad[2]; uint32_t ex[64];
    hipMemcpy(bad, d_bad, 16, hipMemcpyDeviceToHost); hipMemcpy(ex, d_ex, 64 * 4, hipMemcpyDeviceToHost);
    printf("%s: %llu cases in %.3f s, mismatches %llu\n", mode, cases, ms * 1e-3, bad[0]);
    if (!strcmp(mode, "recip")) { for (unsigned k = 0; k < 16 && k < bad[0]; k++) printf("  b = 0x%08x (%a)\n", ex[k], *(float *)&ex[k]); }
    else for (unsigned k = 0; k < 8 && k < bad[1]; k++) printf("  a = 0x%08x  b = 0x%08x\n", ex[2 * k], ex[2 * k + 1]);
    return bad[0] ? 1 : 0;
}
