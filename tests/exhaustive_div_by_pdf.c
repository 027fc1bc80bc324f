/* Proof by enumeration for single-file-vulkan-pathtracing_amd/csrc/pt_math.h: div3_by_pdf().
 * For the one divisor c = 1/(2 pi) rounded to float, the three-instruction sequence
 *     q0 = x * rc;  r = fma(-q0, c, x);  q = fma(r, rc, q0);        rc = RN(1/c)
 * must equal the IEEE-754 correctly rounded x / c for EVERY float x with 2^-100 <= |x| <= 2^120
 * (the guard of the fast path).  All of them are tried; fmaf is the correctly rounded fma on both
 * sides (x86 FMA3 here, v_fma_f32 on the GPU).  Exit code 0 = no mismatch.  Built and run by
 * tests/test_host_and_abi.py.  Compile with -O2 -mfma -ffp-contract=off. */
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <stdlib.h>
static const float c = 0.15915493667125702f;
static float rc;
typedef struct { uint32_t lo, hi; uint64_t bad, tried; } job;
static inline float asf(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static inline uint32_t asu(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static void *run(void *p)
{
    job *j = (job *)p;
    j->bad = j->tried = 0;
    for (uint64_t i = j->lo; i <= j->hi; i++) {
        for (uint32_t sign = 0; sign < 2; sign++) {
            const float x = asf((uint32_t)i | (sign << 31));
            const float want = x / c;
            const float q0 = x * rc;
            const float r = fmaf(-q0, c, x);
            const float q = fmaf(r, rc, q0);
            j->tried++;
            if (asu(q) != asu(want)) j->bad++;
        }
    }
    return 0;
}
int main(int argc, char **argv)
{
    int T = argc > 1 ? atoi(argv[1]) : 8;
    if (T < 1) T = 1;
    if (T > 64) T = 64;
    rc = 1.0f / c;
    if (asu(rc) != asu(6.2831854820251465f)) { printf("rc constant mismatch\n"); return 2; }
    const uint32_t first = asu(0x1p-100f), last = asu(0x1p+120f);   /* positive bit patterns are ordered */
    pthread_t th[64];
    job jb[64];
    const uint64_t span = (uint64_t)last - first + 1;
    for (int t = 0; t < T; t++) {
        jb[t].lo = (uint32_t)(first + span * t / T);
        jb[t].hi = (uint32_t)(first + span * (t + 1) / T - 1);
        pthread_create(&th[t], 0, run, &jb[t]);
    }
    uint64_t bad = 0, tried = 0;
    for (int t = 0; t < T; t++) { pthread_join(th[t], 0); bad += jb[t].bad; tried += jb[t].tried; }
    printf("tried %llu mismatches %llu\n", (unsigned long long)tried, (unsigned long long)bad);
    return bad ? 1 : 0;
}
