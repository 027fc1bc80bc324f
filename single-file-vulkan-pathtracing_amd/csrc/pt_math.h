// pt_math.h -- device-side arithmetic of the radiance loop (gfx950, wave64).
//
// Every function here follows the project's canonical arithmetic: IEEE
// binary32, round-to-nearest-even, no FMA contraction (the whole library is compiled with
// -ffp-contract=off), correctly rounded divide and sqrt, denormals kept.  That is what makes
// the GPU radiance bit-comparable with a CPU restatement of the reference shaders.
//
// Reference anchors (paths relative to the reference repo):
//   pcg / pcg2d / rand        shaders/common.glsl:13-37
//   seed                      shaders/raygen.rgen:47-48
//   primary ray               shaders/raygen.rgen:51-57
//   sampleDirection           shaders/raygen.rgen:14-39
//   hit position / normal     shaders/closesthit.rchit:43-58
//   closest-hit semantics     shaders/raygen.rgen:63-75, main.cpp:497-538 (opaque, no culling)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define PT_MISS 0xFFFFFFFFu
#define PT_LEAF 0x80000000u
// primitives per BVH4 leaf (count-1 lives in bits 28..30 of a leaf word, so <= 8).  Measured on MI355X
// (Grays/s for leaf sizes 1/2/3/4/8): Cornell 11.5/14.3/14.6/13.8/12.7, 1M soup 1.11/1.17/1.17/1.15/1.02,
// 10k-instance grid (both levels) 5.7/5.1/4.8/3.9/2.7 -> then 2 triangles per BLAS leaf, 1 instance per TLAS leaf.
// Re-measured with the final HBM kernel (64-B nodes in four 16-B loads: a node visit costs 4 L1 look-ups per lane,
// a triangle 3): 1M soup 2.43/2.28/2.12/1.99 Grays/s for 1/2/3/4 -> 1 triangle per leaf of the collapsed LBVH.
// (Scenes <= 2048 triangles are traversed through the surface-area BVH4 below unless FAST_BUILD is asked for.)
#define PT_BLAS_LEAF_MAX 1u
#define PT_TLAS_LEAF_MAX 1u
#define PT_SAH_LEAF_MAX 4u  // surface-area BVH4 of small scenes: leaves up to 4 where splitting does not pay (C2 +1.3 % over 2)

namespace ptm {

struct f3 { float x, y, z; };

__device__ __forceinline__ float fdiv(float a, float b) { return __fdiv_rn(a, b); }

// A kernel argument (a record of dwords) copied dword by dword into scalar registers of their own.  The compiler fetches kernel arguments eight or sixteen
// dwords at a time and keeps such a group as ONE register tuple; this kernel holds more scalar values than there are scalar registers, and a tuple
// spilled to vector lanes comes back whole -- eight v_readlane_b32 for `term_pcap`.  Unused copies are dead code; a used one is one s_mov_b32 per launch.
template <class T>
__device__ __forceinline__ T own_sgprs(const T &x)
{
    static_assert(sizeof(T) % 4 == 0, "a record of dwords");
    T r;
    if constexpr (alignof(T) >= 8 && sizeof(T) % 8 == 0) {  // (records of pointers: pairs, as an address operand wants them)
        const unsigned long long *src = (const unsigned long long *)(const void *)&x;
        unsigned long long *dst = (unsigned long long *)(void *)&r;
#pragma unroll
        for (size_t i = 0; i < sizeof(T) / 8; i++) asm("s_mov_b64 %0, %1" : "=s"(dst[i]) : "s"(src[i]));
    } else {
        const uint32_t *src = reinterpret_cast<const uint32_t *>(&x);
        uint32_t *dst = reinterpret_cast<uint32_t *>(&r);
#pragma unroll
        for (size_t i = 0; i < sizeof(T) / 4; i++) asm("s_mov_b32 %0, %1" : "=s"(dst[i]) : "s"(src[i]));
    }
    return r;
}

// x / (1/(2 pi)) for three values at once (raygen.rgen:79-80 divides by the pdf as a true division).
// For THIS divisor, q0 = x*rc, r = fma(-q0, c, x), q = fma(r, rc, q0) with rc = RN(1/c) equals the
// correctly rounded quotient for every float with 2^-100 <= |x| <= 2^120: proven by enumerating all of
// them (tests/exhaustive_div_by_pdf.c, run by the CPU test-suite); anything else takes the real divide.
constexpr float PT_PDF = 0.15915493667125702f;   // 0x1.45f306p-3
constexpr float PT_PDF_RCP = 6.2831854820251465f;  // RN(1 / PT_PDF) = 0x1.921fb6p+2
__device__ __forceinline__ void div3_by_pdf(float &x, float &y, float &z)
{
    const float lo = fminf(fminf(fabsf(x), fabsf(y)), fabsf(z)), hi = fmaxf(fmaxf(fabsf(x), fabsf(y)), fabsf(z));
    if (lo >= 0x1p-100f && hi <= 0x1p+120f) {
        const float qx = x * PT_PDF_RCP, qy = y * PT_PDF_RCP, qz = z * PT_PDF_RCP;
        x = __builtin_fmaf(__builtin_fmaf(-qx, PT_PDF, x), PT_PDF_RCP, qx);
        y = __builtin_fmaf(__builtin_fmaf(-qy, PT_PDF, y), PT_PDF_RCP, qy);
        z = __builtin_fmaf(__builtin_fmaf(-qz, PT_PDF, z), PT_PDF_RCP, qz);
    } else {
        x = fdiv(x, PT_PDF); y = fdiv(y, PT_PDF); z = fdiv(z, PT_PDF);
    }
}
// ---- the correctly rounded quotient by fused multiply-adds (Markstein 1990) ------------------------------------------
// recip_rn(b) = RN(1/b) and quot_rn(a, b, recip_rn(b)) = RN(a/b): three instructions each instead of the ten of the IEEE
// divide expansion (v_div_scale x2, v_rcp, four fma, v_div_fmas, v_div_fixup), and quotients that share a divisor share the
// reciprocal.  The results ARE the IEEE quotients, bit for bit -- not by argument but by enumeration on this chip
// (scripts/ubench/exact_div.hip, profiles/r03_exact_div_proof.txt): the reciprocal for every normal b with
// 2^-126 <= |b| < 2^126, the quotient for all 2^46 pairs of significands -- provided nothing overflows, underflows or
// turns denormal on the way (the sequences are exact scalings by powers of two away from 1 <= a, b < 2 otherwise), which
// is what the guards of the callers below establish: the divisor within 2^+-20 (2^+-40) and every dividend at least
// 2^-100 (2^-60) in magnitude and no larger than the divisor (than 2^60).  A zero dividend takes the IEEE path too (the
// residual would lose the sign of a -0).  Operands outside the guards take the real divide: same bits, old speed.
__device__ __forceinline__ float recip_rn(float b)
{
    const float y0 = __builtin_amdgcn_rcpf(b);
    return __builtin_fmaf(__builtin_fmaf(-b, y0, 1.0f), y0, y0);
}
__device__ __forceinline__ float quot_rn(float a, float b, float y)
{
    const float q0 = a * y;
    return __builtin_fmaf(__builtin_fmaf(-b, q0, a), y, q0);
}
// |b| in [2^-20, 2^20] and the smaller dividend at least 2^-100 (the callers' dividends never exceed the divisor in magnitude)
__device__ __forceinline__ bool quot_guard_dominant(float b, float amin_abs)
{
    const float ab = fabsf(b);
    return __builtin_amdgcn_fmed3f(ab, 0x1p-20f, 0x1p+20f) == ab && amin_abs >= 0x1p-100f;
}
// a1 / b, a2 / b with |a1|, |a2| <= |b| (a hit's barycentric numerators over their sum; direction components over the dominant one)
__device__ __forceinline__ void div2_dominant(float a1, float a2, float b, float &q1, float &q2)
{
    if (quot_guard_dominant(b, fminf(fabsf(a1), fabsf(a2)))) {
        const float y = recip_rn(b);
        q1 = quot_rn(a1, b, y); q2 = quot_rn(a2, b, y);
    } else {
        q1 = fdiv(a1, b); q2 = fdiv(a2, b);
    }
}
// a1 / b, a2 / b, a3 / b with |a_i| <= |b| (a vector over its length)
__device__ __forceinline__ void div3_dominant(float a1, float a2, float a3, float b, float &q1, float &q2, float &q3)
{
    if (quot_guard_dominant(b, fminf(fminf(fabsf(a1), fabsf(a2)), fabsf(a3)))) {
        const float y = recip_rn(b);
        q1 = quot_rn(a1, b, y); q2 = quot_rn(a2, b, y); q3 = quot_rn(a3, b, y);
    } else {
        q1 = fdiv(a1, b); q2 = fdiv(a2, b); q3 = fdiv(a3, b);
    }
}
// NOTE: __fsqrt_rn() lowers to the bare 1-ulp v_sqrt_f32 on gfx950 (ROCm 7.2); __builtin_sqrtf
// gets the correctly rounded expansion (v_sqrt_f32 + two fma corrections), which is what the
// canonical arithmetic requires.
__device__ __forceinline__ float fsqrt(float a) { return __builtin_sqrtf(a); }

// ---- RNG ------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t pcg(uint32_t &state)  // common.glsl:13-19
{
    const uint32_t prev = state * 747796405u + 2891336453u;
    const uint32_t word = ((prev >> ((prev >> 28u) + 4u)) ^ prev) * 277803737u;
    state = prev;
    return (word >> 22u) ^ word;
}

__device__ __forceinline__ uint2 pcg2d(uint2 v)  // common.glsl:21-31
{
    v.x = v.x * 1664525u + 1013904223u;
    v.y = v.y * 1664525u + 1013904223u;
    v.x += v.y * 1664525u;
    v.y += v.x * 1664525u;
    v.x ^= v.x >> 16u;
    v.y ^= v.y >> 16u;
    v.x += v.y * 1664525u;
    v.y += v.x * 1664525u;
    v.x ^= v.x >> 16u;
    v.y ^= v.y >> 16u;
    return v;
}

__device__ __forceinline__ float rnd(uint32_t &seed)  // common.glsl:33-37
{
    // float(val) is v_cvt_f32_u32 = round-to-nearest-even; the constant is 2^-32
    return __uint2float_rn(pcg(seed)) * 2.3283064365386963e-10f;
}

__device__ __forceinline__ uint32_t make_seed(uint32_t px, uint32_t py, uint32_t sample, int32_t frame,
                                              uint32_t spp)  // raygen.rgen:47-48
{
    const uint32_t m = sample + (uint32_t)((int32_t)spp * frame) + 1u;
    const uint2 s = pcg2d(make_uint2(px * m, py * m));
    return s.x + s.y;
}

// ---- sin/cos of a in [0, 2*pi] ---------------------------------------------------------
// Quadrant reduction (Cody-Waite, three-part pi/2) + cephes-style minimax polynomials on
// |r| <= pi/4, Horner form.  Fully specified so host and device agree to the bit; accuracy
// ~1e-7 absolute, far inside what GLSL.std.450 Sin/Cos guarantee (2^-11).
__device__ __forceinline__ void sincos_2pi(float a, float &s, float &c)
{
    const int j = (int)(a * 0.636619772f + 0.5f);
    const float fj = (float)j;
    float r = a - fj * 1.5703125f;
    r = r - fj * 4.837512969970703125e-4f;
    r = r - fj * 7.54978995489188e-8f;
    const float z = r * r;
    float ps = -1.9515295891e-4f * z + 8.3321608736e-3f;
    ps = ps * z - 1.6666654611e-1f;
    ps = ps * z;
    ps = ps * r + r;
    float pc = 2.443315711809948e-5f * z - 1.388731625493765e-3f;
    pc = pc * z + 4.166664568298827e-2f;
    pc = pc * z;
    pc = pc * z;
    pc = pc - 0.5f * z;
    pc = pc + 1.0f;
    const bool swap = j & 1;
    const float s0 = swap ? pc : ps;
    const float c0 = swap ? ps : pc;
    s = (j & 2) ? -s0 : s0;           // j&3: 0 (s,c) 1 (c,-s) 2 (-s,-c) 3 (-c,s)
    c = ((j + 1) & 2) ? -c0 : c0;
}

// ---- camera ----------------------------------------------------------------------------
struct Camera {
    float ox, oy, oz;   // raygen.rgen:55
    float tx, ty, tz;   // raygen.rgen:56: target = (d.x + tx, d.y + ty, tz)
    float w, h;         // float(gl_LaunchSizeEXT.xy)
    float rw, rh;       // RN(1 / w), RN(1 / h) (the host's IEEE divide), or 0: the divides by the launch size take the IEEE expansion
};

// raygen.rgen:51-56 for given jitters: the un-normalised direction target - origin of the camera ray through pixel (px, py)
__device__ __forceinline__ void primary_target(const Camera &cam, uint32_t px, uint32_t py, float jx, float jy, float &vx, float &vy, float &vz)
{
    const float sx = (float)px + jx;
    const float sy = (float)py + jy;
    // screenPos / size (raygen.rgen:52) is a true divide.  The divisor is the same for every ray, so its correctly rounded reciprocal comes with
    // the launch constants and the quotient is quot_rn's three instructions instead of the ten of the IEEE expansion -- the IEEE quotient bit for
    // bit under that sequence's proven conditions (above): divisor in [1, 2^20] (the host passes 0 as the reciprocal otherwise), dividend no
    // larger than the divisor (pixel + jitter <= size) and either +0 -- q0 = +0, residual fma(-b, +0, +0) = +0, result +0, the IEEE 0 / b -- or
    // at least 2^-32 (a pixel >= 1, or rand's smallest non-zero value alone).
    float qx, qy;
    if (cam.rw != 0.0f && cam.rh != 0.0f) {  // (uniform)
        qx = quot_rn(sx, cam.w, cam.rw); qy = quot_rn(sy, cam.h, cam.rh);
    } else {
        qx = fdiv(sx, cam.w); qy = fdiv(sy, cam.h);
    }
    const float dx = qx * 2.0f - 1.0f;  // no aspect-ratio correction (kept)
    const float dy = qy * 2.0f - 1.0f;
    vx = (dx + cam.tx) - cam.ox;
    vy = (dy + cam.ty) - cam.oy;
    vz = cam.tz - cam.oz;
}
__device__ __forceinline__ void primary_ray(const Camera &cam, uint32_t px, uint32_t py, uint32_t &seed,
                                            f3 &org, f3 &dir)  // raygen.rgen:51-57
{
    const float jx = rnd(seed);  // x first, then y
    const float jy = rnd(seed);
    float vx, vy, vz;
    primary_target(cam, px, py, jx, jy, vx, vy, vz);
    const float len = fsqrt((vx * vx + vy * vy) + vz * vz);
    org = { cam.ox, cam.oy, cam.oz };
    div3_dominant(vx, vy, vz, len, dir.x, dir.y, dir.z);
}

// ---- bounce: raygen.rgen:14-39 ---------------------------------------------------------
// createCoordinateSystem alone (raygen.rgen:14-21): depends on the normal only, so k_pack evaluates it once per
// triangle with exactly these operations and k_shade reads it from its LDS tables (one square root, two true divides
// and a cross product less per bounce)
__device__ __forceinline__ void tangent_frame(const f3 n, f3 &T, f3 &B)
{
    const bool bx = fabsf(n.x) > fabsf(n.y);
    const float p = bx ? n.x : n.y, q = n.z;
    const float l = fsqrt(p * p + q * q);
    const float ql = fdiv(q, l), pl = fdiv(p, l);
    const float zl = l > 0.0f ? 0.0f : __builtin_nanf("");
    T = { bx ? ql : zl, bx ? zl : -ql, bx ? -pl : pl };
    B = { n.y * T.z - n.z * T.y, n.z * T.x - n.x * T.z, n.x * T.y - n.y * T.x };
}
// sampleHemisphere + the change of basis (raygen.rgen:23-39) for a given frame
__device__ __forceinline__ f3 sample_direction_frame(float r1, float r2, const f3 n, const f3 T, const f3 B);

__device__ __forceinline__ f3 sample_direction(float r1, float r2, const f3 n)
{
    // createCoordinateSystem (strict >): Nt = normalize(n.z, 0, -n.x) or normalize(0, -n.z, n.y).  Both
    // branches are sqrt(p*p + q*q) and the quotients q/l, p/l with (p, q) = (n.x, n.z) or (n.y, n.z), so
    // they are taken once and placed by selects -- same operands, same bits, no divergent second pass.
    // (-p)/l == -(p/l) exactly; 0/l is +0 for l > 0 and NaN otherwise (l is a square root: never < 0).
    const bool bx = fabsf(n.x) > fabsf(n.y);
    const float p = bx ? n.x : n.y, q = n.z;
    const float l = fsqrt(p * p + q * q);
    const float ql = fdiv(q, l), pl = fdiv(p, l);
    const float zl = l > 0.0f ? 0.0f : __builtin_nanf("");
    const f3 T = { bx ? ql : zl, bx ? zl : -ql, bx ? -pl : pl };
    const f3 B = { n.y * T.z - n.z * T.y, n.z * T.x - n.x * T.z, n.x * T.y - n.y * T.x };
    const float sq = fsqrt(1.0f - r1 * r1);  // uniform hemisphere, pdf 1/(2*pi)
    const float phi = 6.2831854820251465f * r2;
    float sn, cs;
    sincos_2pi(phi, sn, cs);
    const float dx = cs * sq, dy = sn * sq, dz = r1;
    return { (T.x * dx + B.x * dy) + n.x * dz, (T.y * dx + B.y * dy) + n.y * dz,
             (T.z * dx + B.z * dy) + n.z * dz };
}
// ... with sq = sqrt(1 - r1 * r1) given (the fused kernel takes that root beside the camera rays' length)
__device__ __forceinline__ f3 sample_direction_frame_sq(float r1, float r2, float sq, const f3 n, const f3 T, const f3 B)
{
    const float phi = 6.2831854820251465f * r2;
    float sn, cs;
    sincos_2pi(phi, sn, cs);
    const float dx = cs * sq, dy = sn * sq, dz = r1;
    return { (T.x * dx + B.x * dy) + n.x * dz, (T.y * dx + B.y * dy) + n.y * dz,
             (T.z * dx + B.z * dy) + n.z * dz };
}
__device__ __forceinline__ f3 sample_direction_frame(float r1, float r2, const f3 n, const f3 T, const f3 B)
{
    const float sq = fsqrt(1.0f - r1 * r1);  // uniform hemisphere, pdf 1/(2*pi)
    const float phi = 6.2831854820251465f * r2;
    float sn, cs;
    sincos_2pi(phi, sn, cs);
    const float dx = cs * sq, dy = sn * sq, dz = r1;
    return { (T.x * dx + B.x * dy) + n.x * dz, (T.y * dx + B.y * dy) + n.y * dz,
             (T.z * dx + B.z * dy) + n.z * dz };
}

// ---- geometric normal: closesthit.rchit:43-48 -------------------------------------------
__device__ __forceinline__ f3 tri_normal(const f3 v0, const f3 v1, const f3 v2)
{
    const f3 a = { v1.x - v0.x, v1.y - v0.y, v1.z - v0.z };
    const f3 b = { v2.x - v0.x, v2.y - v0.y, v2.z - v0.z };
    const float cx = a.y * b.z - a.z * b.y;
    const float cy = a.z * b.x - a.x * b.z;
    const float cz = a.x * b.y - a.y * b.x;
    const float len = fsqrt((cx * cx + cy * cy) + cz * cz);
    return { -fdiv(cx, len), -fdiv(cy, len), -fdiv(cz, len) };
}

// ---- closest-hit query: watertight ray/triangle (Woop, Benthin, Wald, JCGT 2013) --------
struct RayPre {
    f3 org;
    float Sx, Sy, Sz;
    int kz;  // dominant axis; kx = (kz+1)%3, ky = (kz+2)%3 (winding swap omitted: no culling)
};

__device__ __forceinline__ float sel3(int k, float x, float y, float z) { return k == 0 ? x : (k == 1 ? y : z); }

// SHORT_DIV = false keeps the IEEE expansion: its three divides run one after the other (they pass VCC along), which the
// per-triangle leaf loop of k_extend_lds7 needs to stay within 72 registers
template <bool SHORT_DIV = true>
__device__ __forceinline__ RayPre ray_setup(const f3 org, const f3 dir)
{
    RayPre r;
    int kz = 0;
    if (fabsf(dir.y) > fabsf(dir.x)) kz = 1;
    if (fabsf(dir.z) > fabsf(sel3(kz, dir.x, dir.y, dir.z))) kz = 2;
    // permuted components: (kx,ky,kz) = (1,2,0) (2,0,1) (0,1,2)
    const float dkx = sel3(kz, dir.y, dir.z, dir.x);
    const float dky = sel3(kz, dir.z, dir.x, dir.y);
    const float dkz = sel3(kz, dir.x, dir.y, dir.z);
    // three true divides by the dominant component: |dkx|, |dky| <= |dkz|, and 1/dkz is the reciprocal itself
    if (SHORT_DIV && quot_guard_dominant(dkz, fminf(fabsf(dkx), fabsf(dky)))) {
        const float y = recip_rn(dkz);
        r.Sx = quot_rn(dkx, dkz, y);
        r.Sy = quot_rn(dky, dkz, y);
        r.Sz = y;
    } else {
        r.Sx = fdiv(dkx, dkz);
        r.Sy = fdiv(dky, dkz);
        r.Sz = fdiv(1.0f, dkz);
    }
    r.kz = kz;
    r.org = org;
    return r;
}

// Tests one triangle; on a hit inside (tmin, tmax) returns true with t and the UNDIVIDED
// barycentric numerators V, W and det: u = V/det (weight of v1 = attribs.x), v = W/det (weight of
// v2 = attribs.y).  The two divides are deferred to the hit that finally wins (same operands,
// same bits, fewer IEEE divides).
__device__ __forceinline__ bool tri_test(const RayPre &r, const f3 v0, const f3 v1, const f3 v2, float tmin,
                                         float tmax, float &t, float &Vn, float &Wn, float &detn,
                                         bool *reached_divide = nullptr)  // instrumented builds only
{
    const f3 A = { v0.x - r.org.x, v0.y - r.org.y, v0.z - r.org.z };
    const f3 B = { v1.x - r.org.x, v1.y - r.org.y, v1.z - r.org.z };
    const f3 C = { v2.x - r.org.x, v2.y - r.org.y, v2.z - r.org.z };
    const int kz = r.kz;
    const float Akx = sel3(kz, A.y, A.z, A.x), Aky = sel3(kz, A.z, A.x, A.y), Akz = sel3(kz, A.x, A.y, A.z);
    const float Bkx = sel3(kz, B.y, B.z, B.x), Bky = sel3(kz, B.z, B.x, B.y), Bkz = sel3(kz, B.x, B.y, B.z);
    const float Ckx = sel3(kz, C.y, C.z, C.x), Cky = sel3(kz, C.z, C.x, C.y), Ckz = sel3(kz, C.x, C.y, C.z);
    const float Ax = Akx - r.Sx * Akz, Ay = Aky - r.Sy * Akz;
    const float Bx = Bkx - r.Sx * Bkz, By = Bky - r.Sy * Bkz;
    const float Cx = Ckx - r.Sx * Ckz, Cy = Cky - r.Sy * Ckz;
    const float U = Cx * By - Cy * Bx;
    const float V = Ax * Cy - Ay * Cx;
    const float W = Bx * Ay - By * Ax;
    // a zero edge function counts as inside: shared edges stay watertight
    if ((U < 0.0f || V < 0.0f || W < 0.0f) && (U > 0.0f || V > 0.0f || W > 0.0f)) return false;
    const float det = (U + V) + W;
    if (det == 0.0f) return false;
    if (reached_divide) *reached_divide = true;
    const float Az = r.Sz * Akz, Bz = r.Sz * Bkz, Cz = r.Sz * Ckz;
    const float T = (U * Az + V * Bz) + W * Cz;
    const float tt = fdiv(T, det);
    if (!(tt > tmin && tt < tmax)) return false;  // tMin < t < tMax, NaN rejects
    t = tt;
    Vn = V;
    Wn = W;
    detn = det;
    return true;
}

// Reciprocal for the slab tests only (traversal is not part of the numerical contract: boxes are
// padded by 2^-18 of the scene scale and the far distance by 4e-7, far more than v_rcp_f32's 1 ulp).
__device__ __forceinline__ float safe_inv(float d)
{
    // (|d| clamped from below with d's sign put back: v_max + v_bfi, a compare and a select less per axis than `if (|d| < 1e-20) d = copysign(1e-20, d)`;
    // the same value for every d that is not a NaN)
    float m;  // (the instruction itself: fmaxf() would re-quiet its operand with a v_max x, x first)
    asm("v_max_f32 %0, |%1|, %2" : "=v"(m) : "v"(d), "s"(1e-20f));
    return __builtin_amdgcn_rcpf(copysignf(m, d));
}

// Triangle test on vertices whose components were permuted to (kx, ky, kz) at build time and a ray
// origin permuted the same way: the per-vertex component selects of tri_test() disappear, every
// remaining operation has the same operands, so the result is bit-identical.
__device__ __forceinline__ bool tri_test_perm(const RayPre &r, const f3 orgp, const f3 v0, const f3 v1, const f3 v2,
                                              float tmin, float tmax, float &t, float &Vn, float &Wn, float &detn,
                                              bool *reached_divide = nullptr)  // instrumented builds only
{
    const f3 A = { v0.x - orgp.x, v0.y - orgp.y, v0.z - orgp.z };
    const f3 B = { v1.x - orgp.x, v1.y - orgp.y, v1.z - orgp.z };
    const f3 C = { v2.x - orgp.x, v2.y - orgp.y, v2.z - orgp.z };
    const float Ax = A.x - r.Sx * A.z, Ay = A.y - r.Sy * A.z;
    const float Bx = B.x - r.Sx * B.z, By = B.y - r.Sy * B.z;
    const float Cx = C.x - r.Sx * C.z, Cy = C.y - r.Sy * C.z;
    const float U = Cx * By - Cy * Bx;
    const float V = Ax * Cy - Ay * Cx;
    const float W = Bx * Ay - By * Ax;
    if ((U < 0.0f || V < 0.0f || W < 0.0f) && (U > 0.0f || V > 0.0f || W > 0.0f)) return false;
    const float det = (U + V) + W;
    if (det == 0.0f) return false;
    if (reached_divide) *reached_divide = true;
    const float Az = r.Sz * A.z, Bz = r.Sz * B.z, Cz = r.Sz * C.z;
    const float T = (U * Az + V * Bz) + W * Cz;
    const float tt = fdiv(T, det);
    if (!(tt > tmin && tt < tmax)) return false;
    t = tt;
    Vn = V;
    Wn = W;
    detn = det;
    return true;
}

// Conservative slab test against [tmin, tbest]; only has to never cull a real hit.
__device__ __forceinline__ bool box_test(const f3 mn, const f3 mx, const f3 org, const f3 inv, float tmin,
                                         float tbest, float &tnear)
{
    const float x0 = (mn.x - org.x) * inv.x, x1 = (mx.x - org.x) * inv.x;
    const float y0 = (mn.y - org.y) * inv.y, y1 = (mx.y - org.y) * inv.y;
    const float z0 = (mn.z - org.z) * inv.z, z1 = (mx.z - org.z) * inv.z;
    const float tn = fmaxf(fmaxf(fminf(x0, x1), fminf(y0, y1)), fmaxf(fminf(z0, z1), tmin));
    const float tf = fminf(fminf(fmaxf(x0, x1), fmaxf(y0, y1)), fminf(fmaxf(z0, z1), tbest));
    tnear = tn;
    return tn <= tf * 1.0000004f;
}

}  // namespace ptm

// ---- streamed-once queue traffic ---------------------------------------------------------------
// Queue records, hit records and rays are written by one kernel and read exactly once by the next, tens of MB to GB later.
// NT = true gives them the `nt` cache policy (non-temporal: do not keep the line).  Measured per component, same box, interleaved
// processes (profiles/r03cl_* ... r03cp_*):
//   * k_shade's queue STORES (48 B per surviving path, read by the extend launch after next at the earliest): nt for every scene
//     class -- C2 27.1 -> 28.15 Grays/s (+3.8 %, six of six rounds, tighter spread), config C2 exactly 25.0 -> 27.2, C4 / C5 / C5x +0.3 %;
//   * the extend kernels' hit-record store: plain for the Cornell kernel (nt: -7 %, its reader is the very next launch) and the
//     big-scene kernels; nt in k_extend_inst16 together with its ray loads (C4 +2.4 % with the shade side);
//   * queue LOADS: neutral everywhere (nt only in the instanced kernels, where the whole set was measured together); the 8-wide
//     kernel's ray loads alone: C5 -0.2 %, C5x -1.8 % (profiles/r03ct_ab_e8_nt_loads.log): plain;
//   * term-log stores and k_generate's queue stores: nt (C4 +2.2 %, C2 neutral).
namespace ptm {
typedef float f4v_ __attribute__((ext_vector_type(4)));
typedef float f2v_ __attribute__((ext_vector_type(2)));
typedef unsigned u2v_ __attribute__((ext_vector_type(2)));
template <bool NT> __device__ __forceinline__ float4 ld_stream(const float4 *p)
{
    if (NT) { const f4v_ v = __builtin_nontemporal_load(reinterpret_cast<const f4v_ *>(p)); return make_float4(v.x, v.y, v.z, v.w); }
    return *p;
}
template <bool NT> __device__ __forceinline__ float2 ld_stream(const float2 *p)
{
    if (NT) { const f2v_ v = __builtin_nontemporal_load(reinterpret_cast<const f2v_ *>(p)); return make_float2(v.x, v.y); }
    return *p;
}
template <bool NT> __device__ __forceinline__ uint2 ld_stream(const uint2 *p)
{
    if (NT) { const u2v_ v = __builtin_nontemporal_load(reinterpret_cast<const u2v_ *>(p)); return make_uint2(v.x, v.y); }
    return *p;
}
template <bool NT> __device__ __forceinline__ void st_stream(float4 *p, const float4 a)
{
    if (NT) { const f4v_ v = { a.x, a.y, a.z, a.w }; __builtin_nontemporal_store(v, reinterpret_cast<f4v_ *>(p)); }
    else *p = a;
}
template <bool NT> __device__ __forceinline__ void st_stream(float2 *p, const float2 a)
{
    if (NT) { const f2v_ v = { a.x, a.y }; __builtin_nontemporal_store(v, reinterpret_cast<f2v_ *>(p)); }
    else *p = a;
}
template <bool NT> __device__ __forceinline__ void st_stream(uint2 *p, const uint2 a)
{
    if (NT) { const u2v_ v = { a.x, a.y }; __builtin_nontemporal_store(v, reinterpret_cast<u2v_ *>(p)); }
    else *p = a;
}
}  // namespace ptm
