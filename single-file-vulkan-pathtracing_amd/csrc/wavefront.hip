// wavefront.hip -- the per-pixel radiance loop as a wavefront path tracer on gfx950.
//
// Replaces what runs behind vkCmdTraceRaysKHR (main.cpp:659): raygen.rgen:41-91 with its
// traceRayEXT (raygen.rgen:63-75), closesthit.rchit:50-65 and miss.rmiss:8-12.
//
// Structure.  A *slot* is one (frame, sample group, pixel) triple; it runs its
// samples one after the other ("regeneration"), and its radiance is either accumulated in path
// order (one group) or logged term by term and replayed in order by k_resolve (several groups):
// either way the reference's single `color` accumulator (raygen.rgen:42,76) is reproduced
// add-for-add.  Live paths sit in dense, double-buffered queues (index = queue position, all
// accesses coalesced):
//     qid {slot, sample | depth<<16}, qstate {seed, weight}, qray {origin, direction}
// One round = two kernels over the live queue:
//     k_extend  : closest hit of ray[q] -> hit[q]   persistent threads over the BVH4, LDS short
//                                                   stack + HBM spill, lane refill; nodes and
//                                                   triangles staged in LDS when the scene is small
//                 (k_extend_inst: the same over TLAS + BLAS; k_extend_flat: one wide leaf)
//     k_shade   : hit[q] -> emission/environment radiance, bounce, or the next sample of the
//                 slot; survivors are compacted into the other queue with wave ballots + one
//                 atomic per 1024-path chunk
// k_generate fills the queue for a batch of frames, k_resolve applies raygen.rgen:86-90.
#include "pt_internal.h"
#include "pt_math.h"

#include <hip/hip_ext.h>

#include <algorithm>
#include <cstdlib>
#include <vector>

#include "extend_kernel.h"  // TB, SENTINEL, NormBox, k_extend<>, k_extend_lds7, the slab / stack helpers
#include "extend_inst16.h"  // k_extend_inst16: the two-level kernel over 64-B fp16 nodes with one-dword stack entries

// extend_hbm.hip: k_extend<false, *, true>, compiled with the max-ILP scheduler
const void *ptw_extend_hbm_fn(bool count, bool rec64);
void ptw_launch_extend_hbm(bool count, bool rec64, int grid, size_t smem, hipStream_t st, hipEvent_t ev0, hipEvent_t ev1,
                           const float4 *wide, const uint2 *wide16, const float *norm_c, const float *norm_s,
                           const float *norm_rs, const float4 *tri4, const float4 *rec64_tab, uint32_t n_wide, uint32_t n_tris,
                           const float4 *rayA, const float2 *rayB, float4 *hit, const uint32_t *count_in, uint32_t *count_zero,
                           unsigned long long *stats, uint2 *spill, uint32_t spill_stride, int refill, float tmin,
                           float tmax, int lds_stack, int raw_hit, const uint32_t *perm, const float *ray_tmax);

// extend_hbm.hip: k_extend8 (BVH8)
const void *ptw_extend8_fn(bool count, bool spills, bool waves7);
void ptw_launch_extend8(bool count, bool spills, bool waves7, int grid, size_t smem, hipStream_t st, hipEvent_t ev0, hipEvent_t ev1, const uint4 *nodes8,
                        const float *norm_c, const float *norm_s, const float *norm_rs, const float4 *tri4, const float4 *rec64, const float4 *rayA,
                        const float2 *rayB, float4 *hit, const uint32_t *count_in, uint32_t *count_zero, unsigned long long *stats,
                        uint2 *spill, uint32_t spill_stride, int refill, float tmin, float tmax, int lds_stack, int raw_hit,
                        const uint32_t *perm, const float *ray_tmax);

// ray_sort.hip
size_t ptw_ray_sort_bytes(size_t cap);
const uint32_t *ptw_sort_rays(hipStream_t st, const float4 *rayA, const float2 *rayB, const uint32_t *count, size_t cap,
                              const float *bmin, const float *bmax, int bits, int num_cus, void *scratch);

namespace {

// Exact unsigned division by a run-time constant without the ~28-instruction v_rcp sequence
// (Granlund & Montgomery 1994, N = 32): q = (t + ((n - t) >> s1)) >> s2 with t = mulhi(m, n).
struct FastDiv {
    uint32_t m = 1, s1 = 0, s2 = 0;
    void init(uint32_t d)
    {
        uint32_t l = 0;
        while ((1ull << l) < d) l++;
        m = (uint32_t)((((1ull << l) - d) << 32) / d + 1ull);
        s1 = l < 1u ? l : 1u;
        s2 = l > 0u ? l - 1u : 0u;
    }
    __device__ __forceinline__ uint32_t div(uint32_t n) const
    {
        const uint32_t t = __umulhi(m, n);
        return (t + ((n - t) >> s1)) >> s2;
    }
};

struct RenderConst {
    ptm::Camera cam;
    float env[3];
    float tmin, tmax;
    uint32_t width, height, tiles_x;
    uint32_t spp, max_depth;
    int32_t frame_base;        // frame index of lane 0 of this batch
    uint32_t lanes_active;     // frames in this batch
    uint32_t slots_per_lane;   // n_tiles * 64
    uint32_t groups;           // sample groups per (frame, pixel): slot lane = frame_lane * groups + group
    uint32_t group_size;       // samples per group: group g runs samples [g*group_size, min(spp, (g+1)*group_size))
    uint32_t term_cap;         // radiance-term log capacity per slot = group_size * max_depth (groups > 1)
    uint32_t term_pcap;        // entries of it kept in the dense primary log (the rest is the overflow log)
    uint32_t n_slots;          // all slots of this render (the primary log is term-major: [term_pcap][n_slots])
    FastDiv div_spl, div_groups;  // slot -> frame lane / sample group without integer divides
};

// Where a slot's radiance goes.  groups == 1: one accumulator per slot, added to in path order
// (raygen.rgen:76).  groups > 1: the samples of a pixel are traced by several slots at once, so
// every slot LOGS its non-zero terms in order and k_resolve replays the logs group by group --
// the same float adds in the same order as the reference's single `color`, still bit-exact.
struct Radiance {
    float4 *color;     // [n_slots]              (groups == 1)
    float4 *terms;     // primary log [term_pcap][n_slots], rgb + pad (groups > 1): neighbouring slots write their
                       // k-th term side by side (slot-major rows measured 20 % slower in k_shade)
    float4 *terms_over;  // overflow log [n_slots][term_cap - term_pcap]: rarely touched; sized to a memory budget
    uint32_t *nterm;   // [n_slots]              (groups > 1)
    // terms beyond a slot's term_cap go to a pool shared by all slots, chained backwards per slot (a slot has one
    // live path, so its chain has one writer): {r, g, b, index of the slot's previous pool entry}
    float4 *spill;
    uint32_t *spill_head;          // [n_slots] last pool entry of the slot, SPILL_NONE if none
    unsigned long long *spill_count;
    uint32_t spill_cap;
    unsigned long long *overflow;  // set when the pool is full too: the host re-renders the batch with groups == 1
};
constexpr uint32_t SPILL_NONE = 0xFFFFFFFFu;
constexpr uint32_t SPILL_POOL_ENTRIES = 4u << 20;  // 64 MB

// (the term logs are read once, by k_resolve at the end of the batch, k_generate's queue by the first extend launch: `nt` stores,
// pt_math.h st_stream -- C4 +2.2 %, C2 +0.5 % / -1 % at K = 16 / 2, i.e. neutral: profiles/r03cp_ab_nt_terms_generate.log)
#ifndef PT_NT_TERMS
#define PT_NT_TERMS true
#endif
#ifndef PT_NT_GEN
#define PT_NT_GEN true
#endif
__device__ __forceinline__ void add_radiance(const RenderConst &rc, const Radiance &rad, uint32_t slot, float r, float g,
                                             float b)
{
    if (rc.groups == 1u) {
        float4 c = rad.color[slot];
        c.x = c.x + r;
        c.y = c.y + g;
        c.z = c.z + b;
        rad.color[slot] = c;
    } else {
        const uint32_t k = rad.nterm[slot];
        if (k < rc.term_pcap) ptm::st_stream<PT_NT_TERMS>(rad.terms + ((size_t)k * rc.n_slots + slot), make_float4(r, g, b, 0.f));
        else if (k < rc.term_cap) ptm::st_stream<PT_NT_TERMS>(rad.terms_over + ((size_t)slot * (rc.term_cap - rc.term_pcap) + (k - rc.term_pcap)), make_float4(r, g, b, 0.f));
        else {
            const unsigned long long idx = atomicAdd(rad.spill_count, 1ull);
            if (idx < rad.spill_cap) {
                rad.spill[idx] = make_float4(r, g, b, __uint_as_float(rad.spill_head[slot]));
                rad.spill_head[slot] = (uint32_t)idx;
            } else {
                *rad.overflow = 1ull;  // this batch's film update is discarded and redone with groups == 1
            }
        }
        rad.nterm[slot] = k + 1u;
    }
}

struct QueueView {
    uint2 *id;  // {slot, sample | depth<<16}
    float4 *state;
    float4 *rayA;
    float2 *rayB;
};

__device__ __forceinline__ void slot_pixel(const RenderConst &rc, const uint32_t *__restrict__ tiles, uint32_t slot,
                                           uint32_t &lane_f, uint32_t &group, uint32_t &px, uint32_t &py)
{
    const uint32_t lane = rc.div_spl.div(slot);
    lane_f = rc.div_groups.div(lane);
    group = lane - lane_f * rc.groups;
    const uint32_t local = slot - lane * rc.slots_per_lane;
    const uint32_t g = tiles[local >> 6];  // tile x | tile y << 16
    px = (g & 0xFFFFu) * 8u + (local & 7u);
    py = (g >> 16) * 8u + ((local >> 3) & 7u);
}

// Block-wide ordered compaction of up to ITEMS x 256 survivors: wave ballots for the in-wave
// rank, LDS for the cross-wave prefix, ONE device-scope atomic per chunk for the queue tail.
template <int ITEMS>
__device__ __forceinline__ void chunk_offsets(const bool (&alive)[ITEMS], uint32_t (&dst)[ITEMS], uint32_t *count_out,
                                              uint32_t (*s_wcnt)[4], uint32_t *s_base)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned long long lt = (1ull << lane) - 1ull;
    uint32_t rank[ITEMS];
#pragma unroll
    for (int it = 0; it < ITEMS; it++) {
        const unsigned long long m = __ballot(alive[it]);
        rank[it] = __popcll(m & lt);
        if (lane == 0) s_wcnt[it][wave] = __popcll(m);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t total = 0;
#pragma unroll
        for (int it = 0; it < ITEMS; it++)
            for (int w = 0; w < 4; w++) total += s_wcnt[it][w];
        *s_base = total ? atomicAdd(count_out, total) : 0u;
    }
    __syncthreads();
    uint32_t run = *s_base;
#pragma unroll
    for (int it = 0; it < ITEMS; it++) {
        for (int w = 0; w < 4; w++) {
            if (w == wave) dst[it] = run + rank[it];
            run += s_wcnt[it][w];
        }
    }
    __syncthreads();  // s_wcnt / s_base are reused by the next chunk
}

// ---- generate: sample 0 of every (frame, pixel) slot of the batch ----------------------------
__global__ __launch_bounds__(TB) void k_generate(RenderConst rc, const uint32_t *__restrict__ tiles, uint32_t slot_base,
                                                 uint32_t n_slots, Radiance rad, QueueView out, uint32_t *count_out)
{
    // four slots per thread and ONE queue-tail atomic per 1024 slots, as in k_shade: with one atomic per 256 slots
    // the 133 M slots of 16 frames x 4 sample groups spent 2.9 ms per launch on the ~88 atomics/us a single word takes
    constexpr int GEN_ITEMS = 4;
    constexpr uint32_t CHUNK = TB * GEN_ITEMS;
    __shared__ uint32_t s_wcnt[GEN_ITEMS][4];
    __shared__ uint32_t s_base;
    for (uint32_t base = blockIdx.x * CHUNK; base < n_slots; base += gridDim.x * CHUNK) {
        bool alive[GEN_ITEMS];
        uint32_t o_slot[GEN_ITEMS], o_seed[GEN_ITEMS], o_sample[GEN_ITEMS];
        ptm::f3 o_org[GEN_ITEMS], o_dir[GEN_ITEMS];
#pragma unroll
        for (int it = 0; it < GEN_ITEMS; it++) {
            const uint32_t local = base + it * TB + threadIdx.x;
            const uint32_t slot = slot_base + local;
            alive[it] = false;
            o_slot[it] = slot; o_seed[it] = 0u; o_sample[it] = 0u; o_org[it] = {}; o_dir[it] = {};
            if (local < n_slots) {
                uint32_t f, g, px, py;
                slot_pixel(rc, tiles, slot, f, g, px, py);
                if (rc.groups == 1u) rad.color[slot] = make_float4(0.f, 0.f, 0.f, 0.f);  // raygen.rgen:42
                else {
                    rad.nterm[slot] = 0u;
                    rad.spill_head[slot] = SPILL_NONE;
                }
                const uint32_t sample0 = g * rc.group_size;
                o_sample[it] = sample0;
                if (px < rc.width && py < rc.height && f < rc.lanes_active && sample0 < rc.spp) {
                    alive[it] = true;
                    o_seed[it] = ptm::make_seed(px, py, sample0, rc.frame_base + (int32_t)f, rc.spp);
                    ptm::primary_ray(rc.cam, px, py, o_seed[it], o_org[it], o_dir[it]);
                }
            }
        }
        uint32_t dst[GEN_ITEMS];
        chunk_offsets<GEN_ITEMS>(alive, dst, count_out, s_wcnt, &s_base);
#pragma unroll
        for (int it = 0; it < GEN_ITEMS; it++) {
            if (alive[it]) {
                ptm::st_stream<PT_NT_GEN>(out.id + dst[it], make_uint2(o_slot[it], o_sample[it]));
                ptm::st_stream<PT_NT_GEN>(out.state + dst[it], make_float4(__uint_as_float(o_seed[it]), 1.f, 1.f, 1.f));  // raygen.rgen:59
                ptm::st_stream<PT_NT_GEN>(out.rayA + dst[it], make_float4(o_org[it].x, o_org[it].y, o_org[it].z, o_dir[it].x));
                ptm::st_stream<PT_NT_GEN>(out.rayB + dst[it], make_float2(o_dir[it].y, o_dir[it].z));
            }
        }
    }
}

// ---- extend, two-level variant (BASELINE config C4: instanced scenes) ----------------------------
// TLAS = BVH4 over the instances' world boxes, BLAS = the scene's BVH4 in object space.  Same
// persistent-thread structure; one stack serves both levels: entering an instance pushes an EXIT
// marker, everything above it belongs to the BLAS walk, popping it restores the world-space ray.
// The ray goes to object space un-normalised (Vulkan semantics: t is the same parameter in both
// spaces), so entry distances and the best hit compare across levels.  Not in the reference
// (one identity instance, main.cpp:515-538); semantics in DESIGN.md section 3.
constexpr uint32_t EXIT_MARK = 0x7FFFFFFFu;

template <bool COUNT, bool LDS_BLAS, bool SHADOW = false>
__global__ __launch_bounds__(TB) void k_extend_inst(const float4 *__restrict__ tlas, const float4 *__restrict__ g_blas,
                                                    const float4 *__restrict__ g_tri4, uint32_t n_blas_wide,
                                                    uint32_t n_tris, const float4 *__restrict__ inst6,
                                                    const uint32_t *__restrict__ inst_id, const float4 *__restrict__ rayA,
                                                    const float2 *__restrict__ rayB, float4 *__restrict__ hit,
                                                    uint32_t *__restrict__ hit_inst, const uint32_t *__restrict__ count_in,
                                                    uint32_t *count_zero, unsigned long long *stats,
                                                    uint2 *__restrict__ spill, uint32_t spill_stride, int refill_min_idle,
                                                    float tmin, float tmax, int raw_hit, const float *__restrict__ ray_tmax)
{
    // SHADOW (the NEE pipeline's shadow rays): a per-ray upper bound instead of tmax, any hit below it ends the walk
    extern __shared__ __attribute__((aligned(16))) char smem[];
    uint2 *stack = reinterpret_cast<uint2 *>(smem);  // [LDS_STACK][TB]
    const float4 *blas = g_blas;
    const float4 *tri4 = g_tri4;
    if (LDS_BLAS) {  // the BLAS is shared by every instance: keep it (and 3 permuted triangle copies) in LDS
        float4 *s_blas = reinterpret_cast<float4 *>(smem + (size_t)LDS_STACK * TB * sizeof(uint2));
        float4 *s_tri = s_blas + LDS_NODE_F4 * (size_t)n_blas_wide;
        for (uint32_t i = threadIdx.x; i < 8 * n_blas_wide; i += TB) s_blas[(i >> 3) * LDS_NODE_F4 + (i & 7u)] = g_blas[i];
        for (uint32_t i = threadIdx.x; i < 3 * n_tris; i += TB) {
            const float4 v = g_tri4[i];
            s_tri[i] = make_float4(v.y, v.z, v.x, v.w);
            s_tri[3 * n_tris + i] = make_float4(v.z, v.x, v.y, v.w);
            s_tri[6 * n_tris + i] = v;
        }
        __syncthreads();
        blas = s_blas;
        tri4 = s_tri;
    }
    const uint32_t n = *count_in;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        if (count_zero) *count_zero = 0u;
        if (stats) atomicAdd(stats, (unsigned long long)n);
    }
    lds_u64 *my_stack = (lds_u64 *)reinterpret_cast<unsigned long long *>(stack) + threadIdx.x;
    unsigned long long *my_spill = reinterpret_cast<unsigned long long *>(spill) + (size_t)blockIdx.x * TB + threadIdx.x;
    const float INF = __builtin_inff();
    const int lane = threadIdx.x & 63;
    const unsigned long long lt = (1ull << lane) - 1ull;

    bool have = false, exhausted = false, in_blas = false;
    uint32_t q = 0;
    ptm::f3 org_w{}, dir_w{}, inv_w{};   // world-space ray
    ptm::f3 inv{}, invf{}, on{}, of{}, orgp{};  // ray of the level being walked (on/of: folded slab origins; orgp: origin permuted to kx,ky,kz)
    uint32_t tri_base = 0;
    uint32_t ax = 0, ay = 0, az = 0;      // 48 where the walked level's direction component is negative
    ptm::RayPre pre{};
    float best_t = tmax, best_V = 0.f, best_W = 0.f, best_det = 1.f;
    uint32_t best_pos = PT_MISS, best_prim = PT_MISS, best_ipos = PT_MISS, best_iid = PT_MISS;
    uint32_t cur = SENTINEL, cur_ipos = 0, cur_iid = 0;
    float cur_t = 0.f;
    int sp = 0;
    unsigned long long c_nodes = 0, c_tris = 0;
    const uint32_t wave_base = (blockIdx.x * (TB / 64) + (threadIdx.x >> 6)) * 64u;
    const uint32_t wave_stride = gridDim.x * TB;
    uint32_t cursor = 0;

    auto push = [&](uint32_t w, float t) {
        const unsigned long long e = stack_entry(w, t);
        if (sp < LDS_STACK) my_stack[sp * TB] = e;
        else my_spill[(size_t)(sp - LDS_STACK) * spill_stride] = e;
        sp++;
    };
    auto pop = [&]() -> uint32_t {
        while (sp > 0) {
            sp--;
            unsigned long long e64;
            if (sp < LDS_STACK) e64 = my_stack[sp * TB];
            else e64 = my_spill[(size_t)(sp - LDS_STACK) * spill_stride];
            const uint2 e = make_uint2((uint32_t)e64, (uint32_t)(e64 >> 32));
            if (e.x == EXIT_MARK) {  // the instance is done: back to the world-space ray and the TLAS
                inv = inv_w;
                slab_setup(org_w, inv_w, invf, on, of);
                ax = inv.x < 0.f ? 48u : 0u;
                ay = inv.y < 0.f ? 48u : 0u;
                az = inv.z < 0.f ? 48u : 0u;
                in_blas = false;
                continue;
            }
            if (__uint_as_float(e.y) <= best_t) {
                cur_t = __uint_as_float(e.y);
                return e.x;
            }
        }
        return SENTINEL;
    };

    for (;;) {
        const unsigned long long idle = __ballot(!have);
        const int n_idle = __popcll(idle);
        if (!exhausted && n_idle >= refill_min_idle) {
            if (!have) {
                const uint32_t v = cursor + (uint32_t)__popcll(idle & lt);
                const uint32_t qq = (v >> 6) * wave_stride + wave_base + (v & 63u);
                if (qq < n) {
                    q = qq;
                    const float4 ra = rayA[q];
                    const float2 rb = rayB[q];
                    org_w = { ra.x, ra.y, ra.z };
                    dir_w = { ra.w, rb.x, rb.y };
                    inv_w = { ptm::safe_inv(dir_w.x), ptm::safe_inv(dir_w.y), ptm::safe_inv(dir_w.z) };
                    inv = inv_w;
                    slab_setup(org_w, inv_w, invf, on, of);
                    ax = inv.x < 0.f ? 48u : 0u;
                    ay = inv.y < 0.f ? 48u : 0u;
                    az = inv.z < 0.f ? 48u : 0u;
                    in_blas = false;
                    best_t = SHADOW ? ray_tmax[q] : tmax; best_V = 0.f; best_W = 0.f; best_det = 1.f;
                    best_pos = PT_MISS; best_prim = PT_MISS; best_ipos = PT_MISS; best_iid = PT_MISS;
                    cur = 0u;  // TLAS root
                    cur_t = tmin;
                    sp = 0;
                    have = true;
                }
            }
            cursor += (uint32_t)n_idle;
            exhausted = (cursor >> 6) * wave_stride + wave_base >= n;
        }
        if (__ballot(have) == 0ull) break;

        // ---- node phase (either level).  (Vote-scheduled single steps as in k_extend measured -4 % here, leaving
        // the node loop early when few lanes still descend +-0.)
        while (have && !(cur & PT_LEAF)) {
            float4 nx, fx, ny, fy, nz, fz, cw;
            if (LDS_BLAS && in_blas) {
                const char *nd = reinterpret_cast<const char *>(blas + LDS_NODE_F4 * (size_t)cur);
                nx = PT_F4(nd + ax); fx = PT_F4(nd - ax + 48); ny = PT_F4(nd + ay + 16); fy = PT_F4(nd - ay + 64);
                nz = PT_F4(nd + az + 32); fz = PT_F4(nd - az + 80); cw = PT_F4(nd + 96);
            } else {
                const char *nd = reinterpret_cast<const char *>((LDS_BLAS ? tlas : (in_blas ? g_blas : tlas)) + 8 * (size_t)cur);
                nx = PT_F4(nd + ax); fx = PT_F4(nd - ax + 48); ny = PT_F4(nd + ay + 16); fy = PT_F4(nd - ay + 64);
                nz = PT_F4(nd + az + 32); fz = PT_F4(nd - az + 80); cw = PT_F4(nd + 96);
            }
            if (COUNT) c_nodes++;
            float t0, t1, t2, t3;
            uint32_t w0 = __float_as_uint(cw.x), w1 = __float_as_uint(cw.y), w2 = __float_as_uint(cw.z),
                     w3 = __float_as_uint(cw.w);
            PT_SLAB4(t0, x)
            PT_SLAB4(t1, y)
            PT_SLAB4(t2, z)
            PT_SLAB4(t3, w)
#define PT_CSWAP(TA, WA, TB_, WB)                            \
    {                                                        \
        const bool sw = TB_ < TA;                            \
        const float ta = sw ? TB_ : TA, tb = sw ? TA : TB_;  \
        const uint32_t wa = sw ? WB : WA, wb = sw ? WA : WB; \
        TA = ta; TB_ = tb; WA = wa; WB = wb;                 \
    }
            PT_CSWAP(t0, w0, t1, w1)
            PT_CSWAP(t2, w2, t3, w3)
            PT_CSWAP(t0, w0, t2, w2)
            PT_CSWAP(t1, w1, t3, w3)
            PT_CSWAP(t1, w1, t2, w2)
#undef PT_CSWAP
            if (t3 < INF) push(w3, t3);
            if (t2 < INF) push(w2, t2);
            if (t1 < INF) push(w1, t1);
            if (t0 < INF) { cur = w0; cur_t = t0; }
            else cur = pop();
        }
        // ---- leaf phase
        // entering an instance costs ~130 VALU instructions (ray transform, three true divides, slab set-up): lanes that
        // want to wait until ENTER_MIN of them do, or until no other lane of the wave has triangle work left
        // (extend -6 %, C4 +1.5 %; 8, 16 and 32 measured alike)
        constexpr int ENTER_MIN = 16;
        const int n_enter = __popcll(__ballot(have && cur != SENTINEL && !in_blas));
        const bool others = __ballot(have && cur != SENTINEL && in_blas) != 0ull;
        const bool do_enter = n_enter >= ENTER_MIN || !others;
        if (have) {
            if (cur != SENTINEL && (in_blas || do_enter)) {
                const uint32_t first = cur & 0x0FFFFFFFu, cnt = ((cur >> 28) & 7u) + 1u;
                if (in_blas) {
                    if (COUNT) c_tris += cnt;
                    for (uint32_t k = 0; k < cnt; k++) {
                        const uint32_t pos = first + k;
                        const size_t ti = LDS_BLAS ? (size_t)tri_base + 3 * (size_t)pos : 3 * (size_t)pos;
                        const float4 a = tri4[ti + 0], b = tri4[ti + 1], c = tri4[ti + 2];
                        float t, V, W, det;
                        const bool th = LDS_BLAS
                            ? ptm::tri_test_perm(pre, orgp, { a.x, a.y, a.z }, { b.x, b.y, b.z }, { c.x, c.y, c.z }, tmin, tmax, t, V, W, det)
                            : ptm::tri_test(pre, { a.x, a.y, a.z }, { b.x, b.y, b.z }, { c.x, c.y, c.z }, tmin, tmax, t, V, W, det);
                        if (th) {
                            const uint32_t prim = __float_as_uint(a.w);
                            // closest t; equal t -> lowest (gl_InstanceID, gl_PrimitiveID)
                            if (t < best_t || (t == best_t && (cur_iid < best_iid || (cur_iid == best_iid && prim < best_prim)))) {
                                best_t = t; best_V = V; best_W = W; best_det = det; best_pos = pos; best_prim = prim;
                                best_ipos = cur_ipos; best_iid = cur_iid;
                                if (SHADOW) sp = 0;  // any hit will do
                            }
                        }
                    }
                    cur = pop();
                } else {
                    // TLAS leaf: up to 4 instances; all but the first go back on the stack as
                    // single-instance leaves, the first is entered now
                    for (uint32_t k = cnt - 1u; k >= 1u; k--) push(PT_LEAF | (first + k), cur_t);
                    cur_ipos = first;
                    cur_iid = inst_id[first];
                    const float4 r0 = inst6[6 * (size_t)first + 3], r1 = inst6[6 * (size_t)first + 4],
                                 r2 = inst6[6 * (size_t)first + 5];
                    const ptm::f3 oo = { ((r0.x * org_w.x + r0.y * org_w.y) + r0.z * org_w.z) + r0.w,
                                         ((r1.x * org_w.x + r1.y * org_w.y) + r1.z * org_w.z) + r1.w,
                                         ((r2.x * org_w.x + r2.y * org_w.y) + r2.z * org_w.z) + r2.w };
                    const ptm::f3 od = { (r0.x * dir_w.x + r0.y * dir_w.y) + r0.z * dir_w.z,
                                         (r1.x * dir_w.x + r1.y * dir_w.y) + r1.z * dir_w.z,
                                         (r2.x * dir_w.x + r2.y * dir_w.y) + r2.z * dir_w.z };
                    inv = { ptm::safe_inv(od.x), ptm::safe_inv(od.y), ptm::safe_inv(od.z) };
                    slab_setup(oo, inv, invf, on, of);
                    ax = inv.x < 0.f ? 48u : 0u;
                    ay = inv.y < 0.f ? 48u : 0u;
                    az = inv.z < 0.f ? 48u : 0u;
                    pre = ptm::ray_setup(oo, od);
                    if (LDS_BLAS) {
                        tri_base = (uint32_t)pre.kz * 3u * n_tris;
                        orgp = { ptm::sel3(pre.kz, oo.y, oo.z, oo.x), ptm::sel3(pre.kz, oo.z, oo.x, oo.y),
                                 ptm::sel3(pre.kz, oo.x, oo.y, oo.z) };
                    }
                    push(EXIT_MARK, 0.f);
                    in_blas = true;
                    cur = 0u;  // BLAS root
                    cur_t = tmin;
                }
            }
            if (cur == SENTINEL) {
                const bool miss = best_pos == PT_MISS;
                // raw_hit (render path): (V, W, det) go out undivided and k_shade takes the two quotients at
                // full lane occupancy; here they would run once per finishing lane group
                hit[q] = raw_hit ? make_float4(__uint_as_float(best_pos), best_V, best_W, best_det)
                                 : make_float4(__uint_as_float(best_pos), miss ? 0.f : best_t,
                                               miss ? 0.f : ptm::fdiv(best_V, best_det), miss ? 0.f : ptm::fdiv(best_W, best_det));
                if (!SHADOW) hit_inst[q] = best_ipos;
                have = false;
            }
        }
    }
    if (COUNT) {
        for (int o = 32; o > 0; o >>= 1) {
            c_nodes += __shfl_xor(c_nodes, o, 64);
            c_tris += __shfl_xor(c_tris, o, 64);
        }
        if (lane == 0 && stats) {
            atomicAdd(stats + 2, c_nodes);
            atomicAdd(stats + 3, c_tris);
        }
    }
}

// ---- extend, flat variant: the whole scene is ONE wide leaf ------------------------------------
// For scenes of a few dozen triangles (the Cornell box has 36) a tree only adds divergence: rays
// of a wave take different branches and the wave pays for the union.  Here every lane tests every
// triangle in the same order, so the triangle stream is wave-uniform: it comes through the scalar
// cache into SGPRs (s_load), there is no stack, no LDS traffic and no divergent control flow
// except the rare "some lane passed the edge test" tail.  Same closest-hit definition, same bits.
__global__ __launch_bounds__(TB) void k_extend_flat(const float4 *__restrict__ tri4, uint32_t n_tris,
                                                    const float4 *__restrict__ rayA, const float2 *__restrict__ rayB,
                                                    float4 *__restrict__ hit, const uint32_t *__restrict__ count_in,
                                                    uint32_t *count_zero, unsigned long long *stats, float tmin,
                                                    float tmax, int raw_hit)
{
    const uint32_t n = *count_in;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        if (count_zero) *count_zero = 0u;
        if (stats) atomicAdd(stats, (unsigned long long)n);
    }
    for (uint32_t base = blockIdx.x * TB; base < n; base += gridDim.x * TB) {
        const uint32_t q = min(base + threadIdx.x, n - 1u);  // tail lanes redo the last ray (same value stored)
        const float4 ra = rayA[q];
        const float2 rb = rayB[q];
        const ptm::RayPre pre = ptm::ray_setup({ ra.x, ra.y, ra.z }, { ra.w, rb.x, rb.y });
        float best_t = tmax, best_V = 0.f, best_W = 0.f, best_det = 1.f;
        uint32_t best_pos = PT_MISS, best_prim = PT_MISS;
        for (uint32_t i = 0; i < n_tris; i++) {
            const float4 a = tri4[3 * i + 0], b = tri4[3 * i + 1], c = tri4[3 * i + 2];  // uniform address
            float t, V, W, det;
            if (ptm::tri_test(pre, { a.x, a.y, a.z }, { b.x, b.y, b.z }, { c.x, c.y, c.z }, tmin, tmax, t, V, W, det)) {
                const uint32_t prim = __float_as_uint(a.w);
                if (t < best_t || (t == best_t && prim < best_prim)) {
                    best_t = t; best_V = V; best_W = W; best_det = det; best_pos = i; best_prim = prim;
                }
            }
        }
        const bool miss = best_pos == PT_MISS;
        hit[q] = raw_hit ? make_float4(__uint_as_float(best_pos), best_V, best_W, best_det)
                         : make_float4(__uint_as_float(best_pos), miss ? 0.f : best_t, miss ? 0.f : ptm::fdiv(best_V, best_det),
                                       miss ? 0.f : ptm::fdiv(best_W, best_det));
    }
}

// ---- next-event estimation (PT_PIPELINE_WAVEFRONT_NEE; NOT the reference's estimator, see include/pt_api.h) ----------
// The third queue: one shadow ray per hit whose light sample faces the surface.  contrib = the radiance the path gains if
// the ray reaches the light: ((weight * brdf) * Ke) * (cos_s |cos_l| / d^2 * total light area), .w = the ray's tmax.
struct ShadowQueue {
    float4 *rayA;     // {org.xyz, dir.x}
    float2 *rayB;     // {dir.y, dir.z}
    float4 *contrib;  // {r, g, b, tmax}
    float *tmax;      // the same tmax as a plain array: what the extend kernels read (ray_tmax)
    uint32_t *slot;
};

// One light sample for the hit at `pos` (normal n, brdf, path weight w); the operations and their order are part of the
// pipeline's definition (the CPU checker of the tests restates them, and the two agree bit for bit).  Returns false when no shadow ray is needed.
__device__ __forceinline__ bool nee_sample(const float4 *__restrict__ lights, uint32_t n_lights, float light_area, uint32_t &seed,
                                           const ptm::f3 pos, const ptm::f3 n, float br, float bg, float bb, float wr, float wg,
                                           float wb, ptm::f3 &wi, float4 &contrib)
{
    const float rl = ptm::rnd(seed), ru = ptm::rnd(seed), rv = ptm::rnd(seed);
    const float pick = rl * light_area;
    // first emitter whose running area exceeds pick (the last one if none does): binary search of the cdf
    uint32_t li = 0, hi_ = n_lights - 1u;
    while (li < hi_) {
        const uint32_t mid = (li + hi_) >> 1;
        if (lights[5 * (size_t)mid].w > pick) hi_ = mid; else li = mid + 1u;
    }
    const float4 A = lights[5 * (size_t)li + 0], B = lights[5 * (size_t)li + 1], C = lights[5 * (size_t)li + 2],
                 N = lights[5 * (size_t)li + 3], Ke = lights[5 * (size_t)li + 4];
    const float su = ptm::fsqrt(ru);
    const float b0 = 1.0f - su, b1 = su * (1.0f - rv), b2 = su * rv;
    const float dx = ((A.x * b0 + B.x * b1) + C.x * b2) - pos.x, dy = ((A.y * b0 + B.y * b1) + C.y * b2) - pos.y,
                dz = ((A.z * b0 + B.z * b1) + C.z * b2) - pos.z;
    const float d2 = (dx * dx + dy * dy) + dz * dz;
    if (!(d2 > 0.0f)) return false;
    const float dist = ptm::fsqrt(d2);
    ptm::div3_dominant(dx, dy, dz, dist, wi.x, wi.y, wi.z);
    const float cs = (wi.x * n.x + wi.y * n.y) + wi.z * n.z;
    const float cl = fabsf((wi.x * N.x + wi.y * N.y) + wi.z * N.z);
    if (!(cs > 0.0f && cl > 0.0f)) return false;
    const float fgeo = ptm::fdiv(cs * cl, d2) * light_area;
    contrib = make_float4(((wr * br) * Ke.x) * fgeo, ((wg * bg) * Ke.y) * fgeo, ((wb * bb) * Ke.z) * fgeo, dist * 0.999f);
    return true;
}

// after the shadow rays were traced: the contributions of those that reached their light
__global__ __launch_bounds__(TB) void k_shadow_add(RenderConst rc, Radiance rad, const float4 *__restrict__ sq_hit,
                                                   const float4 *__restrict__ contrib, const uint32_t *__restrict__ slot,
                                                   const uint32_t *__restrict__ count)
{
    const uint32_t n = *count;
    for (uint32_t i = blockIdx.x * TB + threadIdx.x; i < n; i += gridDim.x * TB) {
        if (__float_as_uint(sq_hit[i].x) != PT_MISS) continue;  // occluded
        const float4 c = contrib[i];
        add_radiance(rc, rad, slot[i], c.x, c.y, c.z);
    }
}

// ---- shade: closesthit / miss + the bounce logic of raygen.rgen:76-83, regeneration, compaction
// k_shade: paths per thread and waves per SIMD asked of the compiler.  Measured (C2 / C5 Mrays/s, same box): 4 x 4 waves
// (105 VGPRs) 22 050 / 2 266; 4 x 5 (96 VGPRs, 14 spilled since the term-log tiers) 22 060 / 2 258; 3 x 5 22 260 / 2 271;
// 3 x 6 22 280 / 2 270; 2 x 7 (72 VGPRs, no spills) 22 630 / 2 281 -- all within the run-to-run noise, so the one without
// spills and with the most waves in flight is used.  One queue-tail atomic per 512 paths.
#ifndef PT_SHADE_ITEMS
#define PT_SHADE_ITEMS 2
#endif
// (round 3, after the instancing template -- 70 VGPRs, no spills at 7 waves: ten interleaved processes each, 7 waves median 26.8
// Grays/s, 6 waves 25.8; 5 waves with PT_SHADE_PRELOAD 25.2: profiles/r03ck_ab_c2_shade_7_vs_6_waves.log, r03cj_*)
#ifndef PT_SHADE_WAVES
#define PT_SHADE_WAVES 7
#endif
// (the instanced instantiation carries the position transform and the table gather on top: 25 spilled registers at 7 waves = C4
// -8 %, 8 at 6 waves = +2 %, none at 5 waves / 96 VGPRs = +4 % over the kernel before: profiles/r03cf_ab_c4_inst_frames.log)
#ifndef PT_SHADE_WAVES_INST
#define PT_SHADE_WAVES_INST 5
#endif
// PT_SHADE_PRELOAD=1 requests every queue record of a chunk before the first is used (one memory round trip per chunk
// instead of one per item).  Alone on the chip (one pipeline) k_shade gets 13 % faster with it at 5 waves (91 VGPRs, no
// spills: 104 -> 90 ms per 16 C2 frames); next to the other pipeline's traversal kernel, which is how it runs, nothing
// changes (three interleaved repetitions, profiles/r02_shade_preload.txt) -- the frame is bound by the VALU work of both
// kernels, not by k_shade's latency -- so the simpler code stays the default.
#ifndef PT_SHADE_PRELOAD
#define PT_SHADE_PRELOAD 0
#endif
// INST: the scene is instanced (position and normal go to world space per hit; the single-level instantiations carry none of that code)
template <int SH_ITEMS, bool LDS_TABLES, bool NEE = false, bool INST = false>
__global__ __launch_bounds__(TB, NEE ? 4 : INST ? PT_SHADE_WAVES_INST : PT_SHADE_WAVES) void k_shade(RenderConst rc, const uint32_t *__restrict__ tiles,
                                              const float4 *__restrict__ g_tri4, const float4 *__restrict__ g_shade4,
                                              uint32_t n_tris,
                                              const float4 *__restrict__ hit, Radiance rad, QueueView in,
                                              QueueView out, const uint32_t *__restrict__ count_in, uint32_t *count_out,
                                              const float4 *__restrict__ inst6, const uint32_t *__restrict__ hit_inst,
                                              const float4 *__restrict__ shade64, const float4 *__restrict__ ke4,
                                              const float4 *__restrict__ lights, uint32_t n_lights, float light_area,
                                              ShadowQueue sq, uint32_t *sq_count, const float4 *__restrict__ g_frame4,
                                              const float4 *__restrict__ inst_frame)
{
    __shared__ uint32_t s_wcnt[SH_ITEMS][4];
    __shared__ uint32_t s_base;
    // Under two pipelines the shade launches run back to back -- their durations add up to the wall clock -- while the VALU-bound
    // traversal kernel of the other pipeline fits in between with slack: the shade waves are the critical chain and get issue
    // priority over the traversal waves they share a SIMD with.  Same-box A/B, six rounds: C2 23.59 -> 24.26 Grays/s (+2.9 %,
    // shade 153 -> 138 ms, extend 130 -> 138 ms per 16 frames), C4 +4.6 %; priority 1: none, 2: +1.7 %.  With the tables in HBM
    // (C5 +0.6 %, C5x -0.7 %) the kernel waits for its gathers and keeps the default (profiles/r02i_ab_shade_prio.log).
#ifndef PT_SHADE_PRIO
#define PT_SHADE_PRIO 3
#endif
    if (LDS_TABLES && PT_SHADE_PRIO > 0) __builtin_amdgcn_s_setprio(PT_SHADE_PRIO);
#ifndef PT_SHADE_DENSE_REGEN
#define PT_SHADE_DENSE_REGEN 1
#endif
    // (small scenes only: with the tables in HBM the kernel waits for its gathers, and the extra LDS round trip cost C5 1 %)
    constexpr bool DENSE_REGEN = LDS_TABLES && PT_SHADE_DENSE_REGEN != 0;
    // per wave: one 16-B cell per ended path -- first its job {slot, next sample}, then, written by the lane that took the
    // job, the result {dir.xyz, seed} (dir.x = 2: the slot has no further sample)
    __shared__ float4 s_regen[DENSE_REGEN ? 4 : 1][DENSE_REGEN ? 64 * SH_ITEMS : 1];
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const float4 *tri4 = g_tri4;
    const float4 *shade4 = g_shade4;
    const float4 *frame4 = g_frame4;
    if (LDS_TABLES) {  // small scenes: the per-triangle tables live in LDS, no dependent global gathers
        float4 *s_tri = reinterpret_cast<float4 *>(smem);
        float4 *s_shade = s_tri + 3 * (size_t)n_tris;
        float4 *s_frame = s_shade + 3 * (size_t)n_tris;
        for (uint32_t i = threadIdx.x; i < 3 * n_tris; i += TB) {
            s_tri[i] = g_tri4[i];
            s_shade[i] = g_shade4[i];
        }
        for (uint32_t i = threadIdx.x; i < 2 * n_tris; i += TB) s_frame[i] = g_frame4[i];
        __syncthreads();
        tri4 = s_tri;
        shade4 = s_shade;
        frame4 = s_frame;
    }
    const uint32_t n = *count_in;
    constexpr uint32_t CHUNK = TB * SH_ITEMS;
    for (uint32_t base = blockIdx.x * CHUNK; base < n; base += gridDim.x * CHUNK) {
        bool alive[SH_ITEMS];
        uint32_t o_slot[SH_ITEMS], o_ctr[SH_ITEMS];
        float4 o_state[SH_ITEMS], o_rayA[SH_ITEMS];
        float2 o_rayB[SH_ITEMS];
        bool regen[SH_ITEMS];            // DENSE_REGEN: the path ended, the slot's next sample has to be started
        bool s_alive[SH_ITEMS];          // NEE: a shadow ray for this item
        float4 s_rayA[SH_ITEMS], s_contrib[SH_ITEMS];
        float2 s_rayB[SH_ITEMS];
#if PT_SHADE_PRELOAD
        // every queue record of the chunk is requested before the first one is used: the per-item bodies below store and
        // add to the radiance arrays, which the compiler must assume alias the queue, so without this each item's three
        // loads wait behind the previous item's stores -- one memory round trip per item instead of one per chunk
        uint2 in_id[SH_ITEMS];
        float4 in_st[SH_ITEMS], in_hit[SH_ITEMS];
#pragma unroll
        for (int it = 0; it < SH_ITEMS; it++) {
            const uint32_t q = min(base + it * TB + threadIdx.x, n - 1u);
            in_id[it] = in.id[q];
            in_st[it] = in.state[q];
            in_hit[it] = hit[q];
        }
#endif
#pragma unroll
        for (int it = 0; it < SH_ITEMS; it++) {
            const uint32_t q = base + it * TB + threadIdx.x;
            alive[it] = false;
            regen[it] = false;
            if (NEE) { s_alive[it] = false; o_slot[it] = 0u; }
            if (q >= n) continue;
#if PT_SHADE_PRELOAD
            const uint2 id = in_id[it];
            const float4 st = in_st[it];
            const float4 h = in_hit[it];
#else
            const uint2 id = ptm::ld_stream<INST>(in.id + q);
            const float4 st = ptm::ld_stream<INST>(in.state + q);
            const float4 h = ptm::ld_stream<INST>(hit + q);
#endif
            const uint32_t slot = id.x, ctr = id.y;
            uint32_t sample = ctr & 0xFFFFu, depth = ctr >> 16;
            uint32_t seed = __float_as_uint(st.x);
            float wr = st.y, wg = st.z, wb = st.w;
            const uint32_t pos = __float_as_uint(h.x);
            bool terminated;
            ptm::f3 org{}, dir{};
            if (pos == PT_MISS) {
                // miss.rmiss:10-11 then raygen.rgen:76: color += weight * (0.7,0.6,0.5); break
                add_radiance(rc, rad, slot, wr * rc.env[0], wg * rc.env[1], wb * rc.env[2]);
                terminated = true;
            } else {
                // per-triangle record.  Tables in LDS: {n, brdf.r} {brdf.gb, Ke.rg} {Ke.b} + the three vertices.
                // Tables in HBM: every 16-B load of a wave whose lanes hit different triangles is one L1 look-up
                // per lane, so the record is regrouped (k_pack) into {v0, n.x} {v1, n.y} {v2, n.z} {brdf, emits}
                // + Ke apart: 4 look-ups per hit instead of 6, 1 instead of 3 when the path ends here.
                float4 s0, s1, s2, a{}, b{}, c{};
                if (LDS_TABLES) {
                    s0 = shade4[3 * pos + 0]; s1 = shade4[3 * pos + 1]; s2 = shade4[3 * pos + 2];
                } else {
                    const float4 r3 = shade64[4 * (size_t)pos + 3];
                    const float4 ke = r3.w != 0.f ? ke4[pos] : make_float4(0.f, 0.f, 0.f, 0.f);
                    s0 = make_float4(0.f, 0.f, 0.f, r3.x); s1 = make_float4(r3.y, r3.z, ke.x, ke.y); s2 = make_float4(ke.z, 0.f, 0.f, 0.f);
                }
                // raygen.rgen:76: color += weight * emission.  Adding +0 changes no bit of a
                // non-negative accumulator, so the read-modify-write is skipped for non-emitters
                // (NaN compares false and still takes the add).
                const float er = wr * s1.z, eg = wg * s1.w, eb = wb * s2.x;
                // (NEE: the emitters are sampled explicitly, so running into one counts for camera rays only)
                if ((!NEE || depth == 0u) && !(er == 0.f && eg == 0.f && eb == 0.f)) add_radiance(rc, rad, slot, er, eg, eb);
                depth++;
                terminated = depth >= rc.max_depth;  // raygen.rgen:62 loop bound
                // (NEE samples no light at the path's last hit: that sample stands for the emission the next ray would find,
                // and the reference's sum ends with the hit of ray max_depth - 1, raygen.rgen:62-83)
                if (!terminated) {
                    if (LDS_TABLES) {
                        a = tri4[3 * pos + 0]; b = tri4[3 * pos + 1]; c = tri4[3 * pos + 2];
                    } else {
                        a = shade64[4 * (size_t)pos + 0]; b = shade64[4 * (size_t)pos + 1]; c = shade64[4 * (size_t)pos + 2];
                        s0.x = a.w; s0.y = b.w; s0.z = c.w;
                    }
                    // closesthit.rchit:56-57: position from barycentrics, (v0*b0 + v1*b1) + v2*b2
                    // the hit record carries (V, W, det) of the watertight test; attribs = (V/det, W/det)
                    float hu, hv;  // (0 <= V/det, W/det <= 1: ptm::div2_dominant's exact short division)
                    ptm::div2_dominant(h.y, h.z, h.w, hu, hv);
                    const float b0 = (1.0f - hu) - hv;
                    org = { (a.x * b0 + b.x * hu) + c.x * hv, (a.y * b0 + b.y * hu) + c.y * hv,
                            (a.z * b0 + b.z * hu) + c.z * hv };
                    ptm::f3 nrm = { s0.x, s0.y, s0.z };
                    ptm::f3 tng{};  // instanced scenes with the (instance, triangle) table: the tangent of createCoordinateSystem
                    if (INST) {
                        // instanced scene: position by the object->world matrix, normal by the inverse
                        // transpose, renormalised (the reference's closesthit has one identity instance)
                        const uint32_t ip = hit_inst[q];
                        const float4 m0 = inst6[6 * (size_t)ip + 0], m1 = inst6[6 * (size_t)ip + 1], m2 = inst6[6 * (size_t)ip + 2];
                        const ptm::f3 pw = { ((m0.x * org.x + m0.y * org.y) + m0.z * org.z) + m0.w,
                                             ((m1.x * org.x + m1.y * org.y) + m1.z * org.z) + m1.w,
                                             ((m2.x * org.x + m2.y * org.y) + m2.z * org.z) + m2.w };
                        org = pw;
                        if (inst_frame) {
                            // ... both evaluated once per (instance, triangle) with these very operations (lbvh_build.hip
                            // k_inst_frames): a 32-B gather instead of two square roots and five true divides per hit
                            const size_t e = 2 * ((size_t)ip * n_tris + pos);
                            const float4 f0 = inst_frame[e], f1 = inst_frame[e + 1];
                            nrm = { f0.x, f0.y, f0.z };
                            tng = { f0.w, f1.x, f1.y };
                        } else {
                            const float4 i0 = inst6[6 * (size_t)ip + 3], i1 = inst6[6 * (size_t)ip + 4], i2 = inst6[6 * (size_t)ip + 5];
                            const float nx = (i0.x * nrm.x + i1.x * nrm.y) + i2.x * nrm.z;
                            const float ny = (i0.y * nrm.x + i1.y * nrm.y) + i2.y * nrm.z;
                            const float nz = (i0.z * nrm.x + i1.z * nrm.y) + i2.z * nrm.z;
                            const float l = ptm::fsqrt((nx * nx + ny * ny) + nz * nz);
                            nrm = { ptm::fdiv(nx, l), ptm::fdiv(ny, l), ptm::fdiv(nz, l) };
                        }
                    }
                    if (NEE && n_lights) {  // one light sample -> shadow queue (three random numbers, drawn before the bounce's)
                        ptm::f3 wi;
                        float4 cb;
                        if (nee_sample(lights, n_lights, light_area, seed, org, nrm, s0.w, s1.x, s1.y, wr, wg, wb, wi, cb)) {
                            s_alive[it] = true;
                            s_rayA[it] = make_float4(org.x, org.y, org.z, wi.x);
                            s_rayB[it] = make_float2(wi.y, wi.z);
                            s_contrib[it] = cb;
                        }
                    }
                    const float r1 = ptm::rnd(seed);  // cos(theta) first, azimuth second
                    const float r2 = ptm::rnd(seed);
                    if (LDS_TABLES && !INST) {  // the triangle's tangent frame was evaluated once, by k_pack, with the same operations
                        const float4 f0 = frame4[2 * pos + 0], f1 = frame4[2 * pos + 1];
                        dir = ptm::sample_direction_frame(r1, r2, nrm, { f0.x, f0.y, f0.z }, { f0.w, f1.x, f1.y });
                    } else if (INST && inst_frame) {  // bitangent = the cross product of tangent_frame, same operands
                        const ptm::f3 btg = { nrm.y * tng.z - nrm.z * tng.y, nrm.z * tng.x - nrm.x * tng.z, nrm.x * tng.y - nrm.y * tng.x };
                        dir = ptm::sample_direction_frame(r1, r2, nrm, tng, btg);
                    } else {
                        dir = ptm::sample_direction(r1, r2, nrm);  // raygen.rgen:78
                    }
                    const float dt = (dir.x * nrm.x + dir.y * nrm.y) + dir.z * nrm.z;
                    // raygen.rgen:79-80: weight *= brdf * dot / pdf, pdf = 1/(2*pi) as a true divide
                    float fr = s0.w * dt, fg = s1.x * dt, fb = s1.y * dt;
                    ptm::div3_by_pdf(fr, fg, fb);
                    wr = wr * fr;
                    wg = wg * fg;
                    wb = wb * fb;
                }
            }
            if (terminated) {
                sample++;
                if (DENSE_REGEN) {
                    regen[it] = true;  // the next sample's primary ray is built after the item loop, by densely packed lanes
                } else {
                    uint32_t f, g, px, py;
                    slot_pixel(rc, tiles, slot, f, g, px, py);
                    if (sample < min(rc.spp, (g + 1u) * rc.group_size)) {  // next sample of this slot: raygen.rgen:45-60
                        seed = ptm::make_seed(px, py, sample, rc.frame_base + (int32_t)f, rc.spp);
                        ptm::primary_ray(rc.cam, px, py, seed, org, dir);
                        wr = wg = wb = 1.0f;
                        depth = 0;
                        alive[it] = true;
                    }
                }
            } else {
                alive[it] = true;
            }
            o_slot[it] = slot;
            o_ctr[it] = sample | (depth << 16);
            o_state[it] = make_float4(__uint_as_float(seed), wr, wg, wb);
            o_rayA[it] = make_float4(org.x, org.y, org.z, dir.x);
            o_rayB[it] = make_float2(dir.y, dir.z);
        }
        if (DENSE_REGEN) {
            // Regeneration (raygen.rgen:45-60 for the slot's next sample: pixel of the slot, seed, jitter, camera ray -- five
            // true divides and a square root) used to sit in the per-item branch above, which a wave enters whenever ANY of
            // its lanes ended a path, i.e. always, at ~30 % lane occupancy.  Here the ended paths of all SH_ITEMS items of a
            // wave are handed, through a wave-private piece of LDS, to consecutive lanes: one pass (two when more than 64
            // ended) at 60 % occupancy instead of SH_ITEMS passes at 30 %.  No block barrier: a wave's LDS operations
            // execute in order.  Same operations per path, same bits.
            const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
            const unsigned long long lt = (1ull << lane) - 1ull;
            float4 *cell = s_regen[wave];
            uint32_t rank[SH_ITEMS], total = 0;
#pragma unroll
            for (int it = 0; it < SH_ITEMS; it++) {
                const unsigned long long m = __ballot(regen[it]);
                rank[it] = total + (uint32_t)__popcll(m & lt);
                total += (uint32_t)__popcll(m);
                if (regen[it]) cell[rank[it]] = make_float4(__uint_as_float(o_slot[it]), __uint_as_float(o_ctr[it] & 0xFFFFu), 0.f, 0.f);
            }
            __builtin_amdgcn_wave_barrier();
            for (uint32_t j = (uint32_t)lane; j < total; j += 64u) {
                const float4 jc = cell[j];
                const uint2 job = make_uint2(__float_as_uint(jc.x), __float_as_uint(jc.y));
                uint32_t f, g, px, py;
                slot_pixel(rc, tiles, job.x, f, g, px, py);
                float4 r = make_float4(2.0f, 0.f, 0.f, 0.f);
                if (job.y < min(rc.spp, (g + 1u) * rc.group_size)) {
                    uint32_t seed = ptm::make_seed(px, py, job.y, rc.frame_base + (int32_t)f, rc.spp);
                    ptm::f3 org, dir;
                    ptm::primary_ray(rc.cam, px, py, seed, org, dir);
                    r = make_float4(dir.x, dir.y, dir.z, __uint_as_float(seed));
                }
                cell[j] = r;
            }
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int it = 0; it < SH_ITEMS; it++) {
                const float4 r = regen[it] ? cell[rank[it]] : make_float4(2.0f, 0.f, 0.f, 0.f);
                if (r.x != 2.0f) {
                    alive[it] = true;
                    o_ctr[it] = o_ctr[it] & 0xFFFFu;  // depth 0
                    o_state[it] = make_float4(r.w, 1.0f, 1.0f, 1.0f);  // raygen.rgen:59
                    o_rayA[it] = make_float4(rc.cam.ox, rc.cam.oy, rc.cam.oz, r.x);
                    o_rayB[it] = make_float2(r.y, r.z);
                }
            }
            __builtin_amdgcn_wave_barrier();  // the area is reused by the next chunk
        }
        uint32_t dst[SH_ITEMS];
        if (NEE) {  // the shadow queue, compacted like the path queue (its entries outlive this path's regeneration: own slot copy)
            uint32_t sdst[SH_ITEMS];
            chunk_offsets<SH_ITEMS>(s_alive, sdst, sq_count, s_wcnt, &s_base);
#pragma unroll
            for (int it = 0; it < SH_ITEMS; it++) {
                if (s_alive[it]) {
                    sq.rayA[sdst[it]] = s_rayA[it];
                    sq.rayB[sdst[it]] = s_rayB[it];
                    sq.contrib[sdst[it]] = s_contrib[it];
                    sq.tmax[sdst[it]] = s_contrib[it].w;
                    sq.slot[sdst[it]] = o_slot[it];
                }
            }
        }
        chunk_offsets<SH_ITEMS>(alive, dst, count_out, s_wcnt, &s_base);
#pragma unroll
        for (int it = 0; it < SH_ITEMS; it++) {
            if (alive[it]) {
                ptm::st_stream<true>(out.id + dst[it], make_uint2(o_slot[it], o_ctr[it]));
                ptm::st_stream<true>(out.state + dst[it], o_state[it]);
                ptm::st_stream<true>(out.rayA + dst[it], o_rayA[it]);
                ptm::st_stream<true>(out.rayB + dst[it], o_rayB[it]);
            }
        }
    }
}

// ---- resolve: raygen.rgen:86-90 for every frame of the batch, in frame order -------------------
__device__ __forceinline__ uint8_t to_unorm8(float c)
{
    if (!(c > 0.0f)) return 0;
    if (c > 1.0f) c = 1.0f;
    return (uint8_t)(c * 255.0f + 0.5f);
}

__global__ __launch_bounds__(TB) void k_resolve(RenderConst rc, const uint32_t *__restrict__ tiles, Radiance rad,
                                                float *__restrict__ film, uint8_t *__restrict__ bgra)
{
    const uint32_t local = blockIdx.x * TB + threadIdx.x;
    if (local >= rc.slots_per_lane) return;
    uint32_t f0, g0, px, py;
    slot_pixel(rc, tiles, local, f0, g0, px, py);
    if (px >= rc.width || py >= rc.height) return;
    const size_t pix = (size_t)py * rc.width + px;
    float fr = film[3 * pix + 0], fg = film[3 * pix + 1], fb = film[3 * pix + 2];
    uchar4 img = reinterpret_cast<uchar4 *>(bgra)[pix];  // bytes B,G,R,A
    const float spp = (float)rc.spp;
    for (uint32_t f = 0; f < rc.lanes_active; f++) {
        float4 c = make_float4(0.f, 0.f, 0.f, 0.f);
        if (rc.groups == 1u) {
            c = rad.color[(size_t)f * rc.slots_per_lane + local];
        } else {  // replay the groups' term logs in sample order: the reference's sequence of adds
            for (uint32_t g = 0; g < rc.groups; g++) {
                const size_t slot = ((size_t)f * rc.groups + g) * rc.slots_per_lane + local;
                const uint32_t nt_all = rad.nterm[slot], nt = min(nt_all, rc.term_cap);
                const float4 *to = rad.terms_over + slot * (rc.term_cap - rc.term_pcap);
                for (uint32_t k = 0; k < nt; k++) {
                    const float4 e = k < rc.term_pcap ? rad.terms[(size_t)k * rc.n_slots + slot] : to[k - rc.term_pcap];
                    c.x = c.x + e.x;
                    c.y = c.y + e.y;
                    c.z = c.z + e.z;
                }
                // the few slots with more terms: their pool entries are chained newest-first, so the j-th one in
                // path order is reached by walking m-1-j links (m is small; quadratic in m, rare).  (When the pool
                // itself overflowed the chain is incomplete: the host discards this batch.)
                const uint32_t m = nt_all - nt;
                for (uint32_t j = 0; j < m; j++) {
                    uint32_t idx = rad.spill_head[slot];
                    for (uint32_t w = j + 1; w < m && idx != SPILL_NONE; w++) idx = __float_as_uint(rad.spill[idx].w);
                    if (idx == SPILL_NONE) break;
                    const float4 e = rad.spill[idx];
                    c.x = c.x + e.x;
                    c.y = c.y + e.y;
                    c.z = c.z + e.z;
                }
            }
        }
        const float cr = ptm::fdiv(c.x, spp), cg = ptm::fdiv(c.y, spp), cb = ptm::fdiv(c.z, spp);  // :86
        const int32_t frame = rc.frame_base + (int32_t)f;
        const float ff = (float)frame, f1 = (float)(frame + 1);
        const bool first = frame == 0;  // old * 0: never read the uninitialised image
        // float film (canonical): new = (color + old*frame) / (frame+1)
        fr = ptm::fdiv(cr + (first ? 0.f : fr) * ff, f1);
        fg = ptm::fdiv(cg + (first ? 0.f : fg) * ff, f1);
        fb = ptm::fdiv(cb + (first ? 0.f : fb) * ff, f1);
        // reference display image: rgba8 load -> blend -> clamp + quantise on store
        const float orr = first ? 0.f : ptm::fdiv((float)img.z, 255.0f);
        const float og = first ? 0.f : ptm::fdiv((float)img.y, 255.0f);
        const float ob = first ? 0.f : ptm::fdiv((float)img.x, 255.0f);
        const float oa = first ? 0.f : ptm::fdiv((float)img.w, 255.0f);
        img.z = to_unorm8(ptm::fdiv(cr + orr * ff, f1));
        img.y = to_unorm8(ptm::fdiv(cg + og * ff, f1));
        img.x = to_unorm8(ptm::fdiv(cb + ob * ff, f1));
        img.w = to_unorm8(ptm::fdiv(1.0f + oa * ff, f1));
    }
    film[3 * pix + 0] = fr;
    film[3 * pix + 1] = fg;
    film[3 * pix + 2] = fb;
    reinterpret_cast<uchar4 *>(bgra)[pix] = img;
}

#include "fused_kernel.h"  // k_fused: PT_PIPELINE_FUSED, the whole loop as one persistent kernel (scenes in LDS)

// hit records of the internal layout (sorted position) -> API layout (gl_PrimitiveID)
__global__ __launch_bounds__(TB) void k_hits_to_api(const float4 *__restrict__ hit, const float4 *__restrict__ tri4,
                                                    const uint32_t *__restrict__ hit_inst,
                                                    const uint32_t *__restrict__ inst_id, uint32_t n,
                                                    pt_hit *__restrict__ out)
{
    const uint32_t i = blockIdx.x * TB + threadIdx.x;
    if (i >= n) return;
    const float4 h = hit[i];
    const uint32_t pos = __float_as_uint(h.x);
    pt_hit o;
    o.prim = pos == PT_MISS ? PT_MISS : __float_as_uint(tri4[3 * (size_t)pos].w);
    o.t = h.y; o.u = h.z; o.v = h.w;
    o.inst = pos == PT_MISS ? PT_MISS : (hit_inst ? inst_id[hit_inst[i]] : 0u);
    out[i] = o;
}

// ---- host side ------------------------------------------------------------------------------
struct ExtendPlan {
    uint32_t variant = PT_EXTEND_LDS;  // PT_EXTEND_FLAT / _LDS / _HBM
    bool lds_scene = false;
    size_t smem = 0;
    int grid = 0;
    uint32_t spill_levels = 0;
    int refill = REFILL_MIN_IDLE;
    int lds_stack = LDS_STACK;  // stack entries per lane kept in LDS (single-level kernel)
    bool spill = true;          // false: the scene's exact stack bound fits lds_stack, kernel without spill path
                                // (and with one-dword stack entries: COMPACT in k_extend)
    bool pairs = false;         // ... and every leaf of the BVH4 is one triangle or one fan pair: the PAIRS kernel
    bool waves7 = false;        // 8-wide kernel: the 72-VGPR instantiation, 7 blocks per CU (scenes beyond the Infinity Cache)
    bool bvh8 = false;          // PT_EXTEND_HBM8: the BVH8 and ITS triangle order (s->d_tri4_8, d_shade64_8, d_ke4_8)
    bool topdown4 = false;      // HBM variant over the top-down BVH4 with contiguous children (s->d_wide16t)
    uint32_t n_tlas_lds = 0;    // k_extend_inst16: TLAS nodes staged in LDS (its top levels)
    bool inst16 = false;        // two-level scenes: k_extend_inst16 (64-B fp16 nodes on both levels, one-dword stack entries)
    size_t smem_inst_fallback = 0; int grid_inst_fallback = 0;  // k_extend_inst's launch shape (tmin <= 0 takes it)
    size_t smem_wide_entries = 0;  // LDS bytes of the same plan run by the 8-byte-entry kernel (negative tmin)
};

pt_status plan_extend(pt_scene *s, uint32_t want, ExtendPlan &pl)
{
    pt_ctx *ctx = s->ctx;
    if (want > PT_EXTEND_HBM8) { ctx->err = "unknown extend variant"; return PT_ERR_INVALID_ARG; }
    if (s->broken) {  // an earlier rebuild of the tree ran out of memory (lbvh_build.hip): never launch on null tables
        const pt_status rcb = ptb_repair(s);
        if (rcb != PT_OK) return rcb;
    }
    // AUTO walks scenes beyond L2 (at first; now nearly every scene beyond LDS, below) through the 8-wide tree (64-B nodes with byte planes): fewer distinct lines per ray -- measured on
    // MI355X, same box, three rounds: C5 2 465 -> 2 547 Mrays/s (+3.4 %), C5x 2 405 -> 2 546 (+5.9 %), 36.2 -> 27.7 and 29.5 -> 24.2
    // node visits per ray (profiles/r03_ab_c5_c5x_hbm8_64B_nodes.log); pt_tuning.hbm8 = 0 keeps the BVH4, 1 takes the 8-wide tree
    // for every scene that does not fit LDS
    const uint64_t ws4 = 64ull * (s->n_wide16t ? s->n_wide16t : s->n_wide) + 64ull * s->n_tris;
    // (round 3, last session: with the 8-wide kernel's new vote, refill threshold and spill-free instantiation the crossover fell from
    // 32 MiB of BVH4 nodes + records to ~1 MiB, i.e. ~11 000 triangles -- soups of 2 500 / 5 000 / 12 000 / 20 000 / 50 000 / 100 000 /
    // 200 000 / 400 000 triangles, 8-wide against BVH4 kernel: -4 / -1.7 / +1.7 / +3.8 / +11 / +11 / +18 / +21 %,
    // profiles/r03ca_bvh4_vs_8wide_midsize.log, r03cb_bvh4_vs_8wide_small.log)
    const bool auto8_big = want == PT_EXTEND_AUTO && ctx->tune.hbm8 != 0 && ws4 > (1ull << 20) && s->n_tris > PT_SAH_MAX_TRIS;
    if ((want == PT_EXTEND_HBM8 || (want == PT_EXTEND_AUTO && ctx->tune.hbm8 == 1) || auto8_big) && !s->n_inst && !s->d_wide8) {
        const pt_status rc8 = ptb_ensure_wide8(s);   // built on first request (260 B per triangle nobody else needs)
        if (rc8 != PT_OK) return rc8;
    }
    if (want == PT_EXTEND_HBM8 && (s->n_inst || !s->d_wide8)) { ctx->err = "no 8-wide nodes for this scene (instanced, or <= 2048 triangles)"; return PT_ERR_UNSUPPORTED; }
    if (s->n_inst) {  // two-level scenes: one kernel variant (BVH4s read through L1/L2)
        if (want == PT_EXTEND_FLAT || want == PT_EXTEND_LDS) { ctx->err = "instanced scenes only have the HBM extend variant"; return PT_ERR_UNSUPPORTED; }
        pl.variant = PT_EXTEND_HBM;
        const size_t blas_bytes = 16 * LDS_NODE_F4 * (size_t)s->n_wide + sizeof(float4) * 9 * (size_t)s->n_tris;
        pl.lds_scene = blas_bytes <= 24 * 1024;  // here: the BLAS (shared by all instances) is staged in LDS
        pl.smem = (size_t)LDS_STACK * TB * sizeof(uint2) + (pl.lds_scene ? blas_bytes : 0);
        int per_cu_i = 0;
        PT_HIP(ctx, hipOccupancyMaxActiveBlocksPerMultiprocessor(
                        &per_cu_i, pl.lds_scene ? reinterpret_cast<const void *>(k_extend_inst<false, true>)
                                                : reinterpret_cast<const void *>(k_extend_inst<false, false>), TB, pl.smem));
        per_cu_i = std::max(1, std::min(per_cu_i, 8));
        pl.refill = 64;  // a wave takes new rays only when all its lanes are done: entering an instance (ray transform, three
                         // divides) and the TLAS root are too expensive to run for a few refilled lanes.  C4: 16: 8.2, 32: 8.85,
                         // 48: 9.26, 56: 9.2, 64: 9.38 Grays/s
        pl.refill = pt_tuned(ctx->tune.refill, pl.refill, 1, 64);
        pl.grid = ctx->num_cus * per_cu_i;
        pl.smem_inst_fallback = pl.smem; pl.grid_inst_fallback = pl.grid;
        // the round-2 kernel when both levels fit its 15-bit child codes and the BLAS fits LDS
        const size_t smem16_scene = sizeof(uint32_t) * I16_NODE_DW * (size_t)s->n_wide + sizeof(float4) * 9 * (size_t)s->n_tris;
        const int lds16 = pt_tuned(ctx->tune.lds_stack, 16, 1, 32);
        pl.inst16 = s->d_tlas16 && s->d_wide16 && s->n_inst < 32768u && s->n_tlas16 < 32767u && s->n_wide < 32767u && s->n_tris <= 2047u &&
                    smem16_scene <= 24 * 1024 && ctx->tune.inst16 != 0;
        if (pl.inst16) {
            pl.lds_stack = lds16;
            // top levels of the TLAS staged in LDS next to the BLAS.  8 KB (102 nodes: the top four levels) measured best on
            // C4: 0 / 4 / 8 / 16 / 24 KB -> 11.95 / 12.16 / 12.31 / 10.7 / 11.1 Grays/s (profiles/r02i_ab_c4_tlas_lds.log) --
            // from 16 KB on the four resident blocks leave the other pipeline's k_shade no LDS to run beside them
            const size_t tlas_lds_bytes = (size_t)pt_tuned(ctx->tune.tlas_lds_kb, 8, 0, 96) * 1024;
            pl.n_tlas_lds = (uint32_t)std::min<size_t>(s->n_tlas16, tlas_lds_bytes / (sizeof(uint32_t) * I16_NODE_DW));
            pl.smem = (size_t)lds16 * TB * sizeof(uint32_t) + smem16_scene + sizeof(uint32_t) * I16_NODE_DW * (size_t)pl.n_tlas_lds;
            const void *fn16 = s->pair_leaves ? reinterpret_cast<const void *>(k_extend_inst16<false, true>)
                                              : reinterpret_cast<const void *>(k_extend_inst16<false, false>);
            if (pl.smem > 48 * 1024)
                for (const void *f : { reinterpret_cast<const void *>(k_extend_inst16<false, true>), reinterpret_cast<const void *>(k_extend_inst16<false, false>),
                                       reinterpret_cast<const void *>(k_extend_inst16<true, true>), reinterpret_cast<const void *>(k_extend_inst16<true, false>),
                                       // (the shadow-ray twins of the NEE pipeline: the same launch shape)
                                       reinterpret_cast<const void *>(k_extend_inst16<false, true, true>), reinterpret_cast<const void *>(k_extend_inst16<false, false, true>) })
                    PT_HIP(ctx, hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)pl.smem));
            int per16 = 0;
            PT_HIP(ctx, hipOccupancyMaxActiveBlocksPerMultiprocessor(&per16, fn16, TB, pl.smem));
            // four blocks per CU and refill at 48 idle lanes measured best on the 10 000-instance grid (C4: 4/48 10.73,
            // 5/48 10.33, 4/40 10.66, 4/56 10.35, 4/64 9.74 Grays/s; the fp32 kernel at its best, 4/64: 9.33)
            per16 = pt_tuned(ctx->tune.inst16_blocks, std::min(per16, 4), 1, 8);
            pl.grid = ctx->num_cus * std::max(1, std::min(per16, 8));
            pl.refill = pt_tuned(ctx->tune.refill, 48, 1, 64);
        }
        // TLAS pushes <= 3 per level + 3 extra instances of a leaf, + EXIT, + the BLAS walk: the exact bound of the
        // BVH4 that is TRAVERSED when the builder gave one (the surface-area BVH4 of a small scene can be deeper
        // than the balanced LBVH whose height s->height is), else 3 per level of the collapsed LBVH
        const uint32_t blas_bound = s->stack_need != 0xFFFFFFFFu ? s->stack_need + 1u : 3u * (s->height_tree / 2u + 1u);
        const uint32_t bound_i = 3u * (std::max(s->tlas_height / 2u + 1u, s->tlas16_levels)) + 4u + blas_bound + 2u;
        pl.spill_levels = bound_i > (uint32_t)LDS_STACK ? bound_i - (uint32_t)LDS_STACK : 0u;  // (sized for the 8-entry fallback kernel)
        const size_t need_i = PT_MAX_PIPES * (size_t)std::max(pl.spill_levels, 1u) * (size_t)std::max(pl.grid, pl.grid_inst_fallback) * TB * sizeof(uint2);
        if (need_i > ctx->spill_bytes) {
            (void)hipFree(ctx->d_spill);
            ctx->d_spill = nullptr;
            ctx->spill_bytes = 0;
            PT_HIP(ctx, hipMalloc((void **)&ctx->d_spill, need_i));
            ctx->spill_bytes = need_i;
        }
        return PT_OK;
    }
    if (want == PT_EXTEND_FLAT && s->n_tris > 1024) { ctx->err = "flat extend variant needs <= 1024 triangles"; return PT_ERR_UNSUPPORTED; }
    if (want == PT_EXTEND_FLAT) {  // never chosen by AUTO: the LDS BVH4 with lane refill measured faster even at 36 triangles
        pl.variant = PT_EXTEND_FLAT;
        pl.smem = 0;
        int per_cu = 0;
        PT_HIP(ctx, hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void *>(k_extend_flat), TB, 0));
        pl.grid = ctx->num_cus * std::max(1, std::min(per_cu, 8));
        return PT_OK;
    }
    const size_t scene_bytes = 16 * LDS_NODE_F4 * (size_t)s->n_wide + sizeof(float4) * 9 * (size_t)s->n_tris;  // 3 permuted triangle copies
    // (round 2's 128-B eight-wide node with fp16 planes visited 26 % fewer nodes and fetched as many 128-B LINES -- two 64-B
    // BVH4 siblings share one -- and lost: 458 vs 395 ms of kernel time per 4 frames of C5; the 64-B node above is its successor)
    const bool auto8 = want == PT_EXTEND_AUTO && scene_bytes > 24 * 1024 && s->d_wide8 && (ctx->tune.hbm8 == 1 || auto8_big);
    if (want == PT_EXTEND_HBM8 || auto8) {
        pl.variant = PT_EXTEND_HBM8;
        pl.bvh8 = true;
        pl.lds_scene = false;
        // one stack entry per visited node: at most one per level of the 8-wide tree.  The LDS stack is sized to exactly that (8 M
        // triangles: 10 entries) -- then the kernel is instantiated without the spill column's address arithmetic (C5 +1.9 %, C5x
        // +2.5 %) and the LDS it does not take is there for the co-resident k_shade (9 ... 11 entries instead of 12: C5 +1 %; one
        // entry too few, i.e. the spill kernel: -1.7 %; profiles/r03bk_*, r03bl_*)
        const uint32_t bound8 = s->levels8 + 1u;
        pl.lds_stack = pt_tuned(ctx->tune.lds_stack, (int)std::min(std::max(bound8, 4u), 12u), 1, 32);
        pl.smem = (size_t)pl.lds_stack * TB * sizeof(uint2);
        const bool spills8 = bound8 > (uint32_t)pl.lds_stack;
        // 7 waves per SIMD where the walk waits on HBM (same rule as AUTO ray sorting: nodes + records beyond the Infinity Cache);
        // pt_tuning.extend_blocks = 6 / 7 forces either
        pl.waves7 = !spills8 && pl.lds_stack <= 10 &&
                    (ctx->tune.extend_blocks == 7 || (ctx->tune.extend_blocks < 0 && 64ull * s->n_wide8 + 64ull * s->n_tris > (256ull << 20)));
        int per_cu8 = 0;
        PT_HIP(ctx, hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu8, ptw_extend8_fn(false, spills8, pl.waves7), TB, pl.smem));
        per_cu8 = std::max(1, std::min(per_cu8, 8));
        // with the triangle vote at 16 lanes (launch_extend) the refill optimum moved from 32 idle lanes to 12: C5 2 907 ->
        // 3 240 Mrays/s, C5x 2 715 -> 2 960 for both together (profiles/r03bb_*, r03bc_*: 32: 3 025, 24: 3 140, 16: 3 230, 8: 3 235, 4: 3 170)
        pl.refill = pt_tuned(ctx->tune.refill, 12, 1, 64);
        pl.grid = ctx->num_cus * per_cu8;
        pl.spill_levels = bound8 > (uint32_t)pl.lds_stack ? bound8 - (uint32_t)pl.lds_stack : 0u;
        const size_t need8 = PT_MAX_PIPES * (size_t)std::max(pl.spill_levels, 1u) * (size_t)pl.grid * TB * sizeof(uint2);
        if (need8 > ctx->spill_bytes) {
            (void)hipFree(ctx->d_spill);
            ctx->d_spill = nullptr;
            ctx->spill_bytes = 0;
            PT_HIP(ctx, hipMalloc((void **)&ctx->d_spill, need8));
            ctx->spill_bytes = need8;
        }
        return PT_OK;
    }
    if (want == PT_EXTEND_LDS && scene_bytes > 96 * 1024) { ctx->err = "scene does not fit LDS"; return PT_ERR_UNSUPPORTED; }
    pl.lds_scene = want == PT_EXTEND_LDS || (want == PT_EXTEND_AUTO && scene_bytes <= 24 * 1024);
    pl.variant = pl.lds_scene ? PT_EXTEND_LDS : PT_EXTEND_HBM;
    // deep trees of big scenes: 12 LDS entries measured best on the 1M-triangle soup (4: -15 %, 8: -3 %,
    // 16: -5 %, 24: -16 %: beyond 12 the extra LDS costs occupancy; 9/10/11, which would admit a 7th block per CU: -2.4 %)
    pl.lds_stack = pl.lds_scene ? LDS_STACK : 12;
    // LDS-resident scenes are small enough for an exact stack bound (lbvh_build.hip: wide_stack_need):
    // if it fits 16 LDS entries the kernel is instantiated without the spill path (Cornell: 9)
    // (the no-spill kernel packs child words into 14 bits: <= 2047 triangles, <= 8191 nodes, leaves of <= 4)
    pl.spill = !(pl.lds_scene && s->stack_need <= 16u && s->n_tris <= 2047u && s->n_wide <= 8191u);
    if (!pl.spill) pl.lds_stack = (int)std::max(s->stack_need, 1u);
    pl.pairs = !pl.spill && s->pair_leaves && ctx->tune.pair_kernel != 0;
    pl.smem_wide_entries = (size_t)pl.lds_stack * TB * sizeof(uint2) + (pl.lds_scene ? scene_bytes : 0);
    pl.smem = pl.spill ? pl.smem_wide_entries : (size_t)pl.lds_stack * TB * sizeof(uint32_t) + scene_bytes;
    if (!pl.spill && pl.smem_wide_entries > 48 * 1024) {
        PT_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(k_extend<true, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)pl.smem_wide_entries));
        PT_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(k_extend<true, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)pl.smem_wide_entries));
    }
    const void *fn = pl.pairs ? reinterpret_cast<const void *>(k_extend_lds7p)
                     : !pl.spill ? reinterpret_cast<const void *>(k_extend_lds7)
                     : pl.lds_scene ? reinterpret_cast<const void *>(k_extend<true, false, true>)
                                    : ptw_extend_hbm_fn(false, ctx->tune.rec64 != 0);
    const void *fn_count = pl.pairs ? reinterpret_cast<const void *>(k_extend<true, true, false, true>)
                           : !pl.spill ? reinterpret_cast<const void *>(k_extend<true, true, false>)
                           : pl.lds_scene ? reinterpret_cast<const void *>(k_extend<true, true, true>)
                                          : ptw_extend_hbm_fn(true, ctx->tune.rec64 != 0);
    if (pl.smem > 48 * 1024)
        PT_HIP(ctx, hipFuncSetAttribute(fn_count, hipFuncAttributeMaxDynamicSharedMemorySize, (int)pl.smem));
    if (pl.smem > 48 * 1024) PT_HIP(ctx, hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)pl.smem));
    if (pl.smem > 48 * 1024 && !pl.spill)  // the shadow-ray twins of the two compact kernels (NEE pipeline)
        PT_HIP(ctx, hipFuncSetAttribute(pl.pairs ? reinterpret_cast<const void *>(k_extend_lds7p_sh) : reinterpret_cast<const void *>(k_extend_lds7_sh),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)pl.smem));
    int per_cu = 0;
    PT_HIP(ctx, hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, TB, pl.smem));
    per_cu = std::max(1, std::min(per_cu, 8));
    per_cu = pt_tuned(ctx->tune.extend_blocks, per_cu, 1, per_cu);
    // big scenes (vote-scheduled steps): 32 idle lanes measured best on C5 (16: -2.5 %, 48: -3 %)
    pl.refill = pt_tuned(ctx->tune.refill, pl.lds_scene ? REFILL_MIN_IDLE : 32, 1, 64);
    pl.grid = ctx->num_cus * per_cu;
    // the HBM variant of a scene whose traversed BVH4 is the collapsed LBVH walks the top-down layout of it
    pl.topdown4 = !pl.lds_scene && s->bvh4_builder != 1 && s->d_wide16t && ctx->tune.topdown4 != 0;
    // stack bound: the exact one of the BVH4 that is traversed when its builder computed it (small scenes; the
    // surface-area BVH4 is not bounded by the LBVH's height), else a BVH4 node pushes <= 3 entries per level and the
    // collapsed LBVH's wide height is <= binary height/2 + 1
    const uint32_t bound = pl.topdown4 ? 3u * s->levels4t + 1u
                           : s->stack_need != 0xFFFFFFFFu ? s->stack_need + 1u : 3u * (s->height_tree / 2u + 1u) + 1u;
    pl.spill_levels = bound > (uint32_t)pl.lds_stack ? bound - (uint32_t)pl.lds_stack : 0u;
    const size_t need = PT_MAX_PIPES * (size_t)std::max(pl.spill_levels, 1u) * (size_t)pl.grid * TB * sizeof(uint2);
    if (need > ctx->spill_bytes) {
        (void)hipFree(ctx->d_spill);
        ctx->d_spill = nullptr;
        ctx->spill_bytes = 0;
        PT_HIP(ctx, hipMalloc((void **)&ctx->d_spill, need));
        ctx->spill_bytes = need;
    }
    return PT_OK;
}

void launch_extend(const ExtendPlan &pl, pt_scene *s, const float4 *rayA, const float2 *rayB, float4 *hit,
                   uint32_t *hit_inst, const uint32_t *count_in, uint32_t *count_zero, unsigned long long *stats,
                   float tmin, float tmax, bool count, bool raw_hit, hipStream_t st, int pipe = 0,
                   hipEvent_t ev0 = nullptr, hipEvent_t ev1 = nullptr, const uint32_t *perm = nullptr,
                   const float *ray_tmax = nullptr)
{
    const int raw = raw_hit ? 1 : 0;  // hit records as (pos, V, W, det) for k_shade instead of (pos, t, u, v)
    // hipExtLaunchKernelGGL stamps THIS kernel's start/stop into ev0/ev1 (null = plain launch): under
    // two overlapping pipelines an event recorded between kernels would also count queueing time
    // each concurrently running extend kernel owns its own [spill_levels][grid*TB] region
    const size_t spill_off = (size_t)pipe * std::max(pl.spill_levels, 1u) * (size_t)pl.grid * TB;
    if (s->n_inst && pl.inst16 && tmin > 0.f) {
        // own region of the spill buffer, counted in dwords (the buffer is sized in 8-byte entries for the larger grid)
        uint32_t *sp32 = reinterpret_cast<uint32_t *>(reinterpret_cast<uint2 *>(s->ctx->d_spill) + (size_t)pipe * std::max(pl.spill_levels, 1u) * (size_t)std::max(pl.grid, pl.grid_inst_fallback) * TB);
        const uint32_t str = (uint32_t)pl.grid * TB;
        const NormBox nbt = { s->tlas_norm_c[0], s->tlas_norm_c[1], s->tlas_norm_c[2], s->tlas_norm_s[0], s->tlas_norm_s[1], s->tlas_norm_s[2],
                              s->tlas_norm_rs[0], s->tlas_norm_rs[1], s->tlas_norm_rs[2] };
        const NormBox nbb = { s->norm_c[0], s->norm_c[1], s->norm_c[2], s->norm_s[0], s->norm_s[1], s->norm_s[2], s->norm_rs[0], s->norm_rs[1], s->norm_rs[2] };
        const int enter_min = pt_tuned(s->ctx->tune.enter_min, 16, 1, 64);  // lanes that wait to enter an instance together (8, 16, 24 measured alike within 1 %)
        const int leaf_min = pt_tuned(s->ctx->tune.leaf_min, 8, 1, 64);    // ... and lanes that wait with a triangle leaf (extend_inst16.h)
        // the node loop yields to the lanes waiting with a leaf once fewer than 1/6 of the wave's rays still descend
        // (C4 11.7 -> 12.2 Grays/s; 2, 3, 4, 8 measured within 1 % of it, 0 = never: profiles/r02i_ab_c4_node_yield.log)
        const int node_yield = pt_tuned(s->ctx->tune.node_yield, 6, 0, 64);
#define PT_LAUNCH_INST16(C, P, S)                                                                                           \
    hipExtLaunchKernelGGL((k_extend_inst16<C, P, S>), dim3(pl.grid), dim3(TB), (uint32_t)pl.smem, st, ev0, ev1, 0u, s->d_tlas16, nbt, \
                          reinterpret_cast<const uint4 *>(s->d_wide16), nbb, s->d_tri4, s->n_wide, s->n_tris, s->d_inst6,     \
                          s->d_tlas_prim_of, rayA, rayB, hit, hit_inst, count_in, count_zero, stats, sp32, str, pl.refill, tmin, \
                          tmax, raw, pl.lds_stack, enter_min, leaf_min, node_yield, pl.n_tlas_lds, ray_tmax)
        if (ray_tmax) { if (s->pair_leaves) PT_LAUNCH_INST16(false, true, true); else PT_LAUNCH_INST16(false, false, true); }  // shadow rays (NEE)
        else if (s->pair_leaves) { if (count) PT_LAUNCH_INST16(true, true, false); else PT_LAUNCH_INST16(false, true, false); }
        else { if (count) PT_LAUNCH_INST16(true, false, false); else PT_LAUNCH_INST16(false, false, false); }
#undef PT_LAUNCH_INST16
        return;
    }
    if (s->n_inst) {
        const int grid_i = pl.inst16 ? pl.grid_inst_fallback : pl.grid;
        const size_t smem_i = pl.inst16 ? pl.smem_inst_fallback : pl.smem;
        uint2 *sp = reinterpret_cast<uint2 *>(s->ctx->d_spill) + (size_t)pipe * std::max(pl.spill_levels, 1u) * (size_t)std::max(pl.grid, pl.grid_inst_fallback) * TB;
        const uint32_t str = (uint32_t)grid_i * TB;
#define PT_LAUNCH_INST(C, L, S)                                                                                          \
    hipExtLaunchKernelGGL((k_extend_inst<C, L, S>), dim3(grid_i), dim3(TB), (uint32_t)smem_i, st, ev0, ev1, 0u, s->d_tlas_wide, \
                          s->d_wide, s->d_tri4, s->n_wide, s->n_tris, s->d_inst6, s->d_tlas_prim_of, rayA, rayB, hit,           \
                          hit_inst, count_in, count_zero, stats, sp, str, pl.refill, tmin, tmax, raw, ray_tmax)
        if (ray_tmax) { if (pl.lds_scene) PT_LAUNCH_INST(false, true, true); else PT_LAUNCH_INST(false, false, true); }  // shadow rays (NEE)
        else if (pl.lds_scene) {
            if (count) PT_LAUNCH_INST(true, true, false); else PT_LAUNCH_INST(false, true, false);
        } else {
            if (count) PT_LAUNCH_INST(true, false, false); else PT_LAUNCH_INST(false, false, false);
        }
#undef PT_LAUNCH_INST
        return;
    }
    if (pl.variant == PT_EXTEND_FLAT) {
        hipExtLaunchKernelGGL(k_extend_flat, dim3(pl.grid), dim3(TB), 0u, st, ev0, ev1, 0u, s->d_tri4, s->n_tris, rayA, rayB, hit,
                              count_in, count_zero, stats, tmin, tmax, raw);
        return;
    }
    uint2 *spill = reinterpret_cast<uint2 *>(s->ctx->d_spill) + spill_off;
    const uint32_t stride = (uint32_t)pl.grid * TB;
    if (pl.bvh8) {
        // the vote of the 8-wide kernel (extend8_kernel.h): a triangle step runs once tri_enter lanes wait with leaf triangles
        // (or more than descend), and repeats while tri_stay lanes still hold one.  Majority voting (64) parks ~25 lanes
        // behind every node step -- a node step is 240 instructions, a triangle step 110: 16 measured best (8: -2 %, 12: -0.5 %,
        // 20: equal on C5x, 24: -3 %; repeating triangle steps changes nothing: profiles/r03ba_ab_c5_vote.log)
        const int tri_enter = pt_tuned(s->ctx->tune.tri_enter, 16, 1, 64), tri_stay = pt_tuned(s->ctx->tune.tri_stay, 65, 1, 65);
        ptw_launch_extend8(count, pl.spill_levels > 0u, pl.waves7, pl.grid, pl.smem, st, ev0, ev1, s->d_wide8, s->norm_c, s->norm_s, s->norm_rs, s->d_tri4_8, s->d_shade64_8, rayA, rayB, hit,
                           count_in, count_zero, stats, spill, stride, pl.refill | (tri_enter << 8) | (tri_stay << 16), tmin, tmax, pl.lds_stack, raw, perm, ray_tmax);
        return;
    }
    const NormBox nbox = { s->norm_c[0], s->norm_c[1], s->norm_c[2], s->norm_s[0], s->norm_s[1], s->norm_s[2],
                           s->norm_rs[0], s->norm_rs[1], s->norm_rs[2] };
    // one-dword stack entries truncate the entry distance toward zero, which is only conservative for t >= 0, and the
    // sort of the one-dword keys takes entry distances for positive floats (tmin = 0 could make one -0): a tmin <= 0
    // (not valid in Vulkan, accepted here) runs the same plan through the 8-byte-entry kernel
    const bool no_spill = !pl.spill && tmin > 0.f;
    const size_t smem = (!pl.spill && !no_spill) ? pl.smem_wide_entries : pl.smem;
#define PT_LAUNCH_EXTEND(L, C, S)                                                                                     \
    hipExtLaunchKernelGGL((k_extend<L, C, S>), dim3(pl.grid), dim3(TB), (uint32_t)smem, st, ev0, ev1, 0u, s->d_wide, \
                          s->d_wide16, nbox,                                                                          \
                          s->d_tri4, s->n_wide, s->n_tris, rayA, rayB, hit, count_in, count_zero, stats, spill, stride, \
                          pl.refill, tmin, tmax, pl.lds_stack, raw, nullptr, ray_tmax, nullptr)
    if (no_spill && pl.pairs) {
        if (count)
            hipExtLaunchKernelGGL((k_extend<true, true, false, true>), dim3(pl.grid), dim3(TB), (uint32_t)smem, st, ev0, ev1, 0u, s->d_wide,
                                  s->d_wide16, nbox, s->d_tri4, s->n_wide, s->n_tris, rayA, rayB, hit, count_in, count_zero, stats, spill,
                                  stride, pl.refill, tmin, tmax, pl.lds_stack, raw, nullptr, ray_tmax, nullptr);
        else
            hipExtLaunchKernelGGL(ray_tmax ? k_extend_lds7p_sh : k_extend_lds7p, dim3(pl.grid), dim3(TB), (uint32_t)smem, st, ev0, ev1, 0u, s->d_wide, s->d_wide16, nbox,
                                  s->d_tri4, s->n_wide, s->n_tris, rayA, rayB, hit, count_in, count_zero, stats, spill, stride,
                                  pl.refill, tmin, tmax, pl.lds_stack, raw, nullptr, ray_tmax, nullptr);
    } else if (no_spill) {
        if (count) PT_LAUNCH_EXTEND(true, true, false);
        else
            hipExtLaunchKernelGGL(ray_tmax ? k_extend_lds7_sh : k_extend_lds7, dim3(pl.grid), dim3(TB), (uint32_t)smem, st, ev0, ev1, 0u, s->d_wide, s->d_wide16, nbox,
                                  s->d_tri4, s->n_wide, s->n_tris, rayA, rayB, hit, count_in, count_zero, stats, spill, stride,
                                  pl.refill, tmin, tmax, pl.lds_stack, raw, nullptr, ray_tmax, nullptr);
    } else if (pl.lds_scene) {
        if (count) PT_LAUNCH_EXTEND(true, true, true); else PT_LAUNCH_EXTEND(true, false, true);
    } else {
        ptw_launch_extend_hbm(count, s->ctx->tune.rec64 != 0, pl.grid, smem, st, ev0, ev1, s->d_wide, pl.topdown4 ? reinterpret_cast<const uint2 *>(s->d_wide16t) : s->d_wide16, s->norm_c, s->norm_s, s->norm_rs, s->d_tri4,
                              s->d_shade64, s->n_wide, s->n_tris, rayA, rayB, hit, count_in, count_zero, stats, spill, stride,
                              pl.refill | (pt_tuned(s->ctx->tune.tri_enter, 0, 0, 64) << 8), tmin, tmax, pl.lds_stack, raw, perm, ray_tmax);
    }
#undef PT_LAUNCH_EXTEND
}

// Bytes of workspace per path slot that do not depend on the sample-group shape: two queue sets
// (id 8 + state 16 + rayA 16 + rayB 8), hit 16 + instance 4, term count 4, pool head 4.
// (PT_PIPELINE_FUSED has no queues: 8 B per slot)
constexpr size_t SLOT_BYTES_QUEUES = 2 * (8 + 16 + 16 + 8) + 16 + 4, SLOT_BYTES_META = 4 + 4;
constexpr size_t SLOT_BYTES = SLOT_BYTES_QUEUES + SLOT_BYTES_META;

struct WorkNeed { size_t slots, color, terms, terms_over, total; };
WorkNeed work_need(uint64_t n_slots, uint32_t groups, uint32_t term_cap, uint32_t term_pcap, bool queues = true)
{
    WorkNeed n{};
    n.slots = (size_t)std::max<uint64_t>(n_slots, 1);
    n.color = groups == 1 ? n.slots : 0;
    n.terms = groups > 1 ? n.slots * (size_t)term_pcap : 0;
    n.terms_over = groups > 1 ? n.slots * (size_t)(term_cap - term_pcap) : 0;
    n.total = n.slots * (queues ? SLOT_BYTES : SLOT_BYTES_META) + sizeof(float4) * (n.color + n.terms + n.terms_over) +
              (groups > 1 ? sizeof(float4) * (size_t)SPILL_POOL_ENTRIES : 0);
    return n;
}

// One workspace allocation.  Out of memory (the device's, or the context's PT_MEM_BUDGET_MB) is PT_ERR_OOM, and HIP's
// sticky error is cleared so that the context stays usable.
pt_status work_alloc(pt_ctx *ctx, pt_film::Work &w, void **p, size_t bytes, size_t limit)
{
    *p = nullptr;
    if (limit && w.bytes + bytes > limit) {
        ctx->err = "wavefront workspace exceeds the memory budget (" + std::to_string((w.bytes + bytes) >> 20) + " MB wanted, " +
                   std::to_string(limit >> 20) + " MB allowed): fewer frames_in_flight / sample_groups fit";
        return PT_ERR_OOM;
    }
    const hipError_t e = hipMalloc(p, bytes);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        *p = nullptr;
        ctx->err = std::string("hipMalloc of ") + std::to_string(bytes >> 20) + " MB of wavefront workspace: " + hipGetErrorString(e);
        return e == hipErrorOutOfMemory ? PT_ERR_OOM : PT_ERR_HIP;
    }
    w.bytes += bytes;
    return PT_OK;
}

// Frees the shape-dependent buffers (everything but the tile list and the counters) and zeroes their capacities:
// the state after a failed grow -- the film itself (d_rgb / d_bgra) is untouched and the next render re-allocates.
void free_shape_buffers(pt_film::Work &w)
{
    for (int i = 0; i < 2; i++) {
        (void)hipFree(w.d_qid[i]); (void)hipFree(w.d_qstate[i]); (void)hipFree(w.d_qrayA[i]); (void)hipFree(w.d_qrayB[i]);
        w.d_qid[i] = nullptr; w.d_qstate[i] = w.d_qrayA[i] = nullptr; w.d_qrayB[i] = nullptr;
    }
    (void)hipFree(w.d_hit); (void)hipFree(w.d_hit_inst); (void)hipFree(w.d_nterm); (void)hipFree(w.d_spill_head);
    (void)hipFree(w.d_color); (void)hipFree(w.d_terms); (void)hipFree(w.d_terms_over); (void)hipFree(w.d_spill);
    w.d_hit = nullptr; w.d_hit_inst = nullptr; w.d_nterm = nullptr; w.d_spill_head = nullptr;
    w.d_color = nullptr; w.d_terms = nullptr; w.d_terms_over = nullptr; w.d_spill = nullptr;
    w.cap_slots = w.cap_meta = w.cap_color = w.cap_terms = w.cap_terms_over = 0;
    w.bytes = w.sort_bytes;  // (the ray-sort scratch is not a shape buffer)
}

// Workspace for (rank, world) tiles, `lanes` frames in flight and `groups` sample groups.  Buffers only
// ever grow: a later call with a smaller shape reuses them (hipMalloc of tens of GB costs 100s of ms).
// A grow that does not fit returns PT_ERR_OOM and leaves the film WITHOUT shape buffers (all freed, capacities 0).
pt_status ensure_work(pt_film *f, uint32_t rank, uint32_t world, uint32_t lanes, uint32_t groups, uint32_t term_cap,
                      uint32_t term_pcap, bool queues = true)
{
    // queues = false (PT_PIPELINE_FUSED): no path queues and no hit records, only the per-slot radiance arrays
    pt_ctx *ctx = f->ctx;
    pt_film::Work &w = f->work;
    if (!w.d_tiles || w.rank != rank || w.world != world) {
        (void)hipFree(w.d_tiles);
        w.d_tiles = nullptr;
        const uint32_t tiles_x = (f->w + 7) / 8, tiles_y = (f->h + 7) / 8;
        std::vector<uint32_t> tiles;
        for (uint32_t ty = 0; ty < tiles_y; ty++)
            for (uint32_t tx = 0; tx < tiles_x; tx++)
                if ((tx + ty) % world == rank) tiles.push_back(tx | (ty << 16));
        w.rank = rank; w.world = world;
        w.n_tiles = (uint32_t)tiles.size();
        PT_HIP(ctx, hipMalloc((void **)&w.d_tiles, sizeof(uint32_t) * std::max<size_t>(tiles.size(), 1)));
        if (!tiles.empty())
            PT_HIP(ctx, hipMemcpy(w.d_tiles, tiles.data(), sizeof(uint32_t) * tiles.size(), hipMemcpyHostToDevice));
    }
    const uint64_t n_slots64 = (uint64_t)lanes * groups * w.n_tiles * 64ull;
    if (n_slots64 >= (1ull << 31)) {
        ctx->err = "too many path slots (frames_in_flight x sample_groups x pixels >= 2^31)";
        return PT_ERR_INVALID_ARG;
    }
    if (!w.d_count) PT_HIP(ctx, hipMalloc((void **)&w.d_count, sizeof(uint32_t) * 2 * PT_MAX_PIPES));  // queue sizes, 2 per pipeline
    const WorkNeed need = work_need(n_slots64, groups, term_cap, term_pcap, queues);
    const size_t ns = need.slots;
    const size_t limit = ctx->mem_budget;
    pt_status rc = PT_OK;
#define PT_WORK_ALLOC(PTR, BYTES) \
    if (rc == PT_OK) rc = work_alloc(ctx, w, (void **)&(PTR), (BYTES), limit)
    if (queues && ns > w.cap_slots) {
        // the queue set goes as a whole: free first (peak = the new size, not old + new)
        for (int i = 0; i < 2; i++) {
            (void)hipFree(w.d_qid[i]); (void)hipFree(w.d_qstate[i]); (void)hipFree(w.d_qrayA[i]); (void)hipFree(w.d_qrayB[i]);
            w.d_qid[i] = nullptr; w.d_qstate[i] = w.d_qrayA[i] = nullptr; w.d_qrayB[i] = nullptr;
        }
        (void)hipFree(w.d_hit); (void)hipFree(w.d_hit_inst);
        w.d_hit = nullptr; w.d_hit_inst = nullptr;
        w.bytes -= w.cap_slots * SLOT_BYTES_QUEUES;
        w.cap_slots = 0;
        for (int i = 0; i < 2; i++) {
            PT_WORK_ALLOC(w.d_qid[i], sizeof(uint2) * ns);
            PT_WORK_ALLOC(w.d_qstate[i], sizeof(float4) * ns);
            PT_WORK_ALLOC(w.d_qrayA[i], sizeof(float4) * ns);
            PT_WORK_ALLOC(w.d_qrayB[i], sizeof(float2) * ns);
        }
        PT_WORK_ALLOC(w.d_hit, sizeof(float4) * ns);
        PT_WORK_ALLOC(w.d_hit_inst, sizeof(uint32_t) * ns);
        if (rc == PT_OK) w.cap_slots = ns;
    }
    if (rc == PT_OK && ns > w.cap_meta) {
        (void)hipFree(w.d_nterm); (void)hipFree(w.d_spill_head);
        w.d_nterm = nullptr; w.d_spill_head = nullptr;
        w.bytes -= w.cap_meta * SLOT_BYTES_META;
        w.cap_meta = 0;
        PT_WORK_ALLOC(w.d_nterm, sizeof(uint32_t) * ns);
        PT_WORK_ALLOC(w.d_spill_head, sizeof(uint32_t) * ns);
        if (rc == PT_OK) w.cap_meta = ns;
    }
    if (rc == PT_OK && need.color > w.cap_color) {
        (void)hipFree(w.d_color);
        w.bytes -= sizeof(float4) * w.cap_color;
        w.d_color = nullptr; w.cap_color = 0;
        PT_WORK_ALLOC(w.d_color, sizeof(float4) * need.color);
        if (rc == PT_OK) w.cap_color = need.color;
    }
    // primary log: term_pcap entries per slot (dense, what is normally touched); overflow: the rest of the
    // worst case (one entry per ray), allocated but rarely touched
    if (rc == PT_OK && need.terms > w.cap_terms) {
        (void)hipFree(w.d_terms);
        w.bytes -= sizeof(float4) * w.cap_terms;
        w.d_terms = nullptr; w.cap_terms = 0;
        PT_WORK_ALLOC(w.d_terms, sizeof(float4) * need.terms);
        if (rc == PT_OK) w.cap_terms = need.terms;
    }
    if (rc == PT_OK && need.terms_over > w.cap_terms_over) {
        (void)hipFree(w.d_terms_over);
        w.bytes -= sizeof(float4) * w.cap_terms_over;
        w.d_terms_over = nullptr; w.cap_terms_over = 0;
        PT_WORK_ALLOC(w.d_terms_over, sizeof(float4) * need.terms_over);
        if (rc == PT_OK) w.cap_terms_over = need.terms_over;
    }
    if (rc == PT_OK && groups > 1 && !w.d_spill) PT_WORK_ALLOC(w.d_spill, sizeof(float4) * (size_t)SPILL_POOL_ENTRIES);
#undef PT_WORK_ALLOC
    if (rc != PT_OK) {
        free_shape_buffers(w);
        w.lanes = w.groups = w.term_cap = 0;
        w.n_slots = 0;
        return rc;
    }
    w.lanes = lanes; w.groups = groups; w.term_cap = term_cap;
    w.n_slots = (uint32_t)n_slots64;
    return PT_OK;
}

struct RenderShape {
    uint32_t lanes = 1, groups = 1, group_size = 1, term_cap = 0, term_pcap = 0;
    bool bounded = false;  // term_cap < group_size * max_depth: a full log is detected and the batch redone ungrouped
};

// frames in flight x sample groups: enough live paths (~32M) to fill the chip several times over, and slots that
// do not live longer than they have to
// `shrink`: 0 for the first try; pt_render retries with 1, 2, ... after an out-of-memory workspace grow, each step
// halving the memory the AUTO shape may plan for (explicit frames_in_flight / sample_groups are never overridden).
// `launch_class`: 0 instanced scenes, 1 scenes walked out of L2 / MALL / HBM (no LDS copy), 2 single-level scenes in LDS.  Class 1: the launches take milliseconds per million
// rays and what they gain from being LONG is measured: 1 M-triangle soup, 4 frames of 16 spp -- 4 groups (3.7 M rays per launch)
// 2 617 Mrays/s, 8 groups 2 799, 16 groups (14.8 M) 2 889; 16 frames x 4 groups 2 886, x 8 (29.6 M) 2 926; the 8 M-triangle soup
// at 2 frames +3 % from 8 to 16 groups (profiles/r03au_shapes_c5_c4.log).  So the sample groups of such scenes aim at 128 M live
// paths; the Cornell-class scenes followed in the round's last session (below), instanced scenes aim at 32 M.
RenderShape choose_shape(const pt_film *f, const pt_params *p, int launch_class, int shrink = 0)
{
    RenderShape sh;
    const uint64_t pixels_local = ((uint64_t)((f->w + 7) / 8) * ((f->h + 7) / 8) * 64ull + p->world - 1) / p->world;
    const uint64_t target = 32ull << 20;  // 128 B of queue state per live path
    uint32_t lanes = p->frames_in_flight;
    if (lanes == 0) {
        // up to 32 frames / 64 M paths in flight, in EQUAL batches: 20 frames run as 1 x 20 (measured 22.2 Grays/s on
        // the Cornell box) rather than 16 + 4 (21.3), 40 as 2 x 20; every batch pays the same ~256 rounds of
        // per-launch fixed cost (~27 us per round and pipeline), so fewer and fuller batches are better
        const uint64_t cap = std::max<uint64_t>(1, std::min<uint64_t>(32, 2 * target / std::max<uint64_t>(pixels_local, 1)));
        const uint64_t batches = ((uint64_t)p->frame_count + cap - 1) / cap;
        lanes = (uint32_t)(((uint64_t)p->frame_count + batches - 1) / batches);
    }
    lanes = std::max(1u, std::min(lanes, p->frame_count));
    // A blocking render can check a batch and redo it; PT_FLAG_ASYNC can not, and keeps the worst-case log.
    const bool can_redo = (p->flags & PT_FLAG_ASYNC) == 0;
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) free_b = 0;
    // what the workspace may occupy: the device's free memory plus what this film already holds, within the
    // context's budget (PT_MEM_BUDGET_MB), 7/8 of it planned for
    uint64_t avail = (uint64_t)free_b + f->work.bytes;
    if (f->ctx->mem_budget) avail = std::min<uint64_t>(avail, f->ctx->mem_budget);
    avail = (avail - avail / 8) >> std::min(shrink, 40);
    const uint64_t have_log = f->work.cap_terms_over * sizeof(float4);  // already ours: counts as free
    if (f->ctx->mem_budget) free_b = (size_t)std::min<uint64_t>(free_b, avail);
    // sample groups: split each pixel's samples over several slots; the term logs keep the sum order exact.
    //  (a) few frames asked for: the frames in flight alone cannot fill the chip;
    //  (b) a slot lives group_size x depth rounds and every round costs ~27 us of launch-bound time per pipeline
    //      whatever its queue holds: 16 frames x 4 groups need 64 rounds instead of 256 (Cornell box, 1080p: +4 %),
    //      as long as the slots (<= 160 M: 21 GB of queues + 25 GB of primary log; 288 M for single-level scenes in LDS) allow it.
    uint32_t groups = p->sample_groups;
    if (groups == 0) {
        groups = 1;
        const uint64_t have = std::max<uint64_t>((uint64_t)lanes * pixels_local, 1);
        // live paths aimed at: 128 M for scenes walked out of HBM (above), 32 M for instanced scenes (C4, K = 8: 4 groups 12.74
        // Grays/s, 8 groups 12.52), and -- round 3, last session; 32 M until then -- 256 M for single-level scenes in LDS: a
        // Cornell-class render is better off with MORE slots and fewer rounds.  K = 2 (config C2 exactly): 8 groups 23.7 Grays/s,
        // 16: 25.4, 32: 26.5; K = 1: 16 groups 23.0, 32: 25.3; K = 4: 4 groups 23.7, 16: 26.3, 32 (266 M slots): 27.1; 16 frames as
        // two batches of 8 with 16 groups: 27.1; K = 16: 4 groups (133 M slots, 51 GB) 23.0 / 26.6 / 26.3 in three
        // processes, 8 groups (266 M, 70 GB) 26.5 / 27.1 / 26.5, 16 groups (109 GB) 26.3 / 27.3 / 27.0
        // (profiles/r03br_*, r03bs_*, r03bt_*, r03bu_*)
        // (instanced scenes, on the round's final kernels: C4 at K = 8 with 2 / 4 / 8 / 16 groups 13.40 / 13.77 / 13.88 / 13.70 Grays/s,
        // profiles/r03cs_c4_shapes_final.log -- so they aim at 128 M now as well; the 32 M of the comment above was measured before)
        double want = (double)((launch_class == 2 ? 8 : 4) * target) / (double)have;
        const uint64_t slot_budget = (launch_class == 2 ? 288ull : 160ull) << 20;
        if (can_redo && have * 4 <= slot_budget) want = std::max(want, 4.0);
        else if (can_redo && have * 2 <= slot_budget) want = std::max(want, 2.0);
        if (want >= 2.0) {
            // even groups only (uneven tails measured 8 % slower): the divisor of spp closest to `want`
            uint32_t g = 1;
            double best = 1e30;
            for (uint32_t d = 1; d <= p->spp_per_frame; d++) {
                if (p->spp_per_frame % d) continue;
                const double r = d > want ? d / want : want / d;
                if (r < best) { best = r; g = d; }
            }
            // without the redo the worst-case log (one 16-B term per ray) is allocated in full, so it has to fit:
            // at most 80 GB of the 288 and half of what is free right now
            const uint64_t log_bytes = (uint64_t)lanes * pixels_local * p->spp_per_frame * p->max_depth * 16ull;
            if (g > 1 && (can_redo || (log_bytes <= (80ull << 30) && log_bytes <= have_log + free_b / 2))) groups = g;
        }
    }
    groups = std::max(1u, std::min(groups, p->spp_per_frame));
    // AUTO shapes have to fit the memory there is (queues + primary log; the overflow log is budgeted below): first
    // fewer sample groups (the next smaller divisor of spp), then fewer frames in flight
    auto planned = [&](uint32_t l, uint32_t g) {
        const uint32_t gs = (p->spp_per_frame + g - 1) / g;
        return work_need((uint64_t)l * g * pixels_local, g, g > 1 ? std::min(gs * p->max_depth, gs + 2u) : 0u,
                         g > 1 ? std::min(gs * p->max_depth, gs + 2u) : 0u).total;
    };
    while (planned(lanes, groups) > avail) {
        if (p->sample_groups == 0 && groups > 1) {
            uint32_t g = groups - 1;
            while (g > 1 && p->spp_per_frame % g) g--;
            groups = g;
        } else if (p->frames_in_flight == 0 && lanes > 1) {
            lanes = (lanes + 1) / 2;
        } else {
            break;  // explicit shape (or one frame, one group): ensure_work reports PT_ERR_OOM if it does not fit
        }
    }
    sh.group_size = (p->spp_per_frame + groups - 1) / groups;
    sh.groups = (p->spp_per_frame + sh.group_size - 1) / sh.group_size;  // no empty groups
    const uint32_t worst = sh.groups > 1 ? sh.group_size * p->max_depth : 0u;  // every ray of a slot adds a term
    sh.term_cap = worst;
    sh.term_pcap = std::min(worst, sh.group_size + 2u);  // ~1 term per sample is typical (the miss that ends it)
    if (sh.groups > 1 && can_redo) {
        // overflow log within a budget (16 GB, a quarter of the free memory) instead of the worst case (136 GB for
        // 16 frames x 4 groups at 1080p); a slot that fills it raises a flag and the batch is redone with groups == 1
        const uint64_t n_slots = (uint64_t)lanes * sh.groups * pixels_local;
        const uint64_t room = avail > planned(lanes, sh.groups) ? avail - planned(lanes, sh.groups) : 0;
        const uint64_t budget = std::min<uint64_t>(std::min<uint64_t>(16ull << 30, (have_log + free_b) / 4), room);
        uint64_t ocap = std::min<uint64_t>(worst - sh.term_pcap, budget / std::max<uint64_t>(n_slots * sizeof(float4), 1));
        if (f->ctx->tune.term_ocap >= 0) ocap = std::min<uint64_t>(ocap, (uint64_t)f->ctx->tune.term_ocap);  // tests
        sh.term_cap = sh.term_pcap + (uint32_t)ocap;
        sh.bounded = sh.term_cap < worst;
    }
    sh.lanes = lanes;
    return sh;
}

// The shape of a render and its workspace.  An AUTO shape that does not fit after all (another allocator took the
// memory between hipMemGetInfo and hipMalloc) is planned again for half the memory, down to one frame and one group;
// an explicit shape that does not fit is PT_ERR_OOM.  Either way a failure leaves the film usable.
pt_status shape_and_work(pt_film *f, const pt_params *p_in, RenderShape &sh, int launch_class)
{
    pt_status rc = PT_OK;
    pt_params p_local = *p_in;
    if (p_local.pipeline == PT_PIPELINE_WAVEFRONT_NEE) p_local.sample_groups = 1;  // (up to two radiance terms per hit -- a camera ray's emitter hit, the light sample -- so a sample has more than group_size + 2: the plain accumulator)
    const pt_params *p = &p_local;
    for (int attempt = 0; attempt < 12; attempt++) {
        sh = choose_shape(f, p, launch_class, attempt);
        rc = ensure_work(f, p->rank, p->world, sh.lanes, sh.groups, sh.term_cap, sh.term_pcap);
        if (rc != PT_ERR_OOM) return rc;
        const bool can_shrink = (p->frames_in_flight == 0 && sh.lanes > 1) || (p->sample_groups == 0 && sh.groups > 1);
        if (!can_shrink) return rc;
    }
    return rc;
}

pt_status check_params(pt_scene *s, pt_film *f, const pt_params *p)
{
    pt_ctx *ctx = s->ctx;
    if (p->width != f->w || p->height != f->h) { ctx->err = "params width/height differ from the film's"; return PT_ERR_INVALID_ARG; }
    if (p->world == 0 || p->rank >= p->world) { ctx->err = "rank/world invalid"; return PT_ERR_INVALID_ARG; }
    if (p->spp_per_frame == 0 || p->spp_per_frame > 0xFFFFu || p->max_depth == 0 || p->max_depth > 0xFFFFu) {
        ctx->err = "spp_per_frame and max_depth must be in 1..65535";
        return PT_ERR_INVALID_ARG;
    }
    if (p->frame < 0 || p->frame_count == 0) { ctx->err = "frame must be >= 0 and frame_count >= 1"; return PT_ERR_INVALID_ARG; }
    if (p->pipeline > PT_PIPELINE_FUSED) { ctx->err = "unknown pipeline"; return PT_ERR_UNSUPPORTED; }
    if (p->pipeline == PT_PIPELINE_FUSED && (p->flags & (PT_FLAG_ASYNC | PT_FLAG_COUNT_VISITS))) {
        ctx->err = "the fused pipeline has no asynchronous and no instrumented form";
        return PT_ERR_UNSUPPORTED;
    }
    if (p->pipeline == PT_PIPELINE_WAVEFRONT_NEE) {
        if (p->extend == PT_EXTEND_FLAT) { ctx->err = "the NEE pipeline has no flat extend variant (shadow rays need a per-ray tmax)"; return PT_ERR_UNSUPPORTED; }
        if (p->sample_groups > 1) { ctx->err = "the NEE pipeline runs one sample group per pixel"; return PT_ERR_UNSUPPORTED; }
    }
    return PT_OK;
}

RenderConst make_render_const(const pt_params *p, const pt_film::Work &w, const RenderShape &sh)
{
    RenderConst rc{};
    rc.cam = { p->cam_origin[0], p->cam_origin[1], p->cam_origin[2], p->cam_target[0], p->cam_target[1], p->cam_target[2],
               (float)p->width, (float)p->height };
    for (int k = 0; k < 3; k++) rc.env[k] = p->env[k];
    rc.tmin = p->tmin; rc.tmax = p->tmax;
    rc.width = p->width; rc.height = p->height; rc.tiles_x = (p->width + 7) / 8;
    rc.spp = p->spp_per_frame; rc.max_depth = p->max_depth;
    rc.slots_per_lane = w.n_tiles * 64u;
    rc.groups = sh.groups; rc.group_size = sh.group_size; rc.term_cap = sh.term_cap;
    rc.div_spl.init(std::max(rc.slots_per_lane, 1u)); rc.div_groups.init(std::max(sh.groups, 1u));
    rc.term_pcap = sh.term_pcap;
    rc.n_slots = w.n_slots;
    return rc;
}

// pixels of this rank's 8x8 tiles that lie inside the image (samples started = that x spp x frames)
uint64_t valid_local_pixels(const pt_film *f, const pt_params *p)
{
    uint64_t valid = 0;
    const uint32_t tiles_x = (f->w + 7) / 8, tiles_y = (f->h + 7) / 8;
    for (uint32_t ty = 0; ty < tiles_y; ty++)
        for (uint32_t tx = 0; tx < tiles_x; tx++)
            if ((tx + ty) % p->world == p->rank)
                valid += (uint64_t)std::min(8u, f->w - tx * 8) * std::min(8u, f->h - ty * 8);
    return valid;
}

// what pt_stats.workspace_bytes reports: everything the film's wavefront workspace holds + the context's stack-spill area
uint64_t workspace_bytes(const pt_film *f)
{
    const pt_film::Work &w = f->work;
    return (uint64_t)w.bytes + (uint64_t)w.cap_sq * (16 + 8 + 16 + 4 + 4 + 16) + (uint64_t)f->ctx->spill_bytes;
}

// ---- PT_PIPELINE_FUSED (fused_kernel.h): host side -------------------------------------------------------------------
struct FusedPlan { size_t smem = 0; int grid = 0, lds_stack = 0, refill = 16; };

pt_status plan_fused(pt_scene *s, const ExtendPlan &pl, float tmin, FusedPlan &fp)
{
    pt_ctx *ctx = s->ctx;
    const size_t tables = sizeof(float4) * 5 * (size_t)s->n_tris;  // shade4 + tangent frames (the vertices are the kz = 2 triangle copy)
    if (s->n_inst || pl.variant != PT_EXTEND_LDS || pl.spill || !pl.pairs || !(tmin > 0.f) || tables > 16 * 1024) {
        ctx->err = "PT_PIPELINE_FUSED is for single-level scenes whose BVH4, triangles and shading tables fit LDS (the compact pair-leaf "
                   "kernel's class: <= 2047 triangles in <= 24 KB, stack bound <= 16, tmin > 0)";
        return PT_ERR_UNSUPPORTED;
    }
    fp.lds_stack = pl.lds_stack;
    fp.smem = pl.smem + tables + sizeof(uint32_t) * FS_FIELDS * TB + sizeof(uint32_t) * (TB / 64) * (PT_FUSED_BATCH / 64);
    for (const void *fn : { reinterpret_cast<const void *>(k_fused<false>), reinterpret_cast<const void *>(k_fused<true>) })
        if (fp.smem > 48 * 1024) PT_HIP(ctx, hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)fp.smem));
    int per_cu = 0;
    PT_HIP(ctx, hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void *>(k_fused<false>), TB, fp.smem));
    per_cu = std::max(1, std::min(per_cu, 8));
    per_cu = pt_tuned(ctx->tune.extend_blocks, per_cu, 1, per_cu);
    fp.grid = ctx->num_cus * per_cu;
    fp.refill = pt_tuned(ctx->tune.refill, 16, 1, 64);  // of 64: the share of a wave's live lanes that must wait before the shade block runs
    return PT_OK;
}

// The shape of a fused render: frames in flight as the wavefront pipeline batches them (<= 32, equal batches); sample groups
// only to shorten the tail of a batch -- the last slots handed out run alone at the end, and a slot of 32 samples is up to 256
// rays = ~4 ms of a lane's time against 8 ms for a whole frame: frames x groups >= 16 keeps that tail under ~2 % of a batch.
// Explicit frames_in_flight / sample_groups are taken as given.
void fused_shape_defaults(const pt_params *p, pt_params &q)
{
    q = *p;
    if (q.frames_in_flight == 0) {
        const uint32_t cap = 32;
        const uint32_t batches = (p->frame_count + cap - 1) / cap;
        q.frames_in_flight = (p->frame_count + batches - 1) / batches;
    }
    q.frames_in_flight = std::max(1u, std::min(q.frames_in_flight, p->frame_count));
    if (q.sample_groups == 0) {
        uint32_t g = 1;
        while (g < p->spp_per_frame && (g * q.frames_in_flight < 16u || p->spp_per_frame % g)) g++;
        q.sample_groups = g;
    }
}

pt_status render_fused(pt_scene *s, pt_film *f, const pt_params *p_in, const ExtendPlan &pl, bool nested, bool prepare_only)
{
    pt_ctx *ctx = s->ctx;
    hipStream_t st = ctx->stream;
    FusedPlan fp;
    pt_status rc_ = plan_fused(s, pl, p_in->tmin, fp);
    if (rc_ != PT_OK) return rc_;
    pt_params q;
    fused_shape_defaults(p_in, q);
    const pt_params *p = &q;
    RenderShape sh;
    for (int attempt = 0; attempt < 12; attempt++) {  // (memory taken between hipMemGetInfo and hipMalloc: plan again for half of it)
        sh = choose_shape(f, p, 2, attempt);
        rc_ = ensure_work(f, p->rank, p->world, sh.lanes, sh.groups, sh.term_cap, sh.term_pcap, false);
        if (rc_ != PT_ERR_OOM) break;
    }
    if (!nested) {
        ctx->stats.frames_in_flight = sh.lanes;
        ctx->stats.sample_groups = sh.groups;
    }
    if (rc_ != PT_OK) return rc_;
    ctx->stats.workspace_bytes = workspace_bytes(f);
    if (prepare_only) return PT_OK;
    pt_film::Work &w = f->work;
    unsigned long long *const d_overflow = ctx->d_stats + 6, *const d_spill_count = ctx->d_stats + 7;
    uint32_t spill_cap = sh.bounded ? SPILL_POOL_ENTRIES : 0u;
    if (ctx->tune.term_spill >= 0) spill_cap = std::min<uint32_t>(spill_cap, (uint32_t)ctx->tune.term_spill);
    Radiance rad = { w.d_color, w.d_terms, w.d_terms_over, w.d_nterm, w.d_spill, w.d_spill_head, d_spill_count, spill_cap, d_overflow };
    RenderConst rc = make_render_const(p, w, sh);
    const bool profile = !nested && (p->flags & PT_FLAG_PROFILE) != 0;
    ctx->stats.extend_variant = pl.variant;
    ctx->stats.pipelines = 1;
    if (!nested) PT_HIP(ctx, hipEventRecord(ctx->ev_a, st));
    std::vector<hipEvent_t> evs;
    size_t ev_used = 0;
    for (uint32_t done = 0; w.n_slots > 0 && done < p->frame_count; done += sh.lanes) {
        rc.frame_base = p->frame + (int32_t)done;
        rc.lanes_active = std::min(sh.lanes, p->frame_count - done);
        unsigned long long rays_before = 0;
        if (sh.bounded) {
            PT_HIP(ctx, hipStreamSynchronize(st));
            PT_HIP(ctx, hipMemcpy(&rays_before, ctx->d_stats, sizeof(rays_before), hipMemcpyDeviceToHost));
            PT_HIP(ctx, hipMemsetAsync(d_spill_count, 0, sizeof(unsigned long long), st));
        }
        PT_HIP(ctx, hipMemsetAsync(w.d_count, 0, sizeof(uint32_t), st));  // the slot counter
        const uint32_t n_slots = rc.lanes_active * sh.groups * rc.slots_per_lane;
        hipEvent_t e0 = nullptr, e1 = nullptr;
        if (profile) {
            while (ctx->ev_pool.size() < ev_used + 2) {
                hipEvent_t e = nullptr;
                PT_HIP(ctx, hipEventCreate(&e));
                ctx->ev_pool.push_back(e);
            }
            e0 = ctx->ev_pool[ev_used++]; e1 = ctx->ev_pool[ev_used++];
            evs.push_back(e0); evs.push_back(e1);
        }
        if (sh.groups > 1)
            hipExtLaunchKernelGGL((k_fused<true>), dim3(fp.grid), dim3(TB), (uint32_t)fp.smem, st, e0, e1, 0u, rc, w.d_tiles, rad, s->d_wide, s->d_tri4,
                                  s->d_shade4, s->d_frame4, s->n_wide, s->n_tris, 0u, n_slots, w.d_count, ctx->d_stats, fp.refill, p->tmin, p->tmax, fp.lds_stack);
        else
            hipExtLaunchKernelGGL((k_fused<false>), dim3(fp.grid), dim3(TB), (uint32_t)fp.smem, st, e0, e1, 0u, rc, w.d_tiles, rad, s->d_wide, s->d_tri4,
                                  s->d_shade4, s->d_frame4, s->n_wide, s->n_tris, 0u, n_slots, w.d_count, ctx->d_stats, fp.refill, p->tmin, p->tmax, fp.lds_stack);
        PT_HIP(ctx, hipGetLastError());
        ctx->stats.launches_extend++;
        ctx->stats.rounds++;
        bool redo = false;
        if (sh.bounded) {
            unsigned long long flag = 0;
            PT_HIP(ctx, hipStreamSynchronize(st));
            PT_HIP(ctx, hipMemcpy(&flag, d_overflow, sizeof(flag), hipMemcpyDeviceToHost));
            redo = flag != 0ull;
        }
        if (!redo) {
            k_resolve<<<(rc.slots_per_lane + TB - 1) / TB, TB, 0, st>>>(rc, w.d_tiles, rad, f->d_rgb, f->d_bgra);
            ctx->stats.launches_other++;
        } else {  // a slot filled its term log (scenes where most surfaces emit): the same frames once more with one group
            PT_HIP(ctx, hipMemcpy(ctx->d_stats, &rays_before, sizeof(rays_before), hipMemcpyHostToDevice));
            PT_HIP(ctx, hipMemset(d_overflow, 0, sizeof(unsigned long long)));
            ctx->stats.redone_batches++;
            pt_params r = *p;
            r.frame = rc.frame_base; r.frame_count = rc.lanes_active; r.frames_in_flight = rc.lanes_active; r.sample_groups = 1;
            rc_ = render_fused(s, f, &r, pl, true, false);
            if (rc_ != PT_OK) return rc_;
            rc_ = ensure_work(f, p->rank, p->world, sh.lanes, sh.groups, sh.term_cap, sh.term_pcap, false);
            if (rc_ != PT_OK) return rc_;
            rad = { w.d_color, w.d_terms, w.d_terms_over, w.d_nterm, w.d_spill, w.d_spill_head, d_spill_count, spill_cap, d_overflow };
        }
    }
    if (!nested) PT_HIP(ctx, hipEventRecord(ctx->ev_b, st));
    PT_HIP(ctx, hipStreamSynchronize(st));
    PT_HIP(ctx, hipGetLastError());
    if (!nested) {
        float ms = 0.f;
        PT_HIP(ctx, hipEventElapsedTime(&ms, ctx->ev_a, ctx->ev_b));
        ctx->stats.ms_total += ms;
        ctx->stats.workspace_bytes = workspace_bytes(f);
        ctx->stats.paths += valid_local_pixels(f, p) * p->spp_per_frame * p->frame_count;
    }
    for (size_t i = 0; i + 1 < evs.size(); i += 2) {
        float a = 0.f;
        if (hipEventElapsedTime(&a, evs[i], evs[i + 1]) == hipSuccess) ctx->stats.ms_extend += a;
    }
    return PT_OK;
}

}  // namespace

void ptw_free_work(pt_film *f)
{
    pt_film::Work &w = f->work;
    (void)hipFree(w.d_tiles);
    (void)hipFree(w.d_color);
    (void)hipFree(w.d_terms);
    (void)hipFree(w.d_terms_over);
    (void)hipFree(w.d_nterm);
    (void)hipFree(w.d_spill_head);
    (void)hipFree(w.d_spill);
    for (int i = 0; i < 2; i++) {
        (void)hipFree(w.d_qid[i]);
        (void)hipFree(w.d_qstate[i]);
        (void)hipFree(w.d_qrayA[i]);
        (void)hipFree(w.d_qrayB[i]);
    }
    (void)hipFree(w.d_hit);
    (void)hipFree(w.d_hit_inst);
    (void)hipFree(w.d_count);
    (void)hipFree(w.d_sort);
    (void)hipFree(w.d_sq_rayA); (void)hipFree(w.d_sq_rayB); (void)hipFree(w.d_sq_contrib); (void)hipFree(w.d_sq_slot);
    (void)hipFree(w.d_sq_tmax); (void)hipFree(w.d_sq_hit); (void)hipFree(w.d_sq_count);
    w = pt_film::Work{};
}

pt_status ptw_prepare(pt_scene *s, pt_film *f, const pt_params *p)
{
    pt_status rc_ = check_params(s, f, p);
    if (rc_ != PT_OK) return rc_;
    ExtendPlan pl;
    rc_ = plan_extend(s, p->extend, pl);
    if (rc_ != PT_OK) return rc_;
    if (p->pipeline == PT_PIPELINE_FUSED) return render_fused(s, f, p, pl, false, true);
    RenderShape sh;
    rc_ = shape_and_work(f, p, sh, s->n_inst || pl.variant == PT_EXTEND_FLAT ? 0 : pl.lds_scene ? 2 : 1);
    s->ctx->stats.frames_in_flight = sh.lanes;
    s->ctx->stats.sample_groups = sh.groups;
    if (rc_ != PT_OK) return rc_;
    s->ctx->stats.workspace_bytes = workspace_bytes(f);
    // the other one-time objects of a render: pipeline streams, fork / join events and, with PT_FLAG_PROFILE, the
    // pooled (start, stop) events of every extend / shade launch of one batch (4 per round and pipeline)
    pt_ctx *ctx = s->ctx;
    for (int k = 1; k < 3; k++)
        if (!ctx->pipe_stream[k]) {
            PT_HIP(ctx, hipStreamCreateWithFlags(&ctx->pipe_stream[k], hipStreamNonBlocking));
            PT_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_join[k], hipEventDisableTiming));
        }
    if (!ctx->ev_fork) PT_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming));
    if (!ctx->h_poll) PT_HIP(ctx, hipHostMalloc((void **)&ctx->h_poll, sizeof(uint32_t) * 2 * PT_MAX_PIPES, hipHostMallocDefault));
    for (int k = 0; k < 3; k++)
        for (int j = 0; j < 2; j++)
            if (!ctx->ev_poll[k][j]) PT_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_poll[k][j], hipEventDisableTiming));
    for (int k = 0; k < 3; k++)
        if (!ctx->ev_shade[k]) PT_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_shade[k], hipEventDisableTiming));
    if (p->flags & PT_FLAG_PROFILE) {
        const size_t batches = ((size_t)p->frame_count + sh.lanes - 1) / sh.lanes;
        const size_t want = std::min<size_t>(4ull * sh.group_size * p->max_depth * 3ull * batches, 1u << 16);
        while (ctx->ev_pool.size() < want) {
            hipEvent_t e = nullptr;
            PT_HIP(ctx, hipEventCreate(&e));
            ctx->ev_pool.push_back(e);
        }
    }
    return PT_OK;
}

static pt_status render_impl(pt_scene *s, pt_film *f, const pt_params *p, bool nested)
{
    pt_ctx *ctx = s->ctx;
    hipStream_t st = ctx->stream;
    pt_status rc_ = check_params(s, f, p);
    if (rc_ != PT_OK) return rc_;
    ExtendPlan pl;
    rc_ = plan_extend(s, p->extend, pl);
    if (rc_ != PT_OK) return rc_;
    if (p->pipeline == PT_PIPELINE_FUSED) return render_fused(s, f, p, pl, nested, false);
    RenderShape sh;
    rc_ = shape_and_work(f, p, sh, s->n_inst || pl.variant == PT_EXTEND_FLAT ? 0 : pl.lds_scene ? 2 : 1);
    if (rc_ != PT_OK) return rc_;
    const uint32_t lanes = sh.lanes, groups = sh.groups, group_size = sh.group_size, term_cap = sh.term_cap;
    if (!nested) {
        ctx->stats.frames_in_flight = lanes;
        ctx->stats.sample_groups = groups;
    }
    pt_film::Work &w = f->work;
    unsigned long long *const d_overflow = ctx->d_stats + 6, *const d_spill_count = ctx->d_stats + 7;
    uint32_t spill_cap = sh.bounded ? SPILL_POOL_ENTRIES : 0u;  // worst-case logs never reach the pool
    if (ctx->tune.term_spill >= 0) spill_cap = std::min<uint32_t>(spill_cap, (uint32_t)ctx->tune.term_spill);  // tests
    Radiance rad = { w.d_color, w.d_terms, w.d_terms_over, w.d_nterm, w.d_spill, w.d_spill_head, d_spill_count, spill_cap, d_overflow };

    RenderConst rc = make_render_const(p, w, sh);

    const bool profile = !nested && (p->flags & PT_FLAG_PROFILE) != 0;  // (a redo would re-record the pooled events of its caller)
    const bool count_visits = (p->flags & PT_FLAG_COUNT_VISITS) != 0;
    const bool async = (p->flags & PT_FLAG_ASYNC) != 0;
    if (async && profile) { ctx->err = "PT_FLAG_ASYNC and PT_FLAG_PROFILE exclude each other"; return PT_ERR_INVALID_ARG; }
    std::vector<hipEvent_t> ev_extend, ev_shade;  // (start, stop) pairs filled in by the launches
    size_t ev_used = 0;
    auto new_event = [&]() -> hipEvent_t {  // from the context's pool: creating ~2000 events per call showed in the wall time
        if (ev_used == ctx->ev_pool.size()) {
            hipEvent_t e = nullptr;
            (void)hipEventCreate(&e);
            ctx->ev_pool.push_back(e);
        }
        return ctx->ev_pool[ev_used++];
    };

    // measured on MI355X: 4 paths per thread (one queue-tail atomic per 1024 paths) and 8 blocks per CU;
    // 1 path/thread is 40 % slower, 2 equal, grid size flat between 4 and 16 blocks per CU
    const int shade_grid = ctx->num_cus * 8;
    const size_t shade_smem = sizeof(float4) * 8 * (size_t)s->n_tris;  // tri4 + shade4 + the tangent frames
    const bool shade_lds = shade_smem <= 16 * 1024 && !pl.bvh8;  // per-triangle tables of small scenes are staged in LDS (in the BVH4's order)
    // instanced scenes: world-space normal + tangent per (instance, triangle), built once (lbvh_build.hip); pt_tuning.inst_frames = 0
    // keeps the per-hit transform
    if (s->n_inst && !pl.bvh8 && ctx->tune.inst_frames != 0) {
        const pt_status rcf = ptb_ensure_inst_frames(s);
        if (rcf != PT_OK) return rcf;
    }
    const float4 *inst_frame = s->n_inst && !pl.bvh8 && ctx->tune.inst_frames != 0 ? s->d_inst_frame : nullptr;
    // Several pipelines on separate streams: the slot lanes of a batch are split into parts that run their
    // rounds independently, so the VALU-bound extend of one overlaps the HBM-bound shade of another
    // (measured on MI355X, Cornell box: 1 pipeline 13.4, 2: 15.2, 3: 15.0, 4: 14.1 Grays/s; restricting the
    // kernels' blocks per CU to leave room for each other never helped).  Small batches keep one pipeline.
    struct Pipe {
        hipStream_t st;
        uint32_t slot_begin, n_slots;
        QueueView qv[2];
        float4 *hit;
        uint32_t *hit_inst;
        uint32_t *count;  // [2] queue sizes of this pipeline
        int cur;
        bool done;
        int polls;        // live-count polls queued on this pipeline's stream in the current batch
    };
    // two pipelines overlap one's traversal with the other's shading.  A third: C2 +2 % on one box and -1 % on another
    // (interleaved repetitions), C4 -3 %, C5 -3 %, C5x -5 %; a fourth loses everywhere.  Capping the persistent extend
    // grid below the register-file limit (PT_TUNE_EXTEND_BLOCKS) so that k_shade of the other pipeline can be co-resident
    // changes nothing measurable (C2, 7 -> 5 blocks per CU: within +-1 %).
    // Round 3, with the pipelines free-running (no stream is drained inside a batch any more): two pipelines can still settle
    // with traversal beside traversal and shade beside shade for a whole process (C2, K = 8 ... 16: 22.3-23.0 Grays/s in one
    // pass of a box, 24.0-24.6 in the next); three, held in rotation by the shade rule (below), cannot: C2 at K = 16 25.8-27.1
    // against 23.2-24.6, K = 8 +6 %, K = 4 +2.3 %, K = 2 +1.7 %, K = 1 equal -- where both kernels keep their tables in LDS and
    // the scene has one level.  Two stay on instanced scenes (C4 at K = 8: -1 % with three) and with the tables in HBM (C5 -6 %)
    // (profiles/r03v_c2_pipes.log, r03w_pipes_by_shape.log, r03aj_shade_rule_other_shapes.log).
    const bool three = shade_lds && !s->n_inst && (uint64_t)w.n_slots >= (24ull << 20);
    int n_pipes = three ? 3 : ((uint64_t)w.n_slots >= (4ull << 20) ? 2 : 1);
    n_pipes = pt_tuned(ctx->tune.pipes, n_pipes, 1, PT_MAX_PIPES);
    n_pipes = std::max(1, std::min(n_pipes, std::min<int>(PT_MAX_PIPES, (int)(lanes * groups))));
    ctx->stats.pipelines = (uint32_t)n_pipes;
    for (int k = 1; k < n_pipes; k++)
        if (!ctx->pipe_stream[k]) {
            PT_HIP(ctx, hipStreamCreateWithFlags(&ctx->pipe_stream[k], hipStreamNonBlocking));
            PT_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_join[k], hipEventDisableTiming));
        }
    if (n_pipes > 1 && !ctx->ev_fork) PT_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming));
    // (two pipelines start half a round apart, below; three start together and keep the shade rule, below)
    const bool stagger = ctx->tune.stagger < 0 ? n_pipes == 2 : ctx->tune.stagger == 1;
    if (!ctx->h_poll) PT_HIP(ctx, hipHostMalloc((void **)&ctx->h_poll, sizeof(uint32_t) * 2 * PT_MAX_PIPES, hipHostMallocDefault));
    for (int k = 0; k < n_pipes; k++)
        for (int j = 0; j < 2; j++)
            if (!ctx->ev_poll[k][j]) PT_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_poll[k][j], hipEventDisableTiming));
    for (int k = 0; k < n_pipes; k++)
        if (!ctx->ev_shade[k]) PT_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_shade[k], hipEventDisableTiming));

    // ray sorting (ray_sort.hip): the HBM kernels only; AUTO when the traversal working set does not fit the Infinity Cache
    bool sort_rays = false;
    if ((pl.variant == PT_EXTEND_HBM || pl.variant == PT_EXTEND_HBM8) && !s->n_inst) {
        const uint64_t working_set = pl.bvh8 ? 64ull * s->n_wide8 + 64ull * s->n_tris
                                             : 64ull * (pl.topdown4 ? s->n_wide16t : s->n_wide) + 48ull * s->n_tris;
        sort_rays = working_set > (256ull << 20);
        if (p->flags & PT_FLAG_SORT_RAYS) sort_rays = true;
        if (p->flags & PT_FLAG_NO_SORT_RAYS) sort_rays = false;
    }
    // 4 bits per axis + octant = 15-bit keys = two 8-bit passes (C5x: 6 bits, three passes: +0 %, 4 bits: +3.5 %)
    const int sort_bits = pt_tuned(ctx->tune.sort_bits, 4, 1, 9);
    if (sort_rays) {
        // one scratch area per pipeline, sized for that pipeline's share of the slots (+ slack for the uneven split)
        const size_t per_pipe = ptw_ray_sort_bytes((size_t)w.n_slots / (size_t)n_pipes + (size_t)rc.slots_per_lane + 1);
        const size_t need = per_pipe * (size_t)n_pipes;
        if (need > w.sort_bytes) {
            (void)hipFree(w.d_sort);
            w.d_sort = nullptr;
            w.bytes -= w.sort_bytes;
            w.sort_bytes = 0;
            const hipError_t e = hipMalloc(&w.d_sort, need);
            if (e != hipSuccess) { (void)hipGetLastError(); sort_rays = false; }  // no room: render unsorted
            else { w.sort_bytes = need; w.bytes += need; }
        }
    }
    const bool nee = p->pipeline == PT_PIPELINE_WAVEFRONT_NEE;
    // the emitters the NEE pipeline samples: the scene's, or -- instanced -- every instance's copy of them in world space
    if (nee && s->n_inst) {
        const pt_status rcl = ptb_ensure_inst_lights(s);
        if (rcl != PT_OK) return rcl;
    }
    const float4 *const nee_lights = s->n_inst ? s->d_lights_inst : s->d_lights;
    const uint32_t nee_n_lights = s->n_inst ? s->n_lights_inst : s->n_lights;
    const float nee_light_area = s->n_inst ? s->light_area_inst : s->light_area;
    if (nee && (size_t)w.n_slots > w.cap_sq) {  // the shadow queue: at most one entry per live path and round
        (void)hipFree(w.d_sq_rayA); (void)hipFree(w.d_sq_rayB); (void)hipFree(w.d_sq_contrib); (void)hipFree(w.d_sq_slot);
        (void)hipFree(w.d_sq_tmax); (void)hipFree(w.d_sq_hit);
        w.d_sq_rayA = w.d_sq_contrib = w.d_sq_hit = nullptr; w.d_sq_rayB = nullptr; w.d_sq_slot = nullptr; w.d_sq_tmax = nullptr;
        w.cap_sq = 0;
        const size_t ns = w.n_slots;
        PT_HIP(ctx, hipMalloc((void **)&w.d_sq_rayA, sizeof(float4) * ns));
        PT_HIP(ctx, hipMalloc((void **)&w.d_sq_rayB, sizeof(float2) * ns));
        PT_HIP(ctx, hipMalloc((void **)&w.d_sq_contrib, sizeof(float4) * ns));
        PT_HIP(ctx, hipMalloc((void **)&w.d_sq_slot, sizeof(uint32_t) * ns));
        PT_HIP(ctx, hipMalloc((void **)&w.d_sq_tmax, sizeof(float) * ns));
        PT_HIP(ctx, hipMalloc((void **)&w.d_sq_hit, sizeof(float4) * ns));
        if (!w.d_sq_count) PT_HIP(ctx, hipMalloc((void **)&w.d_sq_count, sizeof(uint32_t) * PT_MAX_PIPES));
        w.cap_sq = ns;
    }
    ctx->stats.extend_variant = pl.variant;
    if (!nested) PT_HIP(ctx, hipEventRecord(ctx->ev_a, st));
    if (w.n_slots > 0) {
        for (uint32_t done = 0; done < p->frame_count; done += lanes) {
            rc.frame_base = p->frame + (int32_t)done;
            rc.lanes_active = std::min(lanes, p->frame_count - done);
            unsigned long long rays_before = 0;
            if (sh.bounded) {  // the exact ray counter as it is before this batch, should the batch have to be redone
                PT_HIP(ctx, hipStreamSynchronize(st));
                PT_HIP(ctx, hipMemcpy(&rays_before, ctx->d_stats, sizeof(rays_before), hipMemcpyDeviceToHost));
            }
            PT_HIP(ctx, hipMemsetAsync(w.d_count, 0, sizeof(uint32_t) * 2 * PT_MAX_PIPES, st));
            if (sh.bounded) PT_HIP(ctx, hipMemsetAsync(d_spill_count, 0, sizeof(unsigned long long), st));
            const uint32_t slot_lanes = rc.lanes_active * groups;
            const int pipes_now = std::min<int>(n_pipes, (int)slot_lanes);
            Pipe pipe[PT_MAX_PIPES];
            for (int k = 0; k < pipes_now; k++) {
                const uint32_t l0 = (uint32_t)((uint64_t)slot_lanes * k / pipes_now);
                const uint32_t l1 = (uint32_t)((uint64_t)slot_lanes * (k + 1) / pipes_now);
                Pipe &pp = pipe[k];
                pp.st = k == 0 ? st : ctx->pipe_stream[k];
                pp.slot_begin = l0 * rc.slots_per_lane;
                pp.n_slots = (l1 - l0) * rc.slots_per_lane;
                for (int i = 0; i < 2; i++)
                    pp.qv[i] = { w.d_qid[i] + pp.slot_begin, w.d_qstate[i] + pp.slot_begin,
                                 w.d_qrayA[i] + pp.slot_begin, w.d_qrayB[i] + pp.slot_begin };
                pp.hit = w.d_hit + pp.slot_begin;
                pp.hit_inst = w.d_hit_inst + pp.slot_begin;
                pp.count = w.d_count + 2 * k;
                pp.cur = 0;
                pp.done = false;
                pp.polls = 0;
            }
            if (pipes_now > 1) {  // the other streams start after the counters are cleared
                PT_HIP(ctx, hipEventRecord(ctx->ev_fork, st));
                for (int k = 1; k < pipes_now; k++) PT_HIP(ctx, hipStreamWaitEvent(ctx->pipe_stream[k], ctx->ev_fork, 0));
            }
            for (int k = 0; k < pipes_now; k++) {
                Pipe &pp = pipe[k];
                const int gen_grid = (int)std::min<uint32_t>((pp.n_slots + 4 * TB - 1) / (4 * TB), (uint32_t)ctx->num_cus * 16u);
                k_generate<<<gen_grid, TB, 0, pp.st>>>(rc, w.d_tiles, pp.slot_begin, pp.n_slots, rad, pp.qv[0], &pp.count[0]);
                ctx->stats.launches_other++;
            }
            const uint32_t max_rounds = group_size * p->max_depth;  // every sample of a slot at full depth
            bool shade_recorded[PT_MAX_PIPES] = {};
            const bool shade_rule = pipes_now == 3 && ((ctx->tune.stagger < 0 && shade_lds) || ctx->tune.stagger == 2);
            for (uint32_t round = 0; round < max_rounds; round++) {
                for (int k = 0; k < pipes_now; k++) {
                    Pipe &pp = pipe[k];
                    if (pp.done) continue;
                    const int cur = pp.cur;
                    hipEvent_t x0 = nullptr, x1 = nullptr, h0 = nullptr, h1 = nullptr;
                    if (profile) { x0 = new_event(); x1 = new_event(); h0 = new_event(); h1 = new_event(); }
                    const uint32_t *perm = nullptr;
                    if (sort_rays && round > 0) {  // (round 0 is the primary rays: one origin, generated tile by tile)
                        const size_t per_pipe = w.sort_bytes / (size_t)n_pipes;
                        perm = ptw_sort_rays(pp.st, pp.qv[cur].rayA, pp.qv[cur].rayB, &pp.count[cur], pp.n_slots, s->bmin, s->bmax, sort_bits,
                                             ctx->num_cus, static_cast<char *>(w.d_sort) + per_pipe * (size_t)k);
                        ctx->stats.launches_other += 1 + 5 * (uint32_t)((3 * sort_bits + 3 + 7) / 8);
                    }
                    // The pipelines start half a round apart: pipeline k > 0 begins its first traversal launch when pipeline
                    // k-1's first one has finished.  Started together they can lock in phase -- traversal beside traversal, shade
                    // beside shade, nothing overlaps what it should -- and whether they do depended on the box: same-box A/B,
                    // five rounds each, 22.45 -> 23.99 Grays/s on a box whose runs were all slow (21.95 ... 23.28) and no
                    // change on one whose runs were all fast (24.1 vs 23.9); C4 unchanged (PT_TUNE_STAGGER=0: off;
                    // profiles/r02i_ab_stagger.log).  Only where the two kernels are of similar length (tables in LDS) and more
                    // than one frame is in flight: the waiting pipeline costs C5 1 % (its traversal launches are four times its
                    // shade launches: nothing to interleave) and a single frame 0.1 ms of latency.  A strict token (one
                    // traversal launch at a time) is 20 % slower: the tail of one traversal launch is where the next one's
                    // blocks start.
                    const bool stag = stagger && shade_lds && rc.lanes_active > 1 && round == 0;
                    if (stag && k > 0) PT_HIP(ctx, hipStreamWaitEvent(pp.st, ctx->ev_fork, 0));
                    launch_extend(pl, s, pp.qv[cur].rayA, pp.qv[cur].rayB, pp.hit, pp.hit_inst, &pp.count[cur], &pp.count[cur ^ 1],
                                  ctx->d_stats, p->tmin, p->tmax, count_visits, true, pp.st, k, x0, x1, perm);
                    if (stag && k + 1 < pipes_now) PT_HIP(ctx, hipEventRecord(ctx->ev_fork, pp.st));
                    // The shade rule (three pipelines): never all three in their shade launch at once.  Pipeline k's shade launch
                    // waits for the end of the most recent shade launch of pipeline k + 1 -- the one a third of a rotation ahead,
                    // whose launch is long over when the three are evenly spread, so in the pattern the rule aims at nobody
                    // waits, and out of it the laggard is held back until the rotation is restored.  Free-running, the three
                    // spend 11-19 % of a frame shade beside shade beside shade (latency-bound, VALUs idle) and which pattern a
                    // process falls into is chance: 24.3-25.7 Grays/s over eight processes of one box; with the rule 25.8-27.1,
                    // mean +6.0 % (profiles/r03ag_c2_rules_distribution.log).  The same rule on the traversal launches, on both,
                    // one position further ahead (= one shade launch at a time), or with two / four pipelines: all slower
                    // (r03af_c2_two_of_three.log, r03ah_c2_shade_rule_variants.log).
                    const int ahead = (k + 1) % pipes_now;
                    if (shade_rule && shade_recorded[ahead]) PT_HIP(ctx, hipStreamWaitEvent(pp.st, ctx->ev_shade[ahead], 0));
                    ShadowQueue sq{};
                    uint32_t *sq_count = nullptr;
                    if (nee) {
                        sq = { w.d_sq_rayA + pp.slot_begin, w.d_sq_rayB + pp.slot_begin, w.d_sq_contrib + pp.slot_begin,
                               w.d_sq_tmax + pp.slot_begin, w.d_sq_slot + pp.slot_begin };
                        sq_count = w.d_sq_count + k;
                        PT_HIP(ctx, hipMemsetAsync(sq_count, 0, sizeof(uint32_t), pp.st));
                    }
#define PT_LAUNCH_SHADE(N, L, E)                                                                                               \
    if (s->n_inst) PT_LAUNCH_SHADE_I(N, L, E, true); else PT_LAUNCH_SHADE_I(N, L, E, false)
#define PT_LAUNCH_SHADE_I(N, L, E, I)                                                                                          \
    hipExtLaunchKernelGGL((k_shade<N, L, E, I>), dim3(shade_grid), dim3(TB), (uint32_t)((L) ? shade_smem : 0), pp.st, h0, h1, 0u, rc, \
                          w.d_tiles, s->d_tri4, s->d_shade4, s->n_tris, pp.hit, rad, pp.qv[cur], pp.qv[cur ^ 1],                \
                          &pp.count[cur], &pp.count[cur ^ 1], s->n_inst ? s->d_inst6 : nullptr, pp.hit_inst,                    \
                          pl.bvh8 ? s->d_shade64_8 : s->d_shade64, pl.bvh8 ? s->d_ke4_8 : s->d_ke4, nee_lights, nee_n_lights,      \
                          nee_light_area, sq, sq_count, s->d_frame4, inst_frame)
                    if (nee) {
                        if (shade_lds) { PT_LAUNCH_SHADE(PT_SHADE_ITEMS, true, true); }
                        else { PT_LAUNCH_SHADE(PT_SHADE_ITEMS, false, true); }
                        if (nee_n_lights) {
                            // the shadow rays of this round: any-hit queries with their own tmax, then the unoccluded terms
                            launch_extend(pl, s, sq.rayA, sq.rayB, w.d_sq_hit + pp.slot_begin, nullptr, sq_count, nullptr, ctx->d_stats,
                                          p->tmin, p->tmax, false, true, pp.st, k, nullptr, nullptr, nullptr, sq.tmax);
                            k_shadow_add<<<shade_grid, TB, 0, pp.st>>>(rc, rad, w.d_sq_hit + pp.slot_begin, sq.contrib, sq.slot, sq_count);
                            ctx->stats.launches_extend++;
                            ctx->stats.launches_other++;
                        }
                    } else if (shade_lds) { PT_LAUNCH_SHADE(PT_SHADE_ITEMS, true, false); }
                    else { PT_LAUNCH_SHADE(PT_SHADE_ITEMS, false, false); }
#undef PT_LAUNCH_SHADE
#undef PT_LAUNCH_SHADE_I
                    if (shade_rule) { PT_HIP(ctx, hipEventRecord(ctx->ev_shade[k], pp.st)); shade_recorded[k] = true; }
                    if (profile) {
                        ev_extend.push_back(x0); ev_extend.push_back(x1);
                        ev_shade.push_back(h0); ev_shade.push_back(h1);
                    }
                    ctx->stats.launches_extend++;
                    ctx->stats.launches_shade++;
                    pp.cur ^= 1;
                }
                ctx->stats.rounds++;
                // Every slot needs >= group_size rounds; after that the live counts are polled every eighth round so that a batch
                // whose paths have all ended stops early.  The host reads the count of the PREVIOUS poll -- eight rounds back --
                // while each stream still holds eight rounds of launches: waiting for the newest count drained both streams,
                // restarted the pipelines in phase (traversal beside traversal) and left one of them idle until the other had
                // caught up: 5.3 ms of the 147 ms of 16 C2 frames (kernel timeline, profiles/r03r_c2_timeline_before.txt).
                // A pipeline found empty has at most sixteen empty rounds queued behind it.
                if (!async && round + 1 >= group_size && ((round + 1) & 7u) == 0u && round + 1 < max_rounds) {
                    bool all_done = true;
                    for (int k = 0; k < pipes_now; k++) {
                        Pipe &pp = pipe[k];
                        if (pp.done) continue;
                        const int j = pp.polls & 1;
                        PT_HIP(ctx, hipMemcpyAsync(ctx->h_poll + 2 * k + j, &pp.count[pp.cur], sizeof(uint32_t), hipMemcpyDeviceToHost, pp.st));
                        PT_HIP(ctx, hipEventRecord(ctx->ev_poll[k][j], pp.st));
                        if (pp.polls > 0) {
                            PT_HIP(ctx, hipEventSynchronize(ctx->ev_poll[k][j ^ 1]));
                            if (ctx->h_poll[2 * k + (j ^ 1)] == 0u) pp.done = true;
                        }
                        pp.polls++;
                        if (!pp.done) all_done = false;
                    }
                    if (all_done) break;
                }
            }
            for (int k = 1; k < pipes_now; k++) {  // join before the resolve reads every pipeline's slots
                PT_HIP(ctx, hipEventRecord(ctx->ev_join[k], ctx->pipe_stream[k]));
                PT_HIP(ctx, hipStreamWaitEvent(st, ctx->ev_join[k], 0));
            }
            bool redo = false;
            if (sh.bounded) {  // did a slot fill its term log?
                unsigned long long flag = 0;
                PT_HIP(ctx, hipStreamSynchronize(st));
                PT_HIP(ctx, hipMemcpy(&flag, d_overflow, sizeof(flag), hipMemcpyDeviceToHost));
                redo = flag != 0ull;
            }
            if (!redo) {
                k_resolve<<<(rc.slots_per_lane + TB - 1) / TB, TB, 0, st>>>(rc, w.d_tiles, rad, f->d_rgb, f->d_bgra);
                ctx->stats.launches_other++;
            } else {
                // Rare (scenes where most surfaces emit): nothing of this batch has touched the film yet.  Put the ray
                // counter back, clear the flag and render the same frames with one slot per (frame, pixel) -- the plain
                // accumulator needs no log -- then return to this call's workspace shape.
                PT_HIP(ctx, hipMemcpy(ctx->d_stats, &rays_before, sizeof(rays_before), hipMemcpyHostToDevice));
                PT_HIP(ctx, hipMemset(d_overflow, 0, sizeof(unsigned long long)));
                ctx->stats.redone_batches++;
                pt_params q = *p;
                q.frame = rc.frame_base;
                q.frame_count = rc.lanes_active;
                q.frames_in_flight = rc.lanes_active;
                q.sample_groups = 1;
                rc_ = render_impl(s, f, &q, true);
                if (rc_ != PT_OK) return rc_;
                rc_ = ensure_work(f, p->rank, p->world, lanes, groups, term_cap, sh.term_pcap);
                if (rc_ != PT_OK) return rc_;
                rad = { w.d_color, w.d_terms, w.d_terms_over, w.d_nterm, w.d_spill, w.d_spill_head, d_spill_count, spill_cap, d_overflow };
            }
        }
    }
    if (!nested) PT_HIP(ctx, hipEventRecord(ctx->ev_b, st));
    if (!async) {
        PT_HIP(ctx, hipStreamSynchronize(st));
        PT_HIP(ctx, hipGetLastError());
        if (!nested) {
            float ms = 0.f;
            PT_HIP(ctx, hipEventElapsedTime(&ms, ctx->ev_a, ctx->ev_b));
            ctx->stats.ms_total += ms;
        }
    }
    if (!nested) {
        ctx->stats.workspace_bytes = workspace_bytes(f);
        ctx->stats.paths += valid_local_pixels(f, p) * p->spp_per_frame * p->frame_count;
    }
    if (profile) {
        for (size_t i = 0; i + 1 < ev_extend.size(); i += 2) {
            float a = 0.f, b = 0.f;
            if (hipEventElapsedTime(&a, ev_extend[i], ev_extend[i + 1]) == hipSuccess) ctx->stats.ms_extend += a;
            if (hipEventElapsedTime(&b, ev_shade[i], ev_shade[i + 1]) == hipSuccess) ctx->stats.ms_shade += b;
        }
    }
    return PT_OK;
}

pt_status ptw_render(pt_scene *s, pt_film *f, const pt_params *p) { return render_impl(s, f, p, false); }

pt_status ptw_trace(pt_scene *s, const float *rays6, uint32_t n, float tmin, float tmax, uint32_t extend, pt_hit *hits)
{
    pt_ctx *ctx = s->ctx;
    hipStream_t st = ctx->stream;
    if (n == 0) return PT_OK;
    ExtendPlan pl;
    pt_status rc_ = plan_extend(s, extend, pl);
    if (rc_ != PT_OK) return rc_;
    std::vector<float4> a(n);
    std::vector<float2> b(n);
    for (uint32_t i = 0; i < n; i++) {
        const float *r = rays6 + 6 * (size_t)i;
        a[i] = make_float4(r[0], r[1], r[2], r[3]);
        b[i] = make_float2(r[4], r[5]);
    }
    float4 *d_a = nullptr, *d_hit = nullptr;
    float2 *d_b = nullptr;
    uint32_t *d_cnt = nullptr, *d_hi = nullptr;
    pt_hit *d_out = nullptr;
    pt_status ret = PT_OK;
    auto fail = [&](hipError_t e, const char *what) {
        ctx->err = std::string(what) + ": " + hipGetErrorString(e);
        ret = PT_ERR_HIP;
    };
    hipError_t e;
    if ((e = hipMalloc((void **)&d_a, sizeof(float4) * n)) != hipSuccess) fail(e, "hipMalloc");
    if (ret == PT_OK && (e = hipMalloc((void **)&d_b, sizeof(float2) * n)) != hipSuccess) fail(e, "hipMalloc");
    if (ret == PT_OK && (e = hipMalloc((void **)&d_hit, sizeof(float4) * n)) != hipSuccess) fail(e, "hipMalloc");
    if (ret == PT_OK && (e = hipMalloc((void **)&d_out, sizeof(pt_hit) * n)) != hipSuccess) fail(e, "hipMalloc");
    if (ret == PT_OK && (e = hipMalloc((void **)&d_cnt, sizeof(uint32_t) * 2)) != hipSuccess) fail(e, "hipMalloc");
    if (ret == PT_OK && (e = hipMalloc((void **)&d_hi, sizeof(uint32_t) * n)) != hipSuccess) fail(e, "hipMalloc");
    if (ret == PT_OK) {
        (void)hipMemcpyAsync(d_a, a.data(), sizeof(float4) * n, hipMemcpyHostToDevice, st);
        (void)hipMemcpyAsync(d_b, b.data(), sizeof(float2) * n, hipMemcpyHostToDevice, st);
        const uint32_t cnt_head[2] = { n, 0u };
        (void)hipMemcpyAsync(d_cnt, cnt_head, sizeof(cnt_head), hipMemcpyHostToDevice, st);
        (void)hipEventRecord(ctx->ev_a, st);
        launch_extend(pl, s, d_a, d_b, d_hit, d_hi, d_cnt, nullptr, ctx->d_stats, tmin, tmax, false, false, st);
        (void)hipEventRecord(ctx->ev_b, st);
        k_hits_to_api<<<(n + TB - 1) / TB, TB, 0, st>>>(d_hit, pl.bvh8 ? s->d_tri4_8 : s->d_tri4, s->n_inst ? d_hi : nullptr, s->d_tlas_prim_of, n, d_out);
        (void)hipMemcpyAsync(hits, d_out, sizeof(pt_hit) * n, hipMemcpyDeviceToHost, st);
        if ((e = hipStreamSynchronize(st)) != hipSuccess) fail(e, "pt_trace");
        else if ((e = hipGetLastError()) != hipSuccess) fail(e, "pt_trace");
        float ms = 0.f;
        if (ret == PT_OK && hipEventElapsedTime(&ms, ctx->ev_a, ctx->ev_b) == hipSuccess) ctx->stats.ms_extend += ms;
        ctx->stats.launches_extend++;
    }
    (void)hipFree(d_a); (void)hipFree(d_b); (void)hipFree(d_hit); (void)hipFree(d_out); (void)hipFree(d_cnt); (void)hipFree(d_hi);
    return ret;
}
