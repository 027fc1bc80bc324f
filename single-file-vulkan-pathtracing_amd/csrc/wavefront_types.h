// wavefront_types.h -- what the kernels of the wavefront pipeline and their host-side scheduler share: the per-render constants,
// where a slot's radiance goes, the path queues, slot -> pixel, and the block-wide ordered compaction.
//
// A *slot* is one (frame, sample group, pixel) triple; it runs its samples one after the other ("regeneration"), and its
// radiance is either accumulated in path order (one group) or logged term by term and replayed in order by k_resolve
// (several groups): either way the reference's single `color` accumulator (raygen.rgen:42, 76) is reproduced add for add.
// Live paths sit in dense, double-buffered queues (index = queue position, every access coalesced):
//     qid {slot, sample | depth << 16}, qstate {seed, weight}, qray {origin, direction}
// The structs are plain data in namespace ptw (they cross translation units as kernel and launcher arguments); the device
// helpers are inline.
#pragma once
#include "pt_internal.h"
#include "pt_math.h"

#ifndef PT_TB_DEFINED
#define PT_TB_DEFINED
namespace {
constexpr int TB = 256;                     // threads per block of every kernel of the library
constexpr uint32_t SENTINEL = 0xFFFFFFFFu;  // "no child" / "no node" in the BVH4 child words
}  // namespace
#endif

namespace ptw {

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
    // HEAD + TAIL slots (PT_PIPELINE_FUSED at few frames per launch, fused_kernel.h MODE 2; `tail` = 0 everywhere else).  A (frame, pixel) is ONE head
    // slot -- samples [0, head_samples), radiance added in LDS like a one-group slot, slot number frame_lane * slots_per_lane + local < n_head -- and
    // `tail` slots of one sample each -- sample head_samples + j, radiance terms logged like a sample group's, slot number
    // n_head + (frame_lane * tail + j) * slots_per_lane + local; the log arrays (nterm, terms, terms_over, spill_head) are indexed by
    // slot - n_head and hold n_tail slots.  The heads go first and are long; the tails fill the end of the launch with short work.
    uint32_t tail, head_samples, n_head, n_tail;
    FastDiv div_tail;
    // PT_PIPELINE_FUSED, single-level scenes: camera rays that cannot reach the scene.  cull_on: every primary ray of a pixel OUTSIDE the pixel
    // rectangle cull = {x0, y0, x1, y1} (the projection of the scene's box, a pixel of slack: render.hip fused_subject_rect) misses the box and
    // with it every triangle, so such a slot is finished where it is handed out: each of its samples is one ray (counted) whose miss adds
    // 1 * env (raygen.rgen:59, 76; miss.rmiss:10) -- cull_sum = that add done spp times, what a slot without a log stores (head + tail: the head slot
    // stands for all spp samples of such a pixel and k_resolve skips its tail logs; several groups: every group logs its samples' terms).
    uint32_t cull_on;
    int32_t cull[4];
    float cull_sum[3];
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
    // This record once more, in device memory (pt_ctx::d_rad; null for the wavefront kernels).  The fused kernels read the log's rarely taken ends
    // (terms_over, the pool) from it where they take them, so that those six pointers are not live in scalar registers through the persistent
    // loop -- with them the head + tail kernel spilled 170 scalar values into vector lanes and read them back with a v_readlane each, 64 of them
    // in the block that logs a term (one blocking frame -4 %, profiles/r06z_*)
    const Radiance *dev;
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

}  // namespace ptw
