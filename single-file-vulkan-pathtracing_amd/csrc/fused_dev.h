// fused_dev.h -- developer instrumentation of the single-level fused kernel (fused_kernel.h), kept out of the kernel's text.
//
// The PRODUCT build defines none of the macros below: `FusedDev` is then an empty struct whose methods are empty inline functions and the two
// `dbg_*` constants are false, so the kernel's ISA is the one without any of this (scripts/isa_cmp.py against the round-5 listing:
// profiles/r06a_fused_dev_isa.txt).  Dev builds (scripts/build_variant.sh NAME "-DPT_FUSED_TIMELINE" ...):
//   PT_FUSED_TIMELINE  per wave {start, first failed slot draw, end, rays} in device clock ticks and per lane its last completed slot with start / end
//                      (scripts/probe_fused_timeline.py)
//   PT_FUSED_HIST      rays started per 50 us of device clock over one launch, and the part started on tail slots (scripts/probe_fused_hist.py)
//   PT_DBG_NO_TERMS    timing experiment, WRONG IMAGES: the term log's stores are skipped (what they cost)
//   PT_DBG_NO_NTERM    timing experiment, WRONG IMAGES: the per-slot term count is not stored
// The wave-level block counters (template parameter COUNT of fused_body) are not dev code: PT_FLAG_COUNT_VISITS on the fused pipeline runs
// that instantiation and bench.py prints its table (pt_get_block_counts).
#pragma once

#ifdef PT_DBG_NO_TERMS
constexpr bool dbg_no_terms = true;
#else
constexpr bool dbg_no_terms = false;
#endif
#ifdef PT_DBG_NO_NTERM
constexpr bool dbg_no_nterm = true;
#else
constexpr bool dbg_no_nterm = false;
#endif

#if defined(PT_FUSED_TIMELINE) || defined(PT_FUSED_HIST)
struct FusedDev {
#ifdef PT_FUSED_TIMELINE
    unsigned long long tl_start = 0ull, tl_oos = 0ull;
    unsigned long long tl_slot_t0 = 0ull, tl_last_t0 = 0ull, tl_last_t1 = 0ull;  // per lane: when its current slot began; its last completed slot
    uint32_t tl_last_slot = 0xFFFFFFFFu, tl_n_slots = 0u;
#endif
#ifdef PT_FUSED_HIST
    uint32_t tl_hb = 0xFFFFFFFFu, tl_hn = 0u, tl_ht = 0u, tl_pass = 0u;  // the histogram bucket being counted, rays started in it (all / on tail slots)
#endif
    __device__ __forceinline__ void kernel_begin()
    {
#ifdef PT_FUSED_TIMELINE
        tl_start = wall_clock64();
#endif
    }
    __device__ __forceinline__ void slot_begin()
    {
#ifdef PT_FUSED_TIMELINE
        tl_slot_t0 = wall_clock64();
#endif
    }
    __device__ __forceinline__ void slot_end(uint32_t slot)
    {
#ifdef PT_FUSED_TIMELINE
        tl_last_slot = slot; tl_last_t0 = tl_slot_t0; tl_last_t1 = wall_clock64(); tl_n_slots++;
#endif
    }
    __device__ __forceinline__ void out_of_slots()
    {
#ifdef PT_FUSED_TIMELINE
        tl_oos = wall_clock64();
#endif
    }
    // once per shade block: `n` rays were set up by this wave (wave-uniform), `tails`: while it draws tail slots
    __device__ __forceinline__ void rays_started(uint32_t n, bool tails, int lane)
    {
#ifdef PT_FUSED_HIST
        // rays started per 50 us of device clock (100 MHz), [bucket][16 words by block]; the second half: those of waves drawing tail slots.
        // Counted in registers; the clock is read on every 8th pass only (a scalar memory read the wave waits for) and a bucket's count is
        // flushed when the bucket changes: one atomic per wave and bucket, spread over 16 words
        if ((tl_pass++ & 7u) == 0u) {
            const uint32_t b = (uint32_t)(wall_clock64() / 5000ull) & 8191u;
            if (b != tl_hb) {
                if (lane == 0 && g_fused_hist && tl_hn) {
                    atomicAdd(g_fused_hist + tl_hb * 16u + (blockIdx.x & 15u), tl_hn);
                    if (tl_ht) atomicAdd(g_fused_hist + (8192u + tl_hb) * 16u + (blockIdx.x & 15u), tl_ht);
                }
                tl_hb = b; tl_hn = 0u; tl_ht = 0u;
            }
        }
        tl_hn += n;
        if (tails) tl_ht += n;
#endif
    }
    __device__ __forceinline__ void kernel_end(int lane, uint32_t n_rays_wave, int tb)
    {
#ifdef PT_FUSED_HIST
        if (lane == 0 && g_fused_hist && tl_hn) {
            atomicAdd(g_fused_hist + (tl_hb & 8191u) * 16u + (blockIdx.x & 15u), tl_hn);
            if (tl_ht) atomicAdd(g_fused_hist + (8192u + (tl_hb & 8191u)) * 16u + (blockIdx.x & 15u), tl_ht);
        }
#endif
#ifdef PT_FUSED_TIMELINE
        if (lane == 0 && g_fused_timeline) {
            unsigned long long *o = g_fused_timeline + 4 * (size_t)(blockIdx.x * (tb / 64) + (threadIdx.x >> 6));
            o[0] = tl_start; o[1] = tl_oos; o[2] = wall_clock64(); o[3] = n_rays_wave;
        }
        if (g_fused_timeline) {  // per lane, after the per-wave records: {last slot | slots done << 32, its start, its end}
            unsigned long long *o = g_fused_timeline + 4 * (size_t)(gridDim.x * (tb / 64)) + 3 * (size_t)(blockIdx.x * tb + threadIdx.x);
            o[0] = (unsigned long long)tl_last_slot | ((unsigned long long)tl_n_slots << 32); o[1] = tl_last_t0; o[2] = tl_last_t1;
        }
#endif
    }
};
#else
struct FusedDev {
    __device__ __forceinline__ void kernel_begin() {}
    __device__ __forceinline__ void slot_begin() {}
    __device__ __forceinline__ void slot_end(uint32_t) {}
    __device__ __forceinline__ void out_of_slots() {}
    __device__ __forceinline__ void rays_started(uint32_t, bool, int) {}
    __device__ __forceinline__ void kernel_end(int, uint32_t, int) {}
};
#endif
