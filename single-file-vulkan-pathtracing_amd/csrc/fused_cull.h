// fused_cull.h -- slots of pixels that cannot see the scene (wavefront_types.h RenderConst::cull), shared by k_fused and k_fused_inst.
// Every camera ray of such a pixel misses the scene's box, hence every triangle: the sample is one ray (counted) whose miss adds 1 * env
// (raygen.rgen:59, 76; miss.rmiss:10).  The slot is finished where it is handed out, with exactly those adds.
#pragma once

namespace ptc {

__device__ __forceinline__ bool pixel_culled(const ptw::RenderConst &rc, uint32_t px, uint32_t py)
{
    return rc.cull_on && ((int32_t)px < rc.cull[0] || (int32_t)px > rc.cull[2] || (int32_t)py < rc.cull[1] || (int32_t)py > rc.cull[3]);
}

// a slot with one accumulator (one sample group; the head slot of a head + tail pixel, which then stands for all its samples): -> samples finished
__device__ __forceinline__ uint32_t finish_plain(const ptw::RenderConst &rc, const ptw::Radiance &rad, uint32_t slot)
{
    rad.color[slot] = make_float4(rc.cull_sum[0], rc.cull_sum[1], rc.cull_sum[2], 0.f);
    return rc.spp;
}

// a slot of sample group g, which logs its terms: one env term per sample (<= group_size <= term_pcap of them): -> samples finished
__device__ __forceinline__ uint32_t finish_group(const ptw::RenderConst &rc, const ptw::Radiance &rad, uint32_t slot, uint32_t g)
{
    const float4 e = make_float4(rc.env[0], rc.env[1], rc.env[2], 0.f);
    const uint32_t n = min(rc.spp, (g + 1u) * rc.group_size) - g * rc.group_size;
    for (uint32_t k = 0; k < n; k++) ptm::st_stream<true>(rad.terms + ((size_t)k * rc.n_slots + slot), e);
    rad.nterm[slot] = n;
    return n;
}

// (wave-uniform control flow) the rays of the slots the wave's lanes just finished: n_full per slot -- spp, or group_size with several groups,
// whose last group may be shorter
template <bool GROUPED>
__device__ __forceinline__ uint32_t rays_finished(const ptw::RenderConst &rc, uint32_t cull_n)
{
    const uint32_t n_full = GROUPED ? rc.group_size : rc.spp;
    uint32_t n = (uint32_t)__popcll(__ballot(cull_n == n_full)) * n_full;
    if (GROUPED) n += (uint32_t)__popcll(__ballot(cull_n != 0u && cull_n != n_full)) * (rc.spp - (rc.groups - 1u) * rc.group_size);
    return n;
}

}  // namespace ptc
