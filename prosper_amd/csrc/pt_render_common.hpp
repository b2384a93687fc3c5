// pt_render_common.hpp — helpers shared by the render kernels (pt_kernels.hip, pt_wavefront.hip).
#pragma once

#include "pt_device.hpp"

namespace ppt
{

__device__ __forceinline__ uint32_t wave_sum(uint32_t v)
{
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

template <bool COUNT>
__device__ void flush_counters(const LaneCounters &c, unsigned long long *counters)
{
    if constexpr (COUNT)
    {
        // order = prosper_pt_counters fields
        const uint32_t vals[kCounterCount] = {
            c.paths, c.closestRays, c.shadowRays, c.nodeVisits, c.triangleTests, c.closestHits, c.anyHitCalls, c.lightSamples,
            c.spotLightSamples, c.skyLookups, c.pixelsWritten, c.historyReads, c.shortIndexHits, c.shortIndexTriangleTests,
            c.nodePhaseSteps, c.trianglePhaseSteps, c.anyHitTexelFetches};
        for (uint32_t i = 0; i < kCounterCount; ++i)
        {
            const uint32_t sum = wave_sum(vals[i]);
            if ((threadIdx.x & 63) == 0 && sum) atomicAdd(&counters[i], (unsigned long long)sum);
        }
    }
}

// Local pixel column -> absolute image column for the stripe partition (prosper_pt_tile_desc).
__device__ __forceinline__ uint32_t local_to_global_x(const RenderParams &p, uint32_t lx)
{
    if (p.stripeWidth == 0) return lx;
    const uint32_t ls = lx / p.stripeWidth;
    return (ls * p.stripeCount + p.stripeIndex) * p.stripeWidth + (lx % p.stripeWidth);
}

struct PathState
{
    Rng rng;
    f3 throughput;
    f3 color;
    f3 o, d;
    uint32_t bounce;
};

template <bool COUNT>
__device__ __forceinline__ void start_path(
    const RenderParams &p, uint32_t px, uint32_t py, uint32_t frameIndex, PathState &st, LaneCounters &cnt)
{
    st.rng = Rng{px, py, frameIndex};
    const f2 j = st.rng.rnd2d01();
    const f2 uv = f2{((float)px + j.x) / (float)p.width, ((float)py + j.y) / (float)p.height};
    Ray ray;
    if (p.pc.flags & PROSPER_PC_FLAG_DEPTH_OF_FIELD)
    {
        const f2 lens = st.rng.rnd2d01();
        ray = thin_lens_camera_ray(p, uv, lens);
    }
    else
        ray = pinhole_camera_ray(p, uv);
    st.o = ray.o;
    st.d = ray.d;
    st.throughput = f3{1.0f, 1.0f, 1.0f};
    st.color = f3{0.0f, 0.0f, 0.0f};
    st.bounce = 0;
    if constexpr (COUNT) cnt.paths++;
}

} // namespace ppt
