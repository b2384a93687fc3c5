// pt_kernels.hpp — host-callable launchers of the gfx950 kernels in pt_kernels.hip.
#pragma once

#include <hip/hip_runtime.h>

#include "pt_scene.hpp"

namespace ppt
{

void launch_flatten_triangles(
    const DeviceScene &s, const uint32_t *triOffsets, uint32_t drawInstanceCount, const uint32_t *drawInstanceFlags,
    WorldTriangle *out, uint32_t total, hipStream_t stream);
void launch_permute_triangles(
    const WorldTriangle *in, const uint32_t *permutation, WorldTriangle *out, uint32_t total, hipStream_t stream);
void launch_render_megakernel(
    const DeviceScene &s, const RenderParams &p, float4 *hdr, unsigned long long *counters, bool countWork,
    hipStream_t stream);
void launch_render_persistent(
    const DeviceScene &s, const RenderParams &p, float4 *hdr, unsigned long long *counters, uint32_t *workCounter,
    bool countWork, hipStream_t stream);
void launch_render_wavefront(
    const DeviceScene &s, const RenderParams &p, float4 *hdr, unsigned long long *counters, const WavefrontBuffers &w,
    uint32_t bvhDepth, bool countWork, hipStream_t stream);
void launch_blit_rgba16f(const float4 *in, void *out, uint32_t count, hipStream_t stream);
void launch_eval_fn(
    uint32_t fn, const float *in, uint32_t inStride, float *out, uint32_t outStride, uint32_t n, hipStream_t stream);

} // namespace ppt
