// pt_kernels.hip — gfx950 kernels of the path-tracing reference pass.
//
//   flatten_triangles   upload: bindless fp16 geometry x instance transforms -> world triangles
//   permute_triangles   upload: reorder world triangles into BVH leaf order
//   render_megakernel   one lane per pixel runs whole paths (rt/reference/main.rgen:225-299)
//   blit_rgba16f        RGBA32F -> RGBA16F (src/render/RtReference.cpp:339-377)
//   eval_fn             device self-test entry (prosper_pt_eval_device_fn)
#include "pt_kernels.hpp"

#include "bvh_encode.hpp"
#include "pt_bc7.hpp"
#include "pt_device.hpp"
#include "pt_render_common.hpp"

namespace ppt
{

// ------------------------------------------------------------------------------------------
// upload kernels
// ------------------------------------------------------------------------------------------

__global__ void flatten_triangles_kernel(
    DeviceScene s, const uint32_t *__restrict__ triOffsets, uint32_t drawInstanceCount,
    const uint32_t *__restrict__ drawInstanceFlags, WorldTriangle *__restrict__ out,
    ShadeTriangle *__restrict__ shadeOut, AlphaTriangle *__restrict__ alphaOut, uint32_t total,
    const uint32_t *__restrict__ leafPosition, WorldTriangle *__restrict__ leafOrder, RawShadeTriangle *__restrict__ rawOut)
{
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= total) return;
    // find the draw instance whose triangle range holds g
    uint32_t lo = 0, hi = drawInstanceCount;
    while (hi - lo > 1)
    {
        const uint32_t mid = (lo + hi) >> 1;
        if (triOffsets[mid] <= g)
            lo = mid;
        else
            hi = mid;
    }
    const uint32_t di = lo;
    const uint32_t prim = g - triOffsets[di];
    const prosper_DrawInstance inst = s.drawInstances[di];
    const prosper_GeometryMetadata m = s.geometryMetadatas[inst.meshIndex];
    const prosper_mat3x4 modelToWorld = s.modelInstanceTransforms[inst.modelInstanceIndex].modelToWorld;
    const uint32_t i0 = load_index(s, m, prim * 3 + 0);
    const uint32_t i1 = load_index(s, m, prim * 3 + 1);
    const uint32_t i2 = load_index(s, m, prim * 3 + 2);
    const f3 v0 = mul_point_mat3x4(load_r16g16b16a16(s, m.bufferIndex, m.positionsOffset, i0), modelToWorld);
    const f3 v1 = mul_point_mat3x4(load_r16g16b16a16(s, m.bufferIndex, m.positionsOffset, i1), modelToWorld);
    const f3 v2 = mul_point_mat3x4(load_r16g16b16a16(s, m.bufferIndex, m.positionsOffset, i2), modelToWorld);
    WorldTriangle t;
    t.v0[0] = v0.x; t.v0[1] = v0.y; t.v0[2] = v0.z;
    t.v1[0] = v1.x; t.v1[1] = v1.y; t.v1[2] = v1.z;
    t.v2[0] = v2.x; t.v2[1] = v2.y; t.v2[2] = v2.z;
    t.drawInstance = di;
    t.primitive = prim;
    const uint32_t diFlags = drawInstanceFlags[di];
    const bool nonOpaque = !(diFlags & kTriFlagOpaque);
    const uint32_t alphaIndex = nonOpaque ? s.alphaOffsets[di] + prim : 0u;
    t.flags = diFlags | (alphaIndex << kTriAlphaShift); // a non-opaque triangle names its alpha record
    out[g] = t;
    if (leafOrder != nullptr) leafOrder[leafPosition[g]] = t; // moved instances: straight into the traversal's (leaf) order
    if (shadeOut == nullptr && rawOut == nullptr) return; // re-flatten after moved instances: the shading records are object-space

    // the shading record of this triangle: decoded (pt_scene.hpp ShadeTriangle: geometry.glsl:220-244 per corner) or, for
    // scenes that keep the 64-byte form, the raw stream values (RawShadeTriangle)
    const uint32_t vi[3] = {i0, i1, i2};
    uint32_t uvBits[3];
    for (int c = 0; c < 3; ++c)
        uvBits[c] = m.texCoord0sOffset == PROSPER_PT_ABSENT ? 0u : geo_u32(s, m.bufferIndex)[m.texCoord0sOffset + vi[c]];
#ifdef PPT_EXPERIMENTS
    if (rawOut != nullptr)
    {
        RawShadeTriangle raw;
        for (int c = 0; c < 3; ++c)
        {
            if (m.positionsOffset == PROSPER_PT_ABSENT)
                raw.position[c][0] = raw.position[c][1] = 0u;
            else
            {
                const u32x2 pp = *(global_u32x2_ptr)(geo_u32(s, m.bufferIndex) + m.positionsOffset + vi[c] * 2);
                raw.position[c][0] = pp.x;
                raw.position[c][1] = pp.y;
            }
            raw.normal[c] = m.normalsOffset == PROSPER_PT_ABSENT ? 0u : geo_u32(s, m.bufferIndex)[m.normalsOffset + vi[c]];
            raw.tangent[c] = m.tangentsOffset == PROSPER_PT_ABSENT ? 0u : geo_u32(s, m.bufferIndex)[m.tangentsOffset + vi[c]];
            raw.uv[c] = uvBits[c];
        }
        raw.flags = (diFlags & kTriFlagShortIndices) | (m.normalsOffset == PROSPER_PT_ABSENT ? kRawNoNormals : 0u) |
                    (m.tangentsOffset == PROSPER_PT_ABSENT ? kRawNoTangents : 0u);
        rawOut[g] = raw;
    }
#endif
    if (shadeOut != nullptr)
    {
        ShadeTriangle sh;
        for (int c = 0; c < 3; ++c)
        {
            const Vertex vtx = load_vertex_through_index_buffer(s, m, prim * 3 + c);
            sh.normalUv[c][0] = vtx.normal.x;
            sh.normalUv[c][1] = vtx.normal.y;
            sh.normalUv[c][2] = vtx.normal.z;
            sh.normalUv[c][3] = __builtin_bit_cast(float, uvBits[c]);
            sh.tangent[c][0] = vtx.tangent.x;
            sh.tangent[c][1] = vtx.tangent.y;
            sh.tangent[c][2] = vtx.tangent.z;
            sh.tangent[c][3] = vtx.tangent.w;
            if (m.positionsOffset == PROSPER_PT_ABSENT)
                sh.position[c][0] = sh.position[c][1] = 0u;
            else
            {
                const u32x2 pp = *(global_u32x2_ptr)(geo_u32(s, m.bufferIndex) + m.positionsOffset + vi[c] * 2);
                sh.position[c][0] = pp.x;
                sh.position[c][1] = pp.y;
            }
        }
        sh.flags = diFlags;
        sh.reserved = 0;
        shadeOut[g] = sh;
    }
    if (nonOpaque)
    {
        // what the any-hit shader reads of this triangle (pt_scene.hpp AlphaTriangle): scene.rahit:20-31
        AlphaTriangle at;
        for (int c = 0; c < 3; ++c) at.uv[c] = uvBits[c];
        at.drawInstance = di;
        at.primitive = prim;
        at.materialIndex = inst.materialIndex;
        at.reserved[0] = at.reserved[1] = 0u;
        at.material = s.alphaMaterials[inst.materialIndex];
        alphaOut[alphaIndex] = at;
    }
}

__global__ void permute_triangles_kernel(
    const WorldTriangle *__restrict__ in, const uint32_t *__restrict__ permutation, WorldTriangle *__restrict__ out,
    uint32_t total)
{
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= total) return;
    const float4 *src = reinterpret_cast<const float4 *>(in + permutation[g]);
    float4 *dst = reinterpret_cast<float4 *>(out + g);
    dst[0] = src[0];
    dst[1] = src[1];
    dst[2] = src[2];
}

void launch_flatten_triangles(
    const DeviceScene &s, const uint32_t *triOffsets, uint32_t drawInstanceCount, const uint32_t *drawInstanceFlags,
    WorldTriangle *out, ShadeTriangle *shadeOut, AlphaTriangle *alphaOut, uint32_t total, hipStream_t stream,
    const uint32_t *leafPosition, WorldTriangle *leafOrder, RawShadeTriangle *rawOut)
{
    if (total == 0) return;
    hipLaunchKernelGGL(
        flatten_triangles_kernel, dim3((total + 255) / 256), dim3(256), 0, stream, s, triOffsets, drawInstanceCount,
        drawInstanceFlags, out, shadeOut, alphaOut, total, leafPosition, leafOrder, rawOut);
}

void launch_permute_triangles(
    const WorldTriangle *in, const uint32_t *permutation, WorldTriangle *out, uint32_t total, hipStream_t stream)
{
    if (total == 0) return;
    hipLaunchKernelGGL(
        permute_triangles_kernel, dim3((total + 255) / 256), dim3(256), 0, stream, in, permutation, out, total);
}

// ---- refit: new boxes for an unchanged tree after instances moved (prosper rebuilds its TLAS every frame,
//      World.cpp:749-802,878-928; here the per-frame cost is two small kernels per tree level) ----
//
// bounds[i] = the exact bounds of the triangles below node i (two float4: lo, hi).  refit_bounds runs level by level from
// the leaves' parents up (`order` lists the nodes by height); encode_nodes then rewrites every node's origin and child
// boxes with the emitter's own code (bvh_encode.hpp), the slack taken from the root's new bounds.

__device__ __forceinline__ EncBox leaf_bounds(const WorldTriangle *__restrict__ tris, int32_t ref)
{
    const uint32_t leaf = (uint32_t)~ref;
    const uint32_t first = leaf >> 3, count = (leaf & 7u) + 1u;
    EncBox b;
    for (int a = 0; a < 3; ++a)
    {
        b.lo[a] = kInf;
        b.hi[a] = -kInf;
    }
    for (uint32_t i = 0; i < count; ++i)
    {
        const WorldTriangle t = tris[first + i];
        for (int a = 0; a < 3; ++a)
        {
            b.lo[a] = enc_min(b.lo[a], enc_min(t.v0[a], enc_min(t.v1[a], t.v2[a])));
            b.hi[a] = enc_max(b.hi[a], enc_max(t.v0[a], enc_max(t.v1[a], t.v2[a])));
        }
    }
    return b;
}
__device__ __forceinline__ EncBox child_bounds(
    const BvhNode &n, uint32_t c, const WorldTriangle *__restrict__ tris, const float4 *__restrict__ bounds)
{
    if (n.child[c] < 0) return leaf_bounds(tris, n.child[c]);
    const float4 lo = bounds[2 * (size_t)n.child[c]], hi = bounds[2 * (size_t)n.child[c] + 1];
    return EncBox{{lo.x, lo.y, lo.z}, {hi.x, hi.y, hi.z}};
}

__global__ __launch_bounds__(256) void refit_bounds_kernel(
    const BvhNode *__restrict__ nodes, const WorldTriangle *__restrict__ tris, float4 *__restrict__ bounds,
    const uint32_t *__restrict__ order, uint32_t first, uint32_t count)
{
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= count) return;
    const uint32_t i = order[first + g];
    const BvhNode n = nodes[i];
    EncBox u;
    for (int a = 0; a < 3; ++a)
    {
        u.lo[a] = kInf;
        u.hi[a] = -kInf;
    }
    for (uint32_t c = 0; c < n.reserved && c < 4u; ++c)
    {
        const EncBox b = child_bounds(n, c, tris, bounds);
        for (int a = 0; a < 3; ++a)
        {
            u.lo[a] = enc_min(u.lo[a], b.lo[a]);
            u.hi[a] = enc_max(u.hi[a], b.hi[a]);
        }
    }
    bounds[2 * (size_t)i] = make_float4(u.lo[0], u.lo[1], u.lo[2], 0.0f);
    bounds[2 * (size_t)i + 1] = make_float4(u.hi[0], u.hi[1], u.hi[2], 0.0f);
}

// `cost` (optional): sum over the nodes of the half-areas of their inner children's boxes - the surface-area measure of how
// many node visits a ray through the scene makes (absolute, not relative to the root's box: an instance that leaves the
// scene's old bounds must not hide the growth of the nodes it stretches); the host compares it with the value right
// after the last build to notice a tree that moved instances have degraded (a heuristic: the float sum is not
// deterministic).
__global__ __launch_bounds__(256) void encode_nodes_kernel(
    BvhNode *__restrict__ nodes, const WorldTriangle *__restrict__ tris, const float4 *__restrict__ bounds,
    uint32_t nodeCount, float padCoeff, float *__restrict__ cost)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nodeCount) return;
    BvhNode n = nodes[i];
    const uint32_t k = n.reserved < 4u ? n.reserved : 4u;
    const float4 rlo = bounds[0], rhi = bounds[1];
    const EncBox scene{{rlo.x, rlo.y, rlo.z}, {rhi.x, rhi.y, rhi.z}};
    EncBox boxes[4];
    float area = 0.0f;
    for (uint32_t c = 0; c < k; ++c)
    {
        boxes[c] = child_bounds(n, c, tris, bounds);
        if (n.child[c] >= 0)
        {
            const float dx = boxes[c].hi[0] - boxes[c].lo[0], dy = boxes[c].hi[1] - boxes[c].lo[1], dz = boxes[c].hi[2] - boxes[c].lo[2];
            area += dx * dy + dy * dz + dz * dx;
        }
    }
    enc_node_boxes(boxes, k, padCoeff, enc_slack(scene), n);
    nodes[i] = n;
    if (cost != nullptr)
    {
        float v = area;
        for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
        if ((threadIdx.x & 63u) == 0u && v != 0.0f) atomicAdd(cost, v);
    }
}

void launch_refit(
    BvhNode *nodes, const WorldTriangle *tris, float4 *bounds, const uint32_t *order, const uint32_t *levelOffsets,
    uint32_t levels, uint32_t nodeCount, float padCoeff, float *cost, hipStream_t stream)
{
    for (uint32_t l = 0; l < levels; ++l)
    {
        const uint32_t first = levelOffsets[l], count = levelOffsets[l + 1] - first;
        if (count == 0) continue;
        hipLaunchKernelGGL(refit_bounds_kernel, dim3((count + 255u) / 256u), dim3(256), 0, stream, nodes, tris, bounds, order, first, count);
    }
    if (nodeCount)
        hipLaunchKernelGGL(encode_nodes_kernel, dim3((nodeCount + 255u) / 256u), dim3(256), 0, stream, nodes, tris, bounds, nodeCount, padCoeff, cost);
}

// ------------------------------------------------------------------------------------------
// path tracing
// ------------------------------------------------------------------------------------------

// One path: rt/reference/main.rgen:225-283.
template <bool COUNT>
__device__ f3 trace_path(
    const DeviceScene &s, const RenderParams &p, uint32_t px, uint32_t py, uint32_t frameIndex,
    const TraversalStack &stack,
    LaneCounters &cnt)
{
    Rng rng{px, py, frameIndex};
    const f2 j = rng.rnd2d01();
    const f2 uv = f2{((float)px + j.x) / (float)p.width, ((float)py + j.y) / (float)p.height};

    f3 color = f3{0.0f, 0.0f, 0.0f};
    f3 throughput = f3{1.0f, 1.0f, 1.0f};
    uint32_t bounce = 0;
    Ray ray;
    if (p.pc.flags & PROSPER_PC_FLAG_DEPTH_OF_FIELD)
    {
        const f2 lens = rng.rnd2d01();
        ray = thin_lens_camera_ray(p, uv, lens);
    }
    else
        ray = pinhole_camera_ray(p, uv);
    if constexpr (COUNT) cnt.paths++;

    while (bounce < PROSPER_RT_MAX_BOUNCES)
    {
        if (bounce >= p.pc.maxBounces) break;
        // traceClosest: main.rgen:62-81
        Hit hit;
        if constexpr (COUNT) cnt.closestRays++;
        const bool found = trace<false, COUNT>(s, ray.o, ray.d, ray.tMin, ray.tMax, pcg(rng.x ^ rng.z), stack, hit, cnt);
        if (!found)
        {
            if (p.pc.flags & PROSPER_PC_FLAG_IBL)
            {
                if constexpr (COUNT) cnt.skyLookups++;
                add_bounce(p.pc.flags, color, throughput * sample_skybox(s, ray.d), bounce);
            }
            break;
        }
        const Surface sf = evaluate_surface<COUNT>(s, ray.d, hit, cnt);
        if (p.pc.drawType != PROSPER_DRAW_TYPE_DEFAULT && p.pc.drawType != PROSPER_DRAW_TYPE_MESHLET_ID)
        {
            color = debug_color(s, p.pc.drawType, hit, sf);
            break;
        }
        // evaluateDirectLighting: main.rgen:195-223
        {
            f3 l, irradiance;
            float d;
            f3 direct = f3{0.0f, 0.0f, 0.0f};
            if (prepare_direct_lighting<COUNT>(s, sf, throughput, rng, l, d, irradiance, cnt))
            {
                // shadow(): main.rgen:49-60, traced only when it can change the result
                Hit sh;
                if constexpr (COUNT) cnt.shadowRays++;
                const f3 brdf = eval_brdf_times_nol(l, sf);
                const f3 lit = direct_lighting_value(s, throughput, irradiance, brdf, 1.0f);
                const f3 blocked = direct_lighting_value(s, throughput, irradiance, brdf, 0.0f);
                bool occluded = false;
                if (shadow_ray_matters(lit, blocked))
                    occluded = trace<true, COUNT>(s, sf.positionWS, l, 0.1f, d, pcg(rng.x ^ rng.y), stack, sh, cnt);
                direct = occluded ? blocked : lit;
            }
            add_bounce(p.pc.flags, color, direct, bounce);
        }
        f3 rd;
        importance_sample_bounce(sf, rng, throughput, rd);
        if (!p.traceDeadPaths && throughput_is_zero(throughput)) break;
        if (bounce > p.pc.rouletteStartBounce)
        {
            if (rng.rnd01() < fmax_(0.05f, 1.0f - max3(throughput))) break;
        }
        ray.o = offset_ray(sf.positionWS, sf.normalWS);
        ray.d = rd;
        ray.tMin = 0.0f;
        ray.tMax = kInf;
        bounce++;
    }
    return color;
}

// One lane per pixel; a wave covers an 8x8 pixel tile, a 256-thread workgroup a 16x16 tile.
// Workgroups are dealt round-robin over the 8 XCDs (b and b+8 share an XCD's L2), so the block id
// is remapped to give each XCD a contiguous band of tiles (speed only, never correctness).
template <bool COUNT>
__global__ __launch_bounds__(256) void render_megakernel(
    DeviceScene s, RenderParams p, float4 *__restrict__ hdr, unsigned long long *__restrict__ counters,
    int32_t *__restrict__ stackOverflow)
{
    __shared__ int32_t ldsStack[kTraversalStackDepth * 256];

    const uint32_t tilesX = (p.localWidth + 15u) / 16u;
    const uint32_t tilesY = (p.height + 15u) / 16u;
    const uint32_t numTiles = tilesX * tilesY;
    const uint32_t perXcd = (numTiles + 7u) / 8u;
    const uint32_t tile = (blockIdx.x % 8u) * perXcd + (blockIdx.x / 8u);
    if (tile >= numTiles) return;

    const uint32_t wave = threadIdx.x >> 6;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t lx = (tile % tilesX) * 16u + (wave & 1u) * 8u + (lane & 7u);
    const uint32_t py = (tile / tilesX) * 16u + (wave >> 1) * 8u + (lane >> 3);
    const TraversalStack stack{(lds_int32 *)ldsStack + wave * (kTraversalStackDepth * 64u) + lane,
                               stackOverflow + blockIdx.x * 256u + threadIdx.x, kTraversalStackDepth, gridDim.x * 256u, 64u};

    LaneCounters cnt = {};
    if (lx < p.localWidth && py < p.height)
    {
        const uint32_t px = local_to_global_x(p, lx);
        float4 *texel = hdr + (size_t)py * p.localWidth + lx;
        float4 history = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        for (uint32_t f = 0; f < p.frameCount; ++f)
        {
            const uint32_t frameIndex = (p.pc.frameIndex + f) % PROSPER_RT_FRAME_PERIOD;
            const f3 color = trace_path<COUNT>(s, p, px, py, frameIndex, stack, cnt);
            // main.rgen:285-298; skipHistory only applies to the first frame of a batch.  Frames
            // after the first find their history in registers (an fp32 store + load is lossless).
            const bool skip = (f == 0 && (p.pc.flags & PROSPER_PC_FLAG_SKIP_HISTORY)) ||
                              !(p.pc.flags & PROSPER_PC_FLAG_ACCUMULATE);
            if (skip)
                history = make_float4(color.x, color.y, color.z, 1.0f);
            else
            {
                if (f == 0) history = *texel;
                if constexpr (COUNT) cnt.historyReads++;
                const float hc = history.w + 1.0f;
                const float invHc = 1.0f / hc;
                history = make_float4(
                    __builtin_fmaf(color.x - history.x, invHc, history.x), __builtin_fmaf(color.y - history.y, invHc, history.y),
                    __builtin_fmaf(color.z - history.z, invHc, history.z), hc);
            }
            if constexpr (COUNT) cnt.pixelsWritten++;
        }
        *texel = history;
    }
    flush_counters<COUNT>(cnt, counters);
}

uint32_t megakernel_grid_blocks(const RenderParams &p)
{
    const uint32_t numTiles = ((p.localWidth + 15u) / 16u) * ((p.height + 15u) / 16u);
    return ((numTiles + 7u) / 8u) * 8u;
}

void launch_render_megakernel(
    const DeviceScene &s, const RenderParams &p, float4 *hdr, unsigned long long *counters, int32_t *stackOverflow,
    bool countWork, hipStream_t stream)
{
    const uint32_t tilesX = (p.localWidth + 15u) / 16u;
    const uint32_t tilesY = (p.height + 15u) / 16u;
    const uint32_t numTiles = tilesX * tilesY;
    if (numTiles == 0) return;
    const uint32_t perXcd = (numTiles + 7u) / 8u;
    const dim3 grid(perXcd * 8u), block(256);
    if (countWork)
        hipLaunchKernelGGL(render_megakernel<true>, grid, block, 0, stream, s, p, hdr, counters, stackOverflow);
    else
        hipLaunchKernelGGL(render_megakernel<false>, grid, block, 0, stream, s, p, hdr, counters, stackOverflow);
}

#ifdef PPT_EXPERIMENTS // (A/B pipeline, measured no faster than the megakernel: profiles/r01_*)
// ------------------------------------------------------------------------------------------
// Persistent waves with path regeneration.
//
// One lane per pixel wastes half the machine on this path: paths end after 1..maxBounces bounces
// (miss, roulette), so within a wave the set of live lanes shrinks every bounce.  Here a fixed set
// of resident waves pulls pixels from one global counter; every loop iteration runs ONE bounce for
// all live lanes (all lanes traverse together, then shade together), and a lane whose path ended
// starts its pixel's next accumulated frame - or claims the next pixel with a wave-aggregated
// atomic (ballot + popcount, one atomic per wave) - before the next iteration.  A pixel's frames
// stay on one lane in order, so the running mean of main.rgen:289-297 is evaluated exactly as by
// frame-at-a-time rendering (same bits), with the history in registers.
// Work order: 8x8-pixel tiles in row-major tile order, so a wave's first claim is one coherent tile.
// ------------------------------------------------------------------------------------------

// One iteration of the while loop of main.rgen:241-283 for one lane; returns true when the path
// has ended (its radiance is then in st.color).
template <bool COUNT>
__device__ __forceinline__ bool path_bounce(
    const DeviceScene &s, const RenderParams &p, PathState &st, const TraversalStack &stack, LaneCounters &cnt)
{
    Hit hit;
    if constexpr (COUNT) cnt.closestRays++;
    const bool found = trace<false, COUNT>(s, st.o, st.d, 0.0f, kInf, pcg(st.rng.x ^ st.rng.z), stack, hit, cnt);
    if (!found)
    {
        if (p.pc.flags & PROSPER_PC_FLAG_IBL)
        {
            if constexpr (COUNT) cnt.skyLookups++;
            add_bounce(p.pc.flags, st.color, st.throughput * sample_skybox(s, st.d), st.bounce);
        }
        return true;
    }
    const Surface sf = evaluate_surface<COUNT>(s, st.d, hit, cnt);
    if (p.pc.drawType != PROSPER_DRAW_TYPE_DEFAULT && p.pc.drawType != PROSPER_DRAW_TYPE_MESHLET_ID)
    {
        st.color = debug_color(s, p.pc.drawType, hit, sf);
        return true;
    }
    {
        f3 l, irradiance;
        float d;
        f3 direct = f3{0.0f, 0.0f, 0.0f};
        if (prepare_direct_lighting<COUNT>(s, sf, st.throughput, st.rng, l, d, irradiance, cnt))
        {
            Hit sh;
            if constexpr (COUNT) cnt.shadowRays++;
            const f3 brdf = eval_brdf_times_nol(l, sf);
            const f3 lit = direct_lighting_value(s, st.throughput, irradiance, brdf, 1.0f);
            const f3 blocked = direct_lighting_value(s, st.throughput, irradiance, brdf, 0.0f);
            bool occluded = false;
            if (shadow_ray_matters(lit, blocked))
                occluded = trace<true, COUNT>(s, sf.positionWS, l, 0.1f, d, pcg(st.rng.x ^ st.rng.y), stack, sh, cnt);
            direct = occluded ? blocked : lit;
        }
        add_bounce(p.pc.flags, st.color, direct, st.bounce);
    }
    f3 rd;
    importance_sample_bounce(sf, st.rng, st.throughput, rd);
    if (!p.traceDeadPaths && throughput_is_zero(st.throughput)) return true;
    if (st.bounce > p.pc.rouletteStartBounce)
    {
        if (st.rng.rnd01() < fmax_(0.05f, 1.0f - max3(st.throughput))) return true;
    }
    st.o = offset_ray(sf.positionWS, sf.normalWS);
    st.d = rd;
    st.bounce++;
    // the loop conditions of main.rgen:241-244, evaluated now instead of next iteration
    return st.bounce >= PROSPER_RT_MAX_BOUNCES || st.bounce >= p.pc.maxBounces;
}

template <bool COUNT>
__global__ __launch_bounds__(256) void render_persistent(
    DeviceScene s, RenderParams p, float4 *__restrict__ hdr, unsigned long long *__restrict__ counters,
    uint32_t *__restrict__ workCounter, int32_t *__restrict__ stackOverflow)
{
    __shared__ int32_t ldsStack[kTraversalStackDepth * 256];
    const uint32_t wave = threadIdx.x >> 6;
    const uint32_t lane = threadIdx.x & 63u;
    const TraversalStack stack{(lds_int32 *)ldsStack + wave * (kTraversalStackDepth * 64u) + lane,
                               stackOverflow + blockIdx.x * 256u + threadIdx.x, kTraversalStackDepth, gridDim.x * 256u, 64u};

    const uint32_t tilesX = (p.localWidth + 7u) / 8u;
    const uint32_t tilesY = (p.height + 7u) / 8u;
    const uint32_t totalWork = tilesX * tilesY * 64u;
    const bool zeroBounces = p.pc.maxBounces == 0;

    LaneCounters cnt = {};
    PathState st = {};
    float4 history = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    uint32_t lx = 0, py = 0, px = 0;
    uint32_t frame = 0;    // accumulated frames finished for the current pixel
    bool havePixel = false; // lane owns a pixel
    bool active = false;    // lane has a path in flight
    bool done = false;      // work counter exhausted

    while (true)
    {
        if (!active && !done)
        {
            if (havePixel && frame >= p.frameCount)
            {
                hdr[(size_t)py * p.localWidth + lx] = history;
                havePixel = false;
            }
            if (!havePixel)
            {
                // wave-aggregated claim of the next work items
                const unsigned long long need = __ballot(1);
                const uint32_t leader = (uint32_t)__builtin_ctzll(need);
                uint32_t base = 0;
                if (lane == leader) base = atomicAdd(workCounter, (uint32_t)__builtin_popcountll(need));
                base = __shfl(base, leader, 64);
                const uint32_t idx = base + (uint32_t)__builtin_popcountll(need & ((1ull << lane) - 1ull));
                if (idx >= totalWork)
                    done = true;
                else
                {
                    const uint32_t tile = idx >> 6, inTile = idx & 63u;
                    lx = (tile % tilesX) * 8u + (inTile & 7u);
                    py = (tile / tilesX) * 8u + (inTile >> 3);
                    if (lx < p.localWidth && py < p.height)
                    {
                        px = local_to_global_x(p, lx);
                        havePixel = true;
                        frame = 0;
                    }
                }
            }
            if (havePixel)
            {
                start_path<COUNT>(p, px, py, (p.pc.frameIndex + frame) % PROSPER_RT_FRAME_PERIOD, st, cnt);
                active = true;
            }
        }
        if (__ballot(active) == 0ull)
        {
            if (__ballot(!done) == 0ull) break;
            continue;
        }
        if (active)
        {
            const bool ended = zeroBounces ? true : path_bounce<COUNT>(s, p, st, stack, cnt);
            if (ended)
            {
                // main.rgen:285-298; skipHistory applies to the first frame of a batch only
                const bool skip = (frame == 0 && (p.pc.flags & PROSPER_PC_FLAG_SKIP_HISTORY)) ||
                                  !(p.pc.flags & PROSPER_PC_FLAG_ACCUMULATE);
                if (skip)
                    history = make_float4(st.color.x, st.color.y, st.color.z, 1.0f);
                else
                {
                    if (frame == 0) history = hdr[(size_t)py * p.localWidth + lx];
                    if constexpr (COUNT) cnt.historyReads++;
                    const float hc = history.w + 1.0f;
                const float invHc = 1.0f / hc;
                history = make_float4(
                    __builtin_fmaf(st.color.x - history.x, invHc, history.x), __builtin_fmaf(st.color.y - history.y, invHc, history.y),
                    __builtin_fmaf(st.color.z - history.z, invHc, history.z), hc);
                }
                if constexpr (COUNT) cnt.pixelsWritten++;
                frame++;
                active = false;
            }
        }
    }
    flush_counters<COUNT>(cnt, counters);
}

uint32_t persistent_grid_blocks()
{
    static uint32_t cached = 0;
    if (cached) return cached;
    int perCu = 0, dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess ||
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&perCu, render_persistent<false>, 256, 0) != hipSuccess || perCu < 1)
    {
        perCu = 2;
        prop.multiProcessorCount = 256;
    }
    cached = (uint32_t)perCu * (uint32_t)prop.multiProcessorCount;
    return cached;
}

void launch_render_persistent(
    const DeviceScene &s, const RenderParams &p, float4 *hdr, unsigned long long *counters, uint32_t *workCounter,
    int32_t *stackOverflow, bool countWork, hipStream_t stream)
{
    const uint32_t tilesX = (p.localWidth + 7u) / 8u;
    const uint32_t tilesY = (p.height + 7u) / 8u;
    const uint32_t waves = tilesX * tilesY;
    if (waves == 0) return;
    (void)hipMemsetAsync(workCounter, 0, sizeof(uint32_t), stream);
    uint32_t blocks = persistent_grid_blocks();
    const uint32_t needed = (waves + 3u) / 4u;
    if (blocks > needed) blocks = needed;
    const dim3 grid(blocks), block(256);
    if (countWork)
        hipLaunchKernelGGL(render_persistent<true>, grid, block, 0, stream, s, p, hdr, counters, workCounter, stackOverflow);
    else
        hipLaunchKernelGGL(render_persistent<false>, grid, block, 0, stream, s, p, hdr, counters, workCounter, stackOverflow);
}

#endif // PPT_EXPERIMENTS

// ------------------------------------------------------------------------------------------
// RGBA32F -> RGBA16F
// ------------------------------------------------------------------------------------------

// De-interleave of gathered rank tiles ("then a de-interleave copy kernel on rank 0"): rank r's tile
// holds the image stripes s with s % ranks == r, rows of localWidth[r] texels; the full image row takes stripe s
// from rank s % ranks at local stripe s / ranks.  One thread per texel (16 B): a stripe of 16 texels is a
// 256-byte run on both sides, so loads and stores stay coalesced.  HBM-bound copy: 32 B per texel.
__global__ __launch_bounds__(256) void deinterleave_tiles_kernel(
    const float4 *__restrict__ tiles, TileLayout layout, float4 *__restrict__ full)
{
    const uint32_t x = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t y = blockIdx.y;
    if (x >= layout.width) return;
    const uint32_t stripe = x / layout.stripeWidth;
    const uint32_t rank = stripe % layout.ranks;
    const uint32_t lx = (stripe / layout.ranks) * layout.stripeWidth + (x - stripe * layout.stripeWidth);
    full[(size_t)y * layout.width + x] = tiles[layout.tileOffset[rank] + (size_t)y * layout.localWidth[rank] + lx];
}

void launch_deinterleave_tiles(const float4 *tiles, const TileLayout &layout, float4 *full, hipStream_t stream)
{
    if (layout.width == 0 || layout.height == 0) return;
    hipLaunchKernelGGL(
        deinterleave_tiles_kernel, dim3((layout.width + 255u) / 256u, layout.height), dim3(256), 0, stream, tiles, layout, full);
}

__global__ void blit_rgba16f_kernel(const float4 *__restrict__ in, uint2 *__restrict__ out, uint32_t count)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const float4 v = in[i];
    out[i] = make_uint2(float_to_half(v.x) | (float_to_half(v.y) << 16), float_to_half(v.z) | (float_to_half(v.w) << 16));
}

void launch_blit_rgba16f(const float4 *in, void *out, uint32_t count, hipStream_t stream)
{
    if (count == 0) return;
    hipLaunchKernelGGL(
        blit_rgba16f_kernel, dim3((count + 255) / 256), dim3(256), 0, stream, in, static_cast<uint2 *>(out), count);
}

// ------------------------------------------------------------------------------------------
// ReSTIR-DI trace (res/shader/rt/direct_illumination/main.rgen:44-165, src/render/rtdi/Trace.cpp:297): a second
// client of the traversal.  One lane per pixel, a wave per 8x8 tile: surface from the G-buffer, the pixel's
// reservoir light, one shadow ray (any-hit included), BRDF, running mean into the HDR image.
// ------------------------------------------------------------------------------------------

// scene/material.glsl:20-32
PPT_D f3 signed_oct_decode(f3 n)
{
    f3 o;
    o.x = n.x - n.y;
    o.y = (n.x + n.y) - 1.0f;
    o.z = n.z * 2.0f - 1.0f;
    o.z = o.z * ((1.0f - fabs_(o.x)) - fabs_(o.y));
    return normalize(o);
}

struct RestirParams
{
    uint32_t drawType, frameIndex, flags, width, height;
    float eye[3];
    float clipToWorld[16]; // column-major
};

__global__ __launch_bounds__(256) void restir_di_trace_kernel(
    DeviceScene s, RestirParams p, const float4 *__restrict__ albedoRoughness, const float4 *__restrict__ normalMetallic,
    const float *__restrict__ nonLinearDepth, const float2 *__restrict__ reservoirs, float4 *__restrict__ hdr,
    int32_t *__restrict__ stackOverflow)
{
    __shared__ int32_t ldsStack[kTraversalStackDepth * 256];
    const uint32_t tilesX = (p.width + 15u) / 16u, tilesY = (p.height + 15u) / 16u;
    const uint32_t numTiles = tilesX * tilesY;
    const uint32_t perXcd = (numTiles + 7u) / 8u;
    const uint32_t tile = (blockIdx.x % 8u) * perXcd + (blockIdx.x / 8u);
    if (tile >= numTiles) return;
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    const uint32_t px = (tile % tilesX) * 16u + (wave & 1u) * 8u + (lane & 7u);
    const uint32_t py = (tile / tilesX) * 16u + (wave >> 1) * 8u + (lane >> 3);
    const TraversalStack stack{(lds_int32 *)ldsStack + wave * (kTraversalStackDepth * 64u) + lane,
                               stackOverflow + blockIdx.x * 256u + threadIdx.x, kTraversalStackDepth, gridDim.x * 256u, 64u};
    if (px >= p.width || py >= p.height) return;
    const size_t i = (size_t)py * p.width + px;

    // main.rgen:113-129
    const f2 uv = f2{(float)px / (float)p.width, (float)py / (float)p.height};
    const float depth = nonLinearDepth[i];
    Surface sf;
    {
        // worldPos, scene/camera.glsl:27-33
        const float *m = p.clipToWorld;
        const float x = uv.x * 2.0f - 1.0f, y = uv.y * 2.0f - 1.0f;
        const float vx = __builtin_fmaf(m[8], depth, __builtin_fmaf(m[4], y, __builtin_fmaf(m[0], x, m[12])));
        const float vy = __builtin_fmaf(m[9], depth, __builtin_fmaf(m[5], y, __builtin_fmaf(m[1], x, m[13])));
        const float vz = __builtin_fmaf(m[10], depth, __builtin_fmaf(m[6], y, __builtin_fmaf(m[2], x, m[14])));
        const float vw = __builtin_fmaf(m[11], depth, __builtin_fmaf(m[7], y, __builtin_fmaf(m[3], x, m[15])));
        sf.positionWS = f3{vx, vy, vz} * (1.0f / vw);
    }
    sf.invViewRayWS = normalize(f3{p.eye[0], p.eye[1], p.eye[2]} - sf.positionWS);
    const float4 ar = albedoRoughness[i], nm = normalMetallic[i];
    sf.material.albedo = f3{ar.x, ar.y, ar.z};
    sf.material.roughness = ar.w;
    sf.material.normal = signed_oct_decode(f3{nm.x, nm.y, nm.w});
    sf.material.metallic = nm.z;
    sf.material.alpha = -1.0f;
    sf.normalWS = sf.material.normal;
    sf.uv = f2{0.0f, 0.0f};
    sf.NoV = saturate(dot(sf.normalWS, sf.invViewRayWS));

    if (p.drawType != PROSPER_DRAW_TYPE_DEFAULT)
    {
        const f3 c = p.drawType == PROSPER_DRAW_TYPE_POSITION ? sf.positionWS : sf.material.albedo;
        hdr[i] = make_float4(c.x, c.y, c.z, 1.0f);
        return;
    }
    // evaluateDirectLightingReSTIR, main.rgen:88-109
    const float2 packed = reservoirs[i];
    const int32_t lightIndex = (int32_t)f2u(packed.x);
    f3 color = f3{0.0f, 0.0f, 0.0f};
    if (!(sf.material.alpha == 0.0f || lightIndex < 0))
    {
        f3 l, irradiance;
        float d;
        sample_light(s, sf.positionWS, (uint32_t)lightIndex, l, d, irradiance);
        if (dot(l, sf.normalWS) > 0.0f)
        {
            LaneCounters cnt = {};
            Hit sh;
            const bool occluded = trace<true, false>(s, sf.positionWS, l, 0.1f, d, pcg(px ^ py), stack, sh, cnt);
            irradiance = irradiance * (occluded ? 0.0f : 1.0f);
            color = (irradiance * eval_brdf_times_nol(l, sf)) * packed.y;
        }
    }
    if ((p.flags & 1u) || !(p.flags & 2u))
        hdr[i] = make_float4(color.x, color.y, color.z, 1.0f);
    else
    {
        const float4 h = hdr[i];
        const float count = h.w + 1.0f;
        const float inv = 1.0f / count;
        hdr[i] = make_float4(
            __builtin_fmaf(color.x - h.x, inv, h.x), __builtin_fmaf(color.y - h.y, inv, h.y),
            __builtin_fmaf(color.z - h.z, inv, h.z), count);
    }
}

uint32_t restir_grid_blocks(uint32_t width, uint32_t height)
{
    const uint32_t numTiles = ((width + 15u) / 16u) * ((height + 15u) / 16u);
    return ((numTiles + 7u) / 8u) * 8u;
}

void launch_restir_di_trace(
    const DeviceScene &s, uint32_t drawType, uint32_t frameIndex, uint32_t flags, uint32_t width, uint32_t height,
    const float eye[3], const float clipToWorld[16], const void *albedoRoughness, const void *normalMetallic,
    const float *nonLinearDepth, const void *reservoirs, float4 *hdr, int32_t *stackOverflow, hipStream_t stream)
{
    if (width == 0 || height == 0) return;
    RestirParams p;
    p.drawType = drawType;
    p.frameIndex = frameIndex;
    p.flags = flags;
    p.width = width;
    p.height = height;
    for (int k = 0; k < 3; ++k) p.eye[k] = eye[k];
    for (int k = 0; k < 16; ++k) p.clipToWorld[k] = clipToWorld[k];
    hipLaunchKernelGGL(
        restir_di_trace_kernel, dim3(restir_grid_blocks(width, height)), dim3(256), 0, stream, s, p,
        static_cast<const float4 *>(albedoRoughness), static_cast<const float4 *>(normalMetallic), nonLinearDepth,
        static_cast<const float2 *>(reservoirs), hdr, stackOverflow);
}

// ------------------------------------------------------------------------------------------
// tone map: the step after the path (res/shader/tone_map.comp:17-60, src/render/ToneMap.cpp:62-128),
// fused with the RGBA32F -> RGBA16F blit it reads through (RtReference.cpp:339-377).  One thread per
// pixel: 16 B in, 4 B out, eight taps of the 442 KB LUT (L2-resident).  Arithmetic: DESIGN.md
// "tone map" (GLSL mod, pow through pow_, folded constants, exact R9G9B9E5 decode, fma-chain trilinear).
// ------------------------------------------------------------------------------------------

PPT_D float mod_(float x, float y) { return x - y * __builtin_floorf(x / y); }

// common/math.glsl:17-44
PPT_D f3 rgb_to_hsv(f3 rgb)
{
    const float value = fmax_(fmax_(rgb.x, rgb.y), rgb.z);
    const float valueMinusChroma = fmin_(fmin_(rgb.x, rgb.y), rgb.z);
    const float chroma = value - valueMinusChroma;
    float hue;
    if (chroma == 0.0f)
        hue = 0.0f;
    else if (value == rgb.x)
        hue = mod_((rgb.y - rgb.z) / chroma, 6.0f);
    else if (value == rgb.y)
        hue = (rgb.z - rgb.x) / chroma + 2.0f;
    else
        hue = (rgb.x - rgb.y) / chroma + 4.0f;
    const float saturation = value == 0.0f ? 0.0f : chroma / value;
    return f3{hue, saturation, value};
}

// common/math.glsl:47-83
PPT_D f3 hsv_to_rgb(f3 hsv)
{
    const float hue = hsv.x, saturation = hsv.y, value = hsv.z;
    const float chroma = value * saturation;
    const float x = chroma * (1.0f - fabs_(mod_(hue, 2.0f) - 1.0f));
    f3 rgb;
    if (hue < 1.0f)
        rgb = f3{chroma, x, 0.0f};
    else if (hue < 2.0f)
        rgb = f3{x, chroma, 0.0f};
    else if (hue < 3.0f)
        rgb = f3{0.0f, chroma, x};
    else if (hue < 4.0f)
        rgb = f3{0.0f, x, chroma};
    else if (hue < 5.0f)
        rgb = f3{x, 0.0f, chroma};
    else
        rgb = f3{chroma, 0.0f, x};
    const float m = value - chroma;
    return f3{rgb.x + m, rgb.y + m, rgb.z + m};
}

PPT_D f3 decode_r9g9b9e5(uint32_t p)
{
    const float scale = u2f(((p >> 27) + 103u) << 23); // 2^(e - 15 - 9)
    return f3{(float)(p & 0x1FFu) * scale, (float)((p >> 9) & 0x1FFu) * scale, (float)((p >> 18) & 0x1FFu) * scale};
}

PPT_D int32_t clamp_texel(int32_t i, int32_t n) { return i < 0 ? 0 : (i >= n ? n - 1 : i); }

// textureLod(sampler3D(lut, linear / clamp-to-edge), uv, 0).xyz
PPT_D f3 sample_lut(const uint32_t *__restrict__ lut, int32_t n, f3 uv)
{
    const float u = __builtin_fmaf(uv.x, (float)n, -0.5f), v = __builtin_fmaf(uv.y, (float)n, -0.5f),
                w = __builtin_fmaf(uv.z, (float)n, -0.5f);
    const float fu = __builtin_floorf(u), fv = __builtin_floorf(v), fw = __builtin_floorf(w);
    const float a = u - fu, b = v - fv, c = w - fw;
    const int32_t is[2] = {clamp_texel(f2i(fu), n), clamp_texel(f2i(fu) + 1, n)};
    const int32_t js[2] = {clamp_texel(f2i(fv), n), clamp_texel(f2i(fv) + 1, n)};
    const int32_t ks[2] = {clamp_texel(f2i(fw), n), clamp_texel(f2i(fw) + 1, n)};
    const float wx[2] = {1.0f - a, a}, wy[2] = {1.0f - b, b}, wz[2] = {1.0f - c, c};
    f3 acc = {0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int z = 0; z < 2; ++z)
#pragma unroll
        for (int y = 0; y < 2; ++y)
#pragma unroll
            for (int x = 0; x < 2; ++x)
            {
                const f3 t = decode_r9g9b9e5(lut[((size_t)ks[z] * n + (size_t)js[y]) * n + (size_t)is[x]]);
                const float wgt = (wx[x] * wy[y]) * wz[z];
                if (x == 0 && y == 0 && z == 0)
                    acc = f3{wgt * t.x, wgt * t.y, wgt * t.z};
                else
                    acc = f3{__builtin_fmaf(wgt, t.x, acc.x), __builtin_fmaf(wgt, t.y, acc.y), __builtin_fmaf(wgt, t.z, acc.z)};
            }
    return acc;
}

PPT_D uint32_t to_unorm8(float x)
{
    if (x != x) return 0u;
    return (uint32_t)__builtin_rintf(clamp_(x, 0.0f, 1.0f) * 255.0f);
}

__global__ __launch_bounds__(256) void tone_map_kernel(
    const float4 *__restrict__ hdr, const uint32_t *__restrict__ lut, uint32_t dim, float exposure, float contrast,
    uint32_t *__restrict__ out, uint32_t count)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const float4 h = hdr[i];
    // the RGBA16F image the tone map reads: the blit rounds to nearest even
    // (v_cvt_f16_f32 / v_cvt_f32_f16: the same values as float_to_half / half_to_float; a NaN's payload may
    // differ but cannot reach the UNORM8 output)
    f3 color = f3{(float)(_Float16)h.x, (float)(_Float16)h.y, (float)(_Float16)h.z};
    color = color * exposure;
    f3 hsv = rgb_to_hsv(color);
    hsv.z = pow_(hsv.z, contrast);
    color = hsv_to_rgb(hsv);
    // tonyMcMapface, tone_map.comp:17-29
    const f3 enc = f3{color.x / (color.x + 1.0f), color.y / (color.y + 1.0f), color.z / (color.z + 1.0f)};
    const float s1 = dim == 48u ? 47.0f / 48.0f : ((float)dim - 1.0f) / (float)dim;
    const float s0 = dim == 48u ? 0.5f / 48.0f : 0.5f / (float)dim;
    const f3 uv = f3{enc.x * s1 + s0, enc.y * s1 + s0, enc.z * s1 + s0};
    color = sample_lut(lut, (int32_t)dim, uv);
    const float invGamma = 1.0f / 2.2f;
    out[i] = to_unorm8(pow_(color.x, invGamma)) | (to_unorm8(pow_(color.y, invGamma)) << 8) |
             (to_unorm8(pow_(color.z, invGamma)) << 16) | 0xFF000000u;
}

void launch_tone_map(
    const float4 *hdr, const uint32_t *lut, uint32_t dim, float exposure, float contrast, void *outRgba8, uint32_t count,
    hipStream_t stream)
{
    if (count == 0) return;
    hipLaunchKernelGGL(
        tone_map_kernel, dim3((count + 255) / 256), dim3(256), 0, stream, hdr, lut, dim, exposure, contrast,
        static_cast<uint32_t *>(outRgba8), count);
}

// ------------------------------------------------------------------------------------------
// device self-test
// ------------------------------------------------------------------------------------------

__global__ void eval_fn_kernel(
    uint32_t fn, const float *__restrict__ in, uint32_t inStride, float *__restrict__ out, uint32_t outStride, uint32_t n)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float *a = in + (size_t)i * inStride;
    float *o = out + (size_t)i * outStride;
    switch (fn)
    {
    case PROSPER_PT_FN_SINCOS: sincos_(a[0], o[0], o[1]); break;
    case PROSPER_PT_FN_POW: o[0] = pow_(a[0], a[1]); break;
    case PROSPER_PT_FN_SRGB_TO_LINEAR: o[0] = srgb_to_linear(a[0]); break;
    case PROSPER_PT_FN_NORMALIZE:
    {
        const f3 r = normalize(f3{a[0], a[1], a[2]});
        o[0] = r.x; o[1] = r.y; o[2] = r.z;
        break;
    }
    case PROSPER_PT_FN_UNPACK_SNORM:
    {
        const uint32_t bits = f2u(a[0]);
        const f3 r = unpack_snorm_r10g10b10(bits);
        o[0] = r.x; o[1] = r.y; o[2] = r.z;
        o[3] = (float)((int32_t)bits >> 30);
        break;
    }
    case PROSPER_PT_FN_ONB:
    {
        const Onb b = orthonormal_basis(f3{a[0], a[1], a[2]});
        o[0] = b.b1.x; o[1] = b.b1.y; o[2] = b.b1.z;
        o[3] = b.b2.x; o[4] = b.b2.y; o[5] = b.b2.z;
        o[6] = b.n.x; o[7] = b.n.y; o[8] = b.n.z;
        break;
    }
    case PROSPER_PT_FN_COSINE_SAMPLE:
    {
        const f3 r = cosine_sample_hemisphere(f3{a[0], a[1], a[2]}, f2{a[3], a[4]});
        o[0] = r.x; o[1] = r.y; o[2] = r.z;
        break;
    }
    case PROSPER_PT_FN_VNDF_SAMPLE:
    {
        const f3 r = sample_visible_trowbridge_reitz(f3{a[0], a[1], a[2]}, a[3], f2{a[4], a[5]});
        o[0] = r.x; o[1] = r.y; o[2] = r.z;
        break;
    }
    case PROSPER_PT_FN_VNDF_PDF:
        o[0] = visible_trowbridge_reitz_pdf(f3{a[0], a[1], a[2]}, f3{a[3], a[4], a[5]}, a[6]);
        break;
    case PROSPER_PT_FN_EVAL_BRDF:
    {
        Surface sf = {};
        sf.normalWS = f3{a[3], a[4], a[5]};
        sf.invViewRayWS = f3{a[6], a[7], a[8]};
        sf.material.albedo = f3{a[9], a[10], a[11]};
        sf.material.roughness = a[12];
        sf.material.metallic = a[13];
        sf.NoV = saturate(dot(sf.normalWS, sf.invViewRayWS));
        const f3 r = eval_brdf_times_nol(f3{a[0], a[1], a[2]}, sf);
        o[0] = r.x; o[1] = r.y; o[2] = r.z;
        break;
    }
    case PROSPER_PT_FN_OFFSET_RAY:
    {
        const f3 r = offset_ray(f3{a[0], a[1], a[2]}, f3{a[3], a[4], a[5]});
        o[0] = r.x; o[1] = r.y; o[2] = r.z;
        break;
    }
    case PROSPER_PT_FN_POINT_LIGHT:
    {
        prosper_PointLight L;
        L.position = prosper_vec4{a[0], a[1], a[2], 0.0f};
        L.radianceAndRadius = prosper_vec4{a[3], a[4], a[5], a[6]};
        f3 l, irr;
        float d;
        eval_point_light(L, f3{a[7], a[8], a[9]}, l, d, irr);
        o[0] = l.x; o[1] = l.y; o[2] = l.z; o[3] = d; o[4] = irr.x; o[5] = irr.y; o[6] = irr.z;
        break;
    }
    case PROSPER_PT_FN_SPOT_LIGHT:
    {
        prosper_SpotLight L;
        L.positionAndAngleOffset = prosper_vec4{a[0], a[1], a[2], a[3]};
        L.radianceAndAngleScale = prosper_vec4{a[4], a[5], a[6], a[7]};
        L.direction = prosper_vec4{a[8], a[9], a[10], 0.0f};
        f3 l, irr;
        float d;
        eval_spot_light(L, f3{a[11], a[12], a[13]}, l, d, irr);
        o[0] = l.x; o[1] = l.y; o[2] = l.z; o[3] = d; o[4] = irr.x; o[5] = irr.y; o[6] = irr.z;
        break;
    }
    case PROSPER_PT_FN_TRIANGLE:
    {
        float t = 0.0f, bu = 0.0f, bv = 0.0f;
        const f3 dir = f3{a[3], a[4], a[5]};
        const bool hit = intersect_triangle(
            f3{a[0], a[1], a[2]}, dir, f3{safe_rcp_dir(dir.x), safe_rcp_dir(dir.y), safe_rcp_dir(dir.z)},
            f3{a[6], a[7], a[8]}, f3{a[9], a[10], a[11]}, f3{a[12], a[13], a[14]}, a[15], a[16], t, bu, bv);
        o[0] = hit ? 1.0f : 0.0f; o[1] = hit ? t : 0.0f; o[2] = hit ? bu : 0.0f; o[3] = hit ? bv : 0.0f;
        break;
    }
    case PROSPER_PT_FN_HALF:
    {
        const uint32_t h = float_to_half(a[0]);
        o[0] = half_to_float(h);
        o[1] = u2f(h);
        break;
    }
    case PROSPER_PT_FN_RNG:
    {
        Rng r{f2u(a[0]), f2u(a[1]), f2u(a[2])};
        o[0] = r.rnd01();
        const f2 u = r.rnd2d01();
        o[1] = u.x; o[2] = u.y;
        o[3] = u2f(pcg(r.x ^ r.z));
        break;
    }
    case PROSPER_PT_FN_BC7_BLOCK:
    {
        const uint32_t block[4] = {f2u(a[0]), f2u(a[1]), f2u(a[2]), f2u(a[3])};
        uint32_t texel[16];
        bc7_decode_block(block, texel);
        for (int k = 0; k < 16; ++k) o[k] = u2f(texel[k]);
        break;
    }
    default: break;
    }
}

// One thread per 4x4 block: decode, then store the four rows into the 8x4-texel tiles of DeviceTexture
// (a block is the left or right half of one tile: four 16-byte stores).
__global__ __launch_bounds__(256) void decode_bc7_kernel(
    const uint4 *__restrict__ blocks, uint32_t blocksX, uint32_t blocksY, uint32_t tilesPerRow, uint32_t *__restrict__ tiled)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= blocksX * blocksY) return;
    const uint4 raw = blocks[i];
    const uint32_t block[4] = {raw.x, raw.y, raw.z, raw.w};
    uint32_t texel[16];
    bc7_decode_block(block, texel);
    const uint32_t bx = i % blocksX, by = i / blocksX;
    uint32_t *tile = tiled + ((size_t)by * tilesPerRow + (bx >> 1)) * (kTexTileW * kTexTileH) + (bx & 1u) * 4u;
    for (uint32_t row = 0; row < 4u; ++row)
        *reinterpret_cast<uint4 *>(tile + row * kTexTileW) =
            make_uint4(texel[row * 4u], texel[row * 4u + 1u], texel[row * 4u + 2u], texel[row * 4u + 3u]);
}

void launch_decode_bc7(
    const void *blocks, uint32_t width, uint32_t height, uint32_t tilesPerRow, void *tiled, hipStream_t stream)
{
    const uint32_t blocksX = width / 4u, blocksY = height / 4u;
    const uint32_t n = blocksX * blocksY;
    if (n == 0) return;
    hipLaunchKernelGGL(
        decode_bc7_kernel, dim3((n + 255u) / 256u), dim3(256), 0, stream, static_cast<const uint4 *>(blocks), blocksX, blocksY,
        tilesPerRow, static_cast<uint32_t *>(tiled));
}

// Row-major RGBA8 texels -> the 8 x 4-texel tiles of DeviceTexture (one thread per texel of the padded extent; texels outside
// the image are zero and never addressed): what prosper_pt_upload_scene did on the host, a texel at a time, before round 4.
__global__ __launch_bounds__(256) void retile_rgba8_kernel(
    const uint32_t *__restrict__ linear, uint32_t width, uint32_t height, uint32_t tilesPerRow, uint32_t tilesY, uint32_t *__restrict__ tiled)
{
    const uint32_t x = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t y = blockIdx.y;
    if (x >= tilesPerRow * kTexTileW || y >= tilesY * kTexTileH) return;
    const uint32_t v = (x < width && y < height) ? linear[(size_t)y * width + x] : 0u;
    tiled[((size_t)(y >> 2) * tilesPerRow + (x >> 3)) * (kTexTileW * kTexTileH) + ((y & 3u) << 3) + (x & 7u)] = v;
}

void launch_retile_rgba8(const void *linear, uint32_t width, uint32_t height, uint32_t tilesPerRow, void *tiled, hipStream_t stream)
{
    const uint32_t tilesY = (height + kTexTileH - 1u) / kTexTileH;
    if (width == 0 || height == 0) return;
    hipLaunchKernelGGL(
        retile_rgba8_kernel, dim3((tilesPerRow * kTexTileW + 255u) / 256u, tilesY * kTexTileH), dim3(256), 0, stream,
        static_cast<const uint32_t *>(linear), width, height, tilesPerRow, tilesY, static_cast<uint32_t *>(tiled));
}

// After prosper_pt_update_materials / _textures touched a MASK / BLEND material: every any-hit record's copy of its
// material's AlphaMaterial again from the (new) table.
__global__ __launch_bounds__(256) void patch_alpha_records_kernel(
    AlphaTriangle *__restrict__ records, uint32_t count, const AlphaMaterial *__restrict__ table, uint32_t materialCount)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const uint32_t m = records[i].materialIndex;
    if (m < materialCount) records[i].material = table[m];
}

void launch_patch_alpha_records(AlphaTriangle *records, uint32_t count, const AlphaMaterial *table, uint32_t materialCount, hipStream_t stream)
{
    if (count == 0) return;
    hipLaunchKernelGGL(patch_alpha_records_kernel, dim3((count + 255u) / 256u), dim3(256), 0, stream, records, count, table, materialCount);
}

// Alpha bounds of a non-opaque material (pt_scene.hpp AlphaMaterial): one thread per cell of 2^shift x 2^shift texels.
// A sample whose footprint STARTS in the cell (i0, j0 of texel_taps / any_hit_record) reads texels (i0, j0), (i1, j0),
// (i0, j1), (i1, j1) with i1 = the wrapped neighbour of i0: one of i0 - 1, i0, i0 + 1 (mirrored repeat can step back),
// so the cell's range grown by one texel on every side - wrapped around for REPEAT, clamped otherwise - holds every
// texel of every such footprint.  With tmin / tmax the extreme alpha bytes in that range, f the filtered value and
// L = sRGBtoLinear:
//     tmin/255 - e1 <= f <= tmax/255 + e1       (convex combination; e1 = kAlphaFilterSlack covers the rounding of the
//                                                weights and of the fma chain, < 6e-7 for values <= 1)
//     L(tmin/255 - e1) - e2 <= L(f) <= L(tmax/255 + e1) + e2   (L monotone up to e2 = kAlphaCurveSlack; the monotonicity
//                                                defect of the device function is measured over every input, test_alpha_bounds)
//     alpha = fl(L(f) * factorA): multiplication by a non-negative constant is monotone under round-to-nearest.
// f >= 0 exactly (non-negative weights and texels), hence alpha >= 0.  lo = the largest byte whose decoded value is <=
// the lower bound, hi = the smallest byte < 255 whose decoded value is >= the upper bound, 255 (= unbounded) if there is
// none.  tmax == 0 makes f, L(f) and alpha exactly 0: hi = 0 without slack.
__global__ __launch_bounds__(256) void build_alpha_bounds_kernel(
    DeviceTexture tex, uint32_t wrapS, uint32_t wrapT, float factorA, uint32_t shift, uint32_t cellsX, uint32_t cellsY,
    uint16_t *__restrict__ out)
{
    const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= cellsX * cellsY) return;
    const int32_t cx = (int32_t)(c % cellsX), cy = (int32_t)(c / cellsX);
    const int32_t size = 1 << shift;
    const int32_t w = (int32_t)tex.width, h = (int32_t)tex.height;
    uint32_t tmin = 255u, tmax = 0u;
    for (int32_t dj = -1; dj <= size; ++dj)
    {
        int32_t j = cy * size + dj;
        j = wrapT == PROSPER_PT_WRAP_REPEAT ? floor_mod(j, h) : (j < 0 ? 0 : (j >= h ? h - 1 : j));
        for (int32_t di = -1; di <= size; ++di)
        {
            int32_t i = cx * size + di;
            i = wrapS == PROSPER_PT_WRAP_REPEAT ? floor_mod(i, w) : (i < 0 ? 0 : (i >= w ? w - 1 : i));
            const uint32_t a = reinterpret_cast<const uint32_t *>(tex.texels)[texel_offset(tex, i, j)] >> 24;
            tmin = a < tmin ? a : tmin;
            tmax = a > tmax ? a : tmax;
        }
    }
    const float k = 1.0f / 255.0f;
    uint32_t lo = 0u, hi = 0u;
    if (tmax != 0u)
    {
        const float lower = fmax_(srgb_to_linear((float)tmin * k - kAlphaFilterSlack) - kAlphaCurveSlack, 0.0f) * factorA;
        const float upper = (srgb_to_linear((float)tmax * k + kAlphaFilterSlack) + kAlphaCurveSlack) * factorA;
        while (lo < 255u && (float)(lo + 1u) * k <= lower) ++lo;
        hi = 255u;
        for (uint32_t b = 0; b < 255u; ++b)
            if ((float)b * k >= upper)
            {
                hi = b;
                break;
            }
        if (hi == 0u) hi = 1u; // hi == 0 is reserved for "alpha is exactly 0" (factorA == 0 lands here: decided by u > hi or exact code)
    }
    out[c] = (uint16_t)(lo | (hi << 8));
}

void launch_build_alpha_bounds(
    const DeviceTexture &tex, uint32_t wrapS, uint32_t wrapT, float factorA, uint32_t shift, uint16_t *out, hipStream_t stream)
{
    const uint32_t cellsX = (tex.width + (1u << shift) - 1u) >> shift, cellsY = (tex.height + (1u << shift) - 1u) >> shift;
    const uint32_t n = cellsX * cellsY;
    if (n == 0) return;
    hipLaunchKernelGGL(
        build_alpha_bounds_kernel, dim3((n + 255u) / 256u), dim3(256), 0, stream, tex, wrapS, wrapT, factorA, shift, cellsX,
        cellsY, out);
}

// The monotonicity defect of the device's sRGBtoLinear over the float bit patterns [firstBits, lastBits] (non-negative
// floats in increasing order): max over x of (max of L over the <= 4096 + 1024 inputs before x) - L(x), as float bits in
// out[0] (0 = monotone), and the number of adjacent pairs with L(next) < L(x) in out[1].  One thread per run of 4096
// inputs, started 1024 inputs early so that a defect across a run boundary is seen.
__global__ __launch_bounds__(256) void srgb_monotonicity_kernel(uint32_t firstBits, uint32_t lastBits, uint32_t *__restrict__ out)
{
    const uint64_t run = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t begin = (uint64_t)firstBits + run * 4096ull;
    if (begin > lastBits) return;
    const uint64_t end = begin + 4095ull < lastBits ? begin + 4095ull : lastBits;
    const uint64_t warm = begin >= (uint64_t)firstBits + 1024ull ? begin - 1024ull : firstBits;
    float runMax = srgb_to_linear(__builtin_bit_cast(float, (uint32_t)warm));
    float prev = runMax, defect = 0.0f;
    uint32_t decreases = 0;
    for (uint64_t b = warm + 1ull; b <= end; ++b)
    {
        const float y = srgb_to_linear(__builtin_bit_cast(float, (uint32_t)b));
        if (b >= begin)
        {
            defect = fmax_(defect, runMax - y);
            decreases += y < prev ? 1u : 0u;
        }
        runMax = fmax_(runMax, y);
        prev = y;
    }
    if (defect > 0.0f) atomicMax(&out[0], __builtin_bit_cast(uint32_t, defect));
    if (decreases) atomicAdd(&out[1], decreases);
}

void launch_srgb_monotonicity(uint32_t firstBits, uint32_t lastBits, uint32_t *out, hipStream_t stream)
{
    const uint64_t runs = ((uint64_t)lastBits - firstBits) / 4096ull + 1ull;
    hipLaunchKernelGGL(srgb_monotonicity_kernel, dim3((uint32_t)((runs + 255ull) / 256ull)), dim3(256), 0, stream, firstBits, lastBits, out);
}

// Interleaves the three tiled RGBA8 textures of a material into a MaterialPack (pt_scene.hpp): one thread per texel of
// the padded pack extent (texels outside the image are zero and never addressed).
// The sky cube with a one-texel border per face (pt_device.hpp fetch_cube_rgb): out[face][j + 1][i + 1] = texel (i, j) of
// the face, taken from the neighbouring face by the seamless-edge rule where (i, j) is outside.
__global__ __launch_bounds__(256) void border_skybox_kernel(const uint16_t *__restrict__ cube, uint32_t n, uint2 *__restrict__ out)
{
    const uint32_t n2 = n + 2u;
    const uint32_t x = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t y = blockIdx.y, face = blockIdx.z;
    if (x >= n2) return;
    out[((size_t)face * n2 + y) * n2 + x] = cube_texel_seamless(cube, (int32_t)n, face, (int32_t)x - 1, (int32_t)y - 1);
}

void launch_border_skybox(const uint16_t *cube, uint32_t faceSize, void *bordered, hipStream_t stream)
{
    hipLaunchKernelGGL(
        border_skybox_kernel, dim3((faceSize + 2u + 255u) / 256u, faceSize + 2u, 6u), dim3(256), 0, stream, cube, faceSize,
        static_cast<uint2 *>(bordered));
}

template <bool COMPACT>
__global__ __launch_bounds__(256) void pack_material_textures_kernel(
    DeviceTexture base, DeviceTexture mr, DeviceTexture normal, MaterialPack pack, void *__restrict__ out, uint32_t paddedW,
    uint32_t paddedH)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t j = blockIdx.y;
    if (i >= paddedW || j >= paddedH) return;
    uint4 t = make_uint4(0u, 0u, 0u, 0u);
    if (i < pack.width && j < pack.height)
    {
        t.x = reinterpret_cast<const uint32_t *>(base.texels)[texel_offset(base, (int32_t)i, (int32_t)j)];
        t.y = reinterpret_cast<const uint32_t *>(mr.texels)[texel_offset(mr, (int32_t)i, (int32_t)j)];
        t.z = reinterpret_cast<const uint32_t *>(normal.texels)[texel_offset(normal, (int32_t)i, (int32_t)j)];
    }
    const uint32_t at = pack_texel_offset(pack, (int32_t)i, (int32_t)j);
    if constexpr (COMPACT)
        // {R G B roughness (MR.g)}, {metallic (MR.b) Nx Ny Nz}
        static_cast<uint2 *>(out)[at] =
            make_uint2((t.x & 0x00FFFFFFu) | ((t.y << 16) & 0xFF000000u), ((t.y >> 16) & 0xFFu) | (t.z << 8));
    else
        static_cast<uint4 *>(out)[at] = t;
}

void launch_pack_material_textures(
    const DeviceTexture &base, const DeviceTexture &mr, const DeviceTexture &normal, const MaterialPack &pack, hipStream_t stream)
{
    const uint32_t paddedW = pack.tilesPerRow * kPackTileW;
    const uint32_t paddedH = ((pack.height + kPackTileH - 1u) / kPackTileH) * kPackTileH;
    const dim3 grid((paddedW + 255u) / 256u, paddedH);
    if (pack.sampler & kPackCompactBit)
        hipLaunchKernelGGL(pack_material_textures_kernel<true>, grid, dim3(256), 0, stream, base, mr, normal, pack, const_cast<void *>(pack.texels), paddedW, paddedH);
    else
        hipLaunchKernelGGL(pack_material_textures_kernel<false>, grid, dim3(256), 0, stream, base, mr, normal, pack, const_cast<void *>(pack.texels), paddedW, paddedH);
}

void launch_eval_fn(
    uint32_t fn, const float *in, uint32_t inStride, float *out, uint32_t outStride, uint32_t n, hipStream_t stream)
{
    if (n == 0) return;
    hipLaunchKernelGGL(eval_fn_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, fn, in, inStride, out, outStride, n);
}

} // namespace ppt
