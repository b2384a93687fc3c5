// pt_kernels.hpp — host-callable launchers of the gfx950 kernels in pt_kernels.hip.
#pragma once

#include <hip/hip_runtime.h>

#include "pt_scene.hpp"

namespace ppt
{

// Stage index of a render kernel: selects its block of 16 work counters and its timing bucket.
enum : uint32_t
{
    kStageGenerate = 0, // wf_generate_extend, or the single kernel of the megakernel/persistent pipelines
    kStageShade = 1,
    kStageTrace = 2,
    kStageAccumulate = 3,
    kStageCount = 4,
};

// Optional per-launch hipEvent timestamps (prosper_pt_set_kernel_timing): one event before every
// launch, one after the last; interval i belongs to stage[i].
struct LaunchTimer
{
    hipEvent_t *events = nullptr;
    uint32_t *stage = nullptr;
    uint32_t count = 0; // intervals recorded so far (events used = count + 1 once closed)
    uint32_t capacity = 0;
    void mark(uint32_t st, hipStream_t stream)
    {
        if (events && count < capacity)
        {
            (void)hipEventRecord(events[count], stream);
            stage[count] = st;
            ++count;
        }
    }
    void close(hipStream_t stream)
    {
        if (events) (void)hipEventRecord(events[count], stream);
    }
};

void launch_flatten_triangles(
    const DeviceScene &s, const uint32_t *triOffsets, uint32_t drawInstanceCount, const uint32_t *drawInstanceFlags,
    WorldTriangle *out, ShadeTriangle *shadeOut, AlphaTriangle *alphaOut, uint32_t total, hipStream_t stream,
    const uint32_t *leafPosition = nullptr, WorldTriangle *leafOrder = nullptr, RawShadeTriangle *rawOut = nullptr);
// refit of an unchanged tree after moved instances: exact bounds level by level (`order` = nodes by height,
// levelOffsets[levels + 1] on the HOST), then every node's boxes re-encoded; *cost += the tree's surface-area measure
void launch_refit(
    BvhNode *nodes, const WorldTriangle *tris, float4 *bounds, const uint32_t *order, const uint32_t *levelOffsets,
    uint32_t levels, uint32_t nodeCount, float padCoeff, float *cost, hipStream_t stream);
void launch_build_alpha_bounds(
    const DeviceTexture &tex, uint32_t wrapS, uint32_t wrapT, float factorA, uint32_t shift, uint16_t *out, hipStream_t stream);
void launch_srgb_monotonicity(uint32_t firstBits, uint32_t lastBits, uint32_t *out, hipStream_t stream);
void launch_permute_triangles(
    const WorldTriangle *in, const uint32_t *permutation, WorldTriangle *out, uint32_t total, hipStream_t stream);
uint32_t restir_grid_blocks(uint32_t width, uint32_t height);
void launch_restir_di_trace(
    const DeviceScene &s, uint32_t drawType, uint32_t frameIndex, uint32_t flags, uint32_t width, uint32_t height,
    const float eye[3], const float clipToWorld[16], const void *albedoRoughness, const void *normalMetallic,
    const float *nonLinearDepth, const void *reservoirs, float4 *hdr, int32_t *stackOverflow, hipStream_t stream);
void launch_tone_map(
    const float4 *hdr, const uint32_t *lut, uint32_t dim, float exposure, float contrast, void *outRgba8, uint32_t count,
    hipStream_t stream);
uint32_t megakernel_grid_blocks(const RenderParams &p);
#ifdef PPT_EXPERIMENTS
uint32_t persistent_grid_blocks();
#endif
uint32_t wavefront_grid_blocks(const WavefrontBuffers &w);
// tiles ordered by the cost of a probe ray (heaviest first) into order[tilesX * tilesY]; `scratch` holds
// tilesX * tilesY + 512 more uint32s; `stackOverflow` as for the render's traversal kernels (the probe's grid is smaller)
#ifdef PPT_EXPERIMENTS
void launch_tile_order(
    const DeviceScene &s, const RenderParams &p, uint32_t tilesX, uint32_t tilesY, uint32_t ldsStackEntries, int32_t *stackOverflow,
    uint32_t *order, uint32_t *scratch, hipStream_t stream);
#endif
// `stackOverflow`: global array of (stack bound - LDS entries) x (grid lanes) ints, or nullptr when the
// BVH's stack bound fits the kernel's LDS stack
void launch_render_megakernel(
    const DeviceScene &s, const RenderParams &p, float4 *hdr, unsigned long long *counters, int32_t *stackOverflow,
    bool countWork, hipStream_t stream);
#ifdef PPT_EXPERIMENTS
void launch_render_persistent(
    const DeviceScene &s, const RenderParams &p, float4 *hdr, unsigned long long *counters, uint32_t *workCounter,
    int32_t *stackOverflow, bool countWork, hipStream_t stream);
#endif
// The wavefront pipeline runs its segment groups as `count` independent chains of launches (generate,
// shade/trace per bounce), chain i on streams[i] with its own launch timer, forked from and joined back
// into the caller's stream around them; the accumulate kernel follows on the caller's stream.  With
// count == 1 (or a batch too small to split) everything runs on the caller's stream.
// `detached`: the (single) chain runs on streams[0] without forking from the caller's stream - it waits only
// for `after` (the previous use of its workspace) - and is joined back before the accumulate kernel.
constexpr uint32_t kMaxChains = 3;
struct WavefrontChains
{
    uint32_t count = 1;
    bool detached = false;
    hipEvent_t after = nullptr;
    hipEvent_t scene = nullptr; // the last prosper_pt_update_transforms: every chain waits for it
    hipEvent_t lights = nullptr; // the last prosper_pt_update_lights
    hipEvent_t materials = nullptr; // the last prosper_pt_update_textures / _materials
    hipStream_t streams[kMaxChains] = {};
    hipEvent_t fork = nullptr;
    hipEvent_t join[kMaxChains] = {};
    LaunchTimer *timers[kMaxChains] = {};
};
// Which traversal-kernel variants a render of a tree with this stack bound takes, and the global scratch they need
struct WavefrontPlan
{
    uint32_t ldsStackEntries;       // 16 / 24 / 32: LDS stack entries per lane of the lane-owned traversal kernels
    uint32_t overflowEntries;       // their stack entries per lane in global memory (stack bound - LDS entries)
    bool sceneInLds;                // a scene of a few KB is traversed out of an LDS copy
    bool tablesInLds;               // wf_shade stages the scene tables in LDS
    bool hipGraph;                  // experiment: a detached chain's launches through a HIP graph
    uint32_t poolVariant;           // 0: wf_trace walks its rays lane-owned (trace_stream); else index + 1 of the ray-pool variant
    uint32_t poolOverflowEntries;   // stack entries per pool slot in global memory
    uint32_t scratchDwordsPerBlock; // ints of `scratch` per workgroup: the larger of the two kernels' needs
};
// what a context's debug options (prosper_pt_debug_options) say about the kernel variants; all zero = the defaults
struct WavefrontOptions
{
    uint32_t ldsStackEntries = 0; // 16 / 24 / 32 forces the LDS stack size
    bool noLdsScene = false, noLdsTables = false;
    uint32_t poolVariant = 0;     // experiments (-DPPT_EXPERIMENTS): ray-pool wf_trace, HIP graph
    bool hipGraph = false;
};
WavefrontPlan wavefront_plan(
    uint32_t stackBound, uint32_t nodeCount, uint32_t triCount, const DeviceScene &s, const WavefrontOptions &opt);
// `scratch`: plan.scratchDwordsPerBlock ints per workgroup of the launch grid (nullptr when that is 0)
void launch_render_wavefront(
    const DeviceScene &s, const RenderParams &p, float4 *hdr, unsigned long long *counters, const WavefrontBuffers &w,
    const WavefrontPlan &plan, int32_t *scratch, uint32_t nodeCount, uint32_t triCount, bool countWork, LaunchTimer *timer,
    const WavefrontChains &chains, hipStream_t stream);
// LDS stack entries (16/24/32) the wavefront traversal kernels use for a tree with this stack bound
uint32_t wavefront_lds_stack_entries(uint32_t stackBound, uint32_t forced);
// kernel variants a render takes: wf_shade with the scene tables staged in LDS; traversal out of an LDS copy of the scene
bool wavefront_shade_tables_in_lds(const DeviceScene &s, bool disabled);
bool wavefront_scene_in_lds(uint32_t ldsStackEntries, uint32_t stackBound, uint32_t nodeCount, uint32_t triCount, bool disabled);
void launch_blit_rgba16f(const float4 *in, void *out, uint32_t count, hipStream_t stream);
// Where each rank's tile sits in the gathered buffer (texel offsets) and how wide its rows are
constexpr uint32_t kMaxRanks = 64;
struct TileLayout
{
    uint32_t width, height, stripeWidth, ranks;
    uint32_t localWidth[kMaxRanks];
    uint64_t tileOffset[kMaxRanks];
};
void launch_deinterleave_tiles(const float4 *tiles, const TileLayout &layout, float4 *full, hipStream_t stream);
// BC7 blocks of one level (row-major, 16 B each; width and height multiples of 4) -> RGBA8 texels in the tiled
// layout of DeviceTexture
void launch_decode_bc7(
    const void *blocks, uint32_t width, uint32_t height, uint32_t tilesPerRow, void *tiled, hipStream_t stream);
// width x height row-major RGBA8 texels (device memory) -> the tiled layout of DeviceTexture, padding zeroed
void launch_retile_rgba8(const void *linear, uint32_t width, uint32_t height, uint32_t tilesPerRow, void *tiled, hipStream_t stream);
// every any-hit record's copy of its material's AlphaMaterial from `table` again
void launch_patch_alpha_records(AlphaTriangle *records, uint32_t count, const AlphaMaterial *table, uint32_t materialCount, hipStream_t stream);
// fills `pack.texels` (device memory, tilesPerRow * 4 x ceil(height / 2) * 2 uint4s) from the three tiled textures
// 6 x n x n RGBA16F texels -> 6 x (n + 2) x (n + 2) with the seamless border
void launch_border_skybox(const uint16_t *cube, uint32_t faceSize, void *bordered, hipStream_t stream);
void launch_pack_material_textures(
    const DeviceTexture &base, const DeviceTexture &mr, const DeviceTexture &normal, const MaterialPack &pack, hipStream_t stream);
void launch_eval_fn(
    uint32_t fn, const float *in, uint32_t inStride, float *out, uint32_t outStride, uint32_t n, hipStream_t stream);

} // namespace ppt
