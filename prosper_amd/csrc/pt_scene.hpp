// pt_scene.hpp — HBM-resident scene layout of the HIP path tracer.
//
// What the RT pass reads through prosper's descriptor sets (src/render/RtReference.cpp:244-257)
// lives here as plain device pointers; the acceleration structure the Vulkan driver would own
// (src/scene/World.cpp:585-802) is this build's own flattened BVH (DESIGN.md "BVH").
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/prosper_pt/prosper_pt.h"

namespace ppt
{

// 4-wide BVH node, 80 B = five 16-byte words.  The four child boxes are stored as binary16 OFFSETS
// from the node's own origin (the minimum corner of the union of its children, an fp32 point),
// rounded outward (lo down, hi up), one 8-byte group per plane: lo[axis][child], hi[axis][child].
// A half has 11 significant bits: stored as absolute coordinates a 5 cm box 10 m from the world
// origin would grow by ~8 mm per side (+60 % surface area, i.e. +60 % false-positive visits at the
// leaf level, where most visits happen); relative to the node the rounding is <= extent / 2048.
// A child reference >= 0 is an inner node index; < 0 is a leaf: ~ref = (firstTriangle << 3) |
// (triangleCount - 1).  An unused child has lo = hi = +inf, which no ray can enter (bvh_build.cpp).
// One node visit tests four boxes: half the dependent fetches per ray of a binary tree.
struct alignas(16) BvhNode
{
    float origin[3];
    uint32_t reserved; // number of children in use (slots 0 .. reserved - 1); the traversal does not read it
    uint16_t lo[3][4];
    uint16_t hi[3][4];
    int32_t child[4];
};
static_assert(sizeof(BvhNode) == 80, "BVH node is 80 B");

constexpr uint32_t kMaxLeafTriangles = 4;
constexpr uint32_t kTraversalStackDepth = 32; // most LDS stack entries per lane a kernel variant has
// worst-case stack entries a tree may need; entries beyond the LDS stack spill to a global array
constexpr uint32_t kMaxStackBound = 96;

// World-space triangle, 48 B = 3 x dwordx4, stored in BVH leaf order.  Vertices are the fp16
// positions of the bindless geometry buffers (reference: src/scene/Mesh.hpp:11-12) decoded and
// multiplied by the instance's modelToWorld on the GPU at upload (kernel flatten_triangles).
struct alignas(16) WorldTriangle
{
    float v0[3];
    uint32_t drawInstance;
    float v1[3];
    uint32_t primitive;
    float v2[3];
    uint32_t flags; // bit 0: opaque geometry (World.cpp:646-651); bit 1: u16-indexed mesh; bits 2..31 of a
                    // non-opaque triangle: its record in DeviceScene::alphaTriangles
};
static_assert(sizeof(WorldTriangle) == 48, "world triangle is 48 B");
constexpr uint32_t kTriFlagOpaque = 1u;
constexpr uint32_t kTriFlagShortIndices = 2u; // the mesh is u16-indexed (only the byte model cares)
constexpr uint32_t kTriAlphaShift = 2u;

// sampleAlpha (scene/materials.glsl:121-147) of one material as the any-hit needs it, 32 B: the base-colour texture
// (tiled RGBA8, as DeviceTexture), its sampler's wrap / filter, baseColorFactor.a, the cutoff, the mode - and
// `bounds`, the material's ALPHA BOUNDS: one {lo, hi} byte pair per cell of 2^cellShift x 2^cellShift texels, bounding
// the final alpha (after the bilinear filter, sRGBtoLinear and the factor) of EVERY sample whose footprint starts in
// that cell.  A candidate whose bounds already decide the test - MASK: hi < cutoff or lo >= cutoff; BLEND: hi == 0,
// u > hi or u <= lo - needs no texel fetch and no pow(); the others run the exact code, so the result is the exact
// code's in every case (DESIGN.md "alpha bounds").  bounds == nullptr: no table (no texture, or a factor outside
// [0, inf)): always exact.
struct alignas(16) AlphaMaterial
{
    const uint8_t *texels;  // nullptr: no base-colour texture, alpha = baseColorFactor.a
    const uint16_t *bounds; // lo | hi << 8 per cell; hi == 255 means unbounded
    uint16_t width, height;
    float factorA;
    float cutoff;
    uint32_t bits; // 0-1 alphaMode, 2-3 wrapS, 4-5 wrapT, 6 nearest filter, 8-11 cellShift
};
static_assert(sizeof(AlphaMaterial) == 32, "alpha material is 32 B");

// What the any-hit shader (rt/scene.rahit:18-39) needs of a non-opaque triangle, in ONE 64-byte line instead of
// draw instance -> 128-byte shading record -> material -> texture -> sampler: the three texCoord0 (two halfs each, as
// the vertex stream holds them), the ids an accepted candidate reports, and a copy of its material's AlphaMaterial.
// Written by flatten_triangles next to the shading records, (non-opaque drawInstance, primitive) order.
struct alignas(64) AlphaTriangle
{
    uint32_t uv[3];
    uint32_t drawInstance;
    uint32_t primitive;
    uint32_t materialIndex;
    uint32_t reserved[2];
    AlphaMaterial material;
};
static_assert(sizeof(AlphaTriangle) == 64, "alpha triangle is 64 B");
constexpr float kAlphaFilterSlack = 4e-6f; // > 6x the rounding of the bilinear weights and the fma chain
constexpr float kAlphaCurveSlack = 4e-6f;  // > any non-monotonicity of the device's sRGBtoLinear (tested over all inputs)

// Decoded object-space corner attributes of one triangle, 128 B = eight 16-byte words, stored in
// (drawInstance, primitive) order (record = triangleOffsets[drawInstance] + primitive).  What
// geometry.glsl:220-256 fetches per hit through drawInstance -> mesh metadata -> index buffer -> four
// vertex streams (three dependent round trips, 3 x (unpack + normalize) x 2) is precomputed ONCE at
// upload by the same device functions, so a hit reads one record: same bits, five hops fewer.
//   q0..q2: normal of corner i (xyz, unpackSnorm + normalize), w = texCoord0 of corner i (two halfs)
//   q3..q5: tangent of corner i (xyz, w = sign)
//   q6: position halfs of corner 0 (x: xy, y: z_) and corner 1 (z, w);  q7: corner 2 (x, y), z = flags
struct alignas(16) ShadeTriangle
{
    float normalUv[3][4];
    float tangent[3][4];
    uint32_t position[3][2];
    uint32_t flags; // bit 1: u16-indexed mesh (kTriFlagShortIndices)
    uint32_t reserved;
};
static_assert(sizeof(ShadeTriangle) == 128, "shade triangle is 128 B");

// The same three corners as the vertex streams hold them, 64 B, kept INSTEAD of ShadeTriangle under
// PROSPER_PT_DEBUG_RAW_RECORDS=1 (DeviceScene::rawShadeTriangles) and decoded per hit by the very functions
// flatten_triangles runs for the decoded record: same bits.  An experiment (round-2 verdict item 6): half the record bytes
// against ~150 more instructions per hit (six unpackSnorm + normalize) - slower on every configuration, also on
// S-sponza-class whose 33.6 MB of records outgrow the L2 (profiles/r03_raw_records.txt).
//   q0: position halfs of corner 0 (xy, z_) and 1;  q1: corner 2, normal 0, normal 1 (snorm10);
//   q2: normal 2, tangents 0..2 (snorm10, sign in the top two bits);  q3: texCoord0 0..2 (two halfs), flags
struct alignas(64) RawShadeTriangle
{
    uint32_t position[3][2];
    uint32_t normal[3];
    uint32_t tangent[3];
    uint32_t uv[3];
    uint32_t flags; // bit 1: u16-indexed mesh; bit 2: the mesh has no normals; bit 3: no tangents
};
static_assert(sizeof(RawShadeTriangle) == 64, "raw shade triangle is 64 B");
constexpr uint32_t kRawNoNormals = 4u, kRawNoTangents = 8u;

// RGBA8 texels in 8 x 4 tiles of one 128-byte cache line each (texel (i, j) at tile (i >> 3, j >> 2),
// row-major inside the tile; the allocation is padded to whole tiles).  The 2 x 2 footprint of a bilinear
// fetch then falls into one line two times out of three instead of always two (rows are `width * 4` bytes
// apart in a linear image): 1.4 instead of 2.1 lines per sample at LOD 0, where neighbouring pixels already
// land on unrelated texels.
struct DeviceTexture
{
    const uint8_t *texels;
    uint32_t width;
    uint32_t height;
    uint32_t tilesPerRow; // (width + 7) / 8
    uint32_t pad;
};
constexpr uint32_t kTexTileW = 8, kTexTileH = 4;

// The three textures of a material - base colour, metallic-roughness, normal - are sampled at the SAME uv of every hit
// (materials.glsl:47-119).  When they have the same extent and sampler (the usual glTF export) their texels are also
// kept interleaved: one uint4 per texel = {base RGBA8, MR RGBA8, normal RGBA8, 0}, in 4 x 2-texel tiles of one 128-byte
// line.  A hit's bilinear footprint is then four 12-byte loads out of ~1.9 lines instead of twelve 4-byte loads out of
// ~4.2 lines, and one footprint computation instead of three.  Same texel values, same filter arithmetic: same bits.
// texels == nullptr: the material is not packable (a texture missing, extents or samplers differ) and samples its
// textures one by one.
// An OPAQUE material reads eight of those twelve bytes (materials.glsl:47-119: base.rgb, MR.g = roughness, MR.b =
// metallic, normal.rgb; base.a only matters to MASK / BLEND): its pack is the COMPACT one (kPackCompactBit in `sampler`),
// one uint2 per texel = {R G B roughness, metallic Nx Ny Nz}, in the same 4 x 2-texel tiles - now one 64-byte sector each.
// Half the texel memory (S-sponza-class: 420 -> 210 MB, inside the 256 MB Infinity Cache) and a footprint of four 8-byte
// loads; every channel still goes through the same byte -> float -> filter arithmetic: same bits.
struct MaterialPack
{
    const void *texels;   // uint4 per texel, or uint2 (compact)
    uint32_t width;
    uint32_t height;
    uint32_t tilesPerRow; // (width + 3) / 4
    uint32_t sampler;     // sampler index | kPackCompactBit
};
constexpr uint32_t kPackTileW = 4, kPackTileH = 2;
constexpr uint32_t kPackCompactBit = 0x80000000u;

// Everything a kernel needs about the scene; passed by value as a kernel argument so every
// pointer arrives in SGPRs.
struct DeviceScene
{
    const BvhNode *nodes;
    const WorldTriangle *triangles;
    const ShadeTriangle *shadeTriangles; // (drawInstance, primitive) order; nullptr when the scene keeps raw records
    const RawShadeTriangle *rawShadeTriangles; // the 64-byte form (PROSPER_PT_DEBUG_RAW_RECORDS=1), else nullptr
    const uint32_t *triangleOffsets;     // first record of each draw instance
    const void *const *geometryBuffers;  // device array of device pointers
    const prosper_GeometryMetadata *geometryMetadatas;
    const prosper_DrawInstance *drawInstances;
    const prosper_ModelInstanceTransforms *modelInstanceTransforms;
    const prosper_MaterialData *materials;
    const DeviceTexture *textures;
    const MaterialPack *materialPacks; // [materialCount]
    const AlphaTriangle *alphaTriangles; // non-opaque triangles, (drawInstance, primitive) order
    const uint32_t *alphaOffsets;        // [drawInstanceCount]: first alpha record of a non-opaque draw instance
    const AlphaMaterial *alphaMaterials; // [materialCount]
    const prosper_pt_sampler_desc *samplers;
    const prosper_DirectionalLightParameters *directionalLight;
    const prosper_PointLightsBuffer *pointLights;
    const prosper_SpotLightsBuffer *spotLights;
    const uint16_t *skybox; // RGBA16F, 6 faces of (skyboxFaceSize + 2)^2 texels: a one-texel seamless border (pt_device.hpp fetch_cube_rgb)
    uint32_t skyboxFaceSize;
    uint32_t pointLightCount; // snapshot of pointLights->count at upload (kept in SGPRs)
    uint32_t spotLightCount;
    uint32_t materialCount; // table sizes, for staging the tables in LDS (wf_shade)
    uint32_t drawInstanceCount;
    uint32_t modelInstanceCount;
    // texel loads of a hit's three textures issued together (sample_material<true>): set for texture sets too big
    // for the caches, where the three dependent HBM round trips per hit are what the shade kernel waits for
    uint32_t batchedTextures;
};

// fields of prosper_pt_counters (uint64 each), per kernel stage
constexpr uint32_t kCounterCount = 17;
static_assert(sizeof(prosper_pt_counters) == kCounterCount * 8, "prosper_pt_counters and the device counter block differ");

// Per-launch constants: push constants, the camera terms the path reads, extent and tile.
struct RenderParams
{
    prosper_ReferencePC pc;
    // rt/ray.glsl:21-35,72 read only these parts of CameraUniforms
    float eye[3];
    float right[3];
    float up[3];
    float fwd[3];
    float aspect;      // cameraToClip[1][1] / cameraToClip[0][0] (ray.glsl:21-24): per-frame constants, divided once
    float tanHalfFovY; // 1 / cameraToClip[1][1]                        on the host (IEEE fp32 division, same bits)
    float cameraToWorld[16];
    uint32_t width;
    uint32_t height;
    uint32_t stripeWidth; // 0 = untiled
    uint32_t stripeIndex;
    uint32_t stripeCount;
    uint32_t localWidth;
    uint32_t frameCount; // consecutive accumulated frames rendered by this launch
    // audit switch (PROSPER_PT_DEBUG_TRACE_DEAD_PATHS=1): keep tracing paths whose throughput is exactly zero, as the
    // GLSL does; the images must not differ (arithmetic contract)
    uint32_t traceDeadPaths;
    // wf_trace: the four segments of a workgroup are traced by ONE of its waves when together they hold at most this many
    // shadow rays and at most this many closest-hit rays (pt_wavefront.hip "sparse segments"); 0 = never
    uint32_t mergeLimit;
};

// Workspace of the wavefront pipeline (pt_wavefront.hip).  Paths live in fixed-length SEGMENTS of
// `segLen` slots, one segment per wave; every stage compacts its survivors to the front of the
// wave's own segment with ballot/popcount, so there is no global queue counter and no atomic.
// slot = frame * pixelsPadded + tile * 64 + laneInTile identifies the (pixel, accumulated frame).
struct WavefrontBuffers
{
    float4 *rayA[2];  // (origin.xyz, closest-ray seed)        ping-pong by bounce parity
    float4 *rayB[2];  // (direction.xyz, rng state x)
    float4 *pathT[2]; // (throughput.rgb, slot); camera paths have no record - their throughput is (1, 1, 1) - only
    uint32_t *cameraSlot; // their slot (generate writes it, the first shade reads it).  An array of its own, indexed like the
                          // others: records of different strides must not share memory - the two chains of an in-order
                          // render are at different bounces at the same time, each in its own segments
    uint2 *pathR[2];  // (rng state y, z)
    uint4 *hit;       // (drawInstance, primitive, bary.u, bary.v), compacted per segment
    uint32_t *hitIdx; // in-segment index of the ray each compacted hit came from
    float4 *shA;      // shadow rays: (origin.xyz, seed)
    float4 *shB;      // (direction.xyz, distance)
    float4 *shC;      // (unoccluded contribution.rgb, slot | nanMask << 28)
    float4 *color;    // per slot: radiance of that path so far
    uint32_t *segRays;   // [nSeg] live rays in rayA/B[cur]
    uint32_t *segHits;   // [nSeg]
    uint32_t *segShadow; // [nSeg]
    uint32_t segLen;
    uint32_t nSeg;
    uint32_t pixelsPadded; // tilesX * tilesY * 64
    uint32_t tilesX;
    uint32_t tilesY;
    // tiles in the order the camera-ray batches take them: heaviest first, by the cost of a probe ray through the tile's
    // centre (pt_wavefront.hip tile order); nullptr = raster order
    const uint32_t *tileOrder;
    // Banded batches (pt_wavefront.hip batch dealing): the segments [x * bandSegments, (x + 1) * bandSegments) - the ones XCD x
    // runs when a launch covers all groups - take the camera-ray batches of the tiles [bandTile[x], bandTile[x + 1]) only, so
    // that the paths an XCD traces start in one band of the image and the part of the scene they see stays in its L2.
    // bandSegments == 0: batches strided over the whole image.
    uint32_t bandSegments;
    uint32_t bandTile[9];
    uint32_t groupBase;  // this launch covers segment groups [groupBase, groupBase + groupCount)
    uint32_t groupCount; // (a group = the 4 segments of one workgroup)
};

} // namespace ppt
