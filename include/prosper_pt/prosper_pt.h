/*
 * prosper_pt/prosper_pt.h — C-ABI of the MI355X path-tracing reference pass.
 *
 * This library sits where prosper records `cb.traceRaysKHR(rgen, miss, hit, callable, W, H, 1)`
 * (reference: src/render/RtReference.cpp:328-330).  prosper has no FFI for this path — the pass is
 * a concrete C++ class (src/render/RtReference.hpp:32-60) — so the entry points below are the
 * smallest plain-C surface that class needs: they replace, one for one,
 *
 *   prosper_pt_create / _destroy      RtReference::init / ~RtReference            RtReference.cpp:92-120
 *                                     (pipeline + SBT creation -> load the gfx950 code object)
 *   prosper_pt_upload_scene           World::updateBuffers + buildNextBlas + buildCurrentTlas
 *                                                                                  src/scene/World.cpp:468-536,585-802
 *                                     and the descriptor sets the pass binds       RtReference.cpp:238-274
 *   prosper_pt_update_lights          lights ring write                            World.cpp:531-535
 *   prosper_pt_update_transforms      instance transforms + TLAS rebuild           World.cpp:359-466,749-802,878-928
 *   prosper_pt_update_textures /      adoption of streamed-in images / materials   src/scene/WorldData.cpp:568-647,2182-2239
 *   prosper_pt_update_materials
 *   prosper_pt_update_meshes          adoption of streamed-in meshes, BLAS once    WorldData.cpp:2003-2110, World.cpp:585-606,
 *                                     a model is complete, inactive until then     909-915
 *   prosper_pt_render                 pushConstants + traceRaysKHR                 RtReference.cpp:278-330
 *                                     + the previous/illumination ping-pong        RtReference.cpp:178-219,332-334
 *   prosper_pt_read_hdr               the RGBA32F "rtIllumination" image           RtReference.cpp:178-187
 *   prosper_pt_blit_rgba16f           blitImage RGBA32F -> RGBA16F                 RtReference.cpp:339-377
 *   prosper_pt_get_counters           (new) deterministic work counters for the roofline model
 *   prosper_pt_comm_* / _gather_tiles (new) multi-GPU: stripes per rank + one RCCL gather + de-interleave kernel
 *                                     (the reference asserts renderArea.offset == 0, RtReference.cpp:327)
 *
 * All entry points are `extern "C"`, take PODs / plain pointers and sizes, never throw and return
 * PROSPER_PT_OK (0) or a negative error code; prosper_pt_last_error() returns the message of the
 * calling thread's last failure.  A context is single-threaded for its callers (the reference makes all pass
 * calls on the main thread, src/Allocators.hpp:9); use one context per GPU.  While meshes stream in the context
 * runs ONE worker thread of its own (prosper_pt_update_meshes), which touches nothing a caller can see.
 *
 * Host pointers inside prosper_pt_scene_view are borrowed for the duration of the call only.
 */
#ifndef PROSPER_PT_H
#define PROSPER_PT_H

#include <stddef.h>
#include <stdint.h>

#include "shader_structs.h"

#ifdef __cplusplus
extern "C" {
#endif

#define PROSPER_PT_ABI_VERSION 4

enum
{
    PROSPER_PT_OK = 0,
    PROSPER_PT_ERR_INVALID_ARGUMENT = -1,
    PROSPER_PT_ERR_NO_DEVICE = -2,  /* no HIP device / HIP extension unusable: never falls back to CPU */
    PROSPER_PT_ERR_HIP = -3,        /* a HIP runtime call failed */
    PROSPER_PT_ERR_NO_SCENE = -4,   /* render before upload_scene */
    PROSPER_PT_ERR_SCENE = -5,      /* scene view failed validation (out-of-range index/offset) */
    PROSPER_PT_ERR_UNSUPPORTED = -6,
};

typedef struct prosper_pt_ctx prosper_pt_ctx;

typedef struct prosper_pt_device_desc
{
    uint32_t struct_size;   /* sizeof(prosper_pt_device_desc) */
    int32_t device_ordinal; /* HIP device index; the pass runs on this GPU only */
    uint32_t flags;         /* PROSPER_PT_CREATE_* */
    uint32_t reserved;
} prosper_pt_device_desc;

enum
{
    /* Default pipeline: wavefront stage kernels (generate+extend / shade / shadow / extend /
     * accumulate) with per-wave ballot compaction.  The two alternatives produce the same pixels
     * and are kept for A/B timing: MEGAKERNEL = one lane per pixel runs whole paths,
     * PERSISTENT = resident waves that regenerate finished paths from a global counter. */
    PROSPER_PT_CREATE_MEGAKERNEL = 1u << 0,
    PROSPER_PT_CREATE_PERSISTENT = 1u << 1,
    /* run the wavefront pipeline as ONE chain of launches on the caller's stream instead of two
     * half-batches on two internal streams (A/B switch; the two-chain default hides launch tails) */
    PROSPER_PT_CREATE_SINGLE_CHAIN = 1u << 2,
};

/* Texel formats of material textures (reference: src/scene/Texture.cpp:217-296 stores UNORM,
 * sRGB decode happens in the shader, materials.glsl:56). */
enum
{
    PROSPER_PT_FORMAT_RGBA8_UNORM = 0,
    /* level 0 of prosper's texture cache as it is (src/scene/Texture.cpp:255-287 compresses every texture whose
     * mip chain divides by 4; src/utils/Dds.cpp:118-131 layout): (width/4)*(height/4) 16-byte blocks, row-major;
     * width and height are multiples of 4.  Decoded once at upload by a HIP kernel to the texels the GPU's
     * BC7 sampler returns in prosper. */
    PROSPER_PT_FORMAT_BC7_UNORM = 1,
};
enum
{
    PROSPER_PT_FILTER_NEAREST = 0,
    PROSPER_PT_FILTER_LINEAR = 1,
};
enum
{
    PROSPER_PT_WRAP_REPEAT = 0,
    PROSPER_PT_WRAP_MIRRORED_REPEAT = 1,
    PROSPER_PT_WRAP_CLAMP_TO_EDGE = 2,
};

/* One entry of the bindless materialTextures[] table (materials.glsl:23-24); mip 0 only, because
 * RT stages have no derivatives and sample LOD 0. Index 0 is the "no texture" slot. */
typedef struct prosper_pt_texture_desc
{
    const void *texels; /* RGBA8: width*height texels, row-major, tightly packed; BC7: the level's blocks */
    uint32_t width;
    uint32_t height;
    uint32_t format; /* PROSPER_PT_FORMAT_* */
    uint32_t reserved;
} prosper_pt_texture_desc;

/* One entry of materialSamplers[] (src/scene/WorldData.cpp:681-720); index 0 = repeat/linear. */
typedef struct prosper_pt_sampler_desc
{
    uint32_t magFilter; /* PROSPER_PT_FILTER_* (LOD 0 => magnification filter is the one used) */
    uint32_t minFilter;
    uint32_t wrapS; /* PROSPER_PT_WRAP_* */
    uint32_t wrapT;
} prosper_pt_sampler_desc;

/* scene::MeshInfo (src/scene/Mesh.hpp:17-23), needed for the triangle count of each mesh and for
 * the opaque flag (World.cpp:646-651 reads m_materials[info.materialIndex].alphaMode). */
typedef struct prosper_pt_mesh_info
{
    uint32_t vertexCount;
    uint32_t indexCount;
    uint32_t meshletCount;
    uint32_t materialIndex;
} prosper_pt_mesh_info;

/* Skybox cube: RGBA16F, mip 0, faces +X,-X,+Y,-Y,+Z,-Z, each faceSize x faceSize, row-major
 * (reference: src/scene/Texture.cpp:589-636, sampled by textureLod(skybox, d, 0) main.rgen:251). */
typedef struct prosper_pt_cube_desc
{
    const uint16_t *texels; /* 6 * faceSize * faceSize * 4 halfs; NULL = no skybox */
    uint32_t faceSize;
    uint32_t reserved;
} prosper_pt_cube_desc;

/* Everything the pass reads through its nine descriptor sets (RtReference.cpp:244-257). */
typedef struct prosper_pt_scene_view
{
    uint32_t struct_size; /* sizeof(prosper_pt_scene_view) */
    uint32_t reserved;

    /* GEOMETRY_SET: bindless geometry buffers + per-mesh metadata (geometry.glsl:7-55) */
    const void *const *geometryBuffers;
    const uint64_t *geometryBufferByteSizes;
    uint32_t geometryBufferCount;
    uint32_t meshCount;
    const prosper_GeometryMetadata *geometryMetadatas; /* [meshCount]; bufferIndex == PROSPER_PT_ABSENT: the mesh has not
                                                          been loaded yet (prosper_pt_update_meshes) */
    const prosper_pt_mesh_info *meshInfos;             /* [meshCount] */

    /* SCENE_INSTANCES_SET (instances.glsl:8-34) */
    const prosper_DrawInstance *drawInstances; /* [drawInstanceCount] */
    uint32_t drawInstanceCount;
    uint32_t modelInstanceCount;
    const prosper_ModelInstanceTransforms *modelInstanceTransforms; /* [modelInstanceCount] */

    /* MATERIAL_DATAS_SET + MATERIAL_TEXTURES_SET (materials.glsl:7-24) */
    const prosper_MaterialData *materials; /* [materialCount], index 0 = default material */
    uint32_t materialCount;
    uint32_t textureCount;
    const prosper_pt_texture_desc *textures; /* [textureCount], index 0 unused by shading */
    const prosper_pt_sampler_desc *samplers; /* [samplerCount], index 0 = default sampler */
    uint32_t samplerCount;
    uint32_t reserved2;

    /* LIGHTS_SET (lights.glsl:6-23) */
    const prosper_DirectionalLightParameters *directionalLight;
    const prosper_PointLightsBuffer *pointLights;
    const prosper_SpotLightsBuffer *spotLights;

    /* SKYBOX_SET binding 0 (skybox.glsl:4) */
    prosper_pt_cube_desc skybox;
} prosper_pt_scene_view;

/* The set of pixels one context renders.  The reference always renders the whole image
 * (asserts renderArea.offset == 0, RtReference.cpp:327); for multi-GPU tiling the image is cut
 * into vertical stripes `stripeWidth` pixels wide and this context renders stripes
 * s with s % stripeCount == stripeIndex.  RNG seeds use absolute pixel coordinates
 * (main.rgen:227-229), so any partition yields the same pixels as a whole-image render.
 * The context's HDR buffer holds only its own pixels, rows of localWidth texels, stripes in
 * ascending order.  NULL / {0,0,1} = whole image. */
typedef struct prosper_pt_tile_desc
{
    uint32_t stripeWidth;
    uint32_t stripeIndex;
    uint32_t stripeCount;
} prosper_pt_tile_desc;

/* Deterministic work counters (exact integers, independent of scheduling) used to price the
 * ALGORITHMIC bytes of a frame.  Collected only by prosper_pt_render calls made with
 * PROSPER_PT_RENDER_COUNT_WORK; accumulate until prosper_pt_reset_counters. */
typedef struct prosper_pt_counters
{
    uint64_t paths;            /* pixels rendered (one path per pixel per frame) */
    uint64_t closestRays;      /* traceClosest calls */
    uint64_t shadowRays;       /* shadow() calls */
    uint64_t nodeVisits;       /* BVH node fetches, all rays */
    uint64_t triangleTests;    /* ray/triangle tests, all rays */
    uint64_t closestHits;      /* evaluateSurface invocations */
    uint64_t anyHitCalls;      /* any-hit invocations (non-opaque candidates) */
    uint64_t lightSamples;     /* sampleLight calls: sun/point */
    uint64_t spotLightSamples; /* sampleLight calls that picked a spot light */
    uint64_t skyLookups;       /* IBL miss lookups */
    uint64_t pixelsWritten;    /* output texels written */
    uint64_t historyReads;     /* output texels whose history was read */
    uint64_t shortIndexHits;   /* of closestHits + anyHitCalls: those on u16-indexed meshes */
    uint64_t shortIndexTriangleTests; /* of triangleTests: those on u16-indexed meshes */
    uint64_t nodePhaseSteps;     /* wavefront pipeline: wave-level steps of the node phase (x64 lanes = issue slots) */
    uint64_t trianglePhaseSteps; /* same for the triangle phase; lane utilisation = visits / (64 * steps) */
    uint64_t anyHitTexelFetches; /* of anyHitCalls: those that fetched texels (not settled by the material's alpha bounds) */
} prosper_pt_counters;

/* Sizes the roofline model needs about the acceleration structure the library built. */
typedef struct prosper_pt_scene_stats
{
    uint64_t triangleCount; /* world-space triangles (instances expanded) */
    uint64_t nodeCount;
    uint32_t nodeBytes;     /* S_node: bytes fetched per node visit */
    uint32_t triangleBytes; /* bytes fetched per ray/triangle test */
    uint32_t maxDepth;
    uint32_t variantFlags;  /* PROSPER_PT_VARIANT_*: the kernel variants the next render of this scene takes */
    uint64_t deviceBytes;   /* HBM resident bytes for the scene */
    double buildSeconds;    /* acceleration structure: from the first flatten kernel to the uploaded nodes (since round 3
                             * the host BVH build runs on the host's threads WHILE the calling thread uploads textures,
                             * sky, lights and alpha tables: this span contains those too) */
    /* (ABI 2) where prosper_pt_upload_scene spent its time: the whole call, the host-side BVH construction alone
     * (the part a host-side rebuild would pay again), and the texture re-tiling / BC7 decode + copies - the last two
     * overlap, so they no longer add up to the first */
    double uploadSeconds;
    double bvhBuildSeconds;
    double textureSeconds;
    /* (ABI 3) non-opaque triangles (each has a 32-byte any-hit record) and the bytes of the materials' alpha bounds */
    uint64_t alphaTriangleCount;
    uint64_t alphaBoundBytes;
} prosper_pt_scene_stats;
enum
{
    PROSPER_PT_VARIANT_LDS_SCENE = 1u << 0,       /* BVH + triangles staged in LDS by the traversal kernels */
    PROSPER_PT_VARIANT_LDS_TABLES = 1u << 1,      /* wf_shade stages instances/transforms/materials/lights in LDS */
    PROSPER_PT_VARIANT_BATCHED_TEXTURES = 1u << 2, /* the twelve texel loads of a hit issued together */
    PROSPER_PT_VARIANT_TEXTURE_PACKS = 1u << 3,  /* some material's base / MR / normal texels are interleaved per texel */
    PROSPER_PT_VARIANT_RAW_RECORDS = 1u << 4,    /* 64-byte raw shading records decoded per hit (debug option rawRecords: an experiment) */
    PROSPER_PT_VARIANT_STACK_SHIFT = 8,           /* bits 8..15: LDS traversal-stack entries (16/24/32) */
};

enum
{
    PROSPER_PT_RENDER_COUNT_WORK = 1u << 0, /* run the instrumented kernels (slower, same pixels) */
    /* Frames in flight, the role of `nextFrame` / the per-frame descriptor sets in RtReference::record
     * (src/render/RtReference.cpp:161-168; prosper keeps 2 frames in flight): the path stages of this render use the
     * context's NEXT workspace (it owns three) and may start before work enqueued earlier on `stream` - including the
     * two previous renders - has finished; they wait only for the render of three calls ago.  The accumulate kernel (history read,
     * output write) runs on `stream`, in order, so the image and everything enqueued after this call behave as
     * without the flag.  The caller promises that no input of this render (scene, lights) is produced by work still
     * pending on `stream`: uploads through this API are synchronous, so that holds unless the caller writes the
     * library's buffers itself.  Same pixels (tested); ignored with PROSPER_PT_RENDER_COUNT_WORK. */
    PROSPER_PT_RENDER_PIPELINED = 1u << 1,
};

/* Tuning and test options of a context.  NOTHING in the library reads the process environment while it uploads, updates or
 * renders: a host application's environment cannot change what this plugin does.  Tests and measurement scripts fill this
 * struct instead (prosper_pt_debug_options_default, change fields, prosper_pt_set_debug_options); options marked "upload"
 * take effect at the next prosper_pt_upload_scene, the others at the next render / update.  No option changes a pixel
 * (every one of them is exercised by a bit-exactness test).  The one concession to shell-driven sweeps: when the variable
 * PROSPER_PT_DEBUG is "1" at prosper_pt_create - and only then, and only there - PROSPER_PT_DEBUG_OPTIONS
 * ("name=value,name=value", the field names below) is parsed into the new context's options.
 * Zero (or -1 where zero is a value) always means "the library's default". */
typedef struct prosper_pt_debug_options
{
    uint32_t struct_size; /* sizeof(prosper_pt_debug_options) */
    /* ---- scene upload ---- */
    int32_t batchedTextures;   /* -1: by texel footprint; 0 / 1: a hit's texel loads one by one / issued together */
    int32_t widePacks;         /* -1: by texel footprint; 1 / 0: 16-byte / compact 8-byte material texture packs */
    int32_t alphaCellShift;    /* -1: default; s: alpha-bound cells of 2^s texels a side */
    uint32_t noTexturePacks;   /* every texture sampled by itself */
    uint32_t noAlphaBounds;    /* every any-hit candidate runs the exact code */
    uint32_t noUploadRefit;    /* keep the host emitter's node bytes (the test that compares them with the device encoder's) */
    uint32_t flatBvh;          /* one SAH tree over everything instead of per-instance subtrees */
    /* ---- hierarchy builder (upload and rebuild) ---- */
    float sahTraversalCost;    /* 0: 1.0 */
    float boxPad;              /* 0: 1.6e-5 (also the minimum) */
    uint32_t leafSize;         /* 0: 4 */
    uint32_t buildThreads;     /* 0: the host's threads */
    uint32_t topEntries;       /* 0: one per four triangles */
    int32_t nodeOrder;         /* -1: 2 (first 4096 nodes breadth-first, then depth-first subtrees); 0 depth-first; 1 breadth-first */
    int32_t childOrder;        /* -1 / 1: smallest box first; 0: build order */
    uint32_t buildTiming;      /* stage times of the hierarchy assembly and of a geometry build to stderr */
    /* ---- render ---- */
    uint32_t segments;         /* target segment count of the wavefront workspace */
    uint32_t segmentLength;    /* segment length in slots (a multiple of 64) */
    uint32_t chains;           /* launch chains of an in-order render (default 2) */
    uint32_t ldsStackEntries;  /* 16 / 24 / 32: LDS traversal-stack entries (deeper entries spill to global memory) */
    uint32_t noLdsScene;       /* keep a small scene in global memory */
    uint32_t noLdsTables;      /* keep the shading tables in global memory */
    uint32_t traceDeadPaths;   /* keep tracing zero-throughput paths, as the GLSL does (audit of the contract's rule) */
    int32_t bandedBatches;     /* 1: every XCD's segments take the camera-ray batches of ONE band of the image instead of batches
                                * strided over all of it (measured slower: profiles/r04_banded_batches.txt); -1 / 0: strided */
    /* ---- updates ---- */
    float rebuildCostRatio;    /* 0: 1.3 - growth of the tree's surface-area measure at which an update also rebuilds */
    uint32_t alwaysRebuild;    /* rebuild with every update, synchronously */
    uint32_t failNextUpdate;   /* the next host-side rebuild fails (the recovery test); cleared by that failure */
    /* ---- measured-slower experiments: only in a library built with -DPPT_EXPERIMENTS (prosper_pt_has_experiments);
     *      setting any of them in the default build fails with PROSPER_PT_ERR_UNSUPPORTED ---- */
    uint32_t poolVariant;      /* wf_trace out of an LDS ray pool (1, 2, 3) */
    uint32_t rawRecords;       /* 64-byte raw shading records decoded per hit (upload) */
    uint32_t tileOrder;        /* camera-ray batches take the tiles by the cost of a probe ray */
    uint32_t hipGraph;         /* a pipelined render's chain of launches through a HIP graph */
    uint32_t pipelinedChains;  /* 2: a frame in flight runs as two chains */
    uint32_t mergeLimit;       /* a workgroup's four segments traced by one wave below this many rays */
} prosper_pt_debug_options;

const char *prosper_pt_last_error(void);
uint32_t prosper_pt_abi_version(void);
/* 1 when the library was built with -DPPT_EXPERIMENTS (the measured-slower kernel variants are compiled in), else 0.
 * In the default build PROSPER_PT_CREATE_PERSISTENT is refused by prosper_pt_create as well. */
uint32_t prosper_pt_has_experiments(void);
void prosper_pt_debug_options_default(prosper_pt_debug_options *out);
int prosper_pt_set_debug_options(prosper_pt_ctx *ctx, const prosper_pt_debug_options *options);
int prosper_pt_get_debug_options(prosper_pt_ctx *ctx, prosper_pt_debug_options *out);

int prosper_pt_create(const prosper_pt_device_desc *desc, prosper_pt_ctx **out_ctx);
void prosper_pt_destroy(prosper_pt_ctx *ctx);

/* Copies the whole scene to HBM, flattens instances x triangles into world space from the
 * fp16 positions (the BVH must see the decoded halfs: World.cpp:635-644) and builds the BVH. */
int prosper_pt_upload_scene(prosper_pt_ctx *ctx, const prosper_pt_scene_view *scene);
/* The three light buffers only (prosper rewrites them every frame: World.cpp:531-535).  An unchanged set costs a memcmp.
 * A changed one is staged by the call and copied by the next render at the head of its own chain of launches into the
 * next of three device versions (ABI 3): the frames in flight keep theirs, nothing synchronises the device. */
int prosper_pt_update_lights(
    prosper_pt_ctx *ctx, const prosper_DirectionalLightParameters *directionalLight,
    const prosper_PointLightsBuffer *pointLights, const prosper_SpotLightsBuffer *spotLights);
/* New instance transforms for the uploaded scene (the whole ModelInstanceTransforms table, `count` = the scene's
 * modelInstanceCount; World::updateScene rewrites it every frame, World.cpp:359-466).  prosper then rebuilds its TLAS on
 * the GPU (World.cpp:749-802, 878-928).  Here (ABI 3) the update is a REFIT on the GPU: the transform table, the
 * world-space triangles and new boxes for the unchanged tree (one small kernel per tree level + one over all nodes) - no
 * host-side build, no device-wide synchronisation.  prosper_pt_update_transforms only STAGES the table (it returns in
 * microseconds; an unchanged table is a no-op); the refit runs at the head of the next render's own chain of launches,
 * into the next of three versions of the transform / triangle / node arrays - like the per-frame TLAS of a Vulkan frame
 * loop - so the frames in flight (PROSPER_PT_RENDER_PIPELINED) go on reading theirs and nothing waits for them.
 * prosper_pt_update_transforms_async with PROSPER_PT_UPDATE_NOW enqueues the refit on `stream` right away instead (any
 * stream, the null stream included); without the flag it stages like prosper_pt_update_transforms and ignores `stream`.  Same
 * pixels as a fresh prosper_pt_upload_scene of the moved scene (hits do not depend on the hierarchy).  A refit cannot keep
 * the tree good when instances travel far: each one leaves the tree's surface-area measure behind, and when that has
 * grown by 30 % (debug option rebuildCostRatio) over its value at the last build, the next update has the moved
 * instances split again - by the context's worker thread, into a new generation of the geometry that the first render
 * after it is done switches to, like streamed-in meshes (prosper_pt_update_meshes); the frame loop goes on refitting and
 * rendering meanwhile.  prosper_pt_finish_mesh_updates waits for it too; prosper_pt_rebuild_hierarchy does the same work
 * at once, synchronously. */
int prosper_pt_update_transforms(prosper_pt_ctx *ctx, const prosper_ModelInstanceTransforms *transforms, uint32_t count);
enum
{
    PROSPER_PT_UPDATE_NOW = 1u << 0, /* enqueue the refit on `stream` inside the call instead of leaving it to the next render */
};
int prosper_pt_update_transforms_async(
    prosper_pt_ctx *ctx, const prosper_ModelInstanceTransforms *transforms, uint32_t count, uint32_t flags, void *stream);
/* ---- incremental adoption: streamed-in textures and materials ----
 * prosper loads a scene in the background and adopts what has arrived a few items per frame: new images get their slot in
 * materialTextures[] (src/scene/WorldData.cpp:2182-2206), a material switches from its placeholder - the default material
 * with the real alpha mode (WorldData.cpp:817-826) - to the real one once its three images are there (:2208-2239), and the
 * next frame's material buffer is rewritten when that happened (:568-586; App.cpp:526-529, 601).  No acceleration structure
 * is touched.  Upload the scene with placeholder textures (1 x 1 texels will do) and placeholder materials, then:
 *   prosper_pt_update_textures   replaces the texels of materialTextures[first .. first + count) (any extent / format);
 *                                the caller's memory is borrowed for the call only.  Materials that sample a replaced
 *                                texture get their texture pack and alpha bounds rebuilt.
 *   prosper_pt_update_materials  replaces MaterialData[first .. first + count); an unchanged entry costs a memcmp (prosper
 *                                rewrites the whole table).  A material's alpha mode must be the one it was uploaded with:
 *                                it decides the opaque flag of the geometry (World.cpp:646-651).
 * Both only stage: the copies, re-tiling / BC7 decode, packs and alpha bounds run on a stream the context owns, and the next
 * render switches to a new version of the material / texture tables at the head of its own chain of launches, like
 * prosper_pt_update_transforms - the frames in flight (PROSPER_PT_RENDER_PIPELINED) keep reading theirs.  (One exception: a
 * changed MASK / BLEND material rewrites the any-hit records in place, behind the frames in flight.)  Every frame shows the
 * scene a fresh prosper_pt_upload_scene of that state would show, bit for bit. */
int prosper_pt_update_textures(prosper_pt_ctx *ctx, const prosper_pt_texture_desc *textures, uint32_t first, uint32_t count);
int prosper_pt_update_materials(prosper_pt_ctx *ctx, const prosper_MaterialData *materials, uint32_t first, uint32_t count);

/* ---- incremental adoption: streamed-in meshes ----
 * prosper's mesh worker fills the geometry buffers in the background; WorldData::pollMeshWorker adopts at most ten finished
 * meshes per frame - their GeometryMetadata and MeshInfo slots, their byte range of a geometry buffer, a new 64 MB buffer
 * now and then (src/scene/WorldData.cpp:2003-2110; at most sMaxGeometryBuffersCount = 100 of them, :31).  Until then a mesh's
 * metadata holds bufferIndex = 0xFFFFFFFF; World::buildNextBlas builds a model's BLAS only once ALL its sub-meshes are there
 * (World.cpp:598-606) and a TLAS instance without a BLAS is inactive (accelerationStructureReference 0, World.cpp:909-915):
 * rays pass through model instances that are still loading.
 * Here: upload the scene with every mesh slot, draw instance and transform it will have; a mesh that has not arrived has
 * geometryMetadatas[i].bufferIndex == PROSPER_PT_ABSENT (the rest of its metadata and its MeshInfo are ignored), and a model
 * instance - a run of draw instances with one modelInstanceIndex - contributes triangles only once every mesh it draws is
 * there.  prosper_pt_update_meshes hands over the meshes that arrived: metadata and MeshInfo as pollMeshWorker stores them,
 * and the bytes the worker wrote (`bytes`, borrowed for the call) with their place in geometryBuffers[metadata.bufferIndex].
 * A buffer index the scene has not seen yet creates that buffer, `bufferByteSize` bytes large (ignored otherwise).
 * The call keeps a copy of the bytes, notes the tables and returns (0.1 ms): a worker thread of the context writes the
 * bytes to the device, lays the triangles out again, builds the subtrees of the model instances that became complete (the
 * others are kept), re-assembles the hierarchy and writes the per-triangle records - all into arrays of its own, on a
 * stream of its own (the calling thread touches no stream) - while the frame loop goes on
 * rendering the geometry it has, frames in flight included.  The first render (or prosper_pt_update_meshes, or
 * prosper_pt_finish_mesh_updates) after the worker is done switches to the new geometry; instances that moved or materials
 * that changed meanwhile are brought up to date by that switch.  This is prosper's own timing: a BLAS is built on the
 * GPU while frames are drawn and its instances appear when it exists.  Meshes that arrive while a build is under way are
 * taken up by the next one, started at the switch.  Textures, material tables, lights and sky are not touched.  Once a state
 * is switched in, a render shows what a fresh prosper_pt_upload_scene of that state would show, bit for bit.
 * prosper_pt_finish_mesh_updates waits until everything handed over so far is in the scene (screenshots, tests, the end of
 * loading).  A mesh can be handed over once; its material's alpha mode is the one the table holds at the time of the call.
 * A build that fails (out of memory, a hierarchy too deep for the traversal) is reported by the call that would have
 * switched to it; the scene stays as it was and the meshes wait for the next build. */
#define PROSPER_PT_MAX_GEOMETRY_BUFFERS 100u
typedef struct prosper_pt_mesh_update
{
    uint32_t meshIndex;
    uint32_t reserved;
    prosper_GeometryMetadata metadata; /* bufferIndex < PROSPER_PT_MAX_GEOMETRY_BUFFERS */
    prosper_pt_mesh_info info;
    const void *bytes;       /* UploadedGeometryData: the mesh's part of the geometry buffer ... */
    uint64_t byteOffset;     /* ... its place in that buffer (a multiple of 4) ... */
    uint64_t byteCount;      /* ... and size; every stream `metadata` names must lie inside */
    uint64_t bufferByteSize; /* size of geometryBuffers[metadata.bufferIndex] if the scene does not have that buffer yet */
} prosper_pt_mesh_update;
int prosper_pt_update_meshes(prosper_pt_ctx *ctx, const prosper_pt_mesh_update *meshes, uint32_t count);
int prosper_pt_finish_mesh_updates(prosper_pt_ctx *ctx);

/* Re-splits the instances that moved since the last build and re-assembles the tree on the host (one subtree per model
 * instance under a re-braided top level); synchronises the device.  prosper_pt_scene_stats.bvhBuildSeconds reports it. */
int prosper_pt_rebuild_hierarchy(prosper_pt_ctx *ctx);
typedef struct prosper_pt_hierarchy_state
{
    uint32_t refits;    /* since the scene was uploaded */
    uint32_t rebuilds;
    float costRatio;    /* sum of the inner boxes' half-areas after the last refit / right after the last build */
    float builtCost;
    uint32_t nodeCount;
    uint32_t levels;    /* kernel launches of a refit = levels + 1 */
    uint32_t meshUpdates;        /* prosper_pt_update_meshes calls that handed meshes over */
    uint32_t geometryInstalls;   /* background geometry builds whose result has become the scene */
    uint32_t geometryBuildRunning; /* 1: a build is under way (or meshes wait for one) */
    uint32_t reserved;
} prosper_pt_hierarchy_state;
/* Waits for the last refit's measure if it is still on its way. */
int prosper_pt_get_hierarchy_state(prosper_pt_ctx *ctx, prosper_pt_hierarchy_state *out);
/* Test hook: the node array as the device holds it (prosper_pt_scene_stats.nodeCount x nodeBytes). */
int prosper_pt_debug_read_nodes(prosper_pt_ctx *ctx, void *out, size_t byte_size);
int prosper_pt_get_scene_stats(prosper_pt_ctx *ctx, prosper_pt_scene_stats *out);

/* Optional: render into caller-owned device memory (localWidth*height RGBA32F texels, 16-byte
 * aligned) instead of the context's own buffer — lets the host framework (e.g. a torch tensor
 * that RCCL gathers) own the HDR tile.  NULL restores the internal buffer.  History is read from
 * and written to the same buffer, like the reference's aliased previous/illumination image. */
int prosper_pt_set_output_buffer(prosper_pt_ctx *ctx, void *device_rgba32f, size_t byte_size);

/* One accumulated frame = one path per pixel (the reference's traceRaysKHR(W,H,1)).
 * `stream` is a hipStream_t (NULL = the null stream); the call only enqueues work.  Stream semantics are
 * those of a kernel launch on `stream`: the render starts after the work already queued there and work
 * queued afterwards sees its result.  (Internally large batches run as two chains of launches on two
 * streams the context owns, forked from and joined back into `stream` with events.) */
int prosper_pt_render(
    prosper_pt_ctx *ctx, const prosper_ReferencePC *pc, const prosper_CameraUniforms *camera,
    uint32_t width, uint32_t height, const prosper_pt_tile_desc *tile, uint32_t render_flags,
    void *stream);

/* `frame_count` consecutive accumulated frames in one launch: exactly the pixels that
 * frame_count calls of prosper_pt_render with frameIndex = (pc->frameIndex + f) % 4096 and
 * PROSPER_PC_FLAG_SKIP_HISTORY honoured on the first of them only would produce (the running mean
 * of main.rgen:289-297 is kept in registers between frames instead of going through HBM). */
int prosper_pt_render_frames(
    prosper_pt_ctx *ctx, const prosper_ReferencePC *pc, const prosper_CameraUniforms *camera,
    uint32_t width, uint32_t height, const prosper_pt_tile_desc *tile, uint32_t frame_count,
    uint32_t render_flags, void *stream);

/* Width in texels of this context's HDR rows for the last render (== width when untiled). */
int prosper_pt_get_local_extent(prosper_pt_ctx *ctx, uint32_t *local_width, uint32_t *height);
/* Device address of the current HDR buffer (for RCCL gathers); valid until the next render with a
 * different extent, set_output_buffer or destroy. */
int prosper_pt_get_hdr_device_ptr(prosper_pt_ctx *ctx, void **out_ptr, size_t *out_bytes);
/* Synchronises `stream` and copies the HDR tile (localWidth*height RGBA32F) to host memory. */
int prosper_pt_read_hdr(prosper_pt_ctx *ctx, float *rgba32f, size_t byte_size, void *stream);
/* RGBA32F -> RGBA16F (round-to-nearest-even), the image later passes consume. */
int prosper_pt_blit_rgba16f(prosper_pt_ctx *ctx, uint16_t *host_rgba16f, size_t byte_size, void *stream);

/* The step after the path: prosper tone-maps the RGBA16F illumination into an RGBA8 UNORM
 * image with res/shader/tone_map.comp (src/render/ToneMap.cpp:62-128): exposure, HSV contrast, the
 * Tony McMapface 3-D LUT (res/texture/tony_mc_mapface.dds, 48^3 R9G9B9E5, linear / clamp sampler), 1/2.2
 * gamma.  prosper_pt_set_tone_map_lut copies dim^3 R9G9B9E5 texels (x fastest) to the device;
 * prosper_pt_tone_map runs blit + tone map in one kernel over the current HDR tile and writes
 * localWidth*height RGBA8 texels to `device_rgba8` (device memory, may be NULL) and/or `host_rgba8`
 * (host memory, may be NULL; synchronises `stream`). */
int prosper_pt_set_tone_map_lut(prosper_pt_ctx *ctx, const uint32_t *lut_r9g9b9e5, uint32_t dim);
int prosper_pt_tone_map(
    prosper_pt_ctx *ctx, float exposure, float contrast, void *device_rgba8, uint8_t *host_rgba8, size_t byte_size,
    void *stream);

/* A second client of the traversal: prosper's ReSTIR-DI trace pass,
 * res/shader/rt/direct_illumination/main.rgen:44-165 dispatched by src/render/rtdi/Trace.cpp:297.  Per pixel:
 * the surface from the G-buffer (world position from the non-linear depth through camera->clipToWorld,
 * signed-octahedral normal), the light its reservoir holds, one shadow ray with the scene's any-hit rules, the
 * BRDF, and the running mean into the context's HDR image (whole image, no stripes).
 * TracePC = res/shader/shared/shader_structs/push_constants/restir_di/trace.h. */
typedef struct prosper_pt_restir_trace_pc
{
    uint32_t drawType;   /* PROSPER_DRAW_TYPE_*: Default traces; Position shows positions, others the albedo */
    uint32_t frameIndex;
    uint32_t flags;      /* bit 0 skipHistory, bit 1 accumulate */
} prosper_pt_restir_trace_pc;
typedef struct prosper_pt_restir_inputs
{
    const void *albedoRoughness;  /* width*height float4: albedo.rgb, roughness        (gbuffer.frag) */
    const void *normalMetallic;   /* width*height float4: octNormal.xy, metallic, octNormal.z */
    const float *nonLinearDepth;  /* width*height */
    const void *reservoirs;       /* width*height float2: bits of the int light index (< 0 = none), weight */
    uint32_t onDevice;            /* 1: the pointers are device memory; 0: host memory, copied by the call */
    uint32_t reserved;
} prosper_pt_restir_inputs;
int prosper_pt_restir_di_trace(
    prosper_pt_ctx *ctx, const prosper_pt_restir_trace_pc *pc, const prosper_CameraUniforms *camera, uint32_t width,
    uint32_t height, const prosper_pt_restir_inputs *inputs, void *stream);

/* ---- multi-GPU: image stripes per rank + ONE gather of the per-rank HDR tiles over RCCL + de-interleave ----
 * (north star: "the image is tiled across the 8 GPUs of one node with an RCCL gather over xGMI of
 * per-tile HDR buffers".)  The reference renders the whole image on one GPU and asserts renderArea.offset == 0
 * (src/render/RtReference.cpp:327): these entry points have no counterpart there.  One process (or thread) per GPU,
 * one context each; rank r renders with prosper_pt_tile_desc{stripeWidth, r, ranks}; pixels are independent
 * (seed = absolute pixel + frame index, main.rgen:229), so nothing is exchanged while rendering.
 *
 * RCCL is loaded on first use (dlopen librccl.so.1); without it these calls fail with PROSPER_PT_ERR_UNSUPPORTED.
 *   prosper_pt_comm_get_unique_id  ncclGetUniqueId: call on ONE rank, hand the 128 bytes to the others by any means
 *   prosper_pt_comm_init           ncclCommInitRank on the context's device (collective over the ranks)
 *   prosper_pt_comm_adopt          use the caller's ncclComm_t instead (borrowed, not destroyed)
 *   prosper_pt_gather_tiles        after a render with a tile: ncclGather of every rank's localWidth*height RGBA32F
 *                                  tile to `root` (grouped ncclSend/ncclRecv with per-rank counts when the stripes do
 *                                  not divide evenly), then on the root a HIP kernel writes the width*height image into
 *                                  `device_full_rgba32f` (ignored on the other ranks; NULL on the root: a buffer the context owns;
 *                                  with one rank: a copy).
 *                                  Enqueue-only.  Default: gather and de-interleave run on a stream the context owns,
 *                                  after the work queued on `stream` so far, and overlap what the caller enqueues next -
 *                                  the next render's accumulate kernel (the one writer of the tile) waits for them by
 *                                  itself; readers of `device_full_rgba32f` call prosper_pt_gather_wait first.
 *                                  PROSPER_PT_GATHER_IN_STREAM runs everything on `stream` instead.
 *   prosper_pt_gather_wait         makes `stream` wait for the last gather (+ de-interleave)
 *   prosper_pt_comm_query          the communicator's own rank count / rank / device and the last gather's device time
 *   prosper_pt_deinterleave_tiles  the root's kernel alone: `device_tiles` = the ranks' tiles back to back in rank order
 */
#define PROSPER_PT_COMM_ID_BYTES 128
enum
{
    PROSPER_PT_GATHER_IN_STREAM = 1u << 0,
};
int prosper_pt_comm_get_unique_id(uint8_t id[PROSPER_PT_COMM_ID_BYTES]);
int prosper_pt_comm_init(prosper_pt_ctx *ctx, const uint8_t id[PROSPER_PT_COMM_ID_BYTES], uint32_t rank, uint32_t ranks);
int prosper_pt_comm_adopt(prosper_pt_ctx *ctx, void *nccl_comm, uint32_t rank, uint32_t ranks);
int prosper_pt_comm_destroy(prosper_pt_ctx *ctx);
int prosper_pt_gather_tiles(
    prosper_pt_ctx *ctx, uint32_t root, void *device_full_rgba32f, size_t byte_size, uint32_t flags, void *stream);
int prosper_pt_gather_wait(prosper_pt_ctx *ctx, void *stream);
/* What the communicator itself reports (ncclCommCount / ncclCommUserRank / ncclCommCuDevice - not what the caller passed
 * to prosper_pt_comm_init), the gathers enqueued so far, and the device time of the last one: its collective + on the
 * root the de-interleave kernel, between two events on the stream it ran on (waits for it; 0 before the first).
 * Without a communicator: ranks 1, rank 0, the context's device. */
typedef struct prosper_pt_comm_info
{
    uint32_t ranks;
    uint32_t rank;
    int32_t device;
    uint32_t gathers;
    float lastGatherMs;
    uint32_t reserved;
} prosper_pt_comm_info;
int prosper_pt_comm_query(prosper_pt_ctx *ctx, prosper_pt_comm_info *out);
/* The root's gathered image: where the last prosper_pt_gather_tiles put it (the caller's buffer, or the context's own
 * when `device_full_rgba32f` was NULL), and a synchronising copy of it to host memory (width*height RGBA32F). */
int prosper_pt_get_gathered_device_ptr(prosper_pt_ctx *ctx, void **out_ptr, uint32_t *width, uint32_t *height);
int prosper_pt_read_gathered(prosper_pt_ctx *ctx, float *rgba32f, size_t byte_size, void *stream);
int prosper_pt_deinterleave_tiles(
    prosper_pt_ctx *ctx, const void *device_tiles, uint32_t ranks, uint32_t stripe_width, uint32_t width, uint32_t height,
    void *device_full_rgba32f, size_t byte_size, void *stream);

int prosper_pt_get_counters(prosper_pt_ctx *ctx, prosper_pt_counters *out, void *stream);
/* The same counters for one kernel stage (index as in prosper_pt_kernel_name): lets the roofline
 * of a single kernel be priced from the work that kernel did. */
int prosper_pt_get_stage_counters(prosper_pt_ctx *ctx, uint32_t stage, prosper_pt_counters *out, void *stream);
int prosper_pt_reset_counters(prosper_pt_ctx *ctx, void *stream);

/* Device time (ms) of the kernels launched by the last `prosper_pt_render(_frames)`, measured with
 * hipEvents around every launch on the stream it was launched on; blocks until they finish.
 * total_ms is the time on the caller's stream (wall time of the render on the device).  kernel_ms[i] is
 * the SUM over the kernel_launches[i] launches of stage i, named by prosper_pt_kernel_name(i); with the
 * two-chain default a launch that shared the GPU with the other chain's launch counts in full, so the
 * kernel_ms can add up to more than total_ms. */
#define PROSPER_PT_MAX_KERNELS 8
int prosper_pt_get_last_render_ms(prosper_pt_ctx *ctx, float *total_ms, float kernel_ms[PROSPER_PT_MAX_KERNELS]);
int prosper_pt_get_last_render_timing(
    prosper_pt_ctx *ctx, float *total_ms, float kernel_ms[PROSPER_PT_MAX_KERNELS],
    uint32_t kernel_launches[PROSPER_PT_MAX_KERNELS]);
const char *prosper_pt_kernel_name(uint32_t index);
/* Enables per-kernel hipEvent timing for subsequent renders (off by default: events add launches).  The readout
 * functions above report the last render that ran with timing on, also after timing has been switched off again
 * and further (untimed) renders have followed - up to two of them with PROSPER_PT_RENDER_PIPELINED. */
int prosper_pt_set_kernel_timing(prosper_pt_ctx *ctx, int enabled);

enum
{
    PROSPER_PT_FN_SINCOS = 0,         /* in: x                                out: sin, cos */
    PROSPER_PT_FN_POW = 1,            /* in: x, y                             out: pow */
    PROSPER_PT_FN_SRGB_TO_LINEAR = 2, /* in: x                                out: y */
    PROSPER_PT_FN_NORMALIZE = 3,      /* in: v3                               out: v3 */
    PROSPER_PT_FN_UNPACK_SNORM = 4,   /* in: bits (u32 in a float)            out: n3, sign */
    PROSPER_PT_FN_ONB = 5,            /* in: n3                               out: rows b1, b2, n */
    PROSPER_PT_FN_COSINE_SAMPLE = 6,  /* in: n3, u2                           out: v3 */
    PROSPER_PT_FN_VNDF_SAMPLE = 7,    /* in: Ve3, alpha, u2                   out: v3 */
    PROSPER_PT_FN_VNDF_PDF = 8,       /* in: Ve3, Le3, alpha                  out: pdf */
    PROSPER_PT_FN_EVAL_BRDF = 9,      /* in: l3, n3, v3, albedo3, rough, metal out: v3 */
    PROSPER_PT_FN_OFFSET_RAY = 10,    /* in: p3, n3                           out: v3 */
    PROSPER_PT_FN_POINT_LIGHT = 11,   /* in: pos3, radiance3, radius, surf3   out: l3, d, irr3 */
    PROSPER_PT_FN_SPOT_LIGHT = 12,    /* in: pos3, off, rad3, scale, dir3, surf3 out: l3, d, irr3 */
    PROSPER_PT_FN_TRIANGLE = 13,      /* in: o3, d3, v0, v1, v2, tmin, tmax   out: hit, t, bu, bv */
    PROSPER_PT_FN_HALF = 14,          /* in: f                                out: unpack(pack(f)), bits */
    PROSPER_PT_FN_RNG = 15,           /* in: px, py, frame (u32 bits)         out: rnd01, rnd2d01, seed */
    PROSPER_PT_FN_BC7_BLOCK = 16,     /* in: 4 words of a block (u32 bits)    out: 16 RGBA8 texels (u32 bits) */
    PROSPER_PT_FN_COUNT = 17,
};

/* Test hook for the alpha bounds (DESIGN.md "alpha bounds"): evaluates the device's sRGBtoLinear on EVERY float whose bit
 * pattern lies in [first_bits, last_bits] (non-negative floats, increasing) and reports how far it is from monotone:
 * *max_defect = max over x of (max of L over the inputs shortly before x) - L(x), 0 if monotone; *decreases = adjacent
 * input pairs whose outputs decrease.  The bounds are valid while max_defect < 4e-6 (kAlphaCurveSlack). */
int prosper_pt_debug_srgb_monotonicity(
    prosper_pt_ctx *ctx, uint32_t first_bits, uint32_t last_bits, float *max_defect, uint64_t *decreases);

/* Device self-test: evaluates device function `fn` (PROSPER_PT_FN_*) element-wise over `n`
 * records of `in_stride` floats and writes `out_stride` floats per record.  Test-only entry that
 * lets tests/ compare every device function with the oracle's restatement bit for bit. */
int prosper_pt_eval_device_fn(
    prosper_pt_ctx *ctx, uint32_t fn, const float *in, uint32_t in_stride, float *out,
    uint32_t out_stride, uint32_t n);

#ifdef __cplusplus
}
#endif

#endif /* PROSPER_PT_H */
