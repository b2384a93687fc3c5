// pt_context.hpp — the context behind the C-ABI handle (private to the library's translation units).
#pragma once

#include <hip/hip_runtime.h>

#include <atomic>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/prosper_pt/prosper_pt.h"
#include "bvh_build.hpp"
#include "pt_kernels.hpp"
#include "pt_scene.hpp"

namespace ppt
{

struct DeviceAllocation
{
    void *ptr = nullptr;
    size_t bytes = 0;
};

// records `msg` as the calling thread's last error (prosper_pt_last_error) and returns `code`
int fail(int code, const std::string &msg);

// Host + device state of the acceleration structure that outlives prosper_pt_upload_scene, so that moved instances can
// be re-fitted without rebuilding the scene (prosper_pt_update_transforms): the world triangles in (drawInstance,
// primitive) order on both sides, the per-instance subtrees (bvh_build.hpp InstancedBvh), the instance table.
// The three light buffers of the scene (World.cpp:531-535 rewrites them every frame), versioned like the instance
// transforms: an update is staged by the call (an unchanged set is a no-op) and copied by the next render's own chain
// into the next of three device copies, so the frames in flight keep theirs and nothing synchronises the device.
// Pinned staging buffers per kind of update (transforms, lights, material tables): an update's copy is enqueued at the head
// of the next render's chain, i.e. BEHIND the frames in flight; with two buffers the update after next may wait on the host
// for that copy - with one buffer more than frames in flight it never does.
constexpr uint32_t kStagingBuffers = 4;

struct LightBlock
{
    alignas(16) prosper_DirectionalLightParameters directional;
    alignas(16) prosper_PointLightsBuffer points; // (16-byte aligned: the kernels read the lights as vec4s)
    alignas(16) prosper_SpotLightsBuffer spots;
};
struct LightState
{
    static constexpr uint32_t kVersions = 3;
    LightBlock *dBlocks[kVersions] = {}; // [0] is the upload's own allocation
    hipEvent_t versionFree[kVersions] = {};
    bool versionUsed[kVersions] = {};
    hipStream_t versionStream[kVersions] = {}; // the stream versionFree was last recorded on
    uint32_t cur = 0;
    LightBlock *mirror = nullptr;         // host copy of what the device holds (or will hold once `pending` is flushed)
    LightBlock *staging[kStagingBuffers] = {}; // pinned
    hipEvent_t stagingDone[kStagingBuffers] = {};
    bool stagingUsed[kStagingBuffers] = {};
    uint32_t stagingNext = 0, pendingStaging = 0;
    bool pending = false;
    hipEvent_t ready = nullptr; // behind the last flush: every later render's chains wait for it
    bool readyRecorded = false;
    uint32_t updates = 0;
    ~LightState()
    {
        delete mirror;
        for (uint32_t i = 0; i < kStagingBuffers; ++i)
        {
            if (staging[i]) (void)hipHostFree(staging[i]);
            if (stagingDone[i]) (void)hipEventDestroy(stagingDone[i]);
        }
        for (hipEvent_t e : versionFree)
            if (e) (void)hipEventDestroy(e);
        if (ready) (void)hipEventDestroy(ready);
    }
};

struct AccelState
{
    WorldTriangle *dFlat = nullptr; // (drawInstance, primitive) order: what flatten_triangles writes
    WorldTriangle *dTris = nullptr; // leaf order: what the traversal reads
    uint32_t *dOffsets = nullptr, *dFlags = nullptr, *dPerm = nullptr;
    BvhNode *dNodes = nullptr;
    size_t nodeCapacityBytes = 0;
    std::vector<uint32_t> triOffsets;
    std::vector<WorldTriangle> flat; // host copy of dFlat as of the last BUILD (flatStale: instances moved since)
    std::vector<InstancedBvh::Range> ranges;      // one per run of draw instances of a model instance
    std::vector<uint32_t> rangeModelInstance;
    std::vector<prosper_ModelInstanceTransforms> transforms;
    InstancedBvh bvh;
    bool instanced = false;
    // a prosper_pt_update_transforms that failed half way leaves transforms, world triangles and nodes out of step: renders
    // are refused and the next update redoes every instance, whatever the transforms it is given
    bool stale = false;
    uint64_t total = 0;
    uint32_t drawInstanceCount = 0;

    // ---- refit of the tree after moved instances (pt_kernels.hip refit_bounds / encode_nodes) ----
    float4 *dNodeBounds = nullptr;     // [2 * nodes]: exact bounds below every node
    uint32_t *dRefitOrder = nullptr;   // node indices by height (leaves' parents first)
    uint32_t *dLeafPosition = nullptr; // [total]: (drawInstance, primitive)-order triangle -> its place in dTris
    size_t refitCapacityNodes = 0;
    std::vector<uint32_t> levelOffsets; // [levels + 1] into dRefitOrder
    uint32_t nodeCount = 0;
    // Surface-area measure of the refitted tree (encode_nodes_kernel), one slot per scene version: a refit writes the
    // slot of the version it produces, so a measure can be read as soon as ITS refit has finished - in a pipelined loop
    // that updates every frame the newest refit has only just been enqueued, but the one of two updates ago is done.
    float *dCost = nullptr;             // [kCostSlots] device
    float *hCost = nullptr;             // [kCostSlots] pinned; slot i valid once costEvent[i] has passed
    static constexpr uint32_t kCostSlots = 3;
    hipEvent_t costEvent[kCostSlots] = {};
    bool costPending[kCostSlots] = {};
    uint64_t costSequence[kCostSlots] = {}; // which refit (a running number) the slot's measure belongs to
    uint64_t refitSequence = 0, costRead = 0; // refits enqueued so far / the newest one whose measure has been taken
    float builtCost = 0.0f;             // the same measure right after the last build
    float lastCostRatio = 1.0f;
    std::vector<uint8_t> movedSinceBuild; // per range: its subtree is out of date in `bvh` and `flat`
    bool flatStale = false;
    // the update's transforms go through pinned staging (a pageable source would make the async copy synchronous)
    prosper_ModelInstanceTransforms *staging[kStagingBuffers] = {};
    hipEvent_t stagingDone[kStagingBuffers] = {};
    bool stagingUsed[kStagingBuffers] = {};
    uint32_t stagingNext = 0;
    // recorded on the updating stream behind the refit: every later render's path stages wait for it
    hipEvent_t sceneEvent = nullptr;
    bool sceneEventRecorded = false;
    uint32_t refits = 0, rebuilds = 0;

    // ---- scene versions: what a refit rewrites - transform table, leaf-order triangles, nodes - exists up to three
    // times, like the per-frame TLAS / instance buffers of a Vulkan frame loop: an update writes the NEXT version while
    // the frames in flight go on reading theirs.  dNodes / dTris / ctx->dTransforms alias version `cur`.  The other
    // versions are allocated by the first update that needs them. ----
    static constexpr uint32_t kVersions = 3;
    WorldTriangle *dTrisV[kVersions] = {};
    BvhNode *dNodesV[kVersions] = {};
    prosper_ModelInstanceTransforms *dTransformsV[kVersions] = {};
    bool nodesCurrent[kVersions] = {}; // the version's node array holds the tree of the last build (child references)
    hipEvent_t versionFree[kVersions] = {}; // behind the last render that read the version
    bool versionUsed[kVersions] = {};
    hipStream_t versionStream[kVersions] = {}; // the stream versionFree was last recorded on
    uint32_t cur = 0;
    // an update waits here until the next consumer of the scene - normally the next render, which runs it at the head of
    // its own chain of launches, beside the frames in flight
    bool pending = false, pendingGeometry = false;
    uint32_t pendingStaging = 0, pendingCount = 0;

    ~AccelState()
    {
        for (uint32_t i = 0; i < kStagingBuffers; ++i)
        {
            if (staging[i]) (void)hipHostFree(staging[i]);
            if (stagingDone[i]) (void)hipEventDestroy(stagingDone[i]);
        }
        if (hCost) (void)hipHostFree(hCost);
        for (hipEvent_t e : costEvent)
            if (e) (void)hipEventDestroy(e);
        if (sceneEvent) (void)hipEventDestroy(sceneEvent);
        for (hipEvent_t e : versionFree)
            if (e) (void)hipEventDestroy(e);
    }
};

// The tables a streamed-in texture or material changes (prosper adopts loaded textures and materials a few per frame:
// src/scene/WorldData.cpp:588-647, 2182-2239; the material buffer is re-uploaded when materialsGeneration moves, :568-586):
// MaterialData[], MaterialPack[], AlphaMaterial[], DeviceTexture[] as ONE device block per version.  Version 0 is what
// prosper_pt_upload_scene made (four allocations of their own, never rewritten); prosper_pt_update_textures / _materials
// change the host mirrors, build the new texel arrays / packs / alpha bounds on `uploadStream`, and the next render copies
// the tables into the next of three rotating blocks at the head of its own chain of launches - the frames in flight keep
// reading theirs (pt_materials.cpp).  Replaced texel arrays / packs / bounds are retired, not freed: a frame in flight may
// still read them (a streamed scene replaces each placeholder once; 256 MB of retired memory buy one synchronisation).
struct MaterialState
{
    static constexpr uint32_t kVersions = 4; // 0: the upload's tables; 1..3: rotating blocks
    std::vector<prosper_MaterialData> materials;
    std::vector<MaterialPack> packs;
    std::vector<AlphaMaterial> alphaMaterials;
    std::vector<DeviceTexture> textures;
    std::vector<prosper_pt_sampler_desc> samplers;
    uint64_t texelBytes = 0;       // level-0 texel footprint of the textures as the caller gave them (4 B per texel)
    uint64_t alphaBoundBytes = 0;
    uint32_t packedMaterials = 0;
    // device side
    uint8_t *dBlocks[kVersions] = {};
    size_t blockBytes = 0, packsOffset = 0, alphaOffset = 0, texturesOffset = 0;
    hipEvent_t versionFree[kVersions] = {};
    bool versionUsed[kVersions] = {};
    hipStream_t versionStream[kVersions] = {};
    uint32_t cur = 0;
    uint8_t *staging[kStagingBuffers] = {}; // pinned images of the block
    hipEvent_t stagingDone[kStagingBuffers] = {};
    bool stagingUsed[kStagingBuffers] = {};
    uint32_t stagingNext = 0;
    bool pending = false;          // the mirrors differ from version `cur`
    uint64_t changes = 0;          // bumped whenever an update changed a mirror (a background geometry build compares)
    bool pendingAlphaPatch = false; // ... in a MASK / BLEND material: the any-hit records' copies must follow
    hipStream_t uploadStream = nullptr; // texel copies, re-tiling / BC7 decode, packs, alpha bounds of an update
    hipEvent_t uploaded = nullptr;      // behind the last of them
    bool uploadedRecorded = false;
    hipEvent_t ready = nullptr;         // behind the last flush: every later render's chains wait for it
    bool readyRecorded = false;
    void *linearStaging = nullptr;      // device: the texels of an update as the caller holds them, before re-tiling
    size_t linearStagingBytes = 0;
    void *pinnedStaging = nullptr;      // host, pinned: the same bytes on their way there (pt_materials.hpp create_device_texture)
    size_t pinnedStagingBytes = 0;
    uint32_t updates = 0;
    // texel arrays, packs and alpha bounds an update replaced: a frame in flight may still read them, so they stay until
    // enough has piled up to be worth ONE device synchronisation (kRetireBytes), or the scene goes
    std::vector<const void *> retired;
    uint64_t retiredBytes = 0;
    static constexpr uint64_t kRetireBytes = 256ull << 20;
    ~MaterialState()
    {
        for (uint32_t i = 0; i < kStagingBuffers; ++i)
        {
            if (staging[i]) (void)hipHostFree(staging[i]);
            if (stagingDone[i]) (void)hipEventDestroy(stagingDone[i]);
        }
        for (hipEvent_t e : versionFree)
            if (e) (void)hipEventDestroy(e);
        if (uploaded) (void)hipEventDestroy(uploaded);
        if (ready) (void)hipEventDestroy(ready);
        if (uploadStream) (void)hipStreamDestroy(uploadStream);
        if (linearStaging) (void)hipFree(linearStaging);
        if (pinnedStaging) (void)hipHostFree(pinnedStaging);
    }
};

// What the geometry of the scene is made from, as the host gave it: prosper_pt_update_meshes changes these mirrors (and the
// device copies they describe) and lays the triangles out again.  The device pointer table has room for every geometry
// buffer prosper can create (sMaxGeometryBuffersCount, WorldData.cpp:31), the metadata table is written in place: a mesh
// that has not arrived has no triangle anywhere, so nothing in flight reads its slot.
struct GeometryState
{
    std::vector<prosper_GeometryMetadata> metadatas;
    std::vector<prosper_pt_mesh_info> infos;
    std::vector<prosper_DrawInstance> drawInstances;
    std::vector<void *> buffers;        // device copies of the geometry buffers (scene allocations)
    std::vector<uint64_t> bufferBytes;
    const void **dBufferTable = nullptr;           // device: [PROSPER_PT_MAX_GEOMETRY_BUFFERS]
    prosper_GeometryMetadata *dMetadatas = nullptr; // device: [meshCount]
    // prosper_pt_update_meshes keeps the arrived bytes in host memory (`arrived`); the worker thread of the next geometry build
    // (MeshBuild, pt_geometry.cpp) copies them to the device, on its own stream, before it lays the triangles out - the
    // calling thread touches no stream.  While a build runs `buffers` belongs to the worker (it allocates new geometry
    // buffers); the calling thread goes by bufferBytes (0: no such buffer yet).
    struct ArrivedMesh
    {
        uint32_t meshIndex = 0, bufferIndex = 0;
        uint64_t byteOffset = 0;
        std::vector<uint8_t> bytes;
        prosper_GeometryMetadata metadata = {};
    };
    std::vector<ArrivedMesh> arrived;
    bool dirty = false;      // meshes arrived that no build has taken up yet
    uint32_t meshUpdates = 0; // calls that handed meshes over
    uint32_t installs = 0;    // background builds whose result became the scene
};
struct MeshBuild;

// multi-GPU state of a context (pt_tiling.cpp): the communicator, its stream and the root's staging buffer
struct TilingState;

// scene-lifetime device memory of a context (prosper_pt.cpp): freed by the next upload / destroy
int device_alloc(prosper_pt_ctx *ctx, size_t bytes, void **out);
void device_free(prosper_pt_ctx *ctx, const void *p);
int upload(prosper_pt_ctx *ctx, const void *src, size_t bytes, void **out);

} // namespace ppt

#define PPT_HIP(call)                                                                                                  \
    do                                                                                                                 \
    {                                                                                                                  \
        const hipError_t e_ = (call);                                                                                  \
        if (e_ != hipSuccess)                                                                                          \
            return ppt::fail(PROSPER_PT_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_));                   \
    } while (0)

struct prosper_pt_ctx
{
    int device = 0;
    uint32_t flags = 0;
    prosper_pt_debug_options debug = {}; // tuning / test options (prosper_pt_set_debug_options); never the environment
    std::vector<ppt::DeviceAllocation> sceneAllocations;
    std::atomic<uint64_t> sceneBytes{0}; // (the worker thread of a mesh build allocates too)
    bool haveScene = false;
    ppt::DeviceScene scene = {};
    prosper_pt_scene_stats stats = {};
    uint32_t packedMaterials = 0; // materials whose three textures are interleaved (MaterialPack)
    bool rawRecords = false;      // the scene keeps 64-byte raw shading records (RawShadeTriangle)
    uint64_t sceneStamp = 0;      // bumped whenever the geometry the rays see changes (upload, refit, rebuild)
    uint64_t alphaTriangleCount = 0, alphaBoundBytes = 0; // any-hit records and bytes of alpha bounds (AlphaMaterial)
    // light buffers are re-uploaded every frame in prosper; keep their device addresses mutable
    ppt::LightState *lights = nullptr; // the scene's light buffers (device copies in the scene's allocation list)
    ppt::MaterialState *materialState = nullptr; // mirrors + versions of the material / texture tables (pt_materials.cpp)
    ppt::GeometryState *geometry = nullptr;       // mirrors of the mesh tables and geometry buffers (prosper_pt_update_meshes)
    ppt::MeshBuild *meshBuild = nullptr;          // the geometry a worker thread is building from them, if any
    std::vector<ppt::AccelState *> retiredAccel;  // replaced generations: frames in flight may still use their events / staging
    std::mutex allocMutex;                        // sceneAllocations / sceneBytes (the worker thread allocates too)
    // the worker thread's stream: a plain one, made by the worker at first need - after prosper_pt_create has given the
    // hardware queues to the work streams (ensure_build_stream)
    hipStream_t buildStream = nullptr;
    // Pinned staging for the large host <-> device copies of a geometry build (pt_geometry.cpp staged_copy): a copy from
    // pageable memory has the runtime pin the caller's pages for its duration, and when those pages are freed soon after - the
    // builder's vectors are - the unmapping goes through the GPU driver and stops every queue of the process for 20-30 ms
    void *pinnedStaging = nullptr;
    size_t pinnedStagingBytes = 0;

    float4 *hdr = nullptr; // current HDR buffer (internal or caller-owned)
    float4 *ownedHdr = nullptr;
    size_t ownedHdrBytes = 0;
    void *externalHdr = nullptr;
    size_t externalHdrBytes = 0;
    uint32_t localWidth = 0, height = 0;

    unsigned long long *dCounters = nullptr; // kStageCount x 16 u64: one block of work counters per kernel stage
    uint32_t *dWorkCounter = nullptr;        // work-distribution counter of the persistent kernel
    // wavefront workspace (one allocation, carved into the WavefrontBuffers arrays)
    // global overflow of the traversal stacks (only for trees whose stack bound exceeds the LDS stack)
    uint64_t wfSlots = 0;

    bool kernelTiming = false;
    static constexpr uint32_t kMaxTimedLaunches = 96;
    hipEvent_t events[kMaxTimedLaunches + 1] = {};
    uint32_t eventStage[kMaxTimedLaunches] = {};
    uint32_t timedLaunches = 0;
    bool timingValid = false;

    void *restirScratch = nullptr; // device copies of host G-buffer inputs (prosper_pt_restir_di_trace)
    size_t restirScratchBytes = 0;
    uint32_t *toneLut = nullptr; // dim^3 R9G9B9E5 texels
    uint32_t toneLutDim = 0;
    void *toneScratch = nullptr; // RGBA8 output when the caller only wants a host copy
    size_t toneScratchBytes = 0;

    // Everything a render has in flight between its first launch and its accumulate kernel: the wavefront
    // workspace, the stack-overflow array and the two launch chains (pt_kernels.hpp WavefrontChains) with their
    // timing events.  kRenderSlots slots = that many frames in flight (PROSPER_PT_RENDER_PIPELINED), the role `nextFrame` and
    // the per-frame descriptor sets play in RtReference::record; everything else uses slot 0.
    struct RenderSlot
    {
        int32_t *stackOverflow = nullptr;
        size_t stackOverflowBytes = 0;
        void *wfBlock = nullptr;
        size_t wfBytes = 0;
        hipEvent_t chainJoin[ppt::kMaxChains] = {};
        hipEvent_t chainEvents[ppt::kMaxChains][kMaxTimedLaunches + 1] = {};
        uint32_t chainStage[ppt::kMaxChains][kMaxTimedLaunches] = {};
        uint32_t chainLaunches[ppt::kMaxChains] = {};
        hipEvent_t free = nullptr; // recorded after the accumulate kernel of the slot's last render
        bool freeRecorded = false;
        // tiles by the cost of a probe ray (pt_wavefront.hip "tile order"), valid for the view in `orderKey`
        uint32_t *tileOrder = nullptr; // [tiles] + scratch [tiles + 512]
        size_t tileOrderTiles = 0;
        struct OrderKey
        {
            float camera[14];
            uint32_t width, height, stripeWidth, stripeIndex, stripeCount, localWidth;
            uint64_t sceneStamp;
        } orderKey = {};
        bool orderValid = false;
    };
    // prosper keeps two frames in flight; a third one fills the machine better at the batch sizes of a multi-GPU
    // rank share (1/4 share 0.71 -> 0.66 ms, C3 19.3 -> 18.9 ms; profiles/r01_pipelined.txt)
    static constexpr uint32_t kRenderSlots = 3;
    // The internal streams, three in all (+ the caller's = the four hardware queues of the device; more streams
    // share queues and serialise): a pipelined render's chain runs on workStreams[slot], the two chains of an
    // in-order render on workStreams[0] and [1].  All ordering between them goes through events.
    hipStream_t workStreams[kRenderSlots] = {};
    // experiment (debug option pipelinedChains = 2): a second chain per frame in flight; created on first use
    hipStream_t extraStreams[kRenderSlots] = {};
    RenderSlot slots[kRenderSlots];
    uint32_t lastSlot = 0;  // of the last render
    uint32_t timedSlot = 0; // of the last render that ran with kernel timing on (timing readout)
    hipEvent_t chainFork = nullptr;

    // stripe partition of the last render (prosper_pt_tile_desc) and the multi-GPU gather state
    uint32_t lastWidth = 0;
    uint32_t stripeWidth = 0, stripeIndex = 0, stripeCount = 1;
    ppt::TilingState *tiling = nullptr;
    ppt::AccelState *accel = nullptr;
    prosper_ModelInstanceTransforms *dTransforms = nullptr; // the scene's transform table (mutable alias)
};
