// pt_geometry.cpp — see pt_geometry.hpp.
#include "pt_geometry.hpp"

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <exception>
#include <memory>
#include <stdexcept>
#include <thread>

#include "pt_kernels.hpp"
#include "pt_materials.hpp"
#include "pt_scene.hpp"

namespace ppt
{

// the context's debug options as the builder and the wavefront launcher take them
BvhBuildOptions build_options(const prosper_pt_ctx *ctx)
{
    const prosper_pt_debug_options &d = ctx->debug;
    BvhBuildOptions o;
    o.sahTraversalCost = d.sahTraversalCost;
    o.boxPad = d.boxPad;
    o.leafSize = d.leafSize;
    o.buildThreads = d.buildThreads;
    o.topEntries = d.topEntries;
    o.nodeOrder = d.nodeOrder;
    o.childOrder = d.childOrder;
    o.buildTiming = d.buildTiming;
    return o;
}

GeometryTarget context_target(prosper_pt_ctx *ctx)
{
    GeometryTarget t;
    t.s = &ctx->scene;
    t.acc = ctx->accel;
    t.stats = &ctx->stats;
    t.alphaTriangleCount = &ctx->alphaTriangleCount;
    t.buildOpt = build_options(ctx);
    t.flatBvh = ctx->debug.flatBvh != 0;
    t.noUploadRefit = ctx->debug.noUploadRefit != 0;
    t.rawRecords = ctx->debug.rawRecords != 0; // (refused by prosper_pt_set_debug_options unless built with -DPPT_EXPERIMENTS)
    return t;
}

// A large copy between host memory the build is about to FREE (the builder's vectors, the arrived meshes' bytes) and the
// device, through the context's pinned staging area and waited for: see prosper_pt_ctx::pinnedStaging.  Small copies take the
// runtime's own staging path.
constexpr size_t kStagedCopyBytes = 64u << 10;
static int staged_copy(prosper_pt_ctx *ctx, hipStream_t stream, void *dst, const void *src, size_t bytes, hipMemcpyKind kind)
{
    if (bytes == 0) return PROSPER_PT_OK;
    if (bytes < kStagedCopyBytes)
    {
        PPT_HIP(hipMemcpyAsync(dst, src, bytes, kind, stream));
        PPT_HIP(hipStreamSynchronize(stream));
        return PROSPER_PT_OK;
    }
    if (bytes > ctx->pinnedStagingBytes)
    {
        if (ctx->pinnedStaging) PPT_HIP(hipHostFree(ctx->pinnedStaging));
        ctx->pinnedStaging = nullptr;
        ctx->pinnedStagingBytes = 0;
        const size_t capacity = bytes + bytes / 4;
        PPT_HIP(hipHostMalloc(&ctx->pinnedStaging, capacity, hipHostMallocDefault));
        ctx->pinnedStagingBytes = capacity;
    }
    if (kind == hipMemcpyHostToDevice)
    {
        std::memcpy(ctx->pinnedStaging, src, bytes);
        PPT_HIP(hipMemcpyAsync(dst, ctx->pinnedStaging, bytes, kind, stream));
        PPT_HIP(hipStreamSynchronize(stream));
    }
    else
    {
        PPT_HIP(hipMemcpyAsync(ctx->pinnedStaging, src, bytes, kind, stream));
        PPT_HIP(hipStreamSynchronize(stream));
        std::memcpy(dst, ctx->pinnedStaging, bytes);
    }
    return PROSPER_PT_OK;
}

// The refit's GPU work on `stream` for the node / triangle arrays of scene version `version`: exact bounds level by level,
// every node re-encoded, the tree's surface-area measure into the version's cost slot (read back through hCost / costEvent).
int enqueue_refit(AccelState *acc, float padCoeff, BvhNode *nodes, const WorldTriangle *tris, uint32_t version, hipStream_t stream)
{
    const uint32_t slot = version % AccelState::kCostSlots;
    PPT_HIP(hipMemsetAsync(acc->dCost + slot, 0, sizeof(float), stream));
    launch_refit(
        nodes, tris, acc->dNodeBounds, acc->dRefitOrder, acc->levelOffsets.data(), (uint32_t)acc->levelOffsets.size() - 1u,
        acc->nodeCount, padCoeff, acc->dCost + slot, stream);
    PPT_HIP(hipGetLastError());
    PPT_HIP(hipMemcpyAsync(acc->hCost + slot, acc->dCost + slot, sizeof(float), hipMemcpyDeviceToHost, stream));
    PPT_HIP(hipEventRecord(acc->costEvent[slot], stream));
    acc->costPending[slot] = true;
    acc->costSequence[slot] = ++acc->refitSequence;
    return PROSPER_PT_OK;
}

// Takes the newest measure that has arrived (wait: also waits for the newest refit's) into lastCostRatio.
int poll_refit_cost(AccelState *acc, bool wait)
{
    for (uint32_t i = 0; i < AccelState::kCostSlots; ++i)
    {
        if (!acc->costPending[i]) continue;
        const bool newest = acc->costSequence[i] == acc->refitSequence;
        if (wait && newest) PPT_HIP(hipEventSynchronize(acc->costEvent[i]));
        if (hipEventQuery(acc->costEvent[i]) != hipSuccess) continue;
        acc->costPending[i] = false;
        if (acc->costSequence[i] > acc->costRead)
        {
            acc->costRead = acc->costSequence[i];
            if (acc->builtCost > 0.0f) acc->lastCostRatio = acc->hCost[i] / acc->builtCost;
        }
    }
    return PROSPER_PT_OK;
}

// Nodes + leaf-order triangles of a freshly built hierarchy to the device (the node array grows when it has to), and
// what a later refit needs: the nodes ordered by height, every triangle's place in the leaf order, the bounds array.
// Everything runs on the target's stream, which has been waited for when this returns.
int upload_hierarchy(prosper_pt_ctx *ctx, GeometryTarget &t, const BvhBuildResult &bvh)
{
    AccelState *acc = t.acc;
    const size_t nodeBytes = bvh.nodes.size() * sizeof(BvhNode);
    if (nodeBytes > acc->nodeCapacityBytes)
    {
        void *d = nullptr;
        const size_t capacity = nodeBytes + nodeBytes / 4 + 4096; // headroom: a rebuild changes the node count a little
        const int rc = device_alloc(ctx, capacity, &d);
        if (rc != PROSPER_PT_OK) return rc;
        // (a rebuild of the context's own hierarchy: its callers have synchronised the device, nothing reads the old
        //  arrays any more; a background build starts without arrays)
        for (uint32_t ver = 0; ver < AccelState::kVersions; ++ver)
            if (acc->dNodesV[ver])
            {
                device_free(ctx, acc->dNodesV[ver]);
                acc->dNodesV[ver] = nullptr;
            }
        acc->dNodes = static_cast<BvhNode *>(d);
        acc->dNodesV[acc->cur] = acc->dNodes;
        acc->nodeCapacityBytes = capacity;
    }
    for (uint32_t ver = 0; ver < AccelState::kVersions; ++ver) acc->nodesCurrent[ver] = ver == acc->cur;
    {
        const int src = staged_copy(ctx, t.stream, acc->dNodes, bvh.nodes.data(), nodeBytes, hipMemcpyHostToDevice);
        if (src != PROSPER_PT_OK) return src;
    }
    t.s->nodes = acc->dNodes;
    if (acc->total)
    {
        const int src = staged_copy(ctx, t.stream, acc->dPerm, bvh.permutation.data(), bvh.permutation.size() * 4, hipMemcpyHostToDevice);
        if (src != PROSPER_PT_OK) return src;
        launch_permute_triangles(acc->dFlat, acc->dPerm, acc->dTris, (uint32_t)acc->total, t.stream);
        PPT_HIP(hipGetLastError());
    }
    PPT_HIP(hipStreamSynchronize(t.stream)); // (the host arrays of `bvh` have been read)
    t.lap("  nodes + permute");

    // ---- refit tables ----
    const size_t n = bvh.nodes.size();
    acc->nodeCount = (uint32_t)n;
    std::vector<uint32_t> height(n, 0u);
    uint32_t levels = 1;
    for (size_t i = n; i-- > 0;) // children are numbered after their parent (emit order and relayout_nodes both)
    {
        uint32_t h = 0;
        for (uint32_t c = 0; c < bvh.nodes[i].reserved && c < 4u; ++c)
        {
            const int32_t ch = bvh.nodes[i].child[c];
            if (ch < 0) continue;
            if ((size_t)ch <= i) return fail(PROSPER_PT_ERR_UNSUPPORTED, "hierarchy: a child node precedes its parent");
            h = std::max(h, height[(size_t)ch] + 1u);
        }
        height[i] = h;
        levels = std::max(levels, h + 1u);
    }
    acc->levelOffsets.assign(levels + 1u, 0u);
    for (size_t i = 0; i < n; ++i) acc->levelOffsets[height[i] + 1u]++;
    for (uint32_t l = 0; l < levels; ++l) acc->levelOffsets[l + 1u] += acc->levelOffsets[l];
    std::vector<uint32_t> order(n ? n : 1), cursor(acc->levelOffsets.begin(), acc->levelOffsets.end() - 1);
    for (size_t i = 0; i < n; ++i) order[cursor[height[i]]++] = (uint32_t)i;
    if (n > acc->refitCapacityNodes)
    {
        const size_t capacity = n + n / 4 + 64;
        void *d = nullptr;
        int rc;
        // (as above: no refit is reading the old tables)
        if (acc->dNodeBounds) device_free(ctx, acc->dNodeBounds);
        if (acc->dRefitOrder) device_free(ctx, acc->dRefitOrder);
        acc->dNodeBounds = nullptr;
        acc->dRefitOrder = nullptr;
        acc->refitCapacityNodes = 0;
        if ((rc = device_alloc(ctx, capacity * 2 * sizeof(float4), &d))) return rc;
        acc->dNodeBounds = static_cast<float4 *>(d);
        if ((rc = device_alloc(ctx, capacity * sizeof(uint32_t), &d))) return rc;
        acc->dRefitOrder = static_cast<uint32_t *>(d);
        acc->refitCapacityNodes = capacity;
    }
    t.lap("  refit tables (host)");
    {
        const int src = staged_copy(ctx, t.stream, acc->dRefitOrder, order.data(), n * sizeof(uint32_t), hipMemcpyHostToDevice);
        if (src != PROSPER_PT_OK) return src;
    }
    if (!acc->dLeafPosition)
    {
        void *d = nullptr;
        int rc;
        if ((rc = device_alloc(ctx, (size_t)(acc->total ? acc->total : 1) * sizeof(uint32_t), &d))) return rc;
        acc->dLeafPosition = static_cast<uint32_t *>(d);
    }
    if (!acc->dCost)
    {
        void *d = nullptr;
        int rc;
        if ((rc = device_alloc(ctx, sizeof(float) * AccelState::kCostSlots, &d))) return rc;
        acc->dCost = static_cast<float *>(d);
        PPT_HIP(hipHostMalloc((void **)&acc->hCost, sizeof(float) * AccelState::kCostSlots, hipHostMallocDefault));
        for (hipEvent_t &e : acc->costEvent) PPT_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        PPT_HIP(hipEventCreateWithFlags(&acc->sceneEvent, hipEventDisableTiming));
    }
    std::vector<uint32_t> position((size_t)acc->total);
    if (acc->total)
    {
        for (size_t leaf = 0; leaf < bvh.permutation.size(); ++leaf) position[bvh.permutation[leaf]] = (uint32_t)leaf;
        const int src = staged_copy(ctx, t.stream, acc->dLeafPosition, position.data(), position.size() * sizeof(uint32_t), hipMemcpyHostToDevice);
        if (src != PROSPER_PT_OK) return src;
    }
    // one refit right away: the device encoder writes the bytes the emitter wrote (tested), and leaves the tree's
    // surface-area measure to compare later refits with
    // (debug option noUploadRefit keeps the emitter's own bytes, for the test that compares the two; the bounds array
    // and the measure are still computed)
    for (bool &pending : acc->costPending) pending = false;
    acc->builtCost = 0.0f;
    t.lap("  table copies queued");
    if (acc->total)
    {
        // (an empty scene keeps the emitter's root - child boxes at +inf - as it is: the encoder has no bounds to write)
        int rc = enqueue_refit(acc, bvh_pad_coefficient(t.buildOpt), acc->dNodes, acc->dTris, acc->cur, t.stream);
        if (rc != PROSPER_PT_OK) return rc;
        t.lap("  refit queued");
        PPT_HIP(hipStreamSynchronize(t.stream));
        t.lap("  refit done");
        if (t.noUploadRefit)
        {
            const int src = staged_copy(ctx, t.stream, acc->dNodes, bvh.nodes.data(), nodeBytes, hipMemcpyHostToDevice);
            if (src != PROSPER_PT_OK) return src;
        }
        const uint32_t slot = acc->cur % AccelState::kCostSlots;
        acc->costPending[slot] = false;
        acc->builtCost = acc->hCost[slot];
    }
    PPT_HIP(hipStreamSynchronize(t.stream)); // (`order` and `position` have been read)
    acc->costRead = acc->refitSequence;
    acc->lastCostRatio = 1.0f;
    acc->movedSinceBuild.assign(acc->ranges.size(), 0);
    acc->flatStale = false;
    return PROSPER_PT_OK;
}

int layout_geometry(const GeometryState &gs, const prosper_MaterialData *materials, GeometryLayout &out)
{
    const uint32_t n = (uint32_t)gs.drawInstances.size();
    out.triOffsets.assign((size_t)n + 1, 0u);
    out.diFlags.assign(n ? n : 1, 0u);
    out.alphaOffsets.assign(n ? n : 1, 0u);
    uint64_t total = 0, alphaTotal = 0;
    for (uint32_t i = 0; i < n;)
    {
        // one subtree per run of draw instances of the same model instance (World.cpp:480-513 emits them together):
        // prosper's TLAS instance (World.cpp:878-928).  It is active once its BLAS exists, and buildNextBlas waits for ALL
        // sub-meshes of the model (World.cpp:598-606, 909-915): a run with a mesh still loading has no triangles.
        const uint32_t mi = gs.drawInstances[i].modelInstanceIndex;
        uint32_t j = i;
        bool complete = true;
        for (; j < n && gs.drawInstances[j].modelInstanceIndex == mi; ++j)
            complete = complete && mesh_loaded(gs.metadatas[gs.drawInstances[j].meshIndex]);
        out.ranges.push_back(InstancedBvh::Range{(uint32_t)total, 0u});
        out.rangeModelInstance.push_back(mi);
        out.rangeComplete.push_back(complete ? 1 : 0);
        for (uint32_t k = i; k < j; ++k)
        {
            const uint32_t mesh = gs.drawInstances[k].meshIndex;
            out.triOffsets[k] = (uint32_t)total;
            if (!mesh_loaded(gs.metadatas[mesh])) continue; // (its MeshInfo is not there yet either)
            const prosper_pt_mesh_info &info = gs.infos[mesh];
            // World.cpp:646-651: eOpaque iff the mesh's material is AlphaMode_Opaque
            out.diFlags[k] = materials[info.materialIndex].alphaMode == PROSPER_ALPHA_MODE_OPAQUE ? kTriFlagOpaque : 0u;
            if (gs.metadatas[mesh].usesShortIndices == 1) out.diFlags[k] |= kTriFlagShortIndices;
            const uint32_t tris = complete ? info.indexCount / 3 : 0u;
            if (!(out.diFlags[k] & kTriFlagOpaque))
            {
                out.alphaOffsets[k] = (uint32_t)alphaTotal;
                alphaTotal += tris;
            }
            total += tris;
            out.ranges.back().count += tris;
        }
        if (total >= (1ull << 28)) return fail(PROSPER_PT_ERR_UNSUPPORTED, "more than 2^28 triangles after instancing");
        i = j;
    }
    out.triOffsets[n] = (uint32_t)total;
    out.total = total;
    out.alphaTotal = alphaTotal;
    return PROSPER_PT_OK;
}


// The layout into the target's AccelState, the world triangles (flatten kernel, copied to the host) and the hierarchy build
// started on the host's threads.  `changed` == nullptr: every subtree is new.  Otherwise the target's InstancedBvh holds the
// subtrees of a previous layout with the same instances, and only the flagged ones are split again.
int begin_geometry(prosper_pt_ctx *ctx, GeometryTarget &t, GeometryLayout &layout, const std::vector<uint8_t> *changed, GeometryJob &job)
{
    DeviceScene &s = *t.s;
    AccelState *acc = t.acc;
    int rc;
    void *d = nullptr;
    job.t0 = std::chrono::steady_clock::now();
    t.lap("(start)");
    job.alphaTotal = layout.alphaTotal;
    const uint32_t drawInstanceCount = (uint32_t)ctx->geometry->drawInstances.size();
    const uint64_t total = layout.total;
    if ((rc = device_alloc(ctx, layout.alphaOffsets.size() * 4, &d))) return rc;
    s.alphaOffsets = static_cast<const uint32_t *>(d);
    PPT_HIP(hipMemcpyAsync(d, layout.alphaOffsets.data(), layout.alphaOffsets.size() * 4, hipMemcpyHostToDevice, t.stream));
    if ((rc = device_alloc(ctx, layout.triOffsets.size() * 4, &d))) return rc;
    acc->dOffsets = static_cast<uint32_t *>(d);
    PPT_HIP(hipMemcpyAsync(d, layout.triOffsets.data(), layout.triOffsets.size() * 4, hipMemcpyHostToDevice, t.stream));
    if ((rc = device_alloc(ctx, layout.diFlags.size() * 4, &d))) return rc;
    acc->dFlags = static_cast<uint32_t *>(d);
    PPT_HIP(hipMemcpyAsync(d, layout.diFlags.data(), layout.diFlags.size() * 4, hipMemcpyHostToDevice, t.stream));
    s.triangleOffsets = acc->dOffsets;

    const size_t triBytes = sizeof(WorldTriangle) * (size_t)(total ? total : 1);
    if ((rc = device_alloc(ctx, triBytes, &d))) return rc;
    acc->dFlat = static_cast<WorldTriangle *>(d);
    t.lap("tables, allocations");
    launch_flatten_triangles(s, acc->dOffsets, drawInstanceCount, acc->dFlags, acc->dFlat, nullptr, nullptr, (uint32_t)total, t.stream);
    PPT_HIP(hipGetLastError());
    acc->flat.resize((size_t)total);
    if (total && (rc = staged_copy(ctx, t.stream, acc->flat.data(), acc->dFlat, sizeof(WorldTriangle) * (size_t)total, hipMemcpyDeviceToHost))) return rc;
    PPT_HIP(hipStreamSynchronize(t.stream)); // (the layout's host arrays have been read, the world triangles are here)
    t.lap("flatten + read back");
    acc->triOffsets.swap(layout.triOffsets);
    acc->ranges.swap(layout.ranges);
    acc->rangeModelInstance.swap(layout.rangeModelInstance);
    acc->total = total;
    acc->drawInstanceCount = drawInstanceCount;

    const BvhBuildOptions buildOpt = t.buildOpt;
    const bool flatBvh = t.flatBvh;
    const bool keep = changed != nullptr && !flatBvh;
    std::vector<uint8_t> flags = keep ? *changed : std::vector<uint8_t>();
    job.build = std::async(std::launch::async, [acc, total, buildOpt, flatBvh, keep, flags]() {
        BuildOutcome out;
        const auto tBuild = std::chrono::steady_clock::now();
        try
        {
            // debug option flatBvh: one SAH tree over all triangles, as round 1 built it (A/B, hierarchy tests)
            if (flatBvh)
                out.bvh = build_bvh(acc->flat.data(), total, buildOpt);
            else
            {
                try
                {
                    out.bvh = keep ? acc->bvh.adopt(acc->flat.data(), total, acc->ranges, flags, buildOpt)
                                   : acc->bvh.build(acc->flat.data(), total, acc->ranges, buildOpt);
                    out.instanced = true;
                }
                catch (const std::exception &)
                {
                    // the subtrees are split without knowing how deep the re-braided top level above them gets: a
                    // spliced tree can pass the traversal's stack bound where one tree over everything does not
                    out.bvh = build_bvh(acc->flat.data(), total, buildOpt);
                }
            }
        }
        catch (const std::exception &ex)
        {
            out.error = ex.what();
        }
        out.seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - tBuild).count();
        return out;
    });
    return PROSPER_PT_OK;
}

// The arrays the flatten kernel fills for shading and any-hit (they live as long as this geometry), the leaf-order
// triangles, and the hierarchy the host's threads built meanwhile.
int finish_geometry(prosper_pt_ctx *ctx, GeometryTarget &t, GeometryJob &job)
{
    DeviceScene &s = *t.s;
    AccelState *acc = t.acc;
    const uint64_t total = acc->total;
    int rc;
    void *d = nullptr;
    if ((rc = device_alloc(ctx, sizeof(AlphaTriangle) * (size_t)(job.alphaTotal ? job.alphaTotal : 1), &d))) return rc;
    s.alphaTriangles = static_cast<const AlphaTriangle *>(d);
    *t.alphaTriangleCount = job.alphaTotal;
    if ((rc = device_alloc(ctx, (size_t)(total ? total : 1) * 4, &d))) return rc;
    acc->dPerm = static_cast<uint32_t *>(d);
    // decoded 128-byte records; debug option rawRecords (an experiment) keeps the raw 64-byte form instead, decoded per hit (same
    // pixels, tested).  Measured and not made a default for any scene size (profiles/r03_raw_records.txt): even on
    // S-sponza-class, whose 33.6 MB of records outgrow the L2 and whose wf_shade runs at 6.7 TB/s, the ~150 instructions of
    // decoding cost more than the 64 bytes save (wf_shade 910 -> 931 us; C4 687 -> 721, C2 220 -> 248, FlightHelmet 97 -> 104)
    void *dShade = nullptr, *dRaw = nullptr;
    if (t.rawRecords)
    {
        if ((rc = device_alloc(ctx, sizeof(RawShadeTriangle) * (size_t)(total ? total : 1), &dRaw))) return rc;
    }
    else if ((rc = device_alloc(ctx, sizeof(ShadeTriangle) * (size_t)(total ? total : 1), &dShade)))
        return rc;
    s.shadeTriangles = static_cast<const ShadeTriangle *>(dShade);
    s.rawShadeTriangles = static_cast<const RawShadeTriangle *>(dRaw);
    const size_t triBytes = sizeof(WorldTriangle) * (size_t)(total ? total : 1);
    void *dTris = nullptr;
    if ((rc = device_alloc(ctx, triBytes, &dTris))) return rc;
    t.lap("record allocations");
    PPT_HIP(hipMemsetAsync(dTris, 0, triBytes, t.stream));
    acc->dTris = static_cast<WorldTriangle *>(dTris);
    acc->dTrisV[acc->cur] = acc->dTris;
    s.triangles = acc->dTris;

    launch_flatten_triangles(
        s, acc->dOffsets, acc->drawInstanceCount, acc->dFlags, acc->dFlat, static_cast<ShadeTriangle *>(dShade),
        const_cast<AlphaTriangle *>(s.alphaTriangles), (uint32_t)total, t.stream, nullptr, nullptr,
        static_cast<RawShadeTriangle *>(dRaw));
    PPT_HIP(hipGetLastError());
    PPT_HIP(hipStreamSynchronize(t.stream));
    t.lap("records");

    BuildOutcome built = job.build.get();
    t.lap("wait for host build");
    if (!built.error.empty()) return fail(PROSPER_PT_ERR_UNSUPPORTED, "BVH build failed: " + built.error);
    acc->instanced = built.instanced;
    if ((rc = upload_hierarchy(ctx, t, built.bvh))) return rc;
    t.lap("hierarchy upload + refit");

    prosper_pt_scene_stats &st = *t.stats;
    st.triangleCount = total;
    st.nodeCount = built.bvh.nodes.size();
    st.nodeBytes = sizeof(BvhNode);
    st.triangleBytes = sizeof(WorldTriangle);
    st.maxDepth = built.bvh.maxDepth;
    st.buildSeconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - job.t0).count();
    st.bvhBuildSeconds = built.seconds;
    st.alphaTriangleCount = job.alphaTotal;
    return PROSPER_PT_OK;
}

// A geometry generation under construction (prosper_pt_update_meshes): the worker thread writes only what hangs off this
// object - a private copy of the scene descriptor, a new AccelState, new device arrays - until the main thread installs it.
struct MeshBuild
{
    DeviceScene scene = {};
    AccelState *acc = nullptr;
    prosper_pt_scene_stats stats = {};
    uint64_t alphaTriangleCount = 0;
    std::future<int> done;    // the worker's return code ...
    std::string error;        // ... and message (fail() is thread-local)
    uint64_t materialChanges = 0; // MaterialState::changes when the build took its copy of the alpha-material table
    void *dAlphaSnapshot = nullptr;
    std::vector<void *> allocations; // what the worker allocated (given back if the build fails)
    std::vector<GeometryState::ArrivedMesh> arrived; // the meshes whose bytes this build writes to the device
    std::vector<void *> newBuffers;  // geometry buffers it created (they stay: their bytes are on the device)
};

// upload / destroy: the worker is waited for, its result dropped (its device arrays are scene allocations)
void discard_mesh_build(prosper_pt_ctx *ctx)
{
    MeshBuild *b = ctx->meshBuild;
    if (!b) return;
    if (b->done.valid()) (void)b->done.get();
    delete b->acc;
    delete b;
    ctx->meshBuild = nullptr;
}

// ---- streamed-in meshes: the new geometry is made by a worker thread, beside the frame loop ----
// The worker's stream, made by the worker thread the first time a context needs it - AFTER prosper_pt_create has given the
// device's hardware queues to the work streams: a plain stream, which then shares a queue with one of those.  What the worker
// enqueues (copies, a few dozen small kernels) waits behind the frames already queued there - three with a paced host, a
// few milliseconds per build.  A high-priority stream gets a hardware queue of its own and never waits, but its first use costs
// 20 ms, and more queues alive cost the frame loop 2-5 % with frames in flight and 10-17 % in order; a plain stream that claims
// a queue BEFORE the work streams do leaves two of those sharing one (profiles/r04_mesh_streams.txt).  Its creation takes 7 ms:
// on the worker, not in the frame loop.
static int ensure_build_stream(prosper_pt_ctx *ctx)
{
    if (!ctx->buildStream) PPT_HIP(hipStreamCreateWithFlags(&ctx->buildStream, hipStreamNonBlocking));
    return PROSPER_PT_OK;
}

// A worker for what the mirrors hold now (the caller has made sure none is running).
static int start_mesh_build_impl(prosper_pt_ctx *ctx, bool rebuild);
int start_mesh_build(prosper_pt_ctx *ctx, bool rebuild)
{
    try
    {
        return start_mesh_build_impl(ctx, rebuild);
    }
    catch (const std::exception &ex)
    {
        return fail(PROSPER_PT_ERR_INVALID_ARGUMENT, std::string("starting the background geometry build: ") + ex.what());
    }
}
static int start_mesh_build_impl(prosper_pt_ctx *ctx, bool rebuild)
{
    GeometryState *gs = ctx->geometry;
    MaterialState *ms = ctx->materialState;
    AccelState *old = ctx->accel;
    auto layout = std::make_shared<GeometryLayout>();
    int rc;
    if ((rc = layout_geometry(*gs, ms->materials.data(), *layout))) return rc;
    // the subtrees that are split again: model instances that became complete, and those that moved since the last build
    auto changed = std::make_shared<std::vector<uint8_t>>(layout->ranges.size(), 0);
    for (size_t r = 0; r < layout->ranges.size(); ++r)
        (*changed)[r] = (r >= old->ranges.size() || layout->ranges[r].count != old->ranges[r].count ||
                         (r < old->movedSinceBuild.size() && old->movedSinceBuild[r])) ? 1 : 0;
    MeshBuild *b = new (std::nothrow) MeshBuild();
    if (b) b->acc = new (std::nothrow) AccelState();
    if (!b || !b->acc)
    {
        delete b;
        return fail(PROSPER_PT_ERR_INVALID_ARGUMENT, "out of host memory");
    }
    // (until the worker runs, a failure below - an exception included - drops the build)
    std::unique_ptr<MeshBuild, void (*)(MeshBuild *)> guard(b, [](MeshBuild *p) {
        delete p->acc;
        delete p;
    });
    b->scene = ctx->scene;
    b->stats = ctx->stats;
    b->acc->transforms = old->transforms; // the newest the caller gave, flushed or not
    b->acc->refits = old->refits;
    b->acc->rebuilds = old->rebuilds + (rebuild ? 1u : 0u);
    // The subtrees of the last build move to the new generation.  (The old one keeps rendering with its node array; should it
    // have to be rebuilt before the switch, prosper_pt_rebuild_hierarchy waits for this build instead.)
    const bool keep = old->instanced && ctx->debug.flatBvh == 0;
    if (keep)
    {
        b->acc->bvh.swap(old->bvh);
        old->instanced = false;
    }
    // the flatten kernel copies every non-opaque triangle's AlphaMaterial into its any-hit record: from a private copy of
    // the table as the mirrors hold it now - the versioned device tables rotate under the frame loop's hands
    b->materialChanges = ms->changes;
    void *d = nullptr;
    const size_t alphaBytes = ms->alphaMaterials.size() * sizeof(AlphaMaterial);
    auto snapshot = std::make_shared<std::vector<AlphaMaterial>>(ms->alphaMaterials);
    if ((rc = device_alloc(ctx, alphaBytes, &d))) return rc;
    b->dAlphaSnapshot = d;
    b->scene.alphaMaterials = static_cast<const AlphaMaterial *>(d);
    GeometryTarget t = context_target(ctx);
    // A build beside the frame loop leaves the application half of the host's threads (of the container's CPU quota, where
    // there is one: bvh_build.cpp cpu_quota).
    if (t.buildOpt.buildThreads == 0) t.buildOpt.buildThreads = std::max(1u, default_build_threads() / 2u);
    t.s = &b->scene;
    t.acc = b->acc;
    t.stats = &b->stats;
    t.alphaTriangleCount = &b->alphaTriangleCount;
    b->arrived.swap(gs->arrived);
    const int device = ctx->device;
    // debug option failNextUpdate: the worker gives up half way (the test of what a failed build leaves behind)
    const bool failHalfWay = ctx->debug.failNextUpdate != 0;
    ctx->debug.failNextUpdate = 0;
    auto work = [ctx, b, t, layout, changed, snapshot, keep, device, alphaBytes, failHalfWay]() mutable -> int {
        ppt::g_allocationLog = &b->allocations;
        auto run = [&]() -> int {
            PPT_HIP(hipSetDevice(device));
            int r = ensure_build_stream(ctx);
            if (r != PROSPER_PT_OK) return r;
            t.stream = ctx->buildStream;
            // the arrived meshes: new geometry buffers, bytes, metadata entries
            GeometryState *gs = ctx->geometry;
            for (const GeometryState::ArrivedMesh &a : b->arrived)
            {
                if (!gs->buffers[a.bufferIndex])
                {
                    void *nb = nullptr;
                    const size_t bytes = (size_t)gs->bufferBytes[a.bufferIndex];
                    PPT_HIP(hipMalloc(&nb, bytes ? bytes : 16)); // (not through the log of what a failed build gives back)
                    {
                        const std::lock_guard<std::mutex> lock(ctx->allocMutex);
                        ctx->sceneAllocations.push_back({nb, bytes});
                        ctx->sceneBytes += bytes;
                    }
                    PPT_HIP(hipMemsetAsync(nb, 0, bytes, t.stream));
                    gs->buffers[a.bufferIndex] = nb;
                    b->newBuffers.push_back(nb);
                    PPT_HIP(hipMemcpyAsync(gs->dBufferTable + a.bufferIndex, &b->newBuffers.back(), sizeof(void *), hipMemcpyHostToDevice, t.stream));
                    PPT_HIP(hipStreamSynchronize(t.stream)); // (newBuffers may move when it grows)
                }
                if (!a.bytes.empty())
                {
                    const int src = staged_copy(ctx, t.stream, static_cast<uint8_t *>(gs->buffers[a.bufferIndex]) + a.byteOffset, a.bytes.data(), a.bytes.size(), hipMemcpyHostToDevice);
                    if (src != PROSPER_PT_OK) return src;
                }
                PPT_HIP(hipMemcpyAsync(gs->dMetadatas + a.meshIndex, &a.metadata, sizeof(prosper_GeometryMetadata), hipMemcpyHostToDevice, t.stream));
            }
            PPT_HIP(hipMemcpyAsync(b->dAlphaSnapshot, snapshot->data(), alphaBytes, hipMemcpyHostToDevice, t.stream));
            PPT_HIP(hipStreamSynchronize(t.stream));
            // its own transform table: the newest transforms as of the start of this build
            AccelState *acc = b->acc;
            void *dT = nullptr;
            r = device_alloc(ctx, sizeof(prosper_ModelInstanceTransforms) * (acc->transforms.size() ? acc->transforms.size() : 1), &dT);
            if (r != PROSPER_PT_OK) return r;
            if (!acc->transforms.empty())
                PPT_HIP(hipMemcpyAsync(dT, acc->transforms.data(), sizeof(prosper_ModelInstanceTransforms) * acc->transforms.size(), hipMemcpyHostToDevice, t.stream));
            acc->dTransformsV[0] = static_cast<prosper_ModelInstanceTransforms *>(dT);
            b->scene.modelInstanceTransforms = acc->dTransformsV[0];
            GeometryJob job;
            if ((r = begin_geometry(ctx, t, *layout, keep ? changed.get() : nullptr, job))) return r;
            if (failHalfWay) return fail(PROSPER_PT_ERR_UNSUPPORTED, "debug option failNextUpdate is set");
            return finish_geometry(ctx, t, job);
        };
        int r;
        try
        {
            r = run();
        }
        catch (const std::exception &ex) // (the C-ABI never throws: the main thread's future.get() must not either)
        {
            r = fail(PROSPER_PT_ERR_INVALID_ARGUMENT, std::string("worker thread: ") + ex.what());
        }
        if (r != PROSPER_PT_OK) b->error = ppt::g_lastErrorStorage;
        return r;
    };
    try
    {
        b->done = std::async(std::launch::async, std::move(work));
    }
    catch (const std::exception &) // (std::system_error: no thread to be had)
    {
    }
    if (!b->done.valid())
    {
        gs->arrived.swap(b->arrived); // (no worker: the meshes wait for the next build)
        return fail(PROSPER_PT_ERR_INVALID_ARGUMENT, "the worker thread of the geometry build could not be started");
    }
    guard.release();
    gs->dirty = false;
    ctx->meshBuild = b;
    return PROSPER_PT_OK;
}

// The finished build becomes the scene; what went on meanwhile - moved instances, changed materials, more meshes - follows.
// wait: block until nothing handed over so far is outstanding.
int poll_mesh_build(prosper_pt_ctx *ctx, bool wait)
{
    if (ctx->meshBuild) PPT_HIP(hipSetDevice(ctx->device));
    while (MeshBuild *b = ctx->meshBuild)
    {
        if (!wait && b->done.wait_for(std::chrono::seconds(0)) != std::future_status::ready) return PROSPER_PT_OK;
        const int built = b->done.get();
        GeometryState *gs = ctx->geometry;
        MaterialState *ms = ctx->materialState;
        ctx->meshBuild = nullptr;
        if (built != PROSPER_PT_OK)
        {
            // the scene stays as it is (without the subtrees that went with the build: its next hierarchy is built whole);
            // the meshes wait for the next prosper_pt_update_meshes / _finish_mesh_updates
            const std::string why = b->error;
            // nothing of it was ever installed: once its stream is idle its arrays can go
            if (ctx->buildStream) (void)hipStreamSynchronize(ctx->buildStream);
            for (GeometryState::ArrivedMesh &a : gs->arrived) b->arrived.push_back(std::move(a)); // (those that came meanwhile, behind)
            gs->arrived.swap(b->arrived);
            for (void *p : b->allocations) device_free(ctx, p);
            device_free(ctx, b->dAlphaSnapshot);
            delete b->acc;
            delete b;
            gs->dirty = true;
            return fail(built, "the background build of the streamed-in meshes failed: " + why);
        }
        AccelState *old = ctx->accel, *acc = b->acc;
        const std::vector<prosper_ModelInstanceTransforms> latest = old->transforms;
        // the old generation: frames in flight may still read its arrays, and its events / pinned staging may be in use
        const void *gone[] = {old->dFlat, old->dPerm, old->dLeafPosition, old->dOffsets, old->dFlags, old->dNodeBounds, old->dRefitOrder,
                              old->dCost, ctx->scene.alphaOffsets, ctx->scene.alphaTriangles, ctx->scene.shadeTriangles,
                              ctx->scene.rawShadeTriangles, b->dAlphaSnapshot};
        for (const void *p : gone) retire(ctx, p);
        for (uint32_t v = 0; v < AccelState::kVersions; ++v)
        {
            retire(ctx, old->dTrisV[v]);
            retire(ctx, old->dNodesV[v]);
            retire(ctx, old->dTransformsV[v]);
        }
        ctx->retiredAccel.push_back(old);
        ctx->accel = acc;
        DeviceScene &s = ctx->scene;
        s.nodes = b->scene.nodes;
        s.triangles = b->scene.triangles;
        s.triangleOffsets = b->scene.triangleOffsets;
        s.alphaOffsets = b->scene.alphaOffsets;
        s.alphaTriangles = b->scene.alphaTriangles;
        s.shadeTriangles = b->scene.shadeTriangles;
        s.rawShadeTriangles = b->scene.rawShadeTriangles;
        s.modelInstanceTransforms = b->scene.modelInstanceTransforms;
        ctx->dTransforms = acc->dTransformsV[0];
        ctx->alphaTriangleCount = b->alphaTriangleCount;
        ctx->stats.triangleCount = b->stats.triangleCount;
        ctx->stats.nodeCount = b->stats.nodeCount;
        ctx->stats.maxDepth = b->stats.maxDepth;
        ctx->stats.buildSeconds = b->stats.buildSeconds;
        ctx->stats.bvhBuildSeconds = b->stats.bvhBuildSeconds;
        ctx->stats.alphaTriangleCount = b->stats.alphaTriangleCount;
        ctx->stats.deviceBytes = ctx->sceneBytes;
        ctx->sceneStamp++;
        gs->installs++;
        // a material that changed while the build ran: its any-hit records are rewritten by the next flush of the tables
        if (ms->changes != b->materialChanges && ctx->alphaTriangleCount)
        {
            ms->pending = true;
            ms->pendingAlphaPatch = true;
        }
        delete b;
        int rc;
        // instances that moved while the build ran: a refit of the new generation at the head of the next render
        if (!latest.empty() && std::memcmp(latest.data(), acc->transforms.data(), sizeof(prosper_ModelInstanceTransforms) * latest.size()) != 0)
            if ((rc = stage_transforms(ctx, latest.data(), (uint32_t)latest.size()))) return rc;
        if ((rc = collect_retired(ctx))) return rc;
        if (gs->dirty && (rc = start_mesh_build(ctx))) return rc;
    }
    if (wait && ctx->geometry && ctx->geometry->dirty && ctx->accel)
    {
        // (a failed build left meshes waiting: try again)
        const int rc = start_mesh_build(ctx);
        return rc != PROSPER_PT_OK ? rc : poll_mesh_build(ctx, true);
    }
    return PROSPER_PT_OK;
}

} // namespace ppt

using namespace ppt;

extern "C" {

// WorldData::pollMeshWorker + World::buildNextBlas for the meshes that arrived (prosper_pt.h "streamed-in meshes").
int prosper_pt_update_meshes(prosper_pt_ctx *ctx, const prosper_pt_mesh_update *meshes, uint32_t count)
{
    if (!ctx || (!meshes && count)) return fail(PROSPER_PT_ERR_INVALID_ARGUMENT, "prosper_pt_update_meshes: null argument");
    if (!ctx->haveScene || !ctx->accel || !ctx->geometry || !ctx->materialState) return fail(PROSPER_PT_ERR_NO_SCENE, "no scene uploaded");
    if (count == 0) return PROSPER_PT_OK;
    GeometryState *gs = ctx->geometry;
    AccelState *acc = ctx->accel;
    MaterialState *ms = ctx->materialState;
    if (acc->stale) return fail(PROSPER_PT_ERR_NO_SCENE, "the last prosper_pt_update_transforms failed: update the transforms again (or upload the scene) first");

    // ---- everything is checked against the mirrors before anything is touched ----
    std::vector<uint64_t> bufferBytes(PROSPER_PT_MAX_GEOMETRY_BUFFERS, 0);
    for (size_t b = 0; b < gs->bufferBytes.size(); ++b) bufferBytes[b] = gs->bufferBytes[b]; // (0: no such buffer yet)
    std::vector<uint8_t> arriving(gs->metadatas.size(), 0);
    for (uint32_t i = 0; i < count; ++i)
    {
        const prosper_pt_mesh_update &u = meshes[i];
        const prosper_GeometryMetadata &m = u.metadata;
        const std::string name = "prosper_pt_update_meshes: mesh " + std::to_string(u.meshIndex);
        if (u.meshIndex >= gs->metadatas.size()) return fail(PROSPER_PT_ERR_INVALID_ARGUMENT, name + " exceeds the scene's meshCount");
        if (mesh_loaded(gs->metadatas[u.meshIndex]) || arriving[u.meshIndex])
            return fail(PROSPER_PT_ERR_INVALID_ARGUMENT, name + " has been loaded already");
        arriving[u.meshIndex] = 1;
        if (m.bufferIndex >= PROSPER_PT_MAX_GEOMETRY_BUFFERS) return fail(PROSPER_PT_ERR_SCENE, name + ": bufferIndex out of range");
        if (bufferBytes[m.bufferIndex] == 0)
        {
            if (u.bufferByteSize == 0 || u.bufferByteSize % 4 != 0)
                return fail(PROSPER_PT_ERR_SCENE, name + ": a new geometry buffer needs its bufferByteSize (a multiple of 4)");
            bufferBytes[m.bufferIndex] = u.bufferByteSize;
        }
        const uint64_t size = bufferBytes[m.bufferIndex];
        if (u.byteOffset % 4 != 0 || u.byteCount % 4 != 0 || u.byteOffset > size || u.byteCount > size - u.byteOffset)
            return fail(PROSPER_PT_ERR_SCENE, name + ": its byte range does not fit the geometry buffer");
        if (u.byteCount && !u.bytes) return fail(PROSPER_PT_ERR_INVALID_ARGUMENT, name + ": bytes missing");
        if (u.info.materialIndex >= ms->materials.size()) return fail(PROSPER_PT_ERR_SCENE, name + ": materialIndex out of range");
        if (u.info.indexCount % 3 != 0) return fail(PROSPER_PT_ERR_SCENE, name + ": indexCount is not a multiple of 3");
        if (m.indicesOffset == PROSPER_PT_ABSENT || m.positionsOffset == PROSPER_PT_ABSENT)
            return fail(PROSPER_PT_ERR_SCENE, name + ": indices/positions are required");
        // every stream inside the bytes that came with the mesh (words of the buffer)
        const uint64_t w0 = u.byteOffset / 4, w1 = (u.byteOffset + u.byteCount) / 4;
        auto inside = [&](uint64_t first, uint64_t words) { return first >= w0 && first <= w1 && words <= w1 - first; };
        const bool shortIdx = m.usesShortIndices == 1;
        const uint64_t indexFirst = shortIdx ? (uint64_t)m.indicesOffset / 2 : (uint64_t)m.indicesOffset;
        const uint64_t indexEnd = shortIdx ? ((uint64_t)m.indicesOffset + u.info.indexCount + 1) / 2 : (uint64_t)m.indicesOffset + u.info.indexCount;
        if (!inside(indexFirst, indexEnd - indexFirst)) return fail(PROSPER_PT_ERR_SCENE, name + ": indices lie outside its bytes");
        if (!inside(m.positionsOffset, 2ull * u.info.vertexCount)) return fail(PROSPER_PT_ERR_SCENE, name + ": positions lie outside its bytes");
        const uint32_t attrs[3] = {m.normalsOffset, m.tangentsOffset, m.texCoord0sOffset};
        for (uint32_t a : attrs)
            if (a != PROSPER_PT_ABSENT && !inside(a, u.info.vertexCount))
                return fail(PROSPER_PT_ERR_SCENE, name + ": an attribute stream lies outside its bytes");
        const uint8_t *indexBytes = static_cast<const uint8_t *>(u.bytes) + ((uint64_t)m.indicesOffset * (shortIdx ? 2u : 4u) - u.byteOffset);
        for (uint32_t k = 0; k < u.info.indexCount; ++k)
        {
            const uint32_t idx = shortIdx ? (uint32_t) reinterpret_cast<const uint16_t *>(indexBytes)[k]
                                          : reinterpret_cast<const uint32_t *>(indexBytes)[k];
            if (idx >= u.info.vertexCount) return fail(PROSPER_PT_ERR_SCENE, name + ": vertex index out of range");
        }
    }

    // ---- the arrived bytes are kept (the caller's memory may go when this returns); the worker thread of the next build
    //      writes them and the metadata into the device's geometry buffers and tables IN PLACE - nothing in flight reads
    //      those places, the meshes have no triangle anywhere yet ----
    PPT_HIP(hipSetDevice(ctx->device));
    int rc;
    if ((rc = poll_mesh_build(ctx, false))) return rc; // (a finished build is installed first: its subtrees are the ones to keep)
    try
    {
        for (uint32_t i = 0; i < count; ++i)
        {
            const prosper_pt_mesh_update &u = meshes[i];
            GeometryState::ArrivedMesh a;
            a.meshIndex = u.meshIndex;
            a.bufferIndex = u.metadata.bufferIndex;
            a.byteOffset = u.byteOffset;
            a.bytes.assign(static_cast<const uint8_t *>(u.bytes), static_cast<const uint8_t *>(u.bytes) + u.byteCount);
            a.metadata = u.metadata;
            gs->arrived.push_back(std::move(a));
        }
    }
    catch (const std::exception &ex)
    {
        return fail(PROSPER_PT_ERR_INVALID_ARGUMENT, std::string("prosper_pt_update_meshes: ") + ex.what());
    }
    for (uint32_t i = 0; i < count; ++i)
    {
        const uint32_t bi = meshes[i].metadata.bufferIndex;
        if (gs->bufferBytes[bi] == 0) gs->bufferBytes[bi] = bufferBytes[bi]; // (a new buffer: no worker knows of it yet)
        gs->metadatas[meshes[i].meshIndex] = meshes[i].metadata;
        gs->infos[meshes[i].meshIndex] = meshes[i].info;
    }
    gs->meshUpdates++;
    gs->dirty = true;
    return ctx->meshBuild ? PROSPER_PT_OK : start_mesh_build(ctx);
}

int prosper_pt_finish_mesh_updates(prosper_pt_ctx *ctx)
{
    if (!ctx) return fail(PROSPER_PT_ERR_INVALID_ARGUMENT, "prosper_pt_finish_mesh_updates: null argument");
    if (!ctx->haveScene || !ctx->accel) return fail(PROSPER_PT_ERR_NO_SCENE, "no scene uploaded");
    PPT_HIP(hipSetDevice(ctx->device));
    return poll_mesh_build(ctx, true);
}

} // extern "C"
