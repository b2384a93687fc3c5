// pt_geometry.hpp — the geometry of a scene as the traversal needs it: where every draw instance's triangles live, the
// hierarchy over them, the per-triangle records (pt_geometry.cpp; private to the library).  Built by
// prosper_pt_upload_scene, rebuilt by prosper_pt_rebuild_hierarchy, and - as a new GENERATION made by a worker thread beside
// the frame loop - after prosper_pt_update_meshes handed over streamed-in meshes or the refits of moved instances have
// degraded the tree.
#pragma once

#include <chrono>
#include <future>
#include <string>
#include <vector>

#include "bvh_build.hpp"
#include "pt_context.hpp"

namespace ppt
{

// (prosper_pt.cpp) the calling thread's last error message; the scene allocations a mesh build's worker thread makes
extern thread_local std::string g_lastErrorStorage;
extern thread_local std::vector<void *> *g_allocationLog;

// the context's debug options as the builder takes them
BvhBuildOptions build_options(const prosper_pt_ctx *ctx);

// Where a hierarchy build puts its results: the context's own scene and AccelState (prosper_pt_upload_scene, the synchronous
// rebuild - null stream, device idle), or the private ones of a background build (prosper_pt_update_meshes: its own stream,
// nothing of the context is written until the result is installed).  The options are copied: the worker thread never reads
// ctx->debug.
struct GeometryTarget
{
    DeviceScene *s = nullptr;
    AccelState *acc = nullptr;
    prosper_pt_scene_stats *stats = nullptr;
    uint64_t *alphaTriangleCount = nullptr;
    hipStream_t stream = nullptr;
    BvhBuildOptions buildOpt;
    bool flatBvh = false, noUploadRefit = false, rawRecords = false;
    // debug option buildTiming: where a build spends its time (stderr)
    std::chrono::steady_clock::time_point tick = std::chrono::steady_clock::now();
    void lap(const char *what)
    {
        const auto now = std::chrono::steady_clock::now();
        if (buildOpt.buildTiming) std::fprintf(stderr, "[geometry] %-24s %.2f ms\n", what, std::chrono::duration<double, std::milli>(now - tick).count());
        tick = now;
    }
};
GeometryTarget context_target(prosper_pt_ctx *ctx);

// the refit's GPU work for the node / triangle arrays of scene version `version`, and the readback of its measure
int enqueue_refit(AccelState *acc, float padCoeff, BvhNode *nodes, const WorldTriangle *tris, uint32_t version, hipStream_t stream);
int poll_refit_cost(AccelState *acc, bool wait);
// nodes + leaf-order triangles of a freshly built hierarchy to the device, and what a later refit needs
int upload_hierarchy(prosper_pt_ctx *ctx, GeometryTarget &t, const BvhBuildResult &bvh);


// ---- the geometry of the scene: where every draw instance's triangles live, the hierarchy over them, the per-triangle
//      records.  prosper_pt_upload_scene builds it from the view, prosper_pt_update_meshes again from the mirrors
//      (GeometryState) once meshes arrived. ----
struct GeometryLayout
{
    std::vector<uint32_t> triOffsets, diFlags, alphaOffsets; // per draw instance (+ 1 for triOffsets)
    std::vector<InstancedBvh::Range> ranges;                 // one per run of draw instances of a model instance
    std::vector<uint32_t> rangeModelInstance;
    std::vector<uint8_t> rangeComplete;                      // every mesh of the run has been loaded
    uint64_t total = 0, alphaTotal = 0;
};

inline bool mesh_loaded(const prosper_GeometryMetadata &m) { return m.bufferIndex != PROSPER_PT_ABSENT; }
int layout_geometry(const GeometryState &gs, const prosper_MaterialData *materials, GeometryLayout &out);

struct BuildOutcome
{
    BvhBuildResult bvh;
    bool instanced = false;
    double seconds = 0.0;
    std::string error;
};
struct GeometryJob
{
    // (a std::async future joins in its destructor: an early return of the caller waits for the build, which reads the
    //  target's AccelState)
    std::future<BuildOutcome> build;
    std::chrono::steady_clock::time_point t0;
    uint64_t alphaTotal = 0;
};

int begin_geometry(prosper_pt_ctx *ctx, GeometryTarget &t, GeometryLayout &layout, const std::vector<uint8_t> *changed, GeometryJob &job);
int finish_geometry(prosper_pt_ctx *ctx, GeometryTarget &t, GeometryJob &job);

// ---- geometry generations built in the background (prosper_pt_update_meshes, drifted instances) ----
// upload / destroy: the worker is waited for, its result dropped
void discard_mesh_build(prosper_pt_ctx *ctx);
// a worker for what the mirrors hold now (the caller has made sure none is running); rebuild: counts as a re-split
int start_mesh_build(prosper_pt_ctx *ctx, bool rebuild = false);
// the finished build becomes the scene; wait: block until nothing handed over so far is outstanding
int poll_mesh_build(prosper_pt_ctx *ctx, bool wait);

// (prosper_pt.cpp) which instances moved, the table into pinned staging; nothing on the GPU
int stage_transforms(prosper_pt_ctx *ctx, const prosper_ModelInstanceTransforms *transforms, uint32_t count);

} // namespace ppt
