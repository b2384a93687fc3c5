// bvh_build.hpp — host-side builder of the flattened BVH the HIP traversal walks.
//
// prosper has no BVH code: it hands triangles to vkCmdBuildAccelerationStructuresKHR
// (src/scene/World.cpp:740,798, ePreferFastTrace :671).  This is the product's replacement: a
// binned-SAH BVH2 over the world-space triangles, emitted in depth-first order as 64-byte nodes
// that hold both children's boxes (pt_scene.hpp).
#pragma once

#include <cstdint>
#include <vector>

#include "pt_scene.hpp"

namespace ppt
{

struct BvhBuildResult
{
    std::vector<BvhNode> nodes;        // nodes[0] is the root and always an inner node
    std::vector<uint32_t> permutation; // leaf-order position -> input triangle index
    uint32_t maxDepth = 0;             // inner nodes on the longest root-to-leaf path
};

// Tuning and test knobs of the builder (prosper_pt_debug_options carries them; zero / negative = the default).  None of
// them changes a pixel: hits do not depend on the hierarchy (DESIGN.md "hit contract").
struct BvhBuildOptions
{
    float sahTraversalCost = 0.0f; // SAH cost of a node visit relative to a triangle test (default 1)
    float boxPad = 0.0f;           // box padding coefficient (default and minimum 1.6e-5)
    uint32_t leafSize = 0;         // most triangles per leaf, 1..8 (default kMaxLeafTriangles)
    uint32_t buildThreads = 0;     // host threads (default: hardware concurrency, at most 32 and at most the container's CPU quota)
    uint32_t topEntries = 0;       // entries of the re-braided top level (default: one per four triangles, 1 k .. 64 k)
    int32_t nodeOrder = -1;        // 0 depth-first as emitted, 1 breadth-first, 2 first 4096 breadth-first (default)
    int32_t childOrder = -1;       // 0: children in build order instead of smallest box first
    uint32_t buildTiming = 0;      // stage times of the assembly to stderr
};

// the host threads a build takes by default (what BvhBuildOptions::buildThreads = 0 means)
unsigned default_build_threads();

// box padding coefficient of the emitter (1.6e-5, or BvhBuildOptions::boxPad): the device refit pads with the same value
float bvh_pad_coefficient(const BvhBuildOptions &opt);

// `triangles` are the GPU-flattened world-space triangles in (drawInstance, primitive) order.
// Throws std::runtime_error if the depth bound of the LDS traversal stack cannot be met.
BvhBuildResult build_bvh(const WorldTriangle *triangles, uint64_t count, const BvhBuildOptions &opt = BvhBuildOptions());

// The same hierarchy built the way prosper builds its acceleration structures (src/scene/World.cpp:585-802: one BLAS
// per mesh, a TLAS over the instances rebuilt every frame): one SAH subtree per model INSTANCE over that instance's
// world-space triangles, and a top level over the instances' boxes whose leaves are those subtrees - spliced into
// ONE 4-wide tree, so the traversal kernels do not know the difference and every box test stays world-space
// arithmetic (hit contract).  The subtrees are kept: after an instance moved, rebuild() re-splits only that
// instance's triangles and the (tiny) top level, then re-emits the nodes - a moved instance costs its own triangles,
// not the scene's.  Subtrees build in parallel on the host's threads.
class InstancedBvh
{
  public:
    struct Range
    {
        uint32_t first; // first world triangle of the instance (in (drawInstance, primitive) order)
        uint32_t count;
    };
    InstancedBvh();
    ~InstancedBvh();
    InstancedBvh(const InstancedBvh &) = delete;
    InstancedBvh &operator=(const InstancedBvh &) = delete;
    // Full build: every instance's subtree, the top level, the emitted nodes.
    BvhBuildResult build(
        const WorldTriangle *triangles, uint64_t count, const std::vector<Range> &instances,
        const BvhBuildOptions &opt = BvhBuildOptions());
    // `triangles` again holds ALL world triangles; only the subtrees of the instances flagged in `changed`
    // (one flag per entry of the `instances` given to build) are rebuilt.
    BvhBuildResult rebuild(
        const WorldTriangle *triangles, const std::vector<uint8_t> &changed, const BvhBuildOptions &opt = BvhBuildOptions());
    // The instances' triangle ranges changed (meshes arrived: prosper_pt_update_meshes): `instances` has one entry per entry
    // of the last build, `changed` flags those whose triangles are new or different - only they are split again; the other
    // subtrees are kept, moved to their new place in the triangle array.  An unflagged instance must hold the triangles it
    // held (same count, same values).
    BvhBuildResult adopt(
        const WorldTriangle *triangles, uint64_t count, const std::vector<Range> &instances, const std::vector<uint8_t> &changed,
        const BvhBuildOptions &opt = BvhBuildOptions());
    [[nodiscard]] size_t instanceCount() const;
    void swap(InstancedBvh &other) noexcept
    {
        Impl *t = m;
        m = other.m;
        other.m = t;
    }

  private:
    struct Impl;
    Impl *m;
};

} // namespace ppt
