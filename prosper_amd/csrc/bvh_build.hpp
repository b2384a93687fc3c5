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

// `triangles` are the GPU-flattened world-space triangles in (drawInstance, primitive) order.
// Throws std::runtime_error if the depth bound of the LDS traversal stack cannot be met.
BvhBuildResult build_bvh(const WorldTriangle *triangles, uint64_t count);

} // namespace ppt
