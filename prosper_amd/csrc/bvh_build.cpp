// bvh_build.cpp — binned-SAH BVH2 builder (host, multi-threaded over top-level subtrees).
#include "bvh_build.hpp"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <future>
#include <limits>
#include <stdexcept>

namespace ppt
{

namespace
{

struct Aabb
{
    float lo[3] = {std::numeric_limits<float>::infinity(), std::numeric_limits<float>::infinity(),
                   std::numeric_limits<float>::infinity()};
    float hi[3] = {-std::numeric_limits<float>::infinity(), -std::numeric_limits<float>::infinity(),
                   -std::numeric_limits<float>::infinity()};
    void grow(const float p[3])
    {
        for (int k = 0; k < 3; ++k)
        {
            lo[k] = std::min(lo[k], p[k]);
            hi[k] = std::max(hi[k], p[k]);
        }
    }
    void grow(const Aabb &b)
    {
        for (int k = 0; k < 3; ++k)
        {
            lo[k] = std::min(lo[k], b.lo[k]);
            hi[k] = std::max(hi[k], b.hi[k]);
        }
    }
    float half_area() const
    {
        const float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
        if (!(dx >= 0.0f)) return 0.0f;
        return dx * dy + dy * dz + dz * dx;
    }
};

struct Prim
{
    Aabb box;
    float centroid[3];
    uint32_t index;
};

// Temporary tree: inner nodes reference children by index into `tmp`, leaves a range of prims.
struct TmpNode
{
    Aabb box;
    int32_t left = -1, right = -1;
    uint32_t first = 0, count = 0;
};

constexpr int kBins = 16;
constexpr float kTraversalCost = 1.0f;
constexpr float kIntersectCost = 1.0f;

uint32_t ceil_log2(uint64_t n)
{
    uint32_t l = 0;
    while ((1ull << l) < n) ++l;
    return l;
}

struct Builder
{
    std::vector<Prim> &prims;
    std::vector<TmpNode> nodes;

    explicit Builder(std::vector<Prim> &p) : prims(p) {}

    // Builds the subtree over prims[first, first+count) and returns its node index in `nodes`.
    int32_t build(uint32_t first, uint32_t count, uint32_t depth)
    {
        const int32_t self = (int32_t)nodes.size();
        nodes.emplace_back();
        Aabb box, cbox;
        for (uint32_t i = 0; i < count; ++i)
        {
            box.grow(prims[first + i].box);
            cbox.grow(prims[first + i].centroid);
        }
        nodes[self].box = box;

        auto make_leaf = [&]() {
            nodes[self].first = first;
            nodes[self].count = count;
            return self;
        };
        if (count == 1) return make_leaf();

        // Keep the tree inside the LDS stack bound: once depth gets close, split by median.
        const bool forceMedian = depth + 3u + ceil_log2(count) >= kTraversalStackDepth;

        int axis = 0;
        uint32_t mid = first + count / 2;
        bool split = false;
        if (!forceMedian)
        {
            float bestCost = std::numeric_limits<float>::infinity();
            int bestAxis = -1, bestBin = -1;
            for (int a = 0; a < 3; ++a)
            {
                const float extent = cbox.hi[a] - cbox.lo[a];
                if (!(extent > 0.0f)) continue;
                Aabb binBox[kBins];
                uint32_t binCount[kBins] = {};
                const float scale = (float)kBins / extent;
                for (uint32_t i = 0; i < count; ++i)
                {
                    const Prim &p = prims[first + i];
                    int b = (int)((p.centroid[a] - cbox.lo[a]) * scale);
                    b = std::min(std::max(b, 0), kBins - 1);
                    binBox[b].grow(p.box);
                    binCount[b]++;
                }
                float rightArea[kBins];
                uint32_t rightCount[kBins];
                Aabb acc;
                uint32_t n = 0;
                for (int b = kBins - 1; b > 0; --b)
                {
                    acc.grow(binBox[b]);
                    n += binCount[b];
                    rightArea[b] = acc.half_area();
                    rightCount[b] = n;
                }
                Aabb accL;
                uint32_t nL = 0;
                for (int b = 0; b < kBins - 1; ++b)
                {
                    accL.grow(binBox[b]);
                    nL += binCount[b];
                    if (nL == 0 || rightCount[b + 1] == 0) continue;
                    const float cost = accL.half_area() * (float)nL + rightArea[b + 1] * (float)rightCount[b + 1];
                    if (cost < bestCost)
                    {
                        bestCost = cost;
                        bestAxis = a;
                        bestBin = b;
                    }
                }
            }
            if (bestAxis >= 0)
            {
                const float parentArea = std::max(box.half_area(), 1e-30f);
                const float splitCost = kTraversalCost + kIntersectCost * bestCost / parentArea;
                const float leafCost = kIntersectCost * (float)count;
                if (count <= kMaxLeafTriangles && leafCost <= splitCost) return make_leaf();
                const float extent = cbox.hi[bestAxis] - cbox.lo[bestAxis];
                const float scale = (float)kBins / extent;
                const float lo = cbox.lo[bestAxis];
                auto it = std::partition(prims.begin() + first, prims.begin() + first + count, [&](const Prim &p) {
                    int b = (int)((p.centroid[bestAxis] - lo) * scale);
                    b = std::min(std::max(b, 0), kBins - 1);
                    return b <= bestBin;
                });
                mid = (uint32_t)(it - prims.begin());
                split = mid > first && mid < first + count;
                axis = bestAxis;
            }
            else if (count <= kMaxLeafTriangles)
                return make_leaf(); // all centroids coincide
        }
        if (!split)
        {
            // median split along the widest centroid axis
            axis = 0;
            if (cbox.hi[1] - cbox.lo[1] > cbox.hi[axis] - cbox.lo[axis]) axis = 1;
            if (cbox.hi[2] - cbox.lo[2] > cbox.hi[axis] - cbox.lo[axis]) axis = 2;
            if (forceMedian && count <= kMaxLeafTriangles) return make_leaf();
            mid = first + count / 2;
            std::nth_element(
                prims.begin() + first, prims.begin() + mid, prims.begin() + first + count,
                [axis](const Prim &a, const Prim &b) { return a.centroid[axis] < b.centroid[axis]; });
        }
        const int32_t l = build(first, mid - first, depth + 1);
        const int32_t r = build(mid, first + count - mid, depth + 1);
        nodes[self].left = l;
        nodes[self].right = r;
        return self;
    }
};

void padded(const Aabb &b, float lo[3], float hi[3])
{
    for (int k = 0; k < 3; ++k)
    {
        // conservative padding: the slab test must never cull a triangle the edge-function test
        // accepts (rounding in (lo - o) * invd and in the triple products)
        const float pad = 1e-5f * (b.hi[k] - b.lo[k]) + 1e-6f * std::max(std::fabs(b.lo[k]), std::fabs(b.hi[k])) + 1e-30f;
        lo[k] = b.lo[k] - pad;
        hi[k] = b.hi[k] + pad;
    }
}

struct Emitter
{
    const std::vector<TmpNode> &tmp;
    const std::vector<Prim> &prims;
    BvhBuildResult &out;

    // reference to a temporary node as stored in a BvhNode child slot
    int32_t emit_ref(int32_t t, uint32_t depth)
    {
        const TmpNode &n = tmp[t];
        if (n.left < 0)
        {
            // leaves longer than kMaxLeafTriangles cannot occur (builder splits them)
            const uint32_t first = (uint32_t)out.permutation.size();
            for (uint32_t i = 0; i < n.count; ++i) out.permutation.push_back(prims[n.first + i].index);
            return ~(int32_t)((first << 3) | (n.count - 1));
        }
        return emit_inner(t, depth);
    }

    int32_t emit_inner(int32_t t, uint32_t depth)
    {
        const int32_t self = (int32_t)out.nodes.size();
        out.nodes.emplace_back();
        out.maxDepth = std::max(out.maxDepth, depth + 1);
        const TmpNode &n = tmp[t];
        float lo0[3], hi0[3], lo1[3], hi1[3];
        padded(tmp[n.left].box, lo0, hi0);
        padded(tmp[n.right].box, lo1, hi1);
        const int32_t c0 = emit_ref(n.left, depth + 1);
        const int32_t c1 = emit_ref(n.right, depth + 1);
        BvhNode &o = out.nodes[self];
        std::memcpy(o.lo0, lo0, 12);
        std::memcpy(o.hi0, hi0, 12);
        std::memcpy(o.lo1, lo1, 12);
        std::memcpy(o.hi1, hi1, 12);
        o.child0 = c0;
        o.child1 = c1;
        o.pad[0] = o.pad[1] = 0;
        return self;
    }
};

} // namespace

BvhBuildResult build_bvh(const WorldTriangle *triangles, uint64_t count)
{
    BvhBuildResult out;
    // An unused child slot gets the box lo = hi = +inf: both slab tests (pt_device.hpp box_entry,
    // pt_trace_stream.hpp box_entry_fma) then report an entry distance of +inf at best, which the
    // traversal treats as a miss.  (A NaN box would slip through min/max, an inverted box passes.)
    const float inf = std::numeric_limits<float>::infinity();
    auto empty_child = [&](float lo[3], float hi[3]) {
        for (int k = 0; k < 3; ++k)
        {
            lo[k] = inf;
            hi[k] = inf;
        }
    };
    if (count >= (1ull << 28)) throw std::runtime_error("too many triangles for the leaf reference encoding");

    std::vector<Prim> prims((size_t)count);
    for (uint64_t i = 0; i < count; ++i)
    {
        const WorldTriangle &t = triangles[i];
        Prim &p = prims[(size_t)i];
        p.box.grow(t.v0);
        p.box.grow(t.v1);
        p.box.grow(t.v2);
        for (int k = 0; k < 3; ++k) p.centroid[k] = 0.5f * (p.box.lo[k] + p.box.hi[k]);
        p.index = (uint32_t)i;
    }

    if (count == 0)
    {
        BvhNode root;
        std::memset(&root, 0, sizeof(root));
        empty_child(root.lo0, root.hi0);
        empty_child(root.lo1, root.hi1);
        root.child0 = ~0; // leaf reference that is never entered
        root.child1 = ~0;
        out.nodes.push_back(root);
        out.maxDepth = 1;
        return out;
    }

    Builder builder(prims);
    builder.nodes.reserve((size_t)count * 2);
    const int32_t rootTmp = builder.build(0, (uint32_t)count, 0);
    out.nodes.reserve(builder.nodes.size());
    out.permutation.reserve((size_t)count);
    Emitter emitter{builder.nodes, prims, out};
    if (builder.nodes[rootTmp].left < 0)
    {
        // the whole scene is one leaf: wrap it in a root whose second child is empty
        out.nodes.emplace_back();
        float lo[3], hi[3];
        padded(builder.nodes[rootTmp].box, lo, hi);
        const int32_t ref = emitter.emit_ref(rootTmp, 1);
        BvhNode &root = out.nodes[0];
        std::memset(&root, 0, sizeof(root));
        std::memcpy(root.lo0, lo, 12);
        std::memcpy(root.hi0, hi, 12);
        empty_child(root.lo1, root.hi1);
        root.child0 = ref;
        root.child1 = ~0;
        out.maxDepth = 1;
    }
    else
        emitter.emit_inner(rootTmp, 0);

    if (out.maxDepth > kTraversalStackDepth)
        throw std::runtime_error("BVH depth exceeds the traversal stack bound");
    return out;
}

} // namespace ppt
