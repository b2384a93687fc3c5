// sbvh_experiment.cpp — would spatial splits pay on this scene?  (round-2 verdict item 5a; tooling, not product)
//
// Reads world-space triangles (9 floats each) and builds two binary hierarchies with the same binned surface-area
// heuristic: (a) object splits only - what bvh_build.cpp does - and (b) with spatial splits of the references (Stich et al.
// 2009), under the one constraint the hit contract allows (DESIGN.md "hit contract"): a split reference's box is
// (the triangle's box) ∩ (the split cell) - the union of a triangle's reference boxes must still contain its guard box, so
// a reference is never shrunk to the clipped triangle.  Prints the surface-area cost of both (expected node visits and
// leaf tests of a random ray), node and reference counts.
//   g++ -O3 -std=c++17 scripts/sbvh_experiment.cpp -o /tmp/sbvh_experiment && /tmp/sbvh_experiment triangles.bin
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <limits>
#include <vector>

struct Box
{
    float lo[3] = {1e30f, 1e30f, 1e30f}, hi[3] = {-1e30f, -1e30f, -1e30f};
    void grow(const Box &b)
    {
        for (int k = 0; k < 3; ++k)
        {
            lo[k] = std::min(lo[k], b.lo[k]);
            hi[k] = std::max(hi[k], b.hi[k]);
        }
    }
    float area() const
    {
        const float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
        if (dx < 0 || dy < 0 || dz < 0) return 0.0f;
        return dx * dy + dy * dz + dz * dx;
    }
};
struct Ref
{
    Box box;
    unsigned tri;
};

static const int kBins = 16;
static bool gSpatial = false;
static double gInnerArea = 0, gLeafAreaCount = 0;
static size_t gInner = 0, gLeaves = 0, gLeafRefs = 0, gSpatialSplits = 0;
static float gRootArea = 1;

static void build(std::vector<Ref> &refs, int depth)
{
    Box box, cbox;
    for (const Ref &r : refs)
    {
        box.grow(r.box);
        for (int k = 0; k < 3; ++k)
        {
            const float c = 0.5f * (r.box.lo[k] + r.box.hi[k]);
            cbox.lo[k] = std::min(cbox.lo[k], c);
            cbox.hi[k] = std::max(cbox.hi[k], c);
        }
    }
    const size_t n = refs.size();
    auto leaf = [&]() {
        gLeaves++;
        gLeafRefs += n;
        gLeafAreaCount += (double)box.area() * (double)n;
    };
    if (n <= 1 || depth > 60) return leaf();
    // object split (binned on centroids)
    float bestObj = std::numeric_limits<float>::infinity();
    int objAxis = -1, objBin = -1;
    for (int a = 0; a < 3; ++a)
    {
        const float ext = cbox.hi[a] - cbox.lo[a];
        if (!(ext > 0)) continue;
        Box bb[kBins];
        size_t bc[kBins] = {};
        for (const Ref &r : refs)
        {
            int b = (int)((0.5f * (r.box.lo[a] + r.box.hi[a]) - cbox.lo[a]) / ext * kBins);
            b = std::min(std::max(b, 0), kBins - 1);
            bb[b].grow(r.box);
            bc[b]++;
        }
        float ra[kBins];
        size_t rc[kBins];
        Box acc;
        size_t c = 0;
        for (int b = kBins - 1; b > 0; --b)
        {
            acc.grow(bb[b]);
            c += bc[b];
            ra[b] = acc.area();
            rc[b] = c;
        }
        Box accL;
        size_t cl = 0;
        for (int b = 0; b < kBins - 1; ++b)
        {
            accL.grow(bb[b]);
            cl += bc[b];
            if (!cl || !rc[b + 1]) continue;
            const float cost = accL.area() * cl + ra[b + 1] * rc[b + 1];
            if (cost < bestObj)
            {
                bestObj = cost;
                objAxis = a;
                objBin = b;
            }
        }
    }
    // spatial split (binned on the node's extent; references chopped to the bins they overlap)
    float bestSp = std::numeric_limits<float>::infinity();
    int spAxis = -1, spBin = -1;
    if (gSpatial && n > 2)
        for (int a = 0; a < 3; ++a)
        {
            const float ext = box.hi[a] - box.lo[a];
            if (!(ext > 0)) continue;
            Box bb[kBins];
            size_t enter[kBins] = {}, leave[kBins] = {};
            const float w = ext / kBins;
            for (const Ref &r : refs)
            {
                int b0 = std::min(std::max((int)((r.box.lo[a] - box.lo[a]) / w), 0), kBins - 1);
                int b1 = std::min(std::max((int)((r.box.hi[a] - box.lo[a]) / w), 0), kBins - 1);
                enter[b0]++;
                leave[b1]++;
                for (int b = b0; b <= b1; ++b)
                {
                    Box piece = r.box; // (triangle box) ∩ (bin slab): what the hit contract allows
                    piece.lo[a] = std::max(piece.lo[a], box.lo[a] + w * b);
                    piece.hi[a] = std::min(piece.hi[a], box.lo[a] + w * (b + 1));
                    bb[b].grow(piece);
                }
            }
            float ra[kBins];
            size_t rc[kBins];
            Box acc;
            size_t c = 0;
            for (int b = kBins - 1; b > 0; --b)
            {
                acc.grow(bb[b]);
                c += leave[b];
                ra[b] = acc.area();
                rc[b] = c;
            }
            Box accL;
            size_t cl = 0;
            for (int b = 0; b < kBins - 1; ++b)
            {
                accL.grow(bb[b]);
                cl += enter[b];
                if (!cl || !rc[b + 1]) continue;
                const float cost = accL.area() * cl + ra[b + 1] * rc[b + 1];
                if (cost < bestSp)
                {
                    bestSp = cost;
                    spAxis = a;
                    spBin = b;
                }
            }
        }
    const float parent = std::max(box.area(), 1e-30f);
    const float best = std::min(bestObj, bestSp);
    if (!(best < std::numeric_limits<float>::infinity()) || (n <= 4 && (float)n <= 1.0f + best / parent)) return leaf();
    std::vector<Ref> L, R;
    if (bestSp < bestObj)
    {
        gSpatialSplits++;
        const float w = (box.hi[spAxis] - box.lo[spAxis]) / kBins;
        const float plane = box.lo[spAxis] + w * (spBin + 1);
        for (const Ref &r : refs)
        {
            if (r.box.hi[spAxis] <= plane)
                L.push_back(r);
            else if (r.box.lo[spAxis] >= plane)
                R.push_back(r);
            else
            {
                Ref l = r, rr = r;
                l.box.hi[spAxis] = plane;
                rr.box.lo[spAxis] = plane;
                L.push_back(l);
                R.push_back(rr);
            }
        }
        if (L.size() == n || R.size() == n) return leaf(); // no progress (all references straddle)
    }
    else
    {
        const float ext = cbox.hi[objAxis] - cbox.lo[objAxis];
        for (const Ref &r : refs)
        {
            int b = (int)((0.5f * (r.box.lo[objAxis] + r.box.hi[objAxis]) - cbox.lo[objAxis]) / ext * kBins);
            b = std::min(std::max(b, 0), kBins - 1);
            (b <= objBin ? L : R).push_back(r);
        }
        if (L.empty() || R.empty()) return leaf();
    }
    gInner++;
    gInnerArea += box.area();
    std::vector<Ref>().swap(refs);
    build(L, depth + 1);
    build(R, depth + 1);
}

int main(int argc, char **argv)
{
    if (argc < 2) return 1;
    FILE *f = std::fopen(argv[1], "rb");
    if (!f) return 1;
    std::fseek(f, 0, SEEK_END);
    const size_t count = (size_t)std::ftell(f) / 36;
    std::fseek(f, 0, SEEK_SET);
    std::vector<float> v(count * 9);
    if (std::fread(v.data(), 36, count, f) != count) return 1;
    std::fclose(f);
    for (int pass = 0; pass < 2; ++pass)
    {
        gSpatial = pass == 1;
        gInnerArea = gLeafAreaCount = 0;
        gInner = gLeaves = gLeafRefs = gSpatialSplits = 0;
        std::vector<Ref> refs(count);
        Box root;
        for (size_t i = 0; i < count; ++i)
        {
            for (int c = 0; c < 3; ++c)
            {
                Box p;
                for (int k = 0; k < 3; ++k) p.lo[k] = p.hi[k] = v[i * 9 + c * 3 + k];
                refs[i].box.grow(p);
            }
            refs[i].tri = (unsigned)i;
            root.grow(refs[i].box);
        }
        gRootArea = root.area();
        build(refs, 0);
        std::printf("%-26s inner nodes %8zu  leaves %8zu  references %8zu (x%.3f)  spatial splits %7zu  "
                    "expected node visits %.2f  expected triangle tests %.2f\n",
                    pass ? "with spatial splits:" : "object splits only:", gInner, gLeaves, gLeafRefs, (double)gLeafRefs / count,
                    gSpatialSplits, gInnerArea / gRootArea, gLeafAreaCount / gRootArea);
    }
    return 0;
}
