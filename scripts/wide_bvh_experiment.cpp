// wide_bvh_experiment.cpp — would an 8-wide node pay?  (round-2 verdict item 5b; tooling, not product, CPU only)
//
// Reads world-space triangles (9 floats each), builds ONE binary hierarchy with a 16-bin surface-area heuristic (what
// bvh_build.cpp does), collapses it to k-wide nodes for k = 4 and k = 8 by the same rule (open the child with the
// largest box until k children or only leaves are left), and walks both with the same rays:
//   * "bounce" rays: a point on an area-weighted random triangle, a cosine-distributed direction about its normal,
//     closest hit wanted - what wf_trace gets after the first bounce;
//   * "shadow" rays: from such a point towards a random point of the scene box, any hit ends the ray.
// Children of a node are visited nearest first, either by their entry distances (a sort per visit: 5 comparators for
// four children, 19 for eight) or in a fixed order per ray octant (children ranked by their box centre along the
// octant's diagonal: no sort, what compressed wide BVHs do).  Printed per k and order: node visits, children tested,
// triangle tests and stack pushes per ray.  The instruction side of the comparison is scripts/unit_costs.hip
// (unit_node_visit8_*): profiles/r03_wide_bvh_experiment.txt multiplies the two.
//   g++ -O3 -std=c++17 scripts/wide_bvh_experiment.cpp -o /tmp/wide_bvh_experiment && /tmp/wide_bvh_experiment tris.bin
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <limits>
#include <numeric>
#include <string>
#include <vector>

namespace
{
struct V3
{
    double x, y, z;
};
V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
V3 operator*(V3 a, double s) { return {a.x * s, a.y * s, a.z * s}; }
double dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
V3 cross(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
double comp(V3 a, int k) { return k == 0 ? a.x : k == 1 ? a.y : a.z; }

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
        return dx * dy + dy * dz + dz * dx;
    }
};

constexpr int kBins = 16;
constexpr int kMaxLeaf = 4;
bool gFullDepth = false; // binary tree down to single triangles (the optimal collapse forms the leaves itself)

struct Node2
{
    Box box;
    int left = -1, right = -1; // inner
    int first = 0, count = 0;  // leaf: range of `order`
};
std::vector<Node2> gNodes;
std::vector<Box> gTriBox;
std::vector<unsigned> gOrder;

int build(int first, int count)
{
    const int self = (int)gNodes.size();
    gNodes.emplace_back();
    Box box, cbox;
    for (int i = first; i < first + count; ++i)
    {
        const Box &b = gTriBox[gOrder[i]];
        box.grow(b);
        for (int k = 0; k < 3; ++k)
        {
            const float c = 0.5f * (b.lo[k] + b.hi[k]);
            cbox.lo[k] = std::min(cbox.lo[k], c);
            cbox.hi[k] = std::max(cbox.hi[k], c);
        }
    }
    gNodes[self].box = box;
    auto leaf = [&]() {
        gNodes[self].first = first;
        gNodes[self].count = count;
        return self;
    };
    if (count <= 1) return leaf();
    float best = std::numeric_limits<float>::infinity();
    int axis = -1, bin = -1;
    for (int a = 0; a < 3; ++a)
    {
        const float ext = cbox.hi[a] - cbox.lo[a];
        if (!(ext > 0)) continue;
        Box bb[kBins];
        int bc[kBins] = {};
        for (int i = first; i < first + count; ++i)
        {
            const Box &r = gTriBox[gOrder[i]];
            int b = (int)((0.5f * (r.lo[a] + r.hi[a]) - cbox.lo[a]) / ext * kBins);
            b = std::min(std::max(b, 0), kBins - 1);
            bb[b].grow(r);
            bc[b]++;
        }
        float ra[kBins];
        int rc[kBins];
        Box acc;
        int c = 0;
        for (int b = kBins - 1; b > 0; --b)
        {
            acc.grow(bb[b]);
            c += bc[b];
            ra[b] = acc.area();
            rc[b] = c;
        }
        Box accL;
        int cl = 0;
        for (int b = 0; b < kBins - 1; ++b)
        {
            accL.grow(bb[b]);
            cl += bc[b];
            if (!cl || !rc[b + 1]) continue;
            const float cost = accL.area() * cl + ra[b + 1] * rc[b + 1];
            if (cost < best)
            {
                best = cost;
                axis = a;
                bin = b;
            }
        }
    }
    const float parent = std::max(box.area(), 1e-30f);
    if (axis < 0)
    {
        if (count <= kMaxLeaf) return leaf();
        // identical centroids: halve
        const int l = build(first, count / 2), r = build(first + count / 2, count - count / 2);
        gNodes[self].left = l;
        gNodes[self].right = r;
        return self;
    }
    if (!gFullDepth && count <= kMaxLeaf && (float)count <= 1.0f + best / parent) return leaf();
    const float ext = cbox.hi[axis] - cbox.lo[axis];
    const auto mid = std::partition(gOrder.begin() + first, gOrder.begin() + first + count, [&](unsigned t) {
        const Box &r = gTriBox[t];
        int b = (int)((0.5f * (r.lo[axis] + r.hi[axis]) - cbox.lo[axis]) / ext * kBins);
        b = std::min(std::max(b, 0), kBins - 1);
        return b <= bin;
    });
    const int nl = (int)(mid - (gOrder.begin() + first));
    if (nl == 0 || nl == count) return count <= kMaxLeaf ? leaf() : (gNodes[self].left = build(first, count / 2), gNodes[self].right = build(first + count / 2, count - count / 2), self);
    const int l = build(first, nl);
    const int r = build(first + nl, count - nl);
    gNodes[self].left = l;
    gNodes[self].right = r;
    return self;
}

// k-wide nodes: child >= 0 is a wide node, child < 0 is ~(binary leaf index)
struct WideNode
{
    int n = 0;
    int child[8];
    Box box[8];
    uint8_t octantOrder[8][8]; // per ray octant: the children, nearest first along the octant's diagonal
};

// WIDE_ORDER=axis: what a traversal kernel can afford without a sort or a per-octant table - the children kept sorted by
// box centre along ONE axis per node (the one their centres spread most along), walked in storage order by a ray that
// travels in + along it and in reverse by the others (one compare + four selects instead of a five-comparator sort)
bool gAxisOrder = false;
void axis_order(WideNode &w)
{
    int axis = 0;
    float bestSpread = -1.0f;
    for (int a = 0; a < 3; ++a)
    {
        float lo = 1e30f, hi = -1e30f;
        for (int i = 0; i < w.n; ++i)
        {
            const float c = 0.5f * (w.box[i].lo[a] + w.box[i].hi[a]);
            lo = std::min(lo, c);
            hi = std::max(hi, c);
        }
        if (hi - lo > bestSpread)
        {
            bestSpread = hi - lo;
            axis = a;
        }
    }
    uint8_t idx[8] = {0, 1, 2, 3, 4, 5, 6, 7};
    std::sort(idx, idx + w.n, [&](uint8_t p, uint8_t q) {
        return w.box[p].lo[axis] + w.box[p].hi[axis] < w.box[q].lo[axis] + w.box[q].hi[axis];
    });
    for (int oct = 0; oct < 8; ++oct)
        for (int i = 0; i < w.n; ++i) w.octantOrder[oct][i] = ((oct >> axis) & 1) ? idx[w.n - 1 - i] : idx[i];
}

int collapse(int k, int root2, std::vector<WideNode> &out)
{
    const int self = (int)out.size();
    out.emplace_back();
    std::vector<int> kids = {gNodes[root2].left, gNodes[root2].right};
    while ((int)kids.size() < k)
    {
        int pick = -1;
        float bestArea = -1.0f;
        for (int i = 0; i < (int)kids.size(); ++i)
            if (gNodes[kids[i]].left >= 0 && gNodes[kids[i]].box.area() > bestArea)
            {
                bestArea = gNodes[kids[i]].box.area();
                pick = i;
            }
        if (pick < 0) break;
        const int open = kids[pick];
        kids[pick] = gNodes[open].left;
        kids.push_back(gNodes[open].right);
    }
    WideNode w;
    w.n = (int)kids.size();
    for (int i = 0; i < w.n; ++i) w.box[i] = gNodes[kids[i]].box;
    for (int oct = 0; oct < 8; ++oct)
    {
        float key[8];
        for (int i = 0; i < w.n; ++i)
        {
            key[i] = 0.0f;
            for (int a = 0; a < 3; ++a)
            {
                const float c = 0.5f * (w.box[i].lo[a] + w.box[i].hi[a]);
                key[i] += ((oct >> a) & 1) ? -c : c; // bit a set: the ray travels in -a
            }
        }
        uint8_t idx[8] = {0, 1, 2, 3, 4, 5, 6, 7};
        std::sort(idx, idx + w.n, [&](uint8_t p, uint8_t q) { return key[p] < key[q]; });
        for (int i = 0; i < 8; ++i) w.octantOrder[oct][i] = idx[i];
    }
    if (gAxisOrder) axis_order(w);
    out[self] = w;
    for (int i = 0; i < (int)kids.size(); ++i)
    {
        const int c = gNodes[kids[i]].left >= 0 ? collapse(k, kids[i], out) : ~kids[i];
        out[self].child[i] = c;
    }
    return self;
}


// ---- the optimal collapse (Ylitie, Karras, Laine 2017, section 3.1): dynamic programming over the binary tree ----
// cost[n][i-1] = the least surface-area cost of representing n's subtree as i children of some wide node; a subtree of at
// most kMaxLeaf triangles may also become ONE leaf.  gNodeCost / gTriCost price a node visit and a triangle test.
double gNodeCost = 1.0, gTriCost = 0.6;
struct Dp
{
    double cost[7];
    uint8_t leaf = 0;      // as one child: a leaf of all its triangles
    uint8_t split[9] = {}; // split[j] = roots given to the left child when the two children share j roots (j = 2 .. k)
    uint8_t lower[7] = {}; // lower[i-1] = 1: cost[i-1] is cost[i-2] (fewer roots are as good)
    int tris = 0;
};
std::vector<Dp> gDp;

void dp_pass(int k)
{
    gDp.assign(gNodes.size(), Dp());
    for (int n = (int)gNodes.size() - 1; n >= 0; --n) // children have larger indices than their parent
    {
        const Node2 &node = gNodes[n];
        Dp &d = gDp[n];
        const double area = node.box.area();
        if (node.left < 0)
        {
            d.tris = node.count;
            d.leaf = 1;
            for (int i = 0; i < k - 1; ++i) d.cost[i] = area * gTriCost * node.count;
            continue;
        }
        const Dp &l = gDp[node.left], &r = gDp[node.right];
        d.tris = l.tris + r.tris;
        double dist[9];
        for (int j = 2; j <= k; ++j)
        {
            dist[j] = std::numeric_limits<double>::infinity();
            for (int i = 1; i < j; ++i)
            {
                if (i > k - 1 || j - i > k - 1) continue;
                const double c = l.cost[i - 1] + r.cost[j - i - 1];
                if (c < dist[j])
                {
                    dist[j] = c;
                    d.split[j] = (uint8_t)i;
                }
            }
        }
        const double inner = area * gNodeCost + dist[k];
        const double asLeaf = d.tris <= kMaxLeaf ? area * gTriCost * d.tris : std::numeric_limits<double>::infinity();
        d.leaf = asLeaf <= inner;
        d.cost[0] = std::min(asLeaf, inner);
        for (int i = 2; i <= k - 1; ++i)
        {
            d.lower[i - 1] = d.cost[i - 2] <= dist[i];
            d.cost[i - 1] = std::min(dist[i], d.cost[i - 2]);
        }
    }
}

// leaves formed by the collapse: triangle ranges of gOrder (a subtree's triangles are contiguous)
struct DpLeaf
{
    int first, count;
};
std::vector<DpLeaf> gDpLeaves;
std::vector<Box> gDpLeafBox;

void dp_roots(int n, int i, std::vector<int> &out) // the (at most i) subtrees that stand for n in its parent's child list
{
    if (i == 1 || gNodes[n].left < 0)
    {
        out.push_back(n);
        return;
    }
    if (gDp[n].lower[i - 1]) return dp_roots(n, i - 1, out);
    const int li = gDp[n].split[i];
    dp_roots(gNodes[n].left, li, out);
    dp_roots(gNodes[n].right, i - li, out);
}

int subtree_first(int n)
{
    while (gNodes[n].left >= 0) n = gNodes[n].left;
    return gNodes[n].first;
}

int dp_collapse(int k, int n, std::vector<WideNode> &out)
{
    const int self = (int)out.size();
    out.emplace_back();
    std::vector<int> kids;
    const int li = gDp[n].split[k];
    dp_roots(gNodes[n].left, li, kids);
    dp_roots(gNodes[n].right, k - li, kids);
    WideNode w;
    w.n = (int)kids.size();
    for (int i = 0; i < w.n; ++i) w.box[i] = gNodes[kids[i]].box;
    for (int oct = 0; oct < 8; ++oct)
    {
        float key[8];
        for (int i = 0; i < w.n; ++i)
        {
            key[i] = 0.0f;
            for (int a = 0; a < 3; ++a)
            {
                const float c = 0.5f * (w.box[i].lo[a] + w.box[i].hi[a]);
                key[i] += ((oct >> a) & 1) ? -c : c;
            }
        }
        uint8_t idx[8] = {0, 1, 2, 3, 4, 5, 6, 7};
        std::sort(idx, idx + w.n, [&](uint8_t p, uint8_t q) { return key[p] < key[q]; });
        for (int i = 0; i < 8; ++i) w.octantOrder[oct][i] = idx[i];
    }
    if (gAxisOrder) axis_order(w);
    out[self] = w;
    for (int i = 0; i < (int)kids.size(); ++i)
    {
        int c;
        if (gDp[kids[i]].leaf || gNodes[kids[i]].left < 0)
        {
            // a leaf of the whole subtree: appended to the binary node list as a new leaf record
            Node2 leaf;
            leaf.box = gNodes[kids[i]].box;
            leaf.first = subtree_first(kids[i]);
            leaf.count = gDp[kids[i]].tris;
            gNodes.push_back(leaf);
            c = ~(int)(gNodes.size() - 1);
        }
        else
            c = dp_collapse(k, kids[i], out);
        out[self].child[i] = c;
    }
    return self;
}

std::vector<float> gTris;
struct Ray
{
    V3 o, d;
    double tMin, tMax;
    bool any;
};
struct Tally
{
    double nodeVisits = 0, childTests = 0, triTests = 0, pushes = 0, hits = 0, maxStack = 0;
};

bool hit_triangle(const Ray &r, unsigned t, double &tOut)
{
    const float *p = gTris.data() + (size_t)t * 9;
    const V3 a{p[0], p[1], p[2]}, b{p[3], p[4], p[5]}, c{p[6], p[7], p[8]};
    const V3 e1 = b - a, e2 = c - a, pv = cross(r.d, e2);
    const double det = dot(e1, pv);
    if (std::fabs(det) < 1e-300) return false;
    const double inv = 1.0 / det;
    const V3 tv = r.o - a;
    const double u = dot(tv, pv) * inv;
    if (u < 0 || u > 1) return false;
    const V3 qv = cross(tv, e1);
    const double v = dot(r.d, qv) * inv;
    if (v < 0 || u + v > 1) return false;
    const double t2 = dot(e2, qv) * inv;
    if (t2 <= r.tMin || t2 >= tOut) return false;
    tOut = t2;
    return true;
}

void walk(const std::vector<WideNode> &nodes, const Ray &r, bool sorted, Tally &t)
{
    const V3 inv{1.0 / r.d.x, 1.0 / r.d.y, 1.0 / r.d.z};
    const int oct = (r.d.x < 0 ? 1 : 0) | (r.d.y < 0 ? 2 : 0) | (r.d.z < 0 ? 4 : 0);
    double tMax = r.tMax;
    std::vector<int> stack;
    stack.push_back(0);
    size_t deepest = 1;
    bool found = false;
    while (!stack.empty())
    {
        const int ref = stack.back();
        stack.pop_back();
        if (ref < 0)
        {
            const Node2 &leaf = gNodes[~ref];
            for (int i = 0; i < leaf.count; ++i)
            {
                t.triTests++;
                if (hit_triangle(r, gOrder[leaf.first + i], tMax)) found = true;
            }
            if (found && r.any) break;
            continue;
        }
        const WideNode &n = nodes[ref];
        t.nodeVisits++;
        t.childTests += n.n;
        double entry[8];
        bool in[8];
        for (int i = 0; i < n.n; ++i)
        {
            double tn = r.tMin, tf = tMax;
            for (int a = 0; a < 3; ++a)
            {
                const double o = comp(r.o, a), iv = comp(inv, a);
                double t0 = (n.box[i].lo[a] - o) * iv, t1 = (n.box[i].hi[a] - o) * iv;
                if (t0 > t1) std::swap(t0, t1);
                tn = std::max(tn, t0);
                tf = std::min(tf, t1);
            }
            in[i] = tn <= tf;
            entry[i] = tn;
        }
        uint8_t order[8];
        int m = 0;
        if (sorted && !r.any)
        {
            for (int i = 0; i < n.n; ++i)
                if (in[i]) order[m++] = (uint8_t)i;
            std::sort(order, order + m, [&](uint8_t p, uint8_t q) { return entry[p] < entry[q]; });
        }
        else
        {
            for (int i = 0; i < n.n; ++i)
                if (in[n.octantOrder[oct][i]]) order[m++] = n.octantOrder[oct][i];
        }
        for (int i = m - 1; i >= 0; --i) stack.push_back(n.child[order[i]]); // nearest on top
        if (m > 1) t.pushes += m - 1;                                         // the nearest is walked directly
        deepest = std::max(deepest, stack.size());
    }
    t.hits += found ? 1 : 0;
    t.maxStack = std::max(t.maxStack, (double)deepest);
}

uint64_t gState = 0x9E3779B97F4A7C15ull;
double rnd()
{
    gState ^= gState << 13;
    gState ^= gState >> 7;
    gState ^= gState << 17;
    return (double)(gState >> 11) * (1.0 / 9007199254740992.0);
}
} // namespace

int main(int argc, char **argv)
{
    if (argc < 2) return 1;
    FILE *f = std::fopen(argv[1], "rb");
    if (!f) return 1;
    std::fseek(f, 0, SEEK_END);
    const size_t count = (size_t)std::ftell(f) / 36;
    std::fseek(f, 0, SEEK_SET);
    gTris.resize(count * 9);
    if (std::fread(gTris.data(), 36, count, f) != count) return 1;
    std::fclose(f);
    const int nRays = argc > 2 ? std::atoi(argv[2]) : 200000;

    gTriBox.resize(count);
    gOrder.resize(count);
    std::iota(gOrder.begin(), gOrder.end(), 0u);
    Box scene;
    std::vector<double> cdf(count);
    double total = 0;
    for (size_t i = 0; i < count; ++i)
    {
        const float *p = gTris.data() + i * 9;
        for (int c = 0; c < 3; ++c)
            for (int k = 0; k < 3; ++k)
            {
                gTriBox[i].lo[k] = std::min(gTriBox[i].lo[k], p[c * 3 + k]);
                gTriBox[i].hi[k] = std::max(gTriBox[i].hi[k], p[c * 3 + k]);
            }
        scene.grow(gTriBox[i]);
        const V3 a{p[0], p[1], p[2]}, b{p[3], p[4], p[5]}, c{p[6], p[7], p[8]};
        const V3 n = cross(b - a, c - a);
        total += 0.5 * std::sqrt(dot(n, n));
        cdf[i] = total;
    }
    // rays
    std::vector<Ray> rays;
    for (int i = 0; i < nRays; ++i)
    {
        const size_t t = (size_t)(std::lower_bound(cdf.begin(), cdf.end(), rnd() * total) - cdf.begin());
        const float *p = gTris.data() + std::min(t, count - 1) * 9;
        const V3 a{p[0], p[1], p[2]}, b{p[3], p[4], p[5]}, c{p[6], p[7], p[8]};
        double u = rnd(), v = rnd();
        if (u + v > 1)
        {
            u = 1 - u;
            v = 1 - v;
        }
        V3 n = cross(b - a, c - a);
        const double len = std::sqrt(dot(n, n));
        if (!(len > 0)) continue;
        n = n * (1.0 / len);
        if (rnd() < 0.5) n = n * -1.0;
        const V3 o = a + (b - a) * u + (c - a) * v + n * 1e-4;
        Ray r;
        r.o = o;
        r.tMin = 1e-5;
        if (i & 1)
        {
            // shadow ray towards a random point of the scene box
            const V3 target{scene.lo[0] + rnd() * (scene.hi[0] - scene.lo[0]), scene.lo[1] + rnd() * (scene.hi[1] - scene.lo[1]),
                            scene.lo[2] + rnd() * (scene.hi[2] - scene.lo[2])};
            V3 d = target - o;
            const double dist = std::sqrt(dot(d, d));
            if (!(dist > 1e-3) || dot(d, n) <= 0) continue;
            r.d = d * (1.0 / dist);
            r.tMax = dist;
            r.any = true;
        }
        else
        {
            // cosine-distributed about n
            const double r1 = rnd(), r2 = rnd(), phi = 6.283185307179586 * r1, s = std::sqrt(r2);
            const V3 h = std::fabs(n.x) < 0.9 ? V3{1, 0, 0} : V3{0, 1, 0};
            V3 tx = cross(h, n);
            tx = tx * (1.0 / std::sqrt(dot(tx, tx)));
            const V3 ty = cross(n, tx);
            r.d = tx * (s * std::cos(phi)) + ty * (s * std::sin(phi)) + n * std::sqrt(1.0 - r2);
            r.tMax = 1e30;
            r.any = false;
        }
        if (r.d.x == 0 || r.d.y == 0 || r.d.z == 0) continue;
        rays.push_back(r);
    }

    if (const char *v = std::getenv("WIDE_TRI_COST")) gTriCost = std::atof(v);
    if (const char *v = std::getenv("WIDE_ORDER")) gAxisOrder = std::string(v) == "axis";
    for (int optimal = 0; optimal < 2; ++optimal)
    {
        gFullDepth = optimal != 0 && !std::getenv("WIDE_KEEP_LEAVES"); // WIDE_KEEP_LEAVES: the optimal collapse over the SAH-terminated tree
        for (int k : {4, 8})
        {
            gNodes.clear();
            gNodes.reserve(count * 3);
            std::iota(gOrder.begin(), gOrder.end(), 0u);
            build(0, (int)count);
            std::vector<WideNode> wide;
            if (optimal)
            {
                dp_pass(k);
                dp_collapse(k, 0, wide);
            }
            else
                collapse(k, 0, wide);
            double fill = 0, leafRefs = 0, leafTris = 0;
            for (const WideNode &w : wide)
            {
                fill += w.n;
                for (int i = 0; i < w.n; ++i)
                    if (w.child[i] < 0)
                    {
                        leafRefs++;
                        leafTris += gNodes[~w.child[i]].count;
                    }
            }
            std::printf("%d-wide, %s: %zu nodes, %.2f children per node, %.0f leaves of %.2f triangles\n", k,
                        optimal ? "optimal collapse of the full-depth binary tree (node cost 1, triangle cost from WIDE_TRI_COST, default 0.6)" : "greedy collapse (largest box first) of the SAH-terminated binary tree",
                        wide.size(), fill / (double)wide.size(), leafRefs, leafTris / leafRefs);
            for (int sorted = 1; sorted >= 0; --sorted)
                for (int any = 0; any < 2; ++any)
                {
                    if (any && sorted) continue; // any-hit rays take the children in a fixed order either way
                    Tally t;
                    size_t n = 0;
                    for (const Ray &r : rays)
                        if ((int)r.any == any)
                        {
                            walk(wide, r, sorted != 0, t);
                            n++;
                        }
                    std::printf("  %-7s rays, children %-25s node visits %6.2f  children tested %7.2f  triangle tests %5.2f  pushes %5.2f  (hit %.2f, deepest stack %.0f; %zu rays)\n",
                                any ? "shadow" : "bounce", any ? "in octant order:" : sorted ? "sorted by entry distance:" : "in octant order:",
                                t.nodeVisits / n, t.childTests / n, t.triTests / n, t.pushes / n, t.hits / n, t.maxStack, n);
                }
        }
    }
    return 0;
}
