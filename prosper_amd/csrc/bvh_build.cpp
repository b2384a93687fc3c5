// bvh_build.cpp — binned-SAH builder (host): one tree over everything (build_bvh), or one subtree per instance built in
// parallel under a top level over the instances (InstancedBvh), both emitted as the same 4-wide nodes.
#include "bvh_build.hpp"

#include "bvh_encode.hpp"

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <functional>
#include <memory>
#include <cstdio>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <future>
#include <limits>
#include <mutex>
#include <queue>
#include <stdexcept>
#include <thread>
#include <type_traits>

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

#ifndef PPT_SAH_BINS
#define PPT_SAH_BINS 16
#endif
constexpr int kBins = PPT_SAH_BINS; // (32 and 64 bins measured in round 4: profiles/r04_sah_bins.txt)
constexpr float kIntersectCost = 1.0f;

// tuning knobs (BvhBuildOptions::sahTraversalCost, leafSize): experiments only
float traversal_cost(const BvhBuildOptions &opt) { return opt.sahTraversalCost > 0.0f ? opt.sahTraversalCost : 1.0f; }
uint32_t max_leaf(const BvhBuildOptions &opt)
{
    const uint32_t n = opt.leafSize ? opt.leafSize : kMaxLeafTriangles;
    return std::min(std::max(n, 1u), 8u);
}

uint32_t ceil_log2(uint64_t n)
{
    uint32_t l = 0;
    while ((1ull << l) < n) ++l;
    return l;
}

// CPUs this process may actually use at once: a container's CPU quota (cgroup v2 cpu.max, v1 cfs_quota_us / cfs_period_us) is
// usually far below the CPUs it can SEE (an MI355X box: 256 visible, a quota of 16), and threads beyond the quota do not run
// in parallel - they use the quota up early and the whole process, its frame loop included, sits out the rest of the period.
unsigned cpu_quota()
{
    static const unsigned cached = [] {
        double quota = -1.0, period = 0.0;
        if (FILE *f = std::fopen("/sys/fs/cgroup/cpu.max", "r"))
        {
            char first[32] = {};
            if (std::fscanf(f, "%31s %lf", first, &period) == 2 && std::strcmp(first, "max") != 0) quota = std::atof(first);
            std::fclose(f);
        }
        else
        {
            if (FILE *q = std::fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r"))
            {
                if (std::fscanf(q, "%lf", &quota) != 1) quota = -1.0;
                std::fclose(q);
            }
            if (FILE *p = std::fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r"))
            {
                if (std::fscanf(p, "%lf", &period) != 1) period = 0.0;
                std::fclose(p);
            }
        }
        if (quota > 0.0 && period > 0.0) return (unsigned)std::max(1.0, std::ceil(quota / period));
        return 0u; // no quota
    }();
    return cached;
}

unsigned host_threads(const BvhBuildOptions &opt)
{
    unsigned threads = std::thread::hardware_concurrency();
    threads = std::max(1u, std::min(threads ? threads : 1u, 32u));
    if (const unsigned quota = cpu_quota()) threads = std::min(threads, quota);
    if (opt.buildThreads) threads = opt.buildThreads;
    return threads;
}

// The host's threads for ONE build: made once by the build's entry point (PoolScope) and reused by every parallel phase of
// it - subtrees, the top level's subranges, the emitter's plan and emit passes.  Making and joining 15 threads per phase cost
// 1.5-3 ms each on the 16-core host of an MI355X box, five times per build.  The pool dies with the build: the library
// keeps no thread of its own alive between calls.
class WorkerPool
{
  public:
    explicit WorkerPool(unsigned workers)
    {
        try
        {
            for (unsigned i = 0; i < workers; ++i) threads.emplace_back([this] { loop(); });
        }
        catch (...) // (no more threads to be had: the ones that did start must be joined before the vector goes)
        {
            shut_down();
            throw;
        }
    }
    ~WorkerPool() { shut_down(); }
    WorkerPool(const WorkerPool &) = delete;
    WorkerPool &operator=(const WorkerPool &) = delete;
    [[nodiscard]] unsigned workers() const { return (unsigned)threads.size(); }
    // job(0) .. job(n - 1) on the pool's threads and the caller's; rethrows the first exception
    void run(size_t n, const std::function<void(size_t)> &job)
    {
        {
            std::lock_guard<std::mutex> g(m);
            current = &job;
            count = n;
            next.store(0);
            active = threads.size();
            failure = nullptr;
            ++generation;
        }
        wake.notify_all();
        work();
        std::exception_ptr failed;
        {
            std::unique_lock<std::mutex> lk(m);
            done.wait(lk, [&] { return active == 0; });
            current = nullptr;
            failed = failure;
        }
        if (failed) std::rethrow_exception(failed);
    }

  private:
    void shut_down()
    {
        {
            std::lock_guard<std::mutex> g(m);
            stop = true;
        }
        wake.notify_all();
        for (std::thread &t : threads) t.join();
        threads.clear();
    }
    void work()
    {
        for (size_t j = next.fetch_add(1); j < count; j = next.fetch_add(1))
        {
            try
            {
                (*current)(j);
            }
            catch (...)
            {
                std::lock_guard<std::mutex> g(m);
                failure = std::current_exception();
            }
        }
    }
    void loop()
    {
        uint64_t seen = 0;
        std::unique_lock<std::mutex> lk(m);
        for (;;)
        {
            wake.wait(lk, [&] { return stop || generation != seen; });
            if (stop) return;
            seen = generation;
            lk.unlock();
            work();
            lk.lock();
            if (--active == 0) done.notify_one();
        }
    }
    std::vector<std::thread> threads;
    std::mutex m;
    std::condition_variable wake, done;
    const std::function<void(size_t)> *current = nullptr;
    size_t count = 0;
    std::atomic<size_t> next{0};
    size_t active = 0;
    uint64_t generation = 0;
    bool stop = false;
    std::exception_ptr failure;
};
thread_local WorkerPool *tlPool = nullptr; // the pool of the build running on this thread, if any
struct PoolScope
{
    WorkerPool *previous;
    std::unique_ptr<WorkerPool> pool;
    explicit PoolScope(const BvhBuildOptions &opt) : previous(tlPool)
    {
        const unsigned threads = host_threads(opt);
        if (!previous && threads > 1)
        {
            pool.reset(new WorkerPool(threads - 1));
            tlPool = pool.get();
        }
    }
    ~PoolScope() { tlPool = previous; }
};

// Runs job(0) .. job(n - 1) on the host's threads (the caller's included); rethrows the first exception.
template <class Job>
void run_parallel(const BvhBuildOptions &opt, size_t n, Job &&job)
{
    if (tlPool && n > 1)
    {
        const std::function<void(size_t)> fn = [&](size_t j) { job(j); };
        tlPool->run(n, fn);
        return;
    }
    const unsigned threads = (unsigned)std::min<size_t>(host_threads(opt), n ? n : 1);
    std::atomic<size_t> next{0};
    std::exception_ptr failure;
    std::mutex failureLock;
    auto worker = [&]() {
        for (size_t j = next.fetch_add(1); j < n; j = next.fetch_add(1))
        {
            try
            {
                job(j);
            }
            catch (...)
            {
                std::lock_guard<std::mutex> g(failureLock);
                failure = std::current_exception();
            }
        }
    };
    std::vector<std::thread> pool;
    for (unsigned t = 1; t < threads; ++t) pool.emplace_back(worker);
    worker();
    for (std::thread &t : pool) t.join();
    if (failure) std::rethrow_exception(failure);
}

struct Builder
{
    std::vector<Prim> &prims;
    const BvhBuildOptions &opt;
    std::vector<TmpNode> nodes;
    const float kTraversalCost = traversal_cost(opt);
    uint32_t kMaxLeaf = max_leaf(opt);
    // build_parallel(): subranges of at most this many primitives are not split here but queued as jobs
    uint32_t deferBelow = 0;
    struct Deferred
    {
        int32_t self;
        uint32_t first, count, depth;
    };
    std::vector<Deferred> deferred;

    Builder(std::vector<Prim> &p, const BvhBuildOptions &o) : prims(p), opt(o) {}

    // The same tree as build(), with the subranges below the first few splits built on the host's threads: every
    // split only permutes its own slice of `prims`, so the slices of different jobs are disjoint.
    int32_t build_parallel(uint32_t first, uint32_t count, uint32_t depth)
    {
        const unsigned threads = host_threads(opt);
        if (threads < 2 || count < 8192u) return build(first, count, depth);
        auto tick = std::chrono::steady_clock::now();
        auto lap = [&](const char *what) {
            const auto now = std::chrono::steady_clock::now();
            if (opt.buildTiming && count >= 32768u)
                std::fprintf(stderr, "[bvh]   %-12s %.1f ms (%u prims)\n", what, std::chrono::duration<double, std::milli>(now - tick).count(), count);
            tick = now;
        };
        // (the nodes above it are walked by ONE thread: as few levels as still give every thread its share below.  Gathering
        //  those nodes' bounds and bins on the pool's threads was tried: waking fifteen threads costs what it saves)
        deferBelow = std::max(1024u, count / threads);
        const int32_t root = build(first, count, depth);
        deferBelow = 0;
        lap("split top");
        std::vector<Deferred> jobs;
        jobs.swap(deferred);
        std::sort(jobs.begin(), jobs.end(), [](const Deferred &a, const Deferred &b) { return a.count > b.count; });
        std::vector<std::vector<TmpNode>> built(jobs.size());
        std::vector<int32_t> roots(jobs.size());
        run_parallel(opt, jobs.size(), [&](size_t j) {
            Builder local(prims, opt);
            local.kMaxLeaf = kMaxLeaf;
            local.nodes.reserve((size_t)jobs[j].count * 2);
            roots[j] = local.build(jobs[j].first, jobs[j].count, jobs[j].depth);
            built[j].swap(local.nodes);
        });
        lap("split jobs");
        size_t gathered = nodes.size();
        for (const std::vector<TmpNode> &part : built) gathered += part.size();
        nodes.reserve(gathered);
        for (size_t j = 0; j < jobs.size(); ++j)
        {
            const int32_t shift = (int32_t)nodes.size();
            for (TmpNode n : built[j])
            {
                if (n.left >= 0)
                {
                    n.left += shift;
                    n.right += shift;
                }
                nodes.push_back(n);
            }
            nodes[(size_t)jobs[j].self] = nodes[(size_t)(shift + roots[j])]; // the placeholder becomes the job's root
            TmpNode &copy = nodes[(size_t)(shift + roots[j])]; // ... and its copy, which nothing points at any more, an empty
            copy.left = copy.right = -1;                       // leaf: whoever counts inner nodes must not depend on how
            copy.first = copy.count = 0;                       // many jobs - i.e. threads - there were
        }
        lap("gather");
        return root;
    }

    // Builds the subtree over prims[first, first+count) and returns its node index in `nodes`.
    int32_t build(uint32_t first, uint32_t count, uint32_t depth)
    {
        const int32_t self = (int32_t)nodes.size();
        nodes.emplace_back();
        if (deferBelow && depth > 0 && count <= deferBelow)
        {
            deferred.push_back(Deferred{self, first, count, depth});
            return self;
        }
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
                if (count <= kMaxLeaf && leafCost <= splitCost) return make_leaf();
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
            else if (count <= kMaxLeaf)
                return make_leaf(); // all centroids coincide
        }
        if (!split)
        {
            // median split along the widest centroid axis
            axis = 0;
            if (cbox.hi[1] - cbox.lo[1] > cbox.hi[axis] - cbox.lo[axis]) axis = 1;
            if (cbox.hi[2] - cbox.lo[2] > cbox.hi[axis] - cbox.lo[axis]) axis = 2;
            if (forceMedian && count <= kMaxLeaf) return make_leaf();
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

// BvhBuildOptions::boxPad (>= 1.6e-5) fattens the boxes (bvh_encode.hpp enc_padded): a different hierarchy for the tests
// that check that hits do not depend on it
float pad_coefficient(const BvhBuildOptions &opt) { return opt.boxPad > 0.0f ? std::max(1.6e-5f, opt.boxPad) : 1.6e-5f; }

EncBox enc_box(const Aabb &b)
{
    EncBox e;
    for (int k = 0; k < 3; ++k)
    {
        e.lo[k] = b.lo[k];
        e.hi[k] = b.hi[k];
    }
    return e;
}

// Renumbers the nodes (speed only; the root stays node 0, leaf references do not change).  The emitter numbers them
// depth-first, which interleaves the few hundred nodes every ray walks through (the top of the tree) with deep ones
// only their own subtree's rays touch: every cache line of the hot set is then half cold.
//   PROSPER_PT_DEBUG_NODE_ORDER = 0   depth-first (as emitted)
//                                 1   breadth-first
//                                 2   the first kTopNodes nodes breadth-first, the subtrees below them depth-first (default)
// Measured (profiles/r02_scheduler_experiments.txt): wf_trace on C3 2562 / 2472 / 2480 us, no difference on C4 and
// FlightHelmet - the traversal is not waiting for its nodes most of the time.
void relayout_nodes(std::vector<BvhNode> &nodes, const BvhBuildOptions &opt)
{
    const int mode = opt.nodeOrder >= 0 ? opt.nodeOrder : 2;
    if (mode == 0 || nodes.size() < 3) return;
    const size_t n = nodes.size();
    const size_t kTopNodes = 4096;
    std::vector<int32_t> order; // order[new] = old
    order.reserve(n);
    std::vector<int32_t> queue;
    queue.push_back(0);
    size_t head = 0;
    const size_t bfsLimit = mode == 1 ? n : std::min(n, kTopNodes);
    while (head < queue.size() && order.size() < bfsLimit)
    {
        const int32_t at = queue[head++];
        order.push_back(at);
        for (int c = 0; c < 4; ++c)
            if (nodes[(size_t)at].child[c] >= 0) queue.push_back(nodes[(size_t)at].child[c]);
    }
    // what is left in the queue: roots of the subtrees below the breadth-first part, each emitted depth-first
    std::vector<int32_t> stack;
    for (; head < queue.size(); ++head)
    {
        stack.push_back(queue[head]);
        while (!stack.empty())
        {
            const int32_t at = stack.back();
            stack.pop_back();
            order.push_back(at);
            for (int c = 3; c >= 0; --c)
                if (nodes[(size_t)at].child[c] >= 0) stack.push_back(nodes[(size_t)at].child[c]);
        }
    }
    if (order.size() != n) throw std::runtime_error("relayout_nodes: the tree does not reach every node");
    std::vector<int32_t> newIndex(n);
    for (size_t i = 0; i < n; ++i) newIndex[(size_t)order[i]] = (int32_t)i;
    std::vector<BvhNode> moved(n);
    for (size_t i = 0; i < n; ++i)
    {
        BvhNode node = nodes[(size_t)order[i]];
        for (int c = 0; c < 4; ++c)
            if (node.child[c] >= 0) node.child[c] = newIndex[(size_t)node.child[c]];
        moved[i] = node;
    }
    nodes.swap(moved);
}

// Collapses the binary SAH tree into 4-wide nodes (a node adopts its grandchildren, largest box
// first, until it has four children or only leaves) and emits them depth-first.  Two passes: plan() walks the
// tree once and records, for every binary node that becomes a 4-wide node, its children and the size of its
// subtree - which fixes every node's index and every leaf's place in the triangle order - so that emit() can
// fill disjoint parts of the output on the host's threads.
struct Emitter
{
    const TmpNode *tmp; // the binary tree (tmpCount nodes)
    size_t tmpCount;
    const std::vector<Prim> &prims;
    BvhBuildResult &out;
    float slack;
    float padCoeff;
    bool sortChildren;

    struct Plan
    {
        int32_t kids[4];
        uint32_t k = 0;
        uint32_t nodes = 0;     // 4-wide nodes in the subtree, this one included
        uint32_t triangles = 0; // triangles in the subtree
        uint32_t stackNeed = 0;
    };
    Plan *plans; // by binary node index; WRITTEN for the nodes that become 4-wide nodes before anything reads them (raw storage)

    uint32_t triangles_of(int32_t t) const { return tmp[(size_t)t].left < 0 ? tmp[(size_t)t].count : plans[(size_t)t].triangles; }

    // the children of the 4-wide node that binary node t becomes (the collapse), in storage order
    void select_kids(int32_t t, Plan &p) const
    {
        int32_t *kids = p.kids;
        uint32_t k = 0;
        if (tmp[t].left < 0)
            kids[k++] = t;
        else
        {
            kids[k++] = tmp[t].left;
            kids[k++] = tmp[t].right;
            while (k < 4)
            {
                int best = -1;
                float bestArea = -1.0f;
                for (uint32_t i = 0; i < k; ++i)
                    if (tmp[kids[i]].left >= 0 && tmp[kids[i]].box.half_area() > bestArea)
                    {
                        bestArea = tmp[kids[i]].box.half_area();
                        best = (int)i;
                    }
                if (best < 0) break;
                const int32_t open = kids[best];
                kids[best] = tmp[open].left;
                kids[k++] = tmp[open].right;
            }
        }
        // Storage order = the order an any-hit (shadow) ray tries the children it enters (descend_any): smallest box
        // first.  A small subtree is cheap to rule out and ends the ray if it occludes; measured on S-sponza-class
        // (shadow + bounce rays, node visits per step): distance-sorted 1235 M, build order 1211 M, largest first
        // 1192 M, most triangles first 1219 M, smallest first 1138 M.  Closest-hit rays sort by distance anyway.
        // PROSPER_PT_DEBUG_CHILD_ORDER=0 keeps the build order (hierarchy-independence tests).
        if (sortChildren)
            std::stable_sort(kids, kids + k, [&](int32_t a, int32_t b) {
                return tmp[a].box.half_area() < tmp[b].box.half_area();
            });
        p.k = k;
    }
    bool inner_kid(int32_t t, int32_t kid) const { return tmp[kid].left >= 0 && kid != t; }
    // sizes and stack need of t's subtree, once its inner children have their plans
    void finish_plan(int32_t t, Plan &p) const
    {
        p.nodes = 1;
        p.triangles = 0;
        uint32_t deepest = 0;
        for (uint32_t c = 0; c < p.k; ++c)
        {
            const int32_t kid = p.kids[c];
            if (!inner_kid(t, kid))
                p.triangles += tmp[kid].count; // a leaf (or a lone leaf wrapped as the root)
            else
            {
                p.nodes += plans[(size_t)kid].nodes;
                p.triangles += plans[(size_t)kid].triangles;
                deepest = std::max(deepest, plans[(size_t)kid].stackNeed);
            }
        }
        // visiting this node leaves at most k-1 siblings on the stack while a child subtree is walked
        p.stackNeed = (p.k > 0 ? p.k - 1 : 0) + deepest;
    }
    void plan(int32_t t)
    {
        Plan p;
        select_kids(t, p);
        for (uint32_t c = 0; c < p.k; ++c)
            if (inner_kid(t, p.kids[c])) plan(p.kids[c]);
        finish_plan(t, p);
        plans[(size_t)t] = p;
    }
    // plan(root) with the subtrees below the first few levels planned on the host's threads (they write disjoint plans)
    void plan_parallel(int32_t root, const BvhBuildOptions &opt)
    {
        const unsigned threads = host_threads(opt);
        if (threads < 2 || tmpCount < 16384)
        {
            plan(root);
            return;
        }
        uint32_t levels = 1;
        for (uint64_t n = 4; n < 8ull * threads; n *= 4) ++levels; // >= 8 subtrees per thread if the tree were full
        std::vector<int32_t> top, frontier; // the shallow nodes in pre-order; the roots below them
        std::vector<std::pair<int32_t, uint32_t>> stack{{root, 0u}};
        while (!stack.empty())
        {
            const auto [t, depth] = stack.back();
            stack.pop_back();
            Plan p;
            select_kids(t, p);
            plans[(size_t)t] = p;
            top.push_back(t);
            for (uint32_t c = 0; c < p.k; ++c)
                if (inner_kid(t, p.kids[c]))
                {
                    if (depth + 1 < levels)
                        stack.push_back({p.kids[c], depth + 1});
                    else
                        frontier.push_back(p.kids[c]);
                }
        }
        run_parallel(opt, frontier.size(), [&](size_t j) { plan(frontier[j]); });
        for (size_t i = top.size(); i-- > 0;) // children after their parents in `top`: backwards, every child is done
        {
            Plan p = plans[(size_t)top[i]];
            finish_plan(top[i], p);
            plans[(size_t)top[i]] = p;
        }
    }

    struct Job
    {
        int32_t t;
        uint32_t self, triBase;
    };

    // Emits the 4-wide node of binary node t at index `self`; its subtree's triangles start at `triBase` of the
    // leaf order.  Subtrees of at most `chunk` nodes are queued in `jobs` instead of being walked (jobs == nullptr:
    // walk everything).
    void emit(int32_t t, uint32_t self, uint32_t triBase, uint32_t chunk, std::vector<Job> *jobs)
    {
        const Plan &p = plans[(size_t)t];
        const int32_t *kids = p.kids;
        const uint32_t k = p.k;
        BvhNode node;
        EncBox boxes[4];
        for (uint32_t c = 0; c < k; ++c) boxes[c] = enc_box(tmp[kids[c]].box);
        enc_node_boxes(boxes, k, padCoeff, slack, node); // unused slots: lo = hi = +inf can never be entered (see build_bvh)
        uint32_t nextNode = self + 1u, nextTri = triBase;
        for (uint32_t c = 0; c < 4; ++c)
        {
            node.child[c] = ~0;
            if (c >= k) continue;
            const TmpNode &kid = tmp[kids[c]];
            if (kid.left < 0)
            {
                // leaves longer than kMaxLeafTriangles cannot occur (the builder splits them)
                for (uint32_t i = 0; i < kid.count; ++i) out.permutation[nextTri + i] = prims[kid.first + i].index;
                node.child[c] = ~(int32_t)((nextTri << 3) | (kid.count - 1));
                nextTri += kid.count;
            }
            else
            {
                node.child[c] = (int32_t)nextNode;
                const Plan &cp = plans[(size_t)kids[c]];
                if (jobs && cp.nodes <= chunk)
                    jobs->push_back(Job{kids[c], nextNode, nextTri});
                else
                    emit(kids[c], nextNode, nextTri, chunk, jobs);
                nextNode += cp.nodes;
                nextTri += cp.triangles;
            }
        }
        out.nodes[self] = node;
    }
};

// Emits the binary tree `tmp` rooted at `root` as 4-wide nodes + leaf-order permutation into `out`.
void emit_tree(
    const TmpNode *tmp, size_t tmpCount, int32_t root, const std::vector<Prim> &prims, uint64_t count, BvhBuildResult &out,
    const BvhBuildOptions &opt)
{
    const float slack = enc_slack(enc_box(tmp[(size_t)root].box));
    static_assert(std::is_trivially_copyable<Emitter::Plan>::value && std::is_trivially_destructible<Emitter::Plan>::value, "raw storage");
    std::unique_ptr<void, void (*)(void *)> planStorage(std::malloc((tmpCount ? tmpCount : 1) * sizeof(Emitter::Plan)), std::free);
    if (!planStorage) throw std::bad_alloc();
    Emitter emitter{tmp, tmpCount, prims, out, slack, pad_coefficient(opt), opt.childOrder != 0, static_cast<Emitter::Plan *>(planStorage.get())};
    auto tick = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
        const auto now = std::chrono::steady_clock::now();
        if (opt.buildTiming) std::fprintf(stderr, "[bvh]   %-12s %.1f ms\n", what, std::chrono::duration<double, std::milli>(now - tick).count());
        tick = now;
    };
    emitter.plan_parallel(root, opt);
    lap("plan");
    const Emitter::Plan &top = emitter.plans[(size_t)root];
    out.maxDepth = top.stackNeed + 1; // entries the traversal stack must hold in the worst case
    if (out.maxDepth > kMaxStackBound)
        throw std::runtime_error("BVH stack bound exceeds the traversal's overflow capacity");
    if (top.triangles != count) throw std::runtime_error("emit_tree: the tree does not hold every triangle");
    out.nodes.assign(top.nodes, BvhNode());
    out.permutation.assign((size_t)count, 0u);
    std::vector<Emitter::Job> jobs;
    const uint32_t chunk = std::max(256u, top.nodes / (8u * host_threads(opt)));
    emitter.emit(root, 0u, 0u, chunk, host_threads(opt) > 1 ? &jobs : nullptr);
    lap("emit top");
    run_parallel(opt, jobs.size(), [&](size_t j) { emitter.emit(jobs[j].t, jobs[j].self, jobs[j].triBase, 0u, nullptr); });
    lap("emit jobs");
    relayout_nodes(out.nodes, opt);
    lap("relayout");
}

} // namespace

float bvh_pad_coefficient(const BvhBuildOptions &opt) { return pad_coefficient(opt); }
unsigned default_build_threads() { return host_threads(BvhBuildOptions()); }

BvhBuildResult build_bvh(const WorldTriangle *triangles, uint64_t count, const BvhBuildOptions &opt)
{
    const PoolScope threads(opt);
    BvhBuildResult out;
    // An unused child slot gets the box lo = hi = +inf: the slab tests (pt_device.hpp,
    // pt_trace_stream.hpp) then report an entry distance of +inf at best, which the traversal treats
    // as a miss.  (A NaN box would slip through min/max, an inverted box passes.)
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
        root.reserved = 0; // no child in use
        for (int a = 0; a < 3; ++a) root.origin[a] = 0.0f;
        for (int a = 0; a < 3; ++a)
            for (int c = 0; c < 4; ++c)
            {
                root.lo[a][c] = 0x7C00u;
                root.hi[a][c] = 0x7C00u;
            }
        for (int c = 0; c < 4; ++c) root.child[c] = ~0; // leaf references that are never entered
        out.nodes.push_back(root);
        out.maxDepth = 1;
        return out;
    }

    Builder builder(prims, opt);
    builder.nodes.reserve((size_t)count * 2);
    const int32_t rootTmp = builder.build_parallel(0, (uint32_t)count, 0);
    emit_tree(builder.nodes.data(), builder.nodes.size(), rootTmp, prims, count, out, opt);
    return out;
}

// ---- per-instance subtrees + top level (InstancedBvh) ----

struct InstancedBvh::Impl
{
    std::vector<Prim> prims;          // all triangles; every instance owns the slice [first, first + count)
    std::vector<Range> instances;
    struct Subtree
    {
        std::vector<TmpNode> nodes;   // child indices local to this vector, leaves index `prims` globally
        int32_t root = -1;
    };
    std::vector<Subtree> subtrees;
    uint64_t count = 0;
    BvhBuildOptions opt; // of the build / rebuild in progress

    void fill_prims(const WorldTriangle *triangles, const Range &r)
    {
        for (uint32_t i = r.first; i < r.first + r.count; ++i)
        {
            const WorldTriangle &t = triangles[i];
            Prim &p = prims[i];
            p.box = Aabb();
            p.box.grow(t.v0);
            p.box.grow(t.v1);
            p.box.grow(t.v2);
            for (int k = 0; k < 3; ++k) p.centroid[k] = 0.5f * (p.box.lo[k] + p.box.hi[k]);
            p.index = i;
        }
    }

    void build_subtree(const WorldTriangle *triangles, size_t i, uint32_t depth0, bool parallel = false)
    {
        const Range &r = instances[i];
        Subtree &st = subtrees[i];
        st.nodes.clear();
        st.root = -1;
        if (r.count == 0) return;
        fill_prims(triangles, r);
        Builder b(prims, opt);
        b.nodes.reserve((size_t)r.count * 2);
        st.root = parallel ? b.build_parallel(r.first, r.count, depth0) : b.build(r.first, r.count, depth0);
        st.nodes.swap(b.nodes);
    }

    // (re)builds the flagged subtrees on the host's threads: they touch disjoint slices of `prims`
    void build_subtrees(const WorldTriangle *triangles, const std::vector<uint8_t> &which)
    {
        const uint32_t depth0 = ceil_log2(instances.size() ? instances.size() : 1) + 1u;
        std::vector<size_t> todo;
        for (size_t i = 0; i < instances.size(); ++i)
            if (which.empty() || which[i]) todo.push_back(i);
        // largest first, so that the last thread to finish holds a small one
        std::sort(todo.begin(), todo.end(), [&](size_t a, size_t b) { return instances[a].count > instances[b].count; });
        // An instance with a large share of the triangles would keep one thread busy long after the others are done (S-sponza-
        // class: the hall, a quarter of the scene): those are split one after the other, each on all threads (the same
        // tree: Builder::build_parallel), the rest one per thread.
        size_t big = 0;
        if (host_threads(opt) > 1)
            while (big < todo.size() && instances[todo[big]].count >= 16384u && (uint64_t)instances[todo[big]].count * 8u >= count) ++big;
        for (size_t j = 0; j < big; ++j) build_subtree(triangles, todo[j], depth0, true);
        run_parallel(opt, todo.size() - big, [&](size_t j) { build_subtree(triangles, todo[big + j], depth0); });
    }

    // Top level over the instances, spliced with their subtrees into one binary tree, emitted as 4-wide nodes.
    // Instances overlap (foliage around the urns, cloth between the columns): with one top-level leaf per instance a
    // ray walks every subtree whose box it crosses - the usual price of a two-level hierarchy (S-sponza-class: +10 % frame
    // time, with the foliage instances +38 %).  So the top level is built over OPENED subtrees ("re-braiding"): starting
    // from the instance roots, the entry with the largest box is replaced by its two children until there is one entry
    // per kTrianglesPerTopEntry triangles: the upper levels of the tree are then split globally, like one SAH tree over
    // everything, and only the opened inner nodes of the subtrees go unused.  Swept on S-sponza-class (frame time, one
    // chain; profiles/r02_bvh_instancing.txt): C3 15.9 / 15.75 / 15.65 / 15.9 ms at 1 k / 4 k / 16 k / 64 k entries (one flat
    // SAH tree: 15.98); C4, whose foliage interleaves with everything, 42.7 / 40.4 / 41.3 / 37.1 ms (flat: 37.9).
    static constexpr size_t kTrianglesPerTopEntry = 4, kMinTopEntries = 1024, kMaxTopEntries = 65536;
    // (a top level of n entries has 2 n - 1 nodes + one unreferenced copy per job of Builder::build_parallel: at most 8 per thread)
    static constexpr size_t kTopSlack = 4096;
    BvhBuildResult assemble()
    {
        BvhBuildResult out;
        const bool timing = opt.buildTiming != 0;
        auto tick = std::chrono::steady_clock::now();
        auto lap = [&](const char *what) {
            const auto now = std::chrono::steady_clock::now();
            if (timing) std::fprintf(stderr, "[bvh] %-14s %.1f ms\n", what, std::chrono::duration<double, std::milli>(now - tick).count());
            tick = now;
        };
        // merged node array: every subtree with its indices shifted, the top-level nodes behind them.  Raw storage filled on
        // the host's threads: 26 MB of nodes for S-sponza-class that a std::vector would first construct, then overwrite.
        std::vector<size_t> offset(subtrees.size(), 0);
        size_t total = 0;
        std::vector<int32_t> roots;
        for (size_t i = 0; i < subtrees.size(); ++i)
        {
            offset[i] = total;
            if (subtrees[i].root < 0) continue;
            roots.push_back((int32_t)(total + (size_t)subtrees[i].root));
            total += subtrees[i].nodes.size();
        }
        if (roots.empty()) return build_bvh(nullptr, 0, opt);
        size_t target = std::min(std::max(kMinTopEntries, (size_t)(count / kTrianglesPerTopEntry)), kMaxTopEntries);
        if (opt.topEntries) target = (size_t)opt.topEntries;
        static_assert(std::is_trivially_copyable<TmpNode>::value && std::is_trivially_destructible<TmpNode>::value, "raw storage");
        struct Raw
        {
            TmpNode *p;
            explicit Raw(size_t n) : p(static_cast<TmpNode *>(std::malloc(n * sizeof(TmpNode))))
            {
                if (!p) throw std::bad_alloc();
            }
            ~Raw() { std::free(p); }
            Raw(const Raw &) = delete;
            Raw &operator=(const Raw &) = delete;
        } storage(total + 2 * target + kTopSlack);
        TmpNode *merged = storage.p;
        run_parallel(opt, subtrees.size(), [&](size_t i) {
            const Subtree &st = subtrees[i];
            if (st.root < 0) return;
            const int32_t shift = (int32_t)offset[i];
            TmpNode *dst = merged + offset[i];
            for (size_t k = 0; k < st.nodes.size(); ++k)
            {
                TmpNode n = st.nodes[k];
                if (n.left >= 0)
                {
                    n.left += shift;
                    n.right += shift;
                }
                dst[k] = n;
            }
        });
        lap("merge");
        // Opening "the entry with the largest box" until there are `target` entries opens the inner nodes in the order of
        // their boxes' areas, and a child's box never has a larger area than its parent's: what gets opened is the
        // k = target - roots inner nodes with the largest areas, whatever the order.  So instead of k rounds of a heap
        // (7 ms for 65 k entries) the k-th largest area is SELECTED (nth_element) and one walk from the roots opens every
        // inner node above it - and as many of those exactly at it as the count still allows.
        std::vector<int32_t> entries; // subtree nodes that become top-level leaves
        if (roots.size() >= target)
            entries = roots;
        else
        {
            const size_t k = target - roots.size();
            std::vector<float> areas;
            areas.reserve(total / 2 + 1);
            for (size_t i = 0; i < total; ++i)
                if (merged[i].left >= 0) areas.push_back(merged[i].box.half_area());
            float tau = -1.0f; // (areas are >= 0: everything opens)
            size_t atTau = 0;
            if (k < areas.size())
            {
                std::nth_element(areas.begin(), areas.begin() + (long)(k - 1), areas.end(), std::greater<float>());
                tau = areas[k - 1];
                size_t above = 0;
                for (size_t i = 0; i < k; ++i) above += areas[i] > tau ? 1u : 0u;
                atTau = k - above;
            }
            std::vector<int32_t> stack(roots.rbegin(), roots.rend());
            while (!stack.empty())
            {
                const int32_t at = stack.back();
                stack.pop_back();
                const TmpNode &n = merged[(size_t)at];
                if (n.left >= 0)
                {
                    const float a = n.box.half_area();
                    if (a > tau || (a == tau && atTau > 0))
                    {
                        if (!(a > tau)) --atTau;
                        stack.push_back(n.right);
                        stack.push_back(n.left);
                        continue;
                    }
                }
                entries.push_back(at); // a triangle leaf, or an inner node that stays closed
            }
        }
        lap("open");

        std::vector<Prim> top(entries.size());
        for (size_t i = 0; i < entries.size(); ++i)
        {
            top[i].box = merged[(size_t)entries[i]].box;
            for (int k = 0; k < 3; ++k) top[i].centroid[k] = 0.5f * (top[i].box.lo[k] + top[i].box.hi[k]);
            top[i].index = (uint32_t)i;
        }
        Builder tb(top, opt);
        tb.kMaxLeaf = 1; // a top-level leaf is exactly one entry
        const int32_t topRoot = tb.build_parallel(0, (uint32_t)top.size(), 0);
        lap("top level");
        const int32_t shift = (int32_t)total;
        size_t mergedCount = total;
        auto entry_of = [&](int32_t topLeaf) { return entries[top[tb.nodes[(size_t)topLeaf].first].index]; };
        int32_t root;
        if (tb.nodes[(size_t)topRoot].left < 0)
            root = entry_of(topRoot);
        else
        {
            root = shift + topRoot;
            if (tb.nodes.size() > 2 * target + kTopSlack) throw std::runtime_error("top level larger than its storage");
            mergedCount = total + tb.nodes.size(); // (top-level leaves are copied too but nothing points at them any more)
            const size_t chunk = 4096, chunks = (tb.nodes.size() + chunk - 1) / chunk;
            run_parallel(opt, chunks, [&](size_t c) {
                const size_t end = std::min(tb.nodes.size(), (c + 1) * chunk);
                for (size_t t = c * chunk; t < end; ++t)
                {
                    TmpNode n = tb.nodes[t];
                    if (n.left >= 0)
                    {
                        n.left = tb.nodes[(size_t)n.left].left < 0 ? entry_of(n.left) : n.left + shift;
                        n.right = tb.nodes[(size_t)n.right].left < 0 ? entry_of(n.right) : n.right + shift;
                    }
                    merged[(size_t)shift + t] = n;
                }
            });
        }
        lap("splice");
        emit_tree(merged, mergedCount, root, prims, count, out, opt);
        lap("emit");
        return out;
    }
};

InstancedBvh::InstancedBvh() : m(new Impl()) {}
InstancedBvh::~InstancedBvh() { delete m; }
size_t InstancedBvh::instanceCount() const { return m->instances.size(); }

static void check_ranges_tile(const std::vector<InstancedBvh::Range> &instances, uint64_t count)
{
    uint64_t covered = 0;
    for (const InstancedBvh::Range &r : instances)
    {
        if ((uint64_t)r.first != covered) throw std::runtime_error("instance triangle ranges must tile the triangle array");
        covered += r.count;
    }
    if (covered != count) throw std::runtime_error("instance triangle ranges must tile the triangle array");
}

BvhBuildResult InstancedBvh::build(
    const WorldTriangle *triangles, uint64_t count, const std::vector<Range> &instances, const BvhBuildOptions &opt)
{
    const PoolScope threads(opt);
    m->opt = opt;
    if (count >= (1ull << 28)) throw std::runtime_error("too many triangles for the leaf reference encoding");
    check_ranges_tile(instances, count);
    m->count = count;
    m->instances = instances;
    m->prims.assign((size_t)count, Prim());
    m->subtrees.assign(instances.size(), Impl::Subtree());
    m->build_subtrees(triangles, std::vector<uint8_t>());
    return m->assemble();
}

BvhBuildResult InstancedBvh::adopt(
    const WorldTriangle *triangles, uint64_t count, const std::vector<Range> &instances, const std::vector<uint8_t> &changed,
    const BvhBuildOptions &opt)
{
    const PoolScope threads(opt);
    m->opt = opt;
    if (count >= (1ull << 28)) throw std::runtime_error("too many triangles for the leaf reference encoding");
    if (instances.size() != m->instances.size() || changed.size() != instances.size())
        throw std::runtime_error("InstancedBvh::adopt: one range and one flag per instance of the last build");
    check_ranges_tile(instances, count);
    std::vector<Prim> moved((size_t)count);
    for (size_t i = 0; i < instances.size(); ++i)
    {
        if (changed[i]) continue; // build_subtree fills its slice
        const Range &was = m->instances[i], &now = instances[i];
        if (was.count != now.count) throw std::runtime_error("InstancedBvh::adopt: an unflagged instance changed its triangle count");
        const uint32_t delta = now.first - was.first; // (modulo 2^32: ranges may move down as well as up)
        for (uint32_t k = 0; k < now.count; ++k)
        {
            Prim p = m->prims[(size_t)was.first + k];
            p.index += delta;
            moved[(size_t)now.first + k] = p;
        }
        for (TmpNode &n : m->subtrees[i].nodes)
            if (n.left < 0) n.first += delta;
    }
    m->prims.swap(moved);
    m->instances = instances;
    m->count = count;
    m->build_subtrees(triangles, changed);
    return m->assemble();
}

BvhBuildResult InstancedBvh::rebuild(const WorldTriangle *triangles, const std::vector<uint8_t> &changed, const BvhBuildOptions &opt)
{
    const PoolScope threads(opt);
    m->opt = opt;
    if (changed.size() != m->instances.size()) throw std::runtime_error("InstancedBvh::rebuild: one flag per instance");
    m->build_subtrees(triangles, changed);
    return m->assemble();
}

} // namespace ppt
