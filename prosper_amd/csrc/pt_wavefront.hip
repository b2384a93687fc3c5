// pt_wavefront.hip — wavefront formulation of the path-tracing reference pass (gfx950).
//
// The megakernel of rt/reference/main.rgen:225-299 runs at ~25 % lane utilisation on 64-wide
// waves (measured: SQ_THREAD_CYCLES_VALU / SQ_INSTS_VALU, profiles/): lanes whose ray missed sit
// through the shading of the others, only the lanes that drew a front-facing light traverse the
// shadow ray, and every traversal loop runs as long as its slowest lane.  Here the loop body is cut
// into stage kernels and each stage only ever sees live work:
//
//   generate_extend   camera rays (ray.glsl:15-78) + first traceClosest, sky on a miss
//   shade             evaluateSurface + evaluateDirectLighting (without visibility) +
//                     importanceSampleBounce + roulette (main.rgen:146-223,90-144,269-276)
//   shadow            shadow() any-hit traversal (main.rgen:49-60), adds the direct term
//   extend            traceClosest of the next bounce (main.rgen:62-81), sky on a miss
//   accumulate        running mean over the batch's frames + history (main.rgen:285-298)
//
// Scheduling is per WAVE: a wave owns one fixed segment of path slots (WavefrontBuffers) through
// all bounces and compacts survivors to the front of its segment with __ballot + popcount.  No
// global queue, no atomics, fully coalesced 16-byte-per-lane state traffic, and the memory layout
// (hence every sum) is identical from run to run.  Results are bit-identical to the megakernel:
// every path keeps its own RNG stream and its contributions are added in the reference's order
// (direct of bounce b, then sky of bounce b+1).
#include "pt_kernels.hpp"

#include "pt_device.hpp"
#include "pt_render_common.hpp"
#ifdef PPT_EXPERIMENTS
#include "pt_trace_pool.hpp"
#endif
#include "pt_trace_stream.hpp"

// Register budgets (measured on C2/C3, profiles/r01_occupancy_ab.txt): the traversal kernels (generate,
// trace) run at 5 waves/SIMD (<= 96 VGPRs, no spill), the shade kernel at 4 (120 VGPRs; 5 would spill).
// The LDS stack is 16 or 24 entries x 64 lanes x 4 B per wave, deeper entries go to global memory.
#define PPT_SHADE_WPE 4
#ifndef PPT_GEN_WPE
#define PPT_GEN_WPE 5
#endif
#ifndef PPT_TRACE_WPE
#define PPT_TRACE_WPE(stack) 5
#endif

namespace ppt
{

namespace
{

constexpr uint32_t kSlotMask = 0x0FFFFFFFu;

struct SegmentId
{
    uint32_t seg;
    uint32_t base;
    bool valid;
};

// Workgroup -> 4 consecutive segments of this launch's group range; block ids are remapped so that each
// XCD (blocks b, b+8, ...) works on one contiguous range of segments, i.e. one band of the image
// (speed only).
__device__ __forceinline__ SegmentId my_segment(const WavefrontBuffers &w)
{
    const uint32_t groups = w.groupCount;
    const uint32_t perXcd = (groups + 7u) / 8u;
    const uint32_t local = (blockIdx.x % 8u) * perXcd + (blockIdx.x / 8u);
    const uint32_t seg = (w.groupBase + local) * 4u + (threadIdx.x >> 6);
    SegmentId id;
    id.seg = seg;
    id.base = seg * w.segLen;
    id.valid = local < groups && seg < w.nSeg;
    return id;
}

__device__ __forceinline__ uint32_t lane_id() { return threadIdx.x & 63u; }

// Exclusive rank of this lane among the lanes whose predicate is set, and the total.
__device__ __forceinline__ uint32_t wave_rank(bool pred, uint32_t &total)
{
    const unsigned long long m = __ballot(pred);
    total = (uint32_t)__builtin_popcountll(m);
    return (uint32_t)__builtin_popcountll(m & ((1ull << lane_id()) - 1ull));
}

__device__ __forceinline__ f3 xyz(float4 v) { return f3{v.x, v.y, v.z}; }
__device__ __forceinline__ uint32_t asu(float f) { return __builtin_bit_cast(uint32_t, f); }
__device__ __forceinline__ float asf(uint32_t u) { return __builtin_bit_cast(float, u); }

// The path state between stages (ray, throughput, generator, hit and shadow records) is written by one stage and read by
// the next one or two: 3.3 GB per 8-spp step that passes through the L2 and the Infinity Cache once.  PPT_STREAM_NT marks
// those accesses non-temporal (global_load / global_store ... nt), so that they do not push the scene - nodes, triangles,
// shading records, texels: what IS re-read - out of the caches.  (profiles/r04_stream_nt.txt)
#ifndef PPT_STREAM_NT
#define PPT_STREAM_NT 7 // bit 0: loads, bit 1: stores, bit 2: the radiance slots too (round 4: C3 12.95 -> 12.50 ms, others -0 .. 1 %)
#endif
typedef float ppt_f4v __attribute__((ext_vector_type(4)));
typedef uint32_t ppt_u4v __attribute__((ext_vector_type(4)));
typedef uint32_t ppt_u2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float4 sld(const float4 *p)
{
#if PPT_STREAM_NT & 1
    const ppt_f4v v = __builtin_nontemporal_load(reinterpret_cast<const ppt_f4v *>(p));
    return make_float4(v.x, v.y, v.z, v.w);
#else
    return *p;
#endif
}
__device__ __forceinline__ uint4 sld(const uint4 *p)
{
#if PPT_STREAM_NT & 1
    const ppt_u4v v = __builtin_nontemporal_load(reinterpret_cast<const ppt_u4v *>(p));
    return make_uint4(v.x, v.y, v.z, v.w);
#else
    return *p;
#endif
}
__device__ __forceinline__ uint2 sld(const uint2 *p)
{
#if PPT_STREAM_NT & 1
    const ppt_u2v v = __builtin_nontemporal_load(reinterpret_cast<const ppt_u2v *>(p));
    return make_uint2(v.x, v.y);
#else
    return *p;
#endif
}
__device__ __forceinline__ uint32_t sld(const uint32_t *p)
{
#if PPT_STREAM_NT & 1
    return __builtin_nontemporal_load(p);
#else
    return *p;
#endif
}
__device__ __forceinline__ void sst(float4 *p, float4 v)
{
#if PPT_STREAM_NT & 2
    __builtin_nontemporal_store(ppt_f4v{v.x, v.y, v.z, v.w}, reinterpret_cast<ppt_f4v *>(p));
#else
    *p = v;
#endif
}
__device__ __forceinline__ void sst(uint4 *p, uint4 v)
{
#if PPT_STREAM_NT & 2
    __builtin_nontemporal_store(ppt_u4v{v.x, v.y, v.z, v.w}, reinterpret_cast<ppt_u4v *>(p));
#else
    *p = v;
#endif
}
__device__ __forceinline__ void sst(uint2 *p, uint2 v)
{
#if PPT_STREAM_NT & 2
    __builtin_nontemporal_store(ppt_u2v{v.x, v.y}, reinterpret_cast<ppt_u2v *>(p));
#else
    *p = v;
#endif
}
__device__ __forceinline__ void sst(uint32_t *p, uint32_t v)
{
#if PPT_STREAM_NT & 2
    __builtin_nontemporal_store(v, p);
#else
    *p = v;
#endif
}

struct SlotPixel
{
    uint32_t lx, py, frame;
    bool valid;
};
__device__ __forceinline__ SlotPixel decode_slot(const WavefrontBuffers &w, const RenderParams &p, uint32_t slot)
{
    SlotPixel r;
    r.frame = slot / w.pixelsPadded;
    const uint32_t rem = slot - r.frame * w.pixelsPadded;
    const uint32_t tile = rem >> 6, inTile = rem & 63u;
    const uint32_t ty = tile / w.tilesX;
    r.lx = (tile - ty * w.tilesX) * 8u + (inTile & 7u);
    r.py = ty * 8u + (inTile >> 3);
    r.valid = r.frame < p.frameCount && r.lx < p.localWidth && r.py < p.height;
    return r;
}

// Which (frame, pixel) lane `lane` of batch `q` of a render's camera rays is.  A batch is 64 paths: NOT one 8x8 tile of one
// frame but a block of 64 / g pixels of a tile in g consecutive frames (g = 64 while that many frames remain, then 32,
// 16, ... 1 - a single frame is the 8x8 tile again).  The g camera rays of a pixel differ by their sub-pixel jitter
// only: they walk the same nodes and mostly hit the same triangle, so the lockstep traversal of the batch diverges far
// less than 64 different pixels do (a dense mesh has a triangle every few pixels), and in wf_shade the hits of a pixel
// sit next to each other - one shading record, one texel footprint for all of them.  Slots keep their meaning (frame *
// pixelsPadded + tile * 64 + pixel in tile): only the order the camera rays are generated in changes.
// (FlightHelmet, 8 spp: wf_generate_extend 1249 -> 918 us; C3 1536 -> 1449 and wf_shade 1004 -> 934; C4 6860 -> 6056.)
struct BatchLane
{
    uint32_t slot, lx, py, frame;
    bool valid;
};
#ifndef PPT_BATCH_MAX_FRAMES
#define PPT_BATCH_MAX_FRAMES 64u
#endif
// `q` counts the batches of the tiles [tileBase, tileBase + tiles) only (the whole image: 0, pixelsPadded >> 6).
__device__ __forceinline__ BatchLane batch_lane(
    const WavefrontBuffers &w, const RenderParams &p, uint32_t q, uint32_t lane, uint32_t tileBase, uint32_t tiles)
{
    constexpr uint32_t kMax = PPT_BATCH_MAX_FRAMES, kMaxShift = kMax == 64u ? 6u : (kMax == 32u ? 5u : (kMax == 16u ? 4u : (kMax == 8u ? 3u : 0u)));
    static_assert(kMaxShift != 0u || kMax == 1u, "PPT_BATCH_MAX_FRAMES is 1, 8, 16, 32 or 64");
    const uint32_t frames = p.frameCount;
    uint32_t shift = kMaxShift, firstFrame = 0u, rem = q;
    const uint32_t fullGroups = frames >> kMaxShift;
    const uint32_t fullBatches = (fullGroups << kMaxShift) * tiles; // batches of the groups of kMax frames
    bool inRange = true;
    if (rem < fullBatches)
    {
        const uint32_t group = rem / (tiles << kMaxShift);
        firstFrame = group << kMaxShift;
        rem -= group * (tiles << kMaxShift);
    }
    else
    {
        rem -= fullBatches;
        firstFrame = fullGroups << kMaxShift;
        inRange = false;
        for (shift = kMaxShift; shift-- != 0u;)
            if (frames & (1u << shift))
            {
                if (rem < (tiles << shift))
                {
                    inRange = true;
                    break;
                }
                rem -= tiles << shift;
                firstFrame += 1u << shift;
            }
        if (!inRange) shift = 0u;
    }
    // 2^shift batches per tile, each a block of 64 >> shift pixels: 8x8, 8x4, 4x4, 4x2, 2x2, 2x1, 1x1
    const uint32_t rank = rem >> shift, part = rem & ((1u << shift) - 1u);
#ifdef PPT_EXPERIMENTS
    const uint32_t tile = (w.tileOrder != nullptr && tileBase + rank < (w.pixelsPadded >> 6)) ? w.tileOrder[tileBase + rank] : tileBase + rank; // (wave-uniform: a scalar load)
#else
    const uint32_t tile = tileBase + rank;
#endif
    const uint32_t wShift = (7u - shift) >> 1, hShift = (6u - shift) >> 1; // log2 of the block's width and height
    const uint32_t pixel = lane & ((64u >> shift) - 1u);
    const uint32_t bx = part & ((8u >> wShift) - 1u), by = part >> (3u - wShift);
    const uint32_t ix = (bx << wShift) + (pixel & ((1u << wShift) - 1u)), iy = (by << hShift) + (pixel >> wShift);
    BatchLane r;
    r.frame = firstFrame + (lane >> (6u - shift));
    const uint32_t ty = tile / w.tilesX;
    r.lx = (tile - ty * w.tilesX) * 8u + ix;
    r.py = ty * 8u + iy;
    r.slot = r.frame * w.pixelsPadded + tile * 64u + iy * 8u + ix;
    r.valid = inRange && rank < tiles && r.frame < frames && r.lx < p.localWidth && r.py < p.height;
    return r;
}

// addBounce (main.rgen:83-88) into the path's radiance slot
__device__ __forceinline__ void add_to_slot(float4 *color, uint32_t slot, uint32_t flags, f3 value, uint32_t bounce)
{
#if PPT_STREAM_NT & 4
    const ppt_f4v cv = __builtin_nontemporal_load(reinterpret_cast<const ppt_f4v *>(color + slot));
    float4 c = make_float4(cv.x, cv.y, cv.z, cv.w);
#else
    float4 c = color[slot];
#endif
    f3 acc = f3{c.x, c.y, c.z};
    add_bounce(flags, acc, value, bounce);
#if PPT_STREAM_NT & 4
    __builtin_nontemporal_store(ppt_f4v{acc.x, acc.y, acc.z, 0.0f}, reinterpret_cast<ppt_f4v *>(color + slot));
#else
    color[slot] = make_float4(acc.x, acc.y, acc.z, 0.0f);
#endif
}

// How a wave walks a stream of rays: lane-owned (trace_stream) or out of its LDS ray pool (trace_pool)
struct StreamTracer
{
    TraversalStack stack;
    template <bool ANY, bool COUNT, class Geom, class Fetch, class Commit>
    __device__ __forceinline__ void run(
        const Geom &g, const DeviceScene &s, uint32_t n, float, LaneCounters &cnt, Fetch &&fetch, Commit &&commit) const
    {
        trace_stream<ANY, COUNT>(g, s, n, stack, cnt, fetch, commit);
    }
};
// 64 rays at a time in lockstep (trace_in's while-while loop), as the camera rays are traced: no phases, no refill.
struct LockstepTracer
{
    TraversalStack stack;
    template <bool ANY, bool COUNT, class Geom, class Fetch, class Commit>
    __device__ __forceinline__ void run(
        const Geom &g, const DeviceScene &s, uint32_t n, float, LaneCounters &cnt, Fetch &&fetch, Commit &&commit) const
    {
        const uint32_t lane = threadIdx.x & 63u;
        for (uint32_t k0 = 0; k0 < n; k0 += 64u)
        {
            const uint32_t k = k0 + lane;
            const bool valid = k < n;
            StreamRay r;
            r.o = f3{0.0f, 0.0f, 0.0f};
            r.d = f3{0.0f, 0.0f, 1.0f};
            r.tMin = 0.0f;
            r.tMax = -1.0f; // a ray that cannot hit anything
            r.seed = 0u;
            if (valid) r = fetch(k);
            Hit hit;
            const bool found = trace_in<ANY, COUNT>(g, s, r.o, r.d, r.tMin, r.tMax, r.seed, stack, hit, cnt);
            commit(valid, k, valid && found, hit, r.d);
        }
    }
};
#ifdef PPT_EXPERIMENTS
template <uint32_t P, uint32_t S, uint32_t B>
struct PoolTracer
{
    RayPool<P, S> pool;
    template <bool ANY, bool COUNT, class Geom, class Fetch, class Commit>
    __device__ __forceinline__ void run(
        const Geom &g, const DeviceScene &s, uint32_t n, float tMin, LaneCounters &cnt, Fetch &&fetch, Commit &&commit) const
    {
        trace_pool<ANY, COUNT, B>(g, s, n, tMin, pool, cnt, fetch, commit);
    }
};
#endif

// Copies the BVH nodes and world triangles into this workgroup's LDS as they are (LdsGeomHalf).
__device__ __forceinline__ LdsGeomHalf stage_scene_in_lds_half(const DeviceScene &s, float4 *lds, uint32_t nodeCount, uint32_t triCount)
{
    const float4 *gn = reinterpret_cast<const float4 *>(s.nodes);
    for (uint32_t i = threadIdx.x; i < nodeCount * kLdsNodeStrideHalf; i += blockDim.x) lds[i] = gn[i];
    float4 *lt = lds + nodeCount * kLdsNodeStrideHalf;
    const float4 *gt = reinterpret_cast<const float4 *>(s.triangles);
    for (uint32_t i = threadIdx.x; i < triCount * 3u; i += blockDim.x) lt[i] = gt[i];
    __syncthreads();
    return LdsGeomHalf{lds, lt};
}

// The BVH nodes - converted to the fp32 image of LdsGeom (pt_device.hpp) - and the world triangles into this workgroup's LDS.
__device__ __forceinline__ LdsGeom stage_scene_in_lds(const DeviceScene &s, float4 *lds, uint32_t nodeCount, uint32_t triCount)
{
    const BvhNode *gn = s.nodes;
    float *lf = reinterpret_cast<float *>(lds);
    // one thread per (node, float of the 36-float image)
    for (uint32_t i = threadIdx.x; i < nodeCount * (kLdsNodeStride * 4u); i += blockDim.x)
    {
        const uint32_t n = i / (kLdsNodeStride * 4u), k = i - n * (kLdsNodeStride * 4u);
        const uint32_t q = k >> 2, c = k & 3u; // float4 index within the image, component
        float v = 0.0f;
        if (q == 0u)
            v = c < 3u ? gn[n].origin[c] : 0.0f;
        else if (q >= 1u && q <= 3u)
            v = half_to_float(gn[n].lo[q - 1u][c]);
        else if (q == 4u)
            v = __builtin_bit_cast(float, gn[n].child[c]);
        else if (q <= 7u)
            v = half_to_float(gn[n].hi[q - 5u][c]);
        lf[i] = v;
    }
    float4 *lt = lds + nodeCount * kLdsNodeStride;
    const float4 *gt = reinterpret_cast<const float4 *>(s.triangles);
    for (uint32_t i = threadIdx.x; i < triCount * 3u; i += blockDim.x) lt[i] = gt[i];
    __syncthreads();
    return LdsGeom{lds, lt};
}

} // namespace

#ifdef PPT_EXPERIMENTS
// ------------------------------------------------------------------------------------------
// tile order: which tiles are expensive?
// ------------------------------------------------------------------------------------------
//
// EXPERIMENT (debug option tileOrder, -DPPT_EXPERIMENTS builds only), measured and NOT the default (profiles/r03_tile_order.txt).
// A segment takes every nSeg-th batch of the render's batch sequence.  In raster order its share of expensive tiles
// (FlightHelmet: a fifth of the tiles see the mesh and cost 15-20x a sky tile) varies from segment to segment like any
// systematic sample of a patchy image - the slowest wave sets the launch's tail (cu_busy 0.78).  Sorted by cost, the
// sequence is monotone and every stride through it gets the same mix.  The cost of a tile = node visits + triangle tests
// + any-hit calls of ONE probe ray through its centre (32 400 rays for a 1920x1080 frame, a few microseconds), binned to
// 256 levels; a counting sort (histogram by the probe kernel, a one-workgroup scan, a scatter) orders the tiles heaviest
// first.  Within a bin the order is whatever the atomics give: it decides which wave traces which tile, never a pixel.
// Result: the waves' loads do even out, and the kernels get SLOWER (FlightHelmet wf_generate_extend 888 -> 977 us, C4 5667 ->
// 6047; steps +2 to +4 %): neighbouring workgroups no longer work on neighbouring tiles, and every wave meets its heavy
// tiles at the same time.  The strided raster order - a stratified sample of the image already - stays.

template <int STACK>
__global__ __launch_bounds__(256) void probe_tiles_kernel(
    DeviceScene s, RenderParams p, uint32_t tilesX, uint32_t tiles, int32_t *__restrict__ stackOverflow, uint32_t overflowStride,
    uint32_t *__restrict__ bins, uint32_t *__restrict__ histogram)
{
    __shared__ int32_t ldsStack[STACK * 256];
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    const TraversalStack stack{(lds_int32 *)ldsStack + (threadIdx.x >> 6) * (STACK * 64u) + (threadIdx.x & 63u),
                               stackOverflow + blockIdx.x * 256u + threadIdx.x, (uint32_t)STACK, overflowStride, 64u};
    if (t >= tiles) return;
    const uint32_t ty = t / tilesX, tx = t - ty * tilesX;
    const uint32_t lx = tx * 8u + 4u, py = ty * 8u + 4u;
    uint32_t cost = 0u;
    if (lx < p.localWidth + 4u && py < p.height + 4u)
    {
        const f2 uv = f2{((float)local_to_global_x(p, lx < p.localWidth ? lx : p.localWidth - 1u)) / (float)p.width,
                         ((float)(py < p.height ? py : p.height - 1u)) / (float)p.height};
        const Ray ray = pinhole_camera_ray(p, uv);
        LaneCounters cnt = {};
        Hit hit;
        (void)trace_in<false, true>(GlobalGeom{s.nodes, s.triangles}, s, ray.o, ray.d, 0.0f, kInf, 0x9E3779B9u, stack, hit, cnt);
        cost = cnt.nodeVisits + cnt.triangleTests + cnt.anyHitCalls;
    }
    const uint32_t bin = cost > 255u ? 255u : cost;
    bins[t] = bin;
    atomicAdd(&histogram[bin], 1u);
}

// histogram[256] -> cursor[b] = tiles in heavier bins (one workgroup of 256)
__global__ __launch_bounds__(256) void tile_order_scan_kernel(const uint32_t *__restrict__ histogram, uint32_t *__restrict__ cursor)
{
    __shared__ uint32_t counts[256];
    counts[threadIdx.x] = histogram[threadIdx.x];
    __syncthreads();
    uint32_t before = 0;
    for (uint32_t b = threadIdx.x + 1u; b < 256u; ++b) before += counts[b];
    cursor[threadIdx.x] = before;
}

__global__ __launch_bounds__(256) void tile_order_scatter_kernel(
    const uint32_t *__restrict__ bins, uint32_t *__restrict__ cursor, uint32_t *__restrict__ order, uint32_t tiles)
{
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= tiles) return;
    order[atomicAdd(&cursor[bins[t]], 1u)] = t;
}

// scratch: bins[tiles], histogram[256], cursor[256] (uint32 each) behind order[tiles]
void launch_tile_order(
    const DeviceScene &s, const RenderParams &p, uint32_t tilesX, uint32_t tilesY, uint32_t ldsStackEntries, int32_t *stackOverflow,
    uint32_t *order, uint32_t *scratch, hipStream_t stream)
{
    const uint32_t tiles = tilesX * tilesY;
    if (tiles == 0) return;
    uint32_t *bins = scratch, *histogram = scratch + tiles, *cursor = histogram + 256;
    (void)hipMemsetAsync(histogram, 0, 256 * sizeof(uint32_t), stream);
    const dim3 grid((tiles + 255u) / 256u), block(256);
    const uint32_t stride = grid.x * 256u;
    if (ldsStackEntries == 16u)
        hipLaunchKernelGGL(probe_tiles_kernel<16>, grid, block, 0, stream, s, p, tilesX, tiles, stackOverflow, stride, bins, histogram);
    else if (ldsStackEntries == 24u)
        hipLaunchKernelGGL(probe_tiles_kernel<24>, grid, block, 0, stream, s, p, tilesX, tiles, stackOverflow, stride, bins, histogram);
    else
        hipLaunchKernelGGL(probe_tiles_kernel<32>, grid, block, 0, stream, s, p, tilesX, tiles, stackOverflow, stride, bins, histogram);
    hipLaunchKernelGGL(tile_order_scan_kernel, dim3(1), dim3(256), 0, stream, histogram, cursor);
    hipLaunchKernelGGL(tile_order_scatter_kernel, grid, block, 0, stream, bins, cursor, order, tiles);
}

#endif // PPT_EXPERIMENTS

// ------------------------------------------------------------------------------------------
// generate + first extend
// ------------------------------------------------------------------------------------------

template <bool COUNT, int STACK, bool LDS_SCENE>
__global__ __launch_bounds__(256, PPT_GEN_WPE) void wf_generate_extend(
    DeviceScene s, RenderParams p, WavefrontBuffers w, uint32_t nodeCount, uint32_t triCount,
    int32_t *__restrict__ stackOverflow, unsigned long long *__restrict__ counters)
{
    __shared__ int32_t ldsStack[STACK * 256];
    __shared__ float4 ldsScene[LDS_SCENE ? kLdsSceneFloat4s : 1];
#ifdef PPT_EXPERIMENT_GEN_LDS_F32
    LdsGeom lg = {};
    if constexpr (LDS_SCENE) lg = stage_scene_in_lds(s, ldsScene, nodeCount, triCount);
#else
    LdsGeomHalf lg = {};
    if constexpr (LDS_SCENE) lg = stage_scene_in_lds_half(s, ldsScene, nodeCount, triCount);
#endif
    const GlobalGeom gg{s.nodes, s.triangles};
    const SegmentId id = my_segment(w);
    if (!id.valid) return;
    const uint32_t lane = lane_id();
    const TraversalStack stack{(lds_int32 *)ldsStack + (threadIdx.x >> 6) * (STACK * 64u) + lane,
                               stackOverflow + blockIdx.x * 256u + threadIdx.x, (uint32_t)STACK, gridDim.x * 256u, 64u, LDS_SCENE};
    const bool traceRays = p.pc.maxBounces > 0;

    LaneCounters cnt = {};
    uint32_t nHit = 0;
    if (!traceRays)
    {
        // maxBounces == 0: the loop of main.rgen:241-244 never runs, every path is black
        for (uint32_t k = lane; k < w.segLen; k += 64u)
        {
            const SlotPixel sp = decode_slot(w, p, id.base + k);
            if (sp.valid)
            {
                if constexpr (COUNT) cnt.paths++;
                w.color[id.base + k] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            }
        }
    }
    else
    {
        // Batch b of segment s is batch number q = b * nSeg + s of the frame's batch sequence (batch_lane): every segment
        // gets batches from all over the image (and all frame groups), so the waves of a launch carry statistically
        // equal work and finish together; consecutive segments still start on neighbouring batches.
        BatchLane bl = {};
        auto fetch = [&](uint32_t k) {
            const uint32_t at = id.base + k; // where the path's records live in this segment
            StreamRay r;
            if (!bl.valid)
            {
                // slot outside the image / batch: a ray that cannot hit anything
                r.o = f3{0.0f, 0.0f, 0.0f};
                r.d = f3{0.0f, 0.0f, 1.0f};
                r.tMin = 0.0f;
                r.tMax = -1.0f;
                r.seed = 0;
                return r;
            }
            PathState st;
            start_path<COUNT>(p, local_to_global_x(p, bl.lx), bl.py, (p.pc.frameIndex + bl.frame) % PROSPER_RT_FRAME_PERIOD,
                              st, cnt);
            if constexpr (COUNT) cnt.closestRays++;
            // a path's state between stages: direction + one word of the generator, the other two words, and for
            // bounce 0 - no throughput record, it is (1, 1, 1) - the radiance slot as one dword (28 B per camera path)
            sst(&w.rayB[0][at], make_float4(st.d.x, st.d.y, st.d.z, asf(st.rng.x)));
            sst(&w.pathR[0][at], make_uint2(st.rng.y, st.rng.z));
            sst(&w.cameraSlot[at], bl.slot);
            r.o = st.o;
            r.d = st.d;
            r.tMin = 0.0f;
            r.tMax = kInf;
            r.seed = pcg(st.rng.x ^ st.rng.z);
            return r;
        };
        auto commit = [&](bool pred, uint32_t k, bool found, const Hit &hit, const f3 &dir) {
            const bool valid = pred && bl.valid;
            if (valid)
            {
                f3 color = f3{0.0f, 0.0f, 0.0f};
                if (!found && (p.pc.flags & PROSPER_PC_FLAG_IBL))
                {
                    if constexpr (COUNT) cnt.skyLookups++;
                    add_bounce(p.pc.flags, color, sample_skybox(s, dir), 0u); // throughput is (1,1,1)
                }
#if PPT_STREAM_NT & 4
                __builtin_nontemporal_store(ppt_f4v{color.x, color.y, color.z, 0.0f}, reinterpret_cast<ppt_f4v *>(w.color + bl.slot));
#else
                w.color[bl.slot] = make_float4(color.x, color.y, color.z, 0.0f);
#endif
            }
            const bool isHit = valid && found;
            uint32_t total;
            const uint32_t pos = nHit + wave_rank(isHit, total);
            if (isHit)
            {
                sst(&w.hit[id.base + pos], make_uint4(hit.drawInstance, hit.primitive, asu(hit.bary.x), asu(hit.bary.y)));
                sst(&w.hitIdx[id.base + pos], k);
            }
            nHit += total;
        };
        // Camera rays of a batch are coherent: 64 of them in lockstep keep ~96 % of the lanes busy on C2, so they skip the
        // stream scheduler and its bookkeeping.  Running them through trace_stream() instead was measured slower on
        // every configuration (profiles/r02_scheduler_experiments.txt: C2 365 -> 572 us, C3 1551 -> 1922, C4 7119 ->
        // 9805, FlightHelmet 1438 -> 1562).
        // ... or, banded (WavefrontBuffers::bandSegments): batch number b * S + j of the batches of ONE band of tiles, S the
        // segments that share the band and j this segment's place among them - the same striding inside the band.
        uint32_t dealStride = w.nSeg, dealFirst = id.seg, tileBase = 0u, tileCount = w.pixelsPadded >> 6;
        if (w.bandSegments != 0u)
        {
            const uint32_t band = id.seg / w.bandSegments;
            const uint32_t firstSeg = band * w.bandSegments;
            dealStride = w.nSeg - firstSeg < w.bandSegments ? w.nSeg - firstSeg : w.bandSegments;
            dealFirst = id.seg - firstSeg;
            tileBase = w.bandTile[band];
            tileCount = w.bandTile[band + 1u] - tileBase;
        }
        for (uint32_t k0 = 0; k0 < w.segLen; k0 += 64u)
        {
            const uint32_t k = k0 + lane;
            bl = batch_lane(w, p, (k0 >> 6) * dealStride + dealFirst, lane, tileBase, tileCount);
            const StreamRay r = fetch(k);
            Hit hit;
            bool found;
            if constexpr (LDS_SCENE)
                found = trace_in<false, COUNT>(lg, s, r.o, r.d, r.tMin, r.tMax, r.seed, stack, hit, cnt);
            else
                found = trace_in<false, COUNT>(gg, s, r.o, r.d, r.tMin, r.tMax, r.seed, stack, hit, cnt);
            commit(true, k, found, hit, r.d);
        }
    }
    if (lane == 0) w.segHits[id.seg] = nHit;
    flush_counters<COUNT>(cnt, counters);
}

// ------------------------------------------------------------------------------------------
// extend: traceClosest of bounce >= 1
// ------------------------------------------------------------------------------------------

// Sparse segments - an EXPERIMENT (debug option mergeLimit, -DPPT_EXPERIMENTS builds only; profiles/r03_sparse_segments.txt).  A wave
// owns a segment, and what survives a stage stays in it: on a sparse image (FlightHelmet: one camera ray in ten hits
// anything) the later stages run waves whose streams hold a few dozen to a few hundred rays (52 % of wf_trace's node steps
// ran with <= 8 lanes there).  The idea: when the four segments of a workgroup together hold few enough rays, ONE of its
// waves traces all of them as one stream - RayMap turns a stream position into the record's offset from the first
// segment's base (the four segments are contiguous), the hits are compacted into the first segment, and from then on the
// paths of the group live there.  Which wave traces a ray does not change its hit, and a path's shadow ray and next
// closest-hit ray are still traced by the same wave in that order (both lists are merged, or neither): same pixels
// (tested).  Measured: fuller waves, and SLOWER the more is merged (FlightHelmet wf_trace 364 -> 445 / 604 us per launch at
// a limit of 256 / 1024 rays, the step 1.88 -> 1.98 / 2.14 ms; also the 1/8 rank share of C2, 0.32 -> 0.41 ms): with a few
// hundred rays per wave the kernel is bound by the latency of each wave's own chain of steps, and a quarter of the waves
// hide a quarter of it.  A sparse stage wants MORE waves, not fuller ones.
struct RayMap
{
    uint32_t c0, c01, c012; // rays in the first one / two / three segments; identity: c0 = ~0
    uint32_t segLen;
    __device__ __forceinline__ uint32_t at(uint32_t i) const
    {
        return i < c0 ? i : (i < c01 ? segLen + (i - c0) : (i < c012 ? 2u * segLen + (i - c01) : 3u * segLen + (i - c012)));
    }
};

// One wave's extend work: traces `n` live rays (buffer set `cur`) of its segment - or of its workgroup's four, through
// `map` - and compacts the hits into the segment `id`; returns their number.
template <bool COUNT, class Geom, class Tracer>
__device__ __forceinline__ uint32_t extend_segment(
    const Geom &g, const DeviceScene &s, const RenderParams &p, const WavefrontBuffers &w, const SegmentId &id, uint32_t n,
    const RayMap &map, uint32_t bounce, uint32_t cur, const Tracer &tracer, LaneCounters &cnt)
{
    const float4 *__restrict__ rayA = w.rayA[cur];
    const float4 *__restrict__ rayB = w.rayB[cur];
    uint32_t nHit = 0;
    auto fetch = [&](uint32_t i) {
        const uint32_t k = map.at(i);
        const float4 a = sld(&rayA[id.base + k]);
        const float4 b = sld(&rayB[id.base + k]);
        if constexpr (COUNT) cnt.closestRays++;
        StreamRay r;
        r.o = xyz(a);
        r.d = xyz(b);
        r.tMin = 0.0f;
        r.tMax = kInf;
        r.seed = asu(a.w);
        return r;
    };
    auto commit = [&](bool pred, uint32_t i, bool found, const Hit &hit, const f3 &dir) {
        const uint32_t k = map.at(i);
        if (pred && !found && (p.pc.flags & PROSPER_PC_FLAG_IBL))
        {
            if constexpr (COUNT) cnt.skyLookups++;
            const float4 t = sld(&w.pathT[cur][id.base + k]);
            add_to_slot(w.color, asu(t.w) & kSlotMask, p.pc.flags, xyz(t) * sample_skybox(s, dir), bounce);
        }
        const bool isHit = pred && found;
        uint32_t total;
        const uint32_t pos = nHit + wave_rank(isHit, total);
        if (isHit)
        {
            sst(&w.hit[id.base + pos], make_uint4(hit.drawInstance, hit.primitive, asu(hit.bary.x), asu(hit.bary.y)));
            sst(&w.hitIdx[id.base + pos], k);
        }
        nHit += total;
    };
    tracer.template run<false, COUNT>(g, s, n, 0.0f, cnt, fetch, commit);
    return nHit;
}

// ------------------------------------------------------------------------------------------
// shade
// ------------------------------------------------------------------------------------------

// The small scene tables every hit walks through - draw instances, instance transforms, materials and the
// light lists - staged into LDS per workgroup when they fit: the shade kernel streams ~200 B of path state
// per hit through the 16 KB vector L1, which keeps evicting them otherwise (an L2 round trip per table on the
// dependent chain hit -> instance -> material / transform -> light).
constexpr uint32_t kLdsTableBytes = 16u * 1024u;
__host__ __device__ inline uint32_t shade_table_bytes(const DeviceScene &s, bool withLights)
{
    const uint32_t scene = s.drawInstanceCount * (uint32_t)sizeof(prosper_DrawInstance) +
                           s.modelInstanceCount * (uint32_t)sizeof(prosper_ModelInstanceTransforms) +
                           s.materialCount * (uint32_t)(sizeof(prosper_MaterialData) + sizeof(MaterialPack)) +
                           (uint32_t)sizeof(prosper_DirectionalLightParameters) + 64u;
    const uint32_t lights = s.pointLightCount * (uint32_t)sizeof(prosper_PointLight) + s.spotLightCount * (uint32_t)sizeof(prosper_SpotLight);
    return scene + (withLights ? lights : 0u);
}
// copies `bytes` (a multiple of 4) from global memory to the next 16-byte aligned LDS offset
__device__ __forceinline__ const void *stage_table(uint32_t *lds, uint32_t &offsetWords, const void *src, uint32_t bytes)
{
    offsetWords = (offsetWords + 3u) & ~3u;
    uint32_t *dst = lds + offsetWords;
    const uint32_t words = bytes / 4u;
    const uint32_t *from = static_cast<const uint32_t *>(src);
    for (uint32_t i = threadIdx.x; i < words; i += blockDim.x) dst[i] = from[i];
    offsetWords += words;
    return dst;
}

// LDS_TABLES: 0 = every table from global memory, 1 = all of them staged in LDS, 2 = all but the light lists (a
// compile-time choice: a pointer that may be either makes every access through it a flat one)
template <bool COUNT, int LDS_TABLES, bool BATCHED_TEXTURES>
__global__ __launch_bounds__(256, PPT_SHADE_WPE) void wf_shade(
    DeviceScene s, RenderParams p, WavefrontBuffers w, uint32_t bounce, uint32_t cur, uint32_t lastBounce,
    unsigned long long *__restrict__ counters)
{
    __shared__ uint32_t ldsTables[LDS_TABLES != 0 ? kLdsTableBytes / 4u : 4u];
    if constexpr (LDS_TABLES != 0)
    {
        uint32_t off = 0;
        s.drawInstances = static_cast<const prosper_DrawInstance *>(
            stage_table(ldsTables, off, s.drawInstances, s.drawInstanceCount * (uint32_t)sizeof(prosper_DrawInstance)));
        s.modelInstanceTransforms = static_cast<const prosper_ModelInstanceTransforms *>(stage_table(
            ldsTables, off, s.modelInstanceTransforms, s.modelInstanceCount * (uint32_t)sizeof(prosper_ModelInstanceTransforms)));
        s.materials = static_cast<const prosper_MaterialData *>(
            stage_table(ldsTables, off, s.materials, s.materialCount * (uint32_t)sizeof(prosper_MaterialData)));
        s.materialPacks = static_cast<const MaterialPack *>(
            stage_table(ldsTables, off, s.materialPacks, s.materialCount * (uint32_t)sizeof(MaterialPack)));
        s.directionalLight = static_cast<const prosper_DirectionalLightParameters *>(
            stage_table(ldsTables, off, s.directionalLight, (uint32_t)sizeof(prosper_DirectionalLightParameters)));
        // only the first `count` lights of each list are ever indexed (sample_light); a thousand lights (C4: 40 KB) do not
        // fit beside the rest and stay in global memory - the instance / transform / material chain still comes from LDS
        if constexpr (LDS_TABLES == 1)
        {
            s.pointLights = static_cast<const prosper_PointLightsBuffer *>(
                stage_table(ldsTables, off, s.pointLights, s.pointLightCount * (uint32_t)sizeof(prosper_PointLight)));
            s.spotLights = static_cast<const prosper_SpotLightsBuffer *>(
                stage_table(ldsTables, off, s.spotLights, s.spotLightCount * (uint32_t)sizeof(prosper_SpotLight)));
        }
        __syncthreads();
    }
    const SegmentId id = my_segment(w);
    if (!id.valid) return;
    const uint32_t lane = lane_id();
    const uint32_t nxt = cur ^ 1u;
    const bool debugDraw =
        p.pc.drawType != PROSPER_DRAW_TYPE_DEFAULT && p.pc.drawType != PROSPER_DRAW_TYPE_MESHLET_ID;

    LaneCounters cnt = {};
    const uint32_t n = w.segHits[id.seg];
    uint32_t nShadow = 0, nNext = 0;
    for (uint32_t j0 = 0; j0 < n; j0 += 64u)
    {
        const uint32_t j = j0 + lane;
        bool wantShadow = false, wantNext = false;
        // shadow-ray record
        f3 shP = {}, shL = {}, shC1 = {};
        float shDist = 0.0f;
        uint32_t shSeed = 0, nanMask = 0;
        // next-ray record
        f3 nO = {}, nD = {}, nT = {};
        Rng rng = {};
        uint32_t slot = 0;
        if (j < n)
        {
            const uint32_t i = sld(&w.hitIdx[id.base + j]);
            const uint4 h = sld(&w.hit[id.base + j]);
            const float4 b = sld(&w.rayB[cur][id.base + i]);
            // bounce 0: no throughput record - it is (1, 1, 1) - and the slot comes from the camera paths' own array
            const float4 t = bounce == 0u ? make_float4(1.0f, 1.0f, 1.0f, asf(sld(&w.cameraSlot[id.base + i]))) : sld(&w.pathT[cur][id.base + i]);
            const uint2 r = sld(&w.pathR[cur][id.base + i]);
            slot = asu(t.w) & kSlotMask;
            rng = Rng{asu(b.w), r.x, r.y};
            const f3 throughput = xyz(t);
            Hit hit;
            hit.drawInstance = h.x;
            hit.primitive = h.y;
            hit.bary = f2{asf(h.z), asf(h.w)};
            hit.t = 0.0f;
            const Surface sf = evaluate_surface<COUNT, BATCHED_TEXTURES>(s, xyz(b), hit, cnt);
            if (debugDraw)
            {
                const f3 c = debug_color(s, p.pc.drawType, hit, sf);
                w.color[slot] = make_float4(c.x, c.y, c.z, 0.0f); // `color =`, not `+=` (main.rgen:262)
            }
            else
            {
                f3 l, irradiance;
                float d;
                if (prepare_direct_lighting<COUNT>(s, sf, throughput, rng, l, d, irradiance, cnt))
                {
                    if constexpr (COUNT) cnt.shadowRays++;
                    shSeed = pcg(rng.x ^ rng.y);
                    shP = sf.positionWS;
                    shL = l;
                    shDist = d;
                    const f3 brdf = eval_brdf_times_nol(l, sf);
                    shC1 = direct_lighting_value(s, throughput, irradiance, brdf, 1.0f);
                    // what the reference adds when the light is occluded is (t * (irr*0*n)) * brdf:
                    // +-0 unless a factor is non-finite; remember which channels come out NaN
                    const f3 c0 = direct_lighting_value(s, throughput, irradiance, brdf, 0.0f);
                    nanMask = (c0.x != c0.x ? 1u : 0u) | (c0.y != c0.y ? 2u : 0u) | (c0.z != c0.z ? 4u : 0u);
                    // a ray that cannot change the radiance bits is not queued (zero-radiance light,
                    // outside the spot cone, beyond the light's range)
                    wantShadow = shadow_ray_matters(shC1, c0);
                }
                if (!lastBounce)
                {
                    f3 rd;
                    f3 tp = throughput;
                    importance_sample_bounce(sf, rng, tp, rd);
                    bool alive = p.traceDeadPaths || !throughput_is_zero(tp);
                    if (alive && bounce > p.pc.rouletteStartBounce)
                        alive = !(rng.rnd01() < fmax_(0.05f, 1.0f - max3(tp)));
                    if (alive)
                    {
                        wantNext = true;
                        nO = offset_ray(sf.positionWS, sf.normalWS);
                        nD = rd;
                        nT = tp;
                    }
                }
            }
        }
        uint32_t total;
        uint32_t pos = nShadow + wave_rank(wantShadow, total);
        if (wantShadow)
        {
            sst(&w.shA[id.base + pos], make_float4(shP.x, shP.y, shP.z, asf(shSeed)));
            sst(&w.shB[id.base + pos], make_float4(shL.x, shL.y, shL.z, shDist));
            sst(&w.shC[id.base + pos], make_float4(shC1.x, shC1.y, shC1.z, asf(slot | (nanMask << 28))));
        }
        nShadow += total;
        pos = nNext + wave_rank(wantNext, total);
        if (wantNext)
        {
            sst(&w.rayA[nxt][id.base + pos], make_float4(nO.x, nO.y, nO.z, asf(pcg(rng.x ^ rng.z))));
            sst(&w.rayB[nxt][id.base + pos], make_float4(nD.x, nD.y, nD.z, asf(rng.x)));
            sst(&w.pathT[nxt][id.base + pos], make_float4(nT.x, nT.y, nT.z, asf(slot)));
            sst(&w.pathR[nxt][id.base + pos], make_uint2(rng.y, rng.z));
        }
        nNext += total;
    }
    if (lane == 0)
    {
        w.segShadow[id.seg] = nShadow;
        w.segRays[id.seg] = nNext;
    }
    flush_counters<COUNT>(cnt, counters);
}

// ------------------------------------------------------------------------------------------
// shadow
// ------------------------------------------------------------------------------------------

// One wave's shadow work: shadow() for the `n` shadow rays shade queued in its segment (or in its workgroup's four,
// through `map`); adds the direct term of bounce `bounce` where the light is visible.
template <bool COUNT, class Geom, class Tracer>
__device__ __forceinline__ void shadow_segment(
    const Geom &g, const DeviceScene &s, const RenderParams &p, const WavefrontBuffers &w, const SegmentId &id, uint32_t n,
    const RayMap &map, uint32_t bounce, const Tracer &tracer, LaneCounters &cnt)
{
    auto fetch = [&](uint32_t i) {
        const uint32_t k = map.at(i);
        const float4 a = sld(&w.shA[id.base + k]);
        const float4 b = sld(&w.shB[id.base + k]);
        StreamRay r;
        r.o = xyz(a);
        r.d = xyz(b);
        r.tMin = 0.1f; // main.rgen:217
        r.tMax = b.w;
        r.seed = asu(a.w);
        return r;
    };
    auto commit = [&](bool pred, uint32_t i, bool occluded, const Hit &, const f3 &) {
        if (pred)
        {
            const float4 c = sld(&w.shC[id.base + map.at(i)]);
            const uint32_t packed = asu(c.w);
            const uint32_t nanMask = packed >> 28;
            if (!occluded || nanMask)
            {
                f3 v = xyz(c);
                if (occluded)
                {
                    const float nan = __builtin_nanf("");
                    v = f3{(nanMask & 1u) ? nan : 0.0f, (nanMask & 2u) ? nan : 0.0f, (nanMask & 4u) ? nan : 0.0f};
                }
                add_to_slot(w.color, packed & kSlotMask, p.pc.flags, v, bounce);
            }
        }
    };
    tracer.template run<true, COUNT>(g, s, n, 0.1f, cnt, fetch, commit);
}

// What a wave of wf_trace traces: its own segment's lists, or - sparse segments, see RayMap - all four of its workgroup's
// (the leader) or nothing (the other three).
struct TraceWork
{
    SegmentId id;      // where the records are read from (base) and the hits compacted into
    uint32_t nShadow, nRays;
    RayMap shadowMap, rayMap;
    bool merged, leader;
};
__device__ __forceinline__ TraceWork trace_work(const RenderParams &p, const WavefrontBuffers &w, const SegmentId &id, bool doExtend)
{
    TraceWork t;
    t.id = id;
    t.nShadow = w.segShadow[id.seg];
    t.nRays = doExtend ? w.segRays[id.seg] : 0u;
    t.shadowMap = t.rayMap = RayMap{~0u, ~0u, ~0u, w.segLen};
    t.merged = t.leader = false;
#ifdef PPT_EXPERIMENTS
    if (p.mergeLimit == 0u) return t;
    const uint32_t limit = p.mergeLimit < w.segLen ? p.mergeLimit : w.segLen;
    const uint32_t seg0 = id.seg & ~3u;
    uint32_t cs[4], cr[4];
#pragma unroll
    for (uint32_t k = 0; k < 4u; ++k)
    {
        const bool valid = seg0 + k < w.nSeg;
        cs[k] = valid ? w.segShadow[seg0 + k] : 0u;
        cr[k] = (valid && doExtend) ? w.segRays[seg0 + k] : 0u;
    }
    const uint32_t totalS = cs[0] + cs[1] + cs[2] + cs[3], totalR = cr[0] + cr[1] + cr[2] + cr[3];
    if (totalS > limit || totalR > limit) return t;
    t.merged = true;
    // the leading wave rotates with the workgroup (the dispatcher deals a workgroup's waves over the SIMDs in order)
    uint32_t leader = (seg0 >> 2) & 3u;
    if (seg0 + leader >= w.nSeg) leader = 0u;
    if ((threadIdx.x >> 6) == leader)
    {
        t.leader = true;
        t.id.seg = seg0;
        t.id.base = seg0 * w.segLen;
        t.nShadow = totalS;
        t.nRays = totalR;
        t.shadowMap = RayMap{cs[0], cs[0] + cs[1], cs[0] + cs[1] + cs[2], w.segLen};
        t.rayMap = RayMap{cr[0], cr[0] + cr[1], cr[0] + cr[1] + cr[2], w.segLen};
    }
    else
    {
        t.nShadow = 0u;
        t.nRays = 0u;
    }
#endif
    return t;
}
// the hit counts of the segments a wave answers for, after its extend work: its own; merged, the leader's hits are the
// first segment's and every other segment of the group is empty from here on
__device__ __forceinline__ void store_hit_counts(const WavefrontBuffers &w, const SegmentId &own, const TraceWork &t, uint32_t nHit)
{
    if (lane_id() != 0u) return;
    if (!t.merged)
    {
        w.segHits[own.seg] = nHit;
        return;
    }
    const uint32_t seg0 = own.seg & ~3u;
    if (t.leader) w.segHits[seg0] = nHit;
    if (own.seg != seg0) w.segHits[own.seg] = 0u;
}

// Shadow rays of bounce `bounce` and (unless it was the last bounce) the closest-hit rays of bounce
// `bounce + 1` in ONE launch: both belong to the same wave-owned segment, neither depends on the
// other, and one launch has one tail instead of two.  The reference adds the direct term of bounce b
// before the sky term of bounce b + 1 (main.rgen:266 then :252); both may touch the same radiance
// slot from different lanes of this wave, hence shadow first, a fence, then extend.
template <bool COUNT, int STACK, bool LDS_SCENE>
__global__ __launch_bounds__(256, PPT_TRACE_WPE(STACK)) void wf_trace(
    DeviceScene s, RenderParams p, WavefrontBuffers w, uint32_t bounce, uint32_t nextCur, uint32_t doExtend,
    uint32_t nodeCount, uint32_t triCount, int32_t *__restrict__ stackOverflow,
    unsigned long long *__restrict__ counters)
{
    __shared__ int32_t ldsStack[STACK * 256];
    __shared__ float4 ldsScene[LDS_SCENE ? kLdsSceneFloat4s : 1];
#ifdef PPT_EXPERIMENT_TRACE_LDS_HALF
    LdsGeomHalf lg = {}; // (A/B: the 80-byte node image, as until round 4)
    if constexpr (LDS_SCENE) lg = stage_scene_in_lds_half(s, ldsScene, nodeCount, triCount);
#else
    LdsGeom lg = {};
    if constexpr (LDS_SCENE) lg = stage_scene_in_lds(s, ldsScene, nodeCount, triCount);
#endif
    const GlobalGeom gg{s.nodes, s.triangles};
    const SegmentId id = my_segment(w);
    if (!id.valid) return;
    const StreamTracer stack{TraversalStack{(lds_int32 *)ldsStack + (threadIdx.x >> 6) * (STACK * 64u) + lane_id(),
                                            stackOverflow + blockIdx.x * 256u + threadIdx.x, (uint32_t)STACK,
                                            gridDim.x * 256u, 64u, LDS_SCENE}};
    LaneCounters cnt = {};
    const TraceWork t = trace_work(p, w, id, doExtend != 0u);
#ifdef PPT_EXPERIMENT_LDS_SCENE_LOCKSTEP
    const LockstepTracer lockstep{stack.stack};
#endif
    if constexpr (LDS_SCENE)
#ifdef PPT_EXPERIMENT_LDS_SCENE_LOCKSTEP
        shadow_segment<COUNT>(lg, s, p, w, t.id, t.nShadow, t.shadowMap, bounce, lockstep, cnt);
#else
        shadow_segment<COUNT>(lg, s, p, w, t.id, t.nShadow, t.shadowMap, bounce, stack, cnt);
#endif
    else
        shadow_segment<COUNT>(gg, s, p, w, t.id, t.nShadow, t.shadowMap, bounce, stack, cnt);
    if (doExtend)
    {
        // same wave, same CU: a workgroup-scope fence (s_waitcnt vmcnt(0)) orders the shadow phase's
        // radiance stores before the extend phase's loads; the vector L1 is write-through
        __threadfence_block();
        uint32_t nHit;
        if constexpr (LDS_SCENE)
#ifdef PPT_EXPERIMENT_LDS_SCENE_LOCKSTEP
            nHit = extend_segment<COUNT>(lg, s, p, w, t.id, t.nRays, t.rayMap, bounce + 1u, nextCur, lockstep, cnt);
#else
            nHit = extend_segment<COUNT>(lg, s, p, w, t.id, t.nRays, t.rayMap, bounce + 1u, nextCur, stack, cnt);
#endif
        else
            nHit = extend_segment<COUNT>(gg, s, p, w, t.id, t.nRays, t.rayMap, bounce + 1u, nextCur, stack, cnt);
        store_hit_counts(w, id, t, nHit);
    }
    flush_counters<COUNT>(cnt, counters);
}

#ifdef PPT_EXPERIMENTS
// wf_trace with the wave's rays in an LDS pool of P slots (pt_trace_pool.hpp) instead of one per lane.
template <bool COUNT, uint32_t P, uint32_t S, uint32_t B, bool LDS_SCENE>
__global__ __launch_bounds__(256, 3) void wf_trace_pool(
    DeviceScene s, RenderParams p, WavefrontBuffers w, uint32_t bounce, uint32_t nextCur, uint32_t doExtend,
    uint32_t nodeCount, uint32_t triCount, int32_t *__restrict__ scratch, uint32_t overflowEntries,
    unsigned long long *__restrict__ counters)
{
    __shared__ uint32_t ldsPool[RayPool<P, S>::kLdsDwords * 4u];
    __shared__ float4 ldsScene[LDS_SCENE ? kLdsSceneFloat4s : 1];
    LdsGeomHalf lg = {};
    if constexpr (LDS_SCENE) lg = stage_scene_in_lds_half(s, ldsScene, nodeCount, triCount);
    const GlobalGeom gg{s.nodes, s.triangles};
    const SegmentId id = my_segment(w);
    if (!id.valid) return;
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const PoolTracer<P, S, B> tracer{
        RayPool<P, S>::carve(ldsPool, wave, reinterpret_cast<uint32_t *>(scratch), blockIdx.x * 4u + wave, overflowEntries)};
    LaneCounters cnt = {};
    RenderParams own = p;
    own.mergeLimit = 0u; // (the pool experiment keeps every wave on its own segment)
    const TraceWork t = trace_work(own, w, id, doExtend != 0u);
    if constexpr (LDS_SCENE)
        shadow_segment<COUNT>(lg, s, p, w, t.id, t.nShadow, t.shadowMap, bounce, tracer, cnt);
    else
        shadow_segment<COUNT>(gg, s, p, w, t.id, t.nShadow, t.shadowMap, bounce, tracer, cnt);
    if (doExtend)
    {
        __threadfence_block();
        uint32_t nHit;
        if constexpr (LDS_SCENE)
            nHit = extend_segment<COUNT>(lg, s, p, w, t.id, t.nRays, t.rayMap, bounce + 1u, nextCur, tracer, cnt);
        else
            nHit = extend_segment<COUNT>(gg, s, p, w, t.id, t.nRays, t.rayMap, bounce + 1u, nextCur, tracer, cnt);
        store_hit_counts(w, id, t, nHit);
    }
    flush_counters<COUNT>(cnt, counters);
}

#endif // PPT_EXPERIMENTS

// ------------------------------------------------------------------------------------------
// accumulate: main.rgen:285-298 over the frames of the batch, in order
// ------------------------------------------------------------------------------------------

template <bool COUNT>
__global__ __launch_bounds__(256) void wf_accumulate(
    RenderParams p, WavefrontBuffers w, float4 *__restrict__ hdr, unsigned long long *__restrict__ counters)
{
    const uint32_t rem = blockIdx.x * blockDim.x + threadIdx.x;
    LaneCounters cnt = {};
    if (rem < w.pixelsPadded)
    {
        const SlotPixel sp = decode_slot(w, p, rem);
        if (sp.valid)
        {
            float4 *texel = hdr + (size_t)sp.py * p.localWidth + sp.lx;
            float4 history = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            for (uint32_t f = 0; f < p.frameCount; ++f)
            {
#if PPT_STREAM_NT & 4
                const ppt_f4v cv = __builtin_nontemporal_load(reinterpret_cast<const ppt_f4v *>(w.color + (size_t)f * w.pixelsPadded + rem)); // (its last use)
                const float4 c = make_float4(cv.x, cv.y, cv.z, cv.w);
#else
                const float4 c = w.color[(size_t)f * w.pixelsPadded + rem];
#endif
                const bool skip = (f == 0 && (p.pc.flags & PROSPER_PC_FLAG_SKIP_HISTORY)) ||
                                  !(p.pc.flags & PROSPER_PC_FLAG_ACCUMULATE);
                if (skip)
                    history = make_float4(c.x, c.y, c.z, 1.0f);
                else
                {
                    if (f == 0) history = *texel;
                    if constexpr (COUNT) cnt.historyReads++;
                    const float hc = history.w + 1.0f;
                const float invHc = 1.0f / hc;
                history = make_float4(
                    __builtin_fmaf(c.x - history.x, invHc, history.x), __builtin_fmaf(c.y - history.y, invHc, history.y),
                    __builtin_fmaf(c.z - history.z, invHc, history.z), hc);
                }
                if constexpr (COUNT) cnt.pixelsWritten++;
            }
            *texel = history;
        }
    }
    flush_counters<COUNT>(cnt, counters);
}

// ------------------------------------------------------------------------------------------
// host-side sequencing
// ------------------------------------------------------------------------------------------

#ifdef PPT_EXPERIMENTS
// ray-pool variants of wf_trace (debug option poolVariant = index): slots per wave, LDS stack entries per slot, batches per
// step.  An experiment kept for its measurements (profiles/r02_pool_experiment.txt): fewer, fuller instructions, but slower.
#define PPT_POOL_VARIANTS(X) X(1, 128, 8, 1) X(2, 96, 10, 1) X(3, 128, 8, 2)

template <bool COUNT, bool LDS_SCENE>
static void launch_trace_pool(
    uint32_t variant, dim3 grid, dim3 block, hipStream_t stream, const DeviceScene &s, const RenderParams &p,
    const WavefrontBuffers &w, uint32_t b, uint32_t nextCur, uint32_t doExtend, uint32_t nodeCount, uint32_t triCount,
    int32_t *scratch, uint32_t overflowEntries, unsigned long long *counters)
{
    switch (variant)
    {
#define PPT_POOL_CASE(index, P, S, B)                                                                                  \
    case index:                                                                                                        \
        hipLaunchKernelGGL(                                                                                            \
            (wf_trace_pool<COUNT, P, S, B, LDS_SCENE>), grid, block, 0, stream, s, p, w, b, nextCur, doExtend, nodeCount,  \
            triCount, scratch, overflowEntries, counters);                                                             \
        break;
        PPT_POOL_VARIANTS(PPT_POOL_CASE)
#undef PPT_POOL_CASE
    default:
        break;
    }
}

#endif // PPT_EXPERIMENTS

template <bool COUNT, int STACK, bool LDS_SCENE>
static void enqueue_wavefront(
    const DeviceScene &s, const RenderParams &p, unsigned long long *counters, const WavefrontBuffers &w,
    const WavefrontPlan &plan, uint32_t nodeCount, uint32_t triCount, int32_t *stackOverflow, LaunchTimer *timer,
    hipStream_t stream)
{
    unsigned long long *cGen = counters + kStageGenerate * kCounterCount, *cShade = counters + kStageShade * kCounterCount,
                       *cTrace = counters + kStageTrace * kCounterCount;
    auto mark = [&](uint32_t st) {
        if (timer) timer->mark(st, stream);
    };
    const dim3 grid(((w.groupCount + 7u) / 8u) * 8u), block(256);
    uint32_t bounces = p.pc.maxBounces < PROSPER_RT_MAX_BOUNCES ? p.pc.maxBounces : PROSPER_RT_MAX_BOUNCES;
    const bool debugDraw =
        p.pc.drawType != PROSPER_DRAW_TYPE_DEFAULT && p.pc.drawType != PROSPER_DRAW_TYPE_MESHLET_ID;
    if (debugDraw && bounces > 1) bounces = 1; // every hit ends its path in the first shade

    mark(kStageGenerate);
    hipLaunchKernelGGL(
        (wf_generate_extend<COUNT, STACK, LDS_SCENE>), grid, block, 0, stream, s, p, w, nodeCount, triCount,
        stackOverflow, cGen);
    for (uint32_t b = 0; b < bounces; ++b)
    {
        const uint32_t cur = b & 1u;
        const uint32_t last = (b + 1u == bounces) ? 1u : 0u;
        mark(kStageShade);
        const int ldsTables = !plan.tablesInLds ? 0 : (shade_table_bytes(s, true) <= kLdsTableBytes ? 1 : 2);
        auto shade = [&](auto kernel) { hipLaunchKernelGGL(kernel, grid, block, 0, stream, s, p, w, b, cur, last, cShade); };
        if (s.batchedTextures)
            ldsTables == 1 ? shade(wf_shade<COUNT, 1, true>) : (ldsTables == 2 ? shade(wf_shade<COUNT, 2, true>) : shade(wf_shade<COUNT, 0, true>));
        else
            ldsTables == 1 ? shade(wf_shade<COUNT, 1, false>) : (ldsTables == 2 ? shade(wf_shade<COUNT, 2, false>) : shade(wf_shade<COUNT, 0, false>));
        if (!debugDraw)
        {
            mark(kStageTrace);
#ifdef PPT_EXPERIMENTS
            if (plan.poolVariant)
                launch_trace_pool<COUNT, LDS_SCENE>(
                    plan.poolVariant, grid, block, stream, s, p, w, b, cur ^ 1u, last ? 0u : 1u, nodeCount, triCount,
                    stackOverflow, plan.poolOverflowEntries, cTrace);
            else
#endif
                hipLaunchKernelGGL(
                    (wf_trace<COUNT, STACK, LDS_SCENE>), grid, block, 0, stream, s, p, w, b, cur ^ 1u, last ? 0u : 1u,
                    nodeCount, triCount, stackOverflow, cTrace);
        }
    }
}

uint32_t wavefront_grid_blocks(const WavefrontBuffers &w)
{
    const uint32_t groups = (w.nSeg + 3u) / 4u;
    return ((groups + 7u) / 8u) * 8u + 8u * kMaxChains; // + the per-chain rounding of a split launch
}

uint32_t wavefront_lds_stack_entries(uint32_t stackBound, uint32_t forced)
{
    // test/tuning hook (debug option ldsStackEntries = 16 | 24 | 32): forces a variant (deeper entries then live in
    // the global overflow array, which is always safe, only slower)
    if (forced == 16u || forced == 24u || forced == 32u) return forced;
    // Deeper trees keep 24 entries in LDS and spill the rest to the global overflow array: the 4-wide
    // tree rarely has more than ~2 entries per level in flight, and a 32-entry LDS stack (32 KB per
    // workgroup) costs more in occupancy than the rare overflow access does (C3: 4.41 vs 3.89 ms/launch).
    return stackBound <= 16u ? 16u : 24u;
}

// Which kernel variants a render of this scene takes (also reported by prosper_pt_get_scene_stats, so that a
// test of a variant can assert it is the one that ran).
bool wavefront_shade_tables_in_lds(const DeviceScene &s, bool disabled)
{
    return shade_table_bytes(s, false) <= kLdsTableBytes && !disabled;
}
bool wavefront_scene_in_lds(uint32_t ldsStackEntries, uint32_t stackBound, uint32_t nodeCount, uint32_t triCount, bool disabled)
{
    // (stackBound <= 16: the LDS-scene kernels have no global stack path at all - TraversalStack::ldsOnly)
    return ldsStackEntries == 16u && stackBound <= 16u && nodeCount * kLdsNodeStride + triCount * 3u <= kLdsSceneFloat4s && !disabled;
}

// The traversal kernels of a render and the global scratch they index: the lane-owned kernels keep `overflowEntries`
// stack entries per lane there, a ray-pool wf_trace its slots' hit / candidate records and deeper stack entries.
WavefrontPlan wavefront_plan(
    uint32_t stackBound, uint32_t nodeCount, uint32_t triCount, const DeviceScene &s, const WavefrontOptions &opt)
{
    WavefrontPlan plan = {};
    plan.ldsStackEntries = wavefront_lds_stack_entries(stackBound, opt.ldsStackEntries);
    plan.overflowEntries = stackBound > plan.ldsStackEntries ? stackBound - plan.ldsStackEntries : 0u;
    plan.sceneInLds = wavefront_scene_in_lds(plan.ldsStackEntries, stackBound, nodeCount, triCount, opt.noLdsScene);
    plan.tablesInLds = wavefront_shade_tables_in_lds(s, opt.noLdsTables);
    uint32_t poolDwords = 0;
#ifdef PPT_EXPERIMENTS
    plan.hipGraph = opt.hipGraph;
    switch (opt.poolVariant)
    {
#define PPT_POOL_CASE(index, P, S, B)                                                                                  \
    case index:                                                                                                        \
        plan.poolVariant = index;                                                                                      \
        plan.poolOverflowEntries = stackBound > S ? stackBound - S : 0u;                                               \
        poolDwords = 4u * RayPool<P, S>::scratch_dwords(plan.poolOverflowEntries);                                     \
        break;
        PPT_POOL_VARIANTS(PPT_POOL_CASE)
#undef PPT_POOL_CASE
    default:
        break;
    }
#endif
    const uint32_t laneDwords = plan.overflowEntries * 256u;
    plan.scratchDwordsPerBlock = laneDwords > poolDwords ? laneDwords : poolDwords;
    return plan;
}

template <bool COUNT>
static void enqueue_for_stack(
    const DeviceScene &s, const RenderParams &p, unsigned long long *counters, const WavefrontBuffers &w,
    const WavefrontPlan &plan, int32_t *stackOverflow, uint32_t nodeCount, uint32_t triCount, LaunchTimer *timer,
    hipStream_t stream)
{
    // a scene of a few KB is traversed out of LDS
    if (plan.sceneInLds)
        enqueue_wavefront<COUNT, 16, true>(s, p, counters, w, plan, nodeCount, triCount, stackOverflow, timer, stream);
    else if (plan.ldsStackEntries == 16u)
        enqueue_wavefront<COUNT, 16, false>(s, p, counters, w, plan, nodeCount, triCount, stackOverflow, timer, stream);
    else if (plan.ldsStackEntries == 24u)
        enqueue_wavefront<COUNT, 24, false>(s, p, counters, w, plan, nodeCount, triCount, stackOverflow, timer, stream);
    else
        enqueue_wavefront<COUNT, 32, false>(s, p, counters, w, plan, nodeCount, triCount, stackOverflow, timer, stream);
}

void launch_render_wavefront(
    const DeviceScene &s, const RenderParams &p, float4 *hdr, unsigned long long *counters, const WavefrontBuffers &w,
    const WavefrontPlan &plan, int32_t *stackOverflow, uint32_t nodeCount, uint32_t triCount, bool countWork,
    LaunchTimer *timer, const WavefrontChains &chains, hipStream_t stream)
{
    if (w.nSeg == 0) return;
    static_assert(kTraversalStackDepth == 32, "largest LDS stack variant");
    const uint32_t groups = (w.nSeg + 3u) / 4u;
    // Two chains only pay when each still fills the machine a few times over (>= 1024 workgroups each).
    uint32_t parts = chains.count < kMaxChains ? chains.count : kMaxChains;
    while (parts > 1u && groups < 256u * parts) --parts;
    const bool ownStreams = parts > 1u || chains.detached;
    const uint32_t per = (groups + parts - 1u) / parts;
    // (detached chains - frames in flight - do not fork from the caller's stream: they wait for their slot's previous user)
    const bool forked = parts > 1u && !chains.detached;
    if (forked) (void)hipEventRecord(chains.fork, stream);
    uint32_t blocksBefore = 0;
    for (uint32_t i = 0; i < parts; ++i)
    {
        WavefrontBuffers part = w;
        part.groupBase = i * per;
        part.groupCount = (part.groupBase + per <= groups) ? per : groups - part.groupBase;
        hipStream_t cs = ownStreams ? chains.streams[i] : stream;
        LaunchTimer *ct = ownStreams ? chains.timers[i] : timer;
        if (forked) (void)hipStreamWaitEvent(cs, chains.fork, 0);
        if (chains.detached && chains.after) (void)hipStreamWaitEvent(cs, chains.after, 0);
        if (ownStreams && chains.scene) (void)hipStreamWaitEvent(cs, chains.scene, 0);
        if (ownStreams && chains.lights) (void)hipStreamWaitEvent(cs, chains.lights, 0);
        if (ownStreams && chains.materials) (void)hipStreamWaitEvent(cs, chains.materials, 0);
        // every chain gets its own region of the stack-overflow array: its kernels index it by their own
        // blockIdx / gridDim (TraversalStack), and the chains run concurrently
        int32_t *ovf = stackOverflow ? stackOverflow + (size_t)blocksBefore * plan.scratchDwordsPerBlock : nullptr;
        if (countWork)
            enqueue_for_stack<true>(s, p, counters, part, plan, ovf, nodeCount, triCount, ct, cs);
#ifdef PPT_EXPERIMENTS
        else if (plan.hipGraph && !ct)
        {
            // EXPERIMENT (debug option hipGraph, profiles/r03_hip_graph.txt): the chain's launches captured into a HIP
            // graph and submitted as one; captured and instantiated anew every time (the kernel arguments change with
            // every frame), so only the DEVICE side of the comparison means anything
            struct Launched
            {
                hipGraphExec_t exec = nullptr;
                hipEvent_t done = nullptr;
            };
            static thread_local Launched previous[8];
            static thread_local uint32_t turn = 0;
            Launched &mine = previous[turn++ & 7u];
            // an executable graph goes once ITS launch has finished: the event recorded behind it says so
            if (mine.exec)
            {
                (void)hipEventSynchronize(mine.done);
                (void)hipGraphExecDestroy(mine.exec);
                mine.exec = nullptr;
            }
            if (!mine.done) (void)hipEventCreateWithFlags(&mine.done, hipEventDisableTiming);
            hipGraph_t graph = nullptr;
            hipGraphExec_t exec = nullptr;
            bool ok = hipStreamBeginCapture(cs, hipStreamCaptureModeRelaxed) == hipSuccess;
            if (ok)
            {
                enqueue_for_stack<false>(s, p, counters, part, plan, ovf, nodeCount, triCount, nullptr, cs);
                // (a failed EndCapture still ends the capture: the stream is usable again either way)
                ok = hipStreamEndCapture(cs, &graph) == hipSuccess && graph != nullptr;
            }
            if (ok) ok = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) == hipSuccess;
            if (ok) ok = hipGraphLaunch(exec, cs) == hipSuccess;
            if (graph) (void)hipGraphDestroy(graph);
            if (ok)
            {
                mine.exec = exec;
                (void)hipEventRecord(mine.done, cs);
            }
            else
            {
                if (exec) (void)hipGraphExecDestroy(exec);
                (void)hipGetLastError();
                hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
                if (hipStreamIsCapturing(cs, &st) == hipSuccess && st != hipStreamCaptureStatusNone)
                {
                    hipGraph_t dropped = nullptr;
                    (void)hipStreamEndCapture(cs, &dropped);
                    if (dropped) (void)hipGraphDestroy(dropped);
                }
                enqueue_for_stack<false>(s, p, counters, part, plan, ovf, nodeCount, triCount, ct, cs);
            }
        }
#endif
        else
            enqueue_for_stack<false>(s, p, counters, part, plan, ovf, nodeCount, triCount, ct, cs);
        blocksBefore += ((part.groupCount + 7u) / 8u) * 8u;
        if (ownStreams)
        {
            if (ct) ct->close(cs);
            (void)hipEventRecord(chains.join[i], cs);
            (void)hipStreamWaitEvent(stream, chains.join[i], 0);
        }
    }
    if (timer) timer->mark(kStageAccumulate, stream);
    unsigned long long *cAcc = counters + kStageAccumulate * kCounterCount;
    const dim3 grid((w.pixelsPadded + 255u) / 256u), block(256);
    if (countWork)
        hipLaunchKernelGGL(wf_accumulate<true>, grid, block, 0, stream, p, w, hdr, cAcc);
    else
        hipLaunchKernelGGL(wf_accumulate<false>, grid, block, 0, stream, p, w, hdr, cAcc);
}

} // namespace ppt
