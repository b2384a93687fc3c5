// pt_tiling.cpp — see pt_tiling.hpp.  C-ABI: prosper_pt_comm_*, prosper_pt_gather_tiles, prosper_pt_gather_wait,
// prosper_pt_deinterleave_tiles (include/prosper_pt/prosper_pt.h, "multi-GPU").
//
// RCCL is loaded on first use (dlopen of librccl.so.1: the copy a host framework already mapped is reused, same
// SONAME), so a single-GPU user of the library never pays for it and the library has no link-time dependency on it.
#include "pt_tiling.hpp"

#include <dlfcn.h>
#include <rccl/rccl.h>

#include <cstring>
#include <mutex>
#include <string>

using namespace ppt;

namespace
{

struct Rccl
{
    void *handle = nullptr;
    decltype(&ncclGetUniqueId) getUniqueId = nullptr;
    decltype(&ncclCommInitRank) commInitRank = nullptr;
    decltype(&ncclCommDestroy) commDestroy = nullptr;
    decltype(&ncclGather) gather = nullptr;
    decltype(&ncclGroupStart) groupStart = nullptr;
    decltype(&ncclGroupEnd) groupEnd = nullptr;
    decltype(&ncclSend) send = nullptr;
    decltype(&ncclRecv) recv = nullptr;
    decltype(&ncclGetErrorString) getErrorString = nullptr;
    decltype(&ncclCommCount) commCount = nullptr;
    decltype(&ncclCommUserRank) commUserRank = nullptr;
    decltype(&ncclCommCuDevice) commCuDevice = nullptr;
    std::string error;
};

Rccl &rccl()
{
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        const char *names[] = {"librccl.so.1", "/opt/rocm/lib/librccl.so.1", "librccl.so"};
        for (const char *n : names)
        {
            r.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
            if (r.handle) break;
        }
        if (!r.handle)
        {
            r.error = std::string("RCCL could not be loaded: ") + dlerror();
            return;
        }
        auto sym = [&](const char *name) -> void * {
            void *p = dlsym(r.handle, name);
            if (!p && r.error.empty()) r.error = std::string("RCCL symbol missing: ") + name;
            return p;
        };
        r.getUniqueId = reinterpret_cast<decltype(r.getUniqueId)>(sym("ncclGetUniqueId"));
        r.commInitRank = reinterpret_cast<decltype(r.commInitRank)>(sym("ncclCommInitRank"));
        r.commDestroy = reinterpret_cast<decltype(r.commDestroy)>(sym("ncclCommDestroy"));
        r.gather = reinterpret_cast<decltype(r.gather)>(sym("ncclGather"));
        r.groupStart = reinterpret_cast<decltype(r.groupStart)>(sym("ncclGroupStart"));
        r.groupEnd = reinterpret_cast<decltype(r.groupEnd)>(sym("ncclGroupEnd"));
        r.send = reinterpret_cast<decltype(r.send)>(sym("ncclSend"));
        r.recv = reinterpret_cast<decltype(r.recv)>(sym("ncclRecv"));
        r.getErrorString = reinterpret_cast<decltype(r.getErrorString)>(sym("ncclGetErrorString"));
        r.commCount = reinterpret_cast<decltype(r.commCount)>(sym("ncclCommCount"));
        r.commUserRank = reinterpret_cast<decltype(r.commUserRank)>(sym("ncclCommUserRank"));
        r.commCuDevice = reinterpret_cast<decltype(r.commCuDevice)>(sym("ncclCommCuDevice"));
    });
    return r;
}

#define PPT_NCCL(call)                                                                                                 \
    do                                                                                                                 \
    {                                                                                                                  \
        const ncclResult_t r_ = (call);                                                                                \
        if (r_ != ncclSuccess)                                                                                         \
            return fail(PROSPER_PT_ERR_HIP, std::string(#call) + ": " + rccl().getErrorString(r_));                    \
    } while (0)

static_assert(sizeof(ncclUniqueId) == PROSPER_PT_COMM_ID_BYTES, "prosper_pt.h carries the id as 128 bytes");

} // namespace

namespace ppt
{

struct TilingState
{
    ncclComm_t comm = nullptr;
    bool ownsComm = false;
    uint32_t rank = 0, ranks = 1;
    hipStream_t commStream = nullptr; // the gather + de-interleave run here, beside the next frame's path stages
    hipEvent_t tileReady = nullptr;   // recorded on the caller's stream: the tile's accumulate kernel is done
    hipEvent_t gatherDone = nullptr;  // recorded on the comm stream behind gather (+ de-interleave on the root)
    bool gatherPending = false;
    hipEvent_t gatherT0 = nullptr, gatherT1 = nullptr; // timing events around the last gather (+ de-interleave), on its stream
    bool gatherTimed = false;
    uint32_t gathers = 0;
    float4 *staging = nullptr; // root: the ranks' tiles back to back, rank order
    size_t stagingBytes = 0;
    float4 *ownedFull = nullptr; // root: the gathered image when the caller passes no destination
    size_t ownedFullBytes = 0;
    float4 *lastFull = nullptr; // where the last gather put the image (root)
    uint32_t lastFullWidth = 0, lastFullHeight = 0;
};

uint32_t tile_local_width(uint32_t width, uint32_t stripeWidth, uint32_t rank, uint32_t ranks)
{
    if (ranks <= 1 || stripeWidth == 0) return width;
    uint32_t n = 0;
    for (uint32_t x = 0; x < width; x += stripeWidth)
        if ((x / stripeWidth) % ranks == rank) n += (x + stripeWidth <= width) ? stripeWidth : (width - x);
    return n;
}

void destroy_tiling(prosper_pt_ctx *ctx)
{
    TilingState *t = ctx->tiling;
    if (!t) return;
    (void)hipSetDevice(ctx->device); // one thread may drive several contexts: everything below belongs to this one's GPU
    if (t->commStream) (void)hipStreamSynchronize(t->commStream);
    if (t->comm && t->ownsComm && rccl().commDestroy) (void)rccl().commDestroy(t->comm);
    if (t->staging) (void)hipFree(t->staging);
    if (t->ownedFull) (void)hipFree(t->ownedFull);
    if (t->tileReady) (void)hipEventDestroy(t->tileReady);
    if (t->gatherDone) (void)hipEventDestroy(t->gatherDone);
    if (t->gatherT0) (void)hipEventDestroy(t->gatherT0);
    if (t->gatherT1) (void)hipEventDestroy(t->gatherT1);
    if (t->commStream) (void)hipStreamDestroy(t->commStream);
    delete t;
    ctx->tiling = nullptr;
}

void wait_for_gather_before_writing_tile(prosper_pt_ctx *ctx, hipStream_t stream)
{
    TilingState *t = ctx->tiling;
    if (t && t->gatherPending) (void)hipStreamWaitEvent(stream, t->gatherDone, 0);
}

} // namespace ppt

namespace
{

int ensure_state(prosper_pt_ctx *ctx)
{
    if (ctx->tiling) return PROSPER_PT_OK;
    PPT_HIP(hipSetDevice(ctx->device));
    TilingState *t = new (std::nothrow) TilingState();
    if (!t) return fail(PROSPER_PT_ERR_INVALID_ARGUMENT, "out of host memory");
    // published only once it is whole: a half-made state (no stream, null events) must never be seen by a later call
    // (the communication stream is a fifth stream beside the caller's and the three work streams: it shares a hardware
    // queue with one of them, so a gather can queue behind a path stage - it still overlaps the other frames' stages)
    hipError_t e = hipStreamCreateWithFlags(&t->commStream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&t->tileReady, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&t->gatherDone, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreate(&t->gatherT0);
    if (e == hipSuccess) e = hipEventCreate(&t->gatherT1);
    if (e != hipSuccess)
    {
        if (t->tileReady) (void)hipEventDestroy(t->tileReady);
        if (t->gatherDone) (void)hipEventDestroy(t->gatherDone);
        if (t->gatherT0) (void)hipEventDestroy(t->gatherT0);
        if (t->gatherT1) (void)hipEventDestroy(t->gatherT1);
        if (t->commStream) (void)hipStreamDestroy(t->commStream);
        delete t;
        return fail(PROSPER_PT_ERR_HIP, std::string("multi-GPU state: ") + hipGetErrorString(e));
    }
    ctx->tiling = t;
    return PROSPER_PT_OK;
}

int fill_layout(uint32_t width, uint32_t height, uint32_t stripeWidth, uint32_t ranks, TileLayout *out)
{
    if (ranks == 0 || ranks > kMaxRanks) return fail(PROSPER_PT_ERR_INVALID_ARGUMENT, "rank count must be 1..64");
    if (ranks > 1 && stripeWidth == 0) return fail(PROSPER_PT_ERR_INVALID_ARGUMENT, "stripe width is zero");
    TileLayout l = {};
    l.width = width;
    l.height = height;
    l.stripeWidth = ranks > 1 ? stripeWidth : (width ? width : 1u);
    l.ranks = ranks;
    uint64_t offset = 0;
    for (uint32_t r = 0; r < ranks; ++r)
    {
        l.localWidth[r] = tile_local_width(width, stripeWidth, r, ranks);
        l.tileOffset[r] = offset;
        offset += (uint64_t)l.localWidth[r] * height;
    }
    *out = l;
    return PROSPER_PT_OK;
}

} // namespace

extern "C" {

int prosper_pt_comm_get_unique_id(uint8_t id[PROSPER_PT_COMM_ID_BYTES])
{
    if (!id) return fail(PROSPER_PT_ERR_INVALID_ARGUMENT, "prosper_pt_comm_get_unique_id: null argument");
    if (!rccl().error.empty()) return fail(PROSPER_PT_ERR_UNSUPPORTED, rccl().error);
    ncclUniqueId uid;
    PPT_NCCL(rccl().getUniqueId(&uid));
    std::memcpy(id, &uid, sizeof(uid));
    return PROSPER_PT_OK;
}

int prosper_pt_comm_init(prosper_pt_ctx *ctx, const uint8_t id[PROSPER_PT_COMM_ID_BYTES], uint32_t rank, uint32_t ranks)
{
    if (!ctx || !id || ranks == 0 || ranks > kMaxRanks || rank >= ranks)
        return fail(PROSPER_PT_ERR_INVALID_ARGUMENT, "prosper_pt_comm_init: bad argument");
    if (!rccl().error.empty()) return fail(PROSPER_PT_ERR_UNSUPPORTED, rccl().error);
    int rc = ensure_state(ctx);
    if (rc != PROSPER_PT_OK) return rc;
    TilingState *t = ctx->tiling;
    if (t->comm) return fail(PROSPER_PT_ERR_INVALID_ARGUMENT, "prosper_pt_comm_init: the context already has a communicator");
    PPT_HIP(hipSetDevice(ctx->device));
    ncclUniqueId uid;
    std::memcpy(&uid, id, sizeof(uid));
    PPT_NCCL(rccl().commInitRank(&t->comm, (int)ranks, uid, (int)rank));
    t->ownsComm = true;
    t->rank = rank;
    t->ranks = ranks;
    return PROSPER_PT_OK;
}

int prosper_pt_comm_adopt(prosper_pt_ctx *ctx, void *nccl_comm, uint32_t rank, uint32_t ranks)
{
    if (!ctx || !nccl_comm || ranks == 0 || ranks > kMaxRanks || rank >= ranks)
        return fail(PROSPER_PT_ERR_INVALID_ARGUMENT, "prosper_pt_comm_adopt: bad argument");
    if (!rccl().error.empty()) return fail(PROSPER_PT_ERR_UNSUPPORTED, rccl().error);
    int rc = ensure_state(ctx);
    if (rc != PROSPER_PT_OK) return rc;
    TilingState *t = ctx->tiling;
    if (t->comm) return fail(PROSPER_PT_ERR_INVALID_ARGUMENT, "prosper_pt_comm_adopt: the context already has a communicator");
    t->comm = static_cast<ncclComm_t>(nccl_comm);
    t->ownsComm = false;
    t->rank = rank;
    t->ranks = ranks;
    return PROSPER_PT_OK;
}

int prosper_pt_comm_destroy(prosper_pt_ctx *ctx)
{
    if (!ctx) return fail(PROSPER_PT_ERR_INVALID_ARGUMENT, "prosper_pt_comm_destroy: null context");
    destroy_tiling(ctx);
    return PROSPER_PT_OK;
}

int prosper_pt_deinterleave_tiles(
    prosper_pt_ctx *ctx, const void *device_tiles, uint32_t ranks, uint32_t stripe_width, uint32_t width, uint32_t height,
    void *device_full_rgba32f, size_t byte_size, void *stream)
{
    if (!ctx || !device_tiles || !device_full_rgba32f)
        return fail(PROSPER_PT_ERR_INVALID_ARGUMENT, "prosper_pt_deinterleave_tiles: null argument");
    if (byte_size < (size_t)width * height * sizeof(float4))
        return fail(PROSPER_PT_ERR_INVALID_ARGUMENT, "prosper_pt_deinterleave_tiles: destination too small");
    TileLayout layout;
    const int rc = fill_layout(width, height, stripe_width, ranks, &layout);
    if (rc != PROSPER_PT_OK) return rc;
    PPT_HIP(hipSetDevice(ctx->device));
    launch_deinterleave_tiles(
        static_cast<const float4 *>(device_tiles), layout, static_cast<float4 *>(device_full_rgba32f),
        static_cast<hipStream_t>(stream));
    PPT_HIP(hipGetLastError());
    return PROSPER_PT_OK;
}

int prosper_pt_gather_tiles(
    prosper_pt_ctx *ctx, uint32_t root, void *device_full_rgba32f, size_t byte_size, uint32_t flags, void *stream)
{
    if (!ctx) return fail(PROSPER_PT_ERR_INVALID_ARGUMENT, "prosper_pt_gather_tiles: null context");
    if (!ctx->hdr || ctx->lastWidth == 0) return fail(PROSPER_PT_ERR_NO_SCENE, "prosper_pt_gather_tiles: nothing has been rendered yet");
    TilingState *t = ctx->tiling;
    const uint32_t ranks = ctx->stripeCount;
    if (ranks > 1 && (!t || !t->comm))
        return fail(PROSPER_PT_ERR_INVALID_ARGUMENT, "prosper_pt_gather_tiles: no communicator (prosper_pt_comm_init / _adopt)");
    if (t && t->comm && (t->ranks != ranks || t->rank != ctx->stripeIndex))
        return fail(PROSPER_PT_ERR_INVALID_ARGUMENT, "prosper_pt_gather_tiles: the last render's tile does not match the communicator's rank / size");
    if (root >= ranks) return fail(PROSPER_PT_ERR_INVALID_ARGUMENT, "prosper_pt_gather_tiles: root out of range");
    const bool isRoot = ranks == 1 || t->rank == root;
    const uint32_t width = ctx->lastWidth, height = ctx->height;
    if (isRoot && device_full_rgba32f && byte_size < (size_t)width * height * sizeof(float4))
        return fail(PROSPER_PT_ERR_INVALID_ARGUMENT, "prosper_pt_gather_tiles: destination too small");
    int rc = ensure_state(ctx);
    if (rc != PROSPER_PT_OK) return rc;
    t = ctx->tiling;
    PPT_HIP(hipSetDevice(ctx->device));
    if (isRoot && !device_full_rgba32f)
    {
        // no destination: the context keeps the gathered image (prosper_pt_get_gathered_device_ptr / _read_gathered)
        const size_t need = (size_t)width * height * sizeof(float4);
        if (t->ownedFullBytes < need)
        {
            PPT_HIP(hipDeviceSynchronize());
            if (t->ownedFull) PPT_HIP(hipFree(t->ownedFull));
            t->ownedFull = nullptr;
            t->ownedFullBytes = 0;
            PPT_HIP(hipMalloc((void **)&t->ownedFull, need ? need : 16));
            t->ownedFullBytes = need;
        }
        device_full_rgba32f = t->ownedFull;
    }
    if (isRoot)
    {
        t->lastFull = static_cast<float4 *>(device_full_rgba32f);
        t->lastFullWidth = width;
        t->lastFullHeight = height;
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    const bool inStream = (flags & PROSPER_PT_GATHER_IN_STREAM) != 0;
    hipStream_t cs = inStream ? s : t->commStream;
    if (!inStream)
    {
        // the tile is complete once everything enqueued on the caller's stream so far has run
        PPT_HIP(hipEventRecord(t->tileReady, s));
        PPT_HIP(hipStreamWaitEvent(cs, t->tileReady, 0));
    }
    else if (t->gatherPending)
        PPT_HIP(hipStreamWaitEvent(s, t->gatherDone, 0)); // staging is still in use by the previous gather

    TileLayout layout;
    rc = fill_layout(width, height, ctx->stripeWidth, ranks, &layout);
    if (rc != PROSPER_PT_OK) return rc;
    const size_t myCount = (size_t)ctx->localWidth * height * 4u; // floats
    PPT_HIP(hipEventRecord(t->gatherT0, cs));
    if (ranks == 1 && !t->comm)
    {
        if (device_full_rgba32f != ctx->hdr)
            PPT_HIP(hipMemcpyAsync(device_full_rgba32f, ctx->hdr, myCount * 4u, hipMemcpyDeviceToDevice, cs));
    }
    else
    {
        if (isRoot)
        {
            const size_t need = (size_t)width * height * sizeof(float4);
            if (t->stagingBytes < need)
            {
                PPT_HIP(hipStreamSynchronize(t->commStream));
                if (t->staging) PPT_HIP(hipFree(t->staging));
                t->staging = nullptr;
                t->stagingBytes = 0;
                PPT_HIP(hipMalloc((void **)&t->staging, need));
                t->stagingBytes = need;
            }
        }
        bool equal = true;
        for (uint32_t r = 1; r < ranks; ++r) equal = equal && layout.localWidth[r] == layout.localWidth[0];
        if (equal)
            // the one data-path collective: every rank's RGBA32F tile to the root (rccl.h ncclGather)
            PPT_NCCL(rccl().gather(ctx->hdr, isRoot ? t->staging : nullptr, myCount, ncclFloat, (int)root, t->comm, cs));
        else
        {
            // stripe counts that do not divide over the ranks: grouped send / recv with per-rank counts
            PPT_NCCL(rccl().groupStart());
            ncclResult_t r1 = rccl().send(ctx->hdr, myCount, ncclFloat, (int)root, t->comm, cs);
            if (isRoot)
                for (uint32_t r = 0; r < ranks && r1 == ncclSuccess; ++r)
                    r1 = rccl().recv(
                        t->staging + layout.tileOffset[r], (size_t)layout.localWidth[r] * height * 4u, ncclFloat, (int)r, t->comm, cs);
            const ncclResult_t r2 = rccl().groupEnd();
            PPT_NCCL(r1);
            PPT_NCCL(r2);
        }
        if (isRoot)
        {
            launch_deinterleave_tiles(t->staging, layout, static_cast<float4 *>(device_full_rgba32f), cs);
            PPT_HIP(hipGetLastError());
        }
    }
    PPT_HIP(hipEventRecord(t->gatherT1, cs));
    t->gatherTimed = true;
    t->gathers++;
    if (!inStream)
    {
        PPT_HIP(hipEventRecord(t->gatherDone, cs));
        t->gatherPending = true;
    }
    return PROSPER_PT_OK;
}

int prosper_pt_comm_query(prosper_pt_ctx *ctx, prosper_pt_comm_info *out)
{
    if (!ctx || !out) return fail(PROSPER_PT_ERR_INVALID_ARGUMENT, "prosper_pt_comm_query: null argument");
    *out = prosper_pt_comm_info{1u, 0u, ctx->device, 0u, 0.0f, 0u};
    TilingState *t = ctx->tiling;
    if (!t) return PROSPER_PT_OK;
    PPT_HIP(hipSetDevice(ctx->device));
    if (t->comm)
    {
        int count = 0, rank = 0, device = 0;
        PPT_NCCL(rccl().commCount(t->comm, &count));
        PPT_NCCL(rccl().commUserRank(t->comm, &rank));
        PPT_NCCL(rccl().commCuDevice(t->comm, &device));
        out->ranks = (uint32_t)count;
        out->rank = (uint32_t)rank;
        out->device = device;
    }
    out->gathers = t->gathers;
    if (t->gatherTimed)
    {
        PPT_HIP(hipEventSynchronize(t->gatherT1));
        float ms = 0.0f;
        PPT_HIP(hipEventElapsedTime(&ms, t->gatherT0, t->gatherT1));
        out->lastGatherMs = ms;
    }
    return PROSPER_PT_OK;
}

int prosper_pt_get_gathered_device_ptr(prosper_pt_ctx *ctx, void **out_ptr, uint32_t *width, uint32_t *height)
{
    if (!ctx || !out_ptr) return fail(PROSPER_PT_ERR_INVALID_ARGUMENT, "prosper_pt_get_gathered_device_ptr: null argument");
    TilingState *t = ctx->tiling;
    if (!t || !t->lastFull) return fail(PROSPER_PT_ERR_NO_SCENE, "this context has not been the root of a gather yet");
    *out_ptr = t->lastFull;
    if (width) *width = t->lastFullWidth;
    if (height) *height = t->lastFullHeight;
    return PROSPER_PT_OK;
}

int prosper_pt_read_gathered(prosper_pt_ctx *ctx, float *rgba32f, size_t byte_size, void *stream)
{
    if (!ctx || !rgba32f) return fail(PROSPER_PT_ERR_INVALID_ARGUMENT, "prosper_pt_read_gathered: null argument");
    TilingState *t = ctx->tiling;
    if (!t || !t->lastFull) return fail(PROSPER_PT_ERR_NO_SCENE, "this context has not been the root of a gather yet");
    const size_t bytes = (size_t)t->lastFullWidth * t->lastFullHeight * sizeof(float4);
    if (byte_size < bytes) return fail(PROSPER_PT_ERR_INVALID_ARGUMENT, "prosper_pt_read_gathered: destination too small");
    PPT_HIP(hipSetDevice(ctx->device));
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (t->gatherPending) PPT_HIP(hipStreamWaitEvent(s, t->gatherDone, 0));
    PPT_HIP(hipMemcpyAsync(rgba32f, t->lastFull, bytes, hipMemcpyDeviceToHost, s));
    PPT_HIP(hipStreamSynchronize(s));
    return PROSPER_PT_OK;
}

int prosper_pt_gather_wait(prosper_pt_ctx *ctx, void *stream)
{
    if (!ctx) return fail(PROSPER_PT_ERR_INVALID_ARGUMENT, "prosper_pt_gather_wait: null context");
    TilingState *t = ctx->tiling;
    if (t && t->gatherPending) PPT_HIP(hipStreamWaitEvent(static_cast<hipStream_t>(stream), t->gatherDone, 0));
    return PROSPER_PT_OK;
}

} // extern "C"
