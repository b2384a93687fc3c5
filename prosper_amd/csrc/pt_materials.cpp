// pt_materials.cpp — see pt_materials.hpp.  C-ABI: prosper_pt_update_textures, prosper_pt_update_materials
// (include/prosper_pt/prosper_pt.h, "incremental adoption").
//
// prosper streams a scene in: the meshes first, then the images a few per frame, and a material switches from its
// placeholder (the default material with the real alpha mode, WorldData.cpp:817-826) to the real one once its three images
// are there (WorldData.cpp:2208-2239); the whole material buffer of the next frame is rewritten when that happened
// (WorldData.cpp:568-586), descriptors of new images are written as they arrive (:2182-2206).  Nothing of that rebuilds an
// acceleration structure.  Here the same two events cost what they touch: a texture update copies and re-tiles the new
// texels (on a stream of its own, beside the frames in flight), a material update rebuilds that material's texture pack and
// alpha bounds, and the next render switches to a new version of the four small tables at the head of its own chain.
#include "pt_materials.hpp"

#include <cmath>
#include <cstring>
#include <string>

#include "pt_kernels.hpp"

namespace ppt
{

size_t texture_staging_bytes(const prosper_pt_texture_desc &t)
{
    const size_t bytes = t.format == PROSPER_PT_FORMAT_BC7_UNORM ? (size_t)(t.width / 4u) * (t.height / 4u) * 16u : (size_t)t.width * t.height * 4u;
    return (bytes + 255u) & ~(size_t)255u;
}

int validate_texture(const prosper_pt_texture_desc &t, uint32_t index)
{
    if (!t.texels || t.width == 0 || t.height == 0 || t.format > PROSPER_PT_FORMAT_BC7_UNORM)
        return fail(PROSPER_PT_ERR_SCENE, "texture " + std::to_string(index) + " is invalid");
    if (t.format == PROSPER_PT_FORMAT_BC7_UNORM && (t.width % 4u != 0u || t.height % 4u != 0u))
        return fail(PROSPER_PT_ERR_SCENE, "BC7 texture " + std::to_string(index) + " is not a whole number of 4x4 blocks");
    return PROSPER_PT_OK;
}

int create_device_texture(prosper_pt_ctx *ctx, const prosper_pt_texture_desc &t, void *staging, hipStream_t stream, DeviceTexture *out, void *pinned)
{
    const uint32_t tilesX = (t.width + kTexTileW - 1u) / kTexTileW, tilesY = (t.height + kTexTileH - 1u) / kTexTileH;
    const size_t tiledBytes = (size_t)tilesX * tilesY * (kTexTileW * kTexTileH) * 4u;
    void *d = nullptr;
    const int rc = device_alloc(ctx, tiledBytes, &d);
    if (rc != PROSPER_PT_OK) return rc;
    if (t.format == PROSPER_PT_FORMAT_BC7_UNORM)
    {
        // the blocks go up as they are and a kernel decodes them into the tiles (pt_bc7.hpp)
        const size_t blockBytes = (size_t)(t.width / 4u) * (t.height / 4u) * 16u;
        const void *src = t.texels;
        if (pinned)
        {
            std::memcpy(pinned, t.texels, blockBytes);
            src = pinned;
        }
        PPT_HIP(hipMemcpyAsync(staging, src, blockBytes, hipMemcpyHostToDevice, stream));
        PPT_HIP(hipMemsetAsync(d, 0, tiledBytes, stream));
        launch_decode_bc7(staging, t.width, t.height, tilesX, d, stream);
    }
    else
    {
        // 8 x 4-texel tiles of one cache line each (pt_scene.hpp DeviceTexture), laid out by a kernel
        const void *src = t.texels;
        if (pinned)
        {
            std::memcpy(pinned, t.texels, (size_t)t.width * t.height * 4u);
            src = pinned;
        }
        PPT_HIP(hipMemcpyAsync(staging, src, (size_t)t.width * t.height * 4u, hipMemcpyHostToDevice, stream));
        launch_retile_rgba8(staging, t.width, t.height, tilesX, d, stream);
    }
    PPT_HIP(hipGetLastError());
    *out = DeviceTexture{static_cast<const uint8_t *>(d), t.width, t.height, tilesX, 0u};
    return PROSPER_PT_OK;
}

int build_material_pack(
    prosper_pt_ctx *ctx, const prosper_MaterialData &m, const std::vector<DeviceTexture> &textures, bool noPacks, bool wide,
    hipStream_t stream, MaterialPack *out)
{
    *out = MaterialPack{nullptr, 0u, 0u, 0u, 0u};
    const uint32_t tb = m.baseColorTextureSampler & 0xFFFFFFu, tm = m.metallicRoughnessTextureSampler & 0xFFFFFFu,
                   tn = m.normalTextureSampler & 0xFFFFFFu;
    const uint32_t sb = m.baseColorTextureSampler >> 24, sm = m.metallicRoughnessTextureSampler >> 24, sn = m.normalTextureSampler >> 24;
    if (noPacks || tb == 0 || tm == 0 || tn == 0 || sb != sm || sb != sn) return PROSPER_PT_OK;
    const DeviceTexture &b = textures[tb], &r = textures[tm], &n = textures[tn];
    if (!b.texels || !r.texels || !n.texels) return PROSPER_PT_OK;
    if (b.width != r.width || b.width != n.width || b.height != r.height || b.height != n.height) return PROSPER_PT_OK;
    if (b.width < 8u || b.height < 8u) return PROSPER_PT_OK; // tiny placeholder textures: nothing to gain
    MaterialPack pk;
    pk.width = b.width;
    pk.height = b.height;
    pk.tilesPerRow = (b.width + kPackTileW - 1u) / kPackTileW;
    // an OPAQUE material never reads base.a: the 8-byte texel
    const bool compact = m.alphaMode == PROSPER_ALPHA_MODE_OPAQUE && !wide;
    pk.sampler = sb | (compact ? kPackCompactBit : 0u);
    const size_t texelCount = (size_t)pk.tilesPerRow * kPackTileW * (((size_t)b.height + kPackTileH - 1u) / kPackTileH) * kPackTileH;
    void *d = nullptr;
    const int rc = device_alloc(ctx, texelCount * (compact ? sizeof(uint2) : sizeof(uint4)), &d);
    if (rc != PROSPER_PT_OK) return rc;
    pk.texels = d;
    launch_pack_material_textures(b, r, n, pk, stream);
    PPT_HIP(hipGetLastError());
    *out = pk;
    return PROSPER_PT_OK;
}

int build_alpha_material(
    prosper_pt_ctx *ctx, const prosper_MaterialData &m, const std::vector<DeviceTexture> &textures,
    const std::vector<prosper_pt_sampler_desc> &samplers, hipStream_t stream, AlphaMaterial *out, uint64_t *boundBytes)
{
    AlphaMaterial am{};
    am.factorA = m.baseColorFactor.w;
    am.cutoff = m.alphaCutoff;
    am.bits = m.alphaMode & 3u;
    *boundBytes = 0;
    const uint32_t tex = m.baseColorTextureSampler & 0xFFFFFFu;
    if (m.alphaMode != PROSPER_ALPHA_MODE_OPAQUE && tex != 0)
    {
        const DeviceTexture &t = textures[tex];
        const prosper_pt_sampler_desc &sd = samplers[m.baseColorTextureSampler >> 24];
        if (!t.texels)
            return fail(PROSPER_PT_ERR_UNSUPPORTED, "the base-colour texture of a MASK / BLEND material has no texels of its own on the device");
        if (t.width > 0xFFFFu || t.height > 0xFFFFu)
            return fail(PROSPER_PT_ERR_UNSUPPORTED, "base-colour texture of a MASK / BLEND material exceeds 65535 texels a side");
        am.texels = t.texels;
        am.width = (uint16_t)t.width;
        am.height = (uint16_t)t.height;
        am.bits |= (sd.wrapS & 3u) << 2 | (sd.wrapT & 3u) << 4 | (sd.magFilter == PROSPER_PT_FILTER_NEAREST ? 64u : 0u);
        // cells of 2 x 2 texels: the table is an eighth of the texture (8 KB for 128^2: it lives in the L1 / L2
        // the texels would have been read through); coarser for textures whose table would pass 2 MB
        uint32_t shift = 1;
        while (((uint64_t)(t.width >> shift) + 1u) * ((t.height >> shift) + 1u) * 2u > (2ull << 20)) ++shift;
        if (ctx->debug.alphaCellShift >= 0) shift = (uint32_t)std::min(15, ctx->debug.alphaCellShift);
        const bool factorOk = std::isfinite(m.baseColorFactor.w) && m.baseColorFactor.w >= 0.0f;
        if (!ctx->debug.noAlphaBounds && factorOk)
        {
            const uint32_t cellsX = (t.width + (1u << shift) - 1u) >> shift, cellsY = (t.height + (1u << shift) - 1u) >> shift;
            void *d = nullptr;
            const int rc = device_alloc(ctx, (size_t)cellsX * cellsY * 2u, &d);
            if (rc != PROSPER_PT_OK) return rc;
            launch_build_alpha_bounds(t, sd.wrapS, sd.wrapT, m.baseColorFactor.w, shift, static_cast<uint16_t *>(d), stream);
            PPT_HIP(hipGetLastError());
            am.bounds = static_cast<const uint16_t *>(d);
            am.bits |= shift << 8;
            *boundBytes = (uint64_t)cellsX * cellsY * 2u;
        }
    }
    *out = am;
    return PROSPER_PT_OK;
}

namespace
{

int ensure_update_state(prosper_pt_ctx *ctx)
{
    MaterialState *ms = ctx->materialState;
    if (!ms->uploadStream) PPT_HIP(hipStreamCreateWithFlags(&ms->uploadStream, hipStreamNonBlocking));
    if (!ms->uploaded) PPT_HIP(hipEventCreateWithFlags(&ms->uploaded, hipEventDisableTiming));
    if (!ms->ready) PPT_HIP(hipEventCreateWithFlags(&ms->ready, hipEventDisableTiming));
    return PROSPER_PT_OK;
}

bool wide_packs(const prosper_pt_ctx *ctx)
{
    // compact packs where the texels outgrow the caches; debug option widePacks = 1 / 0 forces either kind
    if (ctx->debug.widePacks >= 0) return ctx->debug.widePacks != 0;
    return !texel_set_is_big(ctx->materialState->texelBytes);
}

} // namespace

size_t allocation_bytes(prosper_pt_ctx *ctx, const void *p)
{
    const std::lock_guard<std::mutex> lock(ctx->allocMutex);
    for (const DeviceAllocation &a : ctx->sceneAllocations)
        if (a.ptr == p) return a.bytes;
    return 0;
}

// `p` was replaced by an update: nothing new will read it, a frame in flight still may
void retire(prosper_pt_ctx *ctx, const void *p)
{
    if (!p) return;
    MaterialState *ms = ctx->materialState;
    ms->retired.push_back(p);
    ms->retiredBytes += allocation_bytes(ctx, p);
}

// Frees what has been retired once it is worth a device synchronisation (a material slider dragged for a thousand frames
// must not grow the scene without bound; a streamed scene never gets here: it retires a few placeholder texels).
int collect_retired(prosper_pt_ctx *ctx)
{
    MaterialState *ms = ctx->materialState;
    if (ms->retiredBytes < MaterialState::kRetireBytes) return PROSPER_PT_OK;
    PPT_HIP(hipDeviceSynchronize());
    for (const void *p : ms->retired) device_free(ctx, p);
    ms->retired.clear();
    ms->retiredBytes = 0;
    // (the device is idle: the generations a geometry build replaced can go with their events and pinned staging)
    for (AccelState *old : ctx->retiredAccel) delete old;
    ctx->retiredAccel.clear();
    return PROSPER_PT_OK;
}

namespace
{

bool same_pack_inputs(const prosper_MaterialData &a, const prosper_MaterialData &b)
{
    return a.baseColorTextureSampler == b.baseColorTextureSampler && a.metallicRoughnessTextureSampler == b.metallicRoughnessTextureSampler &&
           a.normalTextureSampler == b.normalTextureSampler && a.alphaMode == b.alphaMode;
}

// The pack and the alpha material of material `i` again, from the mirrors, on the upload stream.  `previous`: the
// material's entry before this update (nullptr: its textures' TEXELS changed) - what does not depend on what changed is
// kept: the pack on a material's textures and samplers, the alpha bounds on the base-colour texture, its sampler and
// baseColorFactor.a.
int rebuild_material(prosper_pt_ctx *ctx, uint32_t i, const prosper_MaterialData *previous)
{
    MaterialState *ms = ctx->materialState;
    const prosper_MaterialData &m = ms->materials[i];
    int rc;
    const bool wide = wide_packs(ctx);
    const bool packKindKept = ms->packs[i].texels == nullptr || (((ms->packs[i].sampler & kPackCompactBit) == 0u) == (wide || m.alphaMode != PROSPER_ALPHA_MODE_OPAQUE));
    if (!(previous && same_pack_inputs(*previous, m) && packKindKept))
    {
        MaterialPack pk;
        if ((rc = build_material_pack(ctx, m, ms->textures, ctx->debug.noTexturePacks != 0, wide, ms->uploadStream, &pk))) return rc;
        if ((pk.texels != nullptr) != (ms->packs[i].texels != nullptr)) ms->packedMaterials += pk.texels ? 1u : ~0u;
        retire(ctx, ms->packs[i].texels);
        ms->packs[i] = pk;
    }
    const AlphaMaterial old = ms->alphaMaterials[i];
    const bool boundsKept = previous && previous->baseColorTextureSampler == m.baseColorTextureSampler && previous->alphaMode == m.alphaMode &&
                            std::memcmp(&previous->baseColorFactor.w, &m.baseColorFactor.w, sizeof(float)) == 0;
    if (boundsKept)
    {
        AlphaMaterial am = old; // same texels, same bounds; the cutoff may have moved
        am.cutoff = m.alphaCutoff;
        ms->alphaMaterials[i] = am;
    }
    else
    {
        AlphaMaterial am;
        uint64_t bytes = 0;
        if ((rc = build_alpha_material(ctx, m, ms->textures, ms->samplers, ms->uploadStream, &am, &bytes))) return rc;
        ms->alphaBoundBytes += bytes;
        ms->alphaBoundBytes -= std::min<uint64_t>(ms->alphaBoundBytes, allocation_bytes(ctx, old.bounds));
        retire(ctx, old.bounds);
        ms->alphaMaterials[i] = am;
    }
    if (m.alphaMode != PROSPER_ALPHA_MODE_OPAQUE && std::memcmp(&old, &ms->alphaMaterials[i], sizeof(AlphaMaterial)) != 0)
        ms->pendingAlphaPatch = true;
    return PROSPER_PT_OK;
}

int check_material(const MaterialState *ms, const prosper_MaterialData &m, uint32_t index)
{
    const uint32_t ts[3] = {m.baseColorTextureSampler, m.metallicRoughnessTextureSampler, m.normalTextureSampler};
    for (uint32_t k = 0; k < 3; ++k)
    {
        const uint32_t tex = ts[k] & 0xFFFFFFu, smp = ts[k] >> 24;
        if (tex > 0 && (tex >= ms->textures.size() || smp >= ms->samplers.size()))
            return fail(PROSPER_PT_ERR_SCENE, "material " + std::to_string(index) + " references a missing texture/sampler");
        // (a pack is rebuilt from its three textures' own texels - only when what it depends on changed: an entry that
        //  keeps its textures and samplers keeps its pack, and prosper rewrites the WHOLE table every time)
        if (tex > 0 && !ms->textures[tex].texels && !same_pack_inputs(ms->materials[index], m))
            return fail(
                PROSPER_PT_ERR_UNSUPPORTED, "material " + std::to_string(index) + ": texture " + std::to_string(tex) +
                                                " kept no texels of its own at upload (only packed materials sampled it): update that texture first");
    }
    if (m.alphaMode > PROSPER_ALPHA_MODE_BLEND) return fail(PROSPER_PT_ERR_SCENE, "material alpha mode invalid");
    // the alpha mode decides the opaque flag of the geometry (World.cpp:646-651) and with it the any-hit records; prosper
    // gives a streaming material's placeholder the real mode for the same reason (WorldData.cpp:817-826)
    if (m.alphaMode != ms->materials[index].alphaMode)
        return fail(PROSPER_PT_ERR_UNSUPPORTED, "material " + std::to_string(index) + ": the alpha mode of an uploaded material cannot change (upload the scene again)");
    return PROSPER_PT_OK;
}

} // namespace

int flush_pending_materials(prosper_pt_ctx *ctx, hipStream_t stream)
{
    MaterialState *ms = ctx->materialState;
    if (!ms || !ms->pending) return PROSPER_PT_OK;
    const uint32_t v = ms->cur == 0u ? 1u : (ms->cur % 3u) + 1u;
    int rc;
    if (!ms->dBlocks[v])
    {
        void *d = nullptr;
        if ((rc = device_alloc(ctx, ms->blockBytes, &d))) return rc;
        ms->dBlocks[v] = static_cast<uint8_t *>(d);
    }
    if ((rc = ensure_update_state(ctx))) return rc;
    const uint32_t k = ms->stagingNext;
    if (!ms->staging[k])
    {
        PPT_HIP(hipHostMalloc((void **)&ms->staging[k], ms->blockBytes, hipHostMallocDefault));
        PPT_HIP(hipEventCreateWithFlags(&ms->stagingDone[k], hipEventDisableTiming));
    }
    if (ms->stagingUsed[k]) PPT_HIP(hipEventSynchronize(ms->stagingDone[k])); // the copy of four flushes ago
    ms->stagingUsed[k] = false;
    uint8_t *img = ms->staging[k];
    std::memcpy(img, ms->materials.data(), ms->materials.size() * sizeof(prosper_MaterialData));
    std::memcpy(img + ms->packsOffset, ms->packs.data(), ms->packs.size() * sizeof(MaterialPack));
    std::memcpy(img + ms->alphaOffset, ms->alphaMaterials.data(), ms->alphaMaterials.size() * sizeof(AlphaMaterial));
    std::memcpy(img + ms->texturesOffset, ms->textures.data(), ms->textures.size() * sizeof(DeviceTexture));
    // the block's last readers (three flushes ago), the texel arrays / packs / bounds the tables point at
    if (ms->versionUsed[v]) PPT_HIP(hipStreamWaitEvent(stream, ms->versionFree[v], 0));
    if (ms->uploadedRecorded) PPT_HIP(hipStreamWaitEvent(stream, ms->uploaded, 0));
    // ... and the previous flush, which ran on another render's stream: its rewrite of the any-hit records (they exist
    // once, for every version) must be done before this render reads them.  `ready` is recorded anew below, so this wait is
    // the only thing that orders the render behind the earlier record: without it a frame whose own flush patches nothing
    // could overtake the patch of the flush before (found when a fifth stream changed which streams share a hardware
    // queue: 927 pixels of FlightHelmet's lenses, one frame in six).
    if (ms->readyRecorded) PPT_HIP(hipStreamWaitEvent(stream, ms->ready, 0));
    PPT_HIP(hipMemcpyAsync(ms->dBlocks[v], img, ms->blockBytes, hipMemcpyHostToDevice, stream));
    PPT_HIP(hipEventRecord(ms->stagingDone[k], stream));
    ms->stagingUsed[k] = true;
    ms->stagingNext = (k + 1u) % kStagingBuffers;
    const AlphaMaterial *dAlpha = reinterpret_cast<const AlphaMaterial *>(ms->dBlocks[v] + ms->alphaOffset);
    if (ms->pendingAlphaPatch && ctx->alphaTriangleCount)
    {
        // the any-hit records carry a copy of their material's AlphaMaterial and exist once: they are rewritten in place,
        // behind every render that may still read them (the one kind of update that waits for the frames in flight)
        for (uint32_t i = 0; i < MaterialState::kVersions; ++i)
            if (ms->versionUsed[i]) PPT_HIP(hipStreamWaitEvent(stream, ms->versionFree[i], 0));
        launch_patch_alpha_records(
            const_cast<AlphaTriangle *>(ctx->scene.alphaTriangles), (uint32_t)ctx->alphaTriangleCount, dAlpha,
            (uint32_t)ms->alphaMaterials.size(), stream);
        PPT_HIP(hipGetLastError());
    }
    PPT_HIP(hipEventRecord(ms->ready, stream));
    // ---- commit ----
    ms->readyRecorded = true;
    ms->cur = v;
    ctx->scene.materials = reinterpret_cast<const prosper_MaterialData *>(ms->dBlocks[v]);
    ctx->scene.materialPacks = reinterpret_cast<const MaterialPack *>(ms->dBlocks[v] + ms->packsOffset);
    ctx->scene.alphaMaterials = dAlpha;
    ctx->scene.textures = reinterpret_cast<const DeviceTexture *>(ms->dBlocks[v] + ms->texturesOffset);
    ctx->scene.batchedTextures = ctx->debug.batchedTextures >= 0 ? (ctx->debug.batchedTextures ? 1u : 0u) : (texel_set_is_big(ms->texelBytes) ? 1u : 0u);
    ctx->packedMaterials = ms->packedMaterials;
    ctx->alphaBoundBytes = ms->alphaBoundBytes;
    ctx->stats.alphaBoundBytes = ms->alphaBoundBytes;
    ctx->stats.deviceBytes = ctx->sceneBytes;
    ms->pending = false;
    ms->pendingAlphaPatch = false;
    ms->updates++;
    return PROSPER_PT_OK;
}

} // namespace ppt

using namespace ppt;

extern "C" {

int prosper_pt_update_textures(prosper_pt_ctx *ctx, const prosper_pt_texture_desc *textures, uint32_t first, uint32_t count)
{
    if (!ctx || (!textures && count)) return fail(PROSPER_PT_ERR_INVALID_ARGUMENT, "prosper_pt_update_textures: null argument");
    if (!ctx->haveScene || !ctx->materialState) return fail(PROSPER_PT_ERR_NO_SCENE, "no scene uploaded");
    MaterialState *ms = ctx->materialState;
    if ((uint64_t)first + count > ms->textures.size())
        return fail(PROSPER_PT_ERR_INVALID_ARGUMENT, "prosper_pt_update_textures: range exceeds the scene's textureCount");
    if (count == 0) return PROSPER_PT_OK;
    size_t stagingBytes = 0;
    for (uint32_t i = 0; i < count; ++i)
    {
        const int rc = validate_texture(textures[i], first + i);
        if (rc != PROSPER_PT_OK) return rc;
        stagingBytes += texture_staging_bytes(textures[i]);
    }
    // A material that samples a replaced texture gets its pack rebuilt from its three textures' own texels: a texture whose
    // own copy was released at upload (only packed materials sampled it) must arrive in the same call.
    for (uint32_t m = 0; m < ms->materials.size(); ++m)
    {
        const prosper_MaterialData &md = ms->materials[m];
        const uint32_t t3[3] = {md.baseColorTextureSampler & 0xFFFFFFu, md.metallicRoughnessTextureSampler & 0xFFFFFFu,
                                md.normalTextureSampler & 0xFFFFFFu};
        auto replaced = [&](uint32_t t) { return t >= first && t < first + count; };
        if (!((t3[0] && replaced(t3[0])) || (t3[1] && replaced(t3[1])) || (t3[2] && replaced(t3[2])))) continue;
        for (uint32_t t : t3)
            if (t && !replaced(t) && !ms->textures[t].texels)
                return fail(
                    PROSPER_PT_ERR_UNSUPPORTED, "prosper_pt_update_textures: material " + std::to_string(m) + " also samples texture " + std::to_string(t) +
                                                    ", which kept no texels of its own at upload (only packed materials sampled it): update it in the same call");
    }
    PPT_HIP(hipSetDevice(ctx->device));
    int rc = ensure_update_state(ctx);
    if (rc == PROSPER_PT_OK) rc = collect_retired(ctx);
    if (rc != PROSPER_PT_OK) return rc;
    // the previous update's copies out of the pinned staging area (a frame ago, as a rule: done long since)
    if (ms->uploadedRecorded) PPT_HIP(hipEventSynchronize(ms->uploaded));
    if (stagingBytes > ms->linearStagingBytes)
    {
        // (the outgrown area joins the scene's allocations: kernels of an earlier update may still read it)
        if (ms->linearStaging) ctx->sceneAllocations.push_back({ms->linearStaging, ms->linearStagingBytes});
        ms->linearStaging = nullptr;
        ms->linearStagingBytes = 0;
        PPT_HIP(hipMalloc(&ms->linearStaging, stagingBytes));
        ms->linearStagingBytes = stagingBytes;
    }
    if (stagingBytes > ms->pinnedStagingBytes)
    {
        // (the stream was synchronised at the end of the previous call: nothing reads the old area any more)
        if (ms->pinnedStaging) PPT_HIP(hipHostFree(ms->pinnedStaging));
        ms->pinnedStaging = nullptr;
        ms->pinnedStagingBytes = 0;
        PPT_HIP(hipHostMalloc(&ms->pinnedStaging, stagingBytes, hipHostMallocDefault));
        ms->pinnedStagingBytes = stagingBytes;
    }
    size_t offset = 0;
    std::vector<uint8_t> changed(ms->textures.size(), 0);
    for (uint32_t i = 0; i < count; ++i)
    {
        DeviceTexture dt;
        if ((rc = create_device_texture(
                 ctx, textures[i], static_cast<uint8_t *>(ms->linearStaging) + offset, ms->uploadStream, &dt,
                 static_cast<uint8_t *>(ms->pinnedStaging) + offset)))
        {
            (void)hipStreamSynchronize(ms->uploadStream);
            return rc;
        }
        offset += texture_staging_bytes(textures[i]);
        const DeviceTexture &old = ms->textures[first + i];
        ms->texelBytes += (uint64_t)dt.width * dt.height * 4u;
        ms->texelBytes -= std::min<uint64_t>(ms->texelBytes, (uint64_t)old.width * old.height * 4u);
        retire(ctx, old.texels); // (a frame in flight may still read the previous texel array)
        ms->textures[first + i] = dt;
        changed[first + i] = 1;
    }
    // (the caller's memory has been read - into the pinned staging area - when create_device_texture returns: nothing to wait
    //  for here.  The copies out of that area are waited for by the NEXT call, before it overwrites it.)
    // materials that sample a replaced texture: their packs and alpha bounds hold its texels
    for (uint32_t m = 0; m < ms->materials.size(); ++m)
    {
        const prosper_MaterialData &md = ms->materials[m];
        const uint32_t t3[3] = {md.baseColorTextureSampler & 0xFFFFFFu, md.metallicRoughnessTextureSampler & 0xFFFFFFu,
                                md.normalTextureSampler & 0xFFFFFFu};
        if ((t3[0] && changed[t3[0]]) || (t3[1] && changed[t3[1]]) || (t3[2] && changed[t3[2]]))
            if ((rc = rebuild_material(ctx, m, nullptr))) return rc;
    }
    PPT_HIP(hipEventRecord(ms->uploaded, ms->uploadStream));
    ms->uploadedRecorded = true;
    ms->pending = true;
    ms->changes++;
    return PROSPER_PT_OK;
}

int prosper_pt_update_materials(prosper_pt_ctx *ctx, const prosper_MaterialData *materials, uint32_t first, uint32_t count)
{
    if (!ctx || (!materials && count)) return fail(PROSPER_PT_ERR_INVALID_ARGUMENT, "prosper_pt_update_materials: null argument");
    if (!ctx->haveScene || !ctx->materialState) return fail(PROSPER_PT_ERR_NO_SCENE, "no scene uploaded");
    MaterialState *ms = ctx->materialState;
    if ((uint64_t)first + count > ms->materials.size())
        return fail(PROSPER_PT_ERR_INVALID_ARGUMENT, "prosper_pt_update_materials: range exceeds the scene's materialCount");
    for (uint32_t i = 0; i < count; ++i)
    {
        const int rc = check_material(ms, materials[i], first + i);
        if (rc != PROSPER_PT_OK) return rc;
    }
    bool any = false;
    for (uint32_t i = 0; i < count; ++i)
    {
        // prosper rewrites the whole buffer when ONE material changed (WorldData.cpp:568-586): the others cost a memcmp
        if (std::memcmp(&ms->materials[first + i], &materials[i], sizeof(prosper_MaterialData)) == 0) continue;
        if (!any)
        {
            PPT_HIP(hipSetDevice(ctx->device));
            int rc = ensure_update_state(ctx);
            if (rc == PROSPER_PT_OK) rc = collect_retired(ctx);
            if (rc != PROSPER_PT_OK) return rc;
        }
        any = true;
        const prosper_MaterialData previous = ms->materials[first + i];
        ms->materials[first + i] = materials[i];
        const int rc = rebuild_material(ctx, first + i, &previous);
        if (rc != PROSPER_PT_OK) return rc;
    }
    if (!any) return PROSPER_PT_OK;
    PPT_HIP(hipEventRecord(ms->uploaded, ms->uploadStream));
    ms->uploadedRecorded = true;
    ms->pending = true;
    ms->changes++;
    return PROSPER_PT_OK;
}

} // extern "C"
