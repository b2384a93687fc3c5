// prosper_pt.cpp — C-ABI of the MI355X path-tracing reference pass (include/prosper_pt/prosper_pt.h).
//
// Host side of what prosper does around `cb.traceRaysKHR` (src/render/RtReference.cpp:161-383)
// and of the scene/acceleration-structure upload the pass depends on
// (src/scene/World.cpp:468-536,585-802).  There is no CPU fallback: without a usable HIP device
// every entry point fails with PROSPER_PT_ERR_NO_DEVICE.
#include "../../include/prosper_pt/prosper_pt.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <future>
#include <memory>
#include <cmath>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <exception>
#include <stdexcept>
#include <string>
#include <vector>

#include "bvh_build.hpp"
#include "pt_context.hpp"
#include "pt_geometry.hpp"
#include "pt_kernels.hpp"
#include "pt_materials.hpp"
#include "pt_scene.hpp"
#include "pt_tiling.hpp"

using namespace ppt;

namespace ppt
{
thread_local std::string g_lastErrorStorage;
// set by the worker thread of a mesh build: its scene allocations, so that a build that fails can give them back
thread_local std::vector<void *> *g_allocationLog = nullptr;
int fail(int code, const std::string &msg)
{
    g_lastErrorStorage = msg;
    return code;
}
} // namespace ppt

using RenderSlot = prosper_pt_ctx::RenderSlot;
constexpr uint32_t kStageChains = 4; // unnamed interval of the caller's stream: fork .. join of the chains

namespace ppt
{

int device_alloc(prosper_pt_ctx *ctx, size_t bytes, void **out)
{
    if (bytes == 0) bytes = 16;
    void *p = nullptr;
    PPT_HIP(hipMalloc(&p, bytes));
    if (g_allocationLog) g_allocationLog->push_back(p);
    const std::lock_guard<std::mutex> lock(ctx->allocMutex);
    ctx->sceneAllocations.push_back({p, bytes});
    ctx->sceneBytes += bytes;
    *out = p;
    return PROSPER_PT_OK;
}

// returns a scene allocation early (a texture only its material pack still mirrors, a node array that was outgrown)
void device_free(prosper_pt_ctx *ctx, const void *p)
{
    const std::lock_guard<std::mutex> lock(ctx->allocMutex);
    for (size_t i = 0; i < ctx->sceneAllocations.size(); ++i)
        if (ctx->sceneAllocations[i].ptr == p)
        {
            (void)hipFree(ctx->sceneAllocations[i].ptr);
            ctx->sceneBytes -= ctx->sceneAllocations[i].bytes;
            ctx->sceneAllocations.erase(ctx->sceneAllocations.begin() + (long)i);
            return;
        }
}

int upload(prosper_pt_ctx *ctx, const void *src, size_t bytes, void **out)
{
    const int rc = device_alloc(ctx, bytes, out);
    if (rc != PROSPER_PT_OK) return rc;
    if (bytes) PPT_HIP(hipMemcpy(*out, src, bytes, hipMemcpyHostToDevice));
    return PROSPER_PT_OK;
}

} // namespace ppt

namespace
{

WavefrontOptions wavefront_options(const prosper_pt_ctx *ctx)
{
    const prosper_pt_debug_options &d = ctx->debug;
    WavefrontOptions o;
    o.ldsStackEntries = d.ldsStackEntries;
    o.noLdsScene = d.noLdsScene != 0;
    o.noLdsTables = d.noLdsTables != 0;
    o.poolVariant = d.poolVariant;
    o.hipGraph = d.hipGraph != 0;
    return o;
}

} // namespace

namespace
{

void free_scene(prosper_pt_ctx *ctx)
{
    discard_mesh_build(ctx);
    for (ppt::AccelState *old : ctx->retiredAccel) delete old;
    ctx->retiredAccel.clear();
    for (auto &a : ctx->sceneAllocations) (void)hipFree(a.ptr);
    ctx->sceneAllocations.clear();
    delete ctx->accel; // its device arrays are in the list above
    ctx->accel = nullptr;
    delete ctx->lights;
    ctx->lights = nullptr;
    delete ctx->materialState;
    ctx->materialState = nullptr;
    delete ctx->geometry;
    ctx->geometry = nullptr;
    ctx->dTransforms = nullptr;
    ctx->sceneBytes = 0;
    ctx->haveScene = false;
    ctx->scene = DeviceScene{};
}

// Everything the kernels index with is range-checked here, so a malformed view can only fail the
// upload, never fault the GPU.
int validate_scene(const prosper_pt_scene_view *v)
{
    if (v->struct_size != sizeof(prosper_pt_scene_view))
        return fail(PROSPER_PT_ERR_INVALID_ARGUMENT, "scene view struct_size mismatch");
    if (v->geometryBufferCount && (!v->geometryBuffers || !v->geometryBufferByteSizes))
        return fail(PROSPER_PT_ERR_SCENE, "geometry buffers missing");
    if (v->geometryBufferCount > PROSPER_PT_MAX_GEOMETRY_BUFFERS)
        return fail(PROSPER_PT_ERR_SCENE, "more than PROSPER_PT_MAX_GEOMETRY_BUFFERS geometry buffers");
    if (v->meshCount && (!v->geometryMetadatas || !v->meshInfos)) return fail(PROSPER_PT_ERR_SCENE, "mesh tables missing");
    if (v->drawInstanceCount && !v->drawInstances) return fail(PROSPER_PT_ERR_SCENE, "draw instances missing");
    if (v->modelInstanceCount && !v->modelInstanceTransforms) return fail(PROSPER_PT_ERR_SCENE, "transforms missing");
    if (v->materialCount == 0 || !v->materials) return fail(PROSPER_PT_ERR_SCENE, "materials missing (index 0 is required)");
    if (v->samplerCount == 0 || !v->samplers) return fail(PROSPER_PT_ERR_SCENE, "samplers missing (index 0 is required)");
    if (v->textureCount && !v->textures) return fail(PROSPER_PT_ERR_SCENE, "textures missing");
    if (!v->directionalLight || !v->pointLights || !v->spotLights) return fail(PROSPER_PT_ERR_SCENE, "light buffers missing");
    if (v->pointLights->count > PROSPER_MAX_POINT_LIGHT_COUNT || v->spotLights->count > PROSPER_MAX_SPOT_LIGHT_COUNT)
        return fail(PROSPER_PT_ERR_SCENE, "light count exceeds 1024");
    if (v->skybox.texels && v->skybox.faceSize == 0) return fail(PROSPER_PT_ERR_SCENE, "skybox face size is zero");

    for (uint32_t i = 0; i < v->textureCount; ++i)
    {
        const prosper_pt_texture_desc &t = v->textures[i];
        if (!t.texels || t.width == 0 || t.height == 0 || t.format > PROSPER_PT_FORMAT_BC7_UNORM)
            return fail(PROSPER_PT_ERR_SCENE, "texture " + std::to_string(i) + " is invalid");
        if (t.format == PROSPER_PT_FORMAT_BC7_UNORM && (t.width % 4u != 0u || t.height % 4u != 0u))
            return fail(PROSPER_PT_ERR_SCENE, "BC7 texture " + std::to_string(i) + " is not a whole number of 4x4 blocks");
    }
    for (uint32_t i = 0; i < v->samplerCount; ++i)
    {
        const prosper_pt_sampler_desc &s = v->samplers[i];
        if (s.magFilter > PROSPER_PT_FILTER_LINEAR || s.wrapS > PROSPER_PT_WRAP_CLAMP_TO_EDGE ||
            s.wrapT > PROSPER_PT_WRAP_CLAMP_TO_EDGE)
            return fail(PROSPER_PT_ERR_SCENE, "sampler " + std::to_string(i) + " is invalid");
    }
    for (uint32_t i = 0; i < v->materialCount; ++i)
    {
        const prosper_MaterialData &m = v->materials[i];
        const uint32_t ts[3] = {m.baseColorTextureSampler, m.metallicRoughnessTextureSampler, m.normalTextureSampler};
        for (uint32_t k = 0; k < 3; ++k)
        {
            const uint32_t tex = ts[k] & 0xFFFFFFu, smp = ts[k] >> 24;
            if (tex > 0 && (tex >= v->textureCount || smp >= v->samplerCount))
                return fail(PROSPER_PT_ERR_SCENE, "material " + std::to_string(i) + " references a missing texture/sampler");
        }
        if (m.alphaMode > PROSPER_ALPHA_MODE_BLEND) return fail(PROSPER_PT_ERR_SCENE, "material alpha mode invalid");
    }
    for (uint32_t i = 0; i < v->meshCount; ++i)
    {
        const prosper_GeometryMetadata &m = v->geometryMetadatas[i];
        const prosper_pt_mesh_info &info = v->meshInfos[i];
        const std::string name = "mesh " + std::to_string(i);
        if (m.bufferIndex == PROSPER_PT_ABSENT) continue; // not loaded yet (prosper_pt_update_meshes): nothing else of it is read
        if (m.bufferIndex >= v->geometryBufferCount) return fail(PROSPER_PT_ERR_SCENE, name + ": bufferIndex out of range");
        if (info.materialIndex >= v->materialCount) return fail(PROSPER_PT_ERR_SCENE, name + ": materialIndex out of range");
        if (info.indexCount % 3 != 0) return fail(PROSPER_PT_ERR_SCENE, name + ": indexCount is not a multiple of 3");
        const uint64_t words = v->geometryBufferByteSizes[m.bufferIndex] / 4;
        if (m.indicesOffset == PROSPER_PT_ABSENT || m.positionsOffset == PROSPER_PT_ABSENT)
            return fail(PROSPER_PT_ERR_SCENE, name + ": indices/positions are required");
        const uint64_t indexEndWords = m.usesShortIndices == 1 ? ((uint64_t)m.indicesOffset + info.indexCount + 1) / 2
                                                               : (uint64_t)m.indicesOffset + info.indexCount;
        if (indexEndWords > words) return fail(PROSPER_PT_ERR_SCENE, name + ": indices run past the geometry buffer");
        if ((uint64_t)m.positionsOffset + 2ull * info.vertexCount > words)
            return fail(PROSPER_PT_ERR_SCENE, name + ": positions run past the geometry buffer");
        const uint32_t attrs[3] = {m.normalsOffset, m.tangentsOffset, m.texCoord0sOffset};
        for (uint32_t k = 0; k < 3; ++k)
            if (attrs[k] != PROSPER_PT_ABSENT && (uint64_t)attrs[k] + info.vertexCount > words)
                return fail(PROSPER_PT_ERR_SCENE, name + ": attribute stream runs past the geometry buffer");
        const void *buffer = v->geometryBuffers[m.bufferIndex];
        for (uint32_t k = 0; k < info.indexCount; ++k)
        {
            const uint32_t idx = m.usesShortIndices == 1 ? (uint32_t) static_cast<const uint16_t *>(buffer)[m.indicesOffset + k]
                                                          : static_cast<const uint32_t *>(buffer)[m.indicesOffset + k];
            if (idx >= info.vertexCount) return fail(PROSPER_PT_ERR_SCENE, name + ": vertex index out of range");
        }
    }
    for (uint32_t i = 0; i < v->drawInstanceCount; ++i)
    {
        const prosper_DrawInstance &d = v->drawInstances[i];
        if (d.meshIndex >= v->meshCount || d.materialIndex >= v->materialCount || d.modelInstanceIndex >= v->modelInstanceCount)
            return fail(PROSPER_PT_ERR_SCENE, "draw instance " + std::to_string(i) + " references a missing mesh/material/transform");
    }
    return PROSPER_PT_OK;
}

int upload_scene_impl(prosper_pt_ctx *ctx, const prosper_pt_scene_view *v)
{
    DeviceScene &s = ctx->scene;
    int rc = PROSPER_PT_OK;
    ctx->stats = prosper_pt_scene_stats{};
    const auto tUpload = std::chrono::steady_clock::now();
    auto seconds_since = [](std::chrono::steady_clock::time_point t) {
        return std::chrono::duration<double>(std::chrono::steady_clock::now() - t).count();
    };

    // bindless geometry buffers + pointer table, with the mirrors a later prosper_pt_update_meshes works from
    GeometryState *gs = new (std::nothrow) GeometryState();
    if (!gs) return fail(PROSPER_PT_ERR_INVALID_ARGUMENT, "out of host memory");
    ctx->geometry = gs;
    gs->metadatas.assign(v->geometryMetadatas, v->geometryMetadatas + v->meshCount);
    gs->infos.assign(v->meshInfos, v->meshInfos + v->meshCount);
    gs->drawInstances.assign(v->drawInstances, v->drawInstances + v->drawInstanceCount);
    std::vector<const void *> bufferPtrs(std::max<size_t>(v->geometryBufferCount, PROSPER_PT_MAX_GEOMETRY_BUFFERS), nullptr);
    void *d = nullptr;
    // (fixed sizes: while a mesh build runs the calling thread and the worker both index these, neither may move them)
    gs->buffers.assign(bufferPtrs.size(), nullptr);
    gs->bufferBytes.assign(bufferPtrs.size(), 0);
    for (uint32_t i = 0; i < v->geometryBufferCount; ++i)
    {
        if ((rc = upload(ctx, v->geometryBuffers[i], (size_t)v->geometryBufferByteSizes[i], &d))) return rc;
        bufferPtrs[i] = d;
        gs->buffers[i] = d;
        gs->bufferBytes[i] = v->geometryBufferByteSizes[i];
    }
    if ((rc = upload(ctx, bufferPtrs.data(), bufferPtrs.size() * sizeof(void *), &d))) return rc;
    gs->dBufferTable = static_cast<const void **>(d);
    s.geometryBuffers = gs->dBufferTable;
    if ((rc = upload(ctx, v->geometryMetadatas, sizeof(prosper_GeometryMetadata) * v->meshCount, &d))) return rc;
    gs->dMetadatas = static_cast<prosper_GeometryMetadata *>(d);
    s.geometryMetadatas = gs->dMetadatas;
    if ((rc = upload(ctx, v->drawInstances, sizeof(prosper_DrawInstance) * v->drawInstanceCount, &d))) return rc;
    s.drawInstances = static_cast<const prosper_DrawInstance *>(d);
    if ((rc = upload(ctx, v->modelInstanceTransforms, sizeof(prosper_ModelInstanceTransforms) * v->modelInstanceCount, &d)))
        return rc;
    s.modelInstanceTransforms = static_cast<const prosper_ModelInstanceTransforms *>(d);
    if ((rc = upload(ctx, v->materials, sizeof(prosper_MaterialData) * v->materialCount, &d))) return rc;
    s.materials = static_cast<const prosper_MaterialData *>(d);
    if ((rc = upload(ctx, v->samplers, sizeof(prosper_pt_sampler_desc) * v->samplerCount, &d))) return rc;
    s.samplers = static_cast<const prosper_pt_sampler_desc *>(d);
    s.drawInstanceCount = v->drawInstanceCount;
    s.modelInstanceCount = v->modelInstanceCount;

    // ---- world triangles first: the host-side hierarchy build - the longest step of an upload - runs on the host's threads
    //      while this thread goes on with textures, packs, sky, lights and the alpha tables (begin_geometry); the flatten
    //      kernel runs again in finish_geometry for the shading and any-hit records (the same world triangles a second time) ----
    AccelState *acc = new (std::nothrow) AccelState();
    if (!acc) return fail(PROSPER_PT_ERR_INVALID_ARGUMENT, "out of host memory");
    ctx->accel = acc;
    acc->transforms.assign(v->modelInstanceTransforms, v->modelInstanceTransforms + v->modelInstanceCount);
    ctx->dTransforms = const_cast<prosper_ModelInstanceTransforms *>(s.modelInstanceTransforms);
    acc->dTransformsV[0] = ctx->dTransforms;
    GeometryJob job;
    GeometryTarget target = context_target(ctx);
    {
        GeometryLayout layout;
        if ((rc = layout_geometry(*gs, v->materials, layout))) return rc;
        if ((rc = begin_geometry(ctx, target, layout, nullptr, job))) return rc;
    }

    // textures: one allocation each (256-B aligned by hipMalloc), re-laid out in 8x4-texel tiles of one cache line each
    // (pt_scene.hpp DeviceTexture) by a kernel - until round 4 the host did that a texel at a time, 40 ms for
    // S-sponza-class - + descriptor table.  The tables a later prosper_pt_update_textures / _materials changes are mirrored
    // in ctx->materialState (pt_materials.cpp).
    const auto tTextures = std::chrono::steady_clock::now();
    MaterialState *ms = new (std::nothrow) MaterialState();
    if (!ms) return fail(PROSPER_PT_ERR_INVALID_ARGUMENT, "out of host memory");
    ctx->materialState = ms;
    ms->materials.assign(v->materials, v->materials + v->materialCount);
    ms->samplers.assign(v->samplers, v->samplers + v->samplerCount);
    std::vector<DeviceTexture> &textures = ms->textures;
    textures.assign(v->textureCount ? v->textureCount : 1, DeviceTexture{nullptr, 0u, 0u, 0u, 0u});
    uint64_t texelBytes = 0;
    {
        size_t stagingBytes = 0;
        for (uint32_t i = 0; i < v->textureCount; ++i) stagingBytes = std::max(stagingBytes, texture_staging_bytes(v->textures[i]));
        void *staging = nullptr;
        if (stagingBytes) PPT_HIP(hipMalloc(&staging, stagingBytes));
        for (uint32_t i = 0; i < v->textureCount && rc == PROSPER_PT_OK; ++i)
        {
            texelBytes += (uint64_t)v->textures[i].width * v->textures[i].height * 4u;
            // (one staging area, one stream: the copy of texture i + 1 queues behind the kernel that reads texture i)
            rc = create_device_texture(ctx, v->textures[i], staging, nullptr, &textures[i]);
        }
        const hipError_t e = hipDeviceSynchronize();
        if (staging) (void)hipFree(staging);
        if (rc != PROSPER_PT_OK) return rc;
        PPT_HIP(e);
    }
    ms->texelBytes = texelBytes;
    // beyond the 8 x 4 MB of L2 the texels of a hit come from the Infinity Cache or HBM: overlap their fetches
    // (debug option batchedTextures = 0 / 1 forces either path: same pixels, tested)
    s.batchedTextures = texel_set_is_big(texelBytes) ? 1u : 0u;
    if (ctx->debug.batchedTextures >= 0) s.batchedTextures = ctx->debug.batchedTextures ? 1u : 0u;
    // material texture packs (pt_scene.hpp MaterialPack): base / MR / normal interleaved per texel where a material's
    // three textures share extent and sampler.  Debug option noTexturePacks keeps every material unpacked.
    // Compact packs where the texels outgrow the caches (the threshold of the batched loads above): on a small texture set
    // the bytes are not what the shade kernel waits for, and two kinds of pack in one wave cost a divergent branch
    // (FlightHelmet fixture: +1 % on the step).  Debug option widePacks = 1 / 0 forces the 16-byte / the compact pack.
    std::vector<MaterialPack> &packs = ms->packs;
    packs.assign(v->materialCount, MaterialPack{nullptr, 0u, 0u, 0u, 0u});
    uint32_t packedMaterials = 0;
    bool widePacks = !texel_set_is_big(texelBytes);
    if (ctx->debug.widePacks >= 0) widePacks = ctx->debug.widePacks != 0;
    for (uint32_t i = 0; i < v->materialCount; ++i)
    {
        if ((rc = build_material_pack(ctx, v->materials[i], textures, ctx->debug.noTexturePacks != 0, widePacks, nullptr, &packs[i]))) return rc;
        packedMaterials += packs[i].texels ? 1u : 0u;
    }
    PPT_HIP(hipGetLastError());
    PPT_HIP(hipDeviceSynchronize());
    if ((rc = upload(ctx, packs.data(), packs.size() * sizeof(MaterialPack), &d))) return rc;
    s.materialPacks = static_cast<const MaterialPack *>(d);
    ctx->packedMaterials = packedMaterials;
    ms->packedMaterials = packedMaterials;
    // A texture that only packed materials use is never sampled by itself again (sample_material takes the pack): its
    // own copy goes - half of a packed scene's texel memory.  What keeps a texture: an unpacked material, or the any-hit
    // of a MASK / BLEND material, which reads the base colour's alpha out of the texture itself.
    {
        std::vector<uint8_t> needed(textures.size(), 0);
        needed[0] = 1;
        for (uint32_t i = 0; i < v->materialCount; ++i)
        {
            const prosper_MaterialData &m = v->materials[i];
            const uint32_t t3[3] = {m.baseColorTextureSampler & 0xFFFFFFu, m.metallicRoughnessTextureSampler & 0xFFFFFFu,
                                    m.normalTextureSampler & 0xFFFFFFu};
            if (packs[i].texels == nullptr)
                for (uint32_t t : t3) needed[t] = 1;
            if (m.alphaMode != PROSPER_ALPHA_MODE_OPAQUE) needed[t3[0]] = 1;
        }
        // (a texture no material samples yet stays: it is what a streamed-in material will point at)
        std::vector<uint8_t> sampled(textures.size(), 0);
        for (uint32_t i = 0; i < v->materialCount; ++i)
        {
            const prosper_MaterialData &m = v->materials[i];
            sampled[m.baseColorTextureSampler & 0xFFFFFFu] = sampled[m.metallicRoughnessTextureSampler & 0xFFFFFFu] =
                sampled[m.normalTextureSampler & 0xFFFFFFu] = 1;
        }
        for (uint32_t i = 1; i < v->textureCount; ++i)
            if (!needed[i] && sampled[i] && textures[i].texels)
            {
                device_free(ctx, textures[i].texels);
                textures[i].texels = nullptr;
            }
    }
    if ((rc = upload(ctx, textures.data(), textures.size() * sizeof(DeviceTexture), &d))) return rc;
    s.textures = static_cast<const DeviceTexture *>(d);
    const double textureSeconds = seconds_since(tTextures);

    // lights: one block (directional, point list, spot list), the first of up to three versions
    {
        LightState *ls = new (std::nothrow) LightState();
        if (ls) ls->mirror = new (std::nothrow) LightBlock();
        if (!ls || !ls->mirror)
        {
            delete ls;
            return fail(PROSPER_PT_ERR_INVALID_ARGUMENT, "out of host memory");
        }
        ctx->lights = ls;
        ls->mirror->directional = *v->directionalLight;
        ls->mirror->points = *v->pointLights;
        ls->mirror->spots = *v->spotLights;
        if ((rc = upload(ctx, ls->mirror, sizeof(LightBlock), &d))) return rc;
        ls->dBlocks[0] = static_cast<LightBlock *>(d);
        s.directionalLight = &ls->dBlocks[0]->directional;
        s.pointLights = &ls->dBlocks[0]->points;
        s.spotLights = &ls->dBlocks[0]->spots;
    }
    s.pointLightCount = v->pointLights->count;
    s.spotLightCount = v->spotLights->count;
    s.materialCount = v->materialCount;

    // skybox
    s.skybox = nullptr;
    s.skyboxFaceSize = 0;
    if (v->skybox.texels)
    {
        // kept with a one-texel seamless border per face (pt_device.hpp fetch_cube_rgb); the plain copy is only the source
        const size_t bytes = 6ull * v->skybox.faceSize * v->skybox.faceSize * 4u * sizeof(uint16_t);
        const size_t n2 = (size_t)v->skybox.faceSize + 2u;
        void *plain = nullptr;
        if ((rc = upload(ctx, v->skybox.texels, bytes, &plain))) return rc;
        if ((rc = device_alloc(ctx, 6ull * n2 * n2 * 4u * sizeof(uint16_t), &d))) return rc;
        launch_border_skybox(static_cast<const uint16_t *>(plain), v->skybox.faceSize, d, nullptr);
        PPT_HIP(hipGetLastError());
        PPT_HIP(hipDeviceSynchronize());
        device_free(ctx, plain);
        s.skybox = static_cast<const uint16_t *>(d);
        s.skyboxFaceSize = v->skybox.faceSize;
    }

    // ---- acceleration structure (replaces buildNextBlas/buildCurrentTlas, World.cpp:585-802) ----
    // ---- what the any-hit shader reads: one 32-byte record per non-opaque triangle, one per material, and the
    //      materials' alpha bounds (pt_scene.hpp AlphaTriangle / AlphaMaterial) ----
    {
        std::vector<AlphaMaterial> &alphaMaterials = ms->alphaMaterials;
        alphaMaterials.assign(v->materialCount ? v->materialCount : 1, AlphaMaterial{});
        uint64_t boundBytes = 0;
        for (uint32_t i = 0; i < v->materialCount; ++i)
        {
            uint64_t bytes = 0;
            if ((rc = build_alpha_material(ctx, v->materials[i], textures, ms->samplers, nullptr, &alphaMaterials[i], &bytes))) return rc;
            boundBytes += bytes;
        }
        PPT_HIP(hipGetLastError());
        if ((rc = upload(ctx, alphaMaterials.data(), alphaMaterials.size() * sizeof(AlphaMaterial), &d))) return rc;
        s.alphaMaterials = static_cast<const AlphaMaterial *>(d);
        ctx->alphaBoundBytes = boundBytes;
        ms->alphaBoundBytes = boundBytes;
        // layout of the table block of a later version (pt_materials.cpp flush_pending_materials)
        auto align16 = [](size_t n) { return (n + 15u) & ~(size_t)15u; };
        ms->packsOffset = align16(ms->materials.size() * sizeof(prosper_MaterialData));
        ms->alphaOffset = ms->packsOffset + align16(ms->packs.size() * sizeof(MaterialPack));
        ms->texturesOffset = ms->alphaOffset + align16(ms->alphaMaterials.size() * sizeof(AlphaMaterial));
        ms->blockBytes = ms->texturesOffset + align16(ms->textures.size() * sizeof(DeviceTexture));
    }

    if ((rc = finish_geometry(ctx, target, job))) return rc;
    ctx->rawRecords = target.rawRecords;
    ctx->stats.deviceBytes = ctx->sceneBytes;
    ctx->stats.alphaBoundBytes = ctx->alphaBoundBytes;
    ctx->sceneStamp++;
    ctx->stats.textureSeconds = textureSeconds;
    ctx->stats.uploadSeconds = seconds_since(tUpload);
    return PROSPER_PT_OK;
}

// A slot's workspace may be reused once the kernels of its previous user are done: `free` is recorded behind them.
void wait_for_slot(RenderSlot &slot, hipStream_t stream)
{
    if (slot.freeRecorded) (void)hipStreamWaitEvent(stream, slot.free, 0);
}
void release_slot(RenderSlot &slot, hipStream_t stream)
{
    (void)hipEventRecord(slot.free, stream);
    slot.freeRecorded = true;
}

// Sizes and carves the wavefront workspace for `frames` x (tilesX*tilesY*64) path slots.
int ensure_wavefront_workspace(
    prosper_pt_ctx *ctx, RenderSlot &slot, uint32_t tilesX, uint32_t tilesY, uint32_t frames, bool pipelined, bool banded,
    WavefrontBuffers *out)
{
    const uint64_t pixelsPadded = (uint64_t)tilesX * tilesY * 64u;
    const uint64_t slots = pixelsPadded * frames;
    // One segment per wave.  Segments take their 8x8 tiles strided over the whole batch (wf_generate_extend), so
    // all waves carry statistically equal work and there is nothing to balance dynamically: the best segment
    // count is about one round of resident waves for each of the two launch chains plus a little, ~11 500 segments (swept:
    // profiles/r01_seglen_sweep.txt).  The length is an ODD multiple of 64 slots: at even multiples the waves'
    // concurrent accesses to their segments' records, `segLen * 16` bytes apart, pile onto a few HBM channels.
    // (frames in flight, one chain each: the same total split over the kRenderSlots frames - 3 700)
    // (round 3, small batches re-swept - profiles/r03_seglen_sweep.txt: a 1-spp frame of C2 0.326 -> 0.317 ms at 2560 instead of
    // 3700 segments, FlightHelmet 0.524 -> 0.499, a 1/8 rank share 0.330 -> 0.317, C3 1.826 -> 1.867: 3000)
    uint64_t target = pipelined ? 3000u : 11500u;
    if (ctx->debug.segments >= 64u && ctx->debug.segments <= (1u << 20)) target = ctx->debug.segments; // tuning hook
    uint64_t batches = (slots / target + 32u) / 64u;
    if (batches < 2u) batches = 2u; // small renders: 128-slot segments
    if (batches > 2u && batches % 2u == 0u) batches += 1u;
    // At most 49 batches (3136 slots) per segment.  Re-swept in round 3 (profiles/r03_seglen_sweep.txt; the cap was 33 since
    // round 1): longer per-wave streams keep the later stages' waves fuller - what a sparse image (FlightHelmet: one camera
    // ray in seven hits anything) and expensive, uneven rays (C4's foliage) need - while a dense, texture-heavy scene (C3)
    // prefers short segments, whose waves work on neighbouring tiles.  8 spp, three frames in flight, 33 -> 49 -> 65 batches:
    // FlightHelmet 2.04 -> 1.91 -> 1.87 ms, C4 27.9 -> 27.05 -> 27.15, C2 1.95 -> 1.96 -> 1.94, C3 12.93 -> 13.07 -> 13.14.
    if (batches > 49u) batches = 49u;
    uint64_t segLen = batches * 64u;
    {
        const uint64_t v = ctx->debug.segmentLength; // tuning / test hook
        if (v >= 64u && v <= 8192u && v % 64u == 0u) segLen = v;
    }
    uint64_t nSeg = (slots + segLen - 1u) / segLen;
    // The stride between a segment's tiles is nSeg tiles.  A stride that is nearly a whole number of tile rows
    // would keep a segment in the same few tile columns (correlated work, the balance is gone): add segments
    // until the column step is at least an eighth of a row away from 0.
    if (tilesX >= 16u)
        for (uint32_t tries = 0; tries < tilesX; ++tries)
        {
            const uint64_t cols = (nSeg % ((uint64_t)tilesX * tilesY)) % tilesX;
            if (cols >= tilesX / 8u && tilesX - cols >= tilesX / 8u) break;
            ++nSeg;
        }
    // Banded batches: with a launch that covers all segment groups, XCD x runs the groups [x * perXcd, (x + 1) * perXcd)
    // (pt_wavefront.hip my_segment), i.e. the segments [x * bandSegments, ...).  Each band of segments takes the batches of
    // its own band of tiles - as many tiles as its segments have room for (batches per segment / batches per tile); when
    // rounding leaves a tile without a place, one more round of segments makes room.
    uint32_t bandSegments = 0, bandTile[9] = {};
    if (banded)
    {
        const uint64_t tiles = (uint64_t)tilesX * tilesY, perSegment = segLen / 64u; // batches a segment holds
        for (uint32_t tries = 0; tries < 64u; ++tries)
        {
            const uint64_t groups = (nSeg + 3u) / 4u, perXcd = (groups + 7u) / 8u;
            bandSegments = (uint32_t)(perXcd * 4u);
            uint64_t placed = 0;
            for (uint32_t x = 0; x < 8u; ++x)
            {
                const uint64_t first = (uint64_t)x * bandSegments;
                const uint64_t segs = first >= nSeg ? 0u : std::min<uint64_t>(bandSegments, nSeg - first);
                const uint64_t room = segs * perSegment / frames; // tiles: `frames` batches each
                // what is left, spread evenly over the bands that remain
                const uint64_t want = (tiles - placed + (8u - x) - 1u) / (8u - x);
                bandTile[x] = (uint32_t)placed;
                placed += std::min(room, want);
            }
            bandTile[8] = (uint32_t)placed;
            if (placed >= tiles) break;
            nSeg += 8u;
            bandSegments = 0;
        }
        if (bandSegments == 0) return fail(PROSPER_PT_ERR_UNSUPPORTED, "banded batches: no segment layout covers the image");
    }
    const uint64_t padded = nSeg * segLen;
    // per slot: 2 x (3 x 16 + 8) B ping-pong state, camera slot 4, hit 16 + idx 4, shadow 48, colour 16; + 3 counters per segment
    const size_t bytes = (size_t)padded * (2u * (3u * 16u + 8u) + 4u + 16u + 4u + 48u + 16u) + (size_t)nSeg * 12u + 4096u;
    if (bytes > slot.wfBytes)
    {
        PPT_HIP(hipDeviceSynchronize()); // the other slot's render may be in flight on its own streams
        if (slot.wfBlock) PPT_HIP(hipFree(slot.wfBlock));
        slot.wfBlock = nullptr;
        slot.wfBytes = 0;
        PPT_HIP(hipMalloc(&slot.wfBlock, bytes));
        slot.wfBytes = bytes;
    }
    uint8_t *cursor = static_cast<uint8_t *>(slot.wfBlock);
    auto carve = [&](size_t n) {
        void *r = cursor;
        cursor += (n + 255u) & ~(size_t)255u;
        return r;
    };
    WavefrontBuffers w = {};
    for (int k = 0; k < 2; ++k)
    {
        w.rayA[k] = static_cast<float4 *>(carve(padded * 16u));
        w.rayB[k] = static_cast<float4 *>(carve(padded * 16u));
        w.pathT[k] = static_cast<float4 *>(carve(padded * 16u));
        w.pathR[k] = static_cast<uint2 *>(carve(padded * 8u));
    }
    w.cameraSlot = static_cast<uint32_t *>(carve(padded * 4u));
    w.hit = static_cast<uint4 *>(carve(padded * 16u));
    w.hitIdx = static_cast<uint32_t *>(carve(padded * 4u));
    w.shA = static_cast<float4 *>(carve(padded * 16u));
    w.shB = static_cast<float4 *>(carve(padded * 16u));
    w.shC = static_cast<float4 *>(carve(padded * 16u));
    w.color = static_cast<float4 *>(carve(padded * 16u));
    w.segRays = static_cast<uint32_t *>(carve(nSeg * 4u));
    w.segHits = static_cast<uint32_t *>(carve(nSeg * 4u));
    w.segShadow = static_cast<uint32_t *>(carve(nSeg * 4u));
    w.segLen = (uint32_t)segLen;
    w.nSeg = (uint32_t)nSeg;
    w.pixelsPadded = (uint32_t)pixelsPadded;
    w.tilesX = tilesX;
    w.tilesY = tilesY;
    w.bandSegments = bandSegments;
    for (uint32_t x = 0; x < 9u; ++x) w.bandTile[x] = bandTile[x];
    if ((size_t)(cursor - static_cast<uint8_t *>(slot.wfBlock)) > slot.wfBytes + 0u)
    {
        // carve() rounds every array up to 256 B: re-allocate with the exact carved size
        const size_t need = (size_t)(cursor - static_cast<uint8_t *>(slot.wfBlock));
        PPT_HIP(hipDeviceSynchronize()); // the other slot's render may be in flight on its own streams
        PPT_HIP(hipFree(slot.wfBlock));
        slot.wfBlock = nullptr;
        slot.wfBytes = 0;
        PPT_HIP(hipMalloc(&slot.wfBlock, need));
        slot.wfBytes = need;
        return ensure_wavefront_workspace(ctx, slot, tilesX, tilesY, frames, pipelined, banded, out);
    }
    *out = w;
    return PROSPER_PT_OK;
}

// Makes sure the global stack-overflow array covers `gridBlocks` workgroups of 256 lanes for a kernel
// whose LDS stack holds `ldsEntries` entries; returns nullptr when the tree never needs more.
static int ensure_scratch_dwords(
    prosper_pt_ctx *ctx, RenderSlot &slot, size_t dwordsPerBlock, uint32_t gridBlocks, hipStream_t stream, int32_t **out);
int ensure_stack_overflow(
    prosper_pt_ctx *ctx, RenderSlot &slot, uint32_t ldsEntries, uint32_t gridBlocks, hipStream_t stream, int32_t **out)
{
    const uint32_t bound = ctx->stats.maxDepth;
    return ensure_scratch_dwords(ctx, slot, bound <= ldsEntries ? 0u : (size_t)(bound - ldsEntries) * 256u, gridBlocks, stream, out);
}
// `dwordsPerBlock` ints of kernel scratch per workgroup (stack overflow columns, ray-pool records)
static int ensure_scratch_dwords(
    prosper_pt_ctx *ctx, RenderSlot &slot, size_t dwordsPerBlock, uint32_t gridBlocks, hipStream_t stream, int32_t **out)
{
    (void)ctx;
    *out = nullptr;
    if (dwordsPerBlock == 0u) return PROSPER_PT_OK;
    const size_t bytes = dwordsPerBlock * gridBlocks * sizeof(int32_t);
    if (bytes > slot.stackOverflowBytes)
    {
        (void)stream;
        PPT_HIP(hipDeviceSynchronize());
        if (slot.stackOverflow) PPT_HIP(hipFree(slot.stackOverflow));
        slot.stackOverflow = nullptr;
        slot.stackOverflowBytes = 0;
        PPT_HIP(hipMalloc((void **)&slot.stackOverflow, bytes));
        slot.stackOverflowBytes = bytes;
    }
    *out = slot.stackOverflow;
    return PROSPER_PT_OK;
}

uint32_t compute_local_width(uint32_t width, const prosper_pt_tile_desc *tile)
{
    if (!tile || tile->stripeCount <= 1 || tile->stripeWidth == 0) return width;
    uint32_t n = 0;
    for (uint32_t x = 0; x < width; x += tile->stripeWidth)
    {
        if ((x / tile->stripeWidth) % tile->stripeCount == tile->stripeIndex)
            n += (x + tile->stripeWidth <= width) ? tile->stripeWidth : (width - x);
    }
    return n;
}

// ---- debug options: validation, and the one opt-in reading of the environment (prosper_pt_create) ----

int check_debug_options(const prosper_pt_debug_options &o)
{
#ifndef PPT_EXPERIMENTS
    if (o.poolVariant || o.rawRecords || o.tileOrder || o.hipGraph || o.pipelinedChains || o.mergeLimit)
        return fail(PROSPER_PT_ERR_UNSUPPORTED, "debug options: an experiment was requested, but the library was built without -DPPT_EXPERIMENTS");
#endif
    if (o.ldsStackEntries != 0u && o.ldsStackEntries != 16u && o.ldsStackEntries != 24u && o.ldsStackEntries != 32u)
        return fail(PROSPER_PT_ERR_INVALID_ARGUMENT, "debug options: ldsStackEntries is 0, 16, 24 or 32");
    if (o.segmentLength != 0u && (o.segmentLength < 64u || o.segmentLength > 8192u || o.segmentLength % 64u != 0u))
        return fail(PROSPER_PT_ERR_INVALID_ARGUMENT, "debug options: segmentLength is a multiple of 64 in [64, 8192]");
    if (o.chains > kMaxChains) return fail(PROSPER_PT_ERR_INVALID_ARGUMENT, "debug options: at most three launch chains");
    if (o.leafSize > 8u || o.alphaCellShift > 15 || o.nodeOrder > 2 || o.poolVariant > 3u)
        return fail(PROSPER_PT_ERR_INVALID_ARGUMENT, "debug options: a value is out of range");
    if (!(o.sahTraversalCost >= 0.0f) || !(o.boxPad >= 0.0f) || !(o.rebuildCostRatio >= 0.0f))
        return fail(PROSPER_PT_ERR_INVALID_ARGUMENT, "debug options: a coefficient is negative or NaN");
    return PROSPER_PT_OK;
}

// name -> field, for the "name=value,name=value" form of the environment gate
struct DebugField
{
    const char *name;
    size_t offset;
    char kind; // 'u' uint32, 'i' int32, 'f' float
};
#define PPT_FIELD(member, kind) {#member, offsetof(prosper_pt_debug_options, member), kind}
const DebugField kDebugFields[] = {
    PPT_FIELD(batchedTextures, 'i'), PPT_FIELD(widePacks, 'i'), PPT_FIELD(alphaCellShift, 'i'), PPT_FIELD(noTexturePacks, 'u'),
    PPT_FIELD(noAlphaBounds, 'u'), PPT_FIELD(noUploadRefit, 'u'), PPT_FIELD(flatBvh, 'u'), PPT_FIELD(sahTraversalCost, 'f'),
    PPT_FIELD(boxPad, 'f'), PPT_FIELD(leafSize, 'u'), PPT_FIELD(buildThreads, 'u'), PPT_FIELD(topEntries, 'u'),
    PPT_FIELD(nodeOrder, 'i'), PPT_FIELD(childOrder, 'i'), PPT_FIELD(buildTiming, 'u'), PPT_FIELD(segments, 'u'),
    PPT_FIELD(segmentLength, 'u'), PPT_FIELD(chains, 'u'), PPT_FIELD(ldsStackEntries, 'u'), PPT_FIELD(noLdsScene, 'u'),
    PPT_FIELD(noLdsTables, 'u'), PPT_FIELD(traceDeadPaths, 'u'), PPT_FIELD(bandedBatches, 'i'), PPT_FIELD(rebuildCostRatio, 'f'), PPT_FIELD(alwaysRebuild, 'u'),
    PPT_FIELD(failNextUpdate, 'u'), PPT_FIELD(poolVariant, 'u'), PPT_FIELD(rawRecords, 'u'), PPT_FIELD(tileOrder, 'u'),
    PPT_FIELD(hipGraph, 'u'), PPT_FIELD(pipelinedChains, 'u'), PPT_FIELD(mergeLimit, 'u'),
};
#undef PPT_FIELD

bool parse_debug_options(const char *text, prosper_pt_debug_options *out, std::string *err)
{
    std::string rest = text ? text : "";
    while (!rest.empty())
    {
        const size_t comma = rest.find(',');
        const std::string item = rest.substr(0, comma);
        rest = comma == std::string::npos ? std::string() : rest.substr(comma + 1);
        if (item.empty()) continue;
        const size_t eq = item.find('=');
        const std::string name = item.substr(0, eq), value = eq == std::string::npos ? std::string("1") : item.substr(eq + 1);
        const DebugField *field = nullptr;
        for (const DebugField &f : kDebugFields)
            if (name == f.name) field = &f;
        if (!field)
        {
            *err = "debug options: unknown option '" + name + "'";
            return false;
        }
        char *end = nullptr;
        uint8_t *at = reinterpret_cast<uint8_t *>(out) + field->offset;
        if (field->kind == 'f')
        {
            const float v = std::strtof(value.c_str(), &end);
            std::memcpy(at, &v, sizeof(v));
        }
        else if (field->kind == 'i')
        {
            const int32_t v = (int32_t)std::strtol(value.c_str(), &end, 10);
            std::memcpy(at, &v, sizeof(v));
        }
        else
        {
            const uint32_t v = (uint32_t)std::strtoul(value.c_str(), &end, 10);
            std::memcpy(at, &v, sizeof(v));
        }
        if (end == value.c_str() || *end != '\0')
        {
            *err = "debug options: '" + value + "' is not a value for '" + name + "'";
            return false;
        }
    }
    return true;
}

bool debug_options_from_environment(prosper_pt_debug_options *out, std::string *err)
{
    const char *gate = std::getenv("PROSPER_PT_DEBUG");
    if (!gate || std::strcmp(gate, "1") != 0) return true;
    const char *text = std::getenv("PROSPER_PT_DEBUG_OPTIONS");
    if (!text) return true;
    if (!parse_debug_options(text, out, err)) return false;
    if (check_debug_options(*out) != PROSPER_PT_OK)
    {
        *err = ppt::g_lastErrorStorage;
        return false;
    }
    return true;
}

} // namespace

extern "C" {

const char *prosper_pt_last_error(void) { return ppt::g_lastErrorStorage.c_str(); }
uint32_t prosper_pt_abi_version(void) { return PROSPER_PT_ABI_VERSION; }

uint32_t prosper_pt_has_experiments(void)
{
#ifdef PPT_EXPERIMENTS
    return 1u;
#else
    return 0u;
#endif
}

void prosper_pt_debug_options_default(prosper_pt_debug_options *out)
{
    if (!out) return;
    *out = prosper_pt_debug_options{};
    out->struct_size = (uint32_t)sizeof(prosper_pt_debug_options);
    out->batchedTextures = -1;
    out->widePacks = -1;
    out->alphaCellShift = -1;
    out->nodeOrder = -1;
    out->childOrder = -1;
    out->bandedBatches = -1;
}

int prosper_pt_set_debug_options(prosper_pt_ctx *ctx, const prosper_pt_debug_options *options)
{
    if (!ctx || !options) return fail(PROSPER_PT_ERR_INVALID_ARGUMENT, "prosper_pt_set_debug_options: null argument");
    if (options->struct_size != sizeof(prosper_pt_debug_options))
        return fail(PROSPER_PT_ERR_INVALID_ARGUMENT, "prosper_pt_set_debug_options: struct_size mismatch");
    const int rc = check_debug_options(*options);
    if (rc != PROSPER_PT_OK) return rc;
    ctx->debug = *options;
    return PROSPER_PT_OK;
}

int prosper_pt_get_debug_options(prosper_pt_ctx *ctx, prosper_pt_debug_options *out)
{
    if (!ctx || !out) return fail(PROSPER_PT_ERR_INVALID_ARGUMENT, "prosper_pt_get_debug_options: null argument");
    *out = ctx->debug;
    return PROSPER_PT_OK;
}

int prosper_pt_create(const prosper_pt_device_desc *desc, prosper_pt_ctx **out_ctx)
{
    if (!desc || !out_ctx || desc->struct_size != sizeof(prosper_pt_device_desc))
        return fail(PROSPER_PT_ERR_INVALID_ARGUMENT, "prosper_pt_create: bad descriptor");
    *out_ctx = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0)
        return fail(PROSPER_PT_ERR_NO_DEVICE, "no HIP device is visible (this pass has no CPU fallback)");
    if (desc->device_ordinal < 0 || desc->device_ordinal >= count)
        return fail(PROSPER_PT_ERR_NO_DEVICE, "device ordinal out of range");
    PPT_HIP(hipSetDevice(desc->device_ordinal));
    hipDeviceProp_t prop;
    PPT_HIP(hipGetDeviceProperties(&prop, desc->device_ordinal));
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(PROSPER_PT_ERR_NO_DEVICE, std::string("device is ") + prop.gcnArchName + ", this library carries gfx950 code only");
#ifndef PPT_EXPERIMENTS
    if (desc->flags & PROSPER_PT_CREATE_PERSISTENT)
        return fail(PROSPER_PT_ERR_UNSUPPORTED, "the persistent pipeline is an experiment: build the library with -DPPT_EXPERIMENTS");
#endif
    prosper_pt_ctx *ctx = new (std::nothrow) prosper_pt_ctx();
    if (!ctx) return fail(PROSPER_PT_ERR_INVALID_ARGUMENT, "out of host memory");
    ctx->device = desc->device_ordinal;
    ctx->flags = desc->flags;
    prosper_pt_debug_options_default(&ctx->debug);
    {
        // the ONE place the library looks at the environment, and only when asked to (prosper_pt.h, debug options)
        std::string err;
        if (!debug_options_from_environment(&ctx->debug, &err))
        {
            delete ctx;
            return fail(PROSPER_PT_ERR_INVALID_ARGUMENT, err);
        }
    }
    bool eventsOk = true;
    for (auto &e : ctx->events) eventsOk = eventsOk && hipEventCreate(&e) == hipSuccess;
    for (auto &ws : ctx->workStreams) eventsOk = eventsOk && hipStreamCreateWithFlags(&ws, hipStreamNonBlocking) == hipSuccess;
    for (RenderSlot &slot : ctx->slots)
    {
        for (uint32_t i = 0; i < kMaxChains; ++i)
        {
            eventsOk = eventsOk && hipEventCreateWithFlags(&slot.chainJoin[i], hipEventDisableTiming) == hipSuccess;
            for (auto &e : slot.chainEvents[i]) eventsOk = eventsOk && hipEventCreate(&e) == hipSuccess;
        }
        eventsOk = eventsOk && hipEventCreateWithFlags(&slot.free, hipEventDisableTiming) == hipSuccess;
    }
    eventsOk = eventsOk && hipEventCreateWithFlags(&ctx->chainFork, hipEventDisableTiming) == hipSuccess;
    // Every work stream is used once here, so that the device's four hardware queues go to the caller's stream and these
    // three, in this order: a stream that comes into use later (the two of prosper_pt_update_meshes, made at first need)
    // then SHARES a queue.  Streams that claim queues before the work streams do leave two of those sharing one for the
    // life of the context: 4 % on FlightHelmet, 2 % on S-sponza-class, 60 % on a 256 x 256 frame (profiles/r04_mesh_streams.txt).
    for (auto &ws : ctx->workStreams)
        eventsOk = eventsOk && ws && hipEventRecord(ctx->chainFork, ws) == hipSuccess && hipStreamSynchronize(ws) == hipSuccess;
    if (!eventsOk || hipMalloc((void **)&ctx->dCounters, kStageCount * kCounterCount * sizeof(unsigned long long)) != hipSuccess ||
        hipMemset(ctx->dCounters, 0, kStageCount * kCounterCount * sizeof(unsigned long long)) != hipSuccess ||
        hipMalloc((void **)&ctx->dWorkCounter, 64) != hipSuccess)
    {
        prosper_pt_destroy(ctx);
        return fail(PROSPER_PT_ERR_HIP, "context allocation failed");
    }
    *out_ctx = ctx;
    return PROSPER_PT_OK;
}

void prosper_pt_destroy(prosper_pt_ctx *ctx)
{
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    discard_mesh_build(ctx); // (a worker may still be launching)
    (void)hipDeviceSynchronize();
    destroy_tiling(ctx);
    free_scene(ctx);
    if (ctx->ownedHdr) (void)hipFree(ctx->ownedHdr);
    if (ctx->dCounters) (void)hipFree(ctx->dCounters);
    if (ctx->dWorkCounter) (void)hipFree(ctx->dWorkCounter);
    for (RenderSlot &slot : ctx->slots)
    {
        if (slot.wfBlock) (void)hipFree(slot.wfBlock);
        if (slot.stackOverflow) (void)hipFree(slot.stackOverflow);
        if (slot.tileOrder) (void)hipFree(slot.tileOrder);
    }
    if (ctx->restirScratch) (void)hipFree(ctx->restirScratch);
    if (ctx->toneLut) (void)hipFree(ctx->toneLut);
    if (ctx->toneScratch) (void)hipFree(ctx->toneScratch);
    for (auto &e : ctx->events)
        if (e) (void)hipEventDestroy(e);
    for (RenderSlot &slot : ctx->slots)
    {
        for (uint32_t i = 0; i < kMaxChains; ++i)
        {
            for (auto &e : slot.chainEvents[i])
                if (e) (void)hipEventDestroy(e);
            if (slot.chainJoin[i]) (void)hipEventDestroy(slot.chainJoin[i]);
        }
        if (slot.free) (void)hipEventDestroy(slot.free);
    }
    for (auto &ws : ctx->workStreams)
        if (ws) (void)hipStreamDestroy(ws);
    for (auto &ws : ctx->extraStreams)
        if (ws) (void)hipStreamDestroy(ws);
    if (ctx->buildStream) (void)hipStreamDestroy(ctx->buildStream);
    if (ctx->pinnedStaging) (void)hipHostFree(ctx->pinnedStaging);
    if (ctx->chainFork) (void)hipEventDestroy(ctx->chainFork);
    delete ctx;
}

int prosper_pt_upload_scene(prosper_pt_ctx *ctx, const prosper_pt_scene_view *scene)
{
    if (!ctx || !scene) return fail(PROSPER_PT_ERR_INVALID_ARGUMENT, "prosper_pt_upload_scene: null argument");
    int rc = validate_scene(scene);
    if (rc != PROSPER_PT_OK) return rc;
    PPT_HIP(hipSetDevice(ctx->device));
    discard_mesh_build(ctx); // (a worker may still be launching)
    PPT_HIP(hipDeviceSynchronize());
    free_scene(ctx);
    rc = upload_scene_impl(ctx, scene);
    if (rc != PROSPER_PT_OK)
    {
        free_scene(ctx);
        return rc;
    }
    ctx->haveScene = true;
    return PROSPER_PT_OK;
}

// the staged light set into the next device version, on the stream of the render that is about to read it
static int flush_pending_lights(prosper_pt_ctx *ctx, hipStream_t stream)
{
    LightState *ls = ctx->lights;
    if (!ls || !ls->pending) return PROSPER_PT_OK;
    const uint32_t v = (ls->cur + 1u) % LightState::kVersions;
    if (!ls->dBlocks[v])
    {
        void *d = nullptr;
        const int rc = device_alloc(ctx, sizeof(LightBlock), &d);
        if (rc != PROSPER_PT_OK) return rc;
        ls->dBlocks[v] = static_cast<LightBlock *>(d);
    }
    if (!ls->versionFree[v]) PPT_HIP(hipEventCreateWithFlags(&ls->versionFree[v], hipEventDisableTiming));
    if (!ls->ready) PPT_HIP(hipEventCreateWithFlags(&ls->ready, hipEventDisableTiming));
    if (ls->versionUsed[v]) PPT_HIP(hipStreamWaitEvent(stream, ls->versionFree[v], 0));
    const uint32_t k = ls->pendingStaging;
    PPT_HIP(hipMemcpyAsync(ls->dBlocks[v], ls->staging[k], sizeof(LightBlock), hipMemcpyHostToDevice, stream));
    PPT_HIP(hipEventRecord(ls->stagingDone[k], stream));
    ls->stagingUsed[k] = true;
    PPT_HIP(hipEventRecord(ls->ready, stream));
    ls->readyRecorded = true;
    ls->cur = v;
    ctx->scene.directionalLight = &ls->dBlocks[v]->directional;
    ctx->scene.pointLights = &ls->dBlocks[v]->points;
    ctx->scene.spotLights = &ls->dBlocks[v]->spots;
    ctx->scene.pointLightCount = ls->staging[k]->points.count;
    ctx->scene.spotLightCount = ls->staging[k]->spots.count;
    ls->pending = false;
    ls->updates++;
    return PROSPER_PT_OK;
}

int prosper_pt_update_lights(
    prosper_pt_ctx *ctx, const prosper_DirectionalLightParameters *directionalLight,
    const prosper_PointLightsBuffer *pointLights, const prosper_SpotLightsBuffer *spotLights)
{
    if (!ctx || !directionalLight || !pointLights || !spotLights)
        return fail(PROSPER_PT_ERR_INVALID_ARGUMENT, "prosper_pt_update_lights: null argument");
    if (!ctx->haveScene || !ctx->lights) return fail(PROSPER_PT_ERR_NO_SCENE, "no scene uploaded");
    if (pointLights->count > PROSPER_MAX_POINT_LIGHT_COUNT || spotLights->count > PROSPER_MAX_SPOT_LIGHT_COUNT)
        return fail(PROSPER_PT_ERR_SCENE, "light count exceeds 1024");
    LightState *ls = ctx->lights;
    // prosper rewrites the buffers every frame (World.cpp:531-535) and mostly with what they held: that costs a memcmp
    if (std::memcmp(&ls->mirror->directional, directionalLight, sizeof(*directionalLight)) == 0 &&
        std::memcmp(&ls->mirror->points, pointLights, sizeof(*pointLights)) == 0 &&
        std::memcmp(&ls->mirror->spots, spotLights, sizeof(*spotLights)) == 0)
        return PROSPER_PT_OK;
    PPT_HIP(hipSetDevice(ctx->device));
    const uint32_t k = ls->pending ? ls->pendingStaging : ls->stagingNext;
    if (!ls->pending) ls->stagingNext = (ls->stagingNext + 1u) % kStagingBuffers;
    if (!ls->staging[k])
    {
        PPT_HIP(hipHostMalloc((void **)&ls->staging[k], sizeof(LightBlock), hipHostMallocDefault));
        PPT_HIP(hipEventCreateWithFlags(&ls->stagingDone[k], hipEventDisableTiming));
    }
    if (ls->stagingUsed[k]) PPT_HIP(hipEventSynchronize(ls->stagingDone[k])); // the copy of four updates ago
    ls->stagingUsed[k] = false;
    ls->staging[k]->directional = *directionalLight;
    ls->staging[k]->points = *pointLights;
    ls->staging[k]->spots = *spotLights;
    *ls->mirror = *ls->staging[k];
    ls->pending = true;
    ls->pendingStaging = k;
    return PROSPER_PT_OK;
}

// The synchronous path: re-split the instances that moved since the last build, re-assemble, upload (what
// prosper_pt_update_transforms did before the refit existed; prosper's own TLAS build is of this kind, on the GPU).
static int flush_pending_update(prosper_pt_ctx *ctx, hipStream_t stream);
static int rebuild_hierarchy_impl(prosper_pt_ctx *ctx)
{
    PPT_HIP(hipSetDevice(ctx->device));
    {
        // a background build of streamed-in meshes holds the subtrees: its result is installed first
        const int prc = poll_mesh_build(ctx, true);
        if (prc != PROSPER_PT_OK) return prc;
    }
    AccelState *acc = ctx->accel;
    const auto t0 = std::chrono::steady_clock::now();
    {
        const int frc = flush_pending_update(ctx, nullptr); // the moved triangles must be in the flat array
        if (frc != PROSPER_PT_OK) return frc;
    }
    PPT_HIP(hipDeviceSynchronize()); // renders in flight read the old hierarchy
    acc->stale = true;               // until the new hierarchy is up
    bool any = false;
    for (size_t r = 0; r < acc->ranges.size(); ++r)
    {
        if (!(acc->movedSinceBuild[r] || !acc->instanced) || !acc->ranges[r].count) continue;
        any = any || acc->movedSinceBuild[r];
        PPT_HIP(hipMemcpy(
            acc->flat.data() + acc->ranges[r].first, acc->dFlat + acc->ranges[r].first,
            sizeof(WorldTriangle) * (size_t)acc->ranges[r].count, hipMemcpyDeviceToHost));
    }
    BvhBuildResult bvh;
    const auto tBuild = std::chrono::steady_clock::now();
    try
    {
        if (ctx->debug.failNextUpdate)
        {
            ctx->debug.failNextUpdate = 0;
            throw std::runtime_error("debug option failNextUpdate is set");
        }
        bvh = acc->instanced ? acc->bvh.rebuild(acc->flat.data(), acc->movedSinceBuild, build_options(ctx))
                             : build_bvh(acc->flat.data(), acc->total, build_options(ctx));
    }
    catch (const std::exception &ex)
    {
        return fail(PROSPER_PT_ERR_UNSUPPORTED, std::string("BVH rebuild failed: ") + ex.what());
    }
    const double buildSeconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - tBuild).count();
    GeometryTarget target = context_target(ctx);
    const int rc = upload_hierarchy(ctx, target, bvh);
    if (rc != PROSPER_PT_OK) return rc;
    acc->stale = false;
    acc->rebuilds++;
    ctx->sceneStamp++;
    ctx->stats.nodeCount = bvh.nodes.size();
    ctx->stats.maxDepth = bvh.maxDepth;
    ctx->stats.deviceBytes = ctx->sceneBytes;
    ctx->stats.bvhBuildSeconds = buildSeconds;
    ctx->stats.buildSeconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    return PROSPER_PT_OK;
}

static float rebuild_cost_ratio(const prosper_pt_ctx *ctx)
{
    return ctx->debug.rebuildCostRatio > 0.0f ? std::max(1.0f, ctx->debug.rebuildCostRatio) : 1.3f;
}

// World::updateScene + the per-frame TLAS rebuild (World.cpp:359-466,749-802,878-928) as a REFIT: the new transforms, the
// world triangles and new boxes for the unchanged tree - no host build, no device-wide synchronisation.  Hits do not
// depend on the hierarchy (hit contract), so the image is the one a fresh upload gives.  Two steps:
//   stage_transforms      the call itself: which instances moved, the table into pinned staging; nothing on the GPU
//   flush_pending_update  run by the next consumer of the scene on ITS stream - a pipelined render's own chain - into the
//                         NEXT scene version: the frames in flight go on reading theirs, nothing waits for them
// What a refit cannot do is keep the tree GOOD when instances travel far: every refit leaves the tree's surface-area
// measure behind, and once that has grown by 30 % over its value at the last build the next update rebuilds
// (synchronously: prosper_pt_rebuild_hierarchy).
static int rebuild_hierarchy_impl(prosper_pt_ctx *ctx);

static int flush_pending_update(prosper_pt_ctx *ctx, hipStream_t stream)
{
    AccelState *acc = ctx->accel;
    if (!acc || !acc->pending) return PROSPER_PT_OK;
    const uint32_t v = (acc->cur + 1u) % AccelState::kVersions;
    const size_t transformBytes = sizeof(prosper_ModelInstanceTransforms) * (acc->pendingCount ? acc->pendingCount : 1);
    const size_t triBytes = sizeof(WorldTriangle) * (size_t)(acc->total ? acc->total : 1);
    void *d = nullptr;
    int rc;
    if (!acc->dTransformsV[v])
    {
        if ((rc = device_alloc(ctx, transformBytes, &d))) return rc;
        acc->dTransformsV[v] = static_cast<prosper_ModelInstanceTransforms *>(d);
    }
    if (!acc->dTrisV[v])
    {
        if ((rc = device_alloc(ctx, triBytes, &d))) return rc;
        acc->dTrisV[v] = static_cast<WorldTriangle *>(d);
    }
    if (!acc->dNodesV[v])
    {
        if ((rc = device_alloc(ctx, acc->nodeCapacityBytes, &d))) return rc;
        acc->dNodesV[v] = static_cast<BvhNode *>(d);
        acc->nodesCurrent[v] = false;
    }
    if (!acc->versionFree[v]) PPT_HIP(hipEventCreateWithFlags(&acc->versionFree[v], hipEventDisableTiming));
    // Everything is enqueued for version v through LOCAL names; the context switches to v only after the last call has
    // succeeded.  On a failure the previous version stays current and the update stays pending (the next consumer of the
    // scene tries again); what was half-written into v is rewritten by that retry.
    // the version's last readers (three updates ago), and the previous update: it wrote the node array copied below,
    // and it shares the flat triangle array and the bounds scratch with this one
    if (acc->versionUsed[v]) PPT_HIP(hipStreamWaitEvent(stream, acc->versionFree[v], 0));
    if (acc->sceneEventRecorded) PPT_HIP(hipStreamWaitEvent(stream, acc->sceneEvent, 0));
    if (!acc->nodesCurrent[v])
    {
        PPT_HIP(hipMemcpyAsync(acc->dNodesV[v], acc->dNodes, (size_t)acc->nodeCount * sizeof(BvhNode), hipMemcpyDeviceToDevice, stream));
        acc->nodesCurrent[v] = true;
    }
    const uint32_t k = acc->pendingStaging;
    PPT_HIP(hipMemcpyAsync(acc->dTransformsV[v], acc->staging[k], sizeof(prosper_ModelInstanceTransforms) * acc->pendingCount, hipMemcpyHostToDevice, stream));
    PPT_HIP(hipEventRecord(acc->stagingDone[k], stream));
    acc->stagingUsed[k] = true;
    // world-space triangles again, in both orders (the shading and any-hit records hold object-space attributes and stay
    // as they are), then the boxes.  (Also when only transforms of instances without geometry changed: a version must be
    // whole.)
    DeviceScene next = ctx->scene;
    next.nodes = acc->dNodesV[v];
    next.triangles = acc->dTrisV[v];
    next.modelInstanceTransforms = acc->dTransformsV[v];
    launch_flatten_triangles(
        next, acc->dOffsets, acc->drawInstanceCount, acc->dFlags, acc->dFlat, nullptr, nullptr, (uint32_t)acc->total, stream,
        acc->dLeafPosition, acc->dTrisV[v]);
    PPT_HIP(hipGetLastError());
    acc->flatStale = true; // (the flat array now holds the new pose whatever happens next)
    if (acc->total && (rc = enqueue_refit(acc, bvh_pad_coefficient(build_options(ctx)), acc->dNodesV[v], acc->dTrisV[v], v, stream))) return rc;
    PPT_HIP(hipEventRecord(acc->sceneEvent, stream));
    // ---- commit: the new version becomes the scene ----
    acc->sceneEventRecorded = true;
    acc->cur = v;
    acc->dNodes = acc->dNodesV[v];
    acc->dTris = acc->dTrisV[v];
    ctx->dTransforms = acc->dTransformsV[v];
    ctx->scene = next;
    acc->refits++;
    ctx->sceneStamp++;
    acc->pending = false;
    acc->stale = false;
    return PROSPER_PT_OK;
}

} // extern "C"

int ppt::stage_transforms(prosper_pt_ctx *ctx, const prosper_ModelInstanceTransforms *transforms, uint32_t count)
{
    AccelState *acc = ctx->accel;
    const auto t0 = std::chrono::steady_clock::now();
    // which instances moved (World::updateScene rewrites every transform each frame, World.cpp:359-466; most are unchanged)
    bool any = acc->stale;
    for (size_t r = 0; r < acc->ranges.size(); ++r)
    {
        const uint32_t mi = acc->rangeModelInstance[r];
        if (acc->stale || std::memcmp(&transforms[mi], &acc->transforms[mi], sizeof(prosper_ModelInstanceTransforms)) != 0)
        {
            acc->movedSinceBuild[r] = 1;
            any = true;
        }
    }
    if (!any && std::memcmp(transforms, acc->transforms.data(), sizeof(prosper_ModelInstanceTransforms) * count) == 0) return PROSPER_PT_OK;
    PPT_HIP(hipSetDevice(ctx->device));
    // the measure of the newest refit that has finished (the newest may still be queued behind frames in flight)
    {
        const int prc = poll_refit_cost(acc, false);
        if (prc != PROSPER_PT_OK) return prc;
    }
    // into pinned staging (a pageable source would make the later copy synchronous).  An update that was never consumed
    // is simply replaced: its staging buffer is reused.
    const uint32_t k = acc->pending ? acc->pendingStaging : acc->stagingNext;
    if (!acc->pending) acc->stagingNext = (acc->stagingNext + 1u) % kStagingBuffers;
    if (!acc->staging[k])
    {
        PPT_HIP(hipHostMalloc((void **)&acc->staging[k], sizeof(prosper_ModelInstanceTransforms) * (count ? count : 1), hipHostMallocDefault));
        PPT_HIP(hipEventCreateWithFlags(&acc->stagingDone[k], hipEventDisableTiming));
    }
    if (acc->stagingUsed[k]) PPT_HIP(hipEventSynchronize(acc->stagingDone[k])); // the copy of four updates ago
    acc->stagingUsed[k] = false;
    std::memcpy(acc->staging[k], transforms, sizeof(prosper_ModelInstanceTransforms) * count);
    acc->pending = true;
    acc->pendingStaging = k;
    acc->pendingCount = count;
    acc->transforms.assign(transforms, transforms + count);
    ctx->stats.bvhBuildSeconds = 0.0;
    ctx->stats.buildSeconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (any && ctx->debug.alwaysRebuild) return rebuild_hierarchy_impl(ctx); // (debug option: the synchronous path, every time)
    if (any && acc->lastCostRatio > rebuild_cost_ratio(ctx) && !ctx->meshBuild)
    {
        // The refits have degraded the tree: the instances that moved since the last build are split again - by the worker
        // thread that also takes up streamed-in meshes, into a new geometry generation, while the frame loop goes on
        // refitting and rendering this one (until round 4 this was a 60-75 ms synchronous rebuild in the middle of the
        // frame loop).  The generation is switched in by the first render after it is done.
        return start_mesh_build(ctx, true);
    }
    return PROSPER_PT_OK;
}

extern "C" {

static int update_transforms_impl(prosper_pt_ctx *ctx, const prosper_ModelInstanceTransforms *transforms, uint32_t count, hipStream_t stream, bool flushNow)
{
    const int rc = stage_transforms(ctx, transforms, count);
    if (rc != PROSPER_PT_OK || !flushNow) return rc;
    return flush_pending_update(ctx, stream);
}

static int check_update_arguments(prosper_pt_ctx *ctx, const prosper_ModelInstanceTransforms *transforms, uint32_t count)
{
    if (!ctx || !transforms) return fail(PROSPER_PT_ERR_INVALID_ARGUMENT, "prosper_pt_update_transforms: null argument");
    if (!ctx->haveScene || !ctx->accel) return fail(PROSPER_PT_ERR_NO_SCENE, "no scene uploaded");
    if (count != ctx->accel->transforms.size())
        return fail(PROSPER_PT_ERR_INVALID_ARGUMENT, "prosper_pt_update_transforms: count differs from the scene's modelInstanceCount");
    return PROSPER_PT_OK;
}

int prosper_pt_update_transforms(prosper_pt_ctx *ctx, const prosper_ModelInstanceTransforms *transforms, uint32_t count)
{
    const int rc = check_update_arguments(ctx, transforms, count);
    return rc != PROSPER_PT_OK ? rc : update_transforms_impl(ctx, transforms, count, nullptr, false);
}

int prosper_pt_update_transforms_async(
    prosper_pt_ctx *ctx, const prosper_ModelInstanceTransforms *transforms, uint32_t count, uint32_t flags, void *stream)
{
    if (flags & ~(uint32_t)PROSPER_PT_UPDATE_NOW) return fail(PROSPER_PT_ERR_INVALID_ARGUMENT, "prosper_pt_update_transforms_async: unknown flag");
    const int rc = check_update_arguments(ctx, transforms, count);
    return rc != PROSPER_PT_OK ? rc : update_transforms_impl(ctx, transforms, count, static_cast<hipStream_t>(stream), (flags & PROSPER_PT_UPDATE_NOW) != 0);
}

int prosper_pt_rebuild_hierarchy(prosper_pt_ctx *ctx)
{
    if (!ctx) return fail(PROSPER_PT_ERR_INVALID_ARGUMENT, "prosper_pt_rebuild_hierarchy: null argument");
    if (!ctx->haveScene || !ctx->accel) return fail(PROSPER_PT_ERR_NO_SCENE, "no scene uploaded");
    return rebuild_hierarchy_impl(ctx);
}

int prosper_pt_get_hierarchy_state(prosper_pt_ctx *ctx, prosper_pt_hierarchy_state *out)
{
    if (!ctx || !out) return fail(PROSPER_PT_ERR_INVALID_ARGUMENT, "prosper_pt_get_hierarchy_state: null argument");
    *out = prosper_pt_hierarchy_state{};
    if (ctx->geometry)
    {
        out->meshUpdates = ctx->geometry->meshUpdates;
        out->geometryInstalls = ctx->geometry->installs;
        out->geometryBuildRunning = (ctx->meshBuild || ctx->geometry->dirty) ? 1u : 0u;
    }
    if (!ctx->haveScene || !ctx->accel) return fail(PROSPER_PT_ERR_NO_SCENE, "no scene uploaded");
    AccelState *acc = ctx->accel;
    PPT_HIP(hipSetDevice(ctx->device));
    {
        const int frc = flush_pending_update(ctx, nullptr);
        if (frc != PROSPER_PT_OK) return frc;
    }
    {
        const int prc = poll_refit_cost(acc, true);
        if (prc != PROSPER_PT_OK) return prc;
    }
    out->refits = acc->refits;
    out->rebuilds = acc->rebuilds;
    out->costRatio = acc->lastCostRatio;
    out->builtCost = acc->builtCost;
    out->nodeCount = acc->nodeCount;
    out->levels = (uint32_t)acc->levelOffsets.size() - 1u;
    return PROSPER_PT_OK;
}

int prosper_pt_debug_read_nodes(prosper_pt_ctx *ctx, void *out, size_t byte_size)
{
    if (!ctx || !out) return fail(PROSPER_PT_ERR_INVALID_ARGUMENT, "prosper_pt_debug_read_nodes: null argument");
    if (!ctx->haveScene || !ctx->accel) return fail(PROSPER_PT_ERR_NO_SCENE, "no scene uploaded");
    const size_t bytes = (size_t)ctx->accel->nodeCount * sizeof(BvhNode);
    if (byte_size < bytes) return fail(PROSPER_PT_ERR_INVALID_ARGUMENT, "prosper_pt_debug_read_nodes: destination too small");
    PPT_HIP(hipSetDevice(ctx->device));
    {
        const int frc = flush_pending_update(ctx, nullptr);
        if (frc != PROSPER_PT_OK) return frc;
    }
    PPT_HIP(hipDeviceSynchronize());
    PPT_HIP(hipMemcpy(out, ctx->accel->dNodes, bytes, hipMemcpyDeviceToHost));
    return PROSPER_PT_OK;
}

int prosper_pt_get_scene_stats(prosper_pt_ctx *ctx, prosper_pt_scene_stats *out)
{
    if (!ctx || !out) return fail(PROSPER_PT_ERR_INVALID_ARGUMENT, "prosper_pt_get_scene_stats: null argument");
    if (!ctx->haveScene) return fail(PROSPER_PT_ERR_NO_SCENE, "no scene uploaded");
    *out = ctx->stats;
    // the variants depend on the context's debug options at launch time: report what a render started now would take
    const WavefrontPlan plan = wavefront_plan(
        ctx->stats.maxDepth, (uint32_t)ctx->stats.nodeCount, (uint32_t)ctx->stats.triangleCount, ctx->scene, wavefront_options(ctx));
    const uint32_t ldsEntries = plan.ldsStackEntries;
    out->variantFlags =
        (plan.sceneInLds ? PROSPER_PT_VARIANT_LDS_SCENE : 0u) | (plan.tablesInLds ? PROSPER_PT_VARIANT_LDS_TABLES : 0u) |
        (ctx->scene.batchedTextures ? PROSPER_PT_VARIANT_BATCHED_TEXTURES : 0u) |
        (ctx->packedMaterials ? PROSPER_PT_VARIANT_TEXTURE_PACKS : 0u) | (ctx->rawRecords ? PROSPER_PT_VARIANT_RAW_RECORDS : 0u) |
        (ldsEntries << PROSPER_PT_VARIANT_STACK_SHIFT);
    return PROSPER_PT_OK;
}

int prosper_pt_set_output_buffer(prosper_pt_ctx *ctx, void *device_rgba32f, size_t byte_size)
{
    if (!ctx) return fail(PROSPER_PT_ERR_INVALID_ARGUMENT, "prosper_pt_set_output_buffer: null context");
    if (device_rgba32f && (reinterpret_cast<uintptr_t>(device_rgba32f) & 15u))
        return fail(PROSPER_PT_ERR_INVALID_ARGUMENT, "output buffer must be 16-byte aligned");
    ctx->externalHdr = device_rgba32f;
    ctx->externalHdrBytes = device_rgba32f ? byte_size : 0;
    return PROSPER_PT_OK;
}

// The scene and light versions a render (or a ReSTIR trace) read are free again behind its last kernel on `s` (the
// accumulate kernel follows the path stages on the caller's stream).  One event per version: a reader on ANOTHER stream
// than the previous reader's first waits for that one, so the newest record always stands for every reader so far.
static int mark_versions_read(prosper_pt_ctx *ctx, hipStream_t s)
{
    auto mark = [&](hipEvent_t &event, bool &used, hipStream_t &last) -> int {
        if (!event) PPT_HIP(hipEventCreateWithFlags(&event, hipEventDisableTiming));
        if (used && last != s) PPT_HIP(hipStreamWaitEvent(s, event, 0));
        PPT_HIP(hipEventRecord(event, s));
        used = true;
        last = s;
        return PROSPER_PT_OK;
    };
    if (AccelState *acc = ctx->accel)
    {
        const int rc = mark(acc->versionFree[acc->cur], acc->versionUsed[acc->cur], acc->versionStream[acc->cur]);
        if (rc != PROSPER_PT_OK) return rc;
    }
    if (LightState *ls = ctx->lights)
    {
        const int rc = mark(ls->versionFree[ls->cur], ls->versionUsed[ls->cur], ls->versionStream[ls->cur]);
        if (rc != PROSPER_PT_OK) return rc;
    }
    if (MaterialState *ms = ctx->materialState)
    {
        const int rc = mark(ms->versionFree[ms->cur], ms->versionUsed[ms->cur], ms->versionStream[ms->cur]);
        if (rc != PROSPER_PT_OK) return rc;
    }
    return PROSPER_PT_OK;
}

int prosper_pt_render_frames(
    prosper_pt_ctx *ctx, const prosper_ReferencePC *pc, const prosper_CameraUniforms *camera, uint32_t width,
    uint32_t height, const prosper_pt_tile_desc *tile, uint32_t frame_count, uint32_t render_flags, void *stream)
{
    if (!ctx || !pc || !camera) return fail(PROSPER_PT_ERR_INVALID_ARGUMENT, "prosper_pt_render: null argument");
    if (!ctx->haveScene) return fail(PROSPER_PT_ERR_NO_SCENE, "prosper_pt_render called before prosper_pt_upload_scene");
    if (ctx->meshBuild)
    {
        // streamed-in meshes whose geometry a worker has finished meanwhile: from this render on they are the scene
        const int prc = poll_mesh_build(ctx, false);
        if (prc != PROSPER_PT_OK) return prc;
    }
    if (ctx->accel && ctx->accel->stale && !ctx->accel->pending) // (a staged update gets its chance below)
        return fail(PROSPER_PT_ERR_NO_SCENE, "the last prosper_pt_update_transforms failed: update the transforms again (or upload the scene) before rendering");
    if (width == 0 || height == 0 || frame_count == 0)
        return fail(PROSPER_PT_ERR_INVALID_ARGUMENT, "prosper_pt_render: empty extent or frame count");
    if (pc->drawType >= PROSPER_DRAW_TYPE_COUNT) return fail(PROSPER_PT_ERR_INVALID_ARGUMENT, "drawType out of range");
    const bool tiled = tile && tile->stripeCount > 1 && tile->stripeWidth > 0;
    if (tiled && tile->stripeIndex >= tile->stripeCount)
        return fail(PROSPER_PT_ERR_INVALID_ARGUMENT, "tile stripeIndex >= stripeCount");
    PPT_HIP(hipSetDevice(ctx->device));
    hipStream_t s = static_cast<hipStream_t>(stream);

    const uint32_t localWidth = compute_local_width(width, tile);
    const size_t bytes = (size_t)localWidth * height * sizeof(float4);
    if (ctx->externalHdr)
    {
        if (ctx->externalHdrBytes < bytes) return fail(PROSPER_PT_ERR_INVALID_ARGUMENT, "caller-owned output buffer is too small");
        ctx->hdr = static_cast<float4 *>(ctx->externalHdr);
    }
    else
    {
        if (ctx->ownedHdrBytes < bytes || !ctx->ownedHdr)
        {
            PPT_HIP(hipStreamSynchronize(s));
            if (ctx->ownedHdr) PPT_HIP(hipFree(ctx->ownedHdr));
            ctx->ownedHdr = nullptr;
            PPT_HIP(hipMalloc((void **)&ctx->ownedHdr, bytes ? bytes : 16));
            PPT_HIP(hipMemset(ctx->ownedHdr, 0, bytes ? bytes : 16));
            ctx->ownedHdrBytes = bytes;
        }
        ctx->hdr = ctx->ownedHdr;
    }
    ctx->localWidth = localWidth;
    ctx->height = height;
    ctx->lastWidth = width;
    ctx->stripeWidth = tiled ? tile->stripeWidth : 0;
    ctx->stripeIndex = tiled ? tile->stripeIndex : 0;
    ctx->stripeCount = tiled ? tile->stripeCount : 1;
    // a gather of this tile may still be in flight on the communicator's stream (prosper_pt_gather_tiles): the
    // accumulate kernel, which writes the tile, is enqueued on `s` and must come after it; detached path stages
    // (PROSPER_PT_RENDER_PIPELINED) do not wait for `s` and overlap the gather
    wait_for_gather_before_writing_tile(ctx, s);

    RenderParams p = {};
    p.pc = *pc;
    // rt/ray.glsl:21-35 and scene/camera.glsl:46-51 read these parts of CameraUniforms
    p.eye[0] = camera->eye.x;
    p.eye[1] = camera->eye.y;
    p.eye[2] = camera->eye.z;
    const prosper_mat4 &w2c = camera->worldToCamera;
    p.right[0] = w2c.col[0].x; p.right[1] = w2c.col[1].x; p.right[2] = w2c.col[2].x;
    p.up[0] = w2c.col[0].y; p.up[1] = w2c.col[1].y; p.up[2] = w2c.col[2].y;
    p.fwd[0] = -w2c.col[0].z; p.fwd[1] = -w2c.col[1].z; p.fwd[2] = -w2c.col[2].z;
    {
        // volatile: keep the compiler from folding the two divisions into anything but IEEE fp32 divides
        volatile float c00 = camera->cameraToClip.col[0].x, c11 = camera->cameraToClip.col[1].y;
        p.aspect = c11 / c00;
        p.tanHalfFovY = 1.0f / c11;
    }
    std::memcpy(p.cameraToWorld, &camera->cameraToWorld, 64);
    p.width = width;
    p.height = height;
    p.stripeWidth = tiled ? tile->stripeWidth : 0;
    p.stripeIndex = tiled ? tile->stripeIndex : 0;
    p.stripeCount = tiled ? tile->stripeCount : 1;
    p.localWidth = localWidth;
    p.frameCount = frame_count;
    p.traceDeadPaths = ctx->debug.traceDeadPaths ? 1u : 0u;
    // sparse segments (pt_wavefront.hip RayMap), an experiment that lost (profiles/r03_sparse_segments.txt): debug option
    // mergeLimit = rays up to which a workgroup's four segments are traced by one wave; default never
    p.mergeLimit = ctx->debug.mergeLimit;

    if (localWidth == 0) return PROSPER_PT_OK;
    const bool countWork = (render_flags & PROSPER_PT_RENDER_COUNT_WORK) != 0;
    LaunchTimer timer;
    timer.events = ctx->events;
    timer.stage = ctx->eventStage;
    timer.capacity = prosper_pt_ctx::kMaxTimedLaunches;
    LaunchTimer *tp = ctx->kernelTiming ? &timer : nullptr;
    // a staged prosper_pt_update_transforms runs now, on the stream this render's path stages use: a pipelined render's own
    // chain (beside the frames in flight, which keep reading their scene version), else the caller's stream
    const bool wavefrontPipelined = !(ctx->flags & (PROSPER_PT_CREATE_MEGAKERNEL | PROSPER_PT_CREATE_PERSISTENT)) &&
                                    (render_flags & PROSPER_PT_RENDER_PIPELINED) != 0 && !countWork;
    {
        const uint32_t nextSlot = wavefrontPipelined ? (ctx->lastSlot + 1u) % prosper_pt_ctx::kRenderSlots : 0u;
        hipStream_t us = wavefrontPipelined ? ctx->workStreams[nextSlot] : s;
        int frc = flush_pending_update(ctx, us);
        if (frc == PROSPER_PT_OK) frc = flush_pending_lights(ctx, us);
        if (frc == PROSPER_PT_OK) frc = flush_pending_materials(ctx, us);
        if (frc != PROSPER_PT_OK) return frc;
        if (ctx->lights && ctx->lights->readyRecorded) PPT_HIP(hipStreamWaitEvent(s, ctx->lights->ready, 0));
        if (ctx->materialState && ctx->materialState->readyRecorded) PPT_HIP(hipStreamWaitEvent(s, ctx->materialState->ready, 0));
        // a refit enqueued on another stream (another render's chain, or prosper_pt_update_transforms_async) must be
        // done before anything on the caller's stream reads the scene
        if (ctx->accel && ctx->accel->sceneEventRecorded) PPT_HIP(hipStreamWaitEvent(s, ctx->accel->sceneEvent, 0));
    }
    if (ctx->flags & PROSPER_PT_CREATE_MEGAKERNEL)
    {
        int32_t *ovf = nullptr;
        const int orc = ensure_stack_overflow(ctx, ctx->slots[0], kTraversalStackDepth, megakernel_grid_blocks(p), s, &ovf);
        if (orc != PROSPER_PT_OK) return orc;
        wait_for_slot(ctx->slots[0], s);
        if (tp) tp->mark(kStageGenerate, s);
        launch_render_megakernel(ctx->scene, p, ctx->hdr, ctx->dCounters, ovf, countWork, s);
        release_slot(ctx->slots[0], s);
    }
#ifdef PPT_EXPERIMENTS
    else if (ctx->flags & PROSPER_PT_CREATE_PERSISTENT)
    {
        int32_t *ovf = nullptr;
        const int orc = ensure_stack_overflow(ctx, ctx->slots[0], kTraversalStackDepth, persistent_grid_blocks(), s, &ovf);
        if (orc != PROSPER_PT_OK) return orc;
        wait_for_slot(ctx->slots[0], s);
        if (tp) tp->mark(kStageGenerate, s);
        launch_render_persistent(ctx->scene, p, ctx->hdr, ctx->dCounters, ctx->dWorkCounter, ovf, countWork, s);
        release_slot(ctx->slots[0], s);
    }
#endif
    else
    {
        // wavefront: all frames of the batch are in flight together, in chunks that keep the
        // workspace under kMaxWavefrontSlots path slots
        const uint32_t tilesX = (localWidth + 7u) / 8u, tilesY = (height + 7u) / 8u;
        const uint64_t pixelsPadded = (uint64_t)tilesX * tilesY * 64u;
        constexpr uint64_t kMaxWavefrontSlots = 64ull << 20;
        if (pixelsPadded > kMaxWavefrontSlots) return fail(PROSPER_PT_ERR_UNSUPPORTED, "image too large for the wavefront workspace");
        uint32_t framesPerChunk = (uint32_t)(kMaxWavefrontSlots / pixelsPadded);
        if (framesPerChunk > frame_count) framesPerChunk = frame_count;
        // Frames in flight.  Default: the chains fork from the caller's stream, i.e. after everything enqueued on
        // it so far, and use slot 0.  PROSPER_PT_RENDER_PIPELINED: the path stages (generate / shade / trace) of
        // this render run as ONE chain on the other slot's stream and wait only for that slot's previous render,
        // so they overlap the previous render's remaining work - a 2 M-path batch alone fills 55 % of the GPU,
        // two of them 80 % (profiles/r01_pipelined.txt).  The accumulate kernel stays on the caller's stream, in
        // order: history reads, output writes and everything the caller enqueues later see finished frames.
        const bool pipelined = (render_flags & PROSPER_PT_RENDER_PIPELINED) != 0 && !countWork;
        const uint32_t slotIndex = pipelined ? (ctx->lastSlot + 1u) % prosper_pt_ctx::kRenderSlots : 0u;
        RenderSlot &slot = ctx->slots[slotIndex];
        ctx->lastSlot = slotIndex;
#ifdef PPT_EXPERIMENT_MESH_STREAMS_AFTER_FIRST_RENDER // (A/B: does the worker's stream cost anything once it shares a queue?)
        (void)ensure_build_stream(ctx);
#endif
        WavefrontChains chains;
        LaunchTimer chainTimers[kMaxChains];
        chains.count = (pipelined || (ctx->flags & PROSPER_PT_CREATE_SINGLE_CHAIN)) ? 1u : 2u;
        if (!pipelined && ctx->debug.chains >= 1u && ctx->debug.chains <= kMaxChains) chains.count = ctx->debug.chains; // tuning hook (in-order mode)
        // experiment (profiles/r03_hip_graph.txt, "two chains per frame in flight"): debug option pipelinedChains = 2 splits a
        // pipelined frame's segment groups over two chains, the second on a stream of its own
        bool twoDetached = false;
#ifdef PPT_EXPERIMENTS
        if (pipelined && ctx->debug.pipelinedChains == 2u)
        {
            if (!ctx->extraStreams[slotIndex]) PPT_HIP(hipStreamCreateWithFlags(&ctx->extraStreams[slotIndex], hipStreamNonBlocking));
            twoDetached = true;
            chains.count = 2u;
        }
#endif
        chains.detached = pipelined;
        chains.fork = ctx->chainFork;
        for (uint32_t i = 0; i < kMaxChains; ++i)
        {
            chains.streams[i] = pipelined ? ((twoDetached && i == 1u) ? ctx->extraStreams[slotIndex] : ctx->workStreams[slotIndex]) : ctx->workStreams[i];
            chains.join[i] = slot.chainJoin[i];
            chainTimers[i].events = slot.chainEvents[i];
            chainTimers[i].stage = slot.chainStage[i];
            chainTimers[i].capacity = prosper_pt_ctx::kMaxTimedLaunches;
            chains.timers[i] = tp ? &chainTimers[i] : nullptr;
            slot.chainLaunches[i] = 0;
        }
        for (uint32_t f0 = 0; f0 < frame_count; f0 += framesPerChunk)
        {
            const uint32_t frames = (frame_count - f0 < framesPerChunk) ? frame_count - f0 : framesPerChunk;
            WavefrontBuffers w = {};
            // Banded batches (debug option bandedBatches = 1; round 4, profiles/r04_banded_batches.txt): every XCD's segments take
            // the camera-ray batches of one band of the image, so that the paths an XCD traces START in one part of the scene.
            // Measured and NOT a default: on S-sponza-class wf_trace's FETCH_SIZE moves by 3 % (2.249 -> 2.184 GB per launch) -
            // three of its four launches trace bounce rays and sun shadows that cross the whole hall wherever they start -
            // while the bands' unequal costs bind the step to the slowest XCD: C3 13.0 -> 13.6 ms, C4 27.2 -> 30.7,
            // FlightHelmet (a band of sky is nearly free) 1.88 -> 2.48.
            const bool banded = ctx->debug.bandedBatches > 0;
            const int rc = ensure_wavefront_workspace(ctx, slot, tilesX, tilesY, frames, pipelined, banded, &w);
            if (rc != PROSPER_PT_OK) return rc;
            RenderParams pp = p;
            pp.frameCount = frames;
            pp.pc.frameIndex = (p.pc.frameIndex + f0) % PROSPER_RT_FRAME_PERIOD;
            if (f0 > 0) pp.pc.flags &= ~(uint32_t)PROSPER_PC_FLAG_SKIP_HISTORY;
            const WavefrontPlan plan = wavefront_plan(
                ctx->stats.maxDepth, (uint32_t)ctx->stats.nodeCount, (uint32_t)ctx->stats.triangleCount, ctx->scene, wavefront_options(ctx));
            int32_t *ovf = nullptr;
            const int orc = ensure_scratch_dwords(ctx, slot, plan.scratchDwordsPerBlock, wavefront_grid_blocks(w), s, &ovf);
            if (orc != PROSPER_PT_OK) return orc;
            w.tileOrder = nullptr;
#ifdef PPT_EXPERIMENTS
            // EXPERIMENT (debug option tileOrder; measured slower, profiles/r03_tile_order.txt): the camera-ray batches
            // take the tiles by cost, heaviest first, so that every segment's stride through the sequence gets the same mix;
            // recomputed when the view or the geometry changed since this slot's last order
            const bool tileOrderExperiment = ctx->debug.tileOrder != 0;
            if (frames >= 4u && tileOrderExperiment)
            {
                const size_t tiles = (size_t)tilesX * tilesY;
                RenderSlot::OrderKey key = {};
                std::memcpy(key.camera, pp.eye, sizeof(float) * 12);
                key.camera[12] = pp.aspect;
                key.camera[13] = pp.tanHalfFovY;
                key.width = pp.width;
                key.height = pp.height;
                key.stripeWidth = pp.stripeWidth;
                key.stripeIndex = pp.stripeIndex;
                key.stripeCount = pp.stripeCount;
                key.localWidth = pp.localWidth;
                key.sceneStamp = ctx->sceneStamp;
                hipStream_t os = pipelined ? ctx->workStreams[slotIndex] : s;
                if (slot.tileOrderTiles < tiles)
                {
                    PPT_HIP(hipDeviceSynchronize());
                    if (slot.tileOrder) PPT_HIP(hipFree(slot.tileOrder));
                    slot.tileOrder = nullptr;
                    slot.tileOrderTiles = 0;
                    PPT_HIP(hipMalloc((void **)&slot.tileOrder, (2 * tiles + 512) * sizeof(uint32_t)));
                    slot.tileOrderTiles = tiles;
                    slot.orderValid = false;
                }
                if (!slot.orderValid || std::memcmp(&key, &slot.orderKey, sizeof(key)) != 0)
                {
                    // behind the slot's previous user (it reads the old order) and behind a refit on another stream
                    if (pipelined && slot.freeRecorded) PPT_HIP(hipStreamWaitEvent(os, slot.free, 0));
                    if (!pipelined) wait_for_slot(slot, s);
                    if (pipelined && ctx->accel && ctx->accel->sceneEventRecorded) PPT_HIP(hipStreamWaitEvent(os, ctx->accel->sceneEvent, 0));
                    launch_tile_order(
                        ctx->scene, pp, tilesX, tilesY, plan.ldsStackEntries, ovf, slot.tileOrder, slot.tileOrder + slot.tileOrderTiles, os);
                    PPT_HIP(hipGetLastError());
                    slot.orderKey = key;
                    slot.orderValid = true;
                }
                w.tileOrder = slot.tileOrder;
            }
#endif
            // the slot's previous user (a render of two calls ago, or the previous chunk of this one) must be done
            // with the workspace: detached chains wait for that on their own stream, the others on the caller's
            chains.after = slot.freeRecorded ? slot.free : nullptr;
            chains.scene = (ctx->accel && ctx->accel->sceneEventRecorded) ? ctx->accel->sceneEvent : nullptr;
            chains.lights = (ctx->lights && ctx->lights->readyRecorded) ? ctx->lights->ready : nullptr;
            chains.materials = (ctx->materialState && ctx->materialState->readyRecorded) ? ctx->materialState->ready : nullptr;
            if (!pipelined) wait_for_slot(slot, s);
            if (tp) tp->mark(kStageChains, s);
            launch_render_wavefront(
                ctx->scene, pp, ctx->hdr, ctx->dCounters, w, plan, ovf, (uint32_t)ctx->stats.nodeCount,
                (uint32_t)ctx->stats.triangleCount, countWork, tp, chains, s);
            release_slot(slot, s);
        }
        for (uint32_t i = 0; i < kMaxChains; ++i) slot.chainLaunches[i] = chainTimers[i].count;
        if (tp) ctx->timedSlot = slotIndex;
    }
    PPT_HIP(hipGetLastError());
    {
        const int mrc = mark_versions_read(ctx, s);
        if (mrc != PROSPER_PT_OK) return mrc;
    }
    if (tp)
    {
        tp->close(s);
        ctx->timedLaunches = tp->count;
        ctx->timingValid = true;
    }
    return PROSPER_PT_OK;
}

int prosper_pt_render(
    prosper_pt_ctx *ctx, const prosper_ReferencePC *pc, const prosper_CameraUniforms *camera, uint32_t width,
    uint32_t height, const prosper_pt_tile_desc *tile, uint32_t render_flags, void *stream)
{
    return prosper_pt_render_frames(ctx, pc, camera, width, height, tile, 1, render_flags, stream);
}

int prosper_pt_get_local_extent(prosper_pt_ctx *ctx, uint32_t *local_width, uint32_t *height)
{
    if (!ctx || !local_width || !height) return fail(PROSPER_PT_ERR_INVALID_ARGUMENT, "prosper_pt_get_local_extent: null argument");
    *local_width = ctx->localWidth;
    *height = ctx->height;
    return PROSPER_PT_OK;
}

int prosper_pt_get_hdr_device_ptr(prosper_pt_ctx *ctx, void **out_ptr, size_t *out_bytes)
{
    if (!ctx || !out_ptr) return fail(PROSPER_PT_ERR_INVALID_ARGUMENT, "prosper_pt_get_hdr_device_ptr: null argument");
    if (!ctx->hdr) return fail(PROSPER_PT_ERR_NO_SCENE, "nothing has been rendered yet");
    *out_ptr = ctx->hdr;
    if (out_bytes) *out_bytes = (size_t)ctx->localWidth * ctx->height * sizeof(float4);
    return PROSPER_PT_OK;
}

int prosper_pt_read_hdr(prosper_pt_ctx *ctx, float *rgba32f, size_t byte_size, void *stream)
{
    if (!ctx || !rgba32f) return fail(PROSPER_PT_ERR_INVALID_ARGUMENT, "prosper_pt_read_hdr: null argument");
    if (!ctx->hdr) return fail(PROSPER_PT_ERR_NO_SCENE, "nothing has been rendered yet");
    const size_t bytes = (size_t)ctx->localWidth * ctx->height * sizeof(float4);
    if (byte_size < bytes) return fail(PROSPER_PT_ERR_INVALID_ARGUMENT, "prosper_pt_read_hdr: destination too small");
    PPT_HIP(hipSetDevice(ctx->device));
    hipStream_t s = static_cast<hipStream_t>(stream);
    PPT_HIP(hipMemcpyAsync(rgba32f, ctx->hdr, bytes, hipMemcpyDeviceToHost, s));
    PPT_HIP(hipStreamSynchronize(s));
    return PROSPER_PT_OK;
}

int prosper_pt_blit_rgba16f(prosper_pt_ctx *ctx, uint16_t *host_rgba16f, size_t byte_size, void *stream)
{
    if (!ctx || !host_rgba16f) return fail(PROSPER_PT_ERR_INVALID_ARGUMENT, "prosper_pt_blit_rgba16f: null argument");
    if (!ctx->hdr) return fail(PROSPER_PT_ERR_NO_SCENE, "nothing has been rendered yet");
    const uint32_t count = ctx->localWidth * ctx->height;
    if (byte_size < (size_t)count * 8u) return fail(PROSPER_PT_ERR_INVALID_ARGUMENT, "prosper_pt_blit_rgba16f: destination too small");
    PPT_HIP(hipSetDevice(ctx->device));
    hipStream_t s = static_cast<hipStream_t>(stream);
    void *tmp = nullptr;
    PPT_HIP(hipMalloc(&tmp, (size_t)count * 8u + 16));
    launch_blit_rgba16f(ctx->hdr, tmp, count, s);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(host_rgba16f, tmp, (size_t)count * 8u, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    (void)hipFree(tmp);
    if (e != hipSuccess) return fail(PROSPER_PT_ERR_HIP, std::string("blit: ") + hipGetErrorString(e));
    return PROSPER_PT_OK;
}

int prosper_pt_restir_di_trace(
    prosper_pt_ctx *ctx, const prosper_pt_restir_trace_pc *pc, const prosper_CameraUniforms *camera, uint32_t width,
    uint32_t height, const prosper_pt_restir_inputs *in, void *stream)
{
    if (!ctx || !pc || !camera || !in || !in->albedoRoughness || !in->normalMetallic || !in->nonLinearDepth || !in->reservoirs)
        return fail(PROSPER_PT_ERR_INVALID_ARGUMENT, "prosper_pt_restir_di_trace: null argument");
    if (!ctx->haveScene) return fail(PROSPER_PT_ERR_NO_SCENE, "prosper_pt_restir_di_trace called before prosper_pt_upload_scene");
    if (ctx->meshBuild)
    {
        const int prc = poll_mesh_build(ctx, false);
        if (prc != PROSPER_PT_OK) return prc;
    }
    if (ctx->accel && ctx->accel->stale && !ctx->accel->pending)
        return fail(PROSPER_PT_ERR_NO_SCENE, "the last prosper_pt_update_transforms failed: update the transforms again (or upload the scene) before tracing");
    if (width == 0 || height == 0) return fail(PROSPER_PT_ERR_INVALID_ARGUMENT, "prosper_pt_restir_di_trace: empty extent");
    if (pc->drawType >= PROSPER_DRAW_TYPE_COUNT) return fail(PROSPER_PT_ERR_INVALID_ARGUMENT, "drawType out of range");
    PPT_HIP(hipSetDevice(ctx->device));
    hipStream_t s = static_cast<hipStream_t>(stream);
    {
        int frc = flush_pending_update(ctx, s);
        if (frc == PROSPER_PT_OK) frc = flush_pending_lights(ctx, s);
        if (frc == PROSPER_PT_OK) frc = flush_pending_materials(ctx, s);
        if (frc != PROSPER_PT_OK) return frc;
        if (ctx->materialState && ctx->materialState->readyRecorded) PPT_HIP(hipStreamWaitEvent(s, ctx->materialState->ready, 0));
        if (ctx->accel && ctx->accel->sceneEventRecorded) PPT_HIP(hipStreamWaitEvent(s, ctx->accel->sceneEvent, 0));
        if (ctx->lights && ctx->lights->readyRecorded) PPT_HIP(hipStreamWaitEvent(s, ctx->lights->ready, 0));
    }
    const size_t pixels = (size_t)width * height;
    const size_t bytes = pixels * sizeof(float4);
    if (ctx->externalHdr)
    {
        if (ctx->externalHdrBytes < bytes) return fail(PROSPER_PT_ERR_INVALID_ARGUMENT, "caller-owned output buffer is too small");
        ctx->hdr = static_cast<float4 *>(ctx->externalHdr);
    }
    else
    {
        if (ctx->ownedHdrBytes < bytes || !ctx->ownedHdr)
        {
            PPT_HIP(hipStreamSynchronize(s));
            if (ctx->ownedHdr) PPT_HIP(hipFree(ctx->ownedHdr));
            ctx->ownedHdr = nullptr;
            PPT_HIP(hipMalloc((void **)&ctx->ownedHdr, bytes));
            PPT_HIP(hipMemset(ctx->ownedHdr, 0, bytes));
            ctx->ownedHdrBytes = bytes;
        }
        ctx->hdr = ctx->ownedHdr;
    }
    ctx->localWidth = width;
    ctx->height = height;
    ctx->lastWidth = width;
    ctx->stripeWidth = 0;
    ctx->stripeIndex = 0;
    ctx->stripeCount = 1;
    wait_for_gather_before_writing_tile(ctx, s);

    const void *ar = in->albedoRoughness, *nm = in->normalMetallic, *res = in->reservoirs;
    const float *depth = in->nonLinearDepth;
    if (!in->onDevice)
    {
        // host inputs: one scratch allocation, 16 + 16 + 4 + 8 bytes per pixel
        const size_t need = pixels * 44u + 64u;
        if (ctx->restirScratchBytes < need)
        {
            PPT_HIP(hipStreamSynchronize(s));
            if (ctx->restirScratch) PPT_HIP(hipFree(ctx->restirScratch));
            ctx->restirScratch = nullptr;
            ctx->restirScratchBytes = 0;
            PPT_HIP(hipMalloc(&ctx->restirScratch, need));
            ctx->restirScratchBytes = need;
        }
        uint8_t *base = static_cast<uint8_t *>(ctx->restirScratch);
        PPT_HIP(hipMemcpyAsync(base, in->albedoRoughness, pixels * 16u, hipMemcpyHostToDevice, s));
        PPT_HIP(hipMemcpyAsync(base + pixels * 16u, in->normalMetallic, pixels * 16u, hipMemcpyHostToDevice, s));
        PPT_HIP(hipMemcpyAsync(base + pixels * 32u, in->reservoirs, pixels * 8u, hipMemcpyHostToDevice, s));
        PPT_HIP(hipMemcpyAsync(base + pixels * 40u, in->nonLinearDepth, pixels * 4u, hipMemcpyHostToDevice, s));
        ar = base;
        nm = base + pixels * 16u;
        res = base + pixels * 32u;
        depth = reinterpret_cast<const float *>(base + pixels * 40u);
    }
    int32_t *ovf = nullptr;
    const int orc = ensure_stack_overflow(ctx, ctx->slots[0], kTraversalStackDepth, restir_grid_blocks(width, height), s, &ovf);
    if (orc != PROSPER_PT_OK) return orc;
    wait_for_slot(ctx->slots[0], s);
    const float eye[3] = {camera->eye.x, camera->eye.y, camera->eye.z};
    float c2w[16];
    std::memcpy(c2w, &camera->clipToWorld, 64);
    launch_restir_di_trace(
        ctx->scene, pc->drawType, pc->frameIndex, pc->flags, width, height, eye, c2w, ar, nm, depth, res, ctx->hdr, ovf, s);
    release_slot(ctx->slots[0], s);
    PPT_HIP(hipGetLastError());
    return mark_versions_read(ctx, s);
}

int prosper_pt_set_tone_map_lut(prosper_pt_ctx *ctx, const uint32_t *lut, uint32_t dim)
{
    if (!ctx || !lut || dim < 2 || dim > 256) return fail(PROSPER_PT_ERR_INVALID_ARGUMENT, "prosper_pt_set_tone_map_lut: bad argument");
    PPT_HIP(hipSetDevice(ctx->device));
    PPT_HIP(hipDeviceSynchronize());
    if (ctx->toneLut) PPT_HIP(hipFree(ctx->toneLut));
    ctx->toneLut = nullptr;
    ctx->toneLutDim = 0;
    const size_t bytes = (size_t)dim * dim * dim * sizeof(uint32_t);
    PPT_HIP(hipMalloc((void **)&ctx->toneLut, bytes));
    PPT_HIP(hipMemcpy(ctx->toneLut, lut, bytes, hipMemcpyHostToDevice));
    ctx->toneLutDim = dim;
    return PROSPER_PT_OK;
}

int prosper_pt_tone_map(
    prosper_pt_ctx *ctx, float exposure, float contrast, void *device_rgba8, uint8_t *host_rgba8, size_t byte_size,
    void *stream)
{
    if (!ctx || (!device_rgba8 && !host_rgba8)) return fail(PROSPER_PT_ERR_INVALID_ARGUMENT, "prosper_pt_tone_map: null argument");
    if (!ctx->hdr) return fail(PROSPER_PT_ERR_NO_SCENE, "nothing has been rendered yet");
    if (!ctx->toneLut) return fail(PROSPER_PT_ERR_INVALID_ARGUMENT, "prosper_pt_tone_map: no LUT (prosper_pt_set_tone_map_lut)");
    const uint32_t count = ctx->localWidth * ctx->height;
    const size_t bytes = (size_t)count * 4u;
    if (byte_size < bytes) return fail(PROSPER_PT_ERR_INVALID_ARGUMENT, "prosper_pt_tone_map: destination too small");
    PPT_HIP(hipSetDevice(ctx->device));
    hipStream_t s = static_cast<hipStream_t>(stream);
    void *out = device_rgba8;
    if (!out)
    {
        if (ctx->toneScratchBytes < bytes)
        {
            PPT_HIP(hipStreamSynchronize(s));
            if (ctx->toneScratch) PPT_HIP(hipFree(ctx->toneScratch));
            ctx->toneScratch = nullptr;
            ctx->toneScratchBytes = 0;
            PPT_HIP(hipMalloc(&ctx->toneScratch, bytes + 16));
            ctx->toneScratchBytes = bytes;
        }
        out = ctx->toneScratch;
    }
    launch_tone_map(ctx->hdr, ctx->toneLut, ctx->toneLutDim, exposure, contrast, out, count, s);
    PPT_HIP(hipGetLastError());
    if (host_rgba8)
    {
        PPT_HIP(hipMemcpyAsync(host_rgba8, out, bytes, hipMemcpyDeviceToHost, s));
        PPT_HIP(hipStreamSynchronize(s));
    }
    return PROSPER_PT_OK;
}

static int read_stage_counters(prosper_pt_ctx *ctx, unsigned long long host[kStageCount * kCounterCount], void *stream)
{
    PPT_HIP(hipSetDevice(ctx->device));
    hipStream_t s = static_cast<hipStream_t>(stream);
    PPT_HIP(hipMemcpyAsync(host, ctx->dCounters, kStageCount * kCounterCount * sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
    PPT_HIP(hipStreamSynchronize(s));
    return PROSPER_PT_OK;
}

int prosper_pt_get_counters(prosper_pt_ctx *ctx, prosper_pt_counters *out, void *stream)
{
    if (!ctx || !out) return fail(PROSPER_PT_ERR_INVALID_ARGUMENT, "prosper_pt_get_counters: null argument");
    unsigned long long host[kStageCount * kCounterCount] = {};
    const int rc = read_stage_counters(ctx, host, stream);
    if (rc != PROSPER_PT_OK) return rc;
    static_assert(sizeof(prosper_pt_counters) == kCounterCount * sizeof(uint64_t), "counter layout");
    unsigned long long sum[kCounterCount] = {};
    for (uint32_t st = 0; st < kStageCount; ++st)
        for (uint32_t i = 0; i < kCounterCount; ++i) sum[i] += host[st * kCounterCount + i];
    std::memcpy(out, sum, sizeof(*out));
    return PROSPER_PT_OK;
}

int prosper_pt_get_stage_counters(prosper_pt_ctx *ctx, uint32_t stage, prosper_pt_counters *out, void *stream)
{
    if (!ctx || !out || stage >= kStageCount)
        return fail(PROSPER_PT_ERR_INVALID_ARGUMENT, "prosper_pt_get_stage_counters: bad argument");
    unsigned long long host[kStageCount * kCounterCount] = {};
    const int rc = read_stage_counters(ctx, host, stream);
    if (rc != PROSPER_PT_OK) return rc;
    std::memcpy(out, host + stage * kCounterCount, sizeof(*out));
    return PROSPER_PT_OK;
}

int prosper_pt_reset_counters(prosper_pt_ctx *ctx, void *stream)
{
    if (!ctx) return fail(PROSPER_PT_ERR_INVALID_ARGUMENT, "prosper_pt_reset_counters: null context");
    PPT_HIP(hipSetDevice(ctx->device));
    PPT_HIP(hipMemsetAsync(
        ctx->dCounters, 0, kStageCount * kCounterCount * sizeof(unsigned long long), static_cast<hipStream_t>(stream)));
    return PROSPER_PT_OK;
}

int prosper_pt_set_kernel_timing(prosper_pt_ctx *ctx, int enabled)
{
    if (!ctx) return fail(PROSPER_PT_ERR_INVALID_ARGUMENT, "prosper_pt_set_kernel_timing: null context");
    // switching it off keeps the figures of the last timed render readable (later untimed renders record no events)
    if (enabled && !ctx->kernelTiming) ctx->timingValid = false;
    ctx->kernelTiming = enabled != 0;
    return PROSPER_PT_OK;
}

const char *prosper_pt_kernel_name(uint32_t index)
{
    static const char *names[PROSPER_PT_MAX_KERNELS] = {"wf_generate_extend", "wf_shade", "wf_trace", "wf_accumulate",
                                                        "",                   "",         "",         ""};
    return index < PROSPER_PT_MAX_KERNELS ? names[index] : "";
}

int prosper_pt_get_last_render_ms(prosper_pt_ctx *ctx, float *total_ms, float kernel_ms[PROSPER_PT_MAX_KERNELS])
{
    return prosper_pt_get_last_render_timing(ctx, total_ms, kernel_ms, nullptr);
}

int prosper_pt_get_last_render_timing(
    prosper_pt_ctx *ctx, float *total_ms, float kernel_ms[PROSPER_PT_MAX_KERNELS],
    uint32_t kernel_launches[PROSPER_PT_MAX_KERNELS])
{
    if (!ctx || !total_ms) return fail(PROSPER_PT_ERR_INVALID_ARGUMENT, "prosper_pt_get_last_render_timing: null argument");
    if (!ctx->timingValid)
        return fail(PROSPER_PT_ERR_INVALID_ARGUMENT, "no render has run with kernel timing enabled (prosper_pt_set_kernel_timing)");
    PPT_HIP(hipSetDevice(ctx->device));
    PPT_HIP(hipEventSynchronize(ctx->events[ctx->timedLaunches]));
    float perStage[PROSPER_PT_MAX_KERNELS] = {};
    uint32_t launches[PROSPER_PT_MAX_KERNELS] = {};
    float total = 0.0f;
    for (uint32_t i = 0; i < ctx->timedLaunches; ++i)
    {
        float ms = 0.0f;
        PPT_HIP(hipEventElapsedTime(&ms, ctx->events[i], ctx->events[i + 1]));
        perStage[ctx->eventStage[i]] += ms;
        launches[ctx->eventStage[i]] += 1;
        total += ms;
    }
    // The wavefront chains: their launches ran on the internal streams, between the caller's-stream
    // events counted above as the (unnamed) chains interval.  Their durations are as the device saw
    // them, i.e. a launch that shared the GPU with the other chain's is counted in full: per-stage sums
    // can exceed the wall time in `total_ms`, which stays the caller's-stream time.
    if (perStage[kStageChains] > 0.0f || launches[kStageChains] > 0)
    {
        bool any = false;
        const RenderSlot &slot = ctx->slots[ctx->timedSlot];
        for (uint32_t c = 0; c < kMaxChains; ++c)
            for (uint32_t i = 0; i < slot.chainLaunches[c]; ++i)
            {
                float ms = 0.0f;
                PPT_HIP(hipEventElapsedTime(&ms, slot.chainEvents[c][i], slot.chainEvents[c][i + 1]));
                perStage[slot.chainStage[c][i]] += ms;
                launches[slot.chainStage[c][i]] += 1;
                any = true;
            }
        if (any)
        {
            perStage[kStageChains] = 0.0f;
            launches[kStageChains] = 0;
        }
    }
    *total_ms = total;
    for (int i = 0; i < PROSPER_PT_MAX_KERNELS; ++i)
    {
        if (kernel_ms) kernel_ms[i] = perStage[i];
        if (kernel_launches) kernel_launches[i] = launches[i];
    }
    return PROSPER_PT_OK;
}

int prosper_pt_debug_srgb_monotonicity(
    prosper_pt_ctx *ctx, uint32_t first_bits, uint32_t last_bits, float *max_defect, uint64_t *decreases)
{
    if (!ctx || !max_defect || !decreases || first_bits > last_bits || last_bits >= 0x7F800000u)
        return fail(PROSPER_PT_ERR_INVALID_ARGUMENT, "prosper_pt_debug_srgb_monotonicity: bad argument");
    PPT_HIP(hipSetDevice(ctx->device));
    uint32_t *d = nullptr;
    PPT_HIP(hipMalloc((void **)&d, 8));
    hipError_t e = hipMemset(d, 0, 8);
    uint32_t host[2] = {0u, 0u};
    if (e == hipSuccess)
    {
        launch_srgb_monotonicity(first_bits, last_bits, d, nullptr);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpy(host, d, 8, hipMemcpyDeviceToHost);
    (void)hipFree(d);
    PPT_HIP(e);
    std::memcpy(max_defect, &host[0], 4);
    *decreases = host[1];
    return PROSPER_PT_OK;
}

int prosper_pt_eval_device_fn(
    prosper_pt_ctx *ctx, uint32_t fn, const float *in, uint32_t in_stride, float *out, uint32_t out_stride, uint32_t n)
{
    if (!ctx || !in || !out || fn >= PROSPER_PT_FN_COUNT || in_stride == 0 || out_stride == 0 ||
        (fn == PROSPER_PT_FN_BC7_BLOCK && (in_stride < 4 || out_stride < 16)))
        return fail(PROSPER_PT_ERR_INVALID_ARGUMENT, "prosper_pt_eval_device_fn: bad argument");
    PPT_HIP(hipSetDevice(ctx->device));
    float *dIn = nullptr, *dOut = nullptr;
    const size_t inBytes = (size_t)n * in_stride * 4, outBytes = (size_t)n * out_stride * 4;
    PPT_HIP(hipMalloc((void **)&dIn, inBytes + 16));
    hipError_t e = hipMalloc((void **)&dOut, outBytes + 16);
    if (e == hipSuccess) e = hipMemcpy(dIn, in, inBytes, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemset(dOut, 0, outBytes + 16);
    if (e == hipSuccess)
    {
        launch_eval_fn(fn, dIn, in_stride, dOut, out_stride, n, nullptr);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpy(out, dOut, outBytes, hipMemcpyDeviceToHost);
    (void)hipFree(dIn);
    (void)hipFree(dOut);
    if (e != hipSuccess) return fail(PROSPER_PT_ERR_HIP, std::string("eval_device_fn: ") + hipGetErrorString(e));
    return PROSPER_PT_OK;
}

} // extern "C"
