// pt_materials.hpp — textures, material texture packs, alpha materials: what prosper_pt_upload_scene builds for the whole
// scene and prosper_pt_update_textures / _materials rebuild for what changed (pt_materials.cpp; private to the library).
#pragma once

#include "pt_context.hpp"

namespace ppt
{

// One entry of materialTextures[] on the device: the caller's level-0 texels (RGBA8 rows or BC7 blocks) go to a device
// staging area, a kernel re-tiles / decodes them into a new scene allocation.  `staging` must hold
// texture_staging_bytes(desc) bytes; everything is enqueued on `stream` (the host copy included: the caller's memory may
// go once `stream` has been synchronised).  `pinned` (host, texture_staging_bytes(desc) bytes): the texels go through it - the
// caller's memory has been read when the call returns, and the runtime never pins the caller's pages (which, freed soon
// after, would stop every queue of the process for 20-30 ms while the driver unmaps them: prosper_pt_update_textures).
size_t texture_staging_bytes(const prosper_pt_texture_desc &t);
int validate_texture(const prosper_pt_texture_desc &t, uint32_t index);
int create_device_texture(
    prosper_pt_ctx *ctx, const prosper_pt_texture_desc &t, void *staging, hipStream_t stream, DeviceTexture *out, void *pinned = nullptr);

// The interleaved copy of a material's three textures (pt_scene.hpp MaterialPack) where they share extent and sampler;
// texels == nullptr where the material is not packable.  `wide`: 16-byte texels also for opaque materials.
int build_material_pack(
    prosper_pt_ctx *ctx, const prosper_MaterialData &m, const std::vector<DeviceTexture> &textures, bool noPacks, bool wide,
    hipStream_t stream, MaterialPack *out);

// sampleAlpha of one material as the any-hit needs it, with its alpha bounds (pt_scene.hpp AlphaMaterial)
int build_alpha_material(
    prosper_pt_ctx *ctx, const prosper_MaterialData &m, const std::vector<DeviceTexture> &textures,
    const std::vector<prosper_pt_sampler_desc> &samplers, hipStream_t stream, AlphaMaterial *out, uint64_t *boundBytes);

// compact (8-byte) packs and batched texel fetches where the texel footprint outgrows the caches
inline bool texel_set_is_big(uint64_t texelBytes) { return texelBytes > (32ull << 20); }

// `p` (a scene allocation) was replaced by an update: nothing new will read it, a frame in flight still may.  Freed by
// collect_retired once enough has piled up to be worth ONE device synchronisation, or with the scene.
void retire(prosper_pt_ctx *ctx, const void *p);
int collect_retired(prosper_pt_ctx *ctx);

// the staged tables into the next device version, on the stream of the render that is about to read them
int flush_pending_materials(prosper_pt_ctx *ctx, hipStream_t stream);

} // namespace ppt
