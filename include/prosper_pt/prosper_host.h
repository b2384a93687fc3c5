/*
 * prosper_pt/prosper_host.h — plain-C handles onto the C++ host layer
 * (prosper_amd/csrc/host/{camera,rt_reference}.hpp) so that non-C++ callers — the Python tests and
 * bench.py — drive the same `scene::Camera` / `render::RtReference` code a C++ application links.
 *
 * Replaces, for a headless caller:
 *   scene::Camera::{lookAt, perspective, updateBuffer}   src/scene/Camera.cpp:105-204,366-395
 *   render::RtReference::{init, drawUi, record, recompileShaders, releasePreserved}
 *                                                        src/render/RtReference.hpp:32-60
 *   World::buildAccelerationStructures                   src/scene/World.cpp:538-575
 */
#ifndef PROSPER_HOST_H
#define PROSPER_HOST_H

#include "prosper_pt.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct prosper_host_camera prosper_host_camera;
typedef struct prosper_host_rt_reference prosper_host_rt_reference;

/* RtReference::Options (src/render/RtReference.hpp:44-50) */
typedef struct prosper_host_record_options
{
    uint32_t depthOfField;
    uint32_t ibl;
    uint32_t colorDirty;
    uint32_t drawType; /* prosper_DrawType */
} prosper_host_record_options;

const char *prosper_host_last_error(void);

prosper_host_camera *prosper_host_camera_create(void);
void prosper_host_camera_destroy(prosper_host_camera *camera);
void prosper_host_camera_look_at(prosper_host_camera *camera, const float eye[3], const float target[3], const float up[3]);
void prosper_host_camera_set_parameters(
    prosper_host_camera *camera, float fov, float zN, float zF, float apertureDiameter, float focusDistance);
void prosper_host_camera_update_resolution(prosper_host_camera *camera, uint32_t width, uint32_t height);
void prosper_host_camera_update_buffer(prosper_host_camera *camera, prosper_CameraUniforms *out, float *focalLength);
int prosper_host_camera_changed_this_frame(const prosper_host_camera *camera);
void prosper_host_camera_end_frame(prosper_host_camera *camera);

int prosper_host_rt_reference_create(int32_t deviceOrdinal, uint32_t createFlags, prosper_host_rt_reference **out);
void prosper_host_rt_reference_destroy(prosper_host_rt_reference *pass);
prosper_pt_ctx *prosper_host_rt_reference_context(prosper_host_rt_reference *pass);
/* World::setSceneView + buildAccelerationStructures */
int prosper_host_rt_reference_set_scene(prosper_host_rt_reference *pass, const prosper_pt_scene_view *view);
void prosper_host_rt_reference_draw_ui(
    prosper_host_rt_reference *pass, int accumulate, int clampIndirect, uint32_t rouletteStartBounce,
    uint32_t maxBounces);
void prosper_host_rt_reference_recompile_shaders(prosper_host_rt_reference *pass);
void prosper_host_rt_reference_release_preserved(prosper_host_rt_reference *pass);
/* Camera::updateBuffer + RtReference::record + end of frame; returns the ReferencePC it pushed. */
int prosper_host_rt_reference_record(
    prosper_host_rt_reference *pass, prosper_host_camera *camera, uint32_t width, uint32_t height,
    const prosper_host_record_options *options, uint32_t frameCount, const prosper_pt_tile_desc *tile,
    uint32_t renderFlags, void *stream, prosper_ReferencePC *outPushConstants);

/* render::TiledRtReference (host/tiled_rt_reference.hpp): the pass on one rank of a multi-GPU job - RtReference::record
 * for the rank's interleaved 16-pixel stripes, then the RCCL gather of the ranks' HDR tiles to `root` and the
 * de-interleave kernel there (prosper_pt_gather_tiles).  No counterpart in the reference, which asserts
 * renderArea.offset == 0 (src/render/RtReference.cpp:327).  `commId`: prosper_pt_comm_get_unique_id of one rank.
 * record() returns, on the root, the device pointer of the gathered width*height RGBA32F image; readers enqueue
 * wait_for_gather on their stream first. */
typedef struct prosper_host_tiled_rt_reference prosper_host_tiled_rt_reference;
int prosper_host_tiled_rt_reference_create(
    int32_t deviceOrdinal, uint32_t rank, uint32_t ranks, const uint8_t commId[PROSPER_PT_COMM_ID_BYTES], uint32_t root,
    uint32_t createFlags, prosper_host_tiled_rt_reference **out);
void prosper_host_tiled_rt_reference_destroy(prosper_host_tiled_rt_reference *pass);
prosper_pt_ctx *prosper_host_tiled_rt_reference_context(prosper_host_tiled_rt_reference *pass);
int prosper_host_tiled_rt_reference_set_scene(prosper_host_tiled_rt_reference *pass, const prosper_pt_scene_view *view);
int prosper_host_tiled_rt_reference_record(
    prosper_host_tiled_rt_reference *pass, prosper_host_camera *camera, uint32_t width, uint32_t height,
    const prosper_host_record_options *options, uint32_t frameCount, uint32_t renderFlags, void *stream,
    const float **outIllumination);
int prosper_host_tiled_rt_reference_wait_for_gather(prosper_host_tiled_rt_reference *pass, void *stream);

/* render::ToneMap (host/tone_map.hpp; reference src/render/ToneMap.hpp:16-52): init with the LUT file
 * (res/texture/tony_mc_mapface.dds) or its texels, drawUi's two sliders, record into caller-owned device memory. */
typedef struct prosper_host_tone_map prosper_host_tone_map;
int prosper_host_tone_map_create(prosper_pt_ctx *ctx, const char *lutDdsPath, prosper_host_tone_map **out);
int prosper_host_tone_map_create_from_texels(
    prosper_pt_ctx *ctx, const uint32_t *lutR9G9B9E5, uint32_t dim, prosper_host_tone_map **out);
void prosper_host_tone_map_destroy(prosper_host_tone_map *pass);
void prosper_host_tone_map_draw_ui(prosper_host_tone_map *pass, float exposure, float contrast);
int prosper_host_tone_map_record(prosper_host_tone_map *pass, void *stream, void *deviceRgba8, size_t byteSize);

#ifdef __cplusplus
}
#endif

#endif /* PROSPER_HOST_H */
