// host/rt_reference.cpp — see rt_reference.hpp.
#include "rt_reference.hpp"

#include <cstdio>
#include <cstdlib>
#include <stdexcept>
#include <string>

#include "../../../include/prosper_pt/prosper_host.h"

// prosper keeps its asserts in every build type (readme.md:88-92): programmer errors abort.
#define PROSPER_ASSERT(cond)                                                                                           \
    do                                                                                                                 \
    {                                                                                                                  \
        if (!(cond))                                                                                                   \
        {                                                                                                              \
            std::fprintf(stderr, "%s:%d: assertion failed: %s\n", __FILE__, __LINE__, #cond);                          \
            std::abort();                                                                                              \
        }                                                                                                              \
    } while (0)

namespace scene
{

void World::setSceneView(const prosper_pt_scene_view &view)
{
    m_view = view;
    m_haveView = true;
    m_dirty = true;
}

void World::buildAccelerationStructures(prosper_pt_ctx *ctx)
{
    PROSPER_ASSERT(m_haveView);
    if (m_ctx == ctx && !m_dirty) return;
    if (prosper_pt_upload_scene(ctx, &m_view) != PROSPER_PT_OK)
        throw std::runtime_error(std::string("World::buildAccelerationStructures: ") + prosper_pt_last_error());
    m_ctx = ctx;
    m_dirty = false;
}

void World::updateScene(prosper_pt_ctx *ctx, const prosper_ModelInstanceTransforms *transforms, uint32_t count)
{
    PROSPER_ASSERT(uploadedTo(ctx));
    if (prosper_pt_update_transforms(ctx, transforms, count) != PROSPER_PT_OK)
        throw std::runtime_error(std::string("World::updateScene: ") + prosper_pt_last_error());
}

void World::updateBuffers(prosper_pt_ctx *ctx)
{
    PROSPER_ASSERT(uploadedTo(ctx));
    if (prosper_pt_update_lights(ctx, m_view.directionalLight, m_view.pointLights, m_view.spotLights) != PROSPER_PT_OK)
        throw std::runtime_error(std::string("World::updateBuffers: ") + prosper_pt_last_error());
}

void World::adoptTextures(prosper_pt_ctx *ctx, const prosper_pt_texture_desc *textures, uint32_t firstSlot, uint32_t count)
{
    PROSPER_ASSERT(uploadedTo(ctx));
    if (prosper_pt_update_textures(ctx, textures, firstSlot, count) != PROSPER_PT_OK)
        throw std::runtime_error(std::string("World::adoptTextures: ") + prosper_pt_last_error());
}

void World::uploadMaterialDatas(prosper_pt_ctx *ctx, const prosper_MaterialData *materials, uint32_t count)
{
    PROSPER_ASSERT(uploadedTo(ctx));
    if (prosper_pt_update_materials(ctx, materials, 0, count) != PROSPER_PT_OK)
        throw std::runtime_error(std::string("World::uploadMaterialDatas: ") + prosper_pt_last_error());
}

void World::adoptMeshes(prosper_pt_ctx *ctx, const prosper_pt_mesh_update *meshes, uint32_t count)
{
    PROSPER_ASSERT(uploadedTo(ctx));
    if (prosper_pt_update_meshes(ctx, meshes, count) != PROSPER_PT_OK)
        throw std::runtime_error(std::string("World::adoptMeshes: ") + prosper_pt_last_error());
}

void World::finishMeshAdoption(prosper_pt_ctx *ctx)
{
    PROSPER_ASSERT(uploadedTo(ctx));
    if (prosper_pt_finish_mesh_updates(ctx) != PROSPER_PT_OK)
        throw std::runtime_error(std::string("World::finishMeshAdoption: ") + prosper_pt_last_error());
}

} // namespace scene

namespace render
{

namespace
{

// src/render/RtReference.cpp:31
constexpr uint32_t sFramePeriod = PROSPER_RT_FRAME_PERIOD;

// src/render/RtReference.cpp:68-88
struct ReferencePCFlags
{
    bool skipHistory{false};
    bool accumulate{false};
    bool ibl{false};
    bool depthOfField{false};
    bool clampIndirect{false};
};

uint32_t pcFlags(ReferencePCFlags flags)
{
    uint32_t ret = 0;
    ret |= (uint32_t)flags.skipHistory;
    ret |= (uint32_t)flags.accumulate << 1;
    ret |= (uint32_t)flags.ibl << 2;
    ret |= (uint32_t)flags.depthOfField << 3;
    ret |= (uint32_t)flags.clampIndirect << 4;
    return ret;
}

} // namespace

RtReference::~RtReference()
{
    if (m_ctx) prosper_pt_destroy(m_ctx);
}

void RtReference::init(int32_t deviceOrdinal, uint32_t createFlags)
{
    PROSPER_ASSERT(!m_initialized);
    prosper_pt_device_desc desc = {};
    desc.struct_size = sizeof(desc);
    desc.device_ordinal = deviceOrdinal;
    desc.flags = createFlags;
    if (prosper_pt_create(&desc, &m_ctx) != PROSPER_PT_OK)
        throw std::runtime_error(std::string("RtReference init failed: ") + prosper_pt_last_error());
    m_initialized = true;
}

void RtReference::recompileShaders()
{
    PROSPER_ASSERT(m_initialized);
    m_accumulationDirty = true;
}

void RtReference::drawUi(const UiState &wanted)
{
    PROSPER_ASSERT(m_initialized);
    // RtReference.cpp:148-159: the Accumulate checkbox does not dirty history, the others do;
    // the sliders clamp to [0, maxBounces] and [1, sMaxBounces]
    m_accumulate = wanted.accumulate;
    if (wanted.clampIndirect != m_clampIndirect)
    {
        m_clampIndirect = wanted.clampIndirect;
        m_accumulationDirty = true;
    }
    uint32_t maxBounces = wanted.maxBounces < 1u ? 1u : (wanted.maxBounces > sMaxBounces ? sMaxBounces : wanted.maxBounces);
    uint32_t roulette = wanted.rouletteStartBounce > m_maxBounces ? m_maxBounces : wanted.rouletteStartBounce;
    if (roulette != m_rouletteStartBounce)
    {
        m_rouletteStartBounce = roulette;
        m_accumulationDirty = true;
    }
    if (maxBounces != m_maxBounces)
    {
        m_maxBounces = maxBounces;
        m_accumulationDirty = true;
    }
}

RtReference::UiState RtReference::uiState() const
{
    return UiState{m_accumulate, m_clampIndirect, m_rouletteStartBounce, m_maxBounces};
}

RtReference::Output RtReference::record(
    void *stream, scene::World &world, const scene::Camera &cam, const Rect2D &renderArea, const Options &options,
    uint32_t nextFrame, uint32_t frameCount, const prosper_pt_tile_desc *tile, uint32_t renderFlags)
{
    PROSPER_ASSERT(m_initialized);
    PROSPER_ASSERT(frameCount >= 1);
    // The original selects per-frame descriptor sets with it (two frames in flight).  The library cycles through its
    // render slots by itself: pass PROSPER_PT_RENDER_PIPELINED in renderFlags for the same overlap of frames.
    (void)nextFrame;
    // the caller has run World::buildAccelerationStructures (App.cpp:573-578)
    PROSPER_ASSERT(world.uploadedTo(m_ctx));

    m_frameIndex = (m_frameIndex + 1) % sFramePeriod;

    // RtReference.cpp:189-216: a new previous image (colour dirty or extent change) restarts history
    if (options.colorDirty || !m_havePrevious || renderArea.width != m_previousWidth ||
        renderArea.height != m_previousHeight)
        m_accumulationDirty = true;

    const scene::CameraParameters &camParams = cam.parameters();
    prosper_ReferencePC pcBlock = {};
    pcBlock.drawType = static_cast<uint32_t>(options.drawType);
    ReferencePCFlags flags;
    flags.skipHistory = cam.changedThisFrame() || options.colorDirty || m_accumulationDirty;
    flags.accumulate = m_accumulate;
    flags.ibl = options.ibl;
    flags.depthOfField = options.depthOfField;
    flags.clampIndirect = m_clampIndirect;
    pcBlock.flags = pcFlags(flags);
    pcBlock.frameIndex = m_frameIndex;
    pcBlock.apertureDiameter = camParams.apertureDiameter;
    pcBlock.focusDistance = camParams.focusDistance;
    pcBlock.focalLength = camParams.focalLength;
    pcBlock.rouletteStartBounce = m_rouletteStartBounce;
    pcBlock.maxBounces = m_maxBounces;
    m_lastPC = pcBlock;

    PROSPER_ASSERT(renderArea.offsetX == 0 && renderArea.offsetY == 0); // RtReference.cpp:327
    if (prosper_pt_render_frames(
            m_ctx, &pcBlock, &cam.uniforms(), renderArea.width, renderArea.height, tile, frameCount, renderFlags,
            stream) != PROSPER_PT_OK)
        throw std::runtime_error(std::string("RtReference::record: ") + prosper_pt_last_error());
    // frames after the first of a batch take the following indices
    m_frameIndex = (m_frameIndex + (frameCount - 1)) % sFramePeriod;

    m_havePrevious = true; // the illumination image is preserved as next frame's history (:332-334)
    m_previousWidth = renderArea.width;
    m_previousHeight = renderArea.height;
    m_accumulationDirty = false;

    Output ret;
    void *ptr = nullptr;
    if (prosper_pt_get_hdr_device_ptr(m_ctx, &ptr, nullptr) != PROSPER_PT_OK)
        throw std::runtime_error(std::string("RtReference::record: ") + prosper_pt_last_error());
    ret.illumination = static_cast<const float *>(ptr);
    uint32_t lw = 0, h = 0;
    prosper_pt_get_local_extent(m_ctx, &lw, &h);
    ret.width = lw;
    ret.height = h;
    return ret;
}

void RtReference::releasePreserved()
{
    PROSPER_ASSERT(m_initialized);
    // RtReference.cpp:385-391: the history image goes back to the pool; the next record() finds
    // no valid previous image and restarts accumulation
    m_havePrevious = false;
}

} // namespace render

// ---- plain-C shims over the C++ host classes (include/prosper_pt/prosper_host.h) ----

struct prosper_host_camera
{
    scene::Camera cam;
};
struct prosper_host_rt_reference
{
    render::RtReference pass;
    scene::World world;
};

namespace
{
thread_local std::string g_hostError;
}

extern "C" {

const char *prosper_host_last_error(void) { return g_hostError.c_str(); }

prosper_host_camera *prosper_host_camera_create(void) { return new (std::nothrow) prosper_host_camera(); }
void prosper_host_camera_destroy(prosper_host_camera *c) { delete c; }
void prosper_host_camera_look_at(prosper_host_camera *c, const float eye[3], const float target[3], const float up[3])
{
    scene::CameraTransform t;
    for (int i = 0; i < 3; ++i)
    {
        t.eye[i] = eye[i];
        t.target[i] = target[i];
        t.up[i] = up[i];
    }
    c->cam.lookAt(t);
}
void prosper_host_camera_set_parameters(
    prosper_host_camera *c, float fov, float zN, float zF, float apertureDiameter, float focusDistance)
{
    scene::CameraParameters p;
    p.fov = fov;
    p.zN = zN;
    p.zF = zF;
    p.apertureDiameter = apertureDiameter;
    p.focusDistance = focusDistance;
    c->cam.setParameters(p);
}
void prosper_host_camera_update_resolution(prosper_host_camera *c, uint32_t width, uint32_t height)
{
    c->cam.updateResolution(width, height);
}
void prosper_host_camera_update_buffer(prosper_host_camera *c, prosper_CameraUniforms *out, float *focalLength)
{
    const prosper_CameraUniforms &u = c->cam.updateBuffer();
    if (out) *out = u;
    if (focalLength) *focalLength = c->cam.parameters().focalLength;
}
int prosper_host_camera_changed_this_frame(const prosper_host_camera *c) { return c->cam.changedThisFrame() ? 1 : 0; }
void prosper_host_set_error(const char *message) { g_hostError = message ? message : ""; }
void prosper_host_camera_end_frame(prosper_host_camera *c) { c->cam.endFrame(); }
scene::Camera *prosper_host_camera_object(prosper_host_camera *c) { return &c->cam; }

int prosper_host_rt_reference_create(int32_t deviceOrdinal, uint32_t createFlags, prosper_host_rt_reference **out)
{
    *out = nullptr;
    prosper_host_rt_reference *r = new (std::nothrow) prosper_host_rt_reference();
    if (!r) return PROSPER_PT_ERR_INVALID_ARGUMENT;
    try
    {
        r->pass.init(deviceOrdinal, createFlags);
    }
    catch (const std::exception &e)
    {
        g_hostError = e.what();
        delete r;
        return PROSPER_PT_ERR_NO_DEVICE;
    }
    *out = r;
    return PROSPER_PT_OK;
}
void prosper_host_rt_reference_destroy(prosper_host_rt_reference *r) { delete r; }
prosper_pt_ctx *prosper_host_rt_reference_context(prosper_host_rt_reference *r) { return r->pass.context(); }

int prosper_host_rt_reference_set_scene(prosper_host_rt_reference *r, const prosper_pt_scene_view *view)
{
    try
    {
        r->world.setSceneView(*view);
        r->world.buildAccelerationStructures(r->pass.context());
    }
    catch (const std::exception &e)
    {
        g_hostError = e.what();
        return PROSPER_PT_ERR_SCENE;
    }
    return PROSPER_PT_OK;
}

void prosper_host_rt_reference_draw_ui(
    prosper_host_rt_reference *r, int accumulate, int clampIndirect, uint32_t rouletteStartBounce, uint32_t maxBounces)
{
    render::RtReference::UiState s;
    s.accumulate = accumulate != 0;
    s.clampIndirect = clampIndirect != 0;
    s.rouletteStartBounce = rouletteStartBounce;
    s.maxBounces = maxBounces;
    r->pass.drawUi(s);
}

void prosper_host_rt_reference_recompile_shaders(prosper_host_rt_reference *r) { r->pass.recompileShaders(); }
void prosper_host_rt_reference_release_preserved(prosper_host_rt_reference *r) { r->pass.releasePreserved(); }

int prosper_host_rt_reference_record(
    prosper_host_rt_reference *r, prosper_host_camera *cam, uint32_t width, uint32_t height,
    const prosper_host_record_options *options, uint32_t frameCount, const prosper_pt_tile_desc *tile,
    uint32_t renderFlags, void *stream, prosper_ReferencePC *outPushConstants)
{
    try
    {
        render::RtReference::Options o;
        o.depthOfField = options->depthOfField != 0;
        o.ibl = options->ibl != 0;
        o.colorDirty = options->colorDirty != 0;
        o.drawType = static_cast<scene::DrawType>(options->drawType);
        render::Rect2D area;
        area.width = width;
        area.height = height;
        cam->cam.updateResolution(width, height);
        cam->cam.updateBuffer(); // App::drawFrame does this before Renderer::render (App.cpp:556)
        (void)r->pass.record(stream, r->world, cam->cam, area, o, 0, frameCount, tile, renderFlags);
        cam->cam.endFrame();
        if (outPushConstants) *outPushConstants = r->pass.lastPushConstants();
    }
    catch (const std::exception &e)
    {
        g_hostError = e.what();
        return PROSPER_PT_ERR_HIP;
    }
    return PROSPER_PT_OK;
}

} // extern "C"
