// host/tiled_rt_reference.cpp — see tiled_rt_reference.hpp.
#include "tiled_rt_reference.hpp"

#include <new>
#include <stdexcept>
#include <string>

#include "../../../include/prosper_pt/prosper_host.h"

namespace render
{

void TiledRtReference::createCommId(uint8_t id[sCommIdBytes])
{
    if (prosper_pt_comm_get_unique_id(id) != PROSPER_PT_OK)
        throw std::runtime_error(std::string("TiledRtReference::createCommId: ") + prosper_pt_last_error());
}

void TiledRtReference::init(
    int32_t deviceOrdinal, uint32_t rank, uint32_t ranks, const uint8_t commId[sCommIdBytes], uint32_t root,
    uint32_t createFlags)
{
    m_pass.init(deviceOrdinal, createFlags);
    m_tile = prosper_pt_tile_desc{sStripeWidth, rank, ranks};
    m_root = root;
    if (ranks > 1 && prosper_pt_comm_init(m_pass.context(), commId, rank, ranks) != PROSPER_PT_OK)
        throw std::runtime_error(std::string("TiledRtReference::init: ") + prosper_pt_last_error());
}

TiledRtReference::Output TiledRtReference::record(
    void *stream, scene::World &world, const scene::Camera &cam, const Rect2D &renderArea,
    const RtReference::Options &options, uint32_t nextFrame, uint32_t frameCount, uint32_t renderFlags)
{
    Output out;
    out.tile = m_pass.record(
        stream, world, cam, renderArea, options, nextFrame, frameCount, m_tile.stripeCount > 1 ? &m_tile : nullptr,
        renderFlags);
    out.width = renderArea.width;
    out.height = renderArea.height;
    // the one data-path collective: per-rank HDR tiles to the root over RCCL, de-interleaved there
    if (prosper_pt_gather_tiles(m_pass.context(), m_root, nullptr, 0, 0, stream) != PROSPER_PT_OK)
        throw std::runtime_error(std::string("TiledRtReference::record: ") + prosper_pt_last_error());
    if (isRoot() || m_tile.stripeCount == 1)
    {
        void *full = nullptr;
        if (prosper_pt_get_gathered_device_ptr(m_pass.context(), &full, nullptr, nullptr) != PROSPER_PT_OK)
            throw std::runtime_error(std::string("TiledRtReference::record: ") + prosper_pt_last_error());
        out.illumination = static_cast<const float *>(full);
    }
    return out;
}

void TiledRtReference::waitForGather(void *stream)
{
    if (prosper_pt_gather_wait(m_pass.context(), stream) != PROSPER_PT_OK)
        throw std::runtime_error(std::string("TiledRtReference::waitForGather: ") + prosper_pt_last_error());
}

} // namespace render

// ---- plain-C shims (include/prosper_pt/prosper_host.h) ----

struct prosper_host_tiled_rt_reference
{
    render::TiledRtReference pass;
    scene::World world;
};

extern "C" {

void prosper_host_set_error(const char *message); // rt_reference.cpp

int prosper_host_tiled_rt_reference_create(
    int32_t deviceOrdinal, uint32_t rank, uint32_t ranks, const uint8_t commId[PROSPER_PT_COMM_ID_BYTES], uint32_t root,
    uint32_t createFlags, prosper_host_tiled_rt_reference **out)
{
    *out = nullptr;
    prosper_host_tiled_rt_reference *r = new (std::nothrow) prosper_host_tiled_rt_reference();
    if (!r) return PROSPER_PT_ERR_INVALID_ARGUMENT;
    try
    {
        r->pass.init(deviceOrdinal, rank, ranks, commId, root, createFlags);
    }
    catch (const std::exception &e)
    {
        prosper_host_set_error(e.what());
        delete r;
        return PROSPER_PT_ERR_NO_DEVICE;
    }
    *out = r;
    return PROSPER_PT_OK;
}
void prosper_host_tiled_rt_reference_destroy(prosper_host_tiled_rt_reference *r) { delete r; }
prosper_pt_ctx *prosper_host_tiled_rt_reference_context(prosper_host_tiled_rt_reference *r) { return r->pass.pass().context(); }

int prosper_host_tiled_rt_reference_set_scene(prosper_host_tiled_rt_reference *r, const prosper_pt_scene_view *view)
{
    try
    {
        r->world.setSceneView(*view);
        r->world.buildAccelerationStructures(r->pass.pass().context());
    }
    catch (const std::exception &e)
    {
        prosper_host_set_error(e.what());
        return PROSPER_PT_ERR_SCENE;
    }
    return PROSPER_PT_OK;
}

int prosper_host_tiled_rt_reference_record(
    prosper_host_tiled_rt_reference *r, prosper_host_camera *cam, uint32_t width, uint32_t height,
    const prosper_host_record_options *options, uint32_t frameCount, uint32_t renderFlags, void *stream,
    const float **outIllumination)
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
        scene::Camera &camera = *prosper_host_camera_object(cam);
        camera.updateResolution(width, height);
        camera.updateBuffer();
        const render::TiledRtReference::Output out = r->pass.record(stream, r->world, camera, area, o, 0, frameCount, renderFlags);
        camera.endFrame();
        if (outIllumination) *outIllumination = out.illumination;
    }
    catch (const std::exception &e)
    {
        prosper_host_set_error(e.what());
        return PROSPER_PT_ERR_HIP;
    }
    return PROSPER_PT_OK;
}

int prosper_host_tiled_rt_reference_wait_for_gather(prosper_host_tiled_rt_reference *r, void *stream)
{
    try
    {
        r->pass.waitForGather(stream);
    }
    catch (const std::exception &e)
    {
        prosper_host_set_error(e.what());
        return PROSPER_PT_ERR_HIP;
    }
    return PROSPER_PT_OK;
}

} // extern "C"
