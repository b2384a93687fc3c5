// host/rt_reference.hpp — render::RtReference of the headless host layer.
//
// Same five-method surface as prosper's pass (reference: src/render/RtReference.hpp:32-60) with
// the Vulkan handles replaced: `init` loads the gfx950 code object instead of compiling shaders
// and building an RT pipeline + SBT (RtReference.cpp:104-120,405-707); `record` pushes the same
// ReferencePC and calls prosper_pt_render where the original records cb.traceRaysKHR
// (RtReference.cpp:161-383).  ImGui state (drawUi, RtReference.cpp:148-159) becomes UiState.
#pragma once

#include <cstdint>

#include "../../../include/prosper_pt/prosper_pt.h"
#include "camera.hpp"

namespace scene
{

// src/scene/DrawType.hpp:23-26
enum class DrawType : uint8_t
{
    Default, PrimitiveID, MeshletID, MeshID, MaterialID, Position, ShadingNormal, TexCoord0, Albedo, Roughness,
    Metallic, Count
};

// What the pass needs from scene::World (src/scene/World.hpp:18-82): the flattened scene and
// the moment its acceleration structures are (re)built (App.cpp:573-578 -> World.cpp:538-575).
class World
{
  public:
    // Borrowed pointers: the view's arrays must outlive buildAccelerationStructures().
    void setSceneView(const prosper_pt_scene_view &view);
    // Uploads scene + builds the BVH on `ctx` when the view changed since the last build.
    // Throws std::runtime_error on failure.
    void buildAccelerationStructures(prosper_pt_ctx *ctx);
    // World::updateScene + the per-frame TLAS build (src/scene/World.cpp:359-466,749-802,878-928; App.cpp:516-578 calls
    // them every frame): the whole transform table of the uploaded scene.  Cheap to call unconditionally - an unchanged
    // table is a no-op, a changed one a GPU refit run by the next record().  Throws std::runtime_error on failure.
    void updateScene(prosper_pt_ctx *ctx, const prosper_ModelInstanceTransforms *transforms, uint32_t count);
    // World::updateBuffers' light writes (src/scene/World.cpp:468-536, :531-535: the three light buffers, every frame), from
    // the light pointers of the scene view.  An unchanged set is a no-op.
    void updateBuffers(prosper_pt_ctx *ctx);
    // While the scene streams in (src/scene/WorldData.cpp:588-647 handleDeferredLoading, once per frame from App.cpp:601):
    // the images the texture worker finished this frame take their slots of materialTextures[] (WorldData.cpp:2182-2206
    // updateDescriptorsWithNewTextures; image k of the glTF is slot k + 1) ...
    void adoptTextures(prosper_pt_ctx *ctx, const prosper_pt_texture_desc *textures, uint32_t firstSlot, uint32_t count);
    // ... and uploadMaterialDatas (WorldData.cpp:568-586, every frame from App.cpp:526-529) hands over the whole material
    // table: entries that did not change cost a memcmp, the materials updateMaterials() just switched from their
    // placeholders (WorldData.cpp:2208-2239) get their texture packs and alpha bounds.  Both take effect at the head of the
    // next record()'s launches; frames in flight keep what they started with.  Throw std::runtime_error on failure.
    void uploadMaterialDatas(prosper_pt_ctx *ctx, const prosper_MaterialData *materials, uint32_t count);
    // WorldData::pollMeshWorker (WorldData.cpp:2003-2110): the meshes the mesh worker finished this frame - their metadata and
    // MeshInfo slots and their bytes of a geometry buffer.  The scene was set with every mesh slot, the unloaded ones with
    // bufferIndex 0xFFFFFFFF as prosper keeps them; a model instance shows once all its sub-meshes are there
    // (World::buildNextBlas, World.cpp:598-606, 909-915).  Returns at once: the context builds the geometry beside the
    // frame loop, and the first record() after that shows it.  finishMeshAdoption waits for everything handed over so far.
    void adoptMeshes(prosper_pt_ctx *ctx, const prosper_pt_mesh_update *meshes, uint32_t count);
    void finishMeshAdoption(prosper_pt_ctx *ctx);
    [[nodiscard]] bool uploadedTo(const prosper_pt_ctx *ctx) const { return m_ctx == ctx && !m_dirty; }

  private:
    prosper_pt_scene_view m_view = {};
    bool m_haveView = false;
    bool m_dirty = true;
    const prosper_pt_ctx *m_ctx = nullptr;
};

} // namespace scene

namespace render
{

struct Rect2D
{
    int32_t offsetX = 0, offsetY = 0;
    uint32_t width = 0, height = 0;
};

class RtReference
{
  public:
    static constexpr uint32_t sMaxBounces = PROSPER_RT_MAX_BOUNCES;

    RtReference() noexcept = default;
    ~RtReference();
    RtReference(const RtReference &) = delete;
    RtReference &operator=(const RtReference &) = delete;

    // Throws std::runtime_error when no gfx950 device is usable (there is no CPU fallback).
    void init(int32_t deviceOrdinal, uint32_t createFlags = 0);
    // The kernels are compiled ahead of time; like a successful recompile in the original this
    // only restarts accumulation (RtReference.cpp:140-145).
    void recompileShaders();

    struct UiState
    {
        bool accumulate{true};
        bool clampIndirect{true};
        uint32_t rouletteStartBounce{3};
        uint32_t maxBounces{sMaxBounces};
    };
    // What the ImGui widgets of drawUi() do: changing anything but `accumulate` dirties history.
    void drawUi(const UiState &wanted);
    [[nodiscard]] UiState uiState() const;

    struct Options
    {
        bool depthOfField{false};
        bool ibl{false};
        bool colorDirty{false};
        scene::DrawType drawType{scene::DrawType::Default};
    };
    struct Output
    {
        const float *illumination{nullptr}; // device pointer, RGBA32F, width*height texels
        uint32_t width{0};
        uint32_t height{0};
    };
    // `frameCount` > 1 renders that many consecutive accumulated frames in one launch.
    [[nodiscard]] Output record(
        void *stream, scene::World &world, const scene::Camera &cam, const Rect2D &renderArea, const Options &options,
        uint32_t nextFrame, uint32_t frameCount = 1, const prosper_pt_tile_desc *tile = nullptr,
        uint32_t renderFlags = 0);
    void releasePreserved();

    [[nodiscard]] prosper_pt_ctx *context() const { return m_ctx; }
    [[nodiscard]] const prosper_ReferencePC &lastPushConstants() const { return m_lastPC; }

  private:
    bool m_initialized{false};
    prosper_pt_ctx *m_ctx{nullptr};

    bool m_accumulationDirty{true};
    bool m_accumulate{true};
    bool m_clampIndirect{true};
    uint32_t m_frameIndex{0};
    uint32_t m_rouletteStartBounce{3};
    uint32_t m_maxBounces{sMaxBounces};

    bool m_havePrevious{false};
    uint32_t m_previousWidth{0}, m_previousHeight{0};
    prosper_ReferencePC m_lastPC{};
};

} // namespace render

// the scene::Camera behind a prosper_host_camera handle (for the other shims of the host layer)
struct prosper_host_camera;
extern "C" scene::Camera *prosper_host_camera_object(prosper_host_camera *camera);
