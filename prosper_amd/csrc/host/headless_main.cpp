// host/headless_main.cpp — prosper_headless: a C++ consumer of the host layer, the way prosper's App drives its
// passes (src/App.cpp:516-578 update scene -> build acceleration structures -> Renderer::render ->
// RtReference::record; src/render/Renderer.cpp:406-428), without Vulkan, a window or Python.
//
//   prosper_headless <width> <height> <frames> <out.rgba32f> [device]      (frames = 0: write the packed geometry buffer only)
//
// Builds a small scene in the reference's data contract (packed fp16x4 positions, snorm10 normals / tangents,
// fp16x2 uv, u16 indices in one geometry buffer: src/scene/DeferredLoadingContext.cpp:442-490,775-784), uploads
// it through scene::World, renders `frames` accumulated frames with render::TiledRtReference (one rank: the same
// record() a multi-GPU rank runs) and writes the RGBA32F image as raw floats.  tests/test_host_cpp.py builds the
// same scene through the Python handles and checks the file against the oracle bit for bit.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <vector>

#include "../../../include/prosper_pt/prosper_pt.h"
#include "camera.hpp"
#include "tiled_rt_reference.hpp"

namespace
{

// glm::packHalf: binary16, round to nearest even (the values used here are exactly representable)
uint16_t pack_half(float f)
{
    uint32_t x;
    std::memcpy(&x, &f, 4);
    const uint32_t sign = (x >> 16) & 0x8000u;
    const uint32_t ax = x & 0x7FFFFFFFu;
    if (ax == 0) return (uint16_t)sign;
    const int32_t e = (int32_t)(ax >> 23) - 127;
    const uint32_t m = ax & 0x007FFFFFu;
    if (e < -14 || e > 15 || (m & 0x1FFFu)) throw std::runtime_error("headless scene: value is not an exact normal half");
    return (uint16_t)(sign | (uint32_t)((e + 15) << 10) | (m >> 13));
}

// glm::packSnorm3x10_1x2: round(clamp(v, -1, 1) * (511, 511, 511, 1))
uint32_t pack_snorm(float x, float y, float z, float w)
{
    auto q = [](float v, float s) { return (uint32_t)(int32_t)std::lround(std::fmax(-1.f, std::fmin(1.f, v)) * s); };
    return (q(x, 511.f) & 0x3FFu) | ((q(y, 511.f) & 0x3FFu) << 10) | ((q(z, 511.f) & 0x3FFu) << 20) | ((q(w, 1.f) & 0x3u) << 30);
}

struct Mesh
{
    uint32_t vertexCount, indexCount;
    prosper_GeometryMetadata metadata;
};

// Appends a quad (two triangles) to the geometry buffer in the blob order of DeferredLoadingContext.cpp:775-784
Mesh add_quad(std::vector<uint32_t> &buffer, const float p[4][3], const float n[3], const float t[4])
{
    Mesh mesh = {};
    mesh.vertexCount = 4;
    mesh.indexCount = 6;
    prosper_GeometryMetadata &m = mesh.metadata;
    std::memset(&m, 0xFF, sizeof(m)); // every offset absent
    m.bufferIndex = 0;
    m.usesShortIndices = 1;
    const uint16_t idx[6] = {0, 1, 2, 0, 2, 3};
    m.indicesOffset = (uint32_t)buffer.size() * 2u; // u16 units
    for (int i = 0; i < 6; i += 2) buffer.push_back((uint32_t)idx[i] | ((uint32_t)idx[i + 1] << 16));
    m.positionsOffset = (uint32_t)buffer.size();
    for (int v = 0; v < 4; ++v)
    {
        buffer.push_back((uint32_t)pack_half(p[v][0]) | ((uint32_t)pack_half(p[v][1]) << 16));
        buffer.push_back((uint32_t)pack_half(p[v][2]) | ((uint32_t)pack_half(1.0f) << 16));
    }
    m.normalsOffset = (uint32_t)buffer.size();
    for (int v = 0; v < 4; ++v) buffer.push_back(pack_snorm(n[0], n[1], n[2], 0.f));
    m.tangentsOffset = (uint32_t)buffer.size();
    for (int v = 0; v < 4; ++v) buffer.push_back(pack_snorm(t[0], t[1], t[2], t[3]));
    m.texCoord0sOffset = (uint32_t)buffer.size();
    const float uv[4][2] = {{0, 0}, {1, 0}, {1, 1}, {0, 1}};
    for (int v = 0; v < 4; ++v) buffer.push_back((uint32_t)pack_half(uv[v][0]) | ((uint32_t)pack_half(uv[v][1]) << 16));
    return mesh;
}

prosper_MaterialData material(float r, float g, float b, float metallic, float roughness)
{
    prosper_MaterialData m = {};
    m.baseColorFactor = prosper_vec4{r, g, b, 1.0f};
    m.metallicFactor = metallic;
    m.roughnessFactor = roughness;
    m.alphaCutoff = 0.5f;
    m.alphaMode = PROSPER_ALPHA_MODE_OPAQUE;
    return m;
}

prosper_ModelInstanceTransforms identity_transform()
{
    prosper_ModelInstanceTransforms t = {};
    for (int r = 0; r < 3; ++r)
    {
        float row[4] = {0, 0, 0, 0};
        row[r] = 1.0f;
        t.modelToWorld.col[r] = prosper_vec4{row[0], row[1], row[2], row[3]};
        t.normalToWorld.col[r] = prosper_vec4{row[0], row[1], row[2], row[3]};
    }
    return t;
}

} // namespace

int main(int argc, char **argv)
{
    if (argc < 5)
    {
        std::fprintf(stderr, "usage: %s <width> <height> <frames> <out.rgba32f> [device]\n", argv[0]);
        return 2;
    }
    const uint32_t width = (uint32_t)std::atoi(argv[1]), height = (uint32_t)std::atoi(argv[2]);
    const uint32_t frames = (uint32_t)std::atoi(argv[3]);
    const int32_t device = argc > 5 ? std::atoi(argv[5]) : 0;
    try
    {
        // ---- the scene: a floor, a back wall and a tilted metallic panel; a sun and a point light ----
        std::vector<uint32_t> geometry;
        const float floorP[4][3] = {{-4, 0, 4}, {4, 0, 4}, {4, 0, -4}, {-4, 0, -4}};
        const float wallP[4][3] = {{-4, 0, -4}, {4, 0, -4}, {4, 4, -4}, {-4, 4, -4}};
        const float panelP[4][3] = {{-1, 0.5f, 0}, {1, 0.5f, 0}, {1, 2.5f, -1}, {-1, 2.5f, -1}};
        const float up[3] = {0, 1, 0}, front[3] = {0, 0, 1}, tangent[4] = {1, 0, 0, 1};
        // the panel's normal (0, 1, 2) / sqrt(5): packed snorm10, then re-normalised by the shader (geometry.glsl:102)
        const float panelN[3] = {0.0f, 0.4472136f, 0.8944272f};
        const Mesh meshes[3] = {add_quad(geometry, floorP, up, tangent), add_quad(geometry, wallP, front, tangent),
                                add_quad(geometry, panelP, panelN, tangent)};
        if (frames == 0)
        {
            // no GPU needed: the packed geometry buffer alone (the CPU tests compare it with the Python packer's)
            FILE *out = std::fopen(argv[4], "wb");
            if (!out || std::fwrite(geometry.data(), 4, geometry.size(), out) != geometry.size())
                throw std::runtime_error("cannot write the output file");
            std::fclose(out);
            std::printf("prosper_headless: %zu geometry words\n", geometry.size());
            return 0;
        }
        prosper_GeometryMetadata metadatas[3];
        prosper_pt_mesh_info meshInfos[3];
        for (int i = 0; i < 3; ++i)
        {
            metadatas[i] = meshes[i].metadata;
            meshInfos[i] = prosper_pt_mesh_info{meshes[i].vertexCount, meshes[i].indexCount, 0u, (uint32_t)(i + 1)};
        }
        const prosper_MaterialData materials[4] = {material(1, 1, 1, 1.0f, 1.0f), material(0.75f, 0.75f, 0.75f, 0.0f, 0.875f),
                                                   material(0.25f, 0.5f, 0.75f, 0.0f, 0.5f), material(0.875f, 0.75f, 0.5f, 1.0f, 0.25f)};
        const prosper_DrawInstance drawInstances[3] = {{0, 0, 1}, {1, 1, 2}, {2, 2, 3}};
        const prosper_ModelInstanceTransforms transforms[3] = {identity_transform(), identity_transform(), identity_transform()};
        const uint32_t white = 0xFFFFFFFFu;
        const prosper_pt_texture_desc textures[1] = {{&white, 1, 1, PROSPER_PT_FORMAT_RGBA8_UNORM, 0}};
        const prosper_pt_sampler_desc samplers[1] = {
            {PROSPER_PT_FILTER_LINEAR, PROSPER_PT_FILTER_LINEAR, PROSPER_PT_WRAP_REPEAT, PROSPER_PT_WRAP_REPEAT}};
        prosper_DirectionalLightParameters sun = {};
        sun.irradiance = prosper_vec4{2.0f, 2.0f, 2.0f, 0.0f};
        sun.direction = prosper_vec4{-1.0f, -1.0f, -1.0f, 0.0f};
        static prosper_PointLightsBuffer pointLights;
        static prosper_SpotLightsBuffer spotLights;
        std::memset(&pointLights, 0, sizeof(pointLights));
        std::memset(&spotLights, 0, sizeof(spotLights));
        pointLights.lights[0].radianceAndRadius = prosper_vec4{4.0f, 3.0f, 2.0f, 16.0f};
        pointLights.lights[0].position = prosper_vec4{1.5f, 3.0f, 1.0f, 1.0f};
        pointLights.count = 1;

        const void *buffers[1] = {geometry.data()};
        const uint64_t bufferBytes[1] = {geometry.size() * 4u};
        prosper_pt_scene_view view = {};
        view.struct_size = sizeof(view);
        view.geometryBuffers = buffers;
        view.geometryBufferByteSizes = bufferBytes;
        view.geometryBufferCount = 1;
        view.meshCount = 3;
        view.geometryMetadatas = metadatas;
        view.meshInfos = meshInfos;
        view.drawInstances = drawInstances;
        view.drawInstanceCount = 3;
        view.modelInstanceCount = 3;
        view.modelInstanceTransforms = transforms;
        view.materials = materials;
        view.materialCount = 4;
        view.textureCount = 1;
        view.textures = textures;
        view.samplers = samplers;
        view.samplerCount = 1;
        view.directionalLight = &sun;
        view.pointLights = &pointLights;
        view.spotLights = &spotLights;

        // ---- the frame loop of App::drawFrame, headless ----
        render::TiledRtReference pass;
        pass.init(device, /*rank*/ 0, /*ranks*/ 1, nullptr);
        scene::World world;
        world.setSceneView(view);
        world.buildAccelerationStructures(pass.pass().context());

        scene::Camera camera;
        scene::CameraTransform transform;
        transform.eye[0] = 0.0f, transform.eye[1] = 2.0f, transform.eye[2] = 5.0f;
        transform.target[0] = 0.0f, transform.target[1] = 1.0f, transform.target[2] = 0.0f;
        camera.lookAt(transform);
        camera.setParameters(scene::CameraParameters{});
        camera.updateResolution(width, height);

        render::RtReference::UiState ui;
        ui.maxBounces = 4;
        pass.pass().drawUi(ui);
        render::RtReference::Options options;
        render::Rect2D area;
        area.width = width;
        area.height = height;
        for (uint32_t f = 0; f < frames; ++f)
        {
            // App::drawFrame: the scene's transforms every frame, then the acceleration structures, then the pass
            // (App.cpp:516-578).  Nothing moves here: the update is a no-op, as it is on most of prosper's frames.
            world.updateScene(pass.pass().context(), transforms, 3);
            world.updateBuffers(pass.pass().context());
            camera.updateBuffer();
            (void)pass.record(nullptr, world, camera, area, options, f & 1u, 1, PROSPER_PT_RENDER_PIPELINED);
            camera.endFrame();
        }
        std::vector<float> image((size_t)width * height * 4u);
        if (prosper_pt_read_gathered(pass.pass().context(), image.data(), image.size() * sizeof(float), nullptr) != PROSPER_PT_OK)
            throw std::runtime_error(prosper_pt_last_error());
        FILE *out = std::fopen(argv[4], "wb");
        if (!out || std::fwrite(image.data(), sizeof(float), image.size(), out) != image.size())
            throw std::runtime_error("cannot write the output file");
        std::fclose(out);
        double mean = 0.0;
        for (size_t i = 0; i < image.size(); i += 4) mean += image[i] + image[i + 1] + image[i + 2];
        std::printf("prosper_headless: %ux%u, %u frames, mean radiance %.6f, samples per pixel %.0f\n", width, height, frames,
                    mean / (3.0 * width * height), image[3]);
    }
    catch (const std::exception &e)
    {
        std::fprintf(stderr, "prosper_headless: %s\n", e.what());
        return 1;
    }
    return 0;
}
