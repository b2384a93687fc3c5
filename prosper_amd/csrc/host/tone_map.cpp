// host/tone_map.cpp — see tone_map.hpp.
#include "tone_map.hpp"

#include <cstdio>
#include <cstring>
#include <new>
#include <stdexcept>

namespace render
{

std::vector<uint32_t> readLutDds(const std::string &path, uint32_t &dim)
{
    FILE *f = std::fopen(path.c_str(), "rb");
    if (!f) throw std::runtime_error("cannot open '" + path + "'");
    uint8_t head[148];
    const size_t got = std::fread(head, 1, sizeof(head), f);
    auto u32 = [&](size_t off) {
        uint32_t v;
        std::memcpy(&v, head + off, 4);
        return v;
    };
    // 'DDS ', 124-byte header, pixel format four-cc 'DX10', DXGI_FORMAT_R9G9B9E5_SHAREDEXP (67), TEXTURE3D (4)
    if (got != sizeof(head) || std::memcmp(head, "DDS ", 4) != 0 || u32(4) != 124 || u32(84) != 0x30315844u ||
        u32(128) != 67u || u32(132) != 4u || u32(28) > 1u)
    {
        std::fclose(f);
        throw std::runtime_error("'" + path + "' is not a single-mip 3-D R9G9B9E5 DX10 DDS");
    }
    const uint32_t height = u32(12), width = u32(16), depth = u32(24);
    if (width == 0 || width != height || width != depth || width > 256)
    {
        std::fclose(f);
        throw std::runtime_error("'" + path + "': the LUT must be a cube");
    }
    std::vector<uint32_t> texels((size_t)width * height * depth);
    const size_t n = std::fread(texels.data(), 4, texels.size(), f);
    std::fclose(f);
    if (n != texels.size()) throw std::runtime_error("'" + path + "': truncated payload");
    dim = width;
    return texels;
}

void ToneMap::init(prosper_pt_ctx *ctx, const std::string &lutPath)
{
    uint32_t dim = 0;
    const std::vector<uint32_t> lut = readLutDds(lutPath, dim);
    init(ctx, lut.data(), dim);
}

void ToneMap::init(prosper_pt_ctx *ctx, const uint32_t *lut, uint32_t dim)
{
    if (!ctx) throw std::runtime_error("ToneMap::init: null context");
    if (prosper_pt_set_tone_map_lut(ctx, lut, dim) != PROSPER_PT_OK) throw std::runtime_error(prosper_pt_last_error());
    m_ctx = ctx;
    m_initialized = true;
}

void ToneMap::drawUi(float exposure, float contrast)
{
    // the sliders clamp to [0.001, 10000] (ToneMap.cpp:58-59)
    auto clamp = [](float v) { return v < 0.001f ? 0.001f : (v > 10000.0f ? 10000.0f : v); };
    m_exposure = clamp(exposure);
    m_contrast = clamp(contrast);
}

ToneMap::Output ToneMap::record(void *stream, void *deviceRgba8, size_t byteSize)
{
    if (!m_initialized) throw std::runtime_error("ToneMap::record before init");
    if (prosper_pt_tone_map(m_ctx, m_exposure, m_contrast, deviceRgba8, nullptr, byteSize, stream) != PROSPER_PT_OK)
        throw std::runtime_error(prosper_pt_last_error());
    Output out;
    out.toneMapped = deviceRgba8;
    return out;
}

} // namespace render

// ---- plain-C shims (include/prosper_pt/prosper_host.h) ----
#include "../../../include/prosper_pt/prosper_host.h"

struct prosper_host_tone_map
{
    render::ToneMap pass;
};

extern "C" void prosper_host_set_error(const char *message); // rt_reference.cpp

extern "C" int prosper_host_tone_map_create(prosper_pt_ctx *ctx, const char *lutDdsPath, prosper_host_tone_map **out)
{
    *out = nullptr;
    prosper_host_tone_map *t = new (std::nothrow) prosper_host_tone_map();
    if (!t) return PROSPER_PT_ERR_INVALID_ARGUMENT;
    try
    {
        t->pass.init(ctx, std::string(lutDdsPath ? lutDdsPath : ""));
    }
    catch (const std::exception &e)
    {
        prosper_host_set_error(e.what());
        delete t;
        return PROSPER_PT_ERR_INVALID_ARGUMENT;
    }
    *out = t;
    return PROSPER_PT_OK;
}

extern "C" int prosper_host_tone_map_create_from_texels(
    prosper_pt_ctx *ctx, const uint32_t *lut, uint32_t dim, prosper_host_tone_map **out)
{
    *out = nullptr;
    prosper_host_tone_map *t = new (std::nothrow) prosper_host_tone_map();
    if (!t) return PROSPER_PT_ERR_INVALID_ARGUMENT;
    try
    {
        t->pass.init(ctx, lut, dim);
    }
    catch (const std::exception &e)
    {
        prosper_host_set_error(e.what());
        delete t;
        return PROSPER_PT_ERR_INVALID_ARGUMENT;
    }
    *out = t;
    return PROSPER_PT_OK;
}

extern "C" void prosper_host_tone_map_destroy(prosper_host_tone_map *t) { delete t; }

extern "C" void prosper_host_tone_map_draw_ui(prosper_host_tone_map *t, float exposure, float contrast)
{
    t->pass.drawUi(exposure, contrast);
}

extern "C" int prosper_host_tone_map_record(prosper_host_tone_map *t, void *stream, void *deviceRgba8, size_t byteSize)
{
    try
    {
        (void)t->pass.record(stream, deviceRgba8, byteSize);
    }
    catch (const std::exception &e)
    {
        prosper_host_set_error(e.what());
        return PROSPER_PT_ERR_INVALID_ARGUMENT;
    }
    return PROSPER_PT_OK;
}
