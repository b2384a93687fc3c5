// host/tone_map.hpp — render::ToneMap of the headless host layer.
//
// Same four-method surface as prosper's pass (reference: src/render/ToneMap.hpp:16-52): `init` loads the
// Tony McMapface LUT (ToneMap.cpp:33-44: res/texture/tony_mc_mapface.dds, a DX10-header DDS holding a 48^3
// R9G9B9E5 volume) and hands it to prosper_pt_set_tone_map_lut instead of creating a Vulkan 3-D texture;
// `record` runs prosper_pt_tone_map where the original dispatches tone_map.comp (ToneMap.cpp:62-128); the two
// ImGui sliders of drawUi (ToneMap.cpp:55-60) become plain members.
#pragma once

#include <cstdint>
#include <string>
#include <vector>

#include "../../../include/prosper_pt/prosper_pt.h"

namespace render
{

// DX10-header DDS with a single-mip 3-D R9G9B9E5_SHAREDEXP texture (what src/utils/Dds.cpp accepts for the
// LUT) -> depth*height*width texels, x fastest.  Throws std::runtime_error on anything else.
std::vector<uint32_t> readLutDds(const std::string &path, uint32_t &dim);

class ToneMap
{
  public:
    ToneMap() noexcept = default;
    ToneMap(const ToneMap &) = delete;
    ToneMap &operator=(const ToneMap &) = delete;

    // `ctx` is the context whose HDR image gets tone mapped (RtReference::context()).
    void init(prosper_pt_ctx *ctx, const std::string &lutPath);
    void init(prosper_pt_ctx *ctx, const uint32_t *lutR9G9B9E5, uint32_t dim);
    void recompileShaders() {} // kernels are compiled ahead of time
    void drawUi(float exposure, float contrast);

    struct Output
    {
        void *toneMapped = nullptr; // device pointer, RGBA8 UNORM, the extent of the HDR image
    };
    // Tone maps the context's current HDR image into `deviceRgba8` (caller-owned device memory of
    // localWidth * height * 4 bytes).  Throws std::runtime_error on failure.
    [[nodiscard]] Output record(void *stream, void *deviceRgba8, size_t byteSize);

    [[nodiscard]] float exposure() const { return m_exposure; }
    [[nodiscard]] float contrast() const { return m_contrast; }

  private:
    prosper_pt_ctx *m_ctx = nullptr;
    bool m_initialized = false;
    float m_exposure = 1.0f; // ToneMapPC defaults (shader_structs/push_constants/tone_map.h)
    float m_contrast = 1.0f;
};

} // namespace render
