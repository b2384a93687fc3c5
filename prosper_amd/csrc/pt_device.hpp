// pt_device.hpp — device functions of the HIP path tracer (gfx950), restating prosper's RT
// reference shaders.  Citations are file:line under /root/reference/res/shader/ unless noted.
#pragma once

#include "pt_math.hpp"
#include "pt_scene.hpp"

namespace ppt
{

#define PPT_D __device__ __forceinline__
// 64-byte raw shading records (pt_scene.hpp RawShadeTriangle): an experiment that lost (profiles/r03_raw_records.txt),
// compiled in only with -DPPT_EXPERIMENTS
#ifdef PPT_EXPERIMENTS
#define PPT_RAW_RECORDS(s) ((s).rawShadeTriangles != nullptr)
#else
#define PPT_RAW_RECORDS(s) false
#endif

constexpr uint32_t kMissIndex = 0xFFFFFFFFu; // rt/reference/main.rgen:47

// Work counters of one lane (SURVEY §8d byte model).  Only the COUNT=true kernel variants touch
// them; the timed variants compile every increment away.
struct LaneCounters
{
    uint32_t closestRays, shadowRays, nodeVisits, triangleTests, closestHits, anyHitCalls, lightSamples,
        spotLightSamples, skyLookups, shortIndexHits, paths, pixelsWritten, historyReads, shortIndexTriangleTests,
        nodePhaseSteps, trianglePhaseSteps, anyHitTexelFetches;
};

// ------------------------------------------------------------------------------------------
// F8 RNG — common/random.glsl
// ------------------------------------------------------------------------------------------

// random.glsl:7-12
PPT_D uint32_t pcg(uint32_t v)
{
    const uint32_t state = v * 747796405u + 2891336453u;
    const uint32_t word = ((state >> ((state >> 28) + 4u)) ^ state) * 277803737u;
    return (word >> 22) ^ word;
}

struct Rng
{
    uint32_t x, y, z;
    // random.glsl:17-28
    PPT_D void step()
    {
        x = x * 1664525u + 1013904223u;
        y = y * 1664525u + 1013904223u;
        z = z * 1664525u + 1013904223u;
        x += y * z;
        y += z * x;
        z += x * y;
        x ^= x >> 16;
        y ^= y >> 16;
        z ^= z >> 16;
        x += y * z;
        y += z * x;
        z += x * y;
    }
    // random.glsl:42: float(0xFFFFFFFFu) rounds to 2^32
    static PPT_D float to01(uint32_t u) { return (float)u / 4294967296.0f; }
    // random.glsl:50-63
    PPT_D float rnd01()
    {
        step();
        return to01(x);
    }
    PPT_D f2 rnd2d01()
    {
        step();
        return f2{to01(x), to01(y)};
    }
};

// random.glsl:30-40
PPT_D f3 uint_to_color(uint32_t v)
{
    const uint32_t xr = pcg(v);
    const float k = 1.0f / 1023.0f;
    return f3{(float)((xr >> 20) & 0x3FFu) * k, (float)((xr >> 10) & 0x3FFu) * k, (float)(xr & 0x3FFu) * k};
}

// ------------------------------------------------------------------------------------------
// F10 bindless fetch + decode — scene/geometry.glsl
// ------------------------------------------------------------------------------------------

struct Vertex
{
    f3 position;
    f3 normal;
    f4 tangent;
    f2 uv;
};

// Pointers that are themselves loaded from memory (geometry buffers, texel arrays) are typed as
// global-address-space pointers: the compiler cannot prove it and would emit flat_load for them, which
// occupies the LDS path too and waits on both counters.
typedef const __attribute__((address_space(1))) uint32_t *global_u32_ptr;
typedef const __attribute__((address_space(1))) uint16_t *global_u16_ptr;
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef const __attribute__((address_space(1))) u32x2 *global_u32x2_ptr;

PPT_D global_u32_ptr geo_u32(const DeviceScene &s, uint32_t buffer)
{
    return (global_u32_ptr)s.geometryBuffers[buffer];
}

// geometry.glsl:51-59
PPT_D uint32_t load_index(const DeviceScene &s, const prosper_GeometryMetadata &m, uint32_t index)
{
    if (m.usesShortIndices == 1)
        return (uint32_t)((global_u16_ptr)s.geometryBuffers[m.bufferIndex])[m.indicesOffset + index];
    return geo_u32(s, m.bufferIndex)[m.indicesOffset + index];
}

PPT_D f2 unpack_half2(uint32_t packed) { return f2{half_to_float(packed & 0xFFFFu), half_to_float(packed >> 16)}; }
PPT_D f3 unpack_half3(uint32_t xy, uint32_t z_)
{
    return f3{half_to_float(xy & 0xFFFFu), half_to_float(xy >> 16), half_to_float(z_ & 0xFFFFu)};
}

// geometry.glsl:71-80
PPT_D f2 load_r16g16(const DeviceScene &s, uint32_t buffer, uint32_t offset, uint32_t index)
{
    if (offset == PROSPER_PT_ABSENT) return f2{0.0f, 0.0f};
    const uint32_t packed = geo_u32(s, buffer)[offset + index];
    return f2{half_to_float(packed & 0xFFFFu), half_to_float(packed >> 16)};
}

// geometry.glsl:82-93
PPT_D f3 load_r16g16b16a16(const DeviceScene &s, uint32_t buffer, uint32_t offset, uint32_t index)
{
    if (offset == PROSPER_PT_ABSENT) return f3{0.0f, 0.0f, 0.0f};
    const u32x2 p = *(global_u32x2_ptr)(geo_u32(s, buffer) + offset + index * 2);
    return f3{half_to_float(p.x & 0xFFFFu), half_to_float(p.x >> 16), half_to_float(p.y & 0xFFFFu)};
}

// geometry.glsl:95-103
PPT_D f3 unpack_snorm_r10g10b10(uint32_t packed)
{
    const int32_t sx = (int32_t)(packed << 22) >> 22;
    const int32_t sy = (int32_t)(packed << 12) >> 22;
    const int32_t sz = (int32_t)(packed << 2) >> 22;
    const float k = 1.0f / 511.0f;
    return normalize(f3{fmax_((float)sx * k, -1.0f), fmax_((float)sy * k, -1.0f), fmax_((float)sz * k, -1.0f)});
}

// geometry.glsl:220-244
PPT_D Vertex load_vertex_through_index_buffer(const DeviceScene &s, const prosper_GeometryMetadata &m, uint32_t index)
{
    const uint32_t vi = load_index(s, m, index);
    Vertex v;
    v.position = load_r16g16b16a16(s, m.bufferIndex, m.positionsOffset, vi);
    // geometry.glsl:105-114
    v.normal = m.normalsOffset == PROSPER_PT_ABSENT
                   ? f3{0.0f, 0.0f, 0.0f}
                   : unpack_snorm_r10g10b10(geo_u32(s, m.bufferIndex)[m.normalsOffset + vi]);
    // geometry.glsl:116-127
    if (m.tangentsOffset == PROSPER_PT_ABSENT)
        v.tangent = f4{0.0f, 0.0f, 0.0f, 0.0f};
    else
    {
        const uint32_t packed = geo_u32(s, m.bufferIndex)[m.tangentsOffset + vi];
        const f3 t = unpack_snorm_r10g10b10(packed);
        v.tangent = f4{t.x, t.y, t.z, (float)((int32_t)packed >> 30)};
    }
    v.uv = load_r16g16(s, m.bufferIndex, m.texCoord0sOffset, vi);
    return v;
}

// geometry.glsl:258-270
PPT_D float bary1(float v0, float v1, float v2, float a, float b, float c)
{
    return __builtin_fmaf(v2, c, __builtin_fmaf(v1, b, v0 * a));
}

// geometry.glsl:271-286
PPT_D Vertex interpolate(const Vertex &v0, const Vertex &v1, const Vertex &v2, f2 bc)
{
    const float a = (1.0f - bc.x) - bc.y;
    const float b = bc.x;
    const float c = bc.y;
    Vertex r;
    r.position = f3{bary1(v0.position.x, v1.position.x, v2.position.x, a, b, c),
                    bary1(v0.position.y, v1.position.y, v2.position.y, a, b, c),
                    bary1(v0.position.z, v1.position.z, v2.position.z, a, b, c)};
    r.normal = f3{bary1(v0.normal.x, v1.normal.x, v2.normal.x, a, b, c),
                  bary1(v0.normal.y, v1.normal.y, v2.normal.y, a, b, c),
                  bary1(v0.normal.z, v1.normal.z, v2.normal.z, a, b, c)};
    r.tangent = f4{bary1(v0.tangent.x, v1.tangent.x, v2.tangent.x, a, b, c),
                   bary1(v0.tangent.y, v1.tangent.y, v2.tangent.y, a, b, c),
                   bary1(v0.tangent.z, v1.tangent.z, v2.tangent.z, a, b, c),
                   bary1(v0.tangent.w, v1.tangent.w, v2.tangent.w, a, b, c)};
    r.uv = f2{bary1(v0.uv.x, v1.uv.x, v2.uv.x, a, b, c), bary1(v0.uv.y, v1.uv.y, v2.uv.y, a, b, c)};
    return r;
}

// ------------------------------------------------------------------------------------------
// F11 object -> world — scene/instances.glsl:36-53
// ------------------------------------------------------------------------------------------

// vec4(p,1) * mat3x4
PPT_D f3 mul_point_mat3x4(f3 p, const prosper_mat3x4 &m)
{
    return f3{__builtin_fmaf(p.z, m.col[0].z, __builtin_fmaf(p.y, m.col[0].y, __builtin_fmaf(p.x, m.col[0].x, m.col[0].w))),
              __builtin_fmaf(p.z, m.col[1].z, __builtin_fmaf(p.y, m.col[1].y, __builtin_fmaf(p.x, m.col[1].x, m.col[1].w))),
              __builtin_fmaf(p.z, m.col[2].z, __builtin_fmaf(p.y, m.col[2].y, __builtin_fmaf(p.x, m.col[2].x, m.col[2].w)))};
}
// v * mat3(m)
PPT_D f3 mul_vec_mat3(f3 v, const prosper_mat3x4 &m)
{
    return f3{__builtin_fmaf(v.z, m.col[0].z, __builtin_fmaf(v.y, m.col[0].y, v.x * m.col[0].x)),
              __builtin_fmaf(v.z, m.col[1].z, __builtin_fmaf(v.y, m.col[1].y, v.x * m.col[1].x)),
              __builtin_fmaf(v.z, m.col[2].z, __builtin_fmaf(v.y, m.col[2].y, v.x * m.col[2].x))};
}

PPT_D Vertex transform(const Vertex &v, const prosper_ModelInstanceTransforms &t)
{
    Vertex r;
    r.position = mul_point_mat3x4(v.position, t.modelToWorld);
    r.normal = normalize(mul_vec_mat3(v.normal, t.normalToWorld));
    if (v.tangent.w != 0.0f)
    {
        const f3 tt = normalize(mul_vec_mat3(f3{v.tangent.x, v.tangent.y, v.tangent.z}, t.modelToWorld));
        r.tangent = f4{tt.x, tt.y, tt.z, v.tangent.w};
    }
    else
        r.tangent = v.tangent;
    r.uv = v.uv;
    return r;
}

// ------------------------------------------------------------------------------------------
// Texture sampling (LOD 0; DESIGN.md "Texture contract")
// ------------------------------------------------------------------------------------------

// i mod period in [0, period); power-of-two periods (the usual texture sizes) skip the division
PPT_D int32_t floor_mod(int32_t i, int32_t period)
{
    if ((period & (period - 1)) == 0) return i & (period - 1);
    const int32_t m = i % period;
    return m < 0 ? m + period : m;
}

PPT_D int32_t wrap_coord(int32_t i, int32_t size, uint32_t mode)
{
    if (mode == PROSPER_PT_WRAP_REPEAT) return floor_mod(i, size);
    if (mode == PROSPER_PT_WRAP_MIRRORED_REPEAT)
    {
        const int32_t period = 2 * size;
        const int32_t m = floor_mod(i, period);
        return m < size ? m : period - 1 - m;
    }
    return i < 0 ? 0 : (i >= size ? size - 1 : i);
}

// texel indices of integer coordinates i and i+1: the second is the successor of the first inside
// the wrap period, which is (i+1) mod period without a second division (and without overflow)
PPT_D void wrap_pair(int32_t i, int32_t size, uint32_t mode, int32_t &c0, int32_t &c1)
{
    if (mode == PROSPER_PT_WRAP_REPEAT || mode == PROSPER_PT_WRAP_MIRRORED_REPEAT)
    {
        const int32_t period = mode == PROSPER_PT_WRAP_REPEAT ? size : 2 * size;
        const int32_t m = floor_mod(i, period);
        const int32_t n = m + 1 == period ? 0 : m + 1;
        c0 = m < size ? m : period - 1 - m;
        c1 = n < size ? n : period - 1 - n;
        return;
    }
    c0 = i < 0 ? 0 : (i >= size ? size - 1 : i);
    c1 = i < -1 ? 0 : (i >= size - 1 ? size - 1 : i + 1);
}

// word offset of texel (i, j) in the tiled image
PPT_D uint32_t texel_offset(const DeviceTexture &t, int32_t i, int32_t j)
{
    const uint32_t tile = ((uint32_t)j >> 2) * t.tilesPerRow + ((uint32_t)i >> 3);
    return tile * 32u + ((((uint32_t)j & 3u) << 3) | ((uint32_t)i & 7u));
}
PPT_D f4 unpack_rgba8(uint32_t p)
{
    const float k = 1.0f / 255.0f;
    return f4{(float)(p & 0xFFu) * k, (float)((p >> 8) & 0xFFu) * k, (float)((p >> 16) & 0xFFu) * k, (float)(p >> 24) * k};
}
PPT_D f4 fetch_rgba8(const DeviceTexture &t, int32_t i, int32_t j)
{
    return unpack_rgba8(((global_u32_ptr)t.texels)[texel_offset(t, i, j)]);
}

// One LOD-0 sample in three steps, so that a caller with several textures can compute all coordinates, then have
// all texel loads in flight together, then filter (sample_material: three dependent round trips become one):
//   texel_taps  - wrap + filter footprint: four texel offsets and the bilinear weights (no memory access);
//                 a nearest-filter sample is the footprint (i, j) x 4 with weights (1, 0, 0, 0) - the weighted
//                 sum below then returns that texel's value exactly (texels are finite, 0 * t = 0, x + 0 = x)
//   fetch_taps  - the four loads
//   filter_taps - unpack + weighted sum, the same fma chain as ever
struct TexelTaps
{
    global_u32_ptr texels;
    uint32_t o00, o10, o01, o11;
    float a, b;
};
PPT_D TexelTaps texel_taps(const DeviceTexture &t, const prosper_pt_sampler_desc &sd, f2 uv)
{
    const int32_t w = (int32_t)t.width;
    const int32_t h = (int32_t)t.height;
    TexelTaps k;
    k.texels = (global_u32_ptr)t.texels;
    if (sd.magFilter == PROSPER_PT_FILTER_NEAREST)
    {
        const int32_t i = wrap_coord(f2i(__builtin_floorf(uv.x * (float)w)), w, sd.wrapS);
        const int32_t j = wrap_coord(f2i(__builtin_floorf(uv.y * (float)h)), h, sd.wrapT);
        k.o00 = k.o10 = k.o01 = k.o11 = texel_offset(t, i, j);
        k.a = 0.0f;
        k.b = 0.0f;
        return k;
    }
    const float u = __builtin_fmaf(uv.x, (float)w, -0.5f);
    const float v = __builtin_fmaf(uv.y, (float)h, -0.5f);
    const float fu = __builtin_floorf(u);
    const float fv = __builtin_floorf(v);
    k.a = u - fu;
    k.b = v - fv;
    int32_t i0, i1, j0, j1;
    wrap_pair(f2i(fu), w, sd.wrapS, i0, i1);
    wrap_pair(f2i(fv), h, sd.wrapT, j0, j1);
    k.o00 = texel_offset(t, i0, j0);
    k.o10 = texel_offset(t, i1, j0);
    k.o01 = texel_offset(t, i0, j1);
    k.o11 = texel_offset(t, i1, j1);
    return k;
}
// The same footprint in a MaterialPack (4 x 2-texel tiles of uint4 texels): offsets in uint4 units.
struct PackTaps
{
    uint32_t o00, o10, o01, o11;
    float a, b;
};
PPT_D uint32_t pack_texel_offset(const MaterialPack &t, int32_t i, int32_t j)
{
    const uint32_t tile = ((uint32_t)j >> 1) * t.tilesPerRow + ((uint32_t)i >> 2);
    return tile * 8u + ((((uint32_t)j & 1u) << 2) | ((uint32_t)i & 3u));
}
PPT_D PackTaps pack_taps(const MaterialPack &t, const prosper_pt_sampler_desc &sd, f2 uv)
{
    const int32_t w = (int32_t)t.width;
    const int32_t h = (int32_t)t.height;
    PackTaps k;
    if (sd.magFilter == PROSPER_PT_FILTER_NEAREST)
    {
        const int32_t i = wrap_coord(f2i(__builtin_floorf(uv.x * (float)w)), w, sd.wrapS);
        const int32_t j = wrap_coord(f2i(__builtin_floorf(uv.y * (float)h)), h, sd.wrapT);
        k.o00 = k.o10 = k.o01 = k.o11 = pack_texel_offset(t, i, j);
        k.a = 0.0f;
        k.b = 0.0f;
        return k;
    }
    const float u = __builtin_fmaf(uv.x, (float)w, -0.5f);
    const float v = __builtin_fmaf(uv.y, (float)h, -0.5f);
    const float fu = __builtin_floorf(u);
    const float fv = __builtin_floorf(v);
    k.a = u - fu;
    k.b = v - fv;
    int32_t i0, i1, j0, j1;
    wrap_pair(f2i(fu), w, sd.wrapS, i0, i1);
    wrap_pair(f2i(fv), h, sd.wrapT, j0, j1);
    k.o00 = pack_texel_offset(t, i0, j0);
    k.o10 = pack_texel_offset(t, i1, j0);
    k.o01 = pack_texel_offset(t, i0, j1);
    k.o11 = pack_texel_offset(t, i1, j1);
    return k;
}

struct RawTaps
{
    uint32_t p00, p10, p01, p11;
};
PPT_D RawTaps fetch_taps(const TexelTaps &k)
{
    return RawTaps{k.texels[k.o00], k.texels[k.o10], k.texels[k.o01], k.texels[k.o11]};
}
PPT_D f4 filter_taps(const TexelTaps &k, const RawTaps &r)
{
    const f4 t00 = unpack_rgba8(r.p00);
    const f4 t10 = unpack_rgba8(r.p10);
    const f4 t01 = unpack_rgba8(r.p01);
    const f4 t11 = unpack_rgba8(r.p11);
    const float a = k.a, b = k.b;
    const float w00 = (1.0f - a) * (1.0f - b);
    const float w10 = a * (1.0f - b);
    const float w01 = (1.0f - a) * b;
    const float w11 = a * b;
    return f4{__builtin_fmaf(w11, t11.x, __builtin_fmaf(w01, t01.x, __builtin_fmaf(w10, t10.x, w00 * t00.x))),
              __builtin_fmaf(w11, t11.y, __builtin_fmaf(w01, t01.y, __builtin_fmaf(w10, t10.y, w00 * t00.y))),
              __builtin_fmaf(w11, t11.z, __builtin_fmaf(w01, t01.z, __builtin_fmaf(w10, t10.z, w00 * t00.z))),
              __builtin_fmaf(w11, t11.w, __builtin_fmaf(w01, t01.w, __builtin_fmaf(w10, t10.w, w00 * t00.w)))};
}

PPT_D f4 sample_texture(const DeviceScene &s, uint32_t tex, uint32_t smp, f2 uv)
{
    const DeviceTexture t = s.textures[tex];
    const prosper_pt_sampler_desc sd = s.samplers[smp];
    if (sd.magFilter == PROSPER_PT_FILTER_NEAREST)
    {
        const int32_t w = (int32_t)t.width;
        const int32_t h = (int32_t)t.height;
        const int32_t i = wrap_coord(f2i(__builtin_floorf(uv.x * (float)w)), w, sd.wrapS);
        const int32_t j = wrap_coord(f2i(__builtin_floorf(uv.y * (float)h)), h, sd.wrapT);
        return fetch_rgba8(t, i, j);
    }
    const TexelTaps k = texel_taps(t, sd, uv);
    return filter_taps(k, fetch_taps(k));
}

// Cube face selection per Vulkan 1.3 §16.5.4 (faces +X,-X,+Y,-Y,+Z,-Z)
PPT_D void cube_face_coords(f3 d, uint32_t &face, float &sc, float &tc, float &ma)
{
    const float ax = fabs_(d.x), ay = fabs_(d.y), az = fabs_(d.z);
    if (az >= ax && az >= ay)
    {
        face = d.z < 0.0f ? 5u : 4u;
        sc = d.z < 0.0f ? -d.x : d.x;
        tc = -d.y;
        ma = az;
    }
    else if (ay >= ax)
    {
        face = d.y < 0.0f ? 3u : 2u;
        sc = d.x;
        tc = d.y < 0.0f ? -d.z : d.z;
        ma = ay;
    }
    else
    {
        face = d.x < 0.0f ? 1u : 0u;
        sc = d.x < 0.0f ? d.z : -d.z;
        tc = -d.y;
        ma = ax;
    }
}

PPT_D f3 cube_face_dir(uint32_t face, float sc, float tc)
{
    switch (face)
    {
    case 0: return f3{1.0f, -tc, -sc};
    case 1: return f3{-1.0f, -tc, sc};
    case 2: return f3{sc, 1.0f, tc};
    case 3: return f3{sc, -1.0f, -tc};
    case 4: return f3{sc, -tc, 1.0f};
    default: return f3{-sc, -tc, -1.0f};
    }
}

// Texel (i, j) of a face of the cube as uploaded (6 x n x n RGBA16F), i and j in [-1, n]: outside the face it is the
// texel the seamless-edge rule finds on the neighbouring face.  Runs once per border texel at upload
// (border_skybox_kernel); the path's own lookups read the bordered copy.
PPT_D uint2 cube_texel_seamless(const uint16_t *skybox, int32_t n, uint32_t face, int32_t i, int32_t j)
{
    if (i < 0 || j < 0 || i >= n || j >= n)
    {
        // seamless edge: re-project the centre of the out-of-face texel onto the neighbouring face
        const float invN = 1.0f / (float)n;
        const float sc = __builtin_fmaf(2.0f * ((float)i + 0.5f), invN, -1.0f);
        const float tc = __builtin_fmaf(2.0f * ((float)j + 0.5f), invN, -1.0f);
        const f3 d = cube_face_dir(face, sc, tc);
        float sc2, tc2, ma2;
        cube_face_coords(d, face, sc2, tc2, ma2);
        const float inv2 = 1.0f / ma2;
        const float ss = __builtin_fmaf(0.5f, sc2 * inv2, 0.5f);
        const float tt = __builtin_fmaf(0.5f, tc2 * inv2, 0.5f);
        i = f2i(__builtin_floorf(ss * (float)n));
        j = f2i(__builtin_floorf(tt * (float)n));
        i = i < 0 ? 0 : (i >= n ? n - 1 : i);
        j = j < 0 ? 0 : (j >= n ? n - 1 : j);
    }
    return *reinterpret_cast<const uint2 *>(skybox + 4u * (((size_t)face * n + (size_t)j) * n + (size_t)i));
}

// DeviceScene::skybox holds every face with a one-texel BORDER, (n + 2) x (n + 2): the border texels are the ones
// cube_texel_seamless finds for the out-of-face taps of a bilinear footprint, so a lookup is four plain loads - no
// re-projection branch, which one lane in 250 needs and two waves in five had to walk through.
PPT_D f3 fetch_cube_rgb(const DeviceScene &s, uint32_t face, int32_t i, int32_t j)
{
    const uint32_t n2 = s.skyboxFaceSize + 2u;
    const uint2 p = *reinterpret_cast<const uint2 *>(s.skybox + 4u * (((size_t)face * n2 + (size_t)(j + 1)) * n2 + (size_t)(i + 1)));
    return f3{half_to_float(p.x & 0xFFFFu), half_to_float(p.x >> 16), half_to_float(p.y & 0xFFFFu)};
}

// F18: textureLod(skybox, d, 0).rgb — rt/reference/main.rgen:251
PPT_D f3 sample_skybox(const DeviceScene &s, f3 d)
{
    if (s.skybox == nullptr) return f3{0.0f, 0.0f, 0.0f};
    const int32_t n = (int32_t)s.skyboxFaceSize;
    uint32_t face;
    float sc, tc, ma;
    cube_face_coords(d, face, sc, tc, ma);
    const float invMa = 1.0f / ma;
    const float ss = __builtin_fmaf(0.5f, sc * invMa, 0.5f);
    const float tt = __builtin_fmaf(0.5f, tc * invMa, 0.5f);
    const float u = __builtin_fmaf(ss, (float)n, -0.5f);
    const float v = __builtin_fmaf(tt, (float)n, -0.5f);
    const float fu = __builtin_floorf(u);
    const float fv = __builtin_floorf(v);
    const float a = u - fu;
    const float b = v - fv;
    // ss, tt in [0, 1] put the footprint's first texel in [-1, n - 1]: the clamp changes nothing for a direction with a
    // finite, non-zero largest component and keeps the loads of any other inside the bordered face
    int32_t i0 = f2i(fu);
    int32_t j0 = f2i(fv);
    i0 = i0 < -1 ? -1 : (i0 > n - 1 ? n - 1 : i0);
    j0 = j0 < -1 ? -1 : (j0 > n - 1 ? n - 1 : j0);
    const f3 t00 = fetch_cube_rgb(s, face, i0, j0);
    const f3 t10 = fetch_cube_rgb(s, face, i0 + 1, j0);
    const f3 t01 = fetch_cube_rgb(s, face, i0, j0 + 1);
    const f3 t11 = fetch_cube_rgb(s, face, i0 + 1, j0 + 1);
    const float w00 = (1.0f - a) * (1.0f - b);
    const float w10 = a * (1.0f - b);
    const float w01 = (1.0f - a) * b;
    const float w11 = a * b;
    return f3{__builtin_fmaf(w11, t11.x, __builtin_fmaf(w01, t01.x, __builtin_fmaf(w10, t10.x, w00 * t00.x))),
              __builtin_fmaf(w11, t11.y, __builtin_fmaf(w01, t01.y, __builtin_fmaf(w10, t10.y, w00 * t00.y))),
              __builtin_fmaf(w11, t11.z, __builtin_fmaf(w01, t01.z, __builtin_fmaf(w10, t10.z, w00 * t00.z)))};
}

// ------------------------------------------------------------------------------------------
// F12 materials — scene/materials.glsl
// ------------------------------------------------------------------------------------------

struct Material
{
    f3 albedo;
    f3 normal;
    float roughness;
    float metallic;
    float alpha;
};

// materials.glsl:26-29
PPT_D float srgb_to_linear(float x)
{
    return x <= 0.04045f ? x * (1.0f / 12.92f) : pow_((x + 0.055f) * (1.0f / 1.055f), 2.4f);
}

// materials.glsl:47-119
// BATCHED: all texel loads of the three textures in flight together (costs ~13 more VGPRs: the shade kernel then
// sits at its 128-register limit with a 16-byte spill, which a scene without big textures does not get back)
template <bool BATCHED>
PPT_D Material sample_material(const DeviceScene &s, uint32_t index, f2 uv)
{
    const prosper_MaterialData data = s.materials[index];
    Material ret;
    ret.albedo = f3{0.0f, 0.0f, 0.0f};
    ret.normal = f3{0.0f, 0.0f, 0.0f};
    ret.roughness = 0.0f;
    ret.metallic = 0.0f;

    // The three texture samples do not depend on one another: descriptors first, then all footprints, then all
    // twelve texel loads in flight together, then the filtering - one memory round trip per stage instead of one
    // per texture (the texels of a 315 MB texture set mostly come from HBM).  On a failed alpha mask the
    // reference returns before the other two samples; their values are simply not used then.
    const uint32_t baseTex = data.baseColorTextureSampler & 0xFFFFFFu;
    const uint32_t mrTex = data.metallicRoughnessTextureSampler & 0xFFFFFFu;
    const uint32_t nTex = data.normalTextureSampler & 0xFFFFFFu;
    f4 sBaseT = f4{1.0f, 1.0f, 1.0f, 1.0f}, sMrT = f4{0.0f, 0.0f, 0.0f, 0.0f}, sNT = f4{0.0f, 0.0f, 0.0f, 0.0f};
    const MaterialPack pack = s.materialPacks[index];
    if (pack.texels != nullptr)
    {
        // the three textures interleaved per texel (pt_scene.hpp MaterialPack): one footprint, four 12-byte loads - or
        // four 8-byte ones from the compact pack of an opaque material, which keeps the eight bytes this function reads
        typedef uint32_t u32x3 __attribute__((ext_vector_type(3)));
        typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
        typedef const __attribute__((address_space(1))) u32x3 *global_u32x3_ptr;
        typedef const __attribute__((address_space(1))) u32x2 *global_u32x2_ptr;
        const PackTaps k = pack_taps(pack, s.samplers[pack.sampler & ~kPackCompactBit], uv);
        const global_u32_ptr base = (global_u32_ptr)pack.texels;
        TexelTaps w;
        w.a = k.a;
        w.b = k.b;
        if (pack.sampler & kPackCompactBit)
        {
            // (the compact pack is the one of texture sets that outgrow the caches: PPT_TEXEL_NT reads it past them)
#if defined(PPT_TEXEL_NT) && PPT_TEXEL_NT
            const u32x2 p00 = __builtin_nontemporal_load((global_u32x2_ptr)(base + k.o00 * 2u));
            const u32x2 p10 = __builtin_nontemporal_load((global_u32x2_ptr)(base + k.o10 * 2u));
            const u32x2 p01 = __builtin_nontemporal_load((global_u32x2_ptr)(base + k.o01 * 2u));
            const u32x2 p11 = __builtin_nontemporal_load((global_u32x2_ptr)(base + k.o11 * 2u));
#else
            const u32x2 p00 = *(global_u32x2_ptr)(base + k.o00 * 2u);
            const u32x2 p10 = *(global_u32x2_ptr)(base + k.o10 * 2u);
            const u32x2 p01 = *(global_u32x2_ptr)(base + k.o01 * 2u);
            const u32x2 p11 = *(global_u32x2_ptr)(base + k.o11 * 2u);
#endif
            const f4 lo = filter_taps(w, RawTaps{p00.x, p10.x, p01.x, p11.x}); // R G B roughness
            const f4 hi = filter_taps(w, RawTaps{p00.y, p10.y, p01.y, p11.y}); // metallic Nx Ny Nz
            sBaseT = f4{lo.x, lo.y, lo.z, 1.0f};
            sMrT = f4{0.0f, lo.w, hi.x, 0.0f};
            sNT = f4{hi.y, hi.z, hi.w, 0.0f};
        }
        else
        {
            const u32x3 p00 = *(global_u32x3_ptr)(base + k.o00 * 4u);
            const u32x3 p10 = *(global_u32x3_ptr)(base + k.o10 * 4u);
            const u32x3 p01 = *(global_u32x3_ptr)(base + k.o01 * 4u);
            const u32x3 p11 = *(global_u32x3_ptr)(base + k.o11 * 4u);
            sBaseT = filter_taps(w, RawTaps{p00.x, p10.x, p01.x, p11.x});
            sMrT = filter_taps(w, RawTaps{p00.y, p10.y, p01.y, p11.y});
            sNT = filter_taps(w, RawTaps{p00.z, p10.z, p01.z, p11.z});
        }
    }
    else if constexpr (!BATCHED)
    {
        // one texture after the other: fewer live registers
        if (baseTex > 0) sBaseT = sample_texture(s, baseTex, data.baseColorTextureSampler >> 24, uv);
        if (mrTex > 0) sMrT = sample_texture(s, mrTex, data.metallicRoughnessTextureSampler >> 24, uv);
        if (nTex > 0) sNT = sample_texture(s, nTex, data.normalTextureSampler >> 24, uv);
    }
    else if ((baseTex | mrTex | nTex) != 0u) // untextured materials skip the block as a whole
    {
        DeviceTexture tBase, tMr, tN;
        prosper_pt_sampler_desc sBase, sMr, sN;
        if (baseTex > 0)
        {
            tBase = s.textures[baseTex];
            sBase = s.samplers[data.baseColorTextureSampler >> 24];
        }
        if (mrTex > 0)
        {
            tMr = s.textures[mrTex];
            sMr = s.samplers[data.metallicRoughnessTextureSampler >> 24];
        }
        if (nTex > 0)
        {
            tN = s.textures[nTex];
            sN = s.samplers[data.normalTextureSampler >> 24];
        }
        TexelTaps kBase, kMr, kN;
        if (baseTex > 0) kBase = texel_taps(tBase, sBase, uv);
        if (mrTex > 0) kMr = texel_taps(tMr, sMr, uv);
        if (nTex > 0) kN = texel_taps(tN, sN, uv);
        RawTaps rBase, rMr, rN;
        if (baseTex > 0) rBase = fetch_taps(kBase);
        if (mrTex > 0) rMr = fetch_taps(kMr);
        if (nTex > 0) rN = fetch_taps(kN);
        if (baseTex > 0) sBaseT = filter_taps(kBase, rBase);
        if (mrTex > 0) sMrT = filter_taps(kMr, rMr);
        if (nTex > 0) sNT = filter_taps(kN, rN);
    }

    f4 base = f4{1.0f, 1.0f, 1.0f, 1.0f};
    if (baseTex > 0) base = f4{srgb_to_linear(sBaseT.x), srgb_to_linear(sBaseT.y), srgb_to_linear(sBaseT.z), sBaseT.w};
    base.x *= data.baseColorFactor.x;
    base.y *= data.baseColorFactor.y;
    base.z *= data.baseColorFactor.z;
    base.w *= data.baseColorFactor.w;

    if (data.alphaMode == PROSPER_ALPHA_MODE_BLEND)
        ret.alpha = base.w;
    else
    {
        if (data.alphaMode == PROSPER_ALPHA_MODE_MASK && base.w < data.alphaCutoff)
        {
            ret.alpha = 0.0f;
            return ret;
        }
        ret.alpha = -1.0f;
    }
    ret.albedo = f3{base.x, base.y, base.z};

    if (mrTex > 0)
    {
        ret.roughness = sMrT.y * data.roughnessFactor;
        ret.metallic = sMrT.z * data.metallicFactor;
    }
    else
    {
        ret.roughness = data.roughnessFactor;
        ret.metallic = data.metallicFactor;
    }
    ret.roughness = fmax_(ret.roughness, 0.05f);

    if (nTex > 0)
        ret.normal = f3{__builtin_fmaf(sNT.x, 2.0f, -1.0f), __builtin_fmaf(sNT.y, 2.0f, -1.0f), __builtin_fmaf(sNT.z, 2.0f, -1.0f)};
    else
        ret.normal = f3{-2.0f, -2.0f, -2.0f};
    return ret;
}

// (sampleAlpha, materials.glsl:121-147, lives with the any-hit shader below: any_hit_settle / any_hit_exact / alpha_verdict)

// ------------------------------------------------------------------------------------------
// F15 BRDF — brdf.glsl
// ------------------------------------------------------------------------------------------

struct Surface
{
    f3 positionWS;
    f3 normalWS;
    f3 invViewRayWS;
    f2 uv;
    float NoV;
    Material material;
};

// brdf.glsl:12-19
PPT_D float trowbridge_reitz(float NoH, float alpha)
{
    const float a2 = alpha * alpha;
    const float denom = __builtin_fmaf(NoH * NoH, a2 - 1.0f, 1.0f);
    return a2 / ((kPi * denom) * denom);
}
// brdf.glsl:21-24
PPT_D f3 schlick_fresnel(float VoH, f3 f0)
{
    const float p = pow5(1.0f - VoH);
    return f3{__builtin_fmaf(1.0f - f0.x, p, f0.x), __builtin_fmaf(1.0f - f0.y, p, f0.y), __builtin_fmaf(1.0f - f0.z, p, f0.z)};
}
// brdf.glsl:35-43
PPT_D float schlick_trowbridge_reitz(float NoL, float NoV, float alpha)
{
    float k = alpha * 0.5f;
    k = fmax_(k, 0.0001f);
    const float gl = NoL / __builtin_fmaf(NoL, 1.0f - k, k);
    const float gv = NoV / __builtin_fmaf(NoV, 1.0f - k, k);
    return gl * gv;
}
// brdf.glsl:46-58
PPT_D f3 cook_torrance_brdf(float NoL, float NoV, float NoH, float VoH, f3 f0, float roughness)
{
    const float alpha = roughness * roughness;
    const float D = trowbridge_reitz(NoH, alpha);
    const f3 F = schlick_fresnel(VoH, f0);
    const float G = schlick_trowbridge_reitz(NoL, NoV, alpha);
    const float denom = __builtin_fmaf(4.0f * NoL, NoV, 0.0001f);
    return ((F * D) * G) / denom;
}
// brdf.glsl:60-64
PPT_D f3 fresnel_zero(const Surface &sf)
{
    const float m = sf.material.metallic;
    return f3{mix(0.04f, sf.material.albedo.x, m), mix(0.04f, sf.material.albedo.y, m),
              mix(0.04f, sf.material.albedo.z, m)};
}
// brdf.glsl:9
PPT_D f3 lambert_brdf(f3 c) { return c * kInvPi; }

// brdf.glsl:67-87
PPT_D f3 eval_brdf_times_nol(f3 l, const Surface &sf)
{
    const f3 h = normalize(sf.invViewRayWS + l);
    const float NoL = saturate(dot(sf.normalWS, l));
    const float NoH = saturate(dot(sf.normalWS, h));
    const float VoH = saturate(dot(sf.invViewRayWS, h));
    const f3 f0 = fresnel_zero(sf);
    const float m = sf.material.metallic;
    const f3 cdiff = f3{mix(sf.material.albedo.x * 0.96f, 0.0f, m), mix(sf.material.albedo.y * 0.96f, 0.0f, m),
                        mix(sf.material.albedo.z * 0.96f, 0.0f, m)};
    return (lambert_brdf(cdiff) + cook_torrance_brdf(NoL, sf.NoV, NoH, VoH, f0, sf.material.roughness)) * NoL;
}

// ------------------------------------------------------------------------------------------
// F17 sampling — common/sampling.glsl
// ------------------------------------------------------------------------------------------

// sampling.glsl:18-33
// (sn, cs) = sincos(2 pi u.y): both lobes of importanceSampleBounce turn the same random number into the same
// angle, so the caller evaluates it once, outside the divergent branch
PPT_D f3 cosine_sample_hemisphere(f3 n, float ux, float sn, float cs)
{
    float a = __builtin_fmaf(-2.0f, ux, 1.0f);
    a *= 0.99999f;
    float b = sqrt_(__builtin_fmaf(-a, a, 1.0f));
    b *= 0.99999f;
    return normalize(f3{__builtin_fmaf(b, cs, n.x), __builtin_fmaf(b, sn, n.y), n.z + a});
}
PPT_D f3 cosine_sample_hemisphere(f3 n, f2 u)
{
    float sn, cs;
    sincos_(kTwoPi * u.y, sn, cs);
    return cosine_sample_hemisphere(n, u.x, sn, cs);
}

// sampling.glsl:37-47 (rows b1, b2, n)
struct Onb
{
    f3 b1, b2, n;
    PPT_D f3 to_local(f3 v) const { return f3{dot(b1, v), dot(b2, v), dot(n, v)}; }
    PPT_D f3 to_world(f3 v) const
    {
        return f3{__builtin_fmaf(n.x, v.z, __builtin_fmaf(b2.x, v.y, b1.x * v.x)),
                  __builtin_fmaf(n.y, v.z, __builtin_fmaf(b2.y, v.y, b1.y * v.x)),
                  __builtin_fmaf(n.z, v.z, __builtin_fmaf(b2.z, v.y, b1.z * v.x))};
    }
};
PPT_D Onb orthonormal_basis(f3 n)
{
    const float s = sign_(n.z);
    const float a = -1.0f / (s + n.z);
    const float b = (n.x * n.y) * a;
    Onb o;
    o.b1 = f3{__builtin_fmaf((s * n.x) * n.x, a, 1.0f), s * b, (-s) * n.x};
    o.b2 = f3{b, __builtin_fmaf(n.y * n.y, a, s), -n.y};
    o.n = n;
    return o;
}

// sampling.glsl:53-79
PPT_D f3 sample_visible_trowbridge_reitz(f3 Ve, float alpha, float ux, float sn, float cs)
{
    const f3 Vh = normalize(f3{alpha * Ve.x, alpha * Ve.y, Ve.z});
    const float lensq = __builtin_fmaf(Vh.y, Vh.y, Vh.x * Vh.x);
    f3 T1;
    if (lensq > 0.0f)
    {
        const float inv = 1.0f / sqrt_(lensq);
        T1 = f3{-Vh.y * inv, Vh.x * inv, 0.0f * inv};
    }
    else
        T1 = f3{1.0f, 0.0f, 0.0f};
    const f3 T2 = cross(Vh, T1);
    const float r = sqrt_(ux);
    const float t1 = r * cs;
    float t2 = r * sn;
    const float s = 0.5f * (1.0f + Vh.z);
    const float c1 = __builtin_fmaf(-t1, t1, 1.0f);
    t2 = __builtin_fmaf(s, t2, (1.0f - s) * sqrt_(c1));
    const float k = sqrt_(fmax_(0.0f, __builtin_fmaf(-t2, t2, c1)));
    const f3 Nh = f3{__builtin_fmaf(Vh.x, k, __builtin_fmaf(T2.x, t2, T1.x * t1)),
                     __builtin_fmaf(Vh.y, k, __builtin_fmaf(T2.y, t2, T1.y * t1)),
                     __builtin_fmaf(Vh.z, k, __builtin_fmaf(T2.z, t2, T1.z * t1))};
    const f3 Ne = normalize(f3{alpha * Nh.x, alpha * Nh.y, fmax_(0.0f, Nh.z)});
    return reflect(-Ve, Ne);
}
PPT_D f3 sample_visible_trowbridge_reitz(f3 Ve, float alpha, f2 Us)
{
    float sn, cs;
    sincos_(kTwoPi * Us.y, sn, cs);
    return sample_visible_trowbridge_reitz(Ve, alpha, Us.x, sn, cs);
}

// sampling.glsl:81-93
PPT_D float visible_trowbridge_reitz_pdf(f3 Ve, f3 Le, float alpha)
{
    const f3 N = f3{0.0f, 0.0f, 1.0f};
    const f3 Ne = normalize(Ve + Le);
    const float NoV = saturate(dot(N, Ve));
    const float NoL = saturate(dot(N, Le));
    const float NoH = saturate(dot(N, Ne));
    const float VNDF = ((schlick_trowbridge_reitz(NoL, NoV, alpha) * NoV) * trowbridge_reitz(NoH, alpha)) / Ve.z;
    return VNDF / (4.0f * NoV);
}

// ------------------------------------------------------------------------------------------
// F14 lights — scene/lighting.glsl
// ------------------------------------------------------------------------------------------

// lighting.glsl:15-37
PPT_D void eval_point_light(const prosper_PointLight &light, f3 surfacePos, f3 &l, float &d, f3 &irradiance)
{
    const f3 pos = f3{light.position.x, light.position.y, light.position.z};
    const f3 radiance = f3{light.radianceAndRadius.x, light.radianceAndRadius.y, light.radianceAndRadius.z};
    const float radius = light.radianceAndRadius.w;
    const f3 toLight = pos - surfacePos;
    const float d2 = dot(toLight, toLight);
    d = sqrt_(d2);
    l = toLight / d;
    const float dPerR = d / radius;
    const float dPerR2 = dPerR * dPerR;
    const float dPerR4 = dPerR2 * dPerR2;
    const float att = fmax_(fmin_(1.0f - dPerR4, 1.0f), 0.0f);
    irradiance = (radiance * att) / d2;
}

// lighting.glsl:39-56
PPT_D void eval_spot_light(const prosper_SpotLight &light, f3 surfacePos, f3 &l, float &d, f3 &irradiance)
{
    const f3 pos = f3{light.positionAndAngleOffset.x, light.positionAndAngleOffset.y, light.positionAndAngleOffset.z};
    const f3 toLight = pos - surfacePos;
    const float d2 = dot(toLight, toLight);
    d = sqrt_(d2);
    l = toLight / d;
    const f3 negDir = f3{-light.direction.x, -light.direction.y, -light.direction.z};
    const float cd = dot(negDir, l);
    float att = saturate(__builtin_fmaf(cd, light.radianceAndAngleScale.w, light.positionAndAngleOffset.w));
    att *= att;
    const f3 rad = f3{light.radianceAndAngleScale.x, light.radianceAndAngleScale.y, light.radianceAndAngleScale.z};
    irradiance = (rad * att) / d2;
}

// lighting.glsl:58-89; returns true when a spot light was picked
PPT_D bool sample_light(const DeviceScene &s, f3 surfacePos, uint32_t lightIndex, f3 &l, float &d, f3 &irradiance)
{
    if (lightIndex == 0)
    {
        const prosper_DirectionalLightParameters sun = *s.directionalLight;
        l = -normalize(f3{sun.direction.x, sun.direction.y, sun.direction.z});
        d = 100.0f;
        irradiance = f3{sun.irradiance.x, sun.irradiance.y, sun.irradiance.z};
        return false;
    }
    lightIndex -= 1;
    const bool isPoint = lightIndex < s.pointLightCount;
    const uint32_t spotIndex = lightIndex - s.pointLightCount;
    if (!isPoint && !(spotIndex < s.spotLightCount))
    {
        l = f3{0.0f, 1.0f, 0.0f};
        d = 1.0f;
        irradiance = f3{0.0f, 0.0f, 0.0f};
        return false;
    }
    // Point and spot lights (eval_point_light / eval_spot_light above, lighting.glsl:15-56) run the same
    // arithmetic for the direction, the distance and the inverse-square term: a wave that holds both kinds
    // does that part once and only the attenuation under the branch - operation for operation what the two
    // functions compute.
    f3 pos, radiance;
    float w0;      // radius | angle scale
    float offset;  // - | angle offset
    f3 direction;  // - | spot direction
    if (isPoint)
    {
        const prosper_PointLight light = s.pointLights->lights[lightIndex];
        pos = f3{light.position.x, light.position.y, light.position.z};
        radiance = f3{light.radianceAndRadius.x, light.radianceAndRadius.y, light.radianceAndRadius.z};
        w0 = light.radianceAndRadius.w;
        offset = 0.0f;
        direction = f3{0.0f, 0.0f, 0.0f};
    }
    else
    {
        const prosper_SpotLight light = s.spotLights->lights[spotIndex];
        pos = f3{light.positionAndAngleOffset.x, light.positionAndAngleOffset.y, light.positionAndAngleOffset.z};
        radiance = f3{light.radianceAndAngleScale.x, light.radianceAndAngleScale.y, light.radianceAndAngleScale.z};
        w0 = light.radianceAndAngleScale.w;
        offset = light.positionAndAngleOffset.w;
        direction = f3{light.direction.x, light.direction.y, light.direction.z};
    }
    const f3 toLight = pos - surfacePos;
    const float d2 = dot(toLight, toLight);
    d = sqrt_(d2);
    l = toLight / d;
    float att;
    if (isPoint)
    {
        const float dPerR = d / w0;
        const float dPerR2 = dPerR * dPerR;
        const float dPerR4 = dPerR2 * dPerR2;
        att = fmax_(fmin_(1.0f - dPerR4, 1.0f), 0.0f);
    }
    else
    {
        const f3 negDir = f3{-direction.x, -direction.y, -direction.z};
        const float cd = dot(negDir, l);
        att = saturate(__builtin_fmaf(cd, w0, offset));
        att *= att;
    }
    irradiance = (radiance * att) / d2;
    return !isPoint;
}

// ------------------------------------------------------------------------------------------
// F6/F7 rays — rt/ray.glsl
// ------------------------------------------------------------------------------------------

struct Ray
{
    f3 o, d;
    float tMin, tMax;
};

// ray.glsl:15-43
PPT_D Ray pinhole_camera_ray(const RenderParams &p, f2 uv)
{
    const float ndx = __builtin_fmaf(uv.x, 2.0f, -1.0f);
    const float ndy = __builtin_fmaf(uv.y, 2.0f, -1.0f);
    Ray ray;
    ray.o = f3{p.eye[0], p.eye[1], p.eye[2]};
    ray.tMin = 0.0f;
    ray.tMax = kInf;
    const float aspect = p.aspect;
    const float tanHalfFovY = p.tanHalfFovY;
    const f3 right = f3{p.right[0], p.right[1], p.right[2]};
    const f3 up = f3{p.up[0], p.up[1], p.up[2]};
    const f3 fwd = f3{p.fwd[0], p.fwd[1], p.fwd[2]};
    const f3 tx = ((right * ndx) * tanHalfFovY) * aspect;
    const f3 ty = (up * ndy) * tanHalfFovY;
    ray.d = normalize((tx + ty) + fwd);
    return ray;
}

// ray.glsl:46-78
PPT_D Ray thin_lens_camera_ray(const RenderParams &p, f2 uv, f2 lensOffset)
{
    const Ray pin = pinhole_camera_ray(p, uv);
    const float theta = (lensOffset.x * 2.0f) * kPi;
    const float radius = lensOffset.y;
    float sn, cs;
    sincos_(theta, sn, cs);
    const float u = cs * sqrt_(radius);
    const float v = sn * sqrt_(radius);
    const f3 fwd = f3{p.fwd[0], p.fwd[1], p.fwd[2]};
    const float k = p.pc.focusDistance / dot(pin.d, fwd);
    const f3 focusPoint = f3{__builtin_fmaf(pin.d.x, k, pin.o.x), __builtin_fmaf(pin.d.y, k, pin.o.y), __builtin_fmaf(pin.d.z, k, pin.o.z)};
    const float fStop = p.pc.focalLength / p.pc.apertureDiameter;
    const float coc = p.pc.focalLength / (2.0f * fStop);
    const f3 lensPos = (f3{1.0f, 0.0f, 0.0f} * (u * coc)) + (f3{0.0f, 1.0f, 0.0f} * (v * coc));
    const float *m = p.cameraToWorld; // column-major
    Ray ray;
    ray.o = f3{__builtin_fmaf(m[8], lensPos.z, __builtin_fmaf(m[4], lensPos.y, __builtin_fmaf(m[0], lensPos.x, m[12]))),
               __builtin_fmaf(m[9], lensPos.z, __builtin_fmaf(m[5], lensPos.y, __builtin_fmaf(m[1], lensPos.x, m[13]))),
               __builtin_fmaf(m[10], lensPos.z, __builtin_fmaf(m[6], lensPos.y, __builtin_fmaf(m[2], lensPos.x, m[14])))};
    ray.d = normalize(focusPoint - ray.o);
    ray.tMin = 0.0f;
    ray.tMax = kInf;
    return ray;
}

// ray.glsl:83-103
PPT_D float offset_component(float p, float n)
{
    const int32_t ofI = f2i(256.0f * n);
    const uint32_t moved = f2u(p) + (uint32_t)((p < 0.0f) ? -ofI : ofI);
    return fabs_(p) < (1.0f / 32.0f) ? __builtin_fmaf(1.0f / 65536.0f, n, p) : u2f(moved);
}
PPT_D f3 offset_ray(f3 p, f3 n)
{
    return f3{offset_component(p.x, n.x), offset_component(p.y, n.y), offset_component(p.z, n.z)};
}

// ------------------------------------------------------------------------------------------
// F2-F5 traversal.  prosper leaves this to the Vulkan driver (World.cpp:740,798;
// main.rgen:57,73); the hit contract of this build is in DESIGN.md: world-space triangles from
// the decoded fp16 positions, scalar-triple-product edge functions (watertight on shared edges),
// no culling, tMin < t < tMax, closest = smallest t with ties to the smaller (instance,
// primitive), any-hit (rt/scene.rahit:18-39) on non-opaque geometry.
// ------------------------------------------------------------------------------------------

PPT_D float safe_rcp_dir(float d)
{
    return 1.0f / (fabs_(d) < 1e-30f ? (d < 0.0f ? -1e-30f : 1e-30f) : d);
}

// Hit contract (iii), the box guard: t must lie in the ray's parametric interval through the triangle's
// bounding box grown by 2^-16 of its largest |coordinate|, within a factor 1 + 2^-18.  The edge
// functions are sums of products of size |v - o|^2; for a small triangle far from the ray origin
// rounding lets them accept a ray that passes slightly outside the triangle, possibly outside every box
// a hierarchy keeps for it.  The guard bounds the acceptance zone by construction, so a hierarchy whose
// boxes contain the guard boxes (bvh_build.cpp) and whose slab test is the same monotone arithmetic
// with a larger tolerance (slab_entry) can never cull a valid candidate.
constexpr float kGuardPad = 1.52587890625e-05f;    // 2^-16
constexpr float kGuardTol = 1.000003814697265625f; // 1 + 2^-18
PPT_D bool box_guard(f3 o, f3 invd, f3 v0, f3 v1, f3 v2, float tt)
{
    const float lox = fminf(fminf(v0.x, v1.x), v2.x), hix = fmaxf(fmaxf(v0.x, v1.x), v2.x);
    const float loy = fminf(fminf(v0.y, v1.y), v2.y), hiy = fmaxf(fmaxf(v0.y, v1.y), v2.y);
    const float loz = fminf(fminf(v0.z, v1.z), v2.z), hiz = fmaxf(fmaxf(v0.z, v1.z), v2.z);
    const float mx = fmaxf(fabs_(lox), fabs_(hix));
    const float my = fmaxf(fabs_(loy), fabs_(hiy));
    const float mz = fmaxf(fabs_(loz), fabs_(hiz));
    const float pad = fmaxf(fmaxf(mx, my), mz) * kGuardPad;
    const float ax = ((lox - pad) - o.x) * invd.x, bx = ((hix + pad) - o.x) * invd.x;
    const float ay = ((loy - pad) - o.y) * invd.y, by = ((hiy + pad) - o.y) * invd.y;
    const float az = ((loz - pad) - o.z) * invd.z, bz = ((hiz + pad) - o.z) * invd.z;
    const float tn = fmaxf(fmaxf(fminf(ax, bx), fminf(ay, by)), fminf(az, bz));
    const float tf = fminf(fminf(fmaxf(ax, bx), fmaxf(ay, by)), fmaxf(az, bz));
    return tn <= tt * kGuardTol && tt <= tf * kGuardTol;
}

// The triangle test in two halves, so that a leaf of two triangles can share the second one between
// them (pt_trace_stream.hpp): first the edge functions of hit contract (1), then distance, range and
// box guard, (2) and (3).  intersect_triangle is the two in a row: same operations, same bits.
struct EdgeFunctions
{
    float U, V, W, det;
    bool pass;
};
PPT_D EdgeFunctions edge_functions(f3 o, f3 d, f3 v0, f3 v1, f3 v2)
{
    const f3 A = v0 - o;
    const f3 B = v1 - o;
    const f3 C = v2 - o;
    EdgeFunctions e;
    e.U = dot(d, cross(C, B));
    e.V = dot(d, cross(A, C));
    e.W = dot(d, cross(B, A));
    e.det = (e.U + e.V) + e.W;
    const bool mixed = (e.U < 0.0f || e.V < 0.0f || e.W < 0.0f) && (e.U > 0.0f || e.V > 0.0f || e.W > 0.0f);
    e.pass = !mixed && e.det != 0.0f;
    return e;
}
// `invd` = safe_rcp_dir of each component of d (the traversal keeps it per ray)
PPT_D bool finish_triangle(
    const EdgeFunctions &e, f3 o, f3 d, f3 invd, f3 v0, f3 v1, f3 v2, float tMin, float tMax, float &t, float &bu, float &bv)
{
    // t = projection of the barycentric point on the (unit) direction: (U*(A.d) + V*(B.d) + W*(C.d)) / det.
    // Well conditioned for grazing rays (the plane equation is not).
    const float inv = 1.0f / e.det;
    const float tt =
        __builtin_fmaf(e.W, dot(v2 - o, d), __builtin_fmaf(e.V, dot(v1 - o, d), e.U * dot(v0 - o, d))) * inv;
    if (!(tt > tMin && tt < tMax)) return false;
    if (!box_guard(o, invd, v0, v1, v2, tt)) return false;
    t = tt;
    bu = e.V * inv;
    bv = e.W * inv;
    return true;
}
PPT_D bool intersect_triangle(
    f3 o, f3 d, f3 invd, f3 v0, f3 v1, f3 v2, float tMin, float tMax, float &t, float &bu, float &bv)
{
    const EdgeFunctions e = edge_functions(o, d, v0, v1, v2);
    if (!e.pass) return false;
    return finish_triangle(e, o, d, invd, v0, v1, v2, tMin, tMax, t, bu, bv);
}

struct Hit
{
    uint32_t drawInstance;
    uint32_t primitive;
    f2 bary;
    float t;
};

// rt/scene.rahit:18-39 on the candidate whose record is alphaTriangles[alphaIndex].
//
// The decision is sampleAlpha's (materials.glsl:121-147) in every case.  What changes is how often its texel fetches,
// filter and sRGBtoLinear run: the material's alpha bounds (pt_scene.hpp AlphaMaterial) hold, for the cell the sample's
// footprint starts in, bytes lo <= alpha <= hi valid for every footprint of that cell.  The footprint is computed first
// - the same uv interpolation, the same i0 / j0 the filter would use - and when the bounds settle the comparison the
// exact code is skipped; otherwise it runs on that footprint.
//   any_hit_settle: kAlphaReject / kAlphaAccept when the bounds (or a material without texture) decide, else
//                   kAlphaUndecided with the footprint in `fp`
//   any_hit_exact:  the texel fetches, the filter, sRGBtoLinear and the comparison on that footprint
struct AlphaFootprint
{
    global_u32_ptr texels;
    uint32_t o00, o10, o01, o11;
    float a, b;
    float factorA, cutoff, u;
    uint32_t mode;
};
enum : uint32_t
{
    kAlphaReject = 0,
    kAlphaAccept = 1,
    kAlphaUndecided = 2,
};

PPT_D bool alpha_verdict(uint32_t mode, float linearAlpha, float factorA, float cutoff, float u)
{
    linearAlpha *= factorA;
    float alpha = -1.0f;
    if (mode == PROSPER_ALPHA_MODE_BLEND)
        alpha = linearAlpha;
    else if (mode == PROSPER_ALPHA_MODE_MASK && linearAlpha < cutoff)
        alpha = 0.0f;
    if (alpha == 0.0f) return false;
    if (alpha > 0.0f && u > alpha) return false;
    return true;
}

template <bool COUNT>
PPT_D uint32_t any_hit_settle(
    const DeviceScene &s, uint32_t alphaIndex, f2 bary, uint32_t randomSeed, LaneCounters &cnt, AlphaFootprint &fp)
{
    const uint4 *rp = reinterpret_cast<const uint4 *>(s.alphaTriangles + alphaIndex);
    const uint4 rec = rp[0];             // uv0, uv1, uv2, drawInstance
    const uint4 m0 = rp[2], m1 = rp[3]; // the material's AlphaMaterial
    const uint64_t texelBits = ((uint64_t)m0.y << 32) | m0.x, boundBits = ((uint64_t)m0.w << 32) | m0.z;
    const uint32_t bits = m1.w;
    fp.factorA = __builtin_bit_cast(float, m1.y);
    fp.cutoff = __builtin_bit_cast(float, m1.z);
    fp.mode = bits & 3u;
    fp.u = (float)pcg(randomSeed) / 4294967296.0f; // the ray's one draw (scene.rahit:35); BLEND only
    if constexpr (COUNT)
    {
        cnt.anyHitCalls++;
        const uint32_t di = s.alphaTriangles[alphaIndex].drawInstance, prim = s.alphaTriangles[alphaIndex].primitive;
        const uint32_t record = s.triangleOffsets[di] + prim;
        const uint32_t recordFlags = PPT_RAW_RECORDS(s) ? s.rawShadeTriangles[record].flags : s.shadeTriangles[record].flags;
        cnt.shortIndexHits += (recordFlags & kTriFlagShortIndices) ? 1u : 0u;
    }
    if (texelBits == 0ull) return alpha_verdict(fp.mode, 1.0f, fp.factorA, fp.cutoff, fp.u) ? kAlphaAccept : kAlphaReject;

    // geometry.glsl:246-256: texCoord0 at the hit
    const f2 uv0 = unpack_half2(rec.x), uv1 = unpack_half2(rec.y), uv2 = unpack_half2(rec.z);
    const float a = (1.0f - bary.x) - bary.y;
    const f2 uv = f2{bary1(uv0.x, uv1.x, uv2.x, a, bary.x, bary.y), bary1(uv0.y, uv1.y, uv2.y, a, bary.x, bary.y)};
    DeviceTexture t;
    t.texels = reinterpret_cast<const uint8_t *>(texelBits);
    t.width = m1.x & 0xFFFFu;
    t.height = m1.x >> 16;
    t.tilesPerRow = (t.width + kTexTileW - 1u) / kTexTileW;
    const uint32_t wrapS = (bits >> 2) & 3u, wrapT = (bits >> 4) & 3u;
    const int32_t w = (int32_t)t.width, h = (int32_t)t.height;
    int32_t i0, i1, j0, j1;
    if (bits & 64u) // nearest: the footprint is one texel (texel_taps)
    {
        i0 = i1 = wrap_coord(f2i(__builtin_floorf(uv.x * (float)w)), w, wrapS);
        j0 = j1 = wrap_coord(f2i(__builtin_floorf(uv.y * (float)h)), h, wrapT);
        fp.a = 0.0f;
        fp.b = 0.0f;
    }
    else
    {
        const float tu = __builtin_fmaf(uv.x, (float)w, -0.5f);
        const float tv = __builtin_fmaf(uv.y, (float)h, -0.5f);
        const float fu = __builtin_floorf(tu);
        const float fv = __builtin_floorf(tv);
        fp.a = tu - fu;
        fp.b = tv - fv;
        wrap_pair(f2i(fu), w, wrapS, i0, i1);
        wrap_pair(f2i(fv), h, wrapT, j0, j1);
    }
    if (boundBits != 0ull && (fp.a + fp.b) == (fp.a + fp.b)) // (a non-finite uv makes the weights NaN: exact code only)
    {
        const uint32_t shift = (bits >> 8) & 15u;
        const uint32_t cellsPerRow = (t.width + (1u << shift) - 1u) >> shift;
        const uint32_t cell = ((uint32_t)j0 >> shift) * cellsPerRow + ((uint32_t)i0 >> shift);
        typedef const __attribute__((address_space(1))) uint16_t *global_u16_ptr;
        const uint32_t b = ((global_u16_ptr)boundBits)[cell];
        const float lo = (float)(b & 0xFFu) * (1.0f / 255.0f);
        const float hi = (b >> 8) == 255u ? kInf : (float)(b >> 8) * (1.0f / 255.0f);
        if (fp.mode == PROSPER_ALPHA_MODE_BLEND)
        {
            if (b < 256u) return kAlphaReject;                     // hi == 0: every texel of the cell is 0, alpha == 0 exactly
            if (fp.u > hi) return kAlphaReject;                    // 0 <= alpha <= hi < u
            if (lo > 0.0f && fp.u <= lo) return kAlphaAccept;      // 0 < u <= lo <= alpha
        }
        else if (fp.mode == PROSPER_ALPHA_MODE_MASK)
        {
            if (hi < fp.cutoff) return kAlphaReject;
            if (lo >= fp.cutoff) return kAlphaAccept;
        }
    }
    fp.texels = (global_u32_ptr)t.texels;
    fp.o00 = texel_offset(t, i0, j0);
    fp.o10 = texel_offset(t, i1, j0);
    fp.o01 = texel_offset(t, i0, j1);
    fp.o11 = texel_offset(t, i1, j1);
    return kAlphaUndecided;
}

template <bool COUNT>
PPT_D bool any_hit_exact(const AlphaFootprint &fp, LaneCounters &cnt)
{
    if constexpr (COUNT) cnt.anyHitTexelFetches++;
    TexelTaps k;
    k.texels = fp.texels;
    k.o00 = fp.o00;
    k.o10 = fp.o10;
    k.o01 = fp.o01;
    k.o11 = fp.o11;
    k.a = fp.a;
    k.b = fp.b;
    const float linearAlpha = srgb_to_linear(filter_taps(k, fetch_taps(k)).w);
    return alpha_verdict(fp.mode, linearAlpha, fp.factorA, fp.cutoff, fp.u);
}

// both steps; true = accept the candidate
template <bool COUNT>
PPT_D bool any_hit_record(const DeviceScene &s, uint32_t alphaIndex, f2 bary, uint32_t randomSeed, LaneCounters &cnt)
{
    AlphaFootprint fp;
    const uint32_t v = any_hit_settle<COUNT>(s, alphaIndex, bary, randomSeed, cnt, fp);
    if (v != kAlphaUndecided) return v == kAlphaAccept;
    return any_hit_exact<COUNT>(fp, cnt);
}

// the same for callers that hold (drawInstance, primitive) instead of the triangle's flags word
template <bool COUNT>
PPT_D bool any_hit(
    const DeviceScene &s, uint32_t drawInstance, uint32_t primitive, f2 bary, uint32_t randomSeed, LaneCounters &cnt)
{
    return any_hit_record<COUNT>(s, s.alphaOffsets[drawInstance] + primitive, bary, randomSeed, cnt);
}

// Where the traversal reads BVH nodes and world triangles from.  GlobalGeom: the HBM arrays (through
// L1/L2).  LdsGeom: a copy a workgroup staged in LDS — for scenes of a few KB (a Cornell box) every
// node fetch then costs an LDS access (~64 cycles) instead of a vector-memory round trip (~200+),
// which is what the latency-bound traversal of tiny scenes is made of.  Node stride 80 B and
// triangle stride 48 B keep 16-byte reads of different records spread over 16 bank groups.
struct TriangleData
{
    float4 a, b, c;
};
// one 80-B BvhNode as five 16-B words: q0 = origin.xyz, q1 = lo.x[4] lo.y[4], q2 = lo.z[4] hi.x[4],
// q3 = hi.y[4] hi.z[4] (halfs, two per dword), q4 = child[4]
struct NodeData
{
    uint4 q0, q1, q2, q3, q4;
};
// The traversal stack of one lane: `cap` entries in LDS (entry e at lds[e * 64], conflict-free for
// b32 accesses), anything deeper in a global overflow column (entry e at ovf[e * ovfStride]).  The
// builder bounds the worst case (kMaxStackBound); the LDS part is sized for the common case.
typedef __attribute__((address_space(3))) int32_t lds_int32; // ds_read/ds_write, never a flat access
struct TraversalStack
{
    lds_int32 *lds;
    int32_t *ovf;
    uint32_t cap;
    uint32_t ovfStride;
    uint32_t ldsStride; // 64: one column per lane of the wave
    // the tree's stack bound fits the LDS part (a compile-time constant where it is set: the kernels of LDS-resident scenes):
    // no capacity test, no global path - 1 vector + 5 scalar instructions and two branches less per push and pop
    bool ldsOnly;
    PPT_D void push(int32_t &sp, int32_t v) const
    {
        if (ldsOnly || (uint32_t)sp < cap)
            lds[(uint32_t)sp * ldsStride] = v;
        else
            ovf[(size_t)((uint32_t)sp - cap) * ovfStride] = v;
        ++sp;
    }
    PPT_D int32_t pop(int32_t &sp) const
    {
        --sp;
        if (ldsOnly || (uint32_t)sp < cap) return lds[(uint32_t)sp * ldsStride];
        return ovf[(size_t)((uint32_t)sp - cap) * ovfStride];
    }
    // up to three pushes of one node visit (the far children of a closest-hit ray, farthest first): ONE capacity test for
    // all three - the LDS part nearly always has room for three more - instead of one per push
    PPT_D void push_hit_children(int32_t &sp, const float e[4], const int32_t ref[4]) const
    {
#ifndef PPT_EXPERIMENT_PUSH_EACH
        if (ldsOnly || (uint32_t)sp + 3u <= cap)
        {
            if (e[3] < kInf) lds[(uint32_t)(sp++) * ldsStride] = ref[3];
            if (e[2] < kInf) lds[(uint32_t)(sp++) * ldsStride] = ref[2];
            if (e[1] < kInf) lds[(uint32_t)(sp++) * ldsStride] = ref[1];
            return;
        }
#endif
        if (e[3] < kInf) push(sp, ref[3]);
        if (e[2] < kInf) push(sp, ref[2]);
        if (e[1] < kInf) push(sp, ref[1]);
    }
};

typedef float v2f __attribute__((ext_vector_type(2)));

// (plane - o) for the two binary16 planes packed in one dword, each as ONE v_fma_mix_f32
// (fma(float(half), 1.0, -o): the half -> float conversion rides on the instruction and the
// difference is rounded once, i.e. the value v_cvt_f32_f16 + v_sub_f32 gives).  Written as asm
// because the optimiser folds fma(x, 1, y) back into convert + add.
PPT_D v2f plane_offsets(uint32_t packed, float o)
{
    v2f r;
    asm("v_fma_mix_f32 %0, %1, 1.0, -%2 op_sel_hi:[1,0,0]" : "=v"(r.x) : "v"(packed), "v"(o));
    asm("v_fma_mix_f32 %0, %1, 1.0, -%2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r.y) : "v"(packed), "v"(o));
    return r;
}

// Conservative slab test of one child box from the distances to its three near and three far planes;
// entry distance or +inf on a miss.  The distances are t = (plane - o) * invd: the subtraction stays
// exact when the origin is near the plane (the fused form plane*invd - o*invd cancels catastrophically
// there and needs per-axis error terms that cost as much as the subtraction).  The far distances
// arrive scaled by 1 + 2^-16 (> (1 + 2^-18)^2, the box guard's tolerance on both sides), as does tMaxK.
PPT_D float slab_entry(float nx, float ny, float nz, float fx, float fy, float fz, float tMin, float tMaxK)
{
    const float tn = fmaxf(fmaxf(nx, ny), nz);
    const float tf = fminf(fminf(fx, fy), fz);
    const bool hit = fmaxf(tn, tMin) <= fminf(tf, tMaxK);
    return hit ? tn : kInf;
}

// What the node test needs per ray besides the origin: 1/d per axis, the same scaled by the slab
// tolerance (for the far planes) and which planes are the near ones.
struct RaySlabs
{
    f3 invd;  // safe_rcp_dir(d)
    f3 invdK; // invd * (1 + 2^-16)
};
constexpr float kSlabTol = 1.0000152587890625f; // 1 + 2^-16
PPT_D RaySlabs make_ray_slabs(f3 invd)
{
    return RaySlabs{invd, f3{invd.x * kSlabTol, invd.y * kSlabTol, invd.z * kSlabTol}};
}

// Tests the four children of a node; e[c] = entry distance or +inf, then sorts (e, ref) ascending
// (5-comparator network) so the nearest child is walked first: children that were hit come first,
// e[k] < inf says whether the k-th nearest exists.
// Per axis the near planes are the lo planes when the ray travels in + and the hi planes otherwise, so
// the dwords (two children each) are swapped up front (12 v_cndmask) instead of taking min/max of the
// 24 products; 24 v_fma_mix_f32 + 12 v_pk_mul_f32 then produce the distances.
// SORTED = false leaves the four (entry, child) pairs in storage order: enough for an any-hit ray, whose answer
// does not depend on the order its candidates are met in (descend_any below).
// the common second half of the node test: slab tests of the four children from their twelve near and twelve far
// distances (pairs: children 0|1 and 2|3), then the sort
template <bool SORTED>
PPT_D void finish_node4(
    v2f tnx01, v2f tnx23, v2f tny01, v2f tny23, v2f tnz01, v2f tnz23, v2f tfx01, v2f tfx23, v2f tfy01, v2f tfy23, v2f tfz01,
    v2f tfz23, uint4 children, float tMin, float tMax, float e[4], int32_t ref[4])
{
    const float tMaxK = tMax * kSlabTol;
    e[0] = slab_entry(tnx01.x, tny01.x, tnz01.x, tfx01.x, tfy01.x, tfz01.x, tMin, tMaxK);
    e[1] = slab_entry(tnx01.y, tny01.y, tnz01.y, tfx01.y, tfy01.y, tfz01.y, tMin, tMaxK);
    e[2] = slab_entry(tnx23.x, tny23.x, tnz23.x, tfx23.x, tfy23.x, tfz23.x, tMin, tMaxK);
    e[3] = slab_entry(tnx23.y, tny23.y, tnz23.y, tfx23.y, tfy23.y, tfz23.y, tMin, tMaxK);
    ref[0] = (int32_t)children.x;
    ref[1] = (int32_t)children.y;
    ref[2] = (int32_t)children.z;
    ref[3] = (int32_t)children.w;
#define PPT_CSWAP(i, j)                                                                                                \
    {                                                                                                                  \
        const bool sw = e[j] < e[i];                                                                                   \
        const float te = sw ? e[j] : e[i];                                                                             \
        e[j] = sw ? e[i] : e[j];                                                                                       \
        e[i] = te;                                                                                                     \
        const int32_t tr = sw ? ref[j] : ref[i];                                                                       \
        ref[j] = sw ? ref[i] : ref[j];                                                                                 \
        ref[i] = tr;                                                                                                   \
    }
    if constexpr (SORTED)
    {
        PPT_CSWAP(0, 1)
        PPT_CSWAP(2, 3)
        PPT_CSWAP(0, 2)
        PPT_CSWAP(1, 3)
        PPT_CSWAP(1, 2)
    }
#undef PPT_CSWAP
}

template <bool SORTED = true>
PPT_D void intersect_node4(const NodeData &n, f3 o, const RaySlabs &rs, float tMin, float tMax, float e[4], int32_t ref[4])
{
    // the ray origin relative to the node (one rounding of 2^-24 |o - origin|, inside the builder's slack)
    const float ox = o.x - __builtin_bit_cast(float, n.q0.x);
    const float oy = o.y - __builtin_bit_cast(float, n.q0.y);
    const float oz = o.z - __builtin_bit_cast(float, n.q0.z);
    const bool sx = rs.invd.x < 0.0f, sy = rs.invd.y < 0.0f, sz = rs.invd.z < 0.0f;
    // q1 = lo.x[0..3] lo.y[0..3], q2 = lo.z[0..3] hi.x[0..3], q3 = hi.y[0..3] hi.z[0..3]
    const uint32_t nx01 = sx ? n.q2.z : n.q1.x, nx23 = sx ? n.q2.w : n.q1.y;
    const uint32_t fx01 = sx ? n.q1.x : n.q2.z, fx23 = sx ? n.q1.y : n.q2.w;
    const uint32_t ny01 = sy ? n.q3.x : n.q1.z, ny23 = sy ? n.q3.y : n.q1.w;
    const uint32_t fy01 = sy ? n.q1.z : n.q3.x, fy23 = sy ? n.q1.w : n.q3.y;
    const uint32_t nz01 = sz ? n.q3.z : n.q2.x, nz23 = sz ? n.q3.w : n.q2.y;
    const uint32_t fz01 = sz ? n.q2.x : n.q3.z, fz23 = sz ? n.q2.y : n.q3.w;
    const v2f tnx01 = plane_offsets(nx01, ox) * rs.invd.x, tnx23 = plane_offsets(nx23, ox) * rs.invd.x;
    const v2f tny01 = plane_offsets(ny01, oy) * rs.invd.y, tny23 = plane_offsets(ny23, oy) * rs.invd.y;
    const v2f tnz01 = plane_offsets(nz01, oz) * rs.invd.z, tnz23 = plane_offsets(nz23, oz) * rs.invd.z;
    const v2f tfx01 = plane_offsets(fx01, ox) * rs.invdK.x, tfx23 = plane_offsets(fx23, ox) * rs.invdK.x;
    const v2f tfy01 = plane_offsets(fy01, oy) * rs.invdK.y, tfy23 = plane_offsets(fy23, oy) * rs.invdK.y;
    const v2f tfz01 = plane_offsets(fz01, oz) * rs.invdK.z, tfz23 = plane_offsets(fz23, oz) * rs.invdK.z;
    finish_node4<SORTED>(tnx01, tnx23, tny01, tny23, tnz01, tnz23, tfx01, tfx23, tfy01, tfy23, tfz01, tfz23, n.q4, tMin, tMax, e, ref);
}

struct GlobalGeom
{
    const BvhNode *nodes;
    const WorldTriangle *triangles;
    PPT_D NodeData node(int32_t i) const
    {
        const uint4 *np = reinterpret_cast<const uint4 *>(nodes + i);
#ifdef PPT_EXPERIMENT_SPLIT_NODE_LOADS
        // measurement only (scripts/build_variant.sh split -DPPT_EXPERIMENT_SPLIT_NODE_LOADS): the same 80 bytes as ten
        // 8-byte loads - twice the cache accesses, same lines (profiles/r02_gather_microbench.txt)
        typedef uint32_t u32x2v __attribute__((ext_vector_type(2)));
        u32x2v h[10];
        asm volatile("global_load_dwordx2 %0, %10, off\n\tglobal_load_dwordx2 %1, %10, off offset:8\n\t"
                     "global_load_dwordx2 %2, %10, off offset:16\n\tglobal_load_dwordx2 %3, %10, off offset:24\n\t"
                     "global_load_dwordx2 %4, %10, off offset:32\n\tglobal_load_dwordx2 %5, %10, off offset:40\n\t"
                     "global_load_dwordx2 %6, %10, off offset:48\n\tglobal_load_dwordx2 %7, %10, off offset:56\n\t"
                     "global_load_dwordx2 %8, %10, off offset:64\n\tglobal_load_dwordx2 %9, %10, off offset:72\n\t"
                     "s_waitcnt vmcnt(0)"
                     : "=&v"(h[0]), "=&v"(h[1]), "=&v"(h[2]), "=&v"(h[3]), "=&v"(h[4]), "=&v"(h[5]), "=&v"(h[6]),
                       "=&v"(h[7]), "=&v"(h[8]), "=&v"(h[9])
                     : "v"(np)
                     : "memory");
        return NodeData{uint4{h[0].x, h[0].y, h[1].x, h[1].y}, uint4{h[2].x, h[2].y, h[3].x, h[3].y},
                        uint4{h[4].x, h[4].y, h[5].x, h[5].y}, uint4{h[6].x, h[6].y, h[7].x, h[7].y},
                        uint4{h[8].x, h[8].y, h[9].x, h[9].y}};
#else
        return NodeData{np[0], np[1], np[2], np[3], np[4]};
#endif
    }
    template <bool SORTED>
    PPT_D void test_node(int32_t i, f3 o, const RaySlabs &rs, float tMin, float tMax, float e[4], int32_t ref[4]) const
    {
        intersect_node4<SORTED>(node(i), o, rs, tMin, tMax, e, ref);
    }
    PPT_D TriangleData tri(uint32_t i) const
    {
        const float4 *tp = reinterpret_cast<const float4 *>(triangles + i);
        return TriangleData{tp[0], tp[1], tp[2]};
    }
};
// A node as the kernels of LDS-resident scenes keep it (stage_scene_in_lds converts): the planes as fp32 - exactly the values
// the binary16 ones decode to - one float4 (four children) per plane set, the near / far sets of an axis 64 bytes apart:
//   f4[0] origin.xyz   f4[1] lo.x  f4[2] lo.y  f4[3] lo.z   f4[4] child[0..3]   f4[5] hi.x  f4[6] hi.y  f4[7] hi.z   f4[8] padding
// A ray reads its near planes at  16 (a + 1) + 64 s_a  and its far planes at  16 (a + 5) - 64 s_a  (s_a = 1 where it travels
// in -a): the choice of planes is part of the ADDRESS (two integer instructions per axis) instead of twelve v_cndmask, and
// (plane - o) is a plain v_sub_f32 instead of a v_fma_mix_f32 - same values, same rounding: 9 x 16 B per visit out of LDS
// (256 B/clk per CU) instead of 5 x 16 B, ~15 % fewer issue cycles per node visit.  The 144-byte stride spreads the same
// float4 of neighbouring nodes over different banks.
constexpr uint32_t kLdsNodeStride = 9; // float4s per node in LDS (144 B)
struct LdsGeom
{
    const float4 *nodes;     // LDS: kLdsNodeStride float4s per node (see above)
    const float4 *triangles; // LDS
    template <bool SORTED>
    PPT_D void test_node(int32_t i, f3 o, const RaySlabs &rs, float tMin, float tMax, float e[4], int32_t ref[4]) const
    {
        typedef float f4n __attribute__((ext_vector_type(4)));
        typedef uint32_t u4n __attribute__((ext_vector_type(4)));
        typedef __attribute__((address_space(3))) const f4n *lds_f4;
        typedef __attribute__((address_space(3))) const u4n *lds_u4;
        // (an LDS address is its low 32 bits; a generic pointer into LDS converts to one by dropping the aperture)
        const uint32_t base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void *)nodes + (uint32_t)i * (kLdsNodeStride * 16u);
        // 64 where the ray travels in - along the axis: the sign bit of 1 / d moved to bit 6
        const uint32_t sx = (__builtin_bit_cast(uint32_t, rs.invd.x) >> 25) & 64u;
        const uint32_t sy = (__builtin_bit_cast(uint32_t, rs.invd.y) >> 25) & 64u;
        const uint32_t sz = (__builtin_bit_cast(uint32_t, rs.invd.z) >> 25) & 64u;
        const f4n origin = *(lds_f4)(uintptr_t)(base);
        const f4n nx = *(lds_f4)(uintptr_t)(base + 16u + sx), fx = *(lds_f4)(uintptr_t)(base + 80u - sx);
        const f4n ny = *(lds_f4)(uintptr_t)(base + 32u + sy), fy = *(lds_f4)(uintptr_t)(base + 96u - sy);
        const f4n nz = *(lds_f4)(uintptr_t)(base + 48u + sz), fz = *(lds_f4)(uintptr_t)(base + 112u - sz);
        const u4n ch = *(lds_u4)(uintptr_t)(base + 64u);
        const uint4 children = make_uint4(ch.x, ch.y, ch.z, ch.w);
        const float ox = o.x - origin.x, oy = o.y - origin.y, oz = o.z - origin.z;
        const v2f tnx01 = v2f{nx.x - ox, nx.y - ox} * rs.invd.x, tnx23 = v2f{nx.z - ox, nx.w - ox} * rs.invd.x;
        const v2f tny01 = v2f{ny.x - oy, ny.y - oy} * rs.invd.y, tny23 = v2f{ny.z - oy, ny.w - oy} * rs.invd.y;
        const v2f tnz01 = v2f{nz.x - oz, nz.y - oz} * rs.invd.z, tnz23 = v2f{nz.z - oz, nz.w - oz} * rs.invd.z;
        const v2f tfx01 = v2f{fx.x - ox, fx.y - ox} * rs.invdK.x, tfx23 = v2f{fx.z - ox, fx.w - ox} * rs.invdK.x;
        const v2f tfy01 = v2f{fy.x - oy, fy.y - oy} * rs.invdK.y, tfy23 = v2f{fy.z - oy, fy.w - oy} * rs.invdK.y;
        const v2f tfz01 = v2f{fz.x - oz, fz.y - oz} * rs.invdK.z, tfz23 = v2f{fz.z - oz, fz.w - oz} * rs.invdK.z;
        finish_node4<SORTED>(tnx01, tnx23, tny01, tny23, tnz01, tnz23, tfx01, tfx23, tfy01, tfy23, tfz01, tfz23, children, tMin, tMax, e, ref);
    }
    PPT_D TriangleData tri(uint32_t i) const
    {
        const float4 *tp = triangles + i * 3u;
        return TriangleData{tp[0], tp[1], tp[2]};
    }
};
// The nodes as they are in HBM (80 B) in LDS: what the camera-ray kernel uses - its lockstep loop holds a whole node of
// the fp32 image in registers at once and spills at its 96-VGPR budget with it (9 dwords), the stream scheduler's does not.
constexpr uint32_t kLdsNodeStrideHalf = 5; // float4s per node (80 B)
struct LdsGeomHalf
{
    const float4 *nodes;     // LDS
    const float4 *triangles; // LDS
    PPT_D NodeData node(int32_t i) const
    {
        const uint4 *np = reinterpret_cast<const uint4 *>(nodes + (uint32_t)i * kLdsNodeStrideHalf);
        return NodeData{np[0], np[1], np[2], np[3], np[4]};
    }
    template <bool SORTED>
    PPT_D void test_node(int32_t i, f3 o, const RaySlabs &rs, float tMin, float tMax, float e[4], int32_t ref[4]) const
    {
        intersect_node4<SORTED>(node(i), o, rs, tMin, tMax, e, ref);
    }
    PPT_D TriangleData tri(uint32_t i) const
    {
        const float4 *tp = triangles + i * 3u;
        return TriangleData{tp[0], tp[1], tp[2]};
    }
};
// LDS budget of a staged scene (float4s): nodes * kLdsNodeStride + triangles * 3 must fit
constexpr uint32_t kLdsSceneFloat4s = 768; // 12 KB

// Any-hit rays (shadow(): terminate on the first accepted hit) take the children a node test reports in storage
// order: the first one entered becomes the next node, the others go on the stack.  Whether the ray is occluded does
// not depend on the order - every candidate's acceptance is a function of the ray and the candidate alone - so the
// 25-instruction distance sort is left out.  Returns false when no child was entered.
PPT_D bool descend_any(const float e[4], const int32_t ref[4], const TraversalStack &stack, int32_t &sp, int32_t &node)
{
    bool found = false;
#ifndef PPT_EXPERIMENT_PUSH_EACH
    if (stack.ldsOnly || (uint32_t)sp + 3u <= stack.cap) // one capacity test for the (at most three) pushes of this visit
    {
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (e[k] < kInf)
            {
                if (found)
                    stack.lds[(uint32_t)(sp++) * stack.ldsStride] = ref[k];
                else
                {
                    node = ref[k];
                    found = true;
                }
            }
        return found;
    }
#endif
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (e[k] < kInf)
        {
            if (found)
                stack.push(sp, ref[k]);
            else
            {
                node = ref[k];
                found = true;
            }
        }
    return found;
}

// Shared driver of traceClosest (ANY = false, main.rgen:62-81) and shadow (ANY = true,
// main.rgen:49-60).  `stack` points at this lane's column of the workgroup's LDS stack
// (entry e lives at stack[e * 64]).  Returns true on a hit (ANY: occluded).
template <bool ANY, bool COUNT, class Geom>
PPT_D bool trace_in(
    const Geom &g, const DeviceScene &s, f3 o, f3 d, float tMin, float tMaxIn, uint32_t seed, const TraversalStack &stack,
    Hit &hit, LaneCounters &cnt)
{
    hit.drawInstance = kMissIndex;
    hit.primitive = kMissIndex;
    hit.bary = f2{0.0f, 0.0f};
    hit.t = tMaxIn;
    const f3 invd = f3{safe_rcp_dir(d.x), safe_rcp_dir(d.y), safe_rcp_dir(d.z)};
    const RaySlabs rs = make_ray_slabs(invd);

    // while-while traversal: all lanes first descend inner nodes until every live lane holds a leaf
    // (lanes that already do wait), then all lanes intersect their leaf's triangles.  Under
    // divergence this keeps the (long) triangle code out of the node loop and vice versa.
    int32_t sp = 0;
    int32_t node = 0; // the root is always an inner node
    bool alive = true;
    while (alive)
    {
        while (alive && node >= 0)
        {
            if constexpr (COUNT) cnt.nodeVisits++;
            float e[4];
            int32_t ref[4];
            bool entered;
            if constexpr (ANY)
            {
                g.template test_node<false>(node, o, rs, tMin, hit.t, e, ref);
                entered = descend_any(e, ref, stack, sp, node);
            }
            else
            {
                g.template test_node<true>(node, o, rs, tMin, hit.t, e, ref);
                stack.push_hit_children(sp, e, ref);
                entered = e[0] < kInf;
                if (entered) node = ref[0];
            }
            if (entered)
                ;
            else if (sp == 0)
                alive = false;
            else
                node = stack.pop(sp);
        }
        if (!alive) break;
        {
            const uint32_t ref = (uint32_t)~node;
            const uint32_t first = ref >> 3;
            const uint32_t count = (ref & 7u) + 1u;
            for (uint32_t i = 0; i < count; ++i)
            {
                const TriangleData td = g.tri(first + i);
                const float4 a = td.a, b = td.b, c = td.c;
                if constexpr (COUNT)
                {
                    cnt.triangleTests++;
                    cnt.shortIndexTriangleTests += (__builtin_bit_cast(uint32_t, c.w) & kTriFlagShortIndices) ? 1u : 0u;
                }
                float t, bu, bv;
                if (!intersect_triangle(
                        o, d, invd, f3{a.x, a.y, a.z}, f3{b.x, b.y, b.z}, f3{c.x, c.y, c.z}, tMin, tMaxIn, t, bu, bv))
                    continue;
                const uint32_t di = __builtin_bit_cast(uint32_t, a.w);
                const uint32_t prim = __builtin_bit_cast(uint32_t, b.w);
                const uint32_t flags = __builtin_bit_cast(uint32_t, c.w);
                if (!ANY && hit.drawInstance != kMissIndex)
                {
                    if (t > hit.t) continue;
                    if (t == hit.t && !(di < hit.drawInstance || (di == hit.drawInstance && prim < hit.primitive)))
                        continue;
                }
                if (!(flags & kTriFlagOpaque) && !any_hit_record<COUNT>(s, flags >> kTriAlphaShift, f2{bu, bv}, seed, cnt))
                    continue;
                hit.drawInstance = di;
                hit.primitive = prim;
                hit.bary = f2{bu, bv};
                hit.t = t;
                if (ANY) return true;
            }
        }
        if (sp == 0) break;
        node = stack.pop(sp);
    }
    return hit.drawInstance != kMissIndex;
}

template <bool ANY, bool COUNT>
PPT_D bool trace(
    const DeviceScene &s, f3 o, f3 d, float tMin, float tMaxIn, uint32_t seed, const TraversalStack &stack, Hit &hit,
    LaneCounters &cnt)
{
    return trace_in<ANY, COUNT>(GlobalGeom{s.nodes, s.triangles}, s, o, d, tMin, tMaxIn, seed, stack, hit, cnt);
}

// ------------------------------------------------------------------------------------------
// F9 hit -> shading frame — rt/reference/main.rgen:146-179, :37-45
// ------------------------------------------------------------------------------------------

PPT_D f3 mapped_normal(f3 tsn, f3 normal, f3 tangent, float sgn)
{
    const f3 vB = cross(normal, tangent) * sgn;
    return normalize(((tangent * tsn.x) + (vB * tsn.y)) + (normal * tsn.z));
}

template <bool COUNT, bool BATCHED_TEXTURES = false>
PPT_D Surface evaluate_surface(const DeviceScene &s, f3 rayDir, const Hit &hit, LaneCounters &cnt)
{
    // loadVertexThroughIndexBuffer x 3 (geometry.glsl:220-244) from the triangle's record: decoded at upload (128 B), or
    // - big scenes - the raw stream values (64 B) decoded here by the same functions
    const prosper_DrawInstance inst = s.drawInstances[hit.drawInstance];
    const uint32_t record = s.triangleOffsets[hit.drawInstance] + hit.primitive;
    Vertex v0, v1, v2;
    uint32_t recordFlags;
    if (PPT_RAW_RECORDS(s))
    {
        const uint4 *rec = reinterpret_cast<const uint4 *>(s.rawShadeTriangles + record);
        const uint4 q0 = rec[0], q1 = rec[1], q2 = rec[2], q3 = rec[3];
        recordFlags = q3.w;
        v0.position = unpack_half3(q0.x, q0.y);
        v1.position = unpack_half3(q0.z, q0.w);
        v2.position = unpack_half3(q1.x, q1.y);
        const f3 zero = f3{0.0f, 0.0f, 0.0f};
        const bool noNormals = (recordFlags & kRawNoNormals) != 0u, noTangents = (recordFlags & kRawNoTangents) != 0u;
        v0.normal = noNormals ? zero : unpack_snorm_r10g10b10(q1.z);
        v1.normal = noNormals ? zero : unpack_snorm_r10g10b10(q1.w);
        v2.normal = noNormals ? zero : unpack_snorm_r10g10b10(q2.x);
        const f3 t0 = unpack_snorm_r10g10b10(q2.y), t1 = unpack_snorm_r10g10b10(q2.z), t2 = unpack_snorm_r10g10b10(q2.w);
        v0.tangent = noTangents ? f4{0.0f, 0.0f, 0.0f, 0.0f} : f4{t0.x, t0.y, t0.z, (float)((int32_t)q2.y >> 30)};
        v1.tangent = noTangents ? f4{0.0f, 0.0f, 0.0f, 0.0f} : f4{t1.x, t1.y, t1.z, (float)((int32_t)q2.z >> 30)};
        v2.tangent = noTangents ? f4{0.0f, 0.0f, 0.0f, 0.0f} : f4{t2.x, t2.y, t2.z, (float)((int32_t)q2.w >> 30)};
        v0.uv = unpack_half2(q3.x);
        v1.uv = unpack_half2(q3.y);
        v2.uv = unpack_half2(q3.z);
    }
    else
    {
        const float4 *rec = reinterpret_cast<const float4 *>(s.shadeTriangles + record);
        const float4 q0 = rec[0], q1 = rec[1], q2 = rec[2], q3 = rec[3], q4 = rec[4], q5 = rec[5], q6 = rec[6], q7 = rec[7];
        recordFlags = __builtin_bit_cast(uint32_t, q7.z);
        v0.normal = f3{q0.x, q0.y, q0.z};
        v1.normal = f3{q1.x, q1.y, q1.z};
        v2.normal = f3{q2.x, q2.y, q2.z};
        v0.uv = unpack_half2(__builtin_bit_cast(uint32_t, q0.w));
        v1.uv = unpack_half2(__builtin_bit_cast(uint32_t, q1.w));
        v2.uv = unpack_half2(__builtin_bit_cast(uint32_t, q2.w));
        v0.tangent = f4{q3.x, q3.y, q3.z, q3.w};
        v1.tangent = f4{q4.x, q4.y, q4.z, q4.w};
        v2.tangent = f4{q5.x, q5.y, q5.z, q5.w};
        v0.position = unpack_half3(__builtin_bit_cast(uint32_t, q6.x), __builtin_bit_cast(uint32_t, q6.y));
        v1.position = unpack_half3(__builtin_bit_cast(uint32_t, q6.z), __builtin_bit_cast(uint32_t, q6.w));
        v2.position = unpack_half3(__builtin_bit_cast(uint32_t, q7.x), __builtin_bit_cast(uint32_t, q7.y));
    }
    const Vertex vi = interpolate(v0, v1, v2, hit.bary);
    const Vertex v = transform(vi, s.modelInstanceTransforms[inst.modelInstanceIndex]);
    if constexpr (COUNT)
    {
        cnt.closestHits++;
        cnt.shortIndexHits += (recordFlags & kTriFlagShortIndices) ? 1u : 0u;
    }
    Surface sf;
    sf.positionWS = v.position;
    sf.invViewRayWS = -rayDir;
    sf.uv = v.uv;
    sf.material = sample_material<BATCHED_TEXTURES>(s, inst.materialIndex, v.uv);
    if (sf.material.normal.x != -2.0f && v.tangent.w != 0.0f)
        sf.normalWS = mapped_normal(sf.material.normal, v.normal, f3{v.tangent.x, v.tangent.y, v.tangent.z}, v.tangent.w);
    else
        sf.normalWS = normalize(v.normal);
    sf.NoV = saturate(dot(sf.normalWS, sf.invViewRayWS));
    return sf;
}

// main.rgen:181-193 + debug.glsl:17-38
PPT_D f3 debug_color(const DeviceScene &s, uint32_t drawType, const Hit &hit, const Surface &sf)
{
    const prosper_DrawInstance inst = s.drawInstances[hit.drawInstance];
    switch (drawType)
    {
    case PROSPER_DRAW_TYPE_PRIMITIVE_ID: return uint_to_color(hit.primitive);
    case PROSPER_DRAW_TYPE_MESH_ID: return uint_to_color(inst.meshIndex);
    case PROSPER_DRAW_TYPE_MATERIAL_ID: return uint_to_color(inst.materialIndex);
    case PROSPER_DRAW_TYPE_POSITION: return sf.positionWS;
    case PROSPER_DRAW_TYPE_TEXCOORD0: return f3{sf.uv.x, sf.uv.y, 0.0f};
    case PROSPER_DRAW_TYPE_ALBEDO: return sf.material.albedo;
    case PROSPER_DRAW_TYPE_SHADING_NORMAL:
        return f3{__builtin_fmaf(sf.normalWS.x, 0.5f, 0.5f), __builtin_fmaf(sf.normalWS.y, 0.5f, 0.5f), __builtin_fmaf(sf.normalWS.z, 0.5f, 0.5f)};
    case PROSPER_DRAW_TYPE_ROUGHNESS: return f3{sf.material.roughness, sf.material.roughness, sf.material.roughness};
    case PROSPER_DRAW_TYPE_METALLIC: return f3{sf.material.metallic, sf.material.metallic, sf.material.metallic};
    default: return f3{1.0f, 0.0f, 1.0f};
    }
}

// main.rgen:83-88
PPT_D void add_bounce(uint32_t flags, f3 &acc, f3 color, uint32_t bounce)
{
    if (bounce > 0 && (flags & PROSPER_PC_FLAG_CLAMP_INDIRECT))
        color = f3{clamp_(color.x, 0.0f, 2.0f), clamp_(color.y, 0.0f, 2.0f), clamp_(color.z, 0.0f, 2.0f)};
    acc = acc + color;
}

// main.rgen:90-144
PPT_D void importance_sample_bounce(const Surface &sf, Rng &rng, f3 &throughput, f3 &rd)
{
    const bool specularOnly = sf.material.metallic > 0.999f;
    const float specularWeight = specularOnly ? 1.0f : 0.5f;
    const float diffuseWeight = 1.0f - specularWeight;
    const Onb basis = orthonormal_basis(sf.normalWS);
    const f3 vInBasis = basis.to_local(sf.invViewRayWS);
    const float alpha = sf.material.roughness * sf.material.roughness;
    const bool pickDiffuse = rng.rnd01() < diffuseWeight;
    const f2 u = rng.rnd2d01();
    float sinPhi, cosPhi; // phi = 2 pi u.y in cosineSampleHemisphere and in sampleVisibleTrowbridgeReitz alike
    sincos_(kTwoPi * u.y, sinPhi, cosPhi);
    f3 brdf;
    float NoL;
    float pdf;
    if (pickDiffuse)
    {
        rd = cosine_sample_hemisphere(sf.normalWS, u.x, sinPhi, cosPhi);
        NoL = saturate(dot(sf.normalWS, rd));
        brdf = lambert_brdf(sf.material.albedo);
        pdf = NoL * kInvPi; // sampling.glsl:35
        pdf *= diffuseWeight;
    }
    else
    {
        rd = sample_visible_trowbridge_reitz(vInBasis, alpha, u.x, sinPhi, cosPhi);
        rd = basis.to_world(rd);
        NoL = saturate(dot(sf.normalWS, rd));
        const f3 h = normalize(sf.invViewRayWS + rd);
        const float NoH = saturate(dot(sf.normalWS, h));
        const float VoH = saturate(dot(sf.invViewRayWS, h));
        brdf = cook_torrance_brdf(NoL, sf.NoV, NoH, VoH, fresnel_zero(sf), sf.material.roughness);
        pdf = visible_trowbridge_reitz_pdf(vInBasis, basis.to_local(rd), alpha);
        pdf *= specularWeight;
    }
    const f3 w = (brdf * NoL) / pdf;
    throughput = f3{throughput.x * fmax_(w.x, 0.0f), throughput.y * fmax_(w.y, 0.0f), throughput.z * fmax_(w.z, 0.0f)};
}

// A path whose throughput is exactly (0, 0, 0) after importanceSampleBounce ends there (arithmetic contract,
// DESIGN.md): every later term of main.rgen:241-283 is throughput * X, i.e. +0 for any finite X - and a quarter
// to a half of all bounce rays carry such a throughput (a sampled direction at or below the shading horizon has
// NoL = 0).  NaN components compare unequal to zero: such a path goes on and poisons its pixel as in the GLSL.
PPT_D bool throughput_is_zero(f3 t) { return t.x == 0.0f && t.y == 0.0f && t.z == 0.0f; }

// First half of evaluateDirectLighting (main.rgen:195-214): pick a light, evaluate it.  Returns
// true when a shadow ray towards `l` of length `d` must be traced; `contribution` is then
// throughput * irradiance * lightCount * BRDF (everything but the visibility term).
template <bool COUNT>
PPT_D bool prepare_direct_lighting(
    const DeviceScene &s, const Surface &sf, f3 throughput, Rng &rng, f3 &l, float &d, f3 &irradiance,
    LaneCounters &cnt)
{
    if (sf.material.alpha == 0.0f) return false;
    const uint32_t lightCount = 1u + s.pointLightCount + s.spotLightCount;
    uint32_t lightIndex = f2uint(rng.rnd01() * (float)lightCount);
    if (lightIndex > lightCount - 1u) lightIndex = lightCount - 1u;
    const bool spot = sample_light(s, sf.positionWS, lightIndex, l, d, irradiance);
    if constexpr (COUNT)
    {
        if (spot)
            cnt.spotLightSamples++;
        else
            cnt.lightSamples++;
    }
    (void)spot;
    (void)throughput;
    return dot(l, sf.normalWS) > 0.0f;
}

// Second half (main.rgen:216-222): apply visibility, the uniform-pick weight and the BRDF.
// `brdfTimesNoL` = evalBRDFTimesNoL(l, surface).
PPT_D f3 direct_lighting_value(const DeviceScene &s, f3 throughput, f3 irradiance, f3 brdfTimesNoL, float visibility)
{
    const uint32_t lightCount = 1u + s.pointLightCount + s.spotLightCount;
    irradiance = irradiance * visibility;
    irradiance = irradiance * (float)lightCount;
    return (throughput * irradiance) * brdfTimesNoL;
}
// A shadow ray only has to be traced when it can change the radiance bits.  `lit` / `blocked` are the
// direct term with visibility 1 / 0 (main.rgen:216-222 multiplies the irradiance by the visibility, so
// `blocked` is +-0 per channel unless a factor is non-finite).  When both are +-0 in every channel,
// `color += direct` leaves the accumulator unchanged whatever the traversal finds (it starts at +0
// and x + (+-0) == x for every x that is not -0, which a sum that started at +0 never is): lights
// with zero radiance, samples outside a spot cone or beyond a light's range cost no traversal.
PPT_D bool shadow_ray_matters(f3 lit, f3 blocked)
{
    const bool litZero = lit.x == 0.0f && lit.y == 0.0f && lit.z == 0.0f;
    const bool blockedZero = blocked.x == 0.0f && blocked.y == 0.0f && blocked.z == 0.0f;
    return !(litZero && blockedZero);
}
PPT_D f3 finish_direct_lighting(const DeviceScene &s, const Surface &sf, f3 throughput, f3 l, f3 irradiance, float visibility)
{
    return direct_lighting_value(s, throughput, irradiance, eval_brdf_times_nol(l, sf), visibility);
}

} // namespace ppt
