/*
 * oracle/oracle.c — TEST INFRASTRUCTURE.  See oracle.h for scope and parity status
 * ("parity unpinned" against the Vulkan original; pinned by KATs in tests/golden/).
 *
 * CPU restatement of prosper's path-tracing reference pass.  Citations are file:line relative to
 * /root/reference/.  Arithmetic follows the contract in ora_math.h.
 */
#include "oracle.h"

#include <stdio.h>
#include <stdlib.h>

#ifdef _OPENMP
#include <omp.h>
#endif

#include "ora_math.h"

/* ------------------------------------------------------------------------------------------
 * F8  RNG — res/shader/common/random.glsl
 * ---------------------------------------------------------------------------------------- */

/* random.glsl:7-12 */
uint32_t ora_pcg(uint32_t v)
{
    const uint32_t state = v * 747796405u + 2891336453u;
    const uint32_t word = ((state >> ((state >> 28) + 4u)) ^ state) * 277803737u;
    return (word >> 22) ^ word;
}

/* random.glsl:17-28 */
void ora_pcg3d(uint32_t v[3])
{
    uint32_t x = v[0] * 1664525u + 1013904223u;
    uint32_t y = v[1] * 1664525u + 1013904223u;
    uint32_t z = v[2] * 1664525u + 1013904223u;
    x += y * z;
    y += z * x;
    z += x * y;
    x ^= x >> 16;
    y ^= y >> 16;
    z ^= z >> 16;
    x += y * z;
    y += z * x;
    z += x * y;
    v[0] = x;
    v[1] = y;
    v[2] = z;
}

/* random.glsl:42 — float(0xFFFFFFFFu) rounds to 2^32, so 1.0 is reachable */
static inline float rng_to01(uint32_t u) { return (float)u / 4294967296.0f; }

typedef struct ora_rng { uint32_t s[3]; } ora_rng;

/* random.glsl:50-58 */
static inline float rnd01(ora_rng *r)
{
    ora_pcg3d(r->s);
    return rng_to01(r->s[0]);
}
/* random.glsl:59-63 */
static inline ora_v2 rnd2d01(ora_rng *r)
{
    ora_pcg3d(r->s);
    ora_v2 o = {rng_to01(r->s[0]), rng_to01(r->s[1])};
    return o;
}

/* random.glsl:30-40 */
static ora_v3 uint_to_color(uint32_t x)
{
    const uint32_t xr = ora_pcg(x);
    const uint32_t r = (xr >> 20) & 0x3FFu;
    const uint32_t g = (xr >> 10) & 0x3FFu;
    const uint32_t b = xr & 0x3FFu;
    const float k = 1.0f / 1023.0f;
    return ora_v3_make((float)r * k, (float)g * k, (float)b * k);
}

/* ------------------------------------------------------------------------------------------
 * Scene container
 * ---------------------------------------------------------------------------------------- */

typedef struct ora_tri
{
    ora_v3 v0, v1, v2;
    uint32_t drawInstance;
    uint32_t primitive;
    uint32_t opaque;
} ora_tri;

typedef struct ora_node
{
    float lo[3], hi[3];
    int32_t left;   /* inner: index of left child, right = left + 1; leaf: -1 */
    uint32_t first; /* leaf: first triangle */
    uint32_t count; /* leaf: triangle count */
} ora_node;

struct ora_scene
{
    prosper_pt_scene_view view;
    ora_tri *tris;
    uint64_t triCount;
    ora_node *nodes;
    uint32_t nodeCount;
    int bruteForce;
    int literalGlsl; /* ora_scene_set_literal_glsl: main.rgen:241-283 as written, no zero-throughput rule */
};

/* ------------------------------------------------------------------------------------------
 * F10  bindless fetch + decode — res/shader/scene/geometry.glsl
 * ---------------------------------------------------------------------------------------- */

typedef struct ora_vertex
{
    ora_v3 position;
    ora_v3 normal;
    ora_v4 tangent;
    ora_v2 uv;
} ora_vertex;

static inline const uint32_t *geo_u32(const ora_scene *s, uint32_t buffer)
{
    return (const uint32_t *)s->view.geometryBuffers[buffer];
}

/* geometry.glsl:51-59 */
static inline uint32_t load_index(const ora_scene *s, const prosper_GeometryMetadata *m, uint32_t index)
{
    if (m->usesShortIndices == 1)
        return (uint32_t)((const uint16_t *)s->view.geometryBuffers[m->bufferIndex])[m->indicesOffset + index];
    return geo_u32(s, m->bufferIndex)[m->indicesOffset + index];
}

/* geometry.glsl:71-80 */
static inline ora_v2 load_r16g16(const ora_scene *s, uint32_t buffer, uint32_t offset, uint32_t index)
{
    ora_v2 r = {0.0f, 0.0f};
    if (offset == PROSPER_PT_ABSENT) return r;
    const uint32_t packed = geo_u32(s, buffer)[offset + index];
    r.x = ora_half_to_float((uint16_t)(packed & 0xFFFFu));
    r.y = ora_half_to_float((uint16_t)(packed >> 16));
    return r;
}

/* geometry.glsl:82-93 */
static inline ora_v3 load_r16g16b16a16(const ora_scene *s, uint32_t buffer, uint32_t offset, uint32_t index)
{
    if (offset == PROSPER_PT_ABSENT) return ora_v3_make(0.0f, 0.0f, 0.0f);
    const uint32_t p0 = geo_u32(s, buffer)[offset + index * 2];
    const uint32_t p1 = geo_u32(s, buffer)[offset + index * 2 + 1];
    return ora_v3_make(
        ora_half_to_float((uint16_t)(p0 & 0xFFFFu)), ora_half_to_float((uint16_t)(p0 >> 16)),
        ora_half_to_float((uint16_t)(p1 & 0xFFFFu)));
}

/* geometry.glsl:95-103 */
static ora_v3 unpack_snorm_r10g10b10(uint32_t packed)
{
    const int32_t sx = (int32_t)(packed << 22) >> 22;
    const int32_t sy = (int32_t)(packed << 12) >> 22;
    const int32_t sz = (int32_t)(packed << 2) >> 22;
    const float k = 1.0f / 511.0f;
    const ora_v3 v =
        ora_v3_make(ora_max((float)sx * k, -1.0f), ora_max((float)sy * k, -1.0f), ora_max((float)sz * k, -1.0f));
    return ora_normalize(v);
}

/* geometry.glsl:105-114 */
static inline ora_v3 load_r10g10b10_snorm(const ora_scene *s, uint32_t buffer, uint32_t offset, uint32_t index)
{
    if (offset == PROSPER_PT_ABSENT) return ora_v3_make(0.0f, 0.0f, 0.0f);
    return unpack_snorm_r10g10b10(geo_u32(s, buffer)[offset + index]);
}

/* geometry.glsl:116-127 */
static inline ora_v4 load_tangent_with_sign(const ora_scene *s, uint32_t buffer, uint32_t offset, uint32_t index)
{
    ora_v4 r = {0.0f, 0.0f, 0.0f, 0.0f};
    if (offset == PROSPER_PT_ABSENT) return r;
    const uint32_t packed = geo_u32(s, buffer)[offset + index];
    const ora_v3 t = unpack_snorm_r10g10b10(packed);
    r.x = t.x;
    r.y = t.y;
    r.z = t.z;
    r.w = (float)((int32_t)packed >> 30);
    return r;
}

/* geometry.glsl:220-244 */
static ora_vertex load_vertex_through_index_buffer(
    const ora_scene *s, const prosper_GeometryMetadata *m, uint32_t index)
{
    const uint32_t vi = load_index(s, m, index);
    ora_vertex v;
    v.position = load_r16g16b16a16(s, m->bufferIndex, m->positionsOffset, vi);
    v.normal = load_r10g10b10_snorm(s, m->bufferIndex, m->normalsOffset, vi);
    v.tangent = load_tangent_with_sign(s, m->bufferIndex, m->tangentsOffset, vi);
    v.uv = load_r16g16(s, m->bufferIndex, m->texCoord0sOffset, vi);
    return v;
}

/* geometry.glsl:246-256 */
static ora_v2 load_uv(const ora_scene *s, uint32_t meshIndex, uint32_t index)
{
    const prosper_GeometryMetadata *m = &s->view.geometryMetadatas[meshIndex];
    const uint32_t vi = load_index(s, m, index);
    return load_r16g16(s, m->bufferIndex, m->texCoord0sOffset, vi);
}

/* geometry.glsl:258-270: v0*a + v1*b + v2*c */
static inline float bary1(float v0, float v1, float v2, float a, float b, float c)
{
    return fmaf(v2, c, fmaf(v1, b, v0 * a));
}

/* geometry.glsl:271-295 */
static ora_vertex interpolate_vertex(const ora_vertex *v0, const ora_vertex *v1, const ora_vertex *v2, ora_v2 bc)
{
    const float a = (1.0f - bc.x) - bc.y;
    const float b = bc.x;
    const float c = bc.y;
    ora_vertex r;
    r.position = ora_v3_make(
        bary1(v0->position.x, v1->position.x, v2->position.x, a, b, c),
        bary1(v0->position.y, v1->position.y, v2->position.y, a, b, c),
        bary1(v0->position.z, v1->position.z, v2->position.z, a, b, c));
    r.normal = ora_v3_make(
        bary1(v0->normal.x, v1->normal.x, v2->normal.x, a, b, c),
        bary1(v0->normal.y, v1->normal.y, v2->normal.y, a, b, c),
        bary1(v0->normal.z, v1->normal.z, v2->normal.z, a, b, c));
    r.tangent.x = bary1(v0->tangent.x, v1->tangent.x, v2->tangent.x, a, b, c);
    r.tangent.y = bary1(v0->tangent.y, v1->tangent.y, v2->tangent.y, a, b, c);
    r.tangent.z = bary1(v0->tangent.z, v1->tangent.z, v2->tangent.z, a, b, c);
    r.tangent.w = bary1(v0->tangent.w, v1->tangent.w, v2->tangent.w, a, b, c);
    r.uv.x = bary1(v0->uv.x, v1->uv.x, v2->uv.x, a, b, c);
    r.uv.y = bary1(v0->uv.y, v1->uv.y, v2->uv.y, a, b, c);
    return r;
}

/* ------------------------------------------------------------------------------------------
 * F11  object -> world — res/shader/scene/instances.glsl:36-53
 * ---------------------------------------------------------------------------------------- */

/* vec4(p,1) * mat3x4: component i = dot(vec4(p,1), column i) */
static inline ora_v3 mul_point_mat3x4(ora_v3 p, const prosper_mat3x4 *m)
{
    return ora_v3_make(
        fmaf(p.z, m->col[0].z, fmaf(p.y, m->col[0].y, fmaf(p.x, m->col[0].x, m->col[0].w))),
        fmaf(p.z, m->col[1].z, fmaf(p.y, m->col[1].y, fmaf(p.x, m->col[1].x, m->col[1].w))),
        fmaf(p.z, m->col[2].z, fmaf(p.y, m->col[2].y, fmaf(p.x, m->col[2].x, m->col[2].w))));
}
/* v * mat3(m): component i = dot(v, column i .xyz) */
static inline ora_v3 mul_vec_mat3(ora_v3 v, const prosper_mat3x4 *m)
{
    return ora_v3_make(
        fmaf(v.z, m->col[0].z, fmaf(v.y, m->col[0].y, v.x * m->col[0].x)),
        fmaf(v.z, m->col[1].z, fmaf(v.y, m->col[1].y, v.x * m->col[1].x)),
        fmaf(v.z, m->col[2].z, fmaf(v.y, m->col[2].y, v.x * m->col[2].x)));
}

static ora_vertex transform_vertex(const ora_vertex *v, const prosper_ModelInstanceTransforms *t)
{
    ora_vertex r;
    r.position = mul_point_mat3x4(v->position, &t->modelToWorld);
    r.normal = ora_normalize(mul_vec_mat3(v->normal, &t->normalToWorld));
    if (v->tangent.w != 0.0f)
    {
        const ora_v3 tt =
            ora_normalize(mul_vec_mat3(ora_v3_make(v->tangent.x, v->tangent.y, v->tangent.z), &t->modelToWorld));
        r.tangent.x = tt.x;
        r.tangent.y = tt.y;
        r.tangent.z = tt.z;
        r.tangent.w = v->tangent.w;
    }
    else
        r.tangent = v->tangent;
    r.uv = v->uv;
    return r;
}

/* ------------------------------------------------------------------------------------------
 * Texture sampling (Vulkan 1.3 §16 texel filtering at LOD 0; see DESIGN.md "texture contract")
 * ---------------------------------------------------------------------------------------- */

/* texel index of integer coordinate i (64-bit so that the +1 neighbour of a saturated coordinate is
 * still the mathematical successor) */
static inline int32_t wrap_coord(int64_t i, int32_t size, uint32_t mode)
{
    if (mode == PROSPER_PT_WRAP_REPEAT)
    {
        int64_t m = i % size;
        return (int32_t)(m < 0 ? m + size : m);
    }
    if (mode == PROSPER_PT_WRAP_MIRRORED_REPEAT)
    {
        const int64_t period = 2 * (int64_t)size;
        int64_t m = i % period;
        if (m < 0) m += period;
        return (int32_t)(m < size ? m : period - 1 - m);
    }
    return (int32_t)(i < 0 ? 0 : (i >= size ? size - 1 : i));
}

static inline ora_v4 fetch_rgba8(const prosper_pt_texture_desc *t, int32_t i, int32_t j)
{
    const uint8_t *p = (const uint8_t *)t->texels + 4u * ((size_t)j * t->width + (size_t)i);
    const float k = 1.0f / 255.0f;
    ora_v4 r = {(float)p[0] * k, (float)p[1] * k, (float)p[2] * k, (float)p[3] * k};
    return r;
}

/* texture(sampler2D(materialTextures[tex], materialSamplers[smp]), uv) with implicit LOD 0 */
static ora_v4 sample_texture(const ora_scene *s, uint32_t tex, uint32_t smp, ora_v2 uv)
{
    const prosper_pt_texture_desc *t = &s->view.textures[tex];
    const prosper_pt_sampler_desc *sd = &s->view.samplers[smp];
    const int32_t w = (int32_t)t->width;
    const int32_t h = (int32_t)t->height;
    if (sd->magFilter == PROSPER_PT_FILTER_NEAREST)
    {
        const int32_t i = wrap_coord(ora_f2i(floorf(uv.x * (float)w)), w, sd->wrapS);
        const int32_t j = wrap_coord(ora_f2i(floorf(uv.y * (float)h)), h, sd->wrapT);
        return fetch_rgba8(t, i, j);
    }
    const float u = fmaf(uv.x, (float)w, -0.5f);
    const float v = fmaf(uv.y, (float)h, -0.5f);
    const float fu = floorf(u);
    const float fv = floorf(v);
    const float a = u - fu;
    const float b = v - fv;
    const int32_t i0 = wrap_coord(ora_f2i(fu), w, sd->wrapS);
    const int32_t i1 = wrap_coord((int64_t)ora_f2i(fu) + 1, w, sd->wrapS);
    const int32_t j0 = wrap_coord(ora_f2i(fv), h, sd->wrapT);
    const int32_t j1 = wrap_coord((int64_t)ora_f2i(fv) + 1, h, sd->wrapT);
    const ora_v4 t00 = fetch_rgba8(t, i0, j0);
    const ora_v4 t10 = fetch_rgba8(t, i1, j0);
    const ora_v4 t01 = fetch_rgba8(t, i0, j1);
    const ora_v4 t11 = fetch_rgba8(t, i1, j1);
    const float w00 = (1.0f - a) * (1.0f - b);
    const float w10 = a * (1.0f - b);
    const float w01 = (1.0f - a) * b;
    const float w11 = a * b;
    ora_v4 r;
    r.x = fmaf(w11, t11.x, fmaf(w01, t01.x, fmaf(w10, t10.x, w00 * t00.x)));
    r.y = fmaf(w11, t11.y, fmaf(w01, t01.y, fmaf(w10, t10.y, w00 * t00.y)));
    r.z = fmaf(w11, t11.z, fmaf(w01, t01.z, fmaf(w10, t10.z, w00 * t00.z)));
    r.w = fmaf(w11, t11.w, fmaf(w01, t01.w, fmaf(w10, t10.w, w00 * t00.w)));
    return r;
}

/* Cube face selection and (sc, tc, ma) per Vulkan 1.3 §16.5.4; faces +X,-X,+Y,-Y,+Z,-Z. */
static void cube_face_coords(ora_v3 d, uint32_t *face, float *sc, float *tc, float *ma)
{
    const float ax = ora_abs(d.x);
    const float ay = ora_abs(d.y);
    const float az = ora_abs(d.z);
    if (az >= ax && az >= ay)
    {
        *face = d.z < 0.0f ? 5u : 4u;
        *sc = d.z < 0.0f ? -d.x : d.x;
        *tc = -d.y;
        *ma = az;
    }
    else if (ay >= ax)
    {
        *face = d.y < 0.0f ? 3u : 2u;
        *sc = d.x;
        *tc = d.y < 0.0f ? -d.z : d.z;
        *ma = ay;
    }
    else
    {
        *face = d.x < 0.0f ? 1u : 0u;
        *sc = d.x < 0.0f ? d.z : -d.z;
        *tc = -d.y;
        *ma = ax;
    }
}

/* Direction through the point (sc, tc) of face `face`'s plane at unit distance. */
static ora_v3 cube_face_dir(uint32_t face, float sc, float tc)
{
    switch (face)
    {
    case 0: return ora_v3_make(1.0f, -tc, -sc);
    case 1: return ora_v3_make(-1.0f, -tc, sc);
    case 2: return ora_v3_make(sc, 1.0f, tc);
    case 3: return ora_v3_make(sc, -1.0f, -tc);
    case 4: return ora_v3_make(sc, -tc, 1.0f);
    default: return ora_v3_make(-sc, -tc, -1.0f);
    }
}

static inline ora_v3 fetch_cube_rgb(const ora_scene *s, uint32_t face, int32_t i, int32_t j)
{
    const int32_t n = (int32_t)s->view.skybox.faceSize;
    if (i < 0 || j < 0 || i >= n || j >= n)
    {
        /* Seamless edge: re-project the centre of the out-of-face texel onto the cube and take
         * the texel it lands in on the neighbouring face. */
        const float invN = 1.0f / (float)n;
        const float sc = fmaf(2.0f * ((float)i + 0.5f), invN, -1.0f);
        const float tc = fmaf(2.0f * ((float)j + 0.5f), invN, -1.0f);
        const ora_v3 d = cube_face_dir(face, sc, tc);
        float sc2, tc2, ma2;
        cube_face_coords(d, &face, &sc2, &tc2, &ma2);
        const float inv2 = 1.0f / ma2;
        const float ss = fmaf(0.5f, sc2 * inv2, 0.5f);
        const float tt = fmaf(0.5f, tc2 * inv2, 0.5f);
        i = ora_f2i(floorf(ss * (float)n));
        j = ora_f2i(floorf(tt * (float)n));
        i = i < 0 ? 0 : (i >= n ? n - 1 : i);
        j = j < 0 ? 0 : (j >= n ? n - 1 : j);
    }
    const uint16_t *p = s->view.skybox.texels + 4u * (((size_t)face * (size_t)n + (size_t)j) * (size_t)n + (size_t)i);
    return ora_v3_make(ora_half_to_float(p[0]), ora_half_to_float(p[1]), ora_half_to_float(p[2]));
}

/* F18: textureLod(skybox, d, 0).rgb — main.rgen:251, scene/skybox.glsl:4 */
static ora_v3 sample_skybox(const ora_scene *s, ora_v3 d)
{
    if (s->view.skybox.texels == NULL) return ora_v3_make(0.0f, 0.0f, 0.0f);
    const int32_t n = (int32_t)s->view.skybox.faceSize;
    uint32_t face;
    float sc, tc, ma;
    cube_face_coords(d, &face, &sc, &tc, &ma);
    const float invMa = 1.0f / ma;
    const float ss = fmaf(0.5f, sc * invMa, 0.5f);
    const float tt = fmaf(0.5f, tc * invMa, 0.5f);
    const float u = fmaf(ss, (float)n, -0.5f);
    const float v = fmaf(tt, (float)n, -0.5f);
    const float fu = floorf(u);
    const float fv = floorf(v);
    const float a = u - fu;
    const float b = v - fv;
    const int32_t i0 = ora_f2i(fu);
    const int32_t j0 = ora_f2i(fv);
    const ora_v3 t00 = fetch_cube_rgb(s, face, i0, j0);
    const ora_v3 t10 = fetch_cube_rgb(s, face, i0 + 1, j0);
    const ora_v3 t01 = fetch_cube_rgb(s, face, i0, j0 + 1);
    const ora_v3 t11 = fetch_cube_rgb(s, face, i0 + 1, j0 + 1);
    const float w00 = (1.0f - a) * (1.0f - b);
    const float w10 = a * (1.0f - b);
    const float w01 = (1.0f - a) * b;
    const float w11 = a * b;
    return ora_v3_make(
        fmaf(w11, t11.x, fmaf(w01, t01.x, fmaf(w10, t10.x, w00 * t00.x))),
        fmaf(w11, t11.y, fmaf(w01, t01.y, fmaf(w10, t10.y, w00 * t00.y))),
        fmaf(w11, t11.z, fmaf(w01, t01.z, fmaf(w10, t10.z, w00 * t00.z))));
}

/* ------------------------------------------------------------------------------------------
 * F12  materials — res/shader/scene/materials.glsl
 * ---------------------------------------------------------------------------------------- */

typedef struct ora_material
{
    ora_v3 albedo;
    ora_v3 normal;
    float roughness;
    float metallic;
    float alpha;
} ora_material;

/* materials.glsl:26-29 */
static inline float srgb_to_linear(float x)
{
    return x <= 0.04045f ? x * (1.0f / 12.92f) : ora_pow((x + 0.055f) * (1.0f / 1.055f), 2.4f);
}

/* materials.glsl:47-119 */
static ora_material sample_material(const ora_scene *s, uint32_t index, ora_v2 uv)
{
    const prosper_MaterialData *data = &s->view.materials[index];
    ora_material ret;
    /* fields the GLSL leaves unset on the early return are defined as zero here */
    ret.albedo = ora_v3_make(0.0f, 0.0f, 0.0f);
    ret.normal = ora_v3_make(0.0f, 0.0f, 0.0f);
    ret.roughness = 0.0f;
    ret.metallic = 0.0f;

    ora_v4 base = {1.0f, 1.0f, 1.0f, 1.0f};
    const uint32_t baseTex = data->baseColorTextureSampler & 0xFFFFFFu;
    const uint32_t baseSmp = data->baseColorTextureSampler >> 24;
    if (baseTex > 0)
    {
        const ora_v4 t = sample_texture(s, baseTex, baseSmp, uv);
        base.x = srgb_to_linear(t.x);
        base.y = srgb_to_linear(t.y);
        base.z = srgb_to_linear(t.z);
        base.w = t.w; /* materials.glsl:34-35: alpha is not converted */
    }
    base.x *= data->baseColorFactor.x;
    base.y *= data->baseColorFactor.y;
    base.z *= data->baseColorFactor.z;
    base.w *= data->baseColorFactor.w;

    if (data->alphaMode == PROSPER_ALPHA_MODE_BLEND)
        ret.alpha = base.w;
    else
    {
        if (data->alphaMode == PROSPER_ALPHA_MODE_MASK)
        {
            if (base.w < data->alphaCutoff)
            {
                ret.alpha = 0.0f;
                return ret;
            }
        }
        ret.alpha = -1.0f;
    }
    ret.albedo = ora_v3_make(base.x, base.y, base.z);

    const uint32_t mrTex = data->metallicRoughnessTextureSampler & 0xFFFFFFu;
    const uint32_t mrSmp = data->metallicRoughnessTextureSampler >> 24;
    if (mrTex > 0)
    {
        const ora_v4 mr = sample_texture(s, mrTex, mrSmp, uv);
        ret.roughness = mr.y * data->roughnessFactor;
        ret.metallic = mr.z * data->metallicFactor;
    }
    else
    {
        ret.roughness = data->roughnessFactor;
        ret.metallic = data->metallicFactor;
    }
    ret.roughness = ora_max(ret.roughness, 0.05f);

    const uint32_t nTex = data->normalTextureSampler & 0xFFFFFFu;
    const uint32_t nSmp = data->normalTextureSampler >> 24;
    if (nTex > 0)
    {
        const ora_v4 tn = sample_texture(s, nTex, nSmp, uv);
        ret.normal = ora_v3_make(fmaf(tn.x, 2.0f, -1.0f), fmaf(tn.y, 2.0f, -1.0f), fmaf(tn.z, 2.0f, -1.0f));
    }
    else
        ret.normal = ora_v3_make(-2.0f, -2.0f, -2.0f);
    return ret;
}

/* materials.glsl:121-147 */
static float sample_alpha(const ora_scene *s, uint32_t index, ora_v2 uv)
{
    const prosper_MaterialData *data = &s->view.materials[index];
    float linearAlpha = 1.0f;
    const uint32_t baseTex = data->baseColorTextureSampler & 0xFFFFFFu;
    const uint32_t baseSmp = data->baseColorTextureSampler >> 24;
    if (baseTex > 0) linearAlpha = srgb_to_linear(sample_texture(s, baseTex, baseSmp, uv).w);
    linearAlpha *= data->baseColorFactor.w;
    if (data->alphaMode == PROSPER_ALPHA_MODE_BLEND) return linearAlpha;
    if (data->alphaMode == PROSPER_ALPHA_MODE_MASK)
    {
        if (linearAlpha < data->alphaCutoff) return 0.0f;
    }
    return -1.0f;
}

/* ------------------------------------------------------------------------------------------
 * F15  BRDF — res/shader/brdf.glsl
 * ---------------------------------------------------------------------------------------- */

typedef struct ora_surface
{
    ora_v3 positionWS;
    ora_v3 normalWS;
    ora_v3 invViewRayWS;
    ora_v2 uv;
    float NoV;
    ora_material material;
} ora_surface;

/* brdf.glsl:12-19 */
static inline float trowbridge_reitz(float NoH, float alpha)
{
    const float a2 = alpha * alpha;
    const float denom = fmaf(NoH * NoH, a2 - 1.0f, 1.0f);
    return a2 / ((ORA_PI * denom) * denom);
}
/* brdf.glsl:21-24 */
static inline ora_v3 schlick_fresnel(float VoH, ora_v3 f0)
{
    const float p = ora_pow5(1.0f - VoH);
    return ora_v3_make(fmaf(1.0f - f0.x, p, f0.x), fmaf(1.0f - f0.y, p, f0.y), fmaf(1.0f - f0.z, p, f0.z));
}
/* brdf.glsl:35-43 */
static inline float schlick_trowbridge_reitz(float NoL, float NoV, float alpha)
{
    float k = alpha * 0.5f;
    k = ora_max(k, 0.0001f);
    const float gl = NoL / fmaf(NoL, 1.0f - k, k);
    const float gv = NoV / fmaf(NoV, 1.0f - k, k);
    return gl * gv;
}
/* brdf.glsl:46-58 */
static ora_v3 cook_torrance_brdf(float NoL, float NoV, float NoH, float VoH, ora_v3 f0, float roughness)
{
    const float alpha = roughness * roughness;
    const float D = trowbridge_reitz(NoH, alpha);
    const ora_v3 F = schlick_fresnel(VoH, f0);
    const float G = schlick_trowbridge_reitz(NoL, NoV, alpha);
    const float denom = fmaf(4.0f * NoL, NoV, 0.0001f);
    return ora_divs(ora_scale(ora_scale(F, D), G), denom);
}
/* brdf.glsl:60-64 */
static inline ora_v3 fresnel_zero(const ora_surface *sf)
{
    const float m = sf->material.metallic;
    return ora_v3_make(
        ora_mix(0.04f, sf->material.albedo.x, m), ora_mix(0.04f, sf->material.albedo.y, m),
        ora_mix(0.04f, sf->material.albedo.z, m));
}
/* brdf.glsl:9 */
static inline ora_v3 lambert_brdf(ora_v3 c) { return ora_scale(c, ORA_INV_PI); }

/* brdf.glsl:67-87 */
static ora_v3 eval_brdf_times_nol(ora_v3 l, const ora_surface *sf)
{
    const ora_v3 h = ora_normalize(ora_add(sf->invViewRayWS, l));
    const float NoL = ora_saturate(ora_dot(sf->normalWS, l));
    const float NoH = ora_saturate(ora_dot(sf->normalWS, h));
    const float VoH = ora_saturate(ora_dot(sf->invViewRayWS, h));
    const ora_v3 f0 = fresnel_zero(sf);
    const float m = sf->material.metallic;
    const float k = 0.96f; /* (1 - 0.04) folded by the GLSL front end */
    const ora_v3 cdiff = ora_v3_make(
        ora_mix(sf->material.albedo.x * k, 0.0f, m), ora_mix(sf->material.albedo.y * k, 0.0f, m),
        ora_mix(sf->material.albedo.z * k, 0.0f, m));
    const ora_v3 sum =
        ora_add(lambert_brdf(cdiff), cook_torrance_brdf(NoL, sf->NoV, NoH, VoH, f0, sf->material.roughness));
    return ora_scale(sum, NoL);
}

/* ------------------------------------------------------------------------------------------
 * F17  sampling — res/shader/common/sampling.glsl
 * ---------------------------------------------------------------------------------------- */

/* sampling.glsl:18-33 */
static ora_v3 cosine_sample_hemisphere(ora_v3 n, ora_v2 u)
{
    float a = fmaf(-2.0f, u.x, 1.0f);
    a *= 0.99999f;
    float b = sqrtf(fmaf(-a, a, 1.0f));
    b *= 0.99999f;
    const float phi = (2.0f * ORA_PI) * u.y;
    float sn, cs;
    ora_sincos(phi, &sn, &cs);
    return ora_normalize(ora_v3_make(fmaf(b, cs, n.x), fmaf(b, sn, n.y), n.z + a));
}

/* sampling.glsl:37-47: rows of the returned matrix are b1, b2, n */
typedef struct ora_onb { ora_v3 b1, b2, n; } ora_onb;
static ora_onb orthonormal_basis(ora_v3 n)
{
    const float s = ora_sign(n.z);
    const float a = -1.0f / (s + n.z);
    const float b = (n.x * n.y) * a;
    ora_onb o;
    o.b1 = ora_v3_make(fmaf((s * n.x) * n.x, a, 1.0f), s * b, (-s) * n.x);
    o.b2 = ora_v3_make(b, fmaf(n.y * n.y, a, s), -n.y);
    o.n = n;
    return o;
}
/* M * v with rows b1,b2,n (world -> local) */
static inline ora_v3 onb_to_local(const ora_onb *o, ora_v3 v)
{
    return ora_v3_make(ora_dot(o->b1, v), ora_dot(o->b2, v), ora_dot(o->n, v));
}
/* transpose(M) * v (local -> world): column-weighted sum */
static inline ora_v3 onb_to_world(const ora_onb *o, ora_v3 v)
{
    return ora_v3_make(
        fmaf(o->n.x, v.z, fmaf(o->b2.x, v.y, o->b1.x * v.x)), fmaf(o->n.y, v.z, fmaf(o->b2.y, v.y, o->b1.y * v.x)),
        fmaf(o->n.z, v.z, fmaf(o->b2.z, v.y, o->b1.z * v.x)));
}

/* sampling.glsl:53-79 */
static ora_v3 sample_visible_trowbridge_reitz(ora_v3 Ve, float alpha, ora_v2 Us)
{
    const ora_v3 Vh = ora_normalize(ora_v3_make(alpha * Ve.x, alpha * Ve.y, Ve.z));
    const float lensq = fmaf(Vh.y, Vh.y, Vh.x * Vh.x);
    ora_v3 T1;
    if (lensq > 0.0f)
    {
        const float inv = 1.0f / sqrtf(lensq);
        T1 = ora_v3_make(-Vh.y * inv, Vh.x * inv, 0.0f * inv);
    }
    else
        T1 = ora_v3_make(1.0f, 0.0f, 0.0f);
    const ora_v3 T2 = ora_cross(Vh, T1);
    const float r = sqrtf(Us.x);
    const float phi = (2.0f * ORA_PI) * Us.y;
    float sn, cs;
    ora_sincos(phi, &sn, &cs);
    const float t1 = r * cs;
    float t2 = r * sn;
    const float s = 0.5f * (1.0f + Vh.z);
    const float c1 = fmaf(-t1, t1, 1.0f);
    t2 = fmaf(s, t2, (1.0f - s) * sqrtf(c1));
    const float k = sqrtf(ora_max(0.0f, fmaf(-t2, t2, c1)));
    const ora_v3 Nh = ora_v3_make(
        fmaf(Vh.x, k, fmaf(T2.x, t2, T1.x * t1)), fmaf(Vh.y, k, fmaf(T2.y, t2, T1.y * t1)),
        fmaf(Vh.z, k, fmaf(T2.z, t2, T1.z * t1)));
    const ora_v3 Ne = ora_normalize(ora_v3_make(alpha * Nh.x, alpha * Nh.y, ora_max(0.0f, Nh.z)));
    return ora_reflect(ora_neg(Ve), Ne);
}

/* sampling.glsl:81-93 */
static float visible_trowbridge_reitz_pdf(ora_v3 Ve, ora_v3 Le, float alpha)
{
    const ora_v3 N = ora_v3_make(0.0f, 0.0f, 1.0f);
    const ora_v3 Ne = ora_normalize(ora_add(Ve, Le));
    const float NoV = ora_saturate(ora_dot(N, Ve));
    const float NoL = ora_saturate(ora_dot(N, Le));
    const float NoH = ora_saturate(ora_dot(N, Ne));
    const float VNDF = ((schlick_trowbridge_reitz(NoL, NoV, alpha) * NoV) * trowbridge_reitz(NoH, alpha)) / Ve.z;
    return VNDF / (4.0f * NoV);
}

/* ------------------------------------------------------------------------------------------
 * F14  lights — res/shader/scene/lighting.glsl
 * ---------------------------------------------------------------------------------------- */

/* lighting.glsl:15-37 */
static void eval_point_light(
    const prosper_PointLight *light, ora_v3 surfacePos, ora_v3 *l, float *d, ora_v3 *irradiance)
{
    const ora_v3 pos = ora_v3_make(light->position.x, light->position.y, light->position.z);
    const ora_v3 radiance =
        ora_v3_make(light->radianceAndRadius.x, light->radianceAndRadius.y, light->radianceAndRadius.z);
    const float radius = light->radianceAndRadius.w;
    const ora_v3 toLight = ora_sub(pos, surfacePos);
    const float d2 = ora_dot(toLight, toLight);
    *d = sqrtf(d2);
    *l = ora_divs(toLight, *d);
    const float dPerR = *d / radius;
    const float dPerR2 = dPerR * dPerR;
    const float dPerR4 = dPerR2 * dPerR2;
    const float att = ora_max(ora_min(1.0f - dPerR4, 1.0f), 0.0f);
    *irradiance = ora_divs(ora_scale(radiance, att), d2);
}

/* lighting.glsl:39-56 */
static void eval_spot_light(
    const prosper_SpotLight *light, ora_v3 surfacePos, ora_v3 *l, float *d, ora_v3 *irradiance)
{
    const ora_v3 pos = ora_v3_make(
        light->positionAndAngleOffset.x, light->positionAndAngleOffset.y, light->positionAndAngleOffset.z);
    const ora_v3 toLight = ora_sub(pos, surfacePos);
    const float d2 = ora_dot(toLight, toLight);
    *d = sqrtf(d2);
    *l = ora_divs(toLight, *d);
    const ora_v3 negDir = ora_v3_make(-light->direction.x, -light->direction.y, -light->direction.z);
    const float cd = ora_dot(negDir, *l);
    float att = ora_saturate(fmaf(cd, light->radianceAndAngleScale.w, light->positionAndAngleOffset.w));
    att *= att;
    const ora_v3 rad = ora_v3_make(
        light->radianceAndAngleScale.x, light->radianceAndAngleScale.y, light->radianceAndAngleScale.z);
    *irradiance = ora_divs(ora_scale(rad, att), d2);
}

/* lighting.glsl:58-89; returns 1 when a spot light was picked (for the byte model) */
static int sample_light(
    const ora_scene *s, ora_v3 surfacePos, uint32_t lightIndex, ora_v3 *l, float *d, ora_v3 *irradiance)
{
    const prosper_DirectionalLightParameters *sun = s->view.directionalLight;
    if (lightIndex == 0)
    {
        *l = ora_neg(ora_normalize(ora_v3_make(sun->direction.x, sun->direction.y, sun->direction.z)));
        *d = 100.0f;
        *irradiance = ora_v3_make(sun->irradiance.x, sun->irradiance.y, sun->irradiance.z);
        return 0;
    }
    lightIndex -= 1;
    if (lightIndex < s->view.pointLights->count)
    {
        eval_point_light(&s->view.pointLights->lights[lightIndex], surfacePos, l, d, irradiance);
        return 0;
    }
    lightIndex -= s->view.pointLights->count;
    if (lightIndex < s->view.spotLights->count)
    {
        eval_spot_light(&s->view.spotLights->lights[lightIndex], surfacePos, l, d, irradiance);
        return 1;
    }
    *l = ora_v3_make(0.0f, 1.0f, 0.0f);
    *d = 1.0f;
    *irradiance = ora_v3_make(0.0f, 0.0f, 0.0f);
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * F6/F7  rays — res/shader/rt/ray.glsl, scene/camera.glsl
 * ---------------------------------------------------------------------------------------- */

typedef struct ora_ray
{
    ora_v3 o, d;
    float tMin, tMax;
} ora_ray;

/* camera.glsl:46-51 */
static inline ora_v3 camera_world_fwd(const prosper_CameraUniforms *c)
{
    return ora_v3_make(-c->worldToCamera.col[0].z, -c->worldToCamera.col[1].z, -c->worldToCamera.col[2].z);
}

/* ray.glsl:15-43 */
static ora_ray pinhole_camera_ray(const prosper_CameraUniforms *c, ora_v2 uv)
{
    const float ndx = fmaf(uv.x, 2.0f, -1.0f);
    const float ndy = fmaf(uv.y, 2.0f, -1.0f);
    ora_ray ray;
    ray.o = ora_v3_make(c->eye.x, c->eye.y, c->eye.z);
    ray.tMin = 0.0f;
    ray.tMax = INFINITY;
    const float aspect = c->cameraToClip.col[1].y / c->cameraToClip.col[0].x;
    const float tanHalfFovY = 1.0f / c->cameraToClip.col[1].y;
    const ora_v3 right = ora_v3_make(c->worldToCamera.col[0].x, c->worldToCamera.col[1].x, c->worldToCamera.col[2].x);
    const ora_v3 up = ora_v3_make(c->worldToCamera.col[0].y, c->worldToCamera.col[1].y, c->worldToCamera.col[2].y);
    const ora_v3 fwd = camera_world_fwd(c);
    const ora_v3 tx = ora_scale(ora_scale(ora_scale(right, ndx), tanHalfFovY), aspect);
    const ora_v3 ty = ora_scale(ora_scale(up, ndy), tanHalfFovY);
    ray.d = ora_normalize(ora_add(ora_add(tx, ty), fwd));
    return ray;
}

/* ray.glsl:46-78 */
static ora_ray thin_lens_camera_ray(
    const prosper_CameraUniforms *c, ora_v2 uv, ora_v2 lensOffset, float apertureDiameter, float focusDistance,
    float focalLength)
{
    const ora_ray pin = pinhole_camera_ray(c, uv);
    const float theta = (lensOffset.x * 2.0f) * ORA_PI;
    const float radius = lensOffset.y;
    float sn, cs;
    ora_sincos(theta, &sn, &cs);
    const float u = cs * sqrtf(radius);
    const float v = sn * sqrtf(radius);
    const float k = focusDistance / ora_dot(pin.d, camera_world_fwd(c));
    const ora_v3 focusPoint = ora_v3_make(fmaf(pin.d.x, k, pin.o.x), fmaf(pin.d.y, k, pin.o.y), fmaf(pin.d.z, k, pin.o.z));
    const float fStop = focalLength / apertureDiameter;
    const float coc = focalLength / (2.0f * fStop);
    const ora_v3 lensPos = ora_add(
        ora_scale(ora_v3_make(1.0f, 0.0f, 0.0f), u * coc), ora_scale(ora_v3_make(0.0f, 1.0f, 0.0f), v * coc));
    /* (cameraToWorld * vec4(lensPos, 1)).xyz: columns weighted left to right */
    const prosper_mat4 *m = &c->cameraToWorld;
    ora_ray ray;
    ray.o = ora_v3_make(
        fmaf(m->col[2].x, lensPos.z, fmaf(m->col[1].x, lensPos.y, fmaf(m->col[0].x, lensPos.x, m->col[3].x))),
        fmaf(m->col[2].y, lensPos.z, fmaf(m->col[1].y, lensPos.y, fmaf(m->col[0].y, lensPos.x, m->col[3].y))),
        fmaf(m->col[2].z, lensPos.z, fmaf(m->col[1].z, lensPos.y, fmaf(m->col[0].z, lensPos.x, m->col[3].z))));
    ray.d = ora_normalize(ora_sub(focusPoint, ray.o));
    ray.tMin = 0.0f;
    ray.tMax = INFINITY;
    return ray;
}

/* ray.glsl:83-103 (Wächter & Binder) */
static inline float offset_component(float p, float n)
{
    const float origin = 1.0f / 32.0f;
    const float float_scale = 1.0f / 65536.0f;
    const float int_scale = 256.0f;
    const int32_t ofI = ora_f2i(int_scale * n);
    const int32_t bits = (int32_t)ora_f2u(p);
    const uint32_t moved = (uint32_t)bits + (uint32_t)((p < 0.0f) ? -ofI : ofI);
    const float pI = ora_u2f(moved);
    return ora_abs(p) < origin ? fmaf(float_scale, n, p) : pI;
}
static ora_v3 offset_ray(ora_v3 p, ora_v3 n)
{
    return ora_v3_make(offset_component(p.x, n.x), offset_component(p.y, n.y), offset_component(p.z, n.z));
}

/* ------------------------------------------------------------------------------------------
 * F2-F5  traversal.  prosper delegates this to the Vulkan driver (World.cpp:740,798;
 * main.rgen:57,73), so the arithmetic below is this build's definition (DESIGN.md "hit contract"):
 *   - triangles are the fp16-decoded positions (World.cpp:635-644) times the instance's
 *     modelToWorld, in world space;
 *   - edge functions are scalar triple products (exactly antisymmetric => watertight on shared
 *     edges), no culling, a candidate needs tMin < t < tMax (Vulkan 1.3 ray/triangle rule);
 *   - closest = smallest t, ties broken by the smaller (drawInstanceIndex, primitiveID);
 *   - non-opaque geometry runs the any-hit of rt/scene.rahit:18-39 on every candidate.
 * ---------------------------------------------------------------------------------------- */

/* 1 / d with |d| clamped to 1e-30: the per-axis reciprocal of a ray direction (box guard and slab tests) */
static inline float safe_rcp_dir(float d) { return 1.0f / (ora_abs(d) < 1e-30f ? (d < 0.0f ? -1e-30f : 1e-30f) : d); }

/* Hit contract (iii), the box guard.  The edge functions are sums of products of size |v - o|^2, so for a
 * small triangle far from the ray origin rounding lets them accept a ray that passes a little OUTSIDE the
 * triangle - possibly outside any bounding box an acceleration structure keeps for it, which would make
 * the set of hits depend on that structure.  The contract therefore bounds the acceptance zone by
 * construction: t must lie in the ray's parametric interval through the triangle's bounding box grown by
 * 2^-16 of its largest |coordinate|, within a factor 1 + 2^-18.  Any hierarchy whose boxes contain these
 * guard boxes and whose slab test is the same monotone arithmetic with a tolerance >= 1 + 2^-18 can then
 * never cull a valid candidate, so the hits are the same with a BVH, another BVH, or none. */
#define ORA_GUARD_PAD 1.52587890625e-05f  /* 2^-16 */
#define ORA_GUARD_TOL 1.000003814697265625f /* 1 + 2^-18 */
static int box_guard(ora_v3 o, ora_v3 d, ora_v3 v0, ora_v3 v1, ora_v3 v2, float tt)
{
    const float lox = fminf(fminf(v0.x, v1.x), v2.x), hix = fmaxf(fmaxf(v0.x, v1.x), v2.x);
    const float loy = fminf(fminf(v0.y, v1.y), v2.y), hiy = fmaxf(fmaxf(v0.y, v1.y), v2.y);
    const float loz = fminf(fminf(v0.z, v1.z), v2.z), hiz = fmaxf(fmaxf(v0.z, v1.z), v2.z);
    const float mx = fmaxf(ora_abs(lox), ora_abs(hix));
    const float my = fmaxf(ora_abs(loy), ora_abs(hiy));
    const float mz = fmaxf(ora_abs(loz), ora_abs(hiz));
    const float pad = fmaxf(fmaxf(mx, my), mz) * ORA_GUARD_PAD;
    const float ix = safe_rcp_dir(d.x), iy = safe_rcp_dir(d.y), iz = safe_rcp_dir(d.z);
    const float ax = ((lox - pad) - o.x) * ix, bx = ((hix + pad) - o.x) * ix;
    const float ay = ((loy - pad) - o.y) * iy, by = ((hiy + pad) - o.y) * iy;
    const float az = ((loz - pad) - o.z) * iz, bz = ((hiz + pad) - o.z) * iz;
    const float tn = fmaxf(fmaxf(fminf(ax, bx), fminf(ay, by)), fminf(az, bz));
    const float tf = fminf(fminf(fmaxf(ax, bx), fmaxf(ay, by)), fmaxf(az, bz));
    return tn <= tt * ORA_GUARD_TOL && tt <= tf * ORA_GUARD_TOL;
}

static int intersect_triangle(
    ora_v3 o, ora_v3 d, ora_v3 v0, ora_v3 v1, ora_v3 v2, float tMin, float tMax, float *t, float *bu, float *bv)
{
    const ora_v3 A = ora_sub(v0, o);
    const ora_v3 B = ora_sub(v1, o);
    const ora_v3 C = ora_sub(v2, o);
    const float U = ora_dot(d, ora_cross(C, B));
    const float V = ora_dot(d, ora_cross(A, C));
    const float W = ora_dot(d, ora_cross(B, A));
    if ((U < 0.0f || V < 0.0f || W < 0.0f) && (U > 0.0f || V > 0.0f || W > 0.0f)) return 0;
    const float det = (U + V) + W;
    if (!(det != 0.0f)) return 0; /* zero or NaN */
    /* Distance = the barycentric point's projection on the (unit) ray direction:
     * t = (U*(A.d) + V*(B.d) + W*(C.d)) / det.  Unlike the plane equation dot(A,N)/dot(d,N) this stays
     * well conditioned for grazing rays: the point is a convex combination of the vertices, so t always
     * lies inside the ray's interval through the triangle's bounding box and hit selection cannot depend
     * on which acceleration structure (or none) enumerated the candidates. */
    const float inv = 1.0f / det;
    const float tt = fmaf(W, ora_dot(C, d), fmaf(V, ora_dot(B, d), U * ora_dot(A, d))) * inv;
    if (!(tt > tMin && tt < tMax)) return 0;
    if (!box_guard(o, d, v0, v1, v2, tt)) return 0;
    *t = tt;
    *bu = V * inv;
    *bv = W * inv;
    return 1;
}

typedef struct ora_hit
{
    uint32_t drawInstance;
    uint32_t primitive;
    ora_v2 bary;
    float t;
} ora_hit;
#define ORA_MISS_INDEX 0xFFFFFFFFu

/* rt/scene.rahit:18-39; returns 1 = accept */
static int any_hit(const ora_scene *s, uint32_t drawInstance, uint32_t primitive, ora_v2 bary, uint32_t randomSeed)
{
    const prosper_DrawInstance *inst = &s->view.drawInstances[drawInstance];
    const ora_v2 uv0 = load_uv(s, inst->meshIndex, primitive * 3 + 0);
    const ora_v2 uv1 = load_uv(s, inst->meshIndex, primitive * 3 + 1);
    const ora_v2 uv2 = load_uv(s, inst->meshIndex, primitive * 3 + 2);
    const float a = (1.0f - bary.x) - bary.y;
    ora_v2 uv;
    uv.x = bary1(uv0.x, uv1.x, uv2.x, a, bary.x, bary.y);
    uv.y = bary1(uv0.y, uv1.y, uv2.y, a, bary.x, bary.y);
    const float alpha = sample_alpha(s, inst->materialIndex, uv);
    if (alpha == 0.0f) return 0;
    if (alpha > 0.0f)
    {
        const float u = (float)ora_pcg(randomSeed) / 4294967296.0f;
        if (u > alpha) return 0;
    }
    return 1;
}

static inline void consider_triangle(
    const ora_scene *s, const ora_tri *tri, ora_v3 o, ora_v3 d, float tMin, float rayTMax, uint32_t seed,
    int anyTerminate, ora_hit *best, int *occluded)
{
    float t, bu, bv;
    if (!intersect_triangle(o, d, tri->v0, tri->v1, tri->v2, tMin, rayTMax, &t, &bu, &bv)) return;
    if (!anyTerminate && best->drawInstance != ORA_MISS_INDEX)
    {
        /* closest-hit rule: smaller t wins, equal t goes to the smaller (instance, primitive) */
        if (t > best->t) return;
        const uint64_t key = ((uint64_t)tri->drawInstance << 32) | tri->primitive;
        const uint64_t bestKey = ((uint64_t)best->drawInstance << 32) | best->primitive;
        if (t == best->t && !(key < bestKey)) return;
    }
    ora_v2 bary = {bu, bv};
    if (!tri->opaque && !any_hit(s, tri->drawInstance, tri->primitive, bary, seed)) return;
    if (anyTerminate)
    {
        *occluded = 1;
        return;
    }
    best->drawInstance = tri->drawInstance;
    best->primitive = tri->primitive;
    best->bary = bary;
    best->t = t;
}

/* Slab test of the oracle's BVH: the same (plane - o) * invd arithmetic as the box guard (monotone in the
 * plane, so a box that contains a guard box yields an interval that contains the guard's), tolerance
 * 1 + 2^-16 > (1 + 2^-18)^2 on the far side. */
static inline int ray_box(const float lo[3], const float hi[3], ora_v3 o, ora_v3 invd, float tMin, float tMax)
{
    const float ax = (lo[0] - o.x) * invd.x, bx = (hi[0] - o.x) * invd.x;
    const float ay = (lo[1] - o.y) * invd.y, by = (hi[1] - o.y) * invd.y;
    const float az = (lo[2] - o.z) * invd.z, bz = (hi[2] - o.z) * invd.z;
    const float tn = fmaxf(fmaxf(fminf(ax, bx), fminf(ay, by)), fminf(az, bz));
    const float tf = fminf(fminf(fmaxf(ax, bx), fmaxf(ay, by)), fmaxf(az, bz));
    return fmaxf(tn, tMin) <= fminf(tf, tMax) * 1.0000152587890625f;
}

/* Shared driver of traceClosest (anyTerminate = 0) and shadow (anyTerminate = 1). */
static int trace(
    const ora_scene *s, ora_v3 o, ora_v3 d, float tMin, float tMaxIn, uint32_t seed, int anyTerminate, ora_hit *hit)
{
    ora_hit best;
    best.drawInstance = ORA_MISS_INDEX;
    best.primitive = ORA_MISS_INDEX;
    best.bary.x = 0.0f;
    best.bary.y = 0.0f;
    best.t = tMaxIn;
    int occluded = 0;
    if (s->bruteForce || s->nodeCount == 0)
    {
        for (uint64_t i = 0; i < s->triCount && !occluded; ++i)
            consider_triangle(s, &s->tris[i], o, d, tMin, tMaxIn, seed, anyTerminate, &best, &occluded);
    }
    else
    {
        ora_v3 invd;
        invd.x = safe_rcp_dir(d.x);
        invd.y = safe_rcp_dir(d.y);
        invd.z = safe_rcp_dir(d.z);
        uint32_t stack[128];
        int sp = 0;
        stack[sp++] = 0;
        while (sp > 0 && !occluded)
        {
            const ora_node *n = &s->nodes[stack[--sp]];
            if (!ray_box(n->lo, n->hi, o, invd, tMin, best.t)) continue;
            if (n->left < 0)
            {
                for (uint32_t i = 0; i < n->count && !occluded; ++i)
                    consider_triangle(
                        s, &s->tris[n->first + i], o, d, tMin, tMaxIn, seed, anyTerminate, &best, &occluded);
            }
            else
            {
                stack[sp++] = (uint32_t)n->left;
                stack[sp++] = (uint32_t)n->left + 1u;
            }
        }
    }
    if (anyTerminate) return occluded;
    *hit = best;
    return best.drawInstance != ORA_MISS_INDEX;
}

/* ---- oracle BVH: median split over centroids, leaves of <= 4 triangles ---- */

static void tri_bounds(const ora_tri *t, float lo[3], float hi[3])
{
    const float xs[3] = {t->v0.x, t->v1.x, t->v2.x};
    const float ys[3] = {t->v0.y, t->v1.y, t->v2.y};
    const float zs[3] = {t->v0.z, t->v1.z, t->v2.z};
    lo[0] = fminf(xs[0], fminf(xs[1], xs[2]));
    hi[0] = fmaxf(xs[0], fmaxf(xs[1], xs[2]));
    lo[1] = fminf(ys[0], fminf(ys[1], ys[2]));
    hi[1] = fmaxf(ys[0], fmaxf(ys[1], ys[2]));
    lo[2] = fminf(zs[0], fminf(zs[1], zs[2]));
    hi[2] = fmaxf(zs[0], fmaxf(zs[1], zs[2]));
}

static int g_sort_axis;
static int cmp_centroid(const void *a, const void *b)
{
    const ora_tri *ta = (const ora_tri *)a;
    const ora_tri *tb = (const ora_tri *)b;
    float ca, cb;
    if (g_sort_axis == 0)
    {
        ca = ta->v0.x + ta->v1.x + ta->v2.x;
        cb = tb->v0.x + tb->v1.x + tb->v2.x;
    }
    else if (g_sort_axis == 1)
    {
        ca = ta->v0.y + ta->v1.y + ta->v2.y;
        cb = tb->v0.y + tb->v1.y + tb->v2.y;
    }
    else
    {
        ca = ta->v0.z + ta->v1.z + ta->v2.z;
        cb = tb->v0.z + tb->v1.z + tb->v2.z;
    }
    return (ca > cb) - (ca < cb);
}

static void build_node(ora_scene *s, uint32_t nodeIndex, uint32_t first, uint32_t count)
{
    ora_node *n = &s->nodes[nodeIndex];
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    float clo[3] = {INFINITY, INFINITY, INFINITY}, chi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (uint32_t i = 0; i < count; ++i)
    {
        float l[3], h[3];
        tri_bounds(&s->tris[first + i], l, h);
        for (int k = 0; k < 3; ++k)
        {
            lo[k] = fminf(lo[k], l[k]);
            hi[k] = fmaxf(hi[k], h[k]);
            const float c = 0.5f * (l[k] + h[k]);
            clo[k] = fminf(clo[k], c);
            chi[k] = fmaxf(chi[k], c);
        }
    }
    /* the node box must contain the guard box of every triangle below it (box_guard: the triangle's
     * bounds grown by 2^-16 of its largest |coordinate|); 1.6e-5 > 2^-16 of the node's largest
     * |coordinate| does, and the slab arithmetic is monotone in the planes, so nothing else is needed */
    float mall = 0.0f;
    for (int k = 0; k < 3; ++k) mall = fmaxf(mall, fmaxf(fabsf(lo[k]), fabsf(hi[k])));
    for (int k = 0; k < 3; ++k)
    {
        const float pad = 1.6e-5f * mall + 1e-30f;
        n->lo[k] = lo[k] - pad;
        n->hi[k] = hi[k] + pad;
    }
    if (count <= 4)
    {
        n->left = -1;
        n->first = first;
        n->count = count;
        return;
    }
    int axis = 0;
    if (chi[1] - clo[1] > chi[axis] - clo[axis]) axis = 1;
    if (chi[2] - clo[2] > chi[axis] - clo[axis]) axis = 2;
    g_sort_axis = axis;
    qsort(&s->tris[first], count, sizeof(ora_tri), cmp_centroid);
    const uint32_t half = count / 2;
    const uint32_t left = s->nodeCount;
    s->nodeCount += 2;
    n->left = (int32_t)left;
    n->first = 0;
    n->count = 0;
    build_node(s, left, first, half);
    build_node(s, left + 1, first + half, count - half);
}

ora_scene *ora_scene_create(const prosper_pt_scene_view *view, int brute_force)
{
    ora_scene *s = (ora_scene *)calloc(1, sizeof(ora_scene));
    if (!s) return NULL;
    s->view = *view;
    s->bruteForce = brute_force;
    /* A mesh that is still loading has GeometryMetadata::bufferIndex 0xFFFFFFFF; World::buildNextBlas builds a model's BLAS
     * only once ALL its sub-meshes are there (World.cpp:598-606) and a TLAS instance without a BLAS is inactive
     * (accelerationStructureReference 0, World.cpp:909-915): rays pass through it.  A TLAS instance is a run of draw
     * instances with one modelInstanceIndex (World.cpp:480-513, 878-928). */
    uint8_t *active = (uint8_t *)calloc(view->drawInstanceCount ? view->drawInstanceCount : 1, 1);
    for (uint32_t di = 0; di < view->drawInstanceCount;)
    {
        const uint32_t mi = view->drawInstances[di].modelInstanceIndex;
        uint32_t end = di;
        int complete = 1;
        for (; end < view->drawInstanceCount && view->drawInstances[end].modelInstanceIndex == mi; ++end)
            if (view->geometryMetadatas[view->drawInstances[end].meshIndex].bufferIndex == 0xFFFFFFFFu) complete = 0;
        for (; di < end; ++di) active[di] = (uint8_t)complete;
    }
    uint64_t total = 0;
    for (uint32_t di = 0; di < view->drawInstanceCount; ++di)
        if (active[di]) total += view->meshInfos[view->drawInstances[di].meshIndex].indexCount / 3;
    s->triCount = total;
    s->tris = (ora_tri *)malloc(sizeof(ora_tri) * (total ? total : 1));
    uint64_t w = 0;
    for (uint32_t di = 0; di < view->drawInstanceCount; ++di)
    {
        if (!active[di]) continue;
        const prosper_DrawInstance *inst = &view->drawInstances[di];
        const prosper_GeometryMetadata *m = &view->geometryMetadatas[inst->meshIndex];
        const prosper_pt_mesh_info *info = &view->meshInfos[inst->meshIndex];
        const prosper_ModelInstanceTransforms *trfn = &view->modelInstanceTransforms[inst->modelInstanceIndex];
        /* World.cpp:646-651: geometry is opaque iff the mesh's material is AlphaMode_Opaque */
        const uint32_t opaque = view->materials[info->materialIndex].alphaMode == PROSPER_ALPHA_MODE_OPAQUE;
        const uint32_t triCount = info->indexCount / 3;
        for (uint32_t p = 0; p < triCount; ++p)
        {
            ora_tri *t = &s->tris[w++];
            const uint32_t i0 = load_index(s, m, p * 3 + 0);
            const uint32_t i1 = load_index(s, m, p * 3 + 1);
            const uint32_t i2 = load_index(s, m, p * 3 + 2);
            t->v0 = mul_point_mat3x4(load_r16g16b16a16(s, m->bufferIndex, m->positionsOffset, i0), &trfn->modelToWorld);
            t->v1 = mul_point_mat3x4(load_r16g16b16a16(s, m->bufferIndex, m->positionsOffset, i1), &trfn->modelToWorld);
            t->v2 = mul_point_mat3x4(load_r16g16b16a16(s, m->bufferIndex, m->positionsOffset, i2), &trfn->modelToWorld);
            t->drawInstance = di;
            t->primitive = p;
            t->opaque = opaque;
        }
    }
    free(active);
    if (!brute_force && total > 0)
    {
        s->nodes = (ora_node *)malloc(sizeof(ora_node) * (size_t)(2 * total + 2));
        s->nodeCount = 1;
        build_node(s, 0, 0, (uint32_t)total);
    }
    return s;
}

void ora_scene_destroy(ora_scene *s)
{
    if (!s) return;
    free(s->tris);
    free(s->nodes);
    free(s);
}

uint64_t ora_scene_triangle_count(const ora_scene *s) { return s->triCount; }
void ora_scene_set_literal_glsl(ora_scene *s, int on) { s->literalGlsl = on ? 1 : 0; }

int ora_trace_closest(
    const ora_scene *s, const float origin[3], const float dir[3], float tMin, float tMax, uint32_t randomSeed,
    uint32_t *drawInstanceIndex, uint32_t *primitiveID, float bary[2])
{
    ora_hit h;
    const int r = trace(
        s, ora_v3_make(origin[0], origin[1], origin[2]), ora_v3_make(dir[0], dir[1], dir[2]), tMin, tMax, randomSeed,
        0, &h);
    *drawInstanceIndex = h.drawInstance;
    *primitiveID = h.primitive;
    bary[0] = h.bary.x;
    bary[1] = h.bary.y;
    return r;
}

int ora_trace_shadow(
    const ora_scene *s, const float origin[3], const float dir[3], float tMin, float tMax, uint32_t randomSeed)
{
    ora_hit h;
    return trace(
        s, ora_v3_make(origin[0], origin[1], origin[2]), ora_v3_make(dir[0], dir[1], dir[2]), tMin, tMax, randomSeed,
        1, &h);
}

/* ------------------------------------------------------------------------------------------
 * F1, F9, F13, F16, F19, F20  the integrator — res/shader/rt/reference/main.rgen
 * ---------------------------------------------------------------------------------------- */

typedef struct ora_path_ctx
{
    const ora_scene *scene;
    const prosper_ReferencePC *pc;
    const prosper_CameraUniforms *camera;
    ora_rng rng;
    ora_counters *counters;
} ora_path_ctx;

/* main.rgen:37-45 */
static ora_v3 mapped_normal(ora_v3 tsn, ora_v3 normal, ora_v3 tangent, float sgn)
{
    const ora_v3 vB = ora_scale(ora_cross(normal, tangent), sgn);
    const ora_v3 sum = ora_add(ora_add(ora_scale(tangent, tsn.x), ora_scale(vB, tsn.y)), ora_scale(normal, tsn.z));
    return ora_normalize(sum);
}

/* main.rgen:49-60 */
static float shadow(ora_path_ctx *c, ora_v3 p, ora_v3 l, float tMin, float lDist)
{
    const uint32_t seed = ora_pcg(c->rng.s[0] ^ c->rng.s[1]);
    c->counters->shadowRays++;
    ora_hit h;
    return trace(c->scene, p, l, tMin, lDist, seed, 1, &h) ? 0.0f : 1.0f;
}

/* main.rgen:62-81 */
static ora_hit trace_closest(ora_path_ctx *c, const ora_ray *ray)
{
    const uint32_t seed = ora_pcg(c->rng.s[0] ^ c->rng.s[2]);
    c->counters->closestRays++;
    ora_hit h;
    trace(c->scene, ray->o, ray->d, ray->tMin, ray->tMax, seed, 0, &h);
    return h;
}

/* main.rgen:83-88 */
static void add_bounce(const prosper_ReferencePC *pc, ora_v3 *acc, ora_v3 color, uint32_t bounce)
{
    if (bounce > 0 && (pc->flags & PROSPER_PC_FLAG_CLAMP_INDIRECT))
        color = ora_v3_make(ora_clamp(color.x, 0.0f, 2.0f), ora_clamp(color.y, 0.0f, 2.0f), ora_clamp(color.z, 0.0f, 2.0f));
    *acc = ora_add(*acc, color);
}

/* main.rgen:90-144 */
static void importance_sample_bounce(ora_path_ctx *c, const ora_surface *sf, ora_v3 *throughput, ora_v3 *rd)
{
    const int specularOnly = sf->material.metallic > 0.999f;
    const float specularWeight = specularOnly ? 1.0f : 0.5f;
    const float diffuseWeight = 1.0f - specularWeight;
    const ora_onb basis = orthonormal_basis(sf->normalWS);
    const ora_v3 vInBasis = onb_to_local(&basis, sf->invViewRayWS);
    const float alpha = sf->material.roughness * sf->material.roughness;
    const int pickDiffuse = rnd01(&c->rng) < diffuseWeight;
    ora_v3 brdf;
    float NoL;
    float pdf;
    if (pickDiffuse)
    {
        *rd = cosine_sample_hemisphere(sf->normalWS, rnd2d01(&c->rng));
        NoL = ora_saturate(ora_dot(sf->normalWS, *rd));
        brdf = lambert_brdf(sf->material.albedo);
        pdf = NoL * ORA_INV_PI; /* sampling.glsl:35 */
        pdf *= diffuseWeight;
    }
    else
    {
        *rd = sample_visible_trowbridge_reitz(vInBasis, alpha, rnd2d01(&c->rng));
        *rd = onb_to_world(&basis, *rd);
        NoL = ora_saturate(ora_dot(sf->normalWS, *rd));
        const ora_v3 h = ora_normalize(ora_add(sf->invViewRayWS, *rd));
        const float NoH = ora_saturate(ora_dot(sf->normalWS, h));
        const float VoH = ora_saturate(ora_dot(sf->invViewRayWS, h));
        const ora_v3 f0 = fresnel_zero(sf);
        brdf = cook_torrance_brdf(NoL, sf->NoV, NoH, VoH, f0, sf->material.roughness);
        pdf = visible_trowbridge_reitz_pdf(vInBasis, onb_to_local(&basis, *rd), alpha);
        pdf *= specularWeight;
    }
    const ora_v3 w = ora_divs(ora_scale(brdf, NoL), pdf);
    throughput->x *= ora_max(w.x, 0.0f);
    throughput->y *= ora_max(w.y, 0.0f);
    throughput->z *= ora_max(w.z, 0.0f);
}

/* main.rgen:146-179 */
static ora_surface evaluate_surface(ora_path_ctx *c, const ora_ray *ray, const ora_hit *hit)
{
    const ora_scene *s = c->scene;
    const prosper_DrawInstance *inst = &s->view.drawInstances[hit->drawInstance];
    const prosper_ModelInstanceTransforms *trfn = &s->view.modelInstanceTransforms[inst->modelInstanceIndex];
    const prosper_GeometryMetadata *m = &s->view.geometryMetadatas[inst->meshIndex];
    const ora_vertex v0 = load_vertex_through_index_buffer(s, m, hit->primitive * 3 + 0);
    const ora_vertex v1 = load_vertex_through_index_buffer(s, m, hit->primitive * 3 + 1);
    const ora_vertex v2 = load_vertex_through_index_buffer(s, m, hit->primitive * 3 + 2);
    const ora_vertex vi = interpolate_vertex(&v0, &v1, &v2, hit->bary);
    const ora_vertex v = transform_vertex(&vi, trfn);
    c->counters->closestHits++;

    ora_surface sf;
    sf.positionWS = v.position;
    sf.invViewRayWS = ora_neg(ray->d);
    sf.uv = v.uv;
    sf.material = sample_material(s, inst->materialIndex, v.uv);
    if (sf.material.normal.x != -2.0f && v.tangent.w != 0.0f)
        sf.normalWS =
            mapped_normal(sf.material.normal, v.normal, ora_v3_make(v.tangent.x, v.tangent.y, v.tangent.z), v.tangent.w);
    else
        sf.normalWS = ora_normalize(v.normal);
    sf.NoV = ora_saturate(ora_dot(sf.normalWS, sf.invViewRayWS));
    return sf;
}

/* main.rgen:181-193 + debug.glsl:17-38 */
static ora_v3 debug_color(const ora_scene *s, uint32_t drawType, const ora_hit *hit, const ora_surface *sf)
{
    const prosper_DrawInstance *inst = &s->view.drawInstances[hit->drawInstance];
    switch (drawType)
    {
    case PROSPER_DRAW_TYPE_PRIMITIVE_ID: return uint_to_color(hit->primitive);
    case PROSPER_DRAW_TYPE_MESH_ID: return uint_to_color(inst->meshIndex);
    case PROSPER_DRAW_TYPE_MATERIAL_ID: return uint_to_color(inst->materialIndex);
    case PROSPER_DRAW_TYPE_POSITION: return sf->positionWS;
    case PROSPER_DRAW_TYPE_TEXCOORD0: return ora_v3_make(sf->uv.x, sf->uv.y, 0.0f);
    case PROSPER_DRAW_TYPE_ALBEDO: return sf->material.albedo;
    case PROSPER_DRAW_TYPE_SHADING_NORMAL:
        return ora_v3_make(fmaf(sf->normalWS.x, 0.5f, 0.5f), fmaf(sf->normalWS.y, 0.5f, 0.5f), fmaf(sf->normalWS.z, 0.5f, 0.5f));
    case PROSPER_DRAW_TYPE_ROUGHNESS: return ora_v3_make(sf->material.roughness, sf->material.roughness, sf->material.roughness);
    case PROSPER_DRAW_TYPE_METALLIC: return ora_v3_make(sf->material.metallic, sf->material.metallic, sf->material.metallic);
    default: return ora_v3_make(1.0f, 0.0f, 1.0f);
    }
}

/* main.rgen:195-223 */
static ora_v3 evaluate_direct_lighting(ora_path_ctx *c, const ora_surface *sf, ora_v3 throughput)
{
    if (sf->material.alpha == 0.0f) return ora_v3_make(0.0f, 0.0f, 0.0f);
    const ora_scene *s = c->scene;
    const uint32_t lightCount = 1u + s->view.pointLights->count + s->view.spotLights->count;
    uint32_t lightIndex = ora_f2uint(rnd01(&c->rng) * (float)lightCount);
    if (lightIndex > lightCount - 1u) lightIndex = lightCount - 1u;
    ora_v3 l;
    float d;
    ora_v3 irradiance;
    if (sample_light(s, sf->positionWS, lightIndex, &l, &d, &irradiance))
        c->counters->spotLightSamples++;
    else
        c->counters->lightSamples++;
    if (ora_dot(l, sf->normalWS) <= 0.0f) return ora_v3_make(0.0f, 0.0f, 0.0f);
    irradiance = ora_scale(irradiance, shadow(c, sf->positionWS, l, 0.1f, d));
    irradiance = ora_scale(irradiance, (float)lightCount);
    return ora_mul(ora_mul(throughput, irradiance), eval_brdf_times_nol(l, sf));
}

/* main.rgen:225-283: radiance of one path */
static ora_v3 trace_path(ora_path_ctx *c, uint32_t px, uint32_t py, uint32_t width, uint32_t height)
{
    const prosper_ReferencePC *pc = c->pc;
    c->rng.s[0] = px;
    c->rng.s[1] = py;
    c->rng.s[2] = pc->frameIndex;
    const ora_v2 j = rnd2d01(&c->rng);
    ora_v2 uv = {((float)px + j.x) / (float)width, ((float)py + j.y) / (float)height};

    ora_v3 color = ora_v3_make(0.0f, 0.0f, 0.0f);
    ora_v3 throughput = ora_v3_make(1.0f, 1.0f, 1.0f);
    uint32_t bounce = 0;
    ora_ray ray;
    if (pc->flags & PROSPER_PC_FLAG_DEPTH_OF_FIELD)
    {
        const ora_v2 lens = rnd2d01(&c->rng);
        ray = thin_lens_camera_ray(c->camera, uv, lens, pc->apertureDiameter, pc->focusDistance, pc->focalLength);
    }
    else
        ray = pinhole_camera_ray(c->camera, uv);
    c->counters->paths++;

    while (bounce < PROSPER_RT_MAX_BOUNCES)
    {
        if (bounce >= pc->maxBounces) break;
        const ora_hit hit = trace_closest(c, &ray);
        if (hit.drawInstance == ORA_MISS_INDEX)
        {
            if (pc->flags & PROSPER_PC_FLAG_IBL)
            {
                const ora_v3 sky = sample_skybox(c->scene, ray.d);
                c->counters->skyLookups++;
                add_bounce(pc, &color, ora_mul(throughput, sky), bounce);
            }
            break;
        }
        const ora_surface sf = evaluate_surface(c, &ray, &hit);
        if (pc->drawType != PROSPER_DRAW_TYPE_DEFAULT && pc->drawType != PROSPER_DRAW_TYPE_MESHLET_ID)
        {
            color = debug_color(c->scene, pc->drawType, &hit, &sf);
            break;
        }
        add_bounce(pc, &color, evaluate_direct_lighting(c, &sf, throughput), bounce);
        ora_v3 rd;
        importance_sample_bounce(c, &sf, &throughput, &rd);
        /* arithmetic contract: a path with throughput exactly (0,0,0) ends here - all its later terms are
         * throughput * X = +0 for finite X (NaN components compare unequal and keep the path going).
         * The GLSL has no such line: the literal mode (ora_scene_set_literal_glsl) leaves it out and goes on
         * exactly as main.rgen:269-283 is written, 0 * inf = NaN and NaN ray directions included. */
        if (!c->scene->literalGlsl && throughput.x == 0.0f && throughput.y == 0.0f && throughput.z == 0.0f) break;
        if (bounce > pc->rouletteStartBounce)
        {
            if (rnd01(&c->rng) < ora_max(0.05f, 1.0f - ora_max3(throughput))) break;
        }
        ray.o = offset_ray(sf.positionWS, sf.normalWS);
        ray.d = rd;
        ray.tMin = 0.0f;
        ray.tMax = INFINITY;
        bounce++;
    }
    return color;
}

static inline uint32_t tile_local_width(uint32_t width, const prosper_pt_tile_desc *tile)
{
    if (!tile || tile->stripeCount <= 1 || tile->stripeWidth == 0) return width;
    uint32_t n = 0;
    for (uint32_t x = 0; x < width; x += tile->stripeWidth)
    {
        const uint32_t s = x / tile->stripeWidth;
        if (s % tile->stripeCount == tile->stripeIndex)
            n += (x + tile->stripeWidth <= width) ? tile->stripeWidth : (width - x);
    }
    return n;
}

void ora_render(
    const ora_scene *scene, const prosper_ReferencePC *pc, const prosper_CameraUniforms *camera, uint32_t width,
    uint32_t height, const prosper_pt_tile_desc *tile, float *rgba, int threads, ora_counters *counters)
{
    const int tiled = tile && tile->stripeCount > 1 && tile->stripeWidth > 0;
    const uint32_t localWidth = tile_local_width(width, tile);
    ora_counters total;
    memset(&total, 0, sizeof(total));
#ifdef _OPENMP
    if (threads > 0) omp_set_num_threads(threads);
#else
    (void)threads;
#endif
#pragma omp parallel
    {
        ora_counters local;
        memset(&local, 0, sizeof(local));
        ora_path_ctx ctx;
        ctx.scene = scene;
        ctx.pc = pc;
        ctx.camera = camera;
        ctx.counters = &local;
        /* BASELINE.md's CPU plan: one worker per hardware thread pulling 16x16-pixel tiles from a shared counter
         * (schedule(dynamic, 1) over the tile index is exactly that counter) */
        const int64_t tilesX = ((int64_t)width + 15) / 16, tilesY = ((int64_t)height + 15) / 16;
#pragma omp for schedule(dynamic, 1)
        for (int64_t tileIndex = 0; tileIndex < tilesX * tilesY; ++tileIndex)
        {
            const uint32_t x0 = (uint32_t)(tileIndex % tilesX) * 16u, y0 = (uint32_t)(tileIndex / tilesX) * 16u;
            for (uint32_t ty = 0; ty < 16u; ++ty)
            for (uint32_t tx = 0; tx < 16u; ++tx)
            {
                const uint32_t x = x0 + tx;
                const int64_t y = (int64_t)y0 + ty;
                if (x >= width || y >= (int64_t)height) continue;
                uint32_t lx = x;
                if (tiled)
                {
                    const uint32_t s = x / tile->stripeWidth;
                    if (s % tile->stripeCount != tile->stripeIndex) continue;
                    lx = (s / tile->stripeCount) * tile->stripeWidth + (x % tile->stripeWidth);
                }
                const ora_v3 color = trace_path(&ctx, x, (uint32_t)y, width, height);
                float *out = rgba + 4u * ((size_t)y * localWidth + lx);
                /* main.rgen:285-298 */
                if ((pc->flags & PROSPER_PC_FLAG_SKIP_HISTORY) || !(pc->flags & PROSPER_PC_FLAG_ACCUMULATE))
                {
                    out[0] = color.x;
                    out[1] = color.y;
                    out[2] = color.z;
                    out[3] = 1.0f;
                }
                else
                {
                    const float hc = out[3] + 1.0f;
                    const float inv = 1.0f / hc;
                    out[0] = fmaf(color.x - out[0], inv, out[0]);
                    out[1] = fmaf(color.y - out[1], inv, out[1]);
                    out[2] = fmaf(color.z - out[2], inv, out[2]);
                    out[3] = hc;
                    local.historyReads++;
                }
                local.pixelsWritten++;
            }
        }
#pragma omp critical
        {
            total.paths += local.paths;
            total.closestRays += local.closestRays;
            total.shadowRays += local.shadowRays;
            total.closestHits += local.closestHits;
            total.lightSamples += local.lightSamples;
            total.spotLightSamples += local.spotLightSamples;
            total.skyLookups += local.skyLookups;
            total.pixelsWritten += local.pixelsWritten;
            total.historyReads += local.historyReads;
        }
    }
    if (counters) *counters = total;
}

/* ------------------------------------------------------------------------------------------
 * Host-side mirrors
 * ---------------------------------------------------------------------------------------- */

uint16_t ora_pack_half(float f) { return ora_float_to_half(f); }
float ora_unpack_half(uint16_t h) { return ora_half_to_float(h); }

/* glm::packSnorm3x10_1x2: round(clamp(v, -1, 1) * (511, 511, 511, 1)) into 10/10/10/2 bits */
uint32_t ora_pack_snorm3x10_1x2(const float v[4])
{
    const float scale[4] = {511.0f, 511.0f, 511.0f, 1.0f};
    int32_t q[4];
    for (int i = 0; i < 4; ++i)
    {
        const float c = ora_clamp(v[i], -1.0f, 1.0f) * scale[i];
        q[i] = (int32_t)rintf(c);
    }
    return ((uint32_t)q[0] & 0x3FFu) | (((uint32_t)q[1] & 0x3FFu) << 10) | (((uint32_t)q[2] & 0x3FFu) << 20) |
           (((uint32_t)q[3] & 0x3u) << 30);
}

/* src/scene/DeferredLoadingContext.cpp:442-490 */
void ora_pack_mesh(
    const float *positions, const float *normals, const float *tangents, const float *uvs, uint32_t vertexCount,
    uint64_t *outPositions, uint32_t *outNormals, uint32_t *outTangents, uint32_t *outUvs)
{
    for (uint32_t i = 0; i < vertexCount; ++i)
    {
        if (positions && outPositions)
        {
            const uint64_t x = ora_float_to_half(positions[3 * i + 0]);
            const uint64_t y = ora_float_to_half(positions[3 * i + 1]);
            const uint64_t z = ora_float_to_half(positions[3 * i + 2]);
            const uint64_t w = ora_float_to_half(1.0f);
            outPositions[i] = x | (y << 16) | (z << 32) | (w << 48);
        }
        if (normals && outNormals)
        {
            const float n[4] = {normals[3 * i + 0], normals[3 * i + 1], normals[3 * i + 2], 0.0f};
            outNormals[i] = ora_pack_snorm3x10_1x2(n);
        }
        if (tangents && outTangents) outTangents[i] = ora_pack_snorm3x10_1x2(&tangents[4 * i]);
        if (uvs && outUvs)
            outUvs[i] = (uint32_t)ora_float_to_half(uvs[2 * i + 0]) | ((uint32_t)ora_float_to_half(uvs[2 * i + 1]) << 16);
    }
}

/* 4x4 inverse by cofactors (what glm::inverse does), column-major */
static void mat4_inverse(const float m[16], float out[16])
{
    float inv[16];
    inv[0] = m[5] * m[10] * m[15] - m[5] * m[11] * m[14] - m[9] * m[6] * m[15] + m[9] * m[7] * m[14] + m[13] * m[6] * m[11] - m[13] * m[7] * m[10];
    inv[4] = -m[4] * m[10] * m[15] + m[4] * m[11] * m[14] + m[8] * m[6] * m[15] - m[8] * m[7] * m[14] - m[12] * m[6] * m[11] + m[12] * m[7] * m[10];
    inv[8] = m[4] * m[9] * m[15] - m[4] * m[11] * m[13] - m[8] * m[5] * m[15] + m[8] * m[7] * m[13] + m[12] * m[5] * m[11] - m[12] * m[7] * m[9];
    inv[12] = -m[4] * m[9] * m[14] + m[4] * m[10] * m[13] + m[8] * m[5] * m[14] - m[8] * m[6] * m[13] - m[12] * m[5] * m[10] + m[12] * m[6] * m[9];
    inv[1] = -m[1] * m[10] * m[15] + m[1] * m[11] * m[14] + m[9] * m[2] * m[15] - m[9] * m[3] * m[14] - m[13] * m[2] * m[11] + m[13] * m[3] * m[10];
    inv[5] = m[0] * m[10] * m[15] - m[0] * m[11] * m[14] - m[8] * m[2] * m[15] + m[8] * m[3] * m[14] + m[12] * m[2] * m[11] - m[12] * m[3] * m[10];
    inv[9] = -m[0] * m[9] * m[15] + m[0] * m[11] * m[13] + m[8] * m[1] * m[15] - m[8] * m[3] * m[13] - m[12] * m[1] * m[11] + m[12] * m[3] * m[9];
    inv[13] = m[0] * m[9] * m[14] - m[0] * m[10] * m[13] - m[8] * m[1] * m[14] + m[8] * m[2] * m[13] + m[12] * m[1] * m[10] - m[12] * m[2] * m[9];
    inv[2] = m[1] * m[6] * m[15] - m[1] * m[7] * m[14] - m[5] * m[2] * m[15] + m[5] * m[3] * m[14] + m[13] * m[2] * m[7] - m[13] * m[3] * m[6];
    inv[6] = -m[0] * m[6] * m[15] + m[0] * m[7] * m[14] + m[4] * m[2] * m[15] - m[4] * m[3] * m[14] - m[12] * m[2] * m[7] + m[12] * m[3] * m[6];
    inv[10] = m[0] * m[5] * m[15] - m[0] * m[7] * m[13] - m[4] * m[1] * m[15] + m[4] * m[3] * m[13] + m[12] * m[1] * m[7] - m[12] * m[3] * m[5];
    inv[14] = -m[0] * m[5] * m[14] + m[0] * m[6] * m[13] + m[4] * m[1] * m[14] - m[4] * m[2] * m[13] - m[12] * m[1] * m[6] + m[12] * m[2] * m[5];
    inv[3] = -m[1] * m[6] * m[11] + m[1] * m[7] * m[10] + m[5] * m[2] * m[11] - m[5] * m[3] * m[10] - m[9] * m[2] * m[7] + m[9] * m[3] * m[6];
    inv[7] = m[0] * m[6] * m[11] - m[0] * m[7] * m[10] - m[4] * m[2] * m[11] + m[4] * m[3] * m[10] + m[8] * m[2] * m[7] - m[8] * m[3] * m[6];
    inv[11] = -m[0] * m[5] * m[11] + m[0] * m[7] * m[9] + m[4] * m[1] * m[11] - m[4] * m[3] * m[9] - m[8] * m[1] * m[7] + m[8] * m[3] * m[5];
    inv[15] = m[0] * m[5] * m[10] - m[0] * m[6] * m[9] - m[4] * m[1] * m[10] + m[4] * m[2] * m[9] + m[8] * m[1] * m[6] - m[8] * m[2] * m[5];
    const float det = m[0] * inv[0] + m[1] * inv[4] + m[2] * inv[8] + m[3] * inv[12];
    const float idet = 1.0f / det;
    for (int i = 0; i < 16; ++i) out[i] = inv[i] * idet;
}

static void mat4_mul(const float a[16], const float b[16], float out[16])
{
    float r[16];
    for (int c = 0; c < 4; ++c)
        for (int row = 0; row < 4; ++row)
            r[c * 4 + row] = ((a[0 * 4 + row] * b[c * 4 + 0] + a[1 * 4 + row] * b[c * 4 + 1]) + a[2 * 4 + row] * b[c * 4 + 2]) +
                             a[3 * 4 + row] * b[c * 4 + 3];
    memcpy(out, r, sizeof(r));
}

/* src/scene/Camera.cpp:105-153 (perspective), :366-395 (updateWorldToCamera), :162-204 */
void ora_camera_uniforms(
    const float eye[3], const float target[3], const float up[3], float fovY, float zNear, float zFar, uint32_t width,
    uint32_t height, prosper_CameraUniforms *out, float *focalLength)
{
    memset(out, 0, sizeof(*out));
    const ora_v3 e = ora_v3_make(eye[0], eye[1], eye[2]);
    const ora_v3 fwd = ora_normalize(ora_sub(ora_v3_make(target[0], target[1], target[2]), e));
    const ora_v3 z = ora_neg(fwd);
    const ora_v3 right = ora_normalize(ora_cross(ora_v3_make(up[0], up[1], up[2]), z));
    const ora_v3 newUp = ora_normalize(ora_cross(z, right));
    float w2c[16] = {right.x, newUp.x, z.x, 0.0f, right.y, newUp.y, z.y, 0.0f, right.z, newUp.z, z.z, 0.0f,
                     -ora_dot(right, e), -ora_dot(newUp, e), -ora_dot(z, e), 1.0f};
    float c2w[16];
    mat4_inverse(w2c, c2w);

    const float ar = (float)width / (float)height;
    /* reverse-z: near and far swapped (Camera.cpp:112-115) */
    const float zN = zFar;
    const float zF = zNear;
    const float tf = 1.0f / tanf(fovY * 0.5f);
    const float flipY[16] = {1.0f, 0.0f, 0.0f, 0.0f, 0.0f, -1.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.5f, 0.0f, 0.0f, 0.0f, 0.5f, 1.0f};
    const float proj[16] = {tf / ar, 0.0f, 0.0f, 0.0f, 0.0f, tf, 0.0f, 0.0f, -0.0f, 0.0f, (zF + zN) / (zN - zF), -1.0f,
                            0.0f, 0.0f, 2 * zF * zN / (zN - zF), 0.0f};
    float c2c[16];
    mat4_mul(flipY, proj, c2c);
    float c2cw[16], clipToWorld[16];
    mat4_mul(c2c, w2c, c2cw);
    mat4_inverse(c2cw, clipToWorld);

    memcpy(&out->worldToCamera, w2c, 64);
    memcpy(&out->cameraToWorld, c2w, 64);
    memcpy(&out->cameraToClip, c2c, 64);
    memcpy(&out->clipToWorld, clipToWorld, 64);
    memcpy(&out->previousWorldToCamera, w2c, 64);
    memcpy(&out->previousCameraToClip, c2c, 64);
    out->eye.x = e.x;
    out->eye.y = e.y;
    out->eye.z = e.z;
    out->eye.w = 1.0f;
    out->resolution[0] = width;
    out->resolution[1] = height;
    out->near_ = zNear;
    out->far_ = zFar;
    out->maxViewScale = 1.0f;
    /* Camera.cpp:150-152 with sensorWidth() = 0.035 (Camera.hpp) */
    const float sensorHeight = 0.035f / ar;
    if (focalLength) *focalLength = sensorHeight * tf * 0.5f;
}

/* ------------------------------------------------------------------------------------------
 * Known-answer test dispatcher
 * ---------------------------------------------------------------------------------------- */

/* ------------------------------------------------------------------------------------------
 * ReSTIR-DI trace (res/shader/rt/direct_illumination/main.rgen:44-165): reuses sample_light, the shadow
 * ray with its any-hit, and evalBRDFTimesNoL.
 * ---------------------------------------------------------------------------------------- */

/* scene/material.glsl:20-32 */
static ora_v3 signed_oct_decode(ora_v3 n)
{
    ora_v3 o;
    o.x = n.x - n.y;
    o.y = (n.x + n.y) - 1.0f;
    o.z = n.z * 2.0f - 1.0f;
    o.z = o.z * ((1.0f - ora_abs(o.x)) - ora_abs(o.y));
    return ora_normalize(o);
}

/* scene/camera.glsl:27-33: clipToWorld * vec4(uv * 2 - 1, depth, 1), then xyz / w */
static ora_v3 world_pos(const prosper_CameraUniforms *c, ora_v2 uv, float depth)
{
    const prosper_mat4 *m = &c->clipToWorld;
    const float x = uv.x * 2.0f - 1.0f, y = uv.y * 2.0f - 1.0f;
    const float vx = fmaf(m->col[2].x, depth, fmaf(m->col[1].x, y, fmaf(m->col[0].x, x, m->col[3].x)));
    const float vy = fmaf(m->col[2].y, depth, fmaf(m->col[1].y, y, fmaf(m->col[0].y, x, m->col[3].y)));
    const float vz = fmaf(m->col[2].z, depth, fmaf(m->col[1].z, y, fmaf(m->col[0].z, x, m->col[3].z)));
    const float vw = fmaf(m->col[2].w, depth, fmaf(m->col[1].w, y, fmaf(m->col[0].w, x, m->col[3].w)));
    return ora_scale(ora_v3_make(vx, vy, vz), 1.0f / vw);
}

void ora_restir_di_trace(
    const ora_scene *s, const ora_restir_pc *pc, const prosper_CameraUniforms *camera, uint32_t width, uint32_t height,
    const float *albedoRoughness, const float *normalMetallic, const float *nonLinearDepth, const float *reservoirs,
    float *hdr, int threads)
{
    const int skipHistory = (pc->flags & 1u) != 0, accumulate = (pc->flags & 2u) != 0;
    if (threads > 0) omp_set_num_threads(threads);
#pragma omp parallel for schedule(dynamic, 8)
    for (int64_t py = 0; py < (int64_t)height; ++py)
        for (uint32_t px = 0; px < width; ++px)
        {
            const size_t i = (size_t)py * width + px;
            /* main.rgen:113-129 */
            ora_v2 uv;
            uv.x = (float)px / (float)width;
            uv.y = (float)py / (float)height;
            const float depth = nonLinearDepth[i];
            ora_surface sf;
            sf.positionWS = world_pos(camera, uv, depth);
            sf.invViewRayWS =
                ora_normalize(ora_sub(ora_v3_make(camera->eye.x, camera->eye.y, camera->eye.z), sf.positionWS));
            /* loadFromGbuffer, scene/material.glsl:34-49 */
            sf.material.albedo = ora_v3_make(albedoRoughness[4 * i], albedoRoughness[4 * i + 1], albedoRoughness[4 * i + 2]);
            sf.material.roughness = albedoRoughness[4 * i + 3];
            sf.material.normal =
                signed_oct_decode(ora_v3_make(normalMetallic[4 * i], normalMetallic[4 * i + 1], normalMetallic[4 * i + 3]));
            sf.material.metallic = normalMetallic[4 * i + 2];
            sf.material.alpha = -1.0f;
            sf.normalWS = sf.material.normal;
            sf.uv.x = 0.0f;
            sf.uv.y = 0.0f;
            sf.NoV = ora_saturate(ora_dot(sf.normalWS, sf.invViewRayWS));
            ora_v3 color;
            if (pc->drawType != PROSPER_DRAW_TYPE_DEFAULT)
            {
                /* main.rgen:131-146: Position through commonDebugDraw, everything else shows the albedo channel */
                color = pc->drawType == PROSPER_DRAW_TYPE_POSITION ? sf.positionWS : sf.material.albedo;
                hdr[4 * i] = color.x;
                hdr[4 * i + 1] = color.y;
                hdr[4 * i + 2] = color.z;
                hdr[4 * i + 3] = 1.0f;
                continue;
            }
            /* evaluateDirectLightingReSTIR, main.rgen:88-109 */
            const int32_t lightIndex = (int32_t)ora_f2u(reservoirs[2 * i]);
            const float weight = reservoirs[2 * i + 1];
            color = ora_v3_make(0.0f, 0.0f, 0.0f);
            if (!(sf.material.alpha == 0.0f || lightIndex < 0))
            {
                ora_v3 l, irradiance;
                float d;
                sample_light(s, sf.positionWS, (uint32_t)lightIndex, &l, &d, &irradiance);
                if (ora_dot(l, sf.normalWS) > 0.0f)
                {
                    /* shadow(): seed pcg(px ^ py) - the state (px, py, frameIndex) is never advanced here */
                    ora_hit h;
                    const float visible = trace(s, sf.positionWS, l, 0.1f, d, ora_pcg(px ^ (uint32_t)py), 1, &h) ? 0.0f : 1.0f;
                    irradiance = ora_scale(irradiance, visible);
                    color = ora_scale(ora_mul(irradiance, eval_brdf_times_nol(l, &sf)), weight);
                }
            }
            if (skipHistory || !accumulate)
            {
                hdr[4 * i] = color.x;
                hdr[4 * i + 1] = color.y;
                hdr[4 * i + 2] = color.z;
                hdr[4 * i + 3] = 1.0f;
            }
            else
            {
                /* the same running mean as the reference pass (ora_render): hist + (c - hist) * (1 / count), fused */
                const float count = hdr[4 * i + 3] + 1.0f;
                const float inv = 1.0f / count;
                hdr[4 * i] = fmaf(color.x - hdr[4 * i], inv, hdr[4 * i]);
                hdr[4 * i + 1] = fmaf(color.y - hdr[4 * i + 1], inv, hdr[4 * i + 1]);
                hdr[4 * i + 2] = fmaf(color.z - hdr[4 * i + 2], inv, hdr[4 * i + 2]);
                hdr[4 * i + 3] = count;
            }
        }
}

/* ------------------------------------------------------------------------------------------
 * Tone map (res/shader/tone_map.comp, common/math.glsl:17-85).  Arithmetic contract additions:
 * mod(x, y) = x - y * floor(x / y) (GLSL), pow through ora_pow, constants as the GLSL front end folds
 * them (47/48, 0.5/48, 1/2.2), R9G9B9E5 decode mantissa * 2^(e - 24) (exact), trilinear weights in fp32
 * summed as an fma chain in the order (x0y0z0, x1y0z0, x0y1z0, x1y1z0, x0y0z1, ...), UNORM8 store =
 * round-half-even(saturate(x) * 255) with NaN -> 0.
 * ---------------------------------------------------------------------------------------- */
static inline float ora_mod(float x, float y) { return x - y * floorf(x / y); }

static ora_v3 rgb_to_hsv(ora_v3 rgb) /* math.glsl:17-44 */
{
    const float value = ora_max(ora_max(rgb.x, rgb.y), rgb.z);
    const float valueMinusChroma = ora_min(ora_min(rgb.x, rgb.y), rgb.z);
    const float chroma = value - valueMinusChroma;
    float hue;
    if (chroma == 0.0f)
        hue = 0.0f;
    else if (value == rgb.x)
        hue = ora_mod((rgb.y - rgb.z) / chroma, 6.0f);
    else if (value == rgb.y)
        hue = (rgb.z - rgb.x) / chroma + 2.0f;
    else
        hue = (rgb.x - rgb.y) / chroma + 4.0f;
    const float saturation = value == 0.0f ? 0.0f : chroma / value;
    return ora_v3_make(hue, saturation, value);
}

static ora_v3 hsv_to_rgb(ora_v3 hsv) /* math.glsl:47-83 */
{
    const float hue = hsv.x, saturation = hsv.y, value = hsv.z;
    const float chroma = value * saturation;
    const float x = chroma * (1.0f - ora_abs(ora_mod(hue, 2.0f) - 1.0f));
    ora_v3 rgb;
    if (hue < 1.0f)
        rgb = ora_v3_make(chroma, x, 0.0f);
    else if (hue < 2.0f)
        rgb = ora_v3_make(x, chroma, 0.0f);
    else if (hue < 3.0f)
        rgb = ora_v3_make(0.0f, chroma, x);
    else if (hue < 4.0f)
        rgb = ora_v3_make(0.0f, x, chroma);
    else if (hue < 5.0f)
        rgb = ora_v3_make(x, 0.0f, chroma);
    else
        rgb = ora_v3_make(chroma, 0.0f, x);
    const float m = value - chroma;
    return ora_v3_make(rgb.x + m, rgb.y + m, rgb.z + m);
}

static inline ora_v3 decode_r9g9b9e5(uint32_t p)
{
    const float scale = ora_u2f(((p >> 27) + 103u) << 23); /* 2^(e - 15 - 9) */
    return ora_v3_make((float)(p & 0x1FFu) * scale, (float)((p >> 9) & 0x1FFu) * scale, (float)((p >> 18) & 0x1FFu) * scale);
}

static inline int32_t clamp_texel(int32_t i, int32_t n) { return i < 0 ? 0 : (i >= n ? n - 1 : i); }

/* textureLod(sampler3D(lut, linear/clamp), uv, 0).xyz */
static ora_v3 sample_lut(const uint32_t *lut, int32_t n, ora_v3 uv)
{
    const float u = fmaf(uv.x, (float)n, -0.5f), v = fmaf(uv.y, (float)n, -0.5f), w = fmaf(uv.z, (float)n, -0.5f);
    const float fu = floorf(u), fv = floorf(v), fw = floorf(w);
    const float a = u - fu, b = v - fv, c = w - fw;
    const int32_t i0 = clamp_texel(ora_f2i(fu), n), i1 = clamp_texel(ora_f2i(fu) + 1, n);
    const int32_t j0 = clamp_texel(ora_f2i(fv), n), j1 = clamp_texel(ora_f2i(fv) + 1, n);
    const int32_t k0 = clamp_texel(ora_f2i(fw), n), k1 = clamp_texel(ora_f2i(fw) + 1, n);
    const int32_t is[2] = {i0, i1}, js[2] = {j0, j1}, ks[2] = {k0, k1};
    const float wx[2] = {1.0f - a, a}, wy[2] = {1.0f - b, b}, wz[2] = {1.0f - c, c};
    ora_v3 acc = ora_v3_make(0.0f, 0.0f, 0.0f);
    for (int z = 0; z < 2; ++z)
        for (int y = 0; y < 2; ++y)
            for (int x = 0; x < 2; ++x)
            {
                const ora_v3 t = decode_r9g9b9e5(lut[((size_t)ks[z] * n + (size_t)js[y]) * n + (size_t)is[x]]);
                const float wgt = (wx[x] * wy[y]) * wz[z];
                if (x == 0 && y == 0 && z == 0)
                    acc = ora_v3_make(wgt * t.x, wgt * t.y, wgt * t.z);
                else
                    acc = ora_v3_make(fmaf(wgt, t.x, acc.x), fmaf(wgt, t.y, acc.y), fmaf(wgt, t.z, acc.z));
            }
    return acc;
}

static inline uint8_t to_unorm8(float x)
{
    if (x != x) return 0;
    return (uint8_t)rintf(ora_clamp(x, 0.0f, 1.0f) * 255.0f);
}

void ora_tone_map(
    const float *hdr, const uint32_t *lut, uint32_t dim, float exposure, float contrast, uint8_t *out, uint64_t count)
{
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < (int64_t)count; ++i)
    {
        /* the RGBA16F image the tone map reads (blit of the RGBA32F accumulation, round-to-nearest-even) */
        ora_v3 color = ora_v3_make(
            ora_half_to_float(ora_float_to_half(hdr[4 * i + 0])), ora_half_to_float(ora_float_to_half(hdr[4 * i + 1])),
            ora_half_to_float(ora_float_to_half(hdr[4 * i + 2])));
        color = ora_v3_make(color.x * exposure, color.y * exposure, color.z * exposure);
        ora_v3 hsv = rgb_to_hsv(color);
        hsv.z = ora_pow(hsv.z, contrast);
        color = hsv_to_rgb(hsv);
        /* tonyMcMapface, tone_map.comp:17-29 */
        const ora_v3 enc = ora_v3_make(color.x / (color.x + 1.0f), color.y / (color.y + 1.0f), color.z / (color.z + 1.0f));
        const float k1 = 47.0f / 48.0f, k0 = 0.5f / 48.0f;
        const float kd1 = ((float)dim - 1.0f) / (float)dim, kd0 = 0.5f / (float)dim;
        const float s1 = dim == 48u ? k1 : kd1, s0 = dim == 48u ? k0 : kd0;
        const ora_v3 uv = ora_v3_make(enc.x * s1 + s0, enc.y * s1 + s0, enc.z * s1 + s0);
        color = sample_lut(lut, (int32_t)dim, uv);
        const float invGamma = 1.0f / 2.2f;
        out[4 * i + 0] = to_unorm8(ora_pow(color.x, invGamma));
        out[4 * i + 1] = to_unorm8(ora_pow(color.y, invGamma));
        out[4 * i + 2] = to_unorm8(ora_pow(color.z, invGamma));
        out[4 * i + 3] = 255;
    }
}

int ora_eval_fn(uint32_t fn, const float *in, uint32_t in_stride, float *out, uint32_t out_stride, uint32_t n)
{
    for (uint32_t i = 0; i < n; ++i)
    {
        const float *a = in + (size_t)i * in_stride;
        float *o = out + (size_t)i * out_stride;
        switch (fn)
        {
        case ORA_FN_SINCOS: ora_sincos(a[0], &o[0], &o[1]); break;
        case ORA_FN_POW: o[0] = ora_pow(a[0], a[1]); break;
        case ORA_FN_SRGB_TO_LINEAR: o[0] = srgb_to_linear(a[0]); break;
        case ORA_FN_NORMALIZE:
        {
            const ora_v3 r = ora_normalize(ora_v3_make(a[0], a[1], a[2]));
            o[0] = r.x; o[1] = r.y; o[2] = r.z;
            break;
        }
        case ORA_FN_UNPACK_SNORM:
        {
            const uint32_t bits = ora_f2u(a[0]);
            const ora_v3 r = unpack_snorm_r10g10b10(bits);
            o[0] = r.x; o[1] = r.y; o[2] = r.z;
            o[3] = (float)((int32_t)bits >> 30);
            break;
        }
        case ORA_FN_ONB:
        {
            const ora_onb b = orthonormal_basis(ora_v3_make(a[0], a[1], a[2]));
            o[0] = b.b1.x; o[1] = b.b1.y; o[2] = b.b1.z;
            o[3] = b.b2.x; o[4] = b.b2.y; o[5] = b.b2.z;
            o[6] = b.n.x; o[7] = b.n.y; o[8] = b.n.z;
            break;
        }
        case ORA_FN_COSINE_SAMPLE:
        {
            ora_v2 u = {a[3], a[4]};
            const ora_v3 r = cosine_sample_hemisphere(ora_v3_make(a[0], a[1], a[2]), u);
            o[0] = r.x; o[1] = r.y; o[2] = r.z;
            break;
        }
        case ORA_FN_VNDF_SAMPLE:
        {
            ora_v2 u = {a[4], a[5]};
            const ora_v3 r = sample_visible_trowbridge_reitz(ora_v3_make(a[0], a[1], a[2]), a[3], u);
            o[0] = r.x; o[1] = r.y; o[2] = r.z;
            break;
        }
        case ORA_FN_VNDF_PDF:
            o[0] = visible_trowbridge_reitz_pdf(ora_v3_make(a[0], a[1], a[2]), ora_v3_make(a[3], a[4], a[5]), a[6]);
            break;
        case ORA_FN_EVAL_BRDF:
        {
            ora_surface sf;
            memset(&sf, 0, sizeof(sf));
            sf.normalWS = ora_v3_make(a[3], a[4], a[5]);
            sf.invViewRayWS = ora_v3_make(a[6], a[7], a[8]);
            sf.material.albedo = ora_v3_make(a[9], a[10], a[11]);
            sf.material.roughness = a[12];
            sf.material.metallic = a[13];
            sf.NoV = ora_saturate(ora_dot(sf.normalWS, sf.invViewRayWS));
            const ora_v3 r = eval_brdf_times_nol(ora_v3_make(a[0], a[1], a[2]), &sf);
            o[0] = r.x; o[1] = r.y; o[2] = r.z;
            break;
        }
        case ORA_FN_OFFSET_RAY:
        {
            const ora_v3 r = offset_ray(ora_v3_make(a[0], a[1], a[2]), ora_v3_make(a[3], a[4], a[5]));
            o[0] = r.x; o[1] = r.y; o[2] = r.z;
            break;
        }
        case ORA_FN_POINT_LIGHT:
        {
            prosper_PointLight L;
            L.position.x = a[0]; L.position.y = a[1]; L.position.z = a[2]; L.position.w = 0.0f;
            L.radianceAndRadius.x = a[3]; L.radianceAndRadius.y = a[4]; L.radianceAndRadius.z = a[5];
            L.radianceAndRadius.w = a[6];
            ora_v3 l, irr;
            float d;
            eval_point_light(&L, ora_v3_make(a[7], a[8], a[9]), &l, &d, &irr);
            o[0] = l.x; o[1] = l.y; o[2] = l.z; o[3] = d; o[4] = irr.x; o[5] = irr.y; o[6] = irr.z;
            break;
        }
        case ORA_FN_SPOT_LIGHT:
        {
            prosper_SpotLight L;
            L.positionAndAngleOffset.x = a[0]; L.positionAndAngleOffset.y = a[1]; L.positionAndAngleOffset.z = a[2];
            L.positionAndAngleOffset.w = a[3];
            L.radianceAndAngleScale.x = a[4]; L.radianceAndAngleScale.y = a[5]; L.radianceAndAngleScale.z = a[6];
            L.radianceAndAngleScale.w = a[7];
            L.direction.x = a[8]; L.direction.y = a[9]; L.direction.z = a[10]; L.direction.w = 0.0f;
            ora_v3 l, irr;
            float d;
            eval_spot_light(&L, ora_v3_make(a[11], a[12], a[13]), &l, &d, &irr);
            o[0] = l.x; o[1] = l.y; o[2] = l.z; o[3] = d; o[4] = irr.x; o[5] = irr.y; o[6] = irr.z;
            break;
        }
        case ORA_FN_TRIANGLE:
        {
            float t = 0.0f, bu = 0.0f, bv = 0.0f;
            const int hit = intersect_triangle(
                ora_v3_make(a[0], a[1], a[2]), ora_v3_make(a[3], a[4], a[5]), ora_v3_make(a[6], a[7], a[8]),
                ora_v3_make(a[9], a[10], a[11]), ora_v3_make(a[12], a[13], a[14]), a[15], a[16], &t, &bu, &bv);
            o[0] = (float)hit; o[1] = hit ? t : 0.0f; o[2] = hit ? bu : 0.0f; o[3] = hit ? bv : 0.0f;
            break;
        }
        case ORA_FN_HALF:
        {
            const uint16_t h = ora_float_to_half(a[0]);
            o[0] = ora_half_to_float(h);
            o[1] = ora_u2f((uint32_t)h);
            break;
        }
        case ORA_FN_RNG:
        {
            ora_rng r;
            r.s[0] = ora_f2u(a[0]); r.s[1] = ora_f2u(a[1]); r.s[2] = ora_f2u(a[2]);
            o[0] = rnd01(&r);
            const ora_v2 u = rnd2d01(&r);
            o[1] = u.x; o[2] = u.y;
            o[3] = ora_u2f(ora_pcg(r.s[0] ^ r.s[2]));
            break;
        }
        default: return -1;
        }
    }
    return 0;
}
