/*
 * oracle/oracle.h — TEST INFRASTRUCTURE (parity oracle + CPU baseline), never linked into the
 * product.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.
 *
 * A scalar-fp32 CPU restatement, in plain C, of prosper's path-tracing reference pass:
 * res/shader/rt/reference/main.rgen and everything it includes (rt/{ray,payload}.glsl,
 * rt/scene.{rahit,rchit,rmiss}, brdf.glsl, debug.glsl, common/{math,random,sampling}.glsl,
 * scene/{camera,geometry,instances,lighting,lights,material,materials,skybox,vertex,
 * visible_surface}.glsl) plus the host-side conversions that feed it (src/scene/Camera.cpp,
 * src/scene/DeferredLoadingContext.cpp:442-490, src/scene/WorldData.cpp:1455-1543,
 * src/render/RtReference.cpp:77-88,161-383).  Each function cites the lines it follows.
 *
 * PARITY STATUS: **parity unpinned** against the Vulkan original.  The reference has no tests,
 * golden vectors or fixtures, it cannot be built or run here (Vulkan RT + shaderc +
 * 14 empty submodules), and ray/triangle hit selection lives in the Vulkan driver.
 * What pins this oracle instead: the hand-derived known answers of SURVEY Appendix A and an
 * independent NumPy evaluation of every pure function (tests/golden/, tests/test_oracle_kat.py).
 *
 * The acceleration structure is the oracle's own (median-split BVH2, or brute force over all
 * triangles) and is deliberately NOT the product's: hit selection is defined by arithmetic
 * (ora_intersect_triangle + the closest/tie rule below), not by traversal order, so any correct
 * traversal must return the same hit.
 */
#ifndef ORACLE_H
#define ORACLE_H

#include <stddef.h>
#include <stdint.h>

#include "../include/prosper_pt/prosper_pt.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ora_scene ora_scene;

typedef struct ora_counters
{
    uint64_t paths;
    uint64_t closestRays;
    uint64_t shadowRays;
    uint64_t closestHits;
    uint64_t lightSamples;
    uint64_t spotLightSamples;
    uint64_t skyLookups;
    uint64_t pixelsWritten;
    uint64_t historyReads;
} ora_counters;

/* Borrows every pointer in `view` until ora_scene_destroy.  brute_force != 0 skips the BVH and
 * tests every triangle for every ray (small scenes only). */
ora_scene *ora_scene_create(const prosper_pt_scene_view *view, int brute_force);
void ora_scene_destroy(ora_scene *scene);
uint64_t ora_scene_triangle_count(const ora_scene *scene);
/* The ONE place ora_render departs from the text of main.rgen:241-283 is the arithmetic contract's rule "a path whose
 * throughput is exactly (0,0,0) after importanceSampleBounce ends there" (DESIGN.md section 3).  on != 0 removes
 * the rule: the loop then runs exactly as written - zero-throughput paths go on bouncing, 0 * inf and 0 * NaN
 * terms are added, rays with NaN directions are traced (they miss everything here; the Vulkan specification
 * leaves them undefined).  tests/test_literal_glsl.py measures, pixel by pixel, where the two modes differ. */
void ora_scene_set_literal_glsl(ora_scene *scene, int on);

/* One accumulated frame of the whole image (or of the tile's pixels, same compact layout as
 * prosper_pt_render).  `rgba` holds the history on entry and the new image on return
 * (main.rgen:285-298).  threads <= 0 uses every hardware thread. */
void ora_render(
    const ora_scene *scene, const prosper_ReferencePC *pc, const prosper_CameraUniforms *camera,
    uint32_t width, uint32_t height, const prosper_pt_tile_desc *tile, float *rgba, int threads,
    ora_counters *counters);

/* Single primary-ray probe used by traversal-semantics tests: returns 1 on hit. */
int ora_trace_closest(
    const ora_scene *scene, const float origin[3], const float dir[3], float tMin, float tMax,
    uint32_t randomSeed, uint32_t *drawInstanceIndex, uint32_t *primitiveID, float bary[2]);
int ora_trace_shadow(
    const ora_scene *scene, const float origin[3], const float dir[3], float tMin, float tMax,
    uint32_t randomSeed);

/* ---- pure functions exported for known-answer tests (same ids as PROSPER_PT_FN_*) ---- */
uint32_t ora_pcg(uint32_t v);
void ora_pcg3d(uint32_t v[3]);
uint16_t ora_pack_half(float f);
float ora_unpack_half(uint16_t h);
uint32_t ora_pack_snorm3x10_1x2(const float v[4]);
/* Evaluates function `fn` over n records; see oracle.c:ora_eval_fn for the record layouts. */
int ora_eval_fn(uint32_t fn, const float *in, uint32_t in_stride, float *out, uint32_t out_stride, uint32_t n);

/* A second client of the traversal: the ReSTIR-DI trace pass, res/shader/rt/direct_illumination/
 * main.rgen:44-165 (src/render/rtdi/Trace.cpp:297).  Per pixel: surface from the G-buffer (albedoRoughness,
 * normalMetallic as float4 texels, non-linear depth), the pixel's light reservoir (lightIndex bits, weight),
 * one shadow ray towards that light, BRDF, running-mean accumulation into `hdr` (rgba32f, in/out). */
typedef struct ora_restir_pc
{
    uint32_t drawType;
    uint32_t frameIndex;
    uint32_t flags; /* bit 0 skipHistory, bit 1 accumulate */
} ora_restir_pc;
void ora_restir_di_trace(
    const ora_scene *scene, const ora_restir_pc *pc, const prosper_CameraUniforms *camera, uint32_t width, uint32_t height,
    const float *albedoRoughness, const float *normalMetallic, const float *nonLinearDepth, const float *reservoirs,
    float *hdr, int threads);

/* The step after the path: RGBA32F -> RGBA16F blit (RtReference.cpp:339-377) followed by
 * res/shader/tone_map.comp:17-60 (exposure, HSV contrast, Tony McMapface 3-D LUT lookup, 1/2.2 gamma) into
 * RGBA8 UNORM (ToneMap.cpp:62-128).  `lut` = dim^3 R9G9B9E5 texels (x fastest), sampled like
 * textureLod(sampler3D, uv, 0) with a linear / clamp-to-edge sampler. */
void ora_tone_map(
    const float *hdr_rgba32f, const uint32_t *lut_r9g9b9e5, uint32_t dim, float exposure, float contrast,
    uint8_t *out_rgba8, uint64_t pixel_count);

/* ---- host-side mirrors ---- */
/* Camera::updateWorldToCamera + Camera::perspective + updateBuffer (src/scene/Camera.cpp:105-204,
 * 366-395) for a non-jittered camera; also returns CameraParameters::focalLength. */
void ora_camera_uniforms(
    const float eye[3], const float target[3], const float up[3], float fovY, float zN, float zF,
    uint32_t width, uint32_t height, prosper_CameraUniforms *out, float *focalLength);
/* packMeshData (src/scene/DeferredLoadingContext.cpp:442-490) for one mesh. */
void ora_pack_mesh(
    const float *positions, const float *normals, const float *tangents, const float *uvs,
    uint32_t vertexCount, uint64_t *outPositions, uint32_t *outNormals, uint32_t *outTangents,
    uint32_t *outUvs);

enum
{
    ORA_FN_SINCOS = 0,       /* in: x                     out: sin, cos */
    ORA_FN_POW = 1,          /* in: x, y                  out: pow */
    ORA_FN_SRGB_TO_LINEAR = 2, /* in: x                   out: y */
    ORA_FN_NORMALIZE = 3,    /* in: v3                    out: v3 */
    ORA_FN_UNPACK_SNORM = 4, /* in: bits(u32 as float)    out: v3 normal, w sign */
    ORA_FN_ONB = 5,          /* in: n3                    out: 9 (rows b1,b2,n) */
    ORA_FN_COSINE_SAMPLE = 6, /* in: n3, u2               out: v3 */
    ORA_FN_VNDF_SAMPLE = 7,  /* in: Ve3, alpha, u2        out: v3 */
    ORA_FN_VNDF_PDF = 8,     /* in: Ve3, Le3, alpha       out: pdf */
    ORA_FN_EVAL_BRDF = 9,    /* in: l3,n3,v3,albedo3,rough,metal  out: v3 */
    ORA_FN_OFFSET_RAY = 10,  /* in: p3, n3                out: v3 */
    ORA_FN_POINT_LIGHT = 11, /* in: pos3,radiance3,radius,surf3   out: l3,d,irr3 */
    ORA_FN_SPOT_LIGHT = 12,  /* in: pos3,off,rad3,scale,dir3,surf3 out: l3,d,irr3 */
    ORA_FN_TRIANGLE = 13,    /* in: o3,d3,v0,v1,v2,tmin,tmax out: hit,t,bu,bv */
    ORA_FN_HALF = 14,        /* in: f                     out: unpack(pack(f)) , bits */
    ORA_FN_RNG = 15,         /* in: px,py,frame (as u32 bits) out: 4 draws: rnd01, rnd2d01 x2.. */
    ORA_FN_COUNT = 16,
};

#ifdef __cplusplus
}
#endif

#endif /* ORACLE_H */
