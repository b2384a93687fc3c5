/*
 * prosper_pt/shader_structs.h — host/device POD layouts of the path-tracing reference pass.
 *
 * Byte-identical restatements (plain C, no glm) of the structs prosper shares between its C++
 * host code and its GLSL (reference: res/shader/shared/shader_structs/).  Every struct is a
 * tightly packed little-endian POD; sizes are pinned by static asserts below and by
 * tests/test_structs.py.
 *
 *   ReferencePC                 res/shader/shared/shader_structs/push_constants/rt_reference.h:6-16
 *   CameraUniforms              res/shader/shared/shader_structs/scene/camera.h:11-34
 *   DrawInstance                res/shader/shared/shader_structs/scene/draw_instance.h:11-16
 *   GeometryMetadata            res/shader/shared/shader_structs/scene/geometry_metadata.h:11-29
 *   MaterialData / AlphaMode    res/shader/shared/shader_structs/scene/material_data.h:19-68
 *   ModelInstanceTransforms     res/shader/shared/shader_structs/scene/model_instance_transforms.h:11-15
 *   Directional/Point/SpotLight res/shader/shared/shader_structs/scene/lights.h:16-33
 *   PointLights/SpotLights SSBO res/shader/scene/lights.glsl:11-23, src/scene/Light.hpp:27-59
 *   DrawType                    src/scene/DrawType.hpp:8-10
 */
#ifndef PROSPER_PT_SHADER_STRUCTS_H
#define PROSPER_PT_SHADER_STRUCTS_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#define PROSPER_PT_STATIC_ASSERT(c, m) static_assert(c, m)
#else
#define PROSPER_PT_STATIC_ASSERT(c, m) _Static_assert(c, m)
#endif

/* glm::vec4 / mat4 (column-major, 4 columns of vec4) / mat3x4 (3 columns of vec4). */
typedef struct prosper_vec4 { float x, y, z, w; } prosper_vec4;
typedef struct prosper_mat4 { prosper_vec4 col[4]; } prosper_mat4;
typedef struct prosper_mat3x4 { prosper_vec4 col[3]; } prosper_mat3x4;

/* rt_reference.h:6-16.  Defaults are the struct's own; RtReference overrides them per frame. */
typedef struct prosper_ReferencePC
{
    uint32_t drawType;            /* prosper_DrawType */
    uint32_t flags;               /* PROSPER_PC_FLAG_* */
    uint32_t frameIndex;
    float apertureDiameter;       /* 0.0001f */
    float focusDistance;          /* 1.f */
    float focalLength;            /* 0.f */
    uint32_t rouletteStartBounce; /* 3 */
    uint32_t maxBounces;          /* 3 */
} prosper_ReferencePC;
PROSPER_PT_STATIC_ASSERT(sizeof(prosper_ReferencePC) == 32, "ReferencePC is 32 B");

/* Bit positions from src/render/RtReference.cpp:77-88 / rt/reference/main.rgen:26-30. */
enum
{
    PROSPER_PC_FLAG_SKIP_HISTORY = 1u << 0,
    PROSPER_PC_FLAG_ACCUMULATE = 1u << 1,
    PROSPER_PC_FLAG_IBL = 1u << 2,
    PROSPER_PC_FLAG_DEPTH_OF_FIELD = 1u << 3,
    PROSPER_PC_FLAG_CLAMP_INDIRECT = 1u << 4,
};

/* src/scene/DrawType.hpp:8-10 (order is the enum value). */
typedef enum prosper_DrawType
{
    PROSPER_DRAW_TYPE_DEFAULT = 0,
    PROSPER_DRAW_TYPE_PRIMITIVE_ID = 1,
    PROSPER_DRAW_TYPE_MESHLET_ID = 2, /* treated as Default by the RT pass: main.rgen:259-260 */
    PROSPER_DRAW_TYPE_MESH_ID = 3,
    PROSPER_DRAW_TYPE_MATERIAL_ID = 4,
    PROSPER_DRAW_TYPE_POSITION = 5,
    PROSPER_DRAW_TYPE_SHADING_NORMAL = 6,
    PROSPER_DRAW_TYPE_TEXCOORD0 = 7,
    PROSPER_DRAW_TYPE_ALBEDO = 8,
    PROSPER_DRAW_TYPE_ROUGHNESS = 9,
    PROSPER_DRAW_TYPE_METALLIC = 10,
    PROSPER_DRAW_TYPE_COUNT = 11,
} prosper_DrawType;

/* camera.h:11-34 */
typedef struct prosper_CameraUniforms
{
    prosper_mat4 worldToCamera;
    prosper_mat4 cameraToWorld;
    prosper_mat4 cameraToClip;
    prosper_mat4 clipToWorld;
    prosper_mat4 previousWorldToCamera;
    prosper_mat4 previousCameraToClip;
    prosper_vec4 eye;
    prosper_vec4 nearPlane;
    prosper_vec4 farPlane;
    prosper_vec4 leftPlane;
    prosper_vec4 rightPlane;
    prosper_vec4 topPlane;
    prosper_vec4 bottomPlane;
    uint32_t resolution[2];
    float currentJitter[2];
    float previousJitter[2];
    float near_;
    float far_;
    float maxViewScale;
} prosper_CameraUniforms;
PROSPER_PT_STATIC_ASSERT(sizeof(prosper_CameraUniforms) == 532, "CameraUniforms is 532 B");

/* draw_instance.h:11-16 */
typedef struct prosper_DrawInstance
{
    uint32_t modelInstanceIndex;
    uint32_t meshIndex;
    uint32_t materialIndex;
} prosper_DrawInstance;
PROSPER_PT_STATIC_ASSERT(sizeof(prosper_DrawInstance) == 12, "DrawInstance is 12 B");

/* geometry_metadata.h:11-29.  Offsets are in u32 units into geometry buffer `bufferIndex`
 * (indices/meshletVertices in u16 units when usesShortIndices == 1, meshletTriangles in bytes);
 * 0xFFFFFFFF marks an absent attribute. */
typedef struct prosper_GeometryMetadata
{
    uint32_t bufferIndex;
    uint32_t indicesOffset;
    uint32_t positionsOffset;
    uint32_t normalsOffset;
    uint32_t tangentsOffset;
    uint32_t texCoord0sOffset;
    uint32_t meshletsOffset;
    uint32_t meshletBoundsOffset;
    uint32_t meshletVerticesOffset;
    uint32_t meshletTrianglesByteOffset;
    uint32_t usesShortIndices;
} prosper_GeometryMetadata;
PROSPER_PT_STATIC_ASSERT(sizeof(prosper_GeometryMetadata) == 44, "GeometryMetadata is 44 B");
#define PROSPER_PT_ABSENT 0xFFFFFFFFu

/* material_data.h:36-68 */
enum
{
    PROSPER_ALPHA_MODE_OPAQUE = 0,
    PROSPER_ALPHA_MODE_MASK = 1,
    PROSPER_ALPHA_MODE_BLEND = 2,
};
typedef struct prosper_MaterialData
{
    prosper_vec4 baseColorFactor;             /* 1 */
    float metallicFactor;                     /* 1 */
    float roughnessFactor;                    /* 1 */
    float alphaCutoff;                        /* 0.5 */
    uint32_t alphaMode;                       /* Opaque */
    uint32_t baseColorTextureSampler;         /* (sampler << 24) | texture; texture 0 = none */
    uint32_t metallicRoughnessTextureSampler;
    uint32_t normalTextureSampler;
    uint32_t pad;
} prosper_MaterialData;
PROSPER_PT_STATIC_ASSERT(sizeof(prosper_MaterialData) == 48, "MaterialData is 48 B");

/* model_instance_transforms.h:11-15.  Each mat3x4 holds the three ROWS of the affine transform
 * as its three vec4 columns (World.cpp:405-407: modelToWorld = transpose(M4),
 * normalToWorld = mat3x4(inverse(M4))). */
typedef struct prosper_ModelInstanceTransforms
{
    prosper_mat3x4 modelToWorld;
    prosper_mat3x4 normalToWorld;
} prosper_ModelInstanceTransforms;
PROSPER_PT_STATIC_ASSERT(sizeof(prosper_ModelInstanceTransforms) == 96, "ModelInstanceTransforms is 96 B");

/* lights.h:16-33 */
typedef struct prosper_DirectionalLightParameters
{
    prosper_vec4 irradiance; /* (2,2,2,2) */
    prosper_vec4 direction;  /* (-1,-1,-1,1), un-normalised */
} prosper_DirectionalLightParameters;
typedef struct prosper_PointLight
{
    prosper_vec4 radianceAndRadius;
    prosper_vec4 position;
} prosper_PointLight;
typedef struct prosper_SpotLight
{
    prosper_vec4 radianceAndAngleScale;
    prosper_vec4 positionAndAngleOffset;
    prosper_vec4 direction;
} prosper_SpotLight;
PROSPER_PT_STATIC_ASSERT(sizeof(prosper_DirectionalLightParameters) == 32, "DirectionalLightParameters is 32 B");
PROSPER_PT_STATIC_ASSERT(sizeof(prosper_PointLight) == 32, "PointLight is 32 B");
PROSPER_PT_STATIC_ASSERT(sizeof(prosper_SpotLight) == 48, "SpotLight is 48 B");

/* SSBO images: fixed-capacity array then the count (lights.glsl:11-23, Light.cpp:15-33). */
#define PROSPER_MAX_POINT_LIGHT_COUNT 1024
#define PROSPER_MAX_SPOT_LIGHT_COUNT 1024
typedef struct prosper_PointLightsBuffer
{
    prosper_PointLight lights[PROSPER_MAX_POINT_LIGHT_COUNT];
    uint32_t count;
} prosper_PointLightsBuffer;
typedef struct prosper_SpotLightsBuffer
{
    prosper_SpotLight lights[PROSPER_MAX_SPOT_LIGHT_COUNT];
    uint32_t count;
} prosper_SpotLightsBuffer;
/* The reference's byte sizes are 32772 / 49156 (Light.hpp:45-46,62-63); the C structs pad the
 * trailing count to the struct's 4-byte alignment, which is the same number. */
PROSPER_PT_STATIC_ASSERT(sizeof(prosper_PointLightsBuffer) == 32772, "PointLights SSBO is 32772 B");
PROSPER_PT_STATIC_ASSERT(sizeof(prosper_SpotLightsBuffer) == 49156, "SpotLights SSBO is 49156 B");

/* RtReference::sMaxBounces (src/render/RtReference.hpp:22) = MAX_BOUNCES in main.rgen:241. */
#define PROSPER_RT_MAX_BOUNCES 6
/* sFramePeriod, src/render/RtReference.cpp:31 */
#define PROSPER_RT_FRAME_PERIOD 4096

#ifdef __cplusplus
}
#endif

#endif /* PROSPER_PT_SHADER_STRUCTS_H */
