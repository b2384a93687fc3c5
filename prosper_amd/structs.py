"""ctypes mirrors of include/prosper_pt/{shader_structs,prosper_pt}.h.

Layouts follow prosper's shared host/device structs (reference:
res/shader/shared/shader_structs/scene/*.h, push_constants/rt_reference.h); sizes are checked
against the C headers' static asserts in tests/test_structs.py.
"""
import ctypes as C

import numpy as np


class Vec4(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float), ("z", C.c_float), ("w", C.c_float)]


class Mat4(C.Structure):
    _fields_ = [("col", Vec4 * 4)]


class Mat3x4(C.Structure):
    _fields_ = [("col", Vec4 * 3)]


class ReferencePC(C.Structure):
    """push_constants/rt_reference.h:6-16"""

    _fields_ = [
        ("drawType", C.c_uint32),
        ("flags", C.c_uint32),
        ("frameIndex", C.c_uint32),
        ("apertureDiameter", C.c_float),
        ("focusDistance", C.c_float),
        ("focalLength", C.c_float),
        ("rouletteStartBounce", C.c_uint32),
        ("maxBounces", C.c_uint32),
    ]


PC_FLAG_SKIP_HISTORY = 1 << 0
PC_FLAG_ACCUMULATE = 1 << 1
PC_FLAG_IBL = 1 << 2
PC_FLAG_DEPTH_OF_FIELD = 1 << 3
PC_FLAG_CLAMP_INDIRECT = 1 << 4

# src/scene/DrawType.hpp:8-10
DRAW_TYPES = [
    "Default", "PrimitiveID", "MeshletID", "MeshID", "MaterialID", "Position", "ShadingNormal",
    "TexCoord0", "Albedo", "Roughness", "Metallic",
]
DrawType = {name: i for i, name in enumerate(DRAW_TYPES)}

RT_MAX_BOUNCES = 6      # RtReference::sMaxBounces, src/render/RtReference.hpp:22
RT_FRAME_PERIOD = 4096  # sFramePeriod, src/render/RtReference.cpp:31


class CameraUniforms(C.Structure):
    """scene/camera.h:11-34"""

    _fields_ = [
        ("worldToCamera", Mat4),
        ("cameraToWorld", Mat4),
        ("cameraToClip", Mat4),
        ("clipToWorld", Mat4),
        ("previousWorldToCamera", Mat4),
        ("previousCameraToClip", Mat4),
        ("eye", Vec4),
        ("nearPlane", Vec4),
        ("farPlane", Vec4),
        ("leftPlane", Vec4),
        ("rightPlane", Vec4),
        ("topPlane", Vec4),
        ("bottomPlane", Vec4),
        ("resolution", C.c_uint32 * 2),
        ("currentJitter", C.c_float * 2),
        ("previousJitter", C.c_float * 2),
        ("near_", C.c_float),
        ("far_", C.c_float),
        ("maxViewScale", C.c_float),
    ]


class DrawInstance(C.Structure):
    _fields_ = [("modelInstanceIndex", C.c_uint32), ("meshIndex", C.c_uint32), ("materialIndex", C.c_uint32)]


class GeometryMetadata(C.Structure):
    _fields_ = [
        ("bufferIndex", C.c_uint32),
        ("indicesOffset", C.c_uint32),
        ("positionsOffset", C.c_uint32),
        ("normalsOffset", C.c_uint32),
        ("tangentsOffset", C.c_uint32),
        ("texCoord0sOffset", C.c_uint32),
        ("meshletsOffset", C.c_uint32),
        ("meshletBoundsOffset", C.c_uint32),
        ("meshletVerticesOffset", C.c_uint32),
        ("meshletTrianglesByteOffset", C.c_uint32),
        ("usesShortIndices", C.c_uint32),
    ]


ABSENT = 0xFFFFFFFF
ALPHA_MODE_OPAQUE, ALPHA_MODE_MASK, ALPHA_MODE_BLEND = 0, 1, 2


class MaterialData(C.Structure):
    _fields_ = [
        ("baseColorFactor", Vec4),
        ("metallicFactor", C.c_float),
        ("roughnessFactor", C.c_float),
        ("alphaCutoff", C.c_float),
        ("alphaMode", C.c_uint32),
        ("baseColorTextureSampler", C.c_uint32),
        ("metallicRoughnessTextureSampler", C.c_uint32),
        ("normalTextureSampler", C.c_uint32),
        ("pad", C.c_uint32),
    ]


class ModelInstanceTransforms(C.Structure):
    _fields_ = [("modelToWorld", Mat3x4), ("normalToWorld", Mat3x4)]


class DirectionalLightParameters(C.Structure):
    _fields_ = [("irradiance", Vec4), ("direction", Vec4)]


class PointLight(C.Structure):
    _fields_ = [("radianceAndRadius", Vec4), ("position", Vec4)]


class SpotLight(C.Structure):
    _fields_ = [("radianceAndAngleScale", Vec4), ("positionAndAngleOffset", Vec4), ("direction", Vec4)]


MAX_POINT_LIGHT_COUNT = 1024
MAX_SPOT_LIGHT_COUNT = 1024


class PointLightsBuffer(C.Structure):
    _fields_ = [("lights", PointLight * MAX_POINT_LIGHT_COUNT), ("count", C.c_uint32)]


class SpotLightsBuffer(C.Structure):
    _fields_ = [("lights", SpotLight * MAX_SPOT_LIGHT_COUNT), ("count", C.c_uint32)]


# ---- prosper_pt.h ----

FORMAT_RGBA8_UNORM = 0
FORMAT_BC7_UNORM = 1
FILTER_NEAREST, FILTER_LINEAR = 0, 1
WRAP_REPEAT, WRAP_MIRRORED_REPEAT, WRAP_CLAMP_TO_EDGE = 0, 1, 2

CREATE_MEGAKERNEL = 1 << 0
CREATE_PERSISTENT = 1 << 1
CREATE_SINGLE_CHAIN = 1 << 2
RENDER_COUNT_WORK = 1 << 0
RENDER_PIPELINED = 1 << 1
MAX_KERNELS = 8


class DeviceDesc(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32),
        ("device_ordinal", C.c_int32),
        ("flags", C.c_uint32),
        ("reserved", C.c_uint32),
    ]


class TextureDesc(C.Structure):
    _fields_ = [
        ("texels", C.c_void_p),
        ("width", C.c_uint32),
        ("height", C.c_uint32),
        ("format", C.c_uint32),
        ("reserved", C.c_uint32),
    ]


class SamplerDesc(C.Structure):
    _fields_ = [("magFilter", C.c_uint32), ("minFilter", C.c_uint32), ("wrapS", C.c_uint32), ("wrapT", C.c_uint32)]


class MeshInfo(C.Structure):
    _fields_ = [
        ("vertexCount", C.c_uint32),
        ("indexCount", C.c_uint32),
        ("meshletCount", C.c_uint32),
        ("materialIndex", C.c_uint32),
    ]


class CubeDesc(C.Structure):
    _fields_ = [("texels", C.c_void_p), ("faceSize", C.c_uint32), ("reserved", C.c_uint32)]


class SceneView(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32),
        ("reserved", C.c_uint32),
        ("geometryBuffers", C.POINTER(C.c_void_p)),
        ("geometryBufferByteSizes", C.POINTER(C.c_uint64)),
        ("geometryBufferCount", C.c_uint32),
        ("meshCount", C.c_uint32),
        ("geometryMetadatas", C.POINTER(GeometryMetadata)),
        ("meshInfos", C.POINTER(MeshInfo)),
        ("drawInstances", C.POINTER(DrawInstance)),
        ("drawInstanceCount", C.c_uint32),
        ("modelInstanceCount", C.c_uint32),
        ("modelInstanceTransforms", C.POINTER(ModelInstanceTransforms)),
        ("materials", C.POINTER(MaterialData)),
        ("materialCount", C.c_uint32),
        ("textureCount", C.c_uint32),
        ("textures", C.POINTER(TextureDesc)),
        ("samplers", C.POINTER(SamplerDesc)),
        ("samplerCount", C.c_uint32),
        ("reserved2", C.c_uint32),
        ("directionalLight", C.POINTER(DirectionalLightParameters)),
        ("pointLights", C.POINTER(PointLightsBuffer)),
        ("spotLights", C.POINTER(SpotLightsBuffer)),
        ("skybox", CubeDesc),
    ]


class TileDesc(C.Structure):
    _fields_ = [("stripeWidth", C.c_uint32), ("stripeIndex", C.c_uint32), ("stripeCount", C.c_uint32)]


class Counters(C.Structure):
    _fields_ = [
        ("paths", C.c_uint64),
        ("closestRays", C.c_uint64),
        ("shadowRays", C.c_uint64),
        ("nodeVisits", C.c_uint64),
        ("triangleTests", C.c_uint64),
        ("closestHits", C.c_uint64),
        ("anyHitCalls", C.c_uint64),
        ("lightSamples", C.c_uint64),
        ("spotLightSamples", C.c_uint64),
        ("skyLookups", C.c_uint64),
        ("pixelsWritten", C.c_uint64),
        ("historyReads", C.c_uint64),
        ("shortIndexHits", C.c_uint64),
        ("shortIndexTriangleTests", C.c_uint64),
        ("nodePhaseSteps", C.c_uint64),
        ("trianglePhaseSteps", C.c_uint64),
        ("anyHitTexelFetches", C.c_uint64),
    ]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_ if n != "reserved"}


class SceneStats(C.Structure):
    _fields_ = [
        ("triangleCount", C.c_uint64),
        ("nodeCount", C.c_uint64),
        ("nodeBytes", C.c_uint32),
        ("triangleBytes", C.c_uint32),
        ("maxDepth", C.c_uint32),
        ("variantFlags", C.c_uint32),
        ("deviceBytes", C.c_uint64),
        ("buildSeconds", C.c_double),
        ("uploadSeconds", C.c_double),
        ("bvhBuildSeconds", C.c_double),
        ("textureSeconds", C.c_double),
        ("alphaTriangleCount", C.c_uint64),
        ("alphaBoundBytes", C.c_uint64),
    ]


class HierarchyState(C.Structure):
    _fields_ = [("refits", C.c_uint32), ("rebuilds", C.c_uint32), ("costRatio", C.c_float), ("builtCost", C.c_float),
                ("nodeCount", C.c_uint32), ("levels", C.c_uint32), ("meshUpdates", C.c_uint32), ("geometryInstalls", C.c_uint32),
                ("geometryBuildRunning", C.c_uint32), ("reserved", C.c_uint32)]


class DebugOptions(C.Structure):
    """prosper_pt_debug_options (include/prosper_pt/prosper_pt.h): tuning / test options of a context."""

    _fields_ = [("struct_size", C.c_uint32),
                ("batchedTextures", C.c_int32), ("widePacks", C.c_int32), ("alphaCellShift", C.c_int32),
                ("noTexturePacks", C.c_uint32), ("noAlphaBounds", C.c_uint32), ("noUploadRefit", C.c_uint32), ("flatBvh", C.c_uint32),
                ("sahTraversalCost", C.c_float), ("boxPad", C.c_float), ("leafSize", C.c_uint32), ("buildThreads", C.c_uint32),
                ("topEntries", C.c_uint32), ("nodeOrder", C.c_int32), ("childOrder", C.c_int32), ("buildTiming", C.c_uint32),
                ("segments", C.c_uint32), ("segmentLength", C.c_uint32), ("chains", C.c_uint32), ("ldsStackEntries", C.c_uint32),
                ("noLdsScene", C.c_uint32), ("noLdsTables", C.c_uint32), ("traceDeadPaths", C.c_uint32), ("bandedBatches", C.c_int32),
                ("rebuildCostRatio", C.c_float), ("alwaysRebuild", C.c_uint32), ("failNextUpdate", C.c_uint32),
                ("poolVariant", C.c_uint32), ("rawRecords", C.c_uint32), ("tileOrder", C.c_uint32), ("hipGraph", C.c_uint32),
                ("pipelinedChains", C.c_uint32), ("mergeLimit", C.c_uint32)]


class CommInfo(C.Structure):
    _fields_ = [("ranks", C.c_uint32), ("rank", C.c_uint32), ("device", C.c_int32), ("gathers", C.c_uint32),
                ("lastGatherMs", C.c_float), ("reserved", C.c_uint32)]


class MeshUpdate(C.Structure):
    """prosper_pt_mesh_update: one streamed-in mesh (WorldData::pollMeshWorker)."""
    _fields_ = [("meshIndex", C.c_uint32), ("reserved", C.c_uint32), ("metadata", GeometryMetadata), ("info", MeshInfo),
                ("bytes", C.c_void_p), ("byteOffset", C.c_uint64), ("byteCount", C.c_uint64), ("bufferByteSize", C.c_uint64)]


MAX_GEOMETRY_BUFFERS = 100
GATHER_IN_STREAM = 1
UPDATE_NOW = 1
VARIANT_LDS_SCENE = 1
VARIANT_LDS_TABLES = 2
VARIANT_BATCHED_TEXTURES = 4
VARIANT_TEXTURE_PACKS = 8
VARIANT_RAW_RECORDS = 16
VARIANT_STACK_SHIFT = 8


def as_numpy(struct_array, dtype=np.uint8):
    """View a ctypes structure/array as a numpy byte array (no copy)."""
    return np.frombuffer(struct_array, dtype=dtype)


class RestirTracePC(C.Structure):
    """TracePC, res/shader/shared/shader_structs/push_constants/restir_di/trace.h"""
    _fields_ = [("drawType", C.c_uint32), ("frameIndex", C.c_uint32), ("flags", C.c_uint32)]


class RestirInputs(C.Structure):
    _fields_ = [("albedoRoughness", C.c_void_p), ("normalMetallic", C.c_void_p), ("nonLinearDepth", C.c_void_p),
                ("reservoirs", C.c_void_p), ("onDevice", C.c_uint32), ("reserved", C.c_uint32)]
