"""Data contract: ctypes mirrors == C headers == the reference's stated sizes."""
import ctypes as C
import os
import subprocess
import tempfile

from prosper_amd import structs as S

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

NAMES = {
    "prosper_ReferencePC": S.ReferencePC, "prosper_CameraUniforms": S.CameraUniforms,
    "prosper_DrawInstance": S.DrawInstance, "prosper_GeometryMetadata": S.GeometryMetadata,
    "prosper_MaterialData": S.MaterialData, "prosper_ModelInstanceTransforms": S.ModelInstanceTransforms,
    "prosper_DirectionalLightParameters": S.DirectionalLightParameters, "prosper_PointLight": S.PointLight,
    "prosper_SpotLight": S.SpotLight, "prosper_PointLightsBuffer": S.PointLightsBuffer,
    "prosper_SpotLightsBuffer": S.SpotLightsBuffer, "prosper_pt_device_desc": S.DeviceDesc,
    "prosper_pt_texture_desc": S.TextureDesc, "prosper_pt_sampler_desc": S.SamplerDesc,
    "prosper_pt_mesh_info": S.MeshInfo, "prosper_pt_cube_desc": S.CubeDesc, "prosper_pt_scene_view": S.SceneView,
    "prosper_pt_tile_desc": S.TileDesc, "prosper_pt_counters": S.Counters, "prosper_pt_scene_stats": S.SceneStats,
    "prosper_pt_mesh_update": S.MeshUpdate, "prosper_pt_debug_options": S.DebugOptions, "prosper_pt_comm_info": S.CommInfo,
}


def test_reference_struct_sizes():
    # sizes stated by the reference layouts (12/44/48/96/32/48/32/32 B, 532 B camera, SSBO byte sizes)
    assert C.sizeof(S.ReferencePC) == 32
    assert C.sizeof(S.CameraUniforms) == 532
    assert C.sizeof(S.DrawInstance) == 12
    assert C.sizeof(S.GeometryMetadata) == 44
    assert C.sizeof(S.MaterialData) == 48
    assert C.sizeof(S.ModelInstanceTransforms) == 96
    assert C.sizeof(S.DirectionalLightParameters) == 32
    assert C.sizeof(S.PointLight) == 32
    assert C.sizeof(S.SpotLight) == 48
    assert C.sizeof(S.PointLightsBuffer) == 1024 * 32 + 4   # Light.hpp:45-46
    assert C.sizeof(S.SpotLightsBuffer) == 1024 * 48 + 4    # Light.hpp:62-63
    assert S.PointLightsBuffer.count.offset == 32768 and S.SpotLightsBuffer.count.offset == 49152


def test_ctypes_match_c_headers():
    src = "#include <stdio.h>\n#include \"prosper_pt/prosper_host.h\"\nint main(void){\n"
    for name in NAMES:
        src += '  printf("%s %%zu\\n", sizeof(%s));\n' % (name, name)
    src += "  return 0;}\n"
    with tempfile.TemporaryDirectory() as d:
        c = os.path.join(d, "sizes.c")
        open(c, "w").write(src)
        exe = os.path.join(d, "sizes")
        subprocess.check_call(["gcc", "-std=c11", "-I", os.path.join(ROOT, "include"), c, "-o", exe])
        out = subprocess.check_output([exe]).decode().split("\n")
    sizes = dict(line.split() for line in out if line)
    for name, cls in NAMES.items():
        assert int(sizes[name]) == C.sizeof(cls), name


def test_flag_bits_and_draw_types():
    # RtReference.cpp:77-88, DrawType.hpp:8-10
    assert (S.PC_FLAG_SKIP_HISTORY, S.PC_FLAG_ACCUMULATE, S.PC_FLAG_IBL, S.PC_FLAG_DEPTH_OF_FIELD,
            S.PC_FLAG_CLAMP_INDIRECT) == (1, 2, 4, 8, 16)
    assert S.DRAW_TYPES == ["Default", "PrimitiveID", "MeshletID", "MeshID", "MaterialID", "Position", "ShadingNormal",
                            "TexCoord0", "Albedo", "Roughness", "Metallic"]
