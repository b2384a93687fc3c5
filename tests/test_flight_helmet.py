"""The reference's one real asset (res/glTF/FlightHelmet, prosper's default scene: src/main.cpp:32-33) on the hot path,
from the packed fixture tests/golden/flight_helmet.npz (geometry in prosper's blob format byte for byte, written by
tests/golden/make_flight_helmet.py; prosper_amd/flight_helmet.py says what was reduced and why)."""
import os

import numpy as np
import pytest

from conftest import default_pc, same_bits
from prosper_amd import flight_helmet, structs as S

W, H = 480, 270
REFERENCE_GLTF = "/root/reference/res/glTF/FlightHelmet/glTF/FlightHelmet.gltf"


@pytest.fixture(scope="module")
def helmet():
    return flight_helmet.load_fixture()


def _camera(oracle, world, w=W, h=H):
    c = world.camera
    return oracle.camera_uniforms(c["eye"], c["target"], c["up"], c["fov"], c["zN"], c["zF"], w, h)


def test_fixture_holds_the_asset(helmet):
    assert helmet.triangle_count() == 94722
    assert len(helmet.materials) == 6 and len(helmet.textures) == 16 and len(helmet.metadatas) == 5
    blend = [i for i, m in enumerate(helmet.materials) if m.alphaMode == S.ALPHA_MODE_BLEND]
    assert blend == [3]  # the lenses (glTF material 2 -> index 3, WorldData.cpp:756-828)
    assert all(m.usesShortIndices == 1 for m in helmet.metadatas)  # every primitive has <= 65535 vertices
    assert list(helmet.extra["missing_images"]) == sorted([
        "FlightHelmet_occlusionRoughnessMetallic.png", "FlightHelmet_baseColor1.png",
        "FlightHelmet_occlusionRoughnessMetallic1.png", "FlightHelmet_normal1.png",
        "FlightHelmet_occlusionRoughnessMetallic4.png"])  # /root/reference/.MISSING_LARGE_BLOBS


@pytest.mark.skipif(not os.path.exists(REFERENCE_GLTF), reason="the reference mount is not present (GPU box)")
def test_fixture_geometry_is_the_ingested_asset_byte_for_byte(helmet):
    from prosper_amd import gltf
    src = gltf.load_gltf(REFERENCE_GLTF, load_images=False)
    assert np.array_equal(np.concatenate(src._buffers[0]), np.concatenate(helmet._buffers[0]))
    for a, b in zip(src.metadatas, helmet.metadatas):
        assert bytes(a) == bytes(b)
    for (ma, ta), (mb, tb) in zip(src.model_instances, helmet.model_instances):
        assert ma == mb and np.array_equal(ta, tb)


def test_oracle_renders_the_helmet(oracle, helmet):
    """CPU-side smoke of the fixture: primary hits on all five meshes, BLEND lenses included, finite radiance."""
    w, h = 160, 90
    cam, fl = _camera(oracle, helmet, w, h)
    osc = oracle.OracleScene(helmet)
    ids, _ = osc.render(default_pc(S, fl, draw_type=S.DrawType["MaterialID"], max_bounces=1), cam, w, h)
    colours = {tuple(c) for c in ids[..., :3].reshape(-1, 3).round(5)}
    assert len(colours) >= 5  # background + at least four of the five materials in view
    img, counters = osc.render(default_pc(S, fl, max_bounces=4, ibl=True), cam, w, h)
    assert np.isfinite(img).all() and img[..., :3].max() > 0.0
    assert counters.as_dict()["closestHits"] > w * h // 8


@pytest.mark.gpu
@pytest.mark.parametrize("draw_type", ["Default", "PrimitiveID", "MeshID", "MaterialID", "Position", "ShadingNormal",
                                       "TexCoord0", "Albedo", "Roughness", "Metallic"])
def test_flight_helmet_bit_exact(gpu_ctx, oracle, helmet, draw_type):
    """480x270: every DrawType, and Default over three accumulated frames with IBL (stochastic transparency on the
    lenses, normal maps, u16 indices) - HIP path == oracle, bit for bit."""
    cam, fl = _camera(oracle, helmet)
    gpu_ctx.upload_scene(helmet)
    st = gpu_ctx.scene_stats()
    assert st.triangleCount == 94722
    osc = oracle.OracleScene(helmet)
    frames = (1, 2, 3) if draw_type == "Default" else (1,)
    want = None
    for frame in frames:
        pc = default_pc(S, fl, frame_index=frame, draw_type=S.DrawType[draw_type], max_bounces=4, ibl=True,
                        skip_history=(frame == 1))
        gpu_ctx.render(pc, cam, W, H)
        want, _ = osc.render(pc, cam, W, H, history=want)
    got = gpu_ctx.read_hdr()
    ok = same_bits(got, want).all(axis=2)
    assert ok.all(), "%s: %d of %d pixels differ" % (draw_type, (~ok).sum(), ok.size)
    if draw_type == "Default":
        assert np.isfinite(got).all() and (got[..., 3] == 3).all()


@pytest.mark.gpu
@pytest.mark.parametrize("draw_type", ["Default", "Albedo", "ShadingNormal", "Roughness"])
def test_flight_helmet_with_large_textures_bit_exact(gpu_ctx, oracle, draw_type):
    """The same asset with every texture blown up to 1024 x 1024 (63 MB of texels: past the threshold at which opaque
    materials get the compact 8-byte texture packs and the shade kernel batches its texel loads; the BLEND lenses keep the
    16-byte pack): HIP path == oracle, bit for bit, on what the packs feed."""
    from prosper_amd import flight_helmet
    world = flight_helmet.load_fixture(texture_size=1024)
    cam, fl = _camera(oracle, world)
    gpu_ctx.upload_scene(world)
    assert gpu_ctx.scene_stats().variantFlags & S.VARIANT_TEXTURE_PACKS
    osc = oracle.OracleScene(world)
    frames = (1, 2) if draw_type == "Default" else (1,)
    want = None
    for frame in frames:
        pc = default_pc(S, fl, frame_index=frame, draw_type=S.DrawType[draw_type], max_bounces=4, ibl=True,
                        skip_history=(frame == 1))
        gpu_ctx.render(pc, cam, W, H)
        want, _ = osc.render(pc, cam, W, H, history=want)
    ok = same_bits(gpu_ctx.read_hdr(), want).all(axis=2)
    assert ok.all(), "%s: %d of %d pixels differ" % (draw_type, (~ok).sum(), ok.size)
