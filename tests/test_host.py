"""Host-side mirrors: mesh packing / blob layout (H5), transforms (H2), lights (H4), camera (H3),
tile partition — CPU only."""
import ctypes as C
import math

import numpy as np
import pytest

from prosper_amd import scenes, structs as S, tiling, world as W


def test_pack_mesh_matches_oracle_restatement(oracle):
    rng = np.random.default_rng(3)
    n = 500
    pos = (rng.standard_normal((n, 3)) * 3).astype(np.float32)
    nrm = rng.standard_normal((n, 3)).astype(np.float32)
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    tan = np.concatenate([nrm[::-1], np.where(rng.random((n, 1)) < 0.5, -1.0, 1.0)], axis=1).astype(np.float32)
    uv = (rng.random((n, 2)) * 4 - 1).astype(np.float32)
    packed = W.pack_mesh_data(pos, nrm, tan, uv)
    op = np.zeros(n, np.uint64)
    on, ot, ou = np.zeros(n, np.uint32), np.zeros(n, np.uint32), np.zeros(n, np.uint32)
    oracle.lib().ora_pack_mesh(pos.ctypes.data, nrm.ctypes.data, tan.ctypes.data, uv.ctypes.data, n,
                               op.ctypes.data, on.ctypes.data, ot.ctypes.data, ou.ctypes.data)
    np.testing.assert_array_equal(packed["positions"].reshape(-1).view(np.uint64), op)
    np.testing.assert_array_equal(packed["normals"], on)
    np.testing.assert_array_equal(packed["tangents"], ot)
    np.testing.assert_array_equal(packed["uvs"], ou)
    # positions carry w = 1.0 (packHalf4x16(vec4(p, 1)), DeferredLoadingContext.cpp:454)
    assert ((op >> np.uint64(48)) == 0x3C00).all()


def test_geometry_blob_layout_and_offsets():
    w = W.World()
    m = w.add_material()
    mesh = scenes.box()
    i0 = w.add_mesh(mesh[0], mesh[4], m, normals=mesh[1], tangents=mesh[2], uvs=mesh[3])
    i1 = w.add_mesh(mesh[0], mesh[4], m, normals=mesh[1], force_u32_indices=True)
    md0, md1 = w.metadatas[i0], w.metadatas[i1]
    nv, ni = 24, 36
    # blob order: indices (u16 padded to 4 B) . positions . normals . tangents . uvs
    assert md0.usesShortIndices == 1 and md0.indicesOffset == 0
    assert md0.positionsOffset == ni // 2 and md0.normalsOffset == md0.positionsOffset + 2 * nv
    assert md0.tangentsOffset == md0.normalsOffset + nv and md0.texCoord0sOffset == md0.tangentsOffset + nv
    end0 = md0.texCoord0sOffset + nv
    assert md1.usesShortIndices == 0 and md1.indicesOffset == end0 and md1.positionsOffset == end0 + ni
    assert md1.tangentsOffset == S.ABSENT and md1.texCoord0sOffset == S.ABSENT and md1.meshletsOffset == S.ABSENT
    buf = w.freeze()["geometry_buffers"][0]
    np.testing.assert_array_equal(buf.view(np.uint16)[:ni], mesh[4].astype(np.uint16))
    np.testing.assert_array_equal(buf[md1.indicesOffset:md1.indicesOffset + ni], mesh[4])


def test_instance_transforms_and_draw_instance_order():
    w = W.World()
    m0, m1 = w.add_material(), w.add_material()
    b = scenes.box()
    mesh_a = w.add_mesh(b[0], b[4], m0)
    mesh_b = w.add_mesh(b[0], b[4], m1)
    model = w.add_model([(mesh_a, m0), (mesh_b, m1)])
    single = w.add_model([(mesh_b, m0)])
    M = W.translate((1, 2, 3)) @ W.rotate_y(0.4) @ W.scale((2, 1, 0.5))
    w.add_instance(model, M)
    w.add_instance(single)
    f = w.freeze()
    # DrawInstances: model instances in order, sub-models packed tightly (World.cpp:480-513)
    di = [(d.modelInstanceIndex, d.meshIndex, d.materialIndex) for d in f["draw_instances"][:3]]
    assert di == [(0, mesh_a, m0), (0, mesh_b, m1), (1, mesh_b, m0)]
    t = f["transforms"][0]
    rows = np.array([[t.modelToWorld.col[r].x, t.modelToWorld.col[r].y, t.modelToWorld.col[r].z, t.modelToWorld.col[r].w]
                     for r in range(3)])
    np.testing.assert_allclose(rows, M[:3, :], rtol=1e-6)
    # normal * mat3(normalToWorld) = inverse-transpose(M) * normal
    n2w = np.array([[t.normalToWorld.col[c].x, t.normalToWorld.col[c].y, t.normalToWorld.col[c].z] for c in range(3)])
    n = np.array([0.3, -0.5, 0.8])
    np.testing.assert_allclose(n2w @ n, np.linalg.inv(M[:3, :3]).T @ n, rtol=1e-5)


def test_light_conversion_follows_worlddata():
    w = W.World()
    w.add_point_light((1.0, 0.5, 0.25), 40.0, (1, 2, 3))
    L = w.point_lights.lights[0]
    rad = np.array([1.0, 0.5, 0.25]) * 40.0 / (4 * math.pi)
    np.testing.assert_allclose([L.radianceAndRadius.x, L.radianceAndRadius.y, L.radianceAndRadius.z], rad, rtol=1e-6)
    lum = float(rad @ np.array([0.2126, 0.7152, 0.0722]))
    assert L.radianceAndRadius.w == pytest.approx(math.sqrt(lum / 0.01), rel=1e-6)   # WorldData.cpp:1486-1492
    w.add_spot_light((1, 1, 1), 10.0, (0, 1, 0), (0, -1, 0), math.radians(20), math.radians(35))
    sp = w.spot_lights.lights[0]
    scale = 1.0 / (math.cos(math.radians(20)) - math.cos(math.radians(35)))
    assert sp.radianceAndAngleScale.w == pytest.approx(scale, rel=1e-6)                # WorldData.cpp:1509-1514
    assert sp.positionAndAngleOffset.w == pytest.approx(-math.cos(math.radians(35)) * scale, rel=1e-6)
    w.freeze()
    # punctual lights but no sun: the default directional light is zeroed (WorldData.cpp:1537-1542)
    assert (w.directional.irradiance.x, w.directional.irradiance.y, w.directional.irradiance.z) == (0, 0, 0)
    assert w.point_lights.count == 1 and w.spot_lights.count == 1


def test_default_directional_light_survives_without_punctual_lights():
    w = W.World()
    w.freeze()
    assert w.directional.irradiance.x == 2.0 and w.directional.direction.x == -1.0  # lights.h:9,18-19


def test_host_camera_matches_oracle_restatement(oracle):
    from prosper_amd.rt_reference import Camera
    cam = Camera()
    eye, target, up = (0.0, 1.0, 3.4), (0.0, 1.0, 0.0), (0.0, 1.0, 0.0)
    cam.set_parameters(math.radians(40.0), 0.1, 100.0)
    cam.look_at(eye, target, up)
    cam.update_resolution(1920, 1080)
    assert cam.changed_this_frame()
    u, focal = cam.update_buffer()
    want, want_focal = oracle.camera_uniforms(eye, target, up, math.radians(40.0), 0.1, 100.0, 1920, 1080)
    a = np.frombuffer(u, dtype=np.float32, count=96).copy()       # 6 matrices
    b = np.frombuffer(want, dtype=np.float32, count=96).copy()
    np.testing.assert_allclose(a[:64], b[:64], rtol=1e-5, atol=1e-6)  # w2c, c2w, c2c, clipToWorld
    assert focal == pytest.approx(want_focal, rel=1e-6)
    # Y-flip + reverse-z facts the ray generation relies on
    assert u.cameraToClip.col[1].y < 0 and u.cameraToClip.col[0].x > 0
    assert u.cameraToClip.col[1].y == pytest.approx(-1.0 / math.tan(math.radians(20.0)), rel=1e-6)
    assert (u.eye.x, u.eye.y, u.eye.z, u.eye.w) == (0.0, 1.0, pytest.approx(3.4), 1.0)
    cam.end_frame()
    assert not cam.changed_this_frame()
    cam.look_at(eye, (0.0, 1.0, -1.0), up)
    assert cam.changed_this_frame()


def test_tile_partition_covers_the_image_exactly_once():
    for width, world in ((1920, 1), (1920, 2), (1920, 4), (1920, 8), (3840, 8)):
        assert tiling.check_divisible(width, world)
        cols = []
        for r in range(world):
            lw = tiling.local_width(width, r, world)
            assert lw == width // world
            t = tiling.tile_for_rank(r, world)
            for lx in range(lw):
                if t is None:
                    cols.append(lx)
                else:
                    ls = lx // t.stripeWidth
                    cols.append((ls * t.stripeCount + t.stripeIndex) * t.stripeWidth + lx % t.stripeWidth)
        assert sorted(cols) == list(range(width))
    # 1080 and 2160 rows tile into 8-row wave tiles; stripes are 2 wave tiles wide
    assert 1080 % 8 == 0 and 2160 % 8 == 0 and tiling.STRIPE_WIDTH % 8 == 0


def test_deinterleave_inverts_the_partition():
    h, w, world = 4, 128, 4
    full = np.arange(h * w * 4, dtype=np.float32).reshape(h, w, 4)
    tiles = []
    for r in range(world):
        cols = [x for x in range(w) if (x // 16) % world == r]
        tiles.append(np.ascontiguousarray(full[:, cols, :]))
    np.testing.assert_array_equal(tiling.deinterleave(tiles, w), full)


def test_mesh_ranges_tile_the_geometry_buffers():
    """World.mesh_ranges - the bytes prosper_pt_update_meshes hands over per mesh (UploadedGeometryData's range) - derived from
    the metadata alone: the ranges of a buffer's meshes are disjoint and cover it, and with_meshes_loaded() leaves exactly
    those words of the arrived meshes in place."""
    from prosper_amd import scenes
    for world in (scenes.cornell(), scenes.alpha_wall(), scenes.sponza_class(texture_size=16, sky_size=8, detail=0.25)):
        f = world.freeze()
        ranges = world.mesh_ranges
        assert len(ranges) == len(world.metadatas)
        for b, buf in enumerate(f["geometry_buffers"]):
            mine = sorted((first, words) for (bi, first, words) in ranges if bi == b)
            at = 0
            for first, words in mine:
                assert first == at and words > 0
                at += words
            assert at == buf.size
        loaded = set(range(0, len(world.metadatas), 2))
        partial = world.with_meshes_loaded(loaded).freeze()
        for i, (b, first, words) in enumerate(ranges):
            got = partial["geometry_buffers"][b][first:first + words]
            if i in loaded:
                assert (got == f["geometry_buffers"][b][first:first + words]).all()
                assert partial["metadatas"][i].bufferIndex == b
            else:
                assert not got.any() and partial["metadatas"][i].bufferIndex == S.ABSENT
