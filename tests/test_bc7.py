"""BC7 decoder and the texture-cache DDS reader against an independent decoder (Pillow's "bcn").

The reference holds no BC7 test data (it links a third-party encoder and lets the GPU decode), so the pin is
Pillow: random blocks of every mode must decode to the same bytes.  One documented difference: a reserved block
(mode byte 0) decodes to (0, 0, 0, 0) here, as the format specifies; Pillow returns opaque black.
"""
import os
import shutil
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))

from prosper_amd import bc7, dds, gltf  # noqa: E402

PIL_Image = pytest.importorskip("PIL.Image")


def pillow_decode(blocks):
    return np.stack([np.asarray(PIL_Image.frombytes("RGBA", (4, 4), bytes(b), "bcn", (7,))).reshape(16, 4)
                     for b in blocks])


def random_blocks(mode, count, seed):
    rng = np.random.default_rng(seed)
    blocks = rng.integers(0, 256, (count, 16), dtype=np.uint8)
    blocks[:, 0] = (blocks[:, 0] & ~np.uint8((1 << (mode + 1)) - 1)) | np.uint8(1 << mode)
    return blocks


def test_tables_match_a_fresh_probe_of_pillow():
    import make_bc7_tables
    p2, p3, a2, a3a, a3b = make_bc7_tables.probe()
    assert ["".join(map(str, r)) for r in p2] == list(bc7.PARTITION2)
    assert ["".join(map(str, r)) for r in p3] == list(bc7.PARTITION3)
    assert a2 == list(bc7.ANCHOR2) and a3a == list(bc7.ANCHOR3A) and a3b == list(bc7.ANCHOR3B)


def test_tables_are_consistent():
    for rows, subsets in ((bc7.PARTITION2, 2), (bc7.PARTITION3, 3)):
        assert len(rows) == 64 and len(set(rows)) == 64
        for r in rows:
            assert len(r) == 16 and r[0] == "0" and set(r) == set("012"[:subsets])
    for p in range(64):
        assert bc7.PARTITION2[p][bc7.ANCHOR2[p]] == "1"
        assert bc7.PARTITION3[p][bc7.ANCHOR3A[p]] == "1" and bc7.PARTITION3[p][bc7.ANCHOR3B[p]] == "2"


@pytest.mark.parametrize("mode", range(8))
def test_random_blocks_decode_like_pillow(mode):
    blocks = random_blocks(mode, 1500, 100 + mode)
    assert np.array_equal(bc7.decode_blocks(blocks), pillow_decode(blocks))


def test_every_partition_of_every_multi_subset_mode():
    rng = np.random.default_rng(5)
    for mode, bits in ((0, 4), (1, 6), (2, 6), (3, 6), (7, 6)):
        blocks = random_blocks(mode, 1 << bits, 200 + mode).astype(np.uint16)
        # overwrite the partition field (right after the mode bits) with 0 .. 2^bits - 1
        words = blocks[:, 0] | (blocks[:, 1] << 8)
        mask = ((1 << bits) - 1) << (mode + 1)
        words = (words & ~np.uint16(mask)) | (np.arange(1 << bits, dtype=np.uint16) << (mode + 1))
        blocks[:, 0], blocks[:, 1] = words & 0xFF, words >> 8
        blocks = blocks.astype(np.uint8)
        assert np.array_equal(bc7.decode_blocks(blocks), pillow_decode(blocks)), mode
    del rng


def test_reserved_block_is_transparent_black():
    blocks = random_blocks(3, 4, 9)
    blocks[:, 0] = 0
    assert not bc7.decode_blocks(blocks).any()


def test_solid_colour_mode6_block_by_hand():
    # mode 6, both endpoints (r, g, b, a) = 7-bit 100 with p-bit 1 -> 201, every index 0
    value, pos = 0, 0

    def put(v, n):
        nonlocal value, pos
        value |= v << pos
        pos += n
    put(1 << 6, 7)
    for _ in range(8):
        put(100, 7)
    put(1, 1)
    put(1, 1)
    block = np.frombuffer(value.to_bytes(16, "little"), np.uint8)
    assert (bc7.decode_blocks(block[None]) == 201).all()


def test_image_layout_and_mixed_modes():
    rng = np.random.default_rng(11)
    w, h = 24, 12
    blocks = np.concatenate([random_blocks(m, 3, 300 + m) for m in range(8)])[: (w // 4) * (h // 4)]
    rng.shuffle(blocks)
    img = bc7.decode_image(blocks.tobytes(), w, h)
    ref = np.asarray(PIL_Image.frombytes("RGBA", (w, h), blocks.tobytes(), "bcn", (7,)))
    assert img.shape == (h, w, 4) and np.array_equal(img, ref)
    with pytest.raises(ValueError):
        bc7.decode_image(blocks.tobytes(), 22, 12)


def test_texture_cache_dds_round_trip(tmp_path):
    rng = np.random.default_rng(3)
    # BC7 with a mip chain 16x8 -> 8x4 (prosper stops at 4x4 blocks: Texture.cpp:218-226)
    l0 = np.concatenate([random_blocks(m, 1, 400 + m) for m in range(8)])
    l1 = random_blocks(6, 2, 77)
    path = str(tmp_path / "a.dds")
    dds.write_texture(path, dds.DXGI_FORMAT_BC7_UNORM, 16, 8, [l0.tobytes(), l1.tobytes()])
    levels = dds.read_texture(path, levels=None)
    assert [lv.shape for lv in levels] == [(8, 16, 4), (4, 8, 4)]
    assert np.array_equal(levels[0], bc7.decode_image(l0.tobytes(), 16, 8))
    assert np.array_equal(levels[1], bc7.decode_image(l1.tobytes(), 8, 4))
    assert len(dds.read_texture(path)) == 1
    # RGBA8 (what prosper writes when a level does not divide by 4), odd size, 2 levels
    a = rng.integers(0, 256, (6, 10, 4), dtype=np.uint8)
    b = rng.integers(0, 256, (3, 5, 4), dtype=np.uint8)
    path = str(tmp_path / "b.dds")
    dds.write_texture(path, dds.DXGI_FORMAT_R8G8B8A8_UNORM, 10, 6, [a.tobytes(), b.tobytes()])
    levels = dds.read_texture(path, levels=None)
    assert np.array_equal(levels[0], a) and np.array_equal(levels[1], b)
    # error behaviour of the reference's reader: wrong magic / format / truncated payload
    blob = open(path, "rb").read()
    bad = str(tmp_path / "c.dds")
    open(bad, "wb").write(b"XXXX" + blob[4:])
    with pytest.raises(dds.DdsError):
        dds.read_texture(bad)
    open(bad, "wb").write(blob[:-7])
    with pytest.raises(dds.DdsError):
        dds.read_texture(bad, levels=None)
    with pytest.raises(dds.DdsError):
        dds.write_texture(bad, dds.DXGI_FORMAT_BC7_UNORM, 10, 6, [b"\0" * 16])
    lut = str(tmp_path / "lut.dds")
    dds.write_lut(lut, np.zeros((2, 2, 2), np.uint32))
    with pytest.raises(dds.DdsError):
        dds.read_texture(lut)


def _write_png(path, rgba):
    import struct
    import zlib
    h, w, _ = rgba.shape
    raw = b"".join(b"\0" + rgba[y].tobytes() for y in range(h))

    def chunk(kind, data):
        body = kind + data
        return struct.pack(">I", len(data)) + body + struct.pack(">I", zlib.crc32(body))
    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 6, 0, 0, 0)) +
                chunk(b"IDAT", zlib.compress(raw)) + chunk(b"IEND", b""))


def test_gltf_prefers_prosper_cache(tmp_path):
    import base64
    import json
    rng = np.random.default_rng(21)
    png = rng.integers(0, 256, (8, 8, 4), dtype=np.uint8)
    _write_png(str(tmp_path / "albedo.png"), png)
    pos = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0]], np.float32)
    nrm = np.tile(np.array([0, 0, 1], np.float32), (3, 1))
    idx = np.array([0, 1, 2], np.uint16)
    blob = pos.tobytes() + idx.tobytes() + b"\0\0" + nrm.tobytes()
    doc = {
        "asset": {"version": "2.0"},
        "buffers": [{"byteLength": len(blob), "uri": "data:application/octet-stream;base64," + base64.b64encode(blob).decode()}],
        "bufferViews": [{"buffer": 0, "byteOffset": 0, "byteLength": 36}, {"buffer": 0, "byteOffset": 36, "byteLength": 6},
                        {"buffer": 0, "byteOffset": 44, "byteLength": 36}],
        "accessors": [{"bufferView": 0, "componentType": 5126, "count": 3, "type": "VEC3", "min": [0, 0, 0], "max": [1, 1, 0]},
                      {"bufferView": 1, "componentType": 5123, "count": 3, "type": "SCALAR"},
                      {"bufferView": 2, "componentType": 5126, "count": 3, "type": "VEC3"}],
        "images": [{"uri": "albedo.png"}],
        "textures": [{"source": 0}],
        "materials": [{"pbrMetallicRoughness": {"baseColorTexture": {"index": 0}}}],
        "meshes": [{"primitives": [{"attributes": {"POSITION": 0, "NORMAL": 2}, "indices": 1, "material": 0}]}],
        "nodes": [{"mesh": 0}],
        "scenes": [{"nodes": [0]}],
        "scene": 0,
    }
    path = str(tmp_path / "scene.gltf")
    with open(path, "w") as f:
        json.dump(doc, f)
    # no cache: the PNG's texels
    w0 = gltf.load_gltf(path)
    assert not w0.cached_images and np.array_equal(w0.textures[1], png)
    # with prosper's cache next to it: level 0 of the BC7 file wins, like Texture2D::init reads the cache back
    os.makedirs(tmp_path / "prosper_cache")
    blocks = np.concatenate([random_blocks(m, 1, 500 + m) for m in (6, 1, 3, 5)])
    dds.write_texture(dds.cache_path(str(tmp_path / "albedo.png")), dds.DXGI_FORMAT_BC7_UNORM, 8, 8,
                      [blocks.tobytes(), random_blocks(6, 1, 1).tobytes()])
    cache = dds.cache_path(str(tmp_path / "albedo.png"))
    # an untagged cache is not valid (Texture.cpp:124-160): the PNG's texels are used, as prosper would re-encode
    w_untagged = gltf.load_gltf(path)
    assert not w_untagged.cached_images and w_untagged.stale_cached_images == [cache]
    assert np.array_equal(w_untagged.textures[1], png)
    assert np.array_equal(gltf.load_gltf(path, use_texture_cache="always").textures[1], bc7.decode_image(blocks.tobytes(), 8, 8))
    dds.write_cache_tag(cache, str(tmp_path / "albedo.png"))
    assert dds.cache_valid(cache, str(tmp_path / "albedo.png"))
    assert dds.read_cache_tag(cache) == (5, dds.source_write_time(str(tmp_path / "albedo.png")))
    w1 = gltf.load_gltf(path)
    assert w1.cached_images == [str(tmp_path / "prosper_cache" / "albedo.dds")]
    assert np.array_equal(w1.textures[1], bc7.decode_image(blocks.tobytes(), 8, 8))
    # touching the source makes the cache stale; an old-version tag and a wrong magic too
    st = os.stat(tmp_path / "albedo.png")
    os.utime(tmp_path / "albedo.png", ns=(st.st_atime_ns, st.st_mtime_ns + 10**9))
    assert not dds.cache_valid(cache, str(tmp_path / "albedo.png"))
    assert np.array_equal(gltf.load_gltf(path).textures[1], png)
    dds.write_cache_tag(cache, str(tmp_path / "albedo.png"))
    assert dds.cache_valid(cache, str(tmp_path / "albedo.png"))
    tag = dds.cache_tag_path(cache)
    assert tag.endswith("albedo.prosper_cache_tag")
    blob = open(tag, "rb").read()
    assert len(blob) == 20 and blob[4:12] == b"PRSPRTEX"
    open(tag, "wb").write(b"\x04\0\0\0" + blob[4:])
    assert dds.read_cache_tag(cache) == (4, None) and not dds.cache_valid(cache, str(tmp_path / "albedo.png"))
    open(tag, "wb").write(blob[:4] + b"NOTMAGIC" + blob[12:])
    with pytest.raises(dds.DdsError):
        dds.read_cache_tag(cache)
    assert not dds.cache_valid(cache, str(tmp_path / "albedo.png"))
    open(tag, "wb").write(blob)
    w2 = gltf.load_gltf(path, use_texture_cache=False)
    assert np.array_equal(w2.textures[1], png)


# ------------------------------------------------------------------------------------------------
# the HIP decoder (pt_bc7.hpp) through the C-ABI
# ------------------------------------------------------------------------------------------------

@pytest.mark.gpu
def test_device_block_decode_bit_exact():
    """bc7_decode_block on the GPU (PROSPER_PT_FN_BC7_BLOCK) against the numpy decoder: 2000 random blocks of each
    mode, every partition of the multi-subset modes, reserved blocks."""
    from prosper_amd import capi
    blocks = [random_blocks(m, 2000, 600 + m) for m in range(8)]
    reserved = random_blocks(2, 8, 1)
    reserved[:, 0] = 0
    blocks.append(reserved)
    blocks = np.concatenate(blocks)
    ctx = capi.Context(0)
    try:
        words = blocks.view("<u4").reshape(-1, 4).view(np.float32)
        out = ctx.eval_device_fn(16, words, 4, 16)
    finally:
        ctx.close()
    got = np.ascontiguousarray(out).view("<u4").reshape(-1, 16)
    want = bc7.decode_blocks(blocks).reshape(-1, 16, 4).astype(np.uint32)
    want = want[..., 0] | (want[..., 1] << 8) | (want[..., 2] << 16) | (want[..., 3] << 24)
    bad = (got != want).any(axis=1)
    assert not bad.any(), "%d blocks differ, first at %d" % (bad.sum(), np.nonzero(bad)[0][0])


def _bc7_wall(decoded):
    """A quad wall sampling three BC7 textures (base colour, metallic-roughness, normal) under every wrap mode;
    `decoded` swaps each for its RGBA8 twin decoded by prosper_amd.bc7."""
    import math
    from prosper_amd import scenes, structs as S
    from prosper_amd.world import World
    w = World()
    textures = []
    for k, (tw, th) in enumerate([(16, 8), (32, 32), (4, 4), (24, 12)]):
        n = (tw // 4) * (th // 4)
        blocks = np.concatenate([random_blocks(m, (n + 7) // 8, 700 + 10 * k + m) for m in range(8)])
        np.random.default_rng(k).shuffle(blocks)
        blocks = blocks[:n]
        if decoded:
            textures.append(w.add_texture(bc7.decode_image(blocks.tobytes(), tw, th)))
        else:
            textures.append(w.add_texture_bc7(blocks, tw, th))
    wraps = [S.WRAP_REPEAT, S.WRAP_MIRRORED_REPEAT, S.WRAP_CLAMP_TO_EDGE]
    for k, (flt, wrap) in enumerate([(f, x) for f in (S.FILTER_LINEAR, S.FILTER_NEAREST) for x in wraps]):
        smp = w.add_sampler(flt, flt, wrap, wraps[(k + 1) % 3])
        mat = w.add_material(base_color=(1.0, 1.0, 1.0, 1.0), metallic=0.5, roughness=0.8,
                             base_tex=(textures[k % 4], smp), mr_tex=(textures[(k + 1) % 4], smp),
                             normal_tex=(textures[(k + 2) % 4], smp), alpha_mode=S.ALPHA_MODE_BLEND if k == 2 else S.ALPHA_MODE_OPAQUE)
        cx, cy = (k % 3) - 1.5, (k // 3) - 1.0
        p, n_, t, uv, idx = scenes.quad((cx, cy, 0.0), (cx + 0.95, cy, 0.0), (cx + 0.95, cy + 0.95, 0.0), (cx, cy + 0.95, 0.0))
        uv = uv * (2.3 + 0.4 * k) - (0.7 + 0.3 * k)
        q = scenes._add(w, (p, n_, t, uv, idx), mat)
        w.add_instance(w.add_model([(q, mat)]))
    w.add_point_light((1.0, 1.0, 1.0), 40.0, (0.0, 0.0, 3.0))
    w.camera = dict(eye=(0.0, 0.0, 4.0), target=(0.0, 0.0, 0.0), up=(0.0, 1.0, 0.0), fov=math.radians(50.0), zN=0.1, zF=100.0)
    return w


@pytest.mark.gpu
@pytest.mark.parametrize("draw_type", ["Albedo", "Default"])
def test_bc7_texture_upload_renders_like_its_decoded_twin(draw_type):
    """A scene whose textures go up as BC7 blocks (decoded by decode_bc7_kernel into the tiled layout) renders
    the same bits as the scene with the numpy-decoded RGBA8 twins - and that one equals the oracle."""
    from conftest import default_pc, same_bits
    from oracle import binding as O
    from prosper_amd import capi, structs as S
    w, h = 320, 224
    images = []
    for decoded in (False, True):
        world = _bc7_wall(decoded)
        c = world.camera
        cam, fl = O.camera_uniforms(c["eye"], c["target"], c["up"], c["fov"], c["zN"], c["zF"], w, h)
        pc = default_pc(S, fl, draw_type=S.DrawType[draw_type], max_bounces=2)
        ctx = capi.Context(0)
        try:
            ctx.upload_scene(world)
            ctx.render(pc, cam, w, h)
            images.append(ctx.read_hdr())
        finally:
            ctx.close()
    assert same_bits(images[0], images[1]).all()
    want, _ = O.OracleScene(_bc7_wall(True), brute_force=True).render(pc, cam, w, h)
    assert same_bits(images[1], want).all()
    if draw_type == "Albedo":
        assert len(np.unique(images[0][..., :3].reshape(-1, 3), axis=0)) > 500


@pytest.mark.gpu
def test_bc7_texture_must_be_whole_blocks():
    from prosper_amd import capi
    from prosper_amd.world import Bc7Texture
    world = _bc7_wall(False)
    world.textures[1] = Bc7Texture(world.textures[1].blocks, 18, 8)
    ctx = capi.Context(0)
    try:
        with pytest.raises(capi.ProsperPtError):
            ctx.upload_scene(world)
    finally:
        ctx.close()
