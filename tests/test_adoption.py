"""Incremental adoption (prosper_pt_update_textures / prosper_pt_update_materials): prosper loads a scene in the background
and adopts what has arrived a few items per frame - images get their slots (src/scene/WorldData.cpp:2182-2206), a material
switches from its placeholder (the default material with the real alpha mode, WorldData.cpp:817-826) to the real one once its
three images are there, in order (WorldData.cpp:2208-2239), and the next frame's material buffer is rewritten (:568-586).
Every frame of such a sequence - three in flight, nothing synchronising the device - must show what a fresh
prosper_pt_upload_scene of that frame's state shows, bit for bit."""
import copy
import ctypes

import numpy as np
import pytest

from conftest import default_pc, same_bits
from prosper_amd import capi, scenes, structs as S
from prosper_amd.world import World

pytestmark = pytest.mark.gpu


def _camera(oracle, world, w, h):
    c = world.camera
    return oracle.camera_uniforms(c["eye"], c["target"], c["up"], c["fov"], c["zN"], c["zF"], w, h)


def _placeholder(material):
    """What WorldData::loadMaterials puts in the table until a material's images are there: the default material with the
    real material's alpha mode."""
    m = World._material_struct()
    m.alphaMode = material.alphaMode
    return m


def _texture_indices(m):
    return [t & 0xFFFFFF for t in (m.baseColorTextureSampler, m.metallicRoughnessTextureSampler, m.normalTextureSampler)]


def streamed_state(full, loaded_images):
    """`full` with only its first `loaded_images` images (texture slots 1 .. loaded_images) adopted: the other slots hold
    1 x 1 white texels, and materials are adopted in order while their images are there (WorldData::updateMaterials)."""
    w = copy.copy(full)
    w._frozen = None
    w.textures = [t if i <= loaded_images else np.full((1, 1, 4), 255, np.uint8) for i, t in enumerate(full.textures)]
    w.materials = [full.materials[0]]
    adopting = True
    for m in full.materials[1:]:
        adopting = adopting and all(t <= loaded_images for t in _texture_indices(m))
        w.materials.append(m if adopting else _placeholder(m))
    return w


def _device_buffers(hip, n, nbytes):
    out = []
    for _ in range(n):
        ptr = ctypes.c_void_p()
        assert hip.hipMalloc(ctypes.byref(ptr), ctypes.c_size_t(nbytes)) == 0
        out.append(ptr)
    return out


def _download(hip, ptr, h, w):
    img = np.zeros((h, w, 4), np.float32)
    assert hip.hipMemcpy(ctypes.c_void_p(img.ctypes.data), ptr, ctypes.c_size_t(img.nbytes), 2) == 0  # device to host
    return img


@pytest.mark.parametrize("texture_size,fifth_stream", [(None, False), (256, False), (None, True)])
def test_flight_helmet_streams_in_over_frames_in_flight(gpu_ctx, oracle, texture_size, fifth_stream):
    """The reference's bundled asset: fifteen images and six materials (one of them BLEND) adopted over five frames in
    flight; every frame equals a fresh context's render of that frame's state.
    fifth_stream: one more stream alive in the process changes which streams share a hardware queue, and with it the order
    in which the frames' streams get to run - this is how a missing dependency between two flushes of the material tables
    showed (the frame after the one that rewrote the any-hit records of the BLEND lenses overtook that rewrite)."""
    from prosper_amd import flight_helmet
    extra = None
    if fifth_stream:
        hip0 = ctypes.CDLL("libamdhip64.so")
        extra, ev = ctypes.c_void_p(), ctypes.c_void_p()
        assert hip0.hipStreamCreateWithFlags(ctypes.byref(extra), 1) == 0 and hip0.hipEventCreate(ctypes.byref(ev)) == 0
        assert hip0.hipEventRecord(ev, extra) == 0 and hip0.hipStreamSynchronize(extra) == 0  # (its queue exists from now on)
        hip0.hipEventDestroy(ev)
    full = flight_helmet.load_fixture(texture_size=texture_size)
    images = len(full.textures) - 1
    steps = [0, 3, 6, 9, 12, images]
    w, h = 320, 200
    cam, fl = _camera(oracle, full, w, h)
    pc = default_pc(S, fl, max_bounces=3, ibl=True)
    hip = ctypes.CDLL("libamdhip64.so")
    outs = _device_buffers(hip, len(steps), w * h * 16)
    gpu_ctx.upload_scene(streamed_state(full, 0))
    states = []
    for k, loaded in enumerate(steps):
        state = streamed_state(full, loaded)
        states.append(state)
        if k:
            first = steps[k - 1] + 1
            gpu_ctx.update_textures(full.textures[first:loaded + 1], first)  # the new images
            gpu_ctx.update_materials(state.materials, 0)                     # the whole table, as prosper rewrites it
        gpu_ctx.set_output_buffer(outs[k].value, w * h * 16)
        gpu_ctx.render(pc, cam, w, h, frames=2, flags=S.RENDER_PIPELINED)
    assert hip.hipDeviceSynchronize() == 0
    gpu_ctx.set_output_buffer(0, 0)
    got = [_download(hip, p, h, w) for p in outs]
    for p in outs:
        hip.hipFree(p)
    assert not same_bits(got[0], got[-1]).all()  # the textures do show
    fresh = capi.Context(device=0)
    try:
        for k, state in enumerate(states):
            fresh.upload_scene(state)
            fresh.render(pc, cam, w, h, frames=2)
            same = same_bits(got[k], fresh.read_hdr()).all(axis=2)
            rows, cols = np.nonzero(~same)
            assert same.all(), "frame %d (%d images): %d pixels differ, rows %d-%d, columns %d-%d" % (
                k, steps[k], rows.size, rows.min(), rows.max(), cols.min(), cols.max())
    finally:
        fresh.close()
    # and the last one is the oracle's image of the whole asset
    osc = oracle.OracleScene(full)
    want = None
    for f in (1, 2):
        want, _ = osc.render(default_pc(S, fl, frame_index=f, max_bounces=3, ibl=True, skip_history=(f == 1)), cam, w, h, history=want)
    assert same_bits(got[-1], want).all()
    if extra is not None:
        hip.hipStreamDestroy(extra)


def test_alpha_textures_and_bc7_adopted_between_frames_in_flight(gpu_ctx, oracle):
    """MASK / BLEND materials whose alpha textures arrive late (the any-hit records carry a copy of their material: they are
    rewritten behind the frames in flight), a BC7 image among the arrivals, a material table that changes factor and cutoff
    without any new image: each frame equals a fresh upload of its state, the last one the oracle."""
    full = scenes.alpha_wall()
    # one opaque, fully textured material more, so that a pack is built by an update
    rng = np.random.default_rng(5)
    tex = [full.add_texture(rng.integers(0, 256, size=(32, 32, 4), dtype=np.uint8)) for _ in range(3)]
    mat = full.add_material(base_color=(1.0, 1.0, 1.0, 1.0), metallic=0.5, roughness=0.5, base_tex=(tex[0], 0), mr_tex=(tex[1], 0), normal_tex=(tex[2], 0))
    p, n, t, uv, idx = scenes.quad((-4.0, -2.2, -0.3), (-1.0, -2.2, -0.3), (-1.0, -1.0, -0.3), (-4.0, -1.0, -0.3))
    full.add_instance(full.add_model([(full.add_mesh(p, idx, mat, normals=n, tangents=t, uvs=uv), mat)]))
    images = len(full.textures) - 1
    w, h = 320, 200
    cam, fl = _camera(oracle, full, w, h)
    pc = default_pc(S, fl, max_bounces=3)
    hip = ctypes.CDLL("libamdhip64.so")
    steps = [0, 7, 20, 33, images, images]
    outs = _device_buffers(hip, len(steps), w * h * 16)
    gpu_ctx.upload_scene(streamed_state(full, 0))
    assert gpu_ctx.scene_stats().alphaTriangleCount == 80
    states = []
    for k, loaded in enumerate(steps):
        state = streamed_state(full, loaded)
        if k == len(steps) - 1:
            # no new image: two materials change by themselves (a UI slider, say)
            state.materials = list(state.materials)
            for i in (3, 8):
                m = copy.copy(state.materials[i])
                m.baseColorFactor = S.Vec4(0.5, 0.9, 0.4, 0.6)
                m.alphaCutoff = 0.35
                state.materials[i] = m
            full = state
        states.append(state)
        if k:
            first = steps[k - 1] + 1
            if loaded >= first:
                gpu_ctx.update_textures(state.textures[first:loaded + 1], first)
            gpu_ctx.update_materials(state.materials, 0)
        gpu_ctx.set_output_buffer(outs[k].value, w * h * 16)
        gpu_ctx.render(pc, cam, w, h, frames=2, flags=S.RENDER_PIPELINED)
    assert hip.hipDeviceSynchronize() == 0
    gpu_ctx.set_output_buffer(0, 0)
    got = [_download(hip, p, h, w) for p in outs]
    for p in outs:
        hip.hipFree(p)
    fresh = capi.Context(device=0)
    try:
        for k, state in enumerate(states):
            fresh.upload_scene(state)
            fresh.render(pc, cam, w, h, frames=2)
            assert same_bits(got[k], fresh.read_hdr()).all(), "frame %d (%d images)" % (k, steps[k])
    finally:
        fresh.close()
    osc = oracle.OracleScene(states[-1])
    want = None
    for f in (1, 2):
        want, _ = osc.render(default_pc(S, fl, frame_index=f, max_bounces=3, skip_history=(f == 1)), cam, w, h, history=want)
    assert same_bits(got[-1], want).all()
    # what the call refuses: a changed alpha mode (it decides the geometry's opaque flag), ranges past the tables
    bad = copy.copy(states[-1].materials[2])
    bad.alphaMode = S.ALPHA_MODE_OPAQUE if bad.alphaMode != S.ALPHA_MODE_OPAQUE else S.ALPHA_MODE_MASK
    with pytest.raises(capi.ProsperPtError) as e:
        gpu_ctx.update_materials([bad], 2)
    assert e.value.code == -6
    with pytest.raises(capi.ProsperPtError):
        gpu_ctx.update_materials([states[-1].materials[1]], len(states[-1].materials))
    with pytest.raises(capi.ProsperPtError):
        gpu_ctx.update_textures([states[-1].textures[1]], len(states[-1].textures))
    # a BC7 arrival: the blocks are decoded on the GPU like at upload
    from prosper_amd.world import Bc7Texture
    from test_bc7 import random_blocks
    blocks = np.concatenate([random_blocks(m, 2, 900 + m) for m in range(8)])  # 16 blocks: 16 x 16 texels, every mode
    arrival = Bc7Texture(blocks, 16, 16)
    gpu_ctx.update_textures([arrival], tex[0])
    gpu_ctx.render(pc, cam, w, h, frames=2)
    state = copy.copy(states[-1])
    state._frozen = None
    state.textures = list(state.textures)
    state.textures[tex[0]] = arrival
    fresh = capi.Context(device=0)
    try:
        fresh.upload_scene(state)
        fresh.render(pc, cam, w, h, frames=2)
        assert same_bits(gpu_ctx.read_hdr(), fresh.read_hdr()).all()
    finally:
        fresh.close()


def test_an_unchanged_material_table_costs_nothing(gpu_ctx, oracle, cornell_world):
    """prosper calls uploadMaterialDatas every frame (App.cpp:526-529): with nothing new it must be a no-op here too - no new
    version, no launch."""
    gpu_ctx.upload_scene(cornell_world)
    w, h = 96, 64
    cam, fl = _camera(oracle, cornell_world, w, h)
    pc = default_pc(S, fl, max_bounces=2)
    gpu_ctx.render(pc, cam, w, h)
    before = gpu_ctx.read_hdr()
    bytes_before = gpu_ctx.scene_stats().deviceBytes
    for _ in range(3):
        gpu_ctx.update_materials(cornell_world.freeze()["materials"][:], 0)
        gpu_ctx.render(pc, cam, w, h)
    assert same_bits(before, gpu_ctx.read_hdr()).all()
    assert gpu_ctx.scene_stats().deviceBytes == bytes_before


def test_a_material_slider_does_not_grow_the_scene(gpu_ctx, oracle):
    """A factor dragged for many frames (RtReference's UI keeps rewriting the table): texture packs and alpha bounds are
    rebuilt only for what they depend on - a material's textures and samplers; the base-colour texture, its sampler and
    baseColorFactor.a - so roughness, metallic, colour and cutoff changes allocate nothing, and every frame still equals a
    fresh upload."""
    full = scenes.alpha_wall()
    w, h = 160, 100
    cam, fl = _camera(oracle, full, w, h)
    pc = default_pc(S, fl, max_bounces=2)
    gpu_ctx.upload_scene(full)
    gpu_ctx.render(pc, cam, w, h)
    grown = None
    state = full
    for k in range(12):
        state = copy.copy(state)
        state._frozen = None
        state.materials = list(state.materials)
        for i in (1, 4, 9):
            m = copy.copy(state.materials[i])
            m.roughnessFactor = 0.2 + 0.05 * k
            m.metallicFactor = 0.1 * (k % 3)
            m.alphaCutoff = 0.1 + 0.07 * (k % 5)
            m.baseColorFactor = S.Vec4(0.3 + 0.05 * k, 0.5, 0.7, m.baseColorFactor.w)
            state.materials[i] = m
        gpu_ctx.update_materials(state.materials, 0)
        gpu_ctx.render(pc, cam, w, h, flags=S.RENDER_PIPELINED)
        size = gpu_ctx.scene_stats().deviceBytes
        if k == 3:
            grown = size  # (the first three updates allocate the three table versions and their pinned staging)
        if k > 3:
            assert size == grown, "update %d grew the scene from %d to %d bytes" % (k, grown, size)
    got = gpu_ctx.read_hdr()
    fresh = capi.Context(device=0)
    try:
        fresh.upload_scene(state)
        fresh.render(pc, cam, w, h)
        assert same_bits(got, fresh.read_hdr()).all()
    finally:
        fresh.close()


def test_a_texture_of_a_packed_material_replaced_after_a_full_upload(gpu_ctx, oracle):
    """After a FULL upload the textures only packed materials sample keep no texels of their own (the pack holds them): one
    of a material's three images replaced alone is refused with a message that says so; all three in one call rebuild the
    pack, and the frame equals a fresh upload."""
    world = scenes.sponza_class(texture_size=64, sky_size=32, detail=0.25)
    w, h = 240, 136
    cam, fl = _camera(oracle, world, w, h)
    pc = default_pc(S, fl, max_bounces=3, ibl=True)
    gpu_ctx.upload_scene(world)
    assert gpu_ctx.scene_stats().variantFlags & S.VARIANT_TEXTURE_PACKS
    m = next(m for m in world.materials[1:] if all(_texture_indices(m)))
    tb, tm, tn = _texture_indices(m)
    assert tm == tb + 1 and tn == tb + 2  # (scenes.sponza_class adds a material's three images in a row)
    rng = np.random.default_rng(11)
    fresh_texels = [rng.integers(0, 256, size=np.asarray(world.textures[t]).shape, dtype=np.uint8) for t in (tb, tm, tn)]
    with pytest.raises(capi.ProsperPtError) as e:
        gpu_ctx.update_textures(fresh_texels[:1], tb)
    assert e.value.code == -6 and "update it in the same call" in str(e.value)
    # the whole table rewritten (as prosper does every time) with a factor changed: entries that keep their textures and
    # samplers keep their packs and need no texels
    world = copy.copy(world)
    world._frozen = None
    world.materials = list(world.materials)
    slid = copy.copy(world.materials[world.materials.index(m)])
    slid.roughnessFactor = 0.35
    world.materials[world.materials.index(m)] = slid
    gpu_ctx.update_materials(world.materials, 0)
    gpu_ctx.update_textures(fresh_texels, tb)
    gpu_ctx.render(pc, cam, w, h, frames=2, flags=S.RENDER_PIPELINED)
    got = gpu_ctx.read_hdr()
    state = copy.copy(world)
    state._frozen = None
    state.textures = list(world.textures)
    for t, texels in zip((tb, tm, tn), fresh_texels):
        state.textures[t] = texels
    fresh = capi.Context(device=0)
    try:
        fresh.upload_scene(state)
        fresh.render(pc, cam, w, h, frames=2)
        assert same_bits(got, fresh.read_hdr()).all()
    finally:
        fresh.close()
