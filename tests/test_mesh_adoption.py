"""Incremental adoption of meshes (prosper_pt_update_meshes): prosper's mesh worker fills the geometry buffers in the
background, WorldData::pollMeshWorker adopts up to ten finished meshes per frame (src/scene/WorldData.cpp:2003-2110), a model's
BLAS is built once ALL its sub-meshes are there (World::buildNextBlas, World.cpp:598-606) and a TLAS instance without a BLAS
is inactive (World.cpp:909-915).  Every frame of such a sequence - frames in flight between the calls - must show what a
fresh prosper_pt_upload_scene of that frame's state shows, bit for bit, and the oracle's image of that state."""
import copy
import ctypes

import numpy as np
import pytest

from conftest import default_pc, same_bits
from prosper_amd import capi, scenes, structs as S
from prosper_amd.world import translate


def _camera(oracle, world, w, h):
    c = world.camera
    return oracle.camera_uniforms(c["eye"], c["target"], c["up"], c["fov"], c["zN"], c["zF"], w, h)


def _oracle_image(oracle, world, cam, fl, w, h, **pc):
    osc = oracle.OracleScene(world)
    want = None
    for f in (1, 2):
        want, _ = osc.render(default_pc(S, fl, frame_index=f, skip_history=(f == 1), **pc), cam, w, h, history=want)
    return want


def test_oracle_skips_model_instances_that_are_still_loading(oracle, cornell_world):
    """The oracle's restatement of the rule, pinned on the CPU: a scene in which a model has a mesh still loading renders
    like the scene without that model's instances (draw instance indices shift, their order does not)."""
    full = cornell_world
    # model 0 has three sub-meshes: with one of them missing, none of the three may show
    missing = full.models[0][1][0]
    streaming = full.with_meshes_loaded([i for i in range(len(full.metadatas)) if i != missing])
    without = copy.copy(full)
    without._frozen = None
    without.model_instances = [mi for mi in full.model_instances if mi[0] != 0]
    w, h = 64, 40
    cam, fl = _camera(oracle, full, w, h)
    a = _oracle_image(oracle, streaming, cam, fl, w, h, max_bounces=3)
    b = _oracle_image(oracle, without, cam, fl, w, h, max_bounces=3)
    c = _oracle_image(oracle, full, cam, fl, w, h, max_bounces=3)
    assert same_bits(a, b).all()
    assert not same_bits(a, c).all()
    assert oracle.OracleScene(streaming).triangle_count == without.triangle_count()


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


def _moved(world, instance, offset):
    w = copy.copy(world)
    w._frozen = None
    w.model_instances = list(world.model_instances)
    model, m = w.model_instances[instance]
    w.model_instances[instance] = (model, translate(offset) @ m)
    return w


@pytest.mark.gpu
def test_meshes_stream_in_over_frames_in_flight(gpu_ctx, oracle):
    """S-sponza-class (31 meshes in 15 models, 43 instances; models of up to six sub-meshes) arrives a few meshes per frame,
    an instance moves in between (its subtree is stale when the next meshes come), frames stay in flight: each frame equals
    a fresh context's render of that frame's state; an early and the last one equal the oracle's."""
    full = scenes.sponza_class(texture_size=32, sky_size=16, detail=0.25)
    meshes = len(full.metadatas)
    # in the order a loader would finish them; the steps cut through models
    order = list(range(meshes))
    steps = [0, 4, 9, 10, 18, 27, meshes]
    w, h = 240, 136
    cam, fl = _camera(oracle, full, w, h)
    pc = default_pc(S, fl, max_bounces=3, ibl=True)
    hip = ctypes.CDLL("libamdhip64.so")
    outs = _device_buffers(hip, len(steps), w * h * 16)
    base = full
    gpu_ctx.upload_scene(base.with_meshes_loaded([]))
    assert gpu_ctx.scene_stats().triangleCount == 0
    states = []
    for k, loaded in enumerate(steps):
        if k == 3:
            base = _moved(base, 2, (0.4, 0.1, -0.3))
            gpu_ctx.update_transforms(base)
        if k == 5:
            base = _moved(base, 7, (-0.2, 0.0, 0.5))
            gpu_ctx.update_transforms(base)
        if k:
            gpu_ctx.update_meshes(base, order[steps[k - 1]:loaded])
        states.append(base.with_meshes_loaded(order[:loaded]))
        gpu_ctx.set_output_buffer(outs[k].value, w * h * 16)
        gpu_ctx.render(pc, cam, w, h, frames=2, flags=S.RENDER_PIPELINED)
    assert hip.hipDeviceSynchronize() == 0
    gpu_ctx.set_output_buffer(0, 0)
    got = [_download(hip, p, h, w) for p in outs]
    for p in outs:
        hip.hipFree(p)
    assert gpu_ctx.scene_stats().triangleCount == full.triangle_count()
    fresh = capi.Context(device=0)
    try:
        counts = []
        for k, state in enumerate(states):
            fresh.upload_scene(state)
            counts.append(fresh.scene_stats().triangleCount)
            fresh.render(pc, cam, w, h, frames=2)
            assert same_bits(got[k], fresh.read_hdr()).all(), "frame %d (%d meshes)" % (k, steps[k])
        assert counts[0] == 0 and all(a <= b for a, b in zip(counts, counts[1:])) and counts[1] < counts[-1]
    finally:
        fresh.close()
    assert not same_bits(got[1], got[-1]).all()
    for k in (2, len(steps) - 1):
        assert same_bits(got[k], _oracle_image(oracle, states[k], cam, fl, w, h, max_bounces=3, ibl=True)).all(), "frame %d vs oracle" % k


@pytest.mark.gpu
def test_alpha_meshes_and_a_new_geometry_buffer_arrive(gpu_ctx, oracle):
    """MASK / BLEND quads (their any-hit records are laid out again), a material that changes in the same frame, and meshes
    that live in a geometry buffer the scene has not seen yet (the loader opens a new 64 MB buffer when a mesh no longer fits)."""
    full = scenes.alpha_wall()
    rng = np.random.default_rng(3)
    extra = []
    for k in range(3):
        p, n, t, uv, idx = scenes.quad((-4.0 + k, -2.2, -0.3 - 0.1 * k), (-3.2 + k, -2.2, -0.3), (-3.2 + k, -1.2, -0.3), (-4.0 + k, -1.2, -0.3 - 0.1 * k))
        mesh = full.add_mesh(p, idx, 1 + k, normals=n, tangents=t, uvs=uv, buffer_index=1 + k // 2)
        extra.append(mesh)
        full.add_instance(full.add_model([(mesh, 1 + k)]))
    meshes = len(full.metadatas)
    first = [i for i in range(meshes) if i not in extra]
    w, h = 240, 150
    cam, fl = _camera(oracle, full, w, h)
    pc = default_pc(S, fl, max_bounces=3)
    order = [first[:13], first[13:30], first[30:] + extra[:1], extra[1:]]
    gpu_ctx.upload_scene(full.with_meshes_loaded([], buffers=1))
    loaded = []
    state = full
    for k, arrivals in enumerate(order):
        if k == 2:
            state = copy.copy(full)
            state._frozen = None
            state.materials = list(full.materials)
            m = copy.copy(state.materials[3])
            m.baseColorFactor = S.Vec4(0.5, 0.9, 0.4, 0.6)
            m.alphaCutoff = 0.35
            state.materials[3] = m
            gpu_ctx.update_materials(state.materials, 0)
        gpu_ctx.update_meshes(state, arrivals)
        loaded += arrivals
        gpu_ctx.render(pc, cam, w, h, frames=2, flags=S.RENDER_PIPELINED)
        got = gpu_ctx.read_hdr()
        now = state.with_meshes_loaded(loaded)
        fresh = capi.Context(device=0)
        try:
            fresh.upload_scene(now)
            fresh.render(pc, cam, w, h, frames=2)
            assert same_bits(got, fresh.read_hdr()).all(), "step %d" % k
            assert gpu_ctx.scene_stats().alphaTriangleCount == fresh.scene_stats().alphaTriangleCount
        finally:
            fresh.close()
    assert same_bits(got, _oracle_image(oracle, state, cam, fl, w, h, max_bounces=3)).all()


@pytest.mark.gpu
def test_what_update_meshes_refuses(gpu_ctx, oracle, cornell_world):
    """A mesh handed over twice, a byte range or an index outside what came with the mesh: refused before anything is
    touched - the scene renders as before."""
    full = cornell_world
    meshes = len(full.metadatas)
    w, h = 96, 64
    cam, fl = _camera(oracle, full, w, h)
    pc = default_pc(S, fl, max_bounces=2)
    gpu_ctx.upload_scene(full.with_meshes_loaded(range(meshes - 1)))
    gpu_ctx.render(pc, cam, w, h)
    before = gpu_ctx.read_hdr()
    with pytest.raises(capi.ProsperPtError) as e:
        gpu_ctx.update_meshes(full, [0])
    assert e.value.code == -1 and "loaded already" in str(e.value)
    last = meshes - 1
    buffer_index, first_word, words = full.mesh_ranges[last]
    f = full.freeze()

    def update(**change):
        u = S.MeshUpdate()
        u.meshIndex, u.metadata, u.info = last, full.metadatas[last], full.mesh_infos[last]
        u.bytes = f["geometry_buffers"][buffer_index].ctypes.data + 4 * first_word
        u.byteOffset, u.byteCount, u.bufferByteSize = 4 * first_word, 4 * words, f["geometry_buffers"][buffer_index].nbytes
        for key, value in change.items():
            setattr(u, key, value)
        return capi.lib().prosper_pt_update_meshes(gpu_ctx._h, ctypes.byref(u), 1)

    assert update(meshIndex=meshes) == -1               # no such slot
    assert update(byteCount=4 * words - 4) == -5        # a stream ends outside the bytes
    assert update(byteOffset=4 * first_word + 4) == -5  # ... or starts before them
    assert update(byteCount=1 << 40) == -5              # past the geometry buffer
    info = copy.copy(full.mesh_infos[last])
    info.vertexCount -= 1
    assert update(info=info) == -5                      # an index names a vertex the mesh does not have
    info = copy.copy(full.mesh_infos[last])
    info.materialIndex = len(full.materials)
    assert update(info=info) == -5
    gpu_ctx.render(pc, cam, w, h)
    assert same_bits(before, gpu_ctx.read_hdr()).all()
    assert update() == 0
    gpu_ctx.finish_mesh_updates()
    gpu_ctx.render(pc, cam, w, h)
    fresh = capi.Context(device=0)
    try:
        fresh.upload_scene(full)
        fresh.render(pc, cam, w, h)
        assert same_bits(gpu_ctx.read_hdr(), fresh.read_hdr()).all()
    finally:
        fresh.close()


@pytest.mark.gpu
def test_frames_go_on_while_the_worker_builds_and_what_changed_meanwhile_follows(gpu_ctx, oracle):
    """prosper_pt_update_meshes returns at once; frames rendered until the worker is done show the scene as it was, the first
    one after it the new meshes - never anything in between.  An instance that moved and a MASK material that changed while
    the worker ran are part of what is switched in."""
    full = scenes.sponza_class(foliage=True, texture_size=32, sky_size=16, detail=0.5)
    meshes = len(full.metadatas)
    first = list(range(0, meshes, 2))
    second = [i for i in range(meshes) if i not in first]
    w, h = 200, 120
    cam, fl = _camera(oracle, full, w, h)
    pc = default_pc(S, fl, frame_index=7, max_bounces=2, ibl=True, skip_history=True)  # every render: the same one frame

    def fresh_image(world):
        fresh = capi.Context(device=0)
        try:
            fresh.upload_scene(world)
            fresh.render(pc, cam, w, h)
            return fresh.read_hdr()
        finally:
            fresh.close()

    before = fresh_image(full.with_meshes_loaded(first))
    after = fresh_image(full)
    assert not same_bits(before, after).all()
    gpu_ctx.upload_scene(full.with_meshes_loaded(first))
    gpu_ctx.render(pc, cam, w, h)
    assert same_bits(gpu_ctx.read_hdr(), before).all()
    gpu_ctx.update_meshes(full, second, wait=False)
    state = gpu_ctx.hierarchy_state()
    assert state.meshUpdates == 1 and state.geometryBuildRunning == 1
    seen_before = seen_after = 0
    for _ in range(4000):
        gpu_ctx.render(pc, cam, w, h, flags=S.RENDER_PIPELINED)
        img = gpu_ctx.read_hdr()
        if same_bits(img, before).all():
            assert not seen_after, "the scene went back to the old geometry"
            seen_before += 1
        else:
            assert same_bits(img, after).all(), "a frame shows neither the old nor the new scene"
            seen_after += 1
            if seen_after == 3:
                break
    assert seen_after == 3
    state = gpu_ctx.hierarchy_state()
    assert state.geometryInstalls == 1 and state.geometryBuildRunning == 0

    # again, with an instance moved and a material changed while the worker runs
    gpu_ctx.upload_scene(full.with_meshes_loaded(first))
    gpu_ctx.render(pc, cam, w, h, flags=S.RENDER_PIPELINED)
    gpu_ctx.update_meshes(full, second, wait=False)
    moved = _moved(full, 3, (0.3, 0.2, -0.4))
    moved.materials = list(full.materials)
    masked = [i for i, m in enumerate(full.materials) if m.alphaMode != S.ALPHA_MODE_OPAQUE]
    assert masked
    for i in masked[:2]:
        m = copy.copy(moved.materials[i])
        m.alphaCutoff = 0.8
        m.baseColorFactor = S.Vec4(0.9, 0.4, 0.3, 0.7)
        moved.materials[i] = m
    gpu_ctx.update_transforms(moved)
    gpu_ctx.update_materials(moved.materials, 0)
    gpu_ctx.render(pc, cam, w, h, flags=S.RENDER_PIPELINED)  # (old or new geometry, with the moved instance either way)
    gpu_ctx.finish_mesh_updates()
    gpu_ctx.render(pc, cam, w, h, flags=S.RENDER_PIPELINED)
    assert same_bits(gpu_ctx.read_hdr(), fresh_image(moved)).all()
    # meshes that arrive while a build is under way are taken up by the next one
    gpu_ctx.upload_scene(full.with_meshes_loaded([]))
    for k in range(0, meshes, 3):
        gpu_ctx.update_meshes(full, list(range(k, min(k + 3, meshes))), wait=False)
        gpu_ctx.render(pc, cam, w, h, flags=S.RENDER_PIPELINED)
    gpu_ctx.finish_mesh_updates()
    gpu_ctx.render(pc, cam, w, h, flags=S.RENDER_PIPELINED)
    assert same_bits(gpu_ctx.read_hdr(), after).all()
    state = gpu_ctx.hierarchy_state()
    assert state.meshUpdates == (meshes + 2) // 3 and 1 <= state.geometryInstalls <= state.meshUpdates


@pytest.mark.gpu
def test_flight_helmet_loads_the_way_prosper_loads_it(gpu_ctx, oracle):
    """The reference's bundled asset through prosper's whole loading sequence (WorldData::handleDeferredLoading,
    WorldData.cpp:588-647): first the meshes, a few per frame, under placeholder materials; once all meshes are there the
    images, a few per frame, each material switching from its placeholder when its images have arrived - frames in flight
    throughout, nothing waited for until the end.  The last frame is the oracle's image of the loaded asset; a frame in the
    middle of each phase equals a fresh upload of that state."""
    from prosper_amd import flight_helmet
    from test_adoption import streamed_state
    full = flight_helmet.load_fixture(texture_size=256)
    meshes, images = len(full.metadatas), len(full.textures) - 1
    w, h = 320, 200
    cam, fl = _camera(oracle, full, w, h)
    pc = default_pc(S, fl, max_bounces=3, ibl=True)
    placeholders = streamed_state(full, 0)
    gpu_ctx.upload_scene(placeholders.with_meshes_loaded([]))
    checks = []
    loaded = 0
    while loaded < meshes:
        n = min(2, meshes - loaded)
        gpu_ctx.update_meshes(placeholders, list(range(loaded, loaded + n)), wait=False)
        loaded += n
        gpu_ctx.render(pc, cam, w, h, frames=1, flags=S.RENDER_PIPELINED)
        if loaded == 2 * ((meshes // 2 + 1) // 2):
            gpu_ctx.finish_mesh_updates()
            gpu_ctx.render(pc, cam, w, h, frames=2, flags=S.RENDER_PIPELINED)
            checks.append((gpu_ctx.read_hdr(), placeholders.with_meshes_loaded(range(loaded))))
    gpu_ctx.finish_mesh_updates()  # "Meshes should have been loaded before textures" (WorldData.cpp:601-604)
    arrived = 0
    while arrived < images:
        n = min(4, images - arrived)
        state = streamed_state(full, arrived + n)
        gpu_ctx.update_textures(full.textures[arrived + 1:arrived + 1 + n], arrived + 1)
        gpu_ctx.update_materials(state.materials, 0)
        arrived += n
        gpu_ctx.render(pc, cam, w, h, frames=1, flags=S.RENDER_PIPELINED)
        if arrived == 8:
            gpu_ctx.render(pc, cam, w, h, frames=2, flags=S.RENDER_PIPELINED)
            checks.append((gpu_ctx.read_hdr(), state))
    gpu_ctx.render(pc, cam, w, h, frames=2, flags=S.RENDER_PIPELINED)
    last = gpu_ctx.read_hdr()
    assert len(checks) == 2
    fresh = capi.Context(device=0)
    try:
        for k, (got, state) in enumerate(checks):
            fresh.upload_scene(state)
            fresh.render(pc, cam, w, h, frames=2)
            assert same_bits(got, fresh.read_hdr()).all(), "phase %d" % k
    finally:
        fresh.close()
    assert same_bits(last, _oracle_image(oracle, full, cam, fl, w, h, max_bounces=3, ibl=True)).all()


@pytest.mark.gpu
def test_a_failed_background_build_leaves_the_scene_as_it_was(gpu_ctx, oracle, cornell_world):
    """The worker gives up half way (debug option failNextUpdate): the call that would have switched to its result reports
    it, the scene renders as before, what the worker had allocated is given back, and the meshes are taken up by the next
    build - prosper_pt_finish_mesh_updates starts it."""
    full = cornell_world
    meshes = len(full.metadatas)
    w, h = 96, 64
    cam, fl = _camera(oracle, full, w, h)
    pc = default_pc(S, fl, max_bounces=2)
    gpu_ctx.upload_scene(full.with_meshes_loaded(range(meshes - 2)))
    gpu_ctx.render(pc, cam, w, h)
    before = gpu_ctx.read_hdr()
    bytes_before = gpu_ctx.scene_stats().deviceBytes
    gpu_ctx.set_debug(failNextUpdate=1)
    gpu_ctx.update_meshes(full, [meshes - 2, meshes - 1], wait=False)
    with pytest.raises(capi.ProsperPtError) as e:
        for _ in range(2000):
            gpu_ctx.render(pc, cam, w, h)
    assert e.value.code == -6 and "background build" in str(e.value)
    gpu_ctx.set_debug(failNextUpdate=None)  # (cleared, not left at 0: the context's own options override the process-wide ones)
    gpu_ctx.render(pc, cam, w, h)
    assert same_bits(before, gpu_ctx.read_hdr()).all()
    assert gpu_ctx.scene_stats().deviceBytes == bytes_before
    assert gpu_ctx.hierarchy_state().geometryBuildRunning == 1  # the meshes still wait
    gpu_ctx.finish_mesh_updates()
    gpu_ctx.render(pc, cam, w, h)
    fresh = capi.Context(device=0)
    try:
        fresh.upload_scene(full)
        fresh.render(pc, cam, w, h)
        assert same_bits(gpu_ctx.read_hdr(), fresh.read_hdr()).all()
    finally:
        fresh.close()


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [1, 2, 3])
def test_random_interleavings_of_everything_that_can_happen_while_a_scene_loads(gpu_ctx, oracle, seed):
    """Meshes arriving (waited for or not), instances moving, MASK / BLEND and opaque materials changing, frames in flight,
    in a random order: at every checkpoint - after prosper_pt_finish_mesh_updates - the frame equals a fresh upload of the
    state the calls have described so far."""
    rng = np.random.default_rng(seed)
    full = scenes.sponza_class(foliage=True, texture_size=32, sky_size=16, detail=0.25)
    meshes = len(full.metadatas)
    remaining = [int(i) for i in rng.permutation(meshes)]
    loaded = []
    state = full
    w, h = 200, 120
    cam, fl = _camera(oracle, full, w, h)
    pc = default_pc(S, fl, max_bounces=2, ibl=True)
    gpu_ctx.upload_scene(state.with_meshes_loaded(loaded))
    fresh = capi.Context(device=0)
    try:
        checkpoints = 0
        for step in range(48):
            op = rng.integers(0, 6) if remaining else rng.integers(1, 6)
            if op == 0:
                n = int(rng.integers(1, 5))
                arrivals, remaining = remaining[:n], remaining[n:]
                gpu_ctx.update_meshes(state, arrivals, wait=bool(rng.random() < 0.3))
                loaded += arrivals
            elif op == 1:
                state = _moved(state, int(rng.integers(0, len(state.model_instances))), tuple(rng.normal(0.0, 0.3, 3)))
                gpu_ctx.update_transforms(state)
            elif op == 2:
                state = copy.copy(state)
                state._frozen = None
                state.materials = list(state.materials)
                i = int(rng.integers(1, len(state.materials)))
                m = copy.copy(state.materials[i])
                m.roughnessFactor = float(rng.uniform(0.1, 1.0))
                m.alphaCutoff = float(rng.uniform(0.2, 0.8))
                m.baseColorFactor = S.Vec4(float(rng.uniform(0.3, 1.0)), 0.8, 0.7, float(rng.uniform(0.3, 1.0)) if m.alphaMode else 1.0)
                state.materials[i] = m
                gpu_ctx.update_materials(state.materials, 0)
            elif op in (3, 4):
                gpu_ctx.render(pc, cam, w, h, frames=int(rng.integers(1, 3)), flags=S.RENDER_PIPELINED)
            else:
                gpu_ctx.finish_mesh_updates()
                gpu_ctx.render(pc, cam, w, h, frames=2, flags=S.RENDER_PIPELINED)
                got = gpu_ctx.read_hdr()
                fresh.upload_scene(state.with_meshes_loaded(loaded))
                fresh.render(pc, cam, w, h, frames=2)
                assert same_bits(got, fresh.read_hdr()).all(), "seed %d, step %d, %d meshes" % (seed, step, len(loaded))
                checkpoints += 1
        if remaining:
            gpu_ctx.update_meshes(state, remaining, wait=False)
            loaded += remaining
        gpu_ctx.finish_mesh_updates()
        gpu_ctx.render(pc, cam, w, h, frames=2, flags=S.RENDER_PIPELINED)
        fresh.upload_scene(state)
        fresh.render(pc, cam, w, h, frames=2)
        assert same_bits(gpu_ctx.read_hdr(), fresh.read_hdr()).all()
        assert checkpoints >= 2
    finally:
        fresh.close()
