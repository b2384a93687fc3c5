"""Alpha bounds (pt_scene.hpp AlphaMaterial; DESIGN.md "alpha bounds"): most any-hit candidates (rt/scene.rahit:18-39) are
settled from a per-material table of conservative alpha bounds without fetching a texel - and every decision must still be
sampleAlpha's (scene/materials.glsl:121-147).  The oracle knows nothing of the table: it runs the shader text per candidate."""
import numpy as np
import pytest

from conftest import default_pc, same_bits
from prosper_amd import capi, scenes, structs as S


def _camera(oracle, world, w, h):
    c = world.camera
    return oracle.camera_uniforms(c["eye"], c["target"], c["up"], c["fov"], c["zN"], c["zF"], w, h)


def test_alpha_wall_scene_exercises_both_outcomes(oracle):
    """CPU: the test scene lets rays through its MASK / BLEND quads and stops others (otherwise the GPU test below
    would prove nothing)."""
    world = scenes.alpha_wall()
    w, h = 160, 100
    cam, fl = _camera(oracle, world, w, h)
    osc = oracle.OracleScene(world)
    ids, _ = osc.render(default_pc(S, fl, draw_type=S.DrawType["MaterialID"], max_bounces=1), cam, w, h)
    colours = {tuple(c) for c in ids[..., :3].reshape(-1, 3).round(5)}
    assert len(colours) > 25           # the wall behind + most of the 40 alpha materials are seen
    wall = ids[h // 2, w // 2, :3]     # some pixel inside the grid of quads shows the wall through a quad
    assert any((ids[..., :3] == wall).all(axis=-1)[h // 4: 3 * h // 4, w // 4: 3 * w // 4].ravel())


@pytest.mark.gpu
def test_srgb_to_linear_is_monotone_over_every_input(gpu_ctx):
    """The bounds need L = the device's sRGBtoLinear (pow through the contract's exp2 / log2) to be monotone up to
    kAlphaCurveSlack = 4e-6 on the range filtered UNORM8 values can take: measured on EVERY float in [0, 1.001]."""
    last = int(np.float32(1.001).view(np.uint32))
    defect, decreases = gpu_ctx.srgb_monotonicity(0, last)
    assert defect < 1e-6, "sRGBtoLinear is non-monotone by %g (%d decreasing pairs)" % (defect, decreases)
    assert (decreases == 0) == (defect == 0.0)  # the two figures of the kernel agree with each other


@pytest.mark.gpu
@pytest.mark.parametrize("cell", ["default", "0", "1", "2", "3", "5", "off"])
def test_alpha_wall_bit_exact(gpu_ctx, oracle, cell):
    """Every wrap mode x filter x MASK / BLEND x factor x cutoff, odd texture sizes, UVs over several periods: three
    accumulated frames with shadows and bounces through the quads == the oracle, bit for bit, for every cell size of
    the bounds (1, 2, 4, 8, 32 texels) and without them."""
    world = scenes.alpha_wall()
    w, h = 400, 256
    cam, fl = _camera(oracle, world, w, h)
    if cell == "off":
        capi.debug(noAlphaBounds=1)
    elif cell != "default":
        capi.debug(alphaCellShift=int(cell))
    gpu_ctx.upload_scene(world)
    st = gpu_ctx.scene_stats()
    assert st.alphaTriangleCount == 80 and (st.alphaBoundBytes == 0) == (cell == "off")
    osc = oracle.OracleScene(world)
    want = None
    gpu_ctx.reset_counters()
    for frame in (1, 2, 3):
        pc = default_pc(S, fl, frame_index=frame, max_bounces=3, skip_history=(frame == 1))
        gpu_ctx.render(pc, cam, w, h, flags=S.RENDER_COUNT_WORK)
        want, _ = osc.render(pc, cam, w, h, history=want)
    got = gpu_ctx.read_hdr()
    ok = same_bits(got, want).all(axis=2)
    assert ok.all(), "cell %s: %d of %d pixels differ" % (cell, (~ok).sum(), ok.size)
    c = gpu_ctx.counters().as_dict()
    assert c["anyHitCalls"] > 200000
    if cell == "off":
        # every alpha material of the scene has a texture: without the table each candidate fetches its texels
        assert c["anyHitTexelFetches"] == c["anyHitCalls"]
    else:
        assert c["anyHitTexelFetches"] < c["anyHitCalls"]
    # the timed kernels (no counters) and the batched entry point give the same image
    gpu_ctx.render(default_pc(S, fl, max_bounces=3), cam, w, h, frames=3)
    assert same_bits(gpu_ctx.read_hdr(), want).all()
    for name in ("Albedo", "MaterialID"):
        pc = default_pc(S, fl, draw_type=S.DrawType[name], max_bounces=1)
        gpu_ctx.render(pc, cam, w, h)
        dbg, _ = osc.render(pc, cam, w, h)
        assert same_bits(gpu_ctx.read_hdr(), dbg).all(), name


@pytest.mark.gpu
def test_foliage_candidates_settled_without_texels(gpu_ctx, oracle):
    """C4's foliage (leaf-shaped MASK and BLEND quads, 128^2 alpha textures): the bounds settle most candidates - fewer
    than 30 % still fetch texels - the any-hit count itself (a property of the rays, equal with and without the table)
    and the image do not change."""
    world = scenes.sponza_class(lights=True, foliage=True, texture_size=64, sky_size=32, detail=0.25)
    w, h = 320, 180
    cam, fl = _camera(oracle, world, w, h)
    pc = default_pc(S, fl, max_bounces=4, ibl=True)
    images, counters = [], []
    for off in (False, True):
        if off:
            capi.debug(noAlphaBounds=1)
        gpu_ctx.upload_scene(world)
        gpu_ctx.reset_counters()
        gpu_ctx.render(pc, cam, w, h, frames=2, flags=S.RENDER_COUNT_WORK)
        images.append(gpu_ctx.read_hdr())
        counters.append(gpu_ctx.counters().as_dict())
    assert same_bits(images[0], images[1]).all()
    on, off = counters
    assert on["anyHitCalls"] == off["anyHitCalls"] and off["anyHitTexelFetches"] == off["anyHitCalls"]
    assert on["anyHitTexelFetches"] < 0.3 * on["anyHitCalls"], (on["anyHitTexelFetches"], on["anyHitCalls"])
    want = None
    osc = oracle.OracleScene(world)
    for frame in (1, 2):
        want, _ = osc.render(default_pc(S, fl, frame_index=frame, max_bounces=4, ibl=True, skip_history=(frame == 1)), cam,
                             w, h, history=want)
    assert same_bits(images[0], want).all()
