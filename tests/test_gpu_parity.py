"""HIP path vs the CPU oracle on the same seeded inputs (bit-exact: DESIGN.md "Arithmetic contract").

All calls go through the C-ABI (libprosper_pt.so); the oracle is only the checker.
"""
import numpy as np
import pytest

from conftest import default_pc, same_bits
from prosper_amd import structs as S

pytestmark = pytest.mark.gpu


def _camera(oracle, world, w, h):
    c = world.camera
    return oracle.camera_uniforms(c["eye"], c["target"], c["up"], c["fov"], c["zN"], c["zF"], w, h)


def _rand_inputs(fn, n, rng):
    in_stride, _ = FN_SHAPES[fn]
    return rng.standard_normal((n, in_stride)).astype(np.float32)


FN_SHAPES = {0: (1, 2), 1: (2, 1), 2: (1, 1), 3: (3, 3), 4: (1, 4), 5: (3, 9), 6: (5, 3), 7: (6, 3), 8: (7, 1),
             9: (14, 3), 10: (6, 3), 11: (10, 7), 12: (14, 7), 13: (17, 4), 14: (1, 2), 15: (3, 4)}


def _unit(v):
    return v / np.linalg.norm(v, axis=1, keepdims=True)


def _fn_inputs(fn, n, rng):
    """Inputs in the domain each function sees on the path."""
    u = lambda *shape: rng.random(shape, dtype=np.float32)
    g = lambda *shape: rng.standard_normal(shape).astype(np.float32)
    if fn == 0:   # sincos: angles in [0, 2pi] plus a wider range
        return np.concatenate([u(n // 2, 1) * 6.2831855, g(n - n // 2, 1) * 50.0]).astype(np.float32)
    if fn == 1:   # pow
        return np.stack([u(n) * 4.0, g(n) * 3.0], axis=1).astype(np.float32)
    if fn == 2:   # srgb: every 8-bit code plus random
        codes = (np.arange(256, dtype=np.float32) / np.float32(255.0))[:, None]
        return np.concatenate([codes, u(n, 1)]).astype(np.float32)
    if fn == 3:
        return g(n, 3) * np.float32(10.0)
    if fn == 4:   # snorm10 bit patterns
        bits = rng.integers(0, 2**32, size=(n, 1), dtype=np.uint64).astype(np.uint32)
        return bits.view(np.float32)
    if fn == 5:
        return _unit(g(n, 3)).astype(np.float32)
    if fn == 6:
        return np.concatenate([_unit(g(n, 3)), u(n, 2)], axis=1).astype(np.float32)
    if fn == 7:
        ve = _unit(g(n, 3))
        ve[:, 2] = np.abs(ve[:, 2])
        return np.concatenate([ve, (u(n, 1) * 0.99 + 0.0025), u(n, 2)], axis=1).astype(np.float32)
    if fn == 8:
        ve, le = _unit(g(n, 3)), _unit(g(n, 3))
        ve[:, 2], le[:, 2] = np.abs(ve[:, 2]), np.abs(le[:, 2])
        return np.concatenate([ve, le, (u(n, 1) * 0.99 + 0.0025)], axis=1).astype(np.float32)
    if fn == 9:
        nrm = _unit(g(n, 3))
        l = _unit(nrm + 0.8 * g(n, 3))
        v = _unit(nrm + 0.8 * g(n, 3))
        return np.concatenate([l, nrm, v, u(n, 3), u(n, 1) * 0.95 + 0.05, u(n, 1)], axis=1).astype(np.float32)
    if fn == 10:
        p = g(n, 3) * np.float32(5.0)
        p[: n // 4] *= np.float32(0.004)  # exercise the |p| < 1/32 branch
        return np.concatenate([p, _unit(g(n, 3))], axis=1).astype(np.float32)
    if fn == 11:
        return np.concatenate([g(n, 3) * 4, u(n, 3) * 3, u(n, 1) * 20 + 0.5, g(n, 3) * 4], axis=1).astype(np.float32)
    if fn == 12:
        return np.concatenate([g(n, 3) * 4, g(n, 1), u(n, 3) * 3, u(n, 1) * 8, _unit(g(n, 3)), g(n, 3) * 4],
                              axis=1).astype(np.float32)
    if fn == 13:
        o, tgt = g(n, 3) * 3, g(n, 3)
        d = _unit(tgt - o)
        v0, v1, v2 = tgt + g(n, 3), tgt + g(n, 3), tgt + g(n, 3)
        return np.concatenate([o, d, v0, v1, v2, np.zeros((n, 1)), np.full((n, 1), np.inf)], axis=1).astype(np.float32)
    if fn == 14:
        x = g(n, 1) * np.float32(100.0)
        x[: n // 8] *= np.float32(1e-6)
        x[n // 8: n // 4] *= np.float32(1e4)
        return x.astype(np.float32)
    if fn == 15:
        return rng.integers(0, 4096, size=(n, 3), dtype=np.uint64).astype(np.uint32).view(np.float32)
    raise AssertionError(fn)


@pytest.mark.parametrize("fn", sorted(FN_SHAPES))
def test_device_functions_match_oracle_bitwise(gpu_ctx, oracle, fn):
    """Every device function of pt_device.hpp agrees with the oracle's restatement bit for bit."""
    rng = np.random.default_rng(1234 + fn)
    x = _fn_inputs(fn, 20000, rng)
    in_stride, out_stride = FN_SHAPES[fn]
    want = oracle.eval_fn(fn, x)
    got = gpu_ctx.eval_device_fn(fn, x, in_stride, out_stride)
    ok = same_bits(got, want)
    bad = np.argwhere(~ok)
    assert bad.size == 0, "fn %d: %d mismatches, first at %s: gpu=%r oracle=%r in=%r" % (
        fn, len(bad), bad[0], got[bad[0][0]], want[bad[0][0]], x[bad[0][0]])


DRAW_TYPES = ["PrimitiveID", "MeshID", "MaterialID", "Position", "ShadingNormal", "TexCoord0", "Albedo", "Roughness",
              "Metallic"]


@pytest.mark.parametrize("draw_type", DRAW_TYPES)
def test_cornell_debug_draw_types_bit_exact(gpu_ctx, oracle, cornell_world, draw_type):
    """C1-size DrawType images: traversal + fetch + decode + material, RNG-free after the jitter."""
    w = h = 256
    cam, fl = _camera(oracle, cornell_world, w, h)
    pc = default_pc(S, fl, draw_type=S.DrawType[draw_type], max_bounces=1)
    gpu_ctx.upload_scene(cornell_world)
    gpu_ctx.render(pc, cam, w, h)
    got = gpu_ctx.read_hdr()
    osc = oracle.OracleScene(cornell_world, brute_force=True)
    want, _ = osc.render(pc, cam, w, h)
    ok = same_bits(got, want).all(axis=2)
    assert ok.all(), "%s: %d of %d pixels differ" % (draw_type, (~ok).sum(), ok.size)


@pytest.mark.parametrize("max_bounces,ibl,dof", [(1, False, False), (4, False, False), (6, True, True)])
def test_cornell_radiance_bit_exact_over_accumulated_frames(gpu_ctx, oracle, cornell_world, max_bounces, ibl, dof):
    """C1 (256x256, 1 bounce) and deeper variants: four accumulated frames, every pixel identical."""
    w = h = 256
    cam, fl = _camera(oracle, cornell_world, w, h)
    gpu_ctx.upload_scene(cornell_world)
    osc = oracle.OracleScene(cornell_world, brute_force=True)
    want = None
    for frame in range(1, 5):
        pc = default_pc(S, fl, frame_index=frame, max_bounces=max_bounces, ibl=ibl, dof=dof,
                        skip_history=(frame == 1), roulette=2)
        pc.apertureDiameter = 0.05 if dof else 1e-5
        pc.focusDistance = 3.0
        gpu_ctx.render(pc, cam, w, h)
        want, _ = osc.render(pc, cam, w, h, history=want)
    got = gpu_ctx.read_hdr()
    ok = same_bits(got, want).all(axis=2)
    assert ok.all(), "%d of %d pixels differ; max abs diff %g" % (
        (~ok).sum(), ok.size, np.nanmax(np.abs(got - want)))
    assert np.isfinite(got).all()
    assert (got[..., 3] == 4.0).all()


def test_render_frames_equals_repeated_render(gpu_ctx, oracle, cornell_world):
    """prosper_pt_render_frames(n) == n x prosper_pt_render (history kept in registers vs HBM)."""
    w, h = 320, 200
    cam, fl = _camera(oracle, cornell_world, w, h)
    gpu_ctx.upload_scene(cornell_world)
    for frame in range(1, 6):
        pc = default_pc(S, fl, frame_index=frame, skip_history=(frame == 1))
        gpu_ctx.render(pc, cam, w, h)
    one_by_one = gpu_ctx.read_hdr()
    pc = default_pc(S, fl, frame_index=1, skip_history=True)
    gpu_ctx.render(pc, cam, w, h, frames=5)
    batched = gpu_ctx.read_hdr()
    assert same_bits(one_by_one, batched).all()


def test_counters_match_oracle(gpu_ctx, oracle, cornell_world):
    """BVH-independent work counters are exact integers shared by both implementations."""
    w = h = 128
    cam, fl = _camera(oracle, cornell_world, w, h)
    pc = default_pc(S, fl, max_bounces=4)
    gpu_ctx.upload_scene(cornell_world)
    gpu_ctx.reset_counters()
    gpu_ctx.render(pc, cam, w, h, flags=S.RENDER_COUNT_WORK)
    got = gpu_ctx.counters().as_dict()
    counted = gpu_ctx.read_hdr()
    gpu_ctx.render(pc, cam, w, h)
    assert same_bits(counted, gpu_ctx.read_hdr()).all(), "instrumented kernel changed the image"
    osc = oracle.OracleScene(cornell_world, brute_force=False)
    _, oc = osc.render(pc, cam, w, h)
    want = oc.as_dict()
    for key in ("paths", "closestRays", "shadowRays", "closestHits", "lightSamples", "spotLightSamples", "skyLookups",
                "pixelsWritten", "historyReads"):
        assert got[key] == want[key], key
    assert got["nodeVisits"] > 0 and got["triangleTests"] > 0
