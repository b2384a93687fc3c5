"""ReSTIR-DI trace, a second client of the traversal (res/shader/rt/direct_illumination/main.rgen):
the oracle's restatement on a synthetic G-buffer, and - with -m gpu - the HIP kernel against it bit for bit."""
import numpy as np
import pytest

from conftest import default_pc, same_bits
from prosper_amd import scenes, structs as S

FLAG_SKIP_HISTORY, FLAG_ACCUMULATE = 1, 2


def signed_oct_encode(n):
    """gbuffer.frag:41-58 (the inverse of material.glsl signedOctDecode), float64."""
    n = n / np.abs(n).sum(axis=-1, keepdims=True)
    out = np.empty(n.shape)
    out[..., 1] = n[..., 1] * 0.5 + 0.5
    out[..., 0] = n[..., 0] * 0.5 + out[..., 1]
    out[..., 1] = n[..., 0] * -0.5 + out[..., 1]
    out[..., 2] = np.clip(n[..., 2] * 1e30, 0.0, 1.0)
    return out


def make_gbuffer(oracle, world, w, h, seed=3):
    """A G-buffer for the scene's own camera, built from the oracle's debug renders of the primary hits:
    depth by projecting the hit position with worldToClip, oct-encoded shading normal, albedo / roughness /
    metallic, and a random light reservoir per pixel (some invalid, some out of range)."""
    c = world.camera
    cam, fl = oracle.camera_uniforms(c["eye"], c["target"], c["up"], c["fov"], c["zN"], c["zF"], w, h)
    osc = oracle.OracleScene(world, brute_force=True)

    def debug(name):
        img, _ = osc.render(default_pc(S, fl, draw_type=S.DrawType[name], max_bounces=1), cam, w, h)
        return img[..., :3].astype(np.float64)
    pos, raw_normal, alb = debug("Position"), debug("ShadingNormal"), debug("Albedo")
    rough, metal = debug("Roughness")[..., 0], debug("Metallic")[..., 0]
    hit = raw_normal.sum(axis=-1) > 0.0  # the debug view stores n * 0.5 + 0.5; a miss leaves the texel black
    nrm = raw_normal * 2.0 - 1.0
    nrm = np.where(hit[..., None], nrm, np.array([0.0, 0.0, 1.0]))
    c2c = np.frombuffer(bytes(cam.cameraToClip), np.float32).reshape(4, 4).T.astype(np.float64)
    w2c = np.frombuffer(bytes(cam.worldToCamera), np.float32).reshape(4, 4).T.astype(np.float64)
    m = c2c @ w2c
    clip = np.concatenate([pos, np.ones(pos.shape[:2] + (1,))], axis=-1) @ m.T
    depth = np.where(hit, clip[..., 2] / np.where(clip[..., 3] == 0, 1.0, clip[..., 3]), 0.0)
    ar = np.concatenate([alb, np.maximum(rough, 0.05)[..., None]], axis=-1).astype(np.float32)
    enc = signed_oct_encode(nrm)
    nm = np.stack([enc[..., 0], enc[..., 1], metal, enc[..., 2]], axis=-1).astype(np.float32)
    rng = np.random.default_rng(seed)
    light_count = 1 + world.point_lights.count + world.spot_lights.count
    idx = rng.integers(-1, light_count + 1, size=(h, w)).astype(np.int32)  # -1 invalid, light_count out of range
    res = np.stack([idx.view(np.float32), (rng.random((h, w)) * 4.0).astype(np.float32)], axis=-1)
    return cam, fl, osc, ar, nm, depth.astype(np.float32), res


def test_oracle_restir_trace_properties(oracle):
    world = scenes.cornell()
    w, h = 96, 64
    cam, fl, osc, ar, nm, depth, res = make_gbuffer(oracle, world, w, h)
    img = osc.restir_di_trace((0, 1, FLAG_SKIP_HISTORY | FLAG_ACCUMULATE), cam, ar, nm, depth, res)
    assert np.isfinite(img).all() and (img[..., 3] == 1.0).all()
    idx = res[..., 0].view(np.int32)
    assert (img[idx < 0][:, :3] == 0).all()  # invalid reservoir: no light (main.rgen:92-94)
    lit = img[..., :3].sum(axis=2) > 0
    assert 0.05 < lit.mean() < 0.9
    # world position reconstructed from depth matches the primary hits the G-buffer came from
    pos = osc.restir_di_trace((S.DrawType["Position"], 1, 0), cam, ar, nm, depth, res)
    want, _ = osc.render(default_pc(S, fl, draw_type=S.DrawType["Position"], max_bounces=1), cam, w, h)
    hit = depth != 0
    # uv = px / size (no half-pixel offset, main.rgen:119 "TODO: This is broken"): positions agree to a pixel footprint
    assert np.abs(pos[hit][:, :3] - want[hit][:, :3]).max() < 0.15
    # accumulation: second frame averages with the first
    img2 = osc.restir_di_trace((0, 2, FLAG_ACCUMULATE), cam, ar, nm, depth, res, history=img)
    assert (img2[..., 3] == 2.0).all() and same_bits(img2[..., :3], img[..., :3]).all()  # same reservoirs -> same colour


@pytest.mark.gpu
@pytest.mark.parametrize("scene", ["cornell", "sponza_small"])
def test_gpu_restir_trace_bit_exact(gpu_ctx, oracle, scene):
    world = scenes.cornell() if scene == "cornell" else scenes.sponza_class(
        lights=True, foliage=True, texture_size=64, sky_size=32, detail=0.25)
    w, h = 160, 96
    cam, fl, osc, ar, nm, depth, res = make_gbuffer(oracle, world, w, h)
    gpu_ctx.upload_scene(world)
    want = None
    for frame, flags in ((1, FLAG_SKIP_HISTORY | FLAG_ACCUMULATE), (2, FLAG_ACCUMULATE), (3, FLAG_ACCUMULATE)):
        res[..., 1] *= np.float32(0.9)  # a different weight every frame so that the running mean moves
        gpu_ctx.restir_di_trace(S.RestirTracePC(0, frame, flags), cam, ar, nm, depth, res)
        want = osc.restir_di_trace((0, frame, flags), cam, ar, nm, depth, res, history=want)
    got = gpu_ctx.read_hdr()
    ok = same_bits(got, want).all(axis=2)
    assert ok.all(), "%d of %d pixels differ" % ((~ok).sum(), ok.size)
    assert (got[..., 3] == 3.0).all() and (got[..., :3].sum(axis=2) > 0).mean() > 0.002  # random lights: mostly out of range
    for name in ("Position", "Albedo"):
        gpu_ctx.restir_di_trace(S.RestirTracePC(S.DrawType[name], 1, 0), cam, ar, nm, depth, res)
        assert same_bits(gpu_ctx.read_hdr(), osc.restir_di_trace((S.DrawType[name], 1, 0), cam, ar, nm, depth, res)).all()
