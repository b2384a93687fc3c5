"""A fourth integrator-level known answer that does not go through oracle/oracle.c: TWO bounces and Russian roulette.

The scene of test_integrator_bounce_known_answer.py (a tilted quad, a uniform sky, nothing that emits) plus a wall at
z = 22 facing back at it, maxBounces 3, rouletteStartBounce 0.  A path is then one of

    floor -> sky                     radiance  clamp(t0 * sky)                             (iteration 1 misses)
    floor -> wall -> sky             radiance  clamp(t0 * t1 * sky)  if it survives the roulette of bounce 1
    floor -> wall -> floor, floor -> floor (not from this camera)                          nothing: no light, no sky

with t0, t1 the weights of importanceSampleBounce at the two hits and the roulette of main.rgen:269-276 - only when
bounce > rouletteStartBounce (strictly), on the throughput AFTER the bounce, the path ends when rnd01() < max(0.05,
1 - max(t)), no compensation - all evaluated HERE in NumPy float64 with the reference's draw order (per iteration: light
pick, lobe pick, direction, then the roulette draw if the bounce qualifies; a miss draws nothing).

Pixels are not compared (and are counted) when a direction is within 0.02 of a horizon, when the wall is met within 0.1 of
its edges, or of the floor's far edge, or when the roulette draw is within 1e-5 of its threshold.  The oracle must agree within 5e-5 relative; the HIP
path must agree with the oracle bit for bit.
"""
import math

import numpy as np
import pytest

from conftest import default_pc, same_bits
from prosper_amd import scenes, structs as S

import test_integrator_bounce_known_answer as bounce
from test_integrator_known_answer import normalize, pcg3d, rng_to_01

W, H = bounce.W, bounce.H
RTOL, ATOL = 5e-5, 1e-6
WALL_Z = 22.0


def build_world():
    w = bounce.build_world()
    mat = w.add_material(base_color=bounce.ALBEDO + (1.0,), metallic=bounce.METALLIC, roughness=bounce.ROUGHNESS)
    # normal (0, 0, -1): counter-clockwise seen from -z
    mesh = scenes._add(w, scenes.quad((40, -40, WALL_Z), (-40, -40, WALL_Z), (-40, 40, WALL_Z), (40, 40, WALL_Z)), mat)
    w.add_instance(w.add_model([(mesh, mat)]))
    return w


def numpy_radiance(world, frame_index=1):
    cam = world.camera
    eye, target, up = (np.array(cam[k], np.float64) for k in ("eye", "target", "up"))
    fwd = normalize(target - eye)
    right = normalize(np.cross(fwd, up))
    upv = np.cross(right, fwd)
    tan_half = math.tan(cam["fov"] * 0.5)
    py, px = np.meshgrid(np.arange(H, dtype=np.uint32), np.arange(W, dtype=np.uint32), indexing="ij")
    state = pcg3d(np.stack([px, py, np.full_like(px, frame_index)], axis=-1))      # jitter
    jitter = rng_to_01(state[..., :2]).astype(np.float64)
    uv = (np.stack([px, py], axis=-1).astype(np.float64) + jitter) / np.array([W, H], np.float64)
    nd = uv * 2.0 - 1.0
    d = normalize(nd[..., :1] * right * (tan_half * W / H) - nd[..., 1:] * upv * tan_half + fwd)
    n_geo = np.array([0.0, 0.8, 0.6])
    t = -(eye * n_geo).sum() / (d * n_geo).sum(-1)
    p0 = eye + t[..., None] * d
    assert (p0[..., 2] < WALL_Z - 1.0).all(), "the camera sees the floor in front of the wall only"
    n_floor = normalize(np.array([0.0, 409.0, 307.0]))
    n_wall = np.array([0.0, 0.0, -1.0])
    sky = np.array(bounce.SKY, np.float64)

    def draws(state):
        state = pcg3d(state)                       # light pick (the sun emits nothing)
        state = pcg3d(state)                       # lobe pick
        pick = rng_to_01(state[..., 0]) < np.float32(0.5)
        state = pcg3d(state)                       # direction
        return state, pick, rng_to_01(state[..., :2]).astype(np.float64)

    # iteration 0: the floor
    state, pick0, u0 = draws(state)
    rd0, w0 = bounce.sample_bounce(n_floor, -d, pick0, u0)
    up0 = (rd0 * n_floor).sum(-1)
    t0 = np.where((up0 > 0)[..., None], np.maximum(w0, 0.0), 0.0)
    ok = np.minimum(np.abs(up0), np.abs((rd0 * n_geo).sum(-1))) > 0.02
    alive = up0 > 0                                 # zero throughput ends the path (and would add nothing anyway)

    # iteration 1: wall or sky
    with np.errstate(divide="ignore", invalid="ignore"):
        tw = (WALL_Z - p0[..., 2]) / rd0[..., 2]
    q = p0 + tw[..., None] * rd0
    hits_wall = alive & (rd0[..., 2] > 0) & (np.abs(q[..., 0]) < 40.0) & (q[..., 1] < 40.0)
    near_edge = (rd0[..., 2] > 0) & ((np.abs(np.abs(q[..., 0]) - 40.0) < 0.1) | (np.abs(q[..., 1] - 40.0) < 0.1))
    ok &= ~near_edge & (np.abs(rd0[..., 2]) > 1e-3)
    radiance = np.where((alive & ~hits_wall)[..., None], np.clip(t0 * sky, 0.0, 2.0), 0.0)

    # ... the wall: same draws, then the roulette of bounce 1 > rouletteStartBounce 0
    state1, pick1, u1 = draws(state)
    rd1, w1 = bounce.sample_bounce(n_wall, -rd0, pick1, u1)
    up1 = (rd1 * n_wall).sum(-1)
    t1 = t0 * np.where((up1 > 0)[..., None], np.maximum(w1, 0.0), 0.0)
    state1 = pcg3d(state1)
    rr = rng_to_01(state1[..., 0]).astype(np.float64)
    threshold = np.maximum(0.05, 1.0 - t1.max(-1))
    survives = ~(rr < threshold)
    # iteration 2: sky unless the ray comes down on the floor quad again (from high on the wall it can pass over its far edge)
    down = (rd1 * n_geo).sum(-1)
    with np.errstate(divide="ignore", invalid="ignore"):
        tf = -(q * n_geo).sum(-1) / down
    f = q + tf[..., None] * rd1
    fv = -0.6 * f[..., 1] + 0.8 * f[..., 2]        # the floor's second in-plane coordinate, v = (0, -0.6, 0.8)
    lands = (down < 0) & (np.abs(f[..., 0]) < 40.0) & (np.abs(fv) < 40.0)
    sees_sky = ~lands
    edge2 = (down < 0) & ((np.abs(np.abs(f[..., 0]) - 40.0) < 0.2) | (np.abs(np.abs(fv) - 40.0) < 0.2))
    second = hits_wall & (up1 > 0) & survives & sees_sky
    radiance = np.where(second[..., None], np.clip(t1 * sky, 0.0, 2.0), radiance)
    ok &= ~hits_wall | ((np.abs(up1) > 0.02) & (np.abs(down) > 0.02) & ~edge2 & (np.abs(rr - threshold) > 1e-5))
    return radiance, ok, hits_wall, second, hits_wall & (up1 > 0) & ~survives


def _check(img, want, ok, hits_wall, second, culled):
    got = img[..., :3].astype(np.float64)
    err = np.abs(got - want)
    bad = (err > RTOL * np.abs(want) + ATOL) & ok[..., None]
    assert not bad.any(), "%d channel values off; worst relative %g at %s" % (
        bad.sum(), (err / np.maximum(np.abs(want), 1e-300))[ok].max(), np.argwhere(bad)[0])
    n = W * H
    # not degenerate: a third of the paths meet the wall, hundreds go on to the sky, the roulette ends hundreds, and a
    # path it ended is black although it would have seen the sky
    assert ok.sum() > 0.9 * n and hits_wall.sum() > 0.2 * n
    assert (second & ok).sum() > 300 and (culled & ok).sum() > 300
    assert (got[culled & ok] == 0.0).all() and (got[second & ok].sum(-1) > 0).all()


def test_oracle_matches_the_numpy_two_bounce_roulette(oracle):
    world = build_world()
    cam, fl = bounce._camera(oracle, world)
    osc = oracle.OracleScene(world, brute_force=True)
    for frame in (1, 2):
        img, _ = osc.render(default_pc(S, fl, frame_index=frame, max_bounces=3, ibl=True, roulette=0), cam, W, H)
        _check(img, *numpy_radiance(world, frame_index=frame))


@pytest.mark.gpu
def test_hip_path_matches_oracle_bitwise_and_numpy_two_bounce_roulette(gpu_ctx, oracle):
    world = build_world()
    cam, fl = bounce._camera(oracle, world)
    pc = default_pc(S, fl, max_bounces=3, ibl=True, roulette=0)
    gpu_ctx.upload_scene(world)
    gpu_ctx.render(pc, cam, W, H)
    got = gpu_ctx.read_hdr()
    ref, _ = oracle.OracleScene(world, brute_force=True).render(pc, cam, W, H)
    assert same_bits(got, ref).all()
    _check(got, *numpy_radiance(world))
