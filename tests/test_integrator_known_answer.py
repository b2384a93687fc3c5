"""An integrator-level known answer that does not go through oracle/oracle.c ("Next 1d").

Scene: one Lambert-ish quad (metallic 0, roughness 1) in the plane y = 0, one point light, one spot light, no sun
irradiance, maxBounces 1, one frame.  Every pixel's radiance is then a closed-form function of its jittered primary
ray, evaluated HERE in NumPy float64 straight from the shader text:

    res/shader/common/random.glsl:17-28,42-63   pcg3d, rngTo01 (the jitter and the light pick are integer arithmetic)
    res/shader/rt/ray.glsl:15-43                pinholeCameraRay (Y-flipped projection: SURVEY section 8a, F6)
    res/shader/rt/reference/main.rgen:195-223   evaluateDirectLighting: uniform pick of 1 + P + S lights, x lightCount
    res/shader/scene/lighting.glsl:15-56        point / spot light irradiance
    res/shader/brdf.glsl:9-87                   evalBRDFTimesNoL

The oracle must agree within 1e-5 relative, plus - for pixels in the spot cone's falloff zone only - the amplification
of cd's fp32 rounding by att = saturate(cd * scale + offset)^2 (2 * scale / x relative, 1e-5..1e-4 there), plus an
absolute floor of 1e-6 of the brightest pixel; the HIP path must agree with the oracle bit for bit.  The camera basis is recomputed here from
eye / target / up / fov as well, so the host camera code is cross-checked too.
"""
import math

import numpy as np
import pytest

from conftest import default_pc, same_bits
from prosper_amd import scenes, structs as S
from prosper_amd.world import World

W, H = 160, 120
RTOL, ATOL_OF_MAX = 1e-5, 1e-6


def build_world():
    w = World()
    mat = w.add_material(base_color=(0.8, 0.7, 0.6, 1.0), metallic=0.0, roughness=1.0)
    mesh = scenes._add(w, scenes.quad((-40, 0, 40), (40, 0, 40), (40, 0, -40), (-40, 0, -40)), mat)
    w.add_instance(w.add_model([(mesh, mat)]))
    w.add_point_light((1.0, 0.9, 0.8), 200.0, (1.0, 3.0, 0.5))
    d = np.array([0.3, -1.0, -0.2])
    w.add_spot_light((0.7, 0.8, 1.0), 300.0, (-2.0, 4.0, 1.0), d / np.linalg.norm(d), math.radians(20.0), math.radians(35.0))
    w.camera = dict(eye=(0.0, 2.0, 4.0), target=(0.0, 0.0, 0.0), up=(0.0, 1.0, 0.0), fov=math.radians(40.0), zN=0.1, zF=100.0)
    return w


def pcg3d(v):
    """random.glsl:17-28 on uint32 arrays [..., 3]."""
    v = (v.astype(np.uint64) * 1664525 + 1013904223) & 0xFFFFFFFF
    x, y, z = v[..., 0], v[..., 1], v[..., 2]
    x = (x + y * z) & 0xFFFFFFFF
    y = (y + z * x) & 0xFFFFFFFF
    z = (z + x * y) & 0xFFFFFFFF
    x, y, z = x ^ (x >> 16), y ^ (y >> 16), z ^ (z >> 16)
    x = (x + y * z) & 0xFFFFFFFF
    y = (y + z * x) & 0xFFFFFFFF
    z = (z + x * y) & 0xFFFFFFFF
    return np.stack([x, y, z], axis=-1)


def rng_to_01(u):
    """random.glsl:42: u / float(0xFFFFFFFFu); both conversions round to fp32 (float(0xFFFFFFFF) = 2^32)."""
    return (u.astype(np.float32) / np.float32(4294967296.0)).astype(np.float32)


def saturate(x):
    return np.clip(x, 0.0, 1.0)


def normalize(v):
    return v / np.linalg.norm(v, axis=-1, keepdims=True)


def eval_brdf_times_nol(l, n, v, albedo, roughness, metallic):
    """brdf.glsl:67-87 and the functions it calls, float64."""
    h = normalize(v + l)
    NoL = saturate((n * l).sum(-1))
    NoH = saturate((n * h).sum(-1))
    VoH = saturate((v * h).sum(-1))
    NoV = saturate((n * v).sum(-1))
    f0 = 0.04 * (1.0 - metallic) + albedo * metallic
    c_diff = albedo * (1.0 - 0.04) * (1.0 - metallic)
    alpha = roughness * roughness
    a2 = alpha * alpha
    denom = NoH * NoH * (a2 - 1.0) + 1.0
    D = a2 / (math.pi * denom * denom)
    F = f0 + (1.0 - f0) * ((1.0 - VoH) ** 5.0)[..., None]
    k = max(alpha * 0.5, 0.0001)
    G = (NoL / (NoL * (1.0 - k) + k)) * (NoV / (NoV * (1.0 - k) + k))
    spec = F * (D * G / (4.0 * NoL * NoV + 0.0001))[..., None]
    return (c_diff / math.pi + spec) * NoL[..., None]


def numpy_radiance(world, frame_index=1, lens=None, surface=None):
    """lens = (apertureDiameter, focusDistance): thinLensCameraRay (ray.glsl:46-78) instead of the pinhole ray - one more
    rnd2d01 draw between the jitter and the light pick (main.rgen:236-240).
    surface(p) -> (albedo [..., 3], roughness, metallic) replaces the material factors (textured materials)."""
    f = world.freeze()
    cam = world.camera
    eye, target, up = (np.array(cam[k], np.float64) for k in ("eye", "target", "up"))
    fwd = normalize(target - eye)
    right = normalize(np.cross(fwd, up))
    upv = np.cross(right, fwd)
    tan_half = math.tan(cam["fov"] * 0.5)
    aspect = W / H

    py, px = np.meshgrid(np.arange(H, dtype=np.uint32), np.arange(W, dtype=np.uint32), indexing="ij")
    state = np.stack([px, py, np.full_like(px, frame_index)], axis=-1)
    state = pcg3d(state)                                   # main.rgen:229-231: rnd2d01 jitter
    jitter = rng_to_01(state[..., :2]).astype(np.float64)
    uv = (np.stack([px, py], axis=-1).astype(np.float64) + jitter) / np.array([W, H], np.float64)
    nd = uv * 2.0 - 1.0
    # ray.glsl:21-24 with Camera::perspective's Y flip: cameraToClip[1][1] < 0, so aspect and tanHalfFovY are both
    # negative - their product (the x term) is positive, the y term changes sign
    d = normalize(nd[..., :1] * right * (tan_half * aspect) - nd[..., 1:] * upv * tan_half + fwd)
    origin = np.broadcast_to(eye, d.shape)
    if lens is not None:
        aperture, focus = lens
        state = pcg3d(state)                               # main.rgen:238: the lens offset, rnd2d01
        lo = rng_to_01(state[..., :2]).astype(np.float64)
        theta = lo[..., 0] * 2.0 * 3.14159265
        lu, lv = np.cos(theta) * np.sqrt(lo[..., 1]), np.sin(theta) * np.sqrt(lo[..., 1])
        focus_point = eye + d * (focus / (d * fwd).sum(-1))[..., None]
        coc = aperture / 2.0                               # focalLength / (2 * focalLength / aperture)
        origin = eye + right * (lu * coc)[..., None] + upv * (lv * coc)[..., None]   # cameraToWorld * (lensPos, 1)
        d = normalize(focus_point - origin)
    t = -origin[..., 1] / d[..., 1]
    assert (t > 0).all(), "every primary ray must reach the plane"
    p = origin + t[..., None] * d
    assert (np.abs(p[..., 0]) < 40).all() and (np.abs(p[..., 2]) < 40).all()
    n = np.array([0.0, 1.0, 0.0])
    v = -d

    state = pcg3d(state)                                   # main.rgen:205: rnd01 light pick, fp32 as written
    light_count = 1 + world.point_lights.count + world.spot_lights.count
    assert light_count == 3
    pick = np.minimum((rng_to_01(state[..., 0]) * np.float32(light_count)).astype(np.uint32), light_count - 1)

    mat = f["materials"][1]
    albedo = np.array([mat.baseColorFactor.x, mat.baseColorFactor.y, mat.baseColorFactor.z], np.float64)
    rough = max(float(mat.roughnessFactor), 0.05)
    metal = float(mat.metallicFactor)
    if surface is not None:
        res = surface(p)
        albedo, rough, metal = res[:3]
        if len(res) > 3:
            n = res[3]                                     # the shading normal (instance transform, normal map)

    pl = world.point_lights.lights[0]
    sl = world.spot_lights.lights[0]
    out = np.zeros((H, W, 3), np.float64)
    # sun (pick 0): WorldData.cpp:1537-1542 zeroes its irradiance in a scene with punctual lights only
    sun = world.directional.irradiance
    assert (sun.x, sun.y, sun.z) == (0.0, 0.0, 0.0)

    # point light, lighting.glsl:15-37
    pos = np.array([pl.position.x, pl.position.y, pl.position.z], np.float64)
    radiance = np.array([pl.radianceAndRadius.x, pl.radianceAndRadius.y, pl.radianceAndRadius.z], np.float64)
    radius = float(pl.radianceAndRadius.w)
    to_light = pos - p
    d2 = (to_light * to_light).sum(-1)
    dist = np.sqrt(d2)
    l = to_light / dist[..., None]
    q4 = (dist / radius) ** 4
    att = np.maximum(np.minimum(1.0 - q4, 1.0), 0.0)
    irr = radiance * (att / d2)[..., None]
    lit = (l * n).sum(-1) > 0
    c = irr * light_count * eval_brdf_times_nol(l, n, v, albedo, rough, metal)
    out = np.where(((pick == 1) & lit)[..., None], c, out)
    # conditioning of the range window near the light's radius: 1 - q^4 amplifies the fp32 rounding of q^4 (a few
    # half-ulps, taken as 8 * 2^-24) by q^4 / (1 - q^4)
    cond_point = np.where((pick == 1) & (att > 0) & (att < 1), 8.0 * 2.0 ** -24 * q4 / np.maximum(att, 1e-30), 0.0)
    # ... and of a light within 3 degrees of the shading horizon: NoL = dot(n, l) carries ~8 * 2^-24 of absolute rounding
    cond_point = np.maximum(cond_point, np.where((pick == 1) & lit & ((l * n).sum(-1) < 0.05), 8.0 * 2.0 ** -24 / np.maximum((l * n).sum(-1), 1e-30), 0.0))

    # spot light, lighting.glsl:39-56
    pos = np.array([sl.positionAndAngleOffset.x, sl.positionAndAngleOffset.y, sl.positionAndAngleOffset.z], np.float64)
    radiance = np.array([sl.radianceAndAngleScale.x, sl.radianceAndAngleScale.y, sl.radianceAndAngleScale.z], np.float64)
    scale, offset = float(sl.radianceAndAngleScale.w), float(sl.positionAndAngleOffset.w)
    direction = np.array([sl.direction.x, sl.direction.y, sl.direction.z], np.float64)
    to_light = pos - p
    d2 = (to_light * to_light).sum(-1)
    dist = np.sqrt(d2)
    l = to_light / dist[..., None]
    cd = (l * -direction).sum(-1)
    x = saturate(cd * scale + offset)
    att = x ** 2
    irr = radiance * (att / d2)[..., None]
    lit = (l * n).sum(-1) > 0
    c = irr * light_count * eval_brdf_times_nol(l, n, v, albedo, rough, metal)
    out = np.where(((pick == 2) & lit)[..., None], c, out)
    # conditioning of the spot cone's edge: the fp32 shader's cd = dot(-direction, l) carries a few half-ulps of
    # rounding (taken as 4 * 2^-24 here), which att = saturate(cd * scale + offset)^2 amplifies by 2 * scale / x
    cond = np.where((pick == 2) & (x > 0) & (x < 1), 2.0 * scale * 4.0 * 2.0 ** -24 / np.maximum(x, 1e-30), 0.0)
    cond = np.maximum(cond, np.where((pick == 2) & lit & ((l * n).sum(-1) < 0.05), 8.0 * 2.0 ** -24 / np.maximum((l * n).sum(-1), 1e-30), 0.0))
    return out, pick, np.maximum(cond, cond_point)


def _camera(oracle, world):
    c = world.camera
    return oracle.camera_uniforms(c["eye"], c["target"], c["up"], c["fov"], c["zN"], c["zF"], W, H)


def _check_against_numpy(img, want, pick, cond, rtol=RTOL):
    assert (img[..., 3] == 1.0).all()
    scale = want.max()
    err = np.abs(img[..., :3].astype(np.float64) - want)
    bound = (rtol + cond[..., None] * (rtol / RTOL)) * np.abs(want) + ATOL_OF_MAX * scale
    bad = err > bound
    assert not bad.any(), "%d channel values off; worst %g at %s (want %g)" % (
        bad.sum(), (err / np.maximum(np.abs(want), 1e-300)).max(), np.argwhere(bad)[0], want[tuple(np.argwhere(bad)[0])])
    # the case is not degenerate: all three picks occur, both punctual lights light thousands of pixels
    assert all((pick == k).sum() > 1000 for k in (0, 1, 2))
    assert (cond > 0).sum() > 200 and ((cond > 0) & (cond < 1e-4)).sum() > 0.9 * (cond > 0).sum()  # the falloff zone is small
    plain = (cond == 0) & (want.sum(-1) > 100 * ATOL_OF_MAX * scale)
    assert (err[plain] / want[plain]).max() < rtol  # everywhere else: 1e-5 relative, as stated
    assert ((want.sum(-1) > 0) & (pick == 1)).sum() > 1000 and ((want.sum(-1) > 0) & (pick == 2)).sum() > 500


def test_oracle_matches_the_numpy_float64_integrator(oracle):
    world = build_world()
    want, pick, cond = numpy_radiance(world)
    cam, fl = _camera(oracle, world)
    osc = oracle.OracleScene(world, brute_force=True)
    img, counters = osc.render(default_pc(S, fl, max_bounces=1), cam, W, H)
    _check_against_numpy(img, want, pick, cond)
    c = counters.as_dict()
    assert c["closestHits"] == W * H and c["lightSamples"] + c["spotLightSamples"] == W * H
    assert c["spotLightSamples"] == int((pick == 2).sum())
    # frame 2 has other jitter and other picks: same agreement
    want2, pick2, cond2 = numpy_radiance(world, frame_index=2)
    img2, _ = osc.render(default_pc(S, fl, frame_index=2, max_bounces=1), cam, W, H)
    _check_against_numpy(img2, want2, pick2, cond2)


@pytest.mark.gpu
def test_hip_path_matches_oracle_bitwise_and_numpy_within_tolerance(gpu_ctx, oracle):
    world = build_world()
    want, pick, cond = numpy_radiance(world)
    cam, fl = _camera(oracle, world)
    pc = default_pc(S, fl, max_bounces=1)
    gpu_ctx.upload_scene(world)
    gpu_ctx.render(pc, cam, W, H)
    got = gpu_ctx.read_hdr()
    ref, _ = oracle.OracleScene(world, brute_force=True).render(pc, cam, W, H)
    assert same_bits(got, ref).all()
    _check_against_numpy(got, want, pick, cond)


# ---- the same answer through the thin-lens camera (depth of field): ray.glsl:46-78, main.rgen:236-240 ----

LENS = (0.3, 4.5)  # apertureDiameter, focusDistance: a 15 cm circle of confusion radius at the lens - rays move by centimetres
RTOL_LENS = 5e-5   # the lens ray adds an fp32 normalize(focusPoint - origin) in front of everything: measured 2.4e-5 outside the spot's falloff zone


def _lens_pc(fl, frame_index=1):
    pc = default_pc(S, fl, frame_index=frame_index, max_bounces=1, dof=True)
    pc.apertureDiameter, pc.focusDistance = LENS
    return pc


def test_oracle_matches_the_numpy_float64_integrator_with_depth_of_field(oracle):
    world = build_world()
    cam, fl = _camera(oracle, world)
    osc = oracle.OracleScene(world, brute_force=True)
    for frame in (1, 2):
        want, pick, cond = numpy_radiance(world, frame_index=frame, lens=LENS)
        img, _ = osc.render(_lens_pc(fl, frame), cam, W, H)
        _check_against_numpy(img, want, pick, cond, RTOL_LENS)
        # the lens matters: the pinhole answer is a different image (and a different light pick per pixel: one draw earlier)
        pinhole, pinhole_pick, _ = numpy_radiance(world, frame_index=frame)
        assert (pinhole_pick != pick).mean() > 0.5 and np.abs(pinhole - want).max() > 0.1 * want.max()


@pytest.mark.gpu
def test_hip_path_matches_oracle_bitwise_and_numpy_with_depth_of_field(gpu_ctx, oracle):
    world = build_world()
    want, pick, cond = numpy_radiance(world, lens=LENS)
    cam, fl = _camera(oracle, world)
    pc = _lens_pc(fl)
    gpu_ctx.upload_scene(world)
    gpu_ctx.render(pc, cam, W, H)
    got = gpu_ctx.read_hdr()
    ref, _ = oracle.OracleScene(world, brute_force=True).render(pc, cam, W, H)
    assert same_bits(got, ref).all()
    _check_against_numpy(got, want, pick, cond, RTOL_LENS)


# ---- accumulation (main.rgen:285-298): frames 1..4 of the same answer, running mean with the sample count in alpha ----

def _running_mean(frames):
    """new = hist + (c - hist) / (hist.a + 1), first frame (skipHistory) -> (c, 1)."""
    hist, count = frames[0].copy(), 1.0
    for c in frames[1:]:
        hist = hist + (c - hist) / (count + 1.0)
        count += 1.0
    return hist, count


def _check_accumulated(img, frames):
    want, count = _running_mean([f[0] for f in frames])
    assert (img[..., 3] == count).all()
    scale = want.max()
    cond = np.max([f[2] for f in frames], axis=0)
    err = np.abs(img[..., :3].astype(np.float64) - want)
    # each frame within (RTOL + its conditioning) of its float64 value, so the mean within that of the mean of magnitudes
    mags = np.mean([np.abs(f[0]) for f in frames], axis=0)
    assert (err <= (2 * RTOL + cond[..., None]) * mags + ATOL_OF_MAX * scale).all()
    # and it IS a mean of different images: a pixel lit in one frame and unlit in another (another light pick) holds a fraction
    lit = np.stack([f[0].sum(-1) > 0 for f in frames])
    mixed = lit.any(0) & ~lit.all(0)
    assert mixed.sum() > 1000 and (img[..., :3].sum(-1)[mixed] > 0).all()


def test_oracle_accumulates_the_running_mean_of_the_numpy_frames(oracle):
    world = build_world()
    cam, fl = _camera(oracle, world)
    osc = oracle.OracleScene(world, brute_force=True)
    frames, img = [], None
    for frame in (1, 2, 3, 4):
        frames.append(numpy_radiance(world, frame_index=frame))
        img, _ = osc.render(default_pc(S, fl, frame_index=frame, max_bounces=1, skip_history=(frame == 1)), cam, W, H, history=img)
    _check_accumulated(img, frames)


@pytest.mark.gpu
def test_hip_path_accumulates_bitwise_like_the_oracle_and_the_numpy_mean(gpu_ctx, oracle):
    world = build_world()
    cam, fl = _camera(oracle, world)
    osc = oracle.OracleScene(world, brute_force=True)
    gpu_ctx.upload_scene(world)
    frames, ref = [], None
    for frame in (1, 2, 3, 4):
        frames.append(numpy_radiance(world, frame_index=frame))
        pc = default_pc(S, fl, frame_index=frame, max_bounces=1, skip_history=(frame == 1))
        gpu_ctx.render(pc, cam, W, H)
        ref, _ = osc.render(pc, cam, W, H, history=ref)
    got = gpu_ctx.read_hdr()
    assert same_bits(got, ref).all()
    _check_accumulated(got, frames)


# ---- the sun and an opaque shadow (lighting.glsl:58-70, main.rgen:49-60,195-223) ----

PLATE = (3.0, 7.0, 6.0, 2.0, 6.5)  # x0, x1, y, z0, z1: an opaque plate above and behind the camera


def build_world_sun(plate=None):
    plate = PLATE if plate is None else plate
    w = World()
    mat = w.add_material(base_color=(0.8, 0.7, 0.6, 1.0), metallic=0.0, roughness=1.0)
    mesh = scenes._add(w, scenes.quad((-40, 0, 40), (40, 0, 40), (40, 0, -40), (-40, 0, -40)), mat)
    w.add_instance(w.add_model([(mesh, mat)]))
    x0, x1, y, z0, z1 = plate
    mesh2 = scenes._add(w, scenes.quad((x0, y, z1), (x1, y, z1), (x1, y, z0), (x0, y, z0)), mat)
    w.add_instance(w.add_model([(mesh2, mat)]))
    w.set_directional_light((1.0, 0.9, 0.8), 2.0, (-1.0, -1.0, -1.0))  # un-normalised, as prosper's default (lights.h:9,18-19)
    w.camera = dict(eye=(0.0, 2.0, 4.0), target=(0.0, 0.0, 0.0), up=(0.0, 1.0, 0.0), fov=math.radians(40.0), zN=0.1, zF=100.0)
    return w


def numpy_radiance_sun(world, frame_index=1, plate=None):
    plate = PLATE if plate is None else plate
    cam = world.camera
    eye, target, up = (np.array(cam[k], np.float64) for k in ("eye", "target", "up"))
    fwd = normalize(target - eye)
    right = normalize(np.cross(fwd, up))
    upv = np.cross(right, fwd)
    tan_half = math.tan(cam["fov"] * 0.5)
    py, px = np.meshgrid(np.arange(H, dtype=np.uint32), np.arange(W, dtype=np.uint32), indexing="ij")
    state = pcg3d(np.stack([px, py, np.full_like(px, frame_index)], axis=-1))
    jitter = rng_to_01(state[..., :2]).astype(np.float64)
    uv = (np.stack([px, py], axis=-1).astype(np.float64) + jitter) / np.array([W, H], np.float64)
    nd = uv * 2.0 - 1.0
    d = normalize(nd[..., :1] * right * (tan_half * W / H) - nd[..., 1:] * upv * tan_half + fwd)
    p = eye + (-eye[1] / d[..., 1])[..., None] * d
    n = np.array([0.0, 1.0, 0.0])
    sun = world.directional
    irr = np.array([sun.irradiance.x, sun.irradiance.y, sun.irradiance.z], np.float64)
    l = -normalize(np.array([sun.direction.x, sun.direction.y, sun.direction.z], np.float64))   # lighting.glsl:66
    mat = world.freeze()["materials"][1]
    albedo = np.array([mat.baseColorFactor.x, mat.baseColorFactor.y, mat.baseColorFactor.z], np.float64)
    lv = np.broadcast_to(l, p.shape)
    c = irr * 1.0 * eval_brdf_times_nol(lv, n, -d, albedo, max(float(mat.roughnessFactor), 0.05), float(mat.metallicFactor))
    # the shadow ray p + t l, t in (0.1, 100), against the plate
    x0, x1, y, z0, z1 = plate
    t = y / l[1]
    hx, hz = p[..., 0] + t * l[0], p[..., 2] + t * l[2]
    inside = (hx > x0) & (hx < x1) & (hz > z0) & (hz < z1) & (t < 100.0)   # the sun's shadow ray ends at d = 100
    edge = np.minimum(np.minimum(np.abs(hx - x0), np.abs(hx - x1)), np.minimum(np.abs(hz - z0), np.abs(hz - z1)))
    assert t > 0.1
    return np.where(inside[..., None], 0.0, c), inside, edge > 1e-3


def _check_sun(img, want, shadowed, compared):
    err = np.abs(img[..., :3].astype(np.float64) - want)
    assert (err[compared] <= RTOL * np.abs(want[compared]) + ATOL_OF_MAX * want.max()).all()
    assert (img[..., :3][shadowed & compared] == 0.0).all()
    assert shadowed.sum() > 1000 and (~shadowed).sum() > 5000 and compared.mean() > 0.99


def test_oracle_matches_the_numpy_sun_and_its_shadow(oracle):
    world = build_world_sun()
    cam, fl = _camera(oracle, world)
    osc = oracle.OracleScene(world, brute_force=True)
    for frame in (1, 2):
        img, _ = osc.render(default_pc(S, fl, frame_index=frame, max_bounces=1), cam, W, H)
        _check_sun(img, *numpy_radiance_sun(world, frame_index=frame))


@pytest.mark.gpu
def test_hip_path_matches_oracle_bitwise_and_numpy_sun(gpu_ctx, oracle):
    world = build_world_sun()
    cam, fl = _camera(oracle, world)
    pc = default_pc(S, fl, max_bounces=1)
    gpu_ctx.upload_scene(world)
    gpu_ctx.render(pc, cam, W, H)
    got = gpu_ctx.read_hdr()
    ref, _ = oracle.OracleScene(world, brute_force=True).render(pc, cam, W, H)
    assert same_bits(got, ref).all()
    _check_sun(got, *numpy_radiance_sun(world))


# ---- textures in the shading (materials.glsl:47-119, geometry.glsl:246-295): which texel, which channel, which decode ----

BASE_TEXELS = np.array([[(200, 50, 50, 255), (50, 200, 50, 255)],
                        [(50, 50, 200, 255), (220, 220, 60, 255)]], np.uint8)   # [row j (v), column i (u)]
MR_TEXEL = (0, 180, 90, 255)   # g = roughness, b = metallic (materials.glsl:86-96)
UV_SCALE = 16.0                # the 2 x 2 texture repeats every 5 units: 2.5-unit texels


def build_world_textured(linear=False):
    w = World()
    f = S.FILTER_LINEAR if linear else S.FILTER_NEAREST
    nearest = w.add_sampler(f, f, S.WRAP_REPEAT, S.WRAP_REPEAT)
    base = w.add_texture(BASE_TEXELS)
    mr = w.add_texture(np.tile(np.array(MR_TEXEL, np.uint8), (2, 2, 1)))
    mat = w.add_material(base_color=(1.0, 1.0, 1.0, 1.0), metallic=1.0, roughness=1.0, base_tex=(base, nearest), mr_tex=(mr, nearest))
    mesh = scenes._add(w, scenes.quad((-40, 0, 40), (40, 0, 40), (40, 0, -40), (-40, 0, -40), uv_scale=UV_SCALE), mat)
    w.add_instance(w.add_model([(mesh, mat)]))
    w.add_point_light((1.0, 0.9, 0.8), 200.0, (1.0, 3.0, 0.5))
    d = np.array([0.3, -1.0, -0.2])
    w.add_spot_light((0.7, 0.8, 1.0), 300.0, (-2.0, 4.0, 1.0), d / np.linalg.norm(d), math.radians(20.0), math.radians(35.0))
    w.camera = dict(eye=(0.0, 2.0, 4.0), target=(0.0, 0.0, 0.0), up=(0.0, 1.0, 0.0), fov=math.radians(40.0), zN=0.1, zF=100.0)
    return w


def _srgb_to_linear(x):
    """materials.glsl:26-35."""
    return np.where(x <= 0.04045, x / 12.92, ((x + 0.055) / 1.055) ** 2.4)


def textured_surface(p, away):
    """The quad's corners carry uv (0,0) at (-40, 0, 40), (1,0) at (40, 0, 40), (1,1) at (40, 0, -40), times UV_SCALE."""
    u = (p[..., 0] + 40.0) / 80.0 * UV_SCALE
    v = (40.0 - p[..., 2]) / 80.0 * UV_SCALE
    fu, fv = u * 2.0, v * 2.0                          # nearest: texel = floor(uv * size) mod size (repeat)
    i, j = np.floor(fu).astype(np.int64) % 2, np.floor(fv).astype(np.int64) % 2
    away &= (np.abs(fu - np.round(fu)) > 1e-3) & (np.abs(fv - np.round(fv)) > 1e-3)
    texel = BASE_TEXELS[j, i].astype(np.float64) / 255.0
    albedo = _srgb_to_linear(texel[..., :3])           # sRGB decode of the colour, factor 1
    return albedo, max(MR_TEXEL[1] / 255.0, 0.05), MR_TEXEL[2] / 255.0


def _textured_answer(world, frame_index=1):
    away = np.ones((H, W), bool)
    want, pick, cond = numpy_radiance(world, frame_index=frame_index, surface=lambda p: textured_surface(p, away))
    return want, pick, cond, away


def _check_textured(img, want, pick, cond, away):
    err = np.abs(img[..., :3].astype(np.float64) - want)
    # 2e-5: the metallic surface's specular term and the fp32 pow of the sRGB decode double the plain case's error
    bound = 2.0 * (RTOL + cond[..., None]) * np.abs(want) + ATOL_OF_MAX * want.max()
    assert (err <= bound)[away].all(), "%d of %d compared channel values off" % ((err > bound)[away].sum(), away.sum() * 3)
    assert away.mean() > 0.98
    # all four texels are on screen and tint what they cover (red-ish, green-ish, blue-ish, yellow-ish)
    lit = (want.sum(-1) > 0) & away
    dominant = np.argmax(want, axis=-1)
    assert all((lit & (dominant == c)).sum() > 300 for c in range(3))


def test_oracle_matches_the_numpy_textured_surface(oracle):
    world = build_world_textured()
    cam, fl = _camera(oracle, world)
    osc = oracle.OracleScene(world, brute_force=True)
    for frame in (1, 2):
        img, _ = osc.render(default_pc(S, fl, frame_index=frame, max_bounces=1), cam, W, H)
        _check_textured(img, *_textured_answer(world, frame))


@pytest.mark.gpu
def test_hip_path_matches_oracle_bitwise_and_numpy_textured_surface(gpu_ctx, oracle):
    world = build_world_textured()
    cam, fl = _camera(oracle, world)
    pc = default_pc(S, fl, max_bounces=1)
    gpu_ctx.upload_scene(world)
    gpu_ctx.render(pc, cam, W, H)
    got = gpu_ctx.read_hdr()
    ref, _ = oracle.OracleScene(world, brute_force=True).render(pc, cam, W, H)
    assert same_bits(got, ref).all()
    _check_textured(got, *_textured_answer(world))


# ---- the shading frame: snorm10 normals and tangents, the instance's normal matrix, the normal map ----
# geometry.glsl:95-127 (decode + normalize), instances.glsl:36-53 (normal * mat3(inverse(M)), tangent * mat3(M)),
# main.rgen:37-45 (mappedNormal: B = sgn * cross(N, T)), materials.glsl:104-116 (normal = texel * 2 - 1)

NORMAL_TEXEL = (200, 100, 230, 255)
OBJECT_NORMAL = (0.6, 0.8, 0.0)      # tilted against the flat geometry: packs to (307, 409, 0) / 511
OBJECT_TANGENT = (0.8, -0.6, 0.0)    # packs to (409, -307, 0) / 511, sign +1


def _instance_matrix():
    from prosper_amd.world import rotate_y, scale
    return rotate_y(math.radians(30.0)) @ scale((2.0, 1.0, 0.5))


def build_world_shading_frame():
    w = World()
    lin = w.add_sampler()
    nt = w.add_texture(np.tile(np.array(NORMAL_TEXEL, np.uint8), (2, 2, 1)))
    mat = w.add_material(base_color=(0.8, 0.7, 0.6, 1.0), metallic=0.0, roughness=0.7, normal_tex=(nt, lin))
    pos = np.array([(-30, 0, 120), (30, 0, 120), (30, 0, -120), (-30, 0, -120)], np.float64)   # x2, x0.5 by the instance
    nrm = np.tile(np.array(OBJECT_NORMAL), (4, 1))
    tan = np.tile(np.array(OBJECT_TANGENT + (1.0,)), (4, 1))
    uvs = np.array([[0, 0], [1, 0], [1, 1], [0, 1]], np.float64)
    mesh = w.add_mesh(pos, np.array([0, 1, 2, 0, 2, 3], np.uint32), mat, normals=nrm, tangents=tan, uvs=uvs)
    w.add_instance(w.add_model([(mesh, mat)]), _instance_matrix())
    w.add_point_light((1.0, 0.9, 0.8), 200.0, (1.0, 3.0, 0.5))
    d = np.array([0.3, -1.0, -0.2])
    w.add_spot_light((0.7, 0.8, 1.0), 300.0, (-2.0, 4.0, 1.0), d / np.linalg.norm(d), math.radians(20.0), math.radians(35.0))
    w.camera = dict(eye=(0.0, 2.0, 4.0), target=(0.0, 0.0, 0.0), up=(0.0, 1.0, 0.0), fov=math.radians(40.0), zN=0.1, zF=100.0)
    return w


def shading_normal():
    m = _instance_matrix()[:3, :3]
    n_o = normalize(np.array([307.0, 409.0, 0.0]))         # unpackSnorm10 / 511, clamp, normalize
    t_o = normalize(np.array([409.0, -307.0, 0.0]))
    n_w = normalize(np.linalg.inv(m).T @ n_o)              # n * mat3(inverse(M)): the inverse transpose
    t_w = normalize(m @ t_o)                               # t * mat3(modelToWorld) with modelToWorld = transpose(M)
    b_w = 1.0 * np.cross(n_w, t_w)
    nt = np.array(NORMAL_TEXEL[:3], np.float64) / 255.0 * 2.0 - 1.0
    return normalize(nt[0] * t_w + nt[1] * b_w + nt[2] * n_w)


def _shading_frame_answer(world, frame_index=1):
    n = shading_normal()
    return numpy_radiance(world, frame_index=frame_index,
                          surface=lambda p: (np.array([0.8, 0.7, 0.6]), 0.7, 0.0, n))


def test_oracle_matches_the_numpy_shading_frame(oracle):
    n = shading_normal()
    assert n[1] > 0.5 and abs(n[0]) > 0.1 and abs(n[2]) > 0.1          # tilted off the geometric normal in both directions
    wrong = normalize(_instance_matrix()[:3, :3] @ normalize(np.array([307.0, 409.0, 0.0])))
    assert np.abs(wrong - normalize(np.linalg.inv(_instance_matrix()[:3, :3]).T @ normalize(np.array([307.0, 409.0, 0.0])))).max() > 0.2
    world = build_world_shading_frame()
    cam, fl = _camera(oracle, world)
    osc = oracle.OracleScene(world, brute_force=True)
    for frame in (1, 2):
        want, pick, cond = _shading_frame_answer(world, frame)
        img, _ = osc.render(default_pc(S, fl, frame_index=frame, max_bounces=1), cam, W, H)
        _check_against_numpy(img, want, pick, cond, 5e-5)


@pytest.mark.gpu
def test_hip_path_matches_oracle_bitwise_and_numpy_shading_frame(gpu_ctx, oracle):
    world = build_world_shading_frame()
    want, pick, cond = _shading_frame_answer(world)
    cam, fl = _camera(oracle, world)
    pc = default_pc(S, fl, max_bounces=1)
    gpu_ctx.upload_scene(world)
    gpu_ctx.render(pc, cam, W, H)
    got = gpu_ctx.read_hdr()
    ref, _ = oracle.OracleScene(world, brute_force=True).render(pc, cam, W, H)
    assert same_bits(got, ref).all()
    _check_against_numpy(got, want, pick, cond, 5e-5)


# ---- the debug draw types (debug.glsl:17-38, main.rgen:181-193,259-264, random.glsl:30-40) ----

def _pcg(v):
    """random.glsl:7-12."""
    v = np.asarray(v, np.uint64)
    state = (v * 747796405 + 2891336453) & 0xFFFFFFFF
    word = (((state >> ((state >> 28) + 4)) ^ state) * 277803737) & 0xFFFFFFFF
    return (word >> 22) ^ word


def uint_to_color(x):
    xr = _pcg(x)
    return np.stack([(xr >> 20) & 0x3FF, (xr >> 10) & 0x3FF, xr & 0x3FF], axis=-1).astype(np.float64) / 0x3FF


def _primary_hits(world, frame_index=1):
    """Hit points of the jittered camera rays on the plane y = 0 (the first lines of numpy_radiance)."""
    cam = world.camera
    eye, target, up = (np.array(cam[k], np.float64) for k in ("eye", "target", "up"))
    fwd = normalize(target - eye)
    right = normalize(np.cross(fwd, up))
    upv = np.cross(right, fwd)
    tan_half = math.tan(cam["fov"] * 0.5)
    py, px = np.meshgrid(np.arange(H, dtype=np.uint32), np.arange(W, dtype=np.uint32), indexing="ij")
    state = pcg3d(np.stack([px, py, np.full_like(px, frame_index)], axis=-1))
    uv = (np.stack([px, py], axis=-1).astype(np.float64) + rng_to_01(state[..., :2]).astype(np.float64)) / np.array([W, H], np.float64)
    nd = uv * 2.0 - 1.0
    d = normalize(nd[..., :1] * right * (tan_half * W / H) - nd[..., 1:] * upv * tan_half + fwd)
    return eye + (-eye[1] / d[..., 1])[..., None] * d


def numpy_debug_images(world_textured, world_frame):
    """-> {draw type: (world, image [H, W, 3], compared [H, W])} for one frame."""
    p = _primary_hits(world_textured)
    away = np.ones((H, W), bool)
    albedo, rough, metal = textured_surface(p, away)
    everywhere = np.ones((H, W), bool)
    u = (p[..., 0] + 40.0) / 80.0 * UV_SCALE
    v = (40.0 - p[..., 2]) / 80.0 * UV_SCALE
    tri = (p[..., 2] < -p[..., 0]).astype(np.uint32)       # the quad's triangles (0, 1, 2) and (0, 2, 3) meet on z = -x
    ones = np.ones((H, W, 3))
    return {
        "Position": (world_textured, p, everywhere),
        "TexCoord0": (world_textured, np.stack([u, v, np.zeros_like(u)], axis=-1), everywhere),
        "Albedo": (world_textured, albedo, away),
        "Roughness": (world_textured, ones * rough, everywhere),
        "Metallic": (world_textured, ones * metal, everywhere),
        "PrimitiveID": (world_textured, uint_to_color(tri), np.abs(p[..., 2] + p[..., 0]) > 1e-3),
        "MeshID": (world_textured, ones * uint_to_color(0), everywhere),
        "MaterialID": (world_textured, ones * uint_to_color(1), everywhere),
        "ShadingNormal": (world_frame, ones * (shading_normal() * 0.5 + 0.5), everywhere),
    }


def _check_debug(name, img, want, compared):
    assert (img[..., 3] == 1.0).all()
    err = np.abs(img[..., :3].astype(np.float64) - want)
    # positions and texture coordinates are interpolated from corners 40 units out: fp32 leaves a few 1e-6 of absolute error
    atol = 2e-5 if name in ("Position", "TexCoord0") else 2e-6
    assert (err[compared] <= 2e-5 * np.abs(want[compared]) + atol).all(), name
    assert compared.mean() > 0.98, name


def test_oracle_matches_the_numpy_debug_draw_types(oracle):
    textured, frame = build_world_textured(), build_world_shading_frame()
    scenes_ = {id(textured): oracle.OracleScene(textured, brute_force=True), id(frame): oracle.OracleScene(frame, brute_force=True)}
    for name, (world, want, compared) in numpy_debug_images(textured, frame).items():
        cam, fl = _camera(oracle, world)
        img, _ = scenes_[id(world)].render(default_pc(S, fl, max_bounces=1, draw_type=S.DrawType[name]), cam, W, H)
        _check_debug(name, img, want, compared)
    tri = numpy_debug_images(textured, frame)["PrimitiveID"][1]
    assert len(np.unique(tri.reshape(-1, 3), axis=0)) == 2          # both triangles are on screen


@pytest.mark.gpu
def test_hip_path_matches_oracle_bitwise_and_numpy_debug_draw_types(gpu_ctx, oracle):
    textured, frame = build_world_textured(), build_world_shading_frame()
    answers = numpy_debug_images(textured, frame)
    for world in (textured, frame):
        cam, fl = _camera(oracle, world)
        gpu_ctx.upload_scene(world)
        osc = oracle.OracleScene(world, brute_force=True)
        for name, (wld, want, compared) in answers.items():
            if wld is not world:
                continue
            pc = default_pc(S, fl, max_bounces=1, draw_type=S.DrawType[name])
            gpu_ctx.render(pc, cam, W, H)
            got = gpu_ctx.read_hdr()
            ref, _ = osc.render(pc, cam, W, H)
            assert same_bits(got, ref).all(), name
            _check_debug(name, got, want, compared)


# ---- bilinear filtering of the base colour (LOD 0, repeat): the filtered UNORM value is decoded, not the texels ----

def numpy_bilinear_albedo(p):
    u = (p[..., 0] + 40.0) / 80.0 * UV_SCALE
    v = (40.0 - p[..., 2]) / 80.0 * UV_SCALE
    x, y = u * 2.0 - 0.5, v * 2.0 - 0.5                    # texel space of the 2 x 2 texture, texel centres at integers
    x0, y0 = np.floor(x), np.floor(y)
    a, b = (x - x0)[..., None], (y - y0)[..., None]
    i0, j0 = x0.astype(np.int64) % 2, y0.astype(np.int64) % 2
    i1, j1 = (i0 + 1) % 2, (j0 + 1) % 2
    t = BASE_TEXELS.astype(np.float64)[..., :3] / 255.0
    top = t[j0, i0] * (1.0 - a) + t[j0, i1] * a
    bottom = t[j1, i0] * (1.0 - a) + t[j1, i1] * a
    return _srgb_to_linear(top * (1.0 - b) + bottom * b)


def test_oracle_matches_the_numpy_bilinear_albedo(oracle):
    world = build_world_textured(linear=True)
    cam, fl = _camera(oracle, world)
    img, _ = oracle.OracleScene(world, brute_force=True).render(
        default_pc(S, fl, max_bounces=1, draw_type=S.DrawType["Albedo"]), cam, W, H)
    want = numpy_bilinear_albedo(_primary_hits(world))
    err = np.abs(img[..., :3].astype(np.float64) - want)
    # uv carries ~1e-6 of fp32 interpolation error, a texel is 2.5 units: the filtered value moves by up to ~3e-6 * contrast
    assert (err <= 2e-5).all(), err.max()
    # and the image is not one of flat texels: most pixels hold a blend no texel has
    texels = _srgb_to_linear(BASE_TEXELS.reshape(-1, 4)[:, :3] / 255.0)
    nearest_texel = np.abs(want[..., None, :] - texels).max(-1).min(-1)
    assert (nearest_texel > 0.01).mean() > 0.8


@pytest.mark.gpu
def test_hip_path_matches_oracle_bitwise_and_numpy_bilinear_albedo(gpu_ctx, oracle):
    world = build_world_textured(linear=True)
    cam, fl = _camera(oracle, world)
    pc = default_pc(S, fl, max_bounces=1, draw_type=S.DrawType["Albedo"])
    gpu_ctx.upload_scene(world)
    gpu_ctx.render(pc, cam, W, H)
    got = gpu_ctx.read_hdr()
    ref, _ = oracle.OracleScene(world, brute_force=True).render(pc, cam, W, H)
    assert same_bits(got, ref).all()
    assert (np.abs(got[..., :3].astype(np.float64) - numpy_bilinear_albedo(_primary_hits(world))) <= 2e-5).all()


# ---- a rank's tile of the image: the RNG is seeded with ABSOLUTE pixel coordinates (main.rgen:227-229) ----

def _tile_columns(rank, ranks, stripe=16):
    return np.array([x for x in range(W) if (x // stripe) % ranks == rank])


@pytest.mark.parametrize("rank,ranks", [(1, 2), (3, 5)])
def test_oracle_rank_tile_is_the_numpy_image_at_its_columns(oracle, rank, ranks):
    from prosper_amd import tiling
    world = build_world()
    cam, fl = _camera(oracle, world)
    want, pick, cond = numpy_radiance(world)
    cols = _tile_columns(rank, ranks)
    img, _ = oracle.OracleScene(world, brute_force=True).render(
        default_pc(S, fl, max_bounces=1), cam, W, H, tile=tiling.tile_for_rank(rank, ranks))
    assert img.shape == (H, len(cols), 4)
    err = np.abs(img[..., :3].astype(np.float64) - want[:, cols])
    assert (err <= (RTOL + cond[:, cols, None]) * np.abs(want[:, cols]) + ATOL_OF_MAX * want.max()).all()
    # with local instead of absolute columns in the seed the picks - hence which pixels are lit - would differ
    assert ((want[:, cols].sum(-1) > 0) != (want[:, :len(cols)].sum(-1) > 0)).mean() > 0.2


@pytest.mark.gpu
def test_hip_path_rank_tile_bitwise_and_numpy(gpu_ctx, oracle):
    from prosper_amd import tiling
    world = build_world()
    cam, fl = _camera(oracle, world)
    want, pick, cond = numpy_radiance(world)
    tile = tiling.tile_for_rank(3, 5)
    cols = _tile_columns(3, 5)
    pc = default_pc(S, fl, max_bounces=1)
    gpu_ctx.upload_scene(world)
    gpu_ctx.render(pc, cam, W, H, tile=tile)
    got = gpu_ctx.read_hdr()
    ref, _ = oracle.OracleScene(world, brute_force=True).render(pc, cam, W, H, tile=tile)
    assert same_bits(got, ref).all()
    err = np.abs(got[..., :3].astype(np.float64) - want[:, cols])
    assert (err <= (RTOL + cond[:, cols, None]) * np.abs(want[:, cols]) + ATOL_OF_MAX * want.max()).all()


# ---- draw instances: model-instance order x sub-model order (World.cpp:480-513), ids through the debug draws ----

def build_world_instances():
    """Model A = two strips (meshes 0, 1 with materials 1, 2), model B = one strip (mesh 2, material 3); instances A, B, A
    side by side along x.  DrawInstance i then carries (mesh, material) = (0,1) (1,2) (2,3) (0,1) (1,2)."""
    from prosper_amd.world import translate
    w = World()
    mats = [w.add_material(base_color=(0.2 * (k + 1), 0.5, 0.5, 1.0), metallic=0.0, roughness=1.0) for k in range(3)]

    def strip(x0, mat):
        return scenes._add(w, scenes.quad((x0, 0, 40), (x0 + 1, 0, 40), (x0 + 1, 0, -40), (x0, 0, -40)), mat)
    a = w.add_model([(strip(0.0, mats[0]), mats[0]), (strip(1.0, mats[1]), mats[1])])
    b = w.add_model([(strip(0.0, mats[2]), mats[2])])
    w.add_instance(a, translate((-3.0, 0.0, 0.0)))      # x in [-3, -1)
    w.add_instance(b, translate((-1.0, 0.0, 0.0)))      # x in [-1, 0)
    w.add_instance(a, translate((0.0, 0.0, 0.0)))       # x in [0, 2)
    w.camera = dict(eye=(-0.5, 2.0, 4.0), target=(-0.5, 0.0, 0.0), up=(0.0, 1.0, 0.0), fov=math.radians(40.0), zN=0.1, zF=100.0)
    return w, mats


def numpy_instance_ids(world, mats):
    p = _primary_hits(world)
    x = p[..., 0]
    strip = np.floor(x + 3.0).astype(np.int64)                # 0..4 across the five strips, anything else is a miss
    on = (strip >= 0) & (strip <= 4)
    mesh = np.array([0, 1, 2, 0, 1])[np.clip(strip, 0, 4)]
    material = np.array([mats[0], mats[1], mats[2], mats[0], mats[1]])[np.clip(strip, 0, 4)]
    compared = np.abs(x + 3.0 - np.round(x + 3.0)) > 1e-3
    return on, mesh, material, compared


def _check_ids(img_mesh, img_mat, on, mesh, material, compared):
    for img, ids in ((img_mesh, mesh), (img_mat, material)):
        want = np.where(on[..., None], uint_to_color(ids.astype(np.uint32)), 0.0)   # a miss is black (no sky)
        assert (np.abs(img[..., :3] - want)[compared] < 2e-6).all()
    assert all(((mesh == k) & on & compared).sum() > 200 for k in range(3)) and (~on).sum() > 200


def test_oracle_matches_the_numpy_instance_and_material_ids(oracle):
    world, mats = build_world_instances()
    cam, fl = _camera(oracle, world)
    osc = oracle.OracleScene(world, brute_force=True)
    imgs = [osc.render(default_pc(S, fl, max_bounces=1, draw_type=S.DrawType[t]), cam, W, H)[0] for t in ("MeshID", "MaterialID")]
    _check_ids(*imgs, *numpy_instance_ids(world, mats))


@pytest.mark.gpu
def test_hip_path_matches_oracle_bitwise_and_numpy_instance_ids(gpu_ctx, oracle):
    world, mats = build_world_instances()
    cam, fl = _camera(oracle, world)
    gpu_ctx.upload_scene(world)
    osc = oracle.OracleScene(world, brute_force=True)
    imgs = []
    for t in ("MeshID", "MaterialID"):
        pc = default_pc(S, fl, max_bounces=1, draw_type=S.DrawType[t])
        gpu_ctx.render(pc, cam, W, H)
        imgs.append(gpu_ctx.read_hdr())
        assert same_bits(imgs[-1], osc.render(pc, cam, W, H)[0]).all()
    _check_ids(*imgs, *numpy_instance_ids(world, mats))


# ---- the sun's shadow ray is 100 units long (lighting.glsl:66): the same plate 121 units up the ray casts no shadow ----

FAR_PLATE = (67.0, 71.0, 70.0, 66.0, 70.5)   # PLATE moved by (64, 64, 64) along the ray: t = 70 sqrt(3) = 121


def test_oracle_sun_shadow_ray_ends_at_100(oracle):
    world = build_world_sun(FAR_PLATE)
    cam, fl = _camera(oracle, world)
    img, _ = oracle.OracleScene(world, brute_force=True).render(default_pc(S, fl, max_bounces=1), cam, W, H)
    want, shadowed, compared = numpy_radiance_sun(world, plate=FAR_PLATE)
    assert not shadowed.any() and (want.sum(-1) > 0).all()
    # the plate does lie on the shadow rays of the pixels the near plate shadows
    l = -normalize(np.array([-1.0, -1.0, -1.0]))
    p = _primary_hits(world)
    t = FAR_PLATE[2] / l[1]
    hx, hz = p[..., 0] + t * l[0], p[..., 2] + t * l[2]
    assert ((hx > FAR_PLATE[0]) & (hx < FAR_PLATE[1]) & (hz > FAR_PLATE[3]) & (hz < FAR_PLATE[4])).sum() > 1000 and t > 100.0
    err = np.abs(img[..., :3].astype(np.float64) - want)
    assert (err <= RTOL * np.abs(want) + ATOL_OF_MAX * want.max()).all()


@pytest.mark.gpu
def test_hip_path_sun_shadow_ray_ends_at_100(gpu_ctx, oracle):
    world = build_world_sun(FAR_PLATE)
    cam, fl = _camera(oracle, world)
    pc = default_pc(S, fl, max_bounces=1)
    gpu_ctx.upload_scene(world)
    gpu_ctx.render(pc, cam, W, H)
    got = gpu_ctx.read_hdr()
    ref, _ = oracle.OracleScene(world, brute_force=True).render(pc, cam, W, H)
    assert same_bits(got, ref).all() and (got[..., :3].sum(-1) > 0).all()


# ---- the shadow ray starts at tMin = 0.1 (main.rgen:217): a curb casts no shadow on the 5.8 cm of ground right behind it ----

CURB_X, CURB_H = 0.5, 0.5


def build_world_curb():
    w = build_world_sun(FAR_PLATE)            # the sun, the ground, the (irrelevant) far plate
    mat = 1
    curb = scenes._add(w, scenes.quad((CURB_X, 0, 40), (CURB_X, 0, -40), (CURB_X, CURB_H, -40), (CURB_X, CURB_H, 40)), mat)
    w.add_instance(w.add_model([(curb, mat)]))
    return w


def numpy_radiance_curb(world, frame_index=1):
    want, _, _ = numpy_radiance_sun(world, frame_index=frame_index, plate=FAR_PLATE)
    p = _primary_hits(world, frame_index)
    cam = world.camera
    eye = np.array(cam["eye"], np.float64)
    d = normalize(p - eye)
    # camera rays that meet the curb before the ground are not compared
    with np.errstate(divide="ignore", invalid="ignore"):
        tc = (CURB_X - eye[0]) / d[..., 0]
    yc = eye[1] + tc * d[..., 1]
    sees_curb = (tc > 0) & (yc > -1e-3) & (yc < CURB_H + 1e-3)
    # the shadow ray p + t (1, 1, 1) / sqrt(3) meets the plane x = CURB_X at height CURB_X - p.x, at t = (CURB_X - p.x) sqrt(3)
    gap = CURB_X - p[..., 0]
    t = gap * math.sqrt(3.0)
    shadowed = (gap > 0) & (gap < CURB_H) & (t > 0.1)
    skipped = (gap > 0) & (t <= 0.1)                       # the curb is there, the ray starts beyond it: lit
    edge = (np.abs(t - 0.1) > 2e-3) & (np.abs(gap - CURB_H) > 2e-3) & (np.abs(gap) > 2e-3)
    return np.where(shadowed[..., None], 0.0, want), shadowed, skipped, edge & ~sees_curb


def _check_curb(img, want, shadowed, skipped, compared):
    got = img[..., :3].astype(np.float64)
    err = np.abs(got - want)
    assert (err[compared] <= RTOL * np.abs(want[compared]) + ATOL_OF_MAX * want.max()).all()
    assert (got[shadowed & compared] == 0.0).all() and (got[skipped & compared].sum(-1) > 0).all()
    assert (shadowed & compared).sum() > 1000 and (skipped & compared).sum() > 150 and compared.mean() > 0.8


def test_oracle_shadow_ray_starts_at_tmin(oracle):
    world = build_world_curb()
    cam, fl = _camera(oracle, world)
    osc = oracle.OracleScene(world, brute_force=True)
    for frame in (1, 2):
        img, _ = osc.render(default_pc(S, fl, frame_index=frame, max_bounces=1), cam, W, H)
        _check_curb(img, *numpy_radiance_curb(world, frame))


@pytest.mark.gpu
def test_hip_path_shadow_ray_starts_at_tmin(gpu_ctx, oracle):
    world = build_world_curb()
    cam, fl = _camera(oracle, world)
    pc = default_pc(S, fl, max_bounces=1)
    gpu_ctx.upload_scene(world)
    gpu_ctx.render(pc, cam, W, H)
    got = gpu_ctx.read_hdr()
    ref, _ = oracle.OracleScene(world, brute_force=True).render(pc, cam, W, H)
    assert same_bits(got, ref).all()
    _check_curb(got, *numpy_radiance_curb(world))


# ---- no face culling (main.rgen:62-81: gl_RayFlagsNoneEXT): the quad seen from BELOW is hit and shaded with its own normal ----

def build_world_from_below():
    w = build_world()
    w.camera = dict(w.camera, eye=(0.0, -2.0, 4.0))
    return w


def test_oracle_matches_the_numpy_back_face(oracle):
    """The view vector is on the far side of the normal: NoV = saturate(dot(n, v)) = 0 kills the specular term (its masking
    factor), the diffuse term still sees the lights above the quad - the reference lights back faces from their front."""
    world = build_world_from_below()
    cam, fl = _camera(oracle, world)
    osc = oracle.OracleScene(world, brute_force=True)
    for frame in (1, 2):
        want, pick, cond = numpy_radiance(world, frame_index=frame)
        img, counters = osc.render(default_pc(S, fl, frame_index=frame, max_bounces=1), cam, W, H)
        _check_against_numpy(img, want, pick, cond)
        assert counters.as_dict()["closestHits"] == W * H
    # and that is the diffuse term alone: c_diff / pi * NoL * irradiance, the same for every view direction
    front, _, _ = numpy_radiance(build_world())
    assert want.max() < front.max()


@pytest.mark.gpu
def test_hip_path_matches_oracle_bitwise_and_numpy_back_face(gpu_ctx, oracle):
    world = build_world_from_below()
    want, pick, cond = numpy_radiance(world)
    cam, fl = _camera(oracle, world)
    pc = default_pc(S, fl, max_bounces=1)
    gpu_ctx.upload_scene(world)
    gpu_ctx.render(pc, cam, W, H)
    got = gpu_ctx.read_hdr()
    ref, _ = oracle.OracleScene(world, brute_force=True).render(pc, cam, W, H)
    assert same_bits(got, ref).all()
    _check_against_numpy(got, want, pick, cond)
