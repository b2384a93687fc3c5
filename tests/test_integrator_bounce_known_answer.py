"""A second integrator-level known answer that does not go through oracle/oracle.c: the BOUNCE.

Scene: one quad in the plane 0.8 y + 0.6 z = 0 (metallic 0.5, roughness 0.6), no light that emits (the sun's irradiance
is zero), a uniform sky, maxBounces 2, one frame.  A pixel's radiance is then

    clamp(throughput * sky, 0, 2)      when the sampled bounce direction leaves the plane (main.rgen:249-253, 83-88)
    0                                  when it does not (NoL = 0: the throughput is zero)

with throughput = max(brdf * NoL / pdf, 0) of importanceSampleBounce (main.rgen:90-144), evaluated HERE in NumPy float64
straight from the shader text, with the reference's RNG draw order (jitter rnd2d01, light pick rnd01, lobe pick rnd01,
direction rnd2d01: SURVEY section 8a F1):

    res/shader/common/random.glsl:17-28,42-63     pcg3d, rngTo01
    res/shader/rt/ray.glsl:15-43                  pinholeCameraRay
    res/shader/common/sampling.glsl:18-93         cosineSampleHemisphere, orthonormalBasis, sampleVisibleTrowbridgeReitz, its pdf
    res/shader/brdf.glsl:9-64                     lambertBRFD, cookTorranceBRDF, fresnelZero
    res/shader/scene/geometry.glsl:95-114         the shading normal: snorm10-packed (0, 0.8, 0.6) decodes to normalize(0, 409, 307)

The plane is tilted because orthonormalBasis divides by sign(n.z) + n.z (sampling.glsl:37-47): n.z = 0 is its edge case.
The oracle must agree within 5e-5 relative (measured: 6e-6 at worst, 8e-8 median; the specular lobe's weight is a quotient of
two quantities that both vanish towards grazing directions, so the 1 % of pixels whose bounce direction is within 0.02 of
either horizon are not compared, and counted); the HIP path must agree with the oracle bit for bit.
"""
import math

import numpy as np
import pytest

from conftest import default_pc, same_bits
from prosper_amd import scenes, structs as S
from prosper_amd.world import World

from test_integrator_known_answer import normalize, pcg3d, rng_to_01, saturate

W, H = 160, 120
RTOL, ATOL = 5e-5, 1e-6
SKY = (0.5, 0.75, 2.0)  # 2 * albedo.z * sky.z = 2.4: the indirect clamp to [0, 2] (main.rgen:83-88) is active in one channel
ALBEDO = (0.8, 0.7, 0.6)
ROUGHNESS = 0.6
METALLIC = 0.5  # fresnelZero = mix(0.04, albedo, metallic) is coloured: the specular weight differs per channel
PI = 3.14159265  # math.glsl:4


def build_world(sky_faces=None, sky_texels=None, metallic=METALLIC):
    w = World()
    mat = w.add_material(base_color=ALBEDO + (1.0,), metallic=metallic, roughness=ROUGHNESS)
    # corners +-40 u +- 40 v with u = (1, 0, 0), v = (0, -0.6, 0.8): exactly representable in binary16
    mesh = scenes._add(w, scenes.quad((-40, -24, 32), (40, -24, 32), (40, 24, -32), (-40, 24, -32)), mat)
    w.add_instance(w.add_model([(mesh, mat)]))
    w.set_directional_light((1.0, 1.0, 1.0), 0.0, (-1.0, -1.0, -1.0))
    sky = np.empty((6, 4, 4, 4), np.float16)
    sky[..., :3] = np.asarray(SKY, np.float16) if sky_faces is None else np.asarray(sky_faces, np.float16)[:, None, None, :]
    sky[..., 3] = np.float16(1.0)
    if sky_texels is not None:
        sky = np.asarray(sky_texels, np.float16)
    w.skybox = sky
    # A grazing view (17 degrees above the plane at the centre: NoV from 0.09 to 0.5, so the masking terms and the
    # Fresnel term vary) of the part of the plane around 12 v.  Every hit point has |y| > 2 and |z| > 3, so the bounce
    # origin's offset (256 ulps per component, ray.glsl:83-103) clears the fp32 interpolation error of an 80-unit
    # quad - around y = z = 0 it does not, and some bounce rays meet the quad again there (a property of the
    # reference that float64 cannot predict pixel by pixel).
    w.camera = dict(eye=(0.0, -3.0, 6.5), target=(0.0, -7.2, 9.6), up=(0.0, 0.8, 0.6), fov=math.radians(24.0), zN=0.1, zF=100.0)
    return w


def onb_rows(n):
    """sampling.glsl:37-47 (Duff et al., transposed): rows b1, b2, n; M v = world -> local."""
    s = np.sign(n[2])
    a = -1.0 / (s + n[2])
    b = n[0] * n[1] * a
    return np.array([[1.0 + s * n[0] * n[0] * a, s * b, -s * n[0]], [b, s + n[1] * n[1] * a, -n[1]], n])


def sample_bounce(n, v, pick_diffuse, u, metallic=METALLIC):
    """importanceSampleBounce (main.rgen:90-144) for a surface with shading normal n (3,), view vectors v [..., 3], the
    lobe picks and the direction draws u [..., 2]: -> (direction [..., 3], weight brdf * NoL / pdf [..., 3]), float64."""
    specular_weight = 1.0 if metallic > 0.999 else 0.5             # main.rgen:92-95: a pure metal has no diffuse lobe
    diffuse_weight = 1.0 - specular_weight
    albedo = np.array(ALBEDO, np.float64)
    alpha = ROUGHNESS * ROUGHNESS
    m = onb_rows(n)
    phi = 2.0 * PI * u[..., 1]

    # diffuse lobe: sampling.glsl:18-35, brdf.glsl:9
    a = (1.0 - 2.0 * u[..., 0]) * 0.99999
    b = np.sqrt(1.0 - a * a) * 0.99999
    rd_diff = normalize(n + np.stack([b * np.cos(phi), b * np.sin(phi), a], axis=-1))
    nol_diff = saturate((rd_diff * n).sum(-1))
    with np.errstate(divide="ignore", invalid="ignore"):
        w_diff = (albedo / PI) * nol_diff[..., None] / (nol_diff / PI * diffuse_weight)[..., None]

    # specular lobe: sampling.glsl:53-93, brdf.glsl:12-64
    ve = v @ m.T
    vh = normalize(np.stack([alpha * ve[..., 0], alpha * ve[..., 1], ve[..., 2]], axis=-1))
    lensq = vh[..., 0] ** 2 + vh[..., 1] ** 2
    t1v = np.stack([-vh[..., 1], vh[..., 0], np.zeros_like(lensq)], axis=-1) / np.sqrt(lensq)[..., None]
    assert (lensq > 0).all()
    t2v = np.cross(vh, t1v)
    r = np.sqrt(u[..., 0])
    t1 = r * np.cos(phi)
    t2 = r * np.sin(phi)
    s = 0.5 * (1.0 + vh[..., 2])
    t2 = (1.0 - s) * np.sqrt(1.0 - t1 * t1) + s * t2
    nh = t1[..., None] * t1v + t2[..., None] * t2v + np.sqrt(np.maximum(0.0, 1.0 - t1 * t1 - t2 * t2))[..., None] * vh
    ne = normalize(np.stack([alpha * nh[..., 0], alpha * nh[..., 1], np.maximum(0.0, nh[..., 2])], axis=-1))
    le = -ve + 2.0 * (ne * ve).sum(-1, keepdims=True) * ne      # reflect(-ve, ne)
    rd_spec = le @ m                                            # transpose(M) * le: back to world space
    nol = saturate((rd_spec * n).sum(-1))
    h = normalize(v + rd_spec)
    noh = saturate((n * h).sum(-1))
    voh = saturate((v * h).sum(-1))
    nov = saturate((n * v).sum(-1))
    a2 = alpha * alpha
    denom = noh * noh * (a2 - 1.0) + 1.0
    dterm = a2 / (PI * denom * denom)
    k = max(alpha * 0.5, 0.0001)

    def g(nl, nv):
        return (nl / (nl * (1.0 - k) + k)) * (nv / (nv * (1.0 - k) + k))
    f0 = 0.04 * (1.0 - metallic) + albedo * metallic              # brdf.glsl:60-64
    fterm = f0 + (1.0 - f0) * ((1.0 - voh) ** 5.0)[..., None]
    brdf = fterm * (dterm * g(nol, nov) / (4.0 * nol * nov + 0.0001))[..., None]
    # visibleTrowbridgeReitzPdf in the local frame (sampling.glsl:81-93)
    hl = normalize(ve + le)
    nov_l, nol_l, noh_l = saturate(ve[..., 2]), saturate(le[..., 2]), saturate(hl[..., 2])
    d_l = a2 / (PI * (noh_l * noh_l * (a2 - 1.0) + 1.0) ** 2)
    with np.errstate(divide="ignore", invalid="ignore"):
        pdf = g(nol_l, nov_l) * nov_l * d_l / ve[..., 2] / (4.0 * nov_l) * specular_weight
        w_spec = brdf * (nol / pdf)[..., None]

    rd = np.where(pick_diffuse[..., None], rd_diff, rd_spec)
    weight = np.where(pick_diffuse[..., None], w_diff, w_spec)
    return rd, weight


def numpy_radiance(world, frame_index=1, sky_faces=None, sky_gradient=0, metallic=METALLIC):
    cam = world.camera
    eye, target, up = (np.array(cam[k], np.float64) for k in ("eye", "target", "up"))
    fwd = normalize(target - eye)
    right = normalize(np.cross(fwd, up))
    upv = np.cross(right, fwd)
    tan_half = math.tan(cam["fov"] * 0.5)
    aspect = W / H

    py, px = np.meshgrid(np.arange(H, dtype=np.uint32), np.arange(W, dtype=np.uint32), indexing="ij")
    state = pcg3d(np.stack([px, py, np.full_like(px, frame_index)], axis=-1))      # jitter
    jitter = rng_to_01(state[..., :2]).astype(np.float64)
    uv = (np.stack([px, py], axis=-1).astype(np.float64) + jitter) / np.array([W, H], np.float64)
    nd = uv * 2.0 - 1.0
    d = normalize(nd[..., :1] * right * (tan_half * aspect) - nd[..., 1:] * upv * tan_half + fwd)
    n_geo = np.array([0.0, 0.8, 0.6])
    assert ((d * n_geo).sum(-1) < -0.05).all(), "every primary ray must reach the plane"
    assert (d * n_geo).sum(-1).max() > -0.12 and (d * n_geo).sum(-1).min() < -0.4  # grazing to moderately steep
    t = -(eye * n_geo).sum() / (d * n_geo).sum(-1)
    p = eye + t[..., None] * d
    assert (np.abs(p[..., 0]) < 39).all() and (np.abs(p[..., 1]) > 2.0).all() and (np.abs(p[..., 1]) < 23).all()
    assert (np.abs(p[..., 2]) > 3.0).all()
    n = normalize(np.array([0.0, 409.0, 307.0]))  # packSnorm3x10_1x2(0, 0.8, 0.6) -> (0, 409, 307) / 511 -> normalize
    v = -d

    state = pcg3d(state)   # evaluateDirectLighting's light pick (main.rgen:205): the draw happens, the sun emits nothing
    state = pcg3d(state)   # importanceSampleBounce: lobe pick (main.rgen:100)
    pick_diffuse = rng_to_01(state[..., 0]) < np.float32(0.0 if metallic > 0.999 else 0.5)   # rnd01() < diffuseWeight
    state = pcg3d(state)   # direction (main.rgen:101)
    u = rng_to_01(state[..., :2]).astype(np.float64)

    rd, weight = sample_bounce(n, v, pick_diffuse, u, metallic)
    leaves = (rd * n).sum(-1) > 0.0
    throughput = np.where(leaves[..., None], np.maximum(weight, 0.0), 0.0)
    # not compared: directions within 0.02 of the shading horizon (NoL -> 0: the specular weight is 0 / 0-like) or of
    # the geometric one (the bounce ray may or may not meet the plane again)
    margin = np.minimum(np.abs((rd * n).sum(-1)), np.abs((rd * n_geo).sum(-1)))
    compared = margin > 0.02
    if sky_faces is None:
        sky = np.array(SKY, np.float64)
    else:
        # cube faces +X, -X, +Y, -Y, +Z, -Z by the direction's major axis; every face one colour, so a lookup whose
        # 2 x 2 footprint stays on the face returns it - directions whose second largest component exceeds 0.7 of the
        # largest (within a texel of a 4 x 4 face's border) are not compared
        face = sky_face_of(rd)
        sky = np.asarray(sky_faces, np.float64)[face]
        numpy_radiance.last_faces = np.where(compared & leaves, face, -1)
        srt = np.sort(np.abs(rd), axis=-1)
        compared &= srt[..., 1] < 0.7 * srt[..., 2]
    if sky_gradient:
        sky, inside = gradient_sky(rd, sky_gradient)
        compared = (margin > 0.02) & inside
    radiance = np.clip(throughput * sky, 0.0, 2.0)
    return radiance, pick_diffuse, compared, leaves


def _camera(oracle, world):
    c = world.camera
    return oracle.camera_uniforms(c["eye"], c["target"], c["up"], c["fov"], c["zN"], c["zF"], W, H)


def _check(img, want, pick_diffuse, compared, leaves):
    assert (img[..., 3] == 1.0).all()
    got = img[..., :3].astype(np.float64)
    err = np.abs(got - want)
    bad = (err > RTOL * np.abs(want) + ATOL) & compared[..., None]
    assert not bad.any(), "%d channel values off; worst relative %g at %s" % (
        bad.sum(), (err / np.maximum(np.abs(want), 1e-300))[compared].max(), np.argwhere(bad)[0])
    # the case is not degenerate: both lobes, directions that leave and directions that do not, few pixels skipped,
    # specular weights spread over a decade (not all clamped to 2 or all tiny)
    n = W * H
    assert 0.4 * n < pick_diffuse.sum() < 0.6 * n
    assert compared.sum() > 0.93 * n
    assert (~leaves & ~pick_diffuse).sum() > 50 and (got[~leaves & compared] == 0.0).all()
    spec = want[~pick_diffuse & leaves & compared][:, 0]
    assert spec.size > 0.3 * n and np.percentile(spec, 10) < 0.9 * np.percentile(spec, 90)
    chroma = want[~pick_diffuse & leaves & compared]
    assert (chroma[:, 0] / chroma[:, 2]).std() > 0  # and per channel: F is coloured
    diff = got[pick_diffuse & compared]
    assert np.allclose(diff, np.minimum(2.0 * np.array(ALBEDO) * np.array(SKY), 2.0), rtol=1e-5)  # 2 * albedo * sky


def test_oracle_matches_the_numpy_float64_bounce(oracle):
    world = build_world()
    cam, fl = _camera(oracle, world)
    osc = oracle.OracleScene(world, brute_force=True)
    for frame in (1, 2):
        want, pick, compared, leaves = numpy_radiance(world, frame_index=frame)
        img, counters = osc.render(default_pc(S, fl, frame_index=frame, max_bounces=2, ibl=True), cam, W, H)
        _check(img, want, pick, compared, leaves)
        assert counters.as_dict()["closestHits"] >= W * H


@pytest.mark.gpu
def test_hip_path_matches_oracle_bitwise_and_numpy_bounce(gpu_ctx, oracle):
    world = build_world()
    want, pick, compared, leaves = numpy_radiance(world)
    cam, fl = _camera(oracle, world)
    pc = default_pc(S, fl, max_bounces=2, ibl=True)
    gpu_ctx.upload_scene(world)
    gpu_ctx.render(pc, cam, W, H)
    got = gpu_ctx.read_hdr()
    ref, _ = oracle.OracleScene(world, brute_force=True).render(pc, cam, W, H)
    assert same_bits(got, ref).all()
    _check(got, want, pick, compared, leaves)


SKY_FACES = [(1.0, 0.25, 0.25), (0.25, 1.0, 0.25), (0.25, 0.25, 1.0), (1.0, 1.0, 0.25), (1.0, 0.25, 1.0), (0.25, 1.0, 1.0)]


def sky_face_of(rd):
    a = np.abs(rd)
    major = np.argmax(a, axis=-1)
    return 2 * major + (np.take_along_axis(rd, major[..., None], axis=-1)[..., 0] < 0)


def _check_faces(img, want, compared, leaves):
    got = img[..., :3].astype(np.float64)
    err = np.abs(got - want)
    assert (err[compared] <= RTOL * np.abs(want[compared]) + ATOL).all()
    assert compared.mean() > 0.5 and leaves[compared].mean() > 0.9


def test_oracle_matches_the_numpy_sky_face_lookup(oracle):
    """The same bounce under a sky whose six faces have six colours (skybox.glsl:4, main.rgen:249-253: textureLod on the
    cube with the bounce direction): which face a direction reads."""
    world = build_world(SKY_FACES)
    cam, fl = _camera(oracle, world)
    osc = oracle.OracleScene(world, brute_force=True)
    for frame in (1, 2):
        want, pick, compared, leaves = numpy_radiance(world, frame_index=frame, sky_faces=SKY_FACES)
        img, _ = osc.render(default_pc(S, fl, frame_index=frame, max_bounces=2, ibl=True), cam, W, H)
        _check_faces(img, want, compared, leaves)
        # the bounce directions read at least four of the six faces, hundreds of compared pixels each
        counts = np.bincount(numpy_radiance.last_faces[(numpy_radiance.last_faces >= 0) & compared], minlength=6)
        assert (counts > 300).sum() >= 4, counts


@pytest.mark.gpu
def test_hip_path_matches_oracle_bitwise_and_numpy_sky_faces(gpu_ctx, oracle):
    world = build_world(SKY_FACES)
    want, pick, compared, leaves = numpy_radiance(world, sky_faces=SKY_FACES)
    cam, fl = _camera(oracle, world)
    pc = default_pc(S, fl, max_bounces=2, ibl=True)
    gpu_ctx.upload_scene(world)
    gpu_ctx.render(pc, cam, W, H)
    got = gpu_ctx.read_hdr()
    ref, _ = oracle.OracleScene(world, brute_force=True).render(pc, cam, W, H)
    assert same_bits(got, ref).all()
    _check_faces(got, want, compared, leaves)


# ---- where on a cube face a direction lands: faces that are ramps in s and t (Vulkan spec, cube map face selection) ----

GRADIENT_N = 8


def gradient_texels(n=GRADIENT_N):
    """Face f, texel (i, j) = (i / (n - 1), j / (n - 1), f / 5): bilinear filtering reproduces a ramp exactly between texel centres."""
    sky = np.empty((6, n, n, 4), np.float16)
    j, i = np.meshgrid(np.arange(n), np.arange(n), indexing="ij")
    for f in range(6):
        sky[f, ..., 0] = i / (n - 1)
        sky[f, ..., 1] = j / (n - 1)
        sky[f, ..., 2] = f / 5
    sky[..., 3] = 1
    return sky


def gradient_sky(rd, n):
    """Major axis -> face and (sc, tc): +X (-z, -y), -X (z, -y), +Y (x, z), -Y (x, -z), +Z (x, -y), -Z (-x, -y); s, t = (c / |ma| + 1) / 2."""
    x, y, z = rd[..., 0], rd[..., 1], rd[..., 2]
    face = sky_face_of(rd)
    sc = np.choose(face, [-z, z, x, x, x, -x])
    tc = np.choose(face, [-y, -y, z, -z, -y, -y])
    ma = np.abs(np.choose(face, [x, x, y, y, z, z]))
    px = (sc / ma + 1.0) * 0.5 * n - 0.5          # texel space, centres at integers
    py = (tc / ma + 1.0) * 0.5 * n - 0.5
    inside = (px > 0.02) & (px < n - 1.02) & (py > 0.02) & (py < n - 1.02)   # all four taps on this face
    colour = np.stack([px / (n - 1), py / (n - 1), face / 5.0], axis=-1)
    # the texels are binary16: a ramp step of 1 / 7 is not exact, the filtered value is within 2^-11 relative of the ramp
    return colour, inside


def _check_gradient(img, want, compared, leaves):
    got = img[..., :3].astype(np.float64)
    err = np.abs(got - want)
    assert (err[compared] <= 1.5e-3 * np.abs(want[compared]) + 1e-4).all(), err[compared].max()
    assert compared.mean() > 0.5
    # ramps, not constants: the red and green channels vary across the compared pixels of a face
    lit = compared & leaves & (want[..., 0] > 0)
    assert want[lit][:, 0].std() > 0.02 and want[lit][:, 1].std() > 0.02


def test_oracle_matches_the_numpy_cube_face_coordinates(oracle):
    world = build_world(sky_texels=gradient_texels())
    cam, fl = _camera(oracle, world)
    osc = oracle.OracleScene(world, brute_force=True)
    for frame in (1, 2):
        want, pick, compared, leaves = numpy_radiance(world, frame_index=frame, sky_gradient=GRADIENT_N)
        img, _ = osc.render(default_pc(S, fl, frame_index=frame, max_bounces=2, ibl=True), cam, W, H)
        _check_gradient(img, want, compared, leaves)


@pytest.mark.gpu
def test_hip_path_matches_oracle_bitwise_and_numpy_cube_face_coordinates(gpu_ctx, oracle):
    world = build_world(sky_texels=gradient_texels())
    want, pick, compared, leaves = numpy_radiance(world, sky_gradient=GRADIENT_N)
    cam, fl = _camera(oracle, world)
    pc = default_pc(S, fl, max_bounces=2, ibl=True)
    gpu_ctx.upload_scene(world)
    gpu_ctx.render(pc, cam, W, H)
    got = gpu_ctx.read_hdr()
    ref, _ = oracle.OracleScene(world, brute_force=True).render(pc, cam, W, H)
    assert same_bits(got, ref).all()
    _check_gradient(got, want, compared, leaves)


# ---- a pure metal: metallic > 0.999 takes the specular lobe with weight 1, the lobe-pick draw still happens (main.rgen:92-100) ----

def test_oracle_matches_the_numpy_bounce_of_a_pure_metal(oracle):
    world = build_world(metallic=1.0)
    cam, fl = _camera(oracle, world)
    osc = oracle.OracleScene(world, brute_force=True)
    for frame in (1, 2):
        want, pick, compared, leaves = numpy_radiance(world, frame_index=frame, metallic=1.0)
        assert not pick.any()
        img, _ = osc.render(default_pc(S, fl, frame_index=frame, max_bounces=2, ibl=True), cam, W, H)
        got = img[..., :3].astype(np.float64)
        assert (np.abs(got - want)[compared] <= RTOL * np.abs(want[compared]) + ATOL).all()
        assert compared.mean() > 0.95 and (want[compared & leaves].min(-1) > 0).mean() > 0.99
        # twice the 50 / 50 material's specular weight would be 2x off: the weights are F G2 / G1 with F0 = albedo
        assert 0.3 < np.median(want[compared & leaves][:, 0] / SKY[0]) < 1.0


@pytest.mark.gpu
def test_hip_path_matches_oracle_bitwise_and_numpy_pure_metal(gpu_ctx, oracle):
    world = build_world(metallic=1.0)
    want, pick, compared, leaves = numpy_radiance(world, metallic=1.0)
    cam, fl = _camera(oracle, world)
    pc = default_pc(S, fl, max_bounces=2, ibl=True)
    gpu_ctx.upload_scene(world)
    gpu_ctx.render(pc, cam, W, H)
    got = gpu_ctx.read_hdr()
    ref, _ = oracle.OracleScene(world, brute_force=True).render(pc, cam, W, H)
    assert same_bits(got, ref).all()
    assert (np.abs(got[..., :3].astype(np.float64) - want)[compared] <= RTOL * np.abs(want[compared]) + ATOL).all()
