"""Seed-fixed procedural scenes for the BASELINE.json configs.

prosper bundles neither a Cornell box nor Sponza (only res/glTF/FlightHelmet, with textures and
res/env/storm.ktx missing from the mount), so the benchmark scenes are generated here and packed
with the reference's vertex formats through `World`:

  cornell()        C1/C2: box + 2 blocks + BLEND quad + MASK quad, point + spot light
  sponza_class()   C3:    ~262k-triangle atrium, 25 materials, 1024^2 textures, env cube, sun
  sponza_class(lights=True, foliage=True)  C4: + 512 point + 512 spot lights + 20k alpha quads
"""
import math

import numpy as np

from . import structs as S
from .world import World, rotate_x, rotate_y, rotate_z, scale, translate


def pcg3d(v):
    """res/shader/common/random.glsl:17-28 on uint32 arrays [..., 3] (used to seed textures/lights)."""
    v = np.asarray(v, dtype=np.uint32).copy()
    with np.errstate(over="ignore"):
        v = v * np.uint32(1664525) + np.uint32(1013904223)
        x, y, z = v[..., 0].copy(), v[..., 1].copy(), v[..., 2].copy()
        x += y * z
        y += z * x
        z += x * y
        x ^= x >> np.uint32(16)
        y ^= y >> np.uint32(16)
        z ^= z >> np.uint32(16)
        x += y * z
        y += z * x
        z += x * y
    return np.stack([x, y, z], axis=-1)


def _rand01(idx, kind, seed):
    idx = np.asarray(idx, dtype=np.uint32)
    v = np.stack([idx, np.full_like(idx, kind), np.full_like(idx, seed)], axis=-1)
    return pcg3d(v).astype(np.float64) / 4294967296.0


# ---------------------------------------------------------------------------------------------
# mesh helpers: all return (positions[N,3], normals[N,3], tangents[N,4], uvs[N,2], indices[M])
# ---------------------------------------------------------------------------------------------

def quad(p0, p1, p2, p3, uv_scale=1.0):
    """Quad p0,p1,p2,p3 counter-clockwise seen from the side its normal points to."""
    p = np.array([p0, p1, p2, p3], dtype=np.float64)
    e1, e2 = p[1] - p[0], p[3] - p[0]
    n = np.cross(e1, e2)
    n /= np.linalg.norm(n)
    t = e1 / np.linalg.norm(e1)
    normals = np.tile(n, (4, 1))
    tangents = np.tile(np.append(t, 1.0), (4, 1))
    uvs = np.array([[0, 0], [1, 0], [1, 1], [0, 1]], dtype=np.float64) * uv_scale
    idx = np.array([0, 1, 2, 0, 2, 3], dtype=np.uint32)
    return p, normals, tangents, uvs, idx


def merge(parts):
    pos, nor, tan, uv, idx = [], [], [], [], []
    base = 0
    for p, n, t, u, i in parts:
        pos.append(p)
        nor.append(n)
        tan.append(t)
        uv.append(u)
        idx.append(np.asarray(i, np.uint32) + np.uint32(base))
        base += len(p)
    return (np.concatenate(pos), np.concatenate(nor), np.concatenate(tan), np.concatenate(uv), np.concatenate(idx))


def box(lo=(-0.5, 0.0, -0.5), hi=(0.5, 1.0, 0.5), uv_scale=1.0):
    x0, y0, z0 = lo
    x1, y1, z1 = hi
    return merge([
        quad((x0, y0, z1), (x1, y0, z1), (x1, y1, z1), (x0, y1, z1), uv_scale),  # +z
        quad((x1, y0, z0), (x0, y0, z0), (x0, y1, z0), (x1, y1, z0), uv_scale),  # -z
        quad((x1, y0, z1), (x1, y0, z0), (x1, y1, z0), (x1, y1, z1), uv_scale),  # +x
        quad((x0, y0, z0), (x0, y0, z1), (x0, y1, z1), (x0, y1, z0), uv_scale),  # -x
        quad((x0, y1, z1), (x1, y1, z1), (x1, y1, z0), (x0, y1, z0), uv_scale),  # +y
        quad((x0, y0, z0), (x1, y0, z0), (x1, y0, z1), (x0, y0, z1), uv_scale),  # -y
    ])


def grid(nx, nz, size_x, size_z, height_fn=None, uv_scale=1.0):
    """Tessellated XZ plane facing +y, optionally displaced."""
    xs = np.linspace(-0.5 * size_x, 0.5 * size_x, nx + 1)
    zs = np.linspace(-0.5 * size_z, 0.5 * size_z, nz + 1)
    gx, gz = np.meshgrid(xs, zs, indexing="xy")
    gy = np.zeros_like(gx) if height_fn is None else height_fn(gx, gz)
    pos = np.stack([gx, gy, gz], axis=-1).reshape(-1, 3)
    # finite-difference normals
    dydx = np.gradient(gy, xs, axis=1)
    dydz = np.gradient(gy, zs, axis=0)
    nrm = np.stack([-dydx, np.ones_like(gy), -dydz], axis=-1).reshape(-1, 3)
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    tan = np.stack([np.ones_like(gy), dydx, np.zeros_like(gy)], axis=-1).reshape(-1, 3)
    tan /= np.linalg.norm(tan, axis=1, keepdims=True)
    tan = np.concatenate([tan, np.ones((tan.shape[0], 1))], axis=1)
    uv = np.stack([(gx / size_x + 0.5) * uv_scale, (gz / size_z + 0.5) * uv_scale], axis=-1).reshape(-1, 2)
    i = (np.arange(nz)[:, None] * (nx + 1) + np.arange(nx)[None, :]).reshape(-1)
    idx = np.stack([i, i + nx + 1, i + 1, i + 1, i + nx + 1, i + nx + 2], axis=1).reshape(-1)
    return pos, nrm, tan, uv, idx.astype(np.uint32)


def cylinder(radius, height, segments, rings, uv_scale=1.0, arc=2.0 * math.pi, arc_start=0.0):
    """Open cylinder (or arc of one) around +y, normals pointing outwards."""
    th = arc_start + np.linspace(0.0, arc, segments + 1)
    ys = np.linspace(0.0, height, rings + 1)
    gt, gy = np.meshgrid(th, ys, indexing="xy")
    pos = np.stack([radius * np.cos(gt), gy, radius * np.sin(gt)], axis=-1).reshape(-1, 3)
    nrm = np.stack([np.cos(gt), np.zeros_like(gt), np.sin(gt)], axis=-1).reshape(-1, 3)
    tan = np.stack([-np.sin(gt), np.zeros_like(gt), np.cos(gt), np.ones_like(gt)], axis=-1).reshape(-1, 4)
    uv = np.stack([(gt - arc_start) / arc * uv_scale * 2.0, gy / max(height, 1e-6) * uv_scale], axis=-1).reshape(-1, 2)
    i = (np.arange(rings)[:, None] * (segments + 1) + np.arange(segments)[None, :]).reshape(-1)
    idx = np.stack([i, i + segments + 1, i + 1, i + 1, i + segments + 1, i + segments + 2], axis=1).reshape(-1)
    return pos, nrm, tan, uv, idx.astype(np.uint32)


def transform_mesh(mesh, m):
    p, n, t, u, i = mesh
    m = np.asarray(m, np.float64)
    p2 = p @ m[:3, :3].T + m[:3, 3]
    nm = np.linalg.inv(m[:3, :3]).T
    n2 = n @ nm.T
    n2 /= np.linalg.norm(n2, axis=1, keepdims=True)
    t3 = t[:, :3] @ m[:3, :3].T
    t3 /= np.linalg.norm(t3, axis=1, keepdims=True)
    return p2, n2, np.concatenate([t3, t[:, 3:4]], axis=1), u, i


def _add(world, mesh, material, **kw):
    p, n, t, u, i = mesh
    return world.add_mesh(p, i, material, normals=n, tangents=t, uvs=u, **kw)


# ---------------------------------------------------------------------------------------------
# procedural textures (value distributions via pcg3d(texel, seed))
# ---------------------------------------------------------------------------------------------

def noise_texture(size, seed, base=(0.5, 0.5, 0.5), amplitude=0.25, cells=16, alpha=255):
    """sRGB-encoded value noise: smooth lattice noise + per-texel grain."""
    ys, xs = np.mgrid[0:size, 0:size]
    lattice = _rand01(np.arange((cells + 1) * (cells + 1)), 7, seed)[:, 0].reshape(cells + 1, cells + 1)
    lattice[-1, :] = lattice[0, :]
    lattice[:, -1] = lattice[:, 0]
    fx = xs / size * cells
    fy = ys / size * cells
    ix, iy = fx.astype(int), fy.astype(int)
    ax, ay = fx - ix, fy - iy
    ax, ay = ax * ax * (3 - 2 * ax), ay * ay * (3 - 2 * ay)
    v = (lattice[iy, ix] * (1 - ax) * (1 - ay) + lattice[iy, ix + 1] * ax * (1 - ay) +
         lattice[iy + 1, ix] * (1 - ax) * ay + lattice[iy + 1, ix + 1] * ax * ay)
    grain = _rand01((ys * size + xs).reshape(-1), 11, seed)[:, 0].reshape(size, size)
    val = (v - 0.5) * 2.0 * amplitude + (grain - 0.5) * 0.08
    rgb = np.clip(np.asarray(base)[None, None, :] + val[..., None], 0.0, 1.0)
    out = np.empty((size, size, 4), np.uint8)
    out[..., :3] = np.round(rgb * 255.0).astype(np.uint8)
    out[..., 3] = alpha
    return out


def metallic_roughness_texture(size, seed, roughness=0.7, metallic=0.0, amplitude=0.2):
    n = noise_texture(size, seed, base=(0, roughness, metallic), amplitude=amplitude, cells=8)
    n[..., 0] = 255
    if metallic == 0.0:
        n[..., 2] = 0
    return n


def normal_texture(size, seed, strength=0.35, cells=24):
    h = noise_texture(size, seed, base=(0.5, 0.5, 0.5), amplitude=0.5, cells=cells)[..., 0].astype(np.float64) / 255.0
    dx = (np.roll(h, -1, axis=1) - np.roll(h, 1, axis=1)) * strength * cells * 0.5
    dy = (np.roll(h, -1, axis=0) - np.roll(h, 1, axis=0)) * strength * cells * 0.5
    n = np.stack([-dx, -dy, np.ones_like(h)], axis=-1)
    n /= np.linalg.norm(n, axis=-1, keepdims=True)
    out = np.empty((size, size, 4), np.uint8)
    out[..., :3] = np.round((n * 0.5 + 0.5) * 255.0).astype(np.uint8)
    out[..., 3] = 255
    return out


def checker_alpha_texture(size=64, squares=8, rgb=(200, 200, 60)):
    ys, xs = np.mgrid[0:size, 0:size]
    on = (((xs * squares) // size + (ys * squares) // size) % 2) == 0
    out = np.empty((size, size, 4), np.uint8)
    out[..., 0], out[..., 1], out[..., 2] = rgb
    out[..., 3] = np.where(on, 255, 0)
    return out


def leaf_alpha_texture(size=128, seed=0x1EAF, blend=False):
    ys, xs = np.mgrid[0:size, 0:size]
    u = (xs + 0.5) / size * 2.0 - 1.0
    v = (ys + 0.5) / size * 2.0 - 1.0
    # leaf-shaped mask: |u| < (1-v^2)^1.5 * 0.6
    inside = np.abs(u) < np.power(np.clip(1.0 - v * v, 0.0, 1.0), 1.5) * 0.6
    tex = noise_texture(size, seed, base=(0.18, 0.42, 0.12), amplitude=0.15, cells=8)
    if blend:
        # soft alpha in [0.2, 0.9] inside, 0 outside
        a = np.clip(0.9 - np.abs(u) * 1.2 - np.abs(v) * 0.3, 0.2, 0.9)
        tex[..., 3] = np.where(inside, np.round(a * 255.0), 0).astype(np.uint8)
    else:
        tex[..., 3] = np.where(inside, 255, 0)
    return tex


def sky_cube(face_size=512, sun_dir=(1.0, 1.0, 1.0), sun_radiance=50.0):
    """Analytic sky gradient + sun disc, RGBA16F, faces +X,-X,+Y,-Y,+Z,-Z (Vulkan cube layout)."""
    n = face_size
    c = (np.arange(n) + 0.5) / n * 2.0 - 1.0
    sc, tc = np.meshgrid(c, c, indexing="xy")
    one = np.ones_like(sc)
    dirs = [
        np.stack([one, -tc, -sc], -1), np.stack([-one, -tc, sc], -1),
        np.stack([sc, one, tc], -1), np.stack([sc, -one, -tc], -1),
        np.stack([sc, -tc, one], -1), np.stack([-sc, -tc, -one], -1),
    ]
    sun = np.asarray(sun_dir, np.float64)
    sun /= np.linalg.norm(sun)
    out = np.empty((6, n, n, 4), np.float16)
    for f, d in enumerate(dirs):
        d = d / np.linalg.norm(d, axis=-1, keepdims=True)
        up = np.clip(d[..., 1], -1.0, 1.0)
        t = np.clip(up * 0.5 + 0.5, 0.0, 1.0)
        horizon = np.array([0.85, 0.9, 1.0]) * 0.9
        zenith = np.array([0.15, 0.35, 0.9]) * 1.2
        ground = np.array([0.25, 0.22, 0.2]) * 0.4
        sky = np.where(up[..., None] >= 0.0,
                       horizon + (zenith - horizon) * np.power(np.clip(up, 0, 1), 0.6)[..., None],
                       horizon + (ground - horizon) * np.power(np.clip(-up, 0, 1), 0.4)[..., None])
        cosang = np.clip((d * sun).sum(-1), -1.0, 1.0)
        disc = np.clip((cosang - 0.9985) / (1.0 - 0.9985), 0.0, 1.0)
        glow = np.power(np.clip(cosang, 0.0, 1.0), 64.0) * 1.5
        rgb = sky + (disc * sun_radiance + glow)[..., None] * np.array([1.0, 0.95, 0.85])
        out[f, ..., :3] = np.clip(rgb, 0.0, sun_radiance).astype(np.float16)
        out[f, ..., 3] = np.float16(1.0)
        del t
    return out


# ---------------------------------------------------------------------------------------------
# S-cornell (C1, C2)
# ---------------------------------------------------------------------------------------------

def cornell(with_skybox=False):
    w = World()
    white = w.add_material(base_color=(0.73, 0.73, 0.73, 1.0), metallic=0.0, roughness=1.0)
    red = w.add_material(base_color=(0.65, 0.05, 0.05, 1.0), metallic=0.0, roughness=0.9)
    green = w.add_material(base_color=(0.12, 0.45, 0.15, 1.0), metallic=0.0, roughness=0.8)
    metal = w.add_material(base_color=(0.9, 0.85, 0.7, 1.0), metallic=1.0, roughness=0.2)
    blend = w.add_material(base_color=(0.2, 0.4, 0.9, 0.5), metallic=0.0, roughness=0.5,
                           alpha_mode=S.ALPHA_MODE_BLEND)
    checker = w.add_texture(checker_alpha_texture())
    nearest_clamp = w.add_sampler(S.FILTER_LINEAR, S.FILTER_LINEAR, S.WRAP_CLAMP_TO_EDGE, S.WRAP_MIRRORED_REPEAT)
    mask = w.add_material(base_color=(1.0, 1.0, 1.0, 1.0), metallic=0.0, roughness=0.6, alpha_cutoff=0.5,
                          alpha_mode=S.ALPHA_MODE_MASK, base_tex=(checker, nearest_clamp))

    # room: x in [-1,1], y in [0,2], z in [-1,1], open towards +z
    room_white = merge([
        quad((-1, 0, 1), (1, 0, 1), (1, 0, -1), (-1, 0, -1)),   # floor (normal +y)
        quad((-1, 2, -1), (1, 2, -1), (1, 2, 1), (-1, 2, 1)),   # ceiling (normal -y)
        quad((-1, 0, -1), (1, 0, -1), (1, 2, -1), (-1, 2, -1)),  # back wall (normal +z)
    ])
    left = quad((-1, 0, 1), (-1, 0, -1), (-1, 2, -1), (-1, 2, 1))   # normal +x
    right = quad((1, 0, -1), (1, 0, 1), (1, 2, 1), (1, 2, -1))      # normal -x
    m_white = _add(w, room_white, white)
    m_left = _add(w, left, red)
    m_right = _add(w, right, green)
    room = w.add_model([(m_white, white), (m_left, red), (m_right, green)])
    w.add_instance(room)

    cube = _add(w, box(), white)
    block_white = w.add_model([(cube, white)])
    block_metal = w.add_model([(cube, metal)])
    # tall block, rotated, back left; short block, rotated, front right (instance transforms: T6)
    w.add_instance(block_white, translate((-0.35, 0.0, -0.3)) @ rotate_y(math.radians(17.0)) @ scale((0.6, 1.2, 0.6)))
    w.add_instance(block_metal, translate((0.4, 0.0, 0.35)) @ rotate_y(math.radians(-18.0)) @ scale((0.55, 0.55, 0.55)))

    q_blend = _add(w, quad((-0.9, 0.9, 0.55), (-0.2, 0.9, 0.55), (-0.2, 1.6, 0.55), (-0.9, 1.6, 0.55)), blend)
    q_mask = _add(w, quad((0.15, 0.7, 0.75), (0.9, 0.7, 0.75), (0.9, 1.45, 0.75), (0.15, 1.45, 0.75), uv_scale=1.5), mask)
    w.add_instance(w.add_model([(q_blend, blend)]))
    w.add_instance(w.add_model([(q_mask, mask)]))

    w.add_point_light((1.0, 1.0, 1.0), 20.0, (0.0, 1.9, 0.0))
    inner, outer = math.radians(20.0), math.radians(35.0)
    w.add_spot_light((1.0, 0.9, 0.8), 30.0, (0.6, 1.8, 0.8), (-0.4, -1.0, -0.5), inner, outer)
    # no sun: punctual lights only zero the default directional light (WorldData.cpp:1537-1542)
    w.camera = dict(eye=(0.0, 1.0, 3.4), target=(0.0, 1.0, 0.0), up=(0.0, 1.0, 0.0),
                    fov=math.radians(40.0), zN=0.1, zF=100.0)
    if with_skybox:
        w.skybox = sky_cube(64)
    return w


# ---------------------------------------------------------------------------------------------
# S-sponza-class (C3-C5)
# ---------------------------------------------------------------------------------------------

def sponza_class(lights=False, foliage=False, texture_size=1024, sky_size=512, detail=1.0):
    """Procedural atrium: floor tiles, two storeys of colonnades with arches, walls, roof beams,
    hanging cloth; ~262k triangles at detail=1 (48 mesh primitives, 12 models, 40 instances)."""
    w = World()
    rng_seed = 0x5EED
    tex_sampler = 0
    mats = []
    palette = [
        (0.62, 0.58, 0.52), (0.55, 0.50, 0.45), (0.70, 0.66, 0.60), (0.45, 0.42, 0.40), (0.66, 0.55, 0.42),
        (0.50, 0.30, 0.22), (0.35, 0.36, 0.40), (0.58, 0.60, 0.62), (0.40, 0.25, 0.18), (0.72, 0.70, 0.65),
        (0.60, 0.12, 0.10), (0.10, 0.25, 0.55), (0.15, 0.45, 0.20), (0.75, 0.62, 0.20), (0.48, 0.46, 0.44),
        (0.52, 0.48, 0.40), (0.64, 0.60, 0.58), (0.30, 0.30, 0.32), (0.80, 0.78, 0.74), (0.57, 0.44, 0.33),
        (0.42, 0.40, 0.36), (0.68, 0.64, 0.54), (0.36, 0.32, 0.30), (0.90, 0.72, 0.30), (0.76, 0.76, 0.80),
    ]
    for i, base in enumerate(palette):
        metallic = 1.0 if i in (23, 24) else 0.0
        rough = 0.35 if metallic else 0.55 + 0.4 * ((i * 7) % 10) / 10.0
        bt = w.add_texture(noise_texture(texture_size, rng_seed + 3 * i, base=base, amplitude=0.18, cells=12 + i))
        mr = w.add_texture(metallic_roughness_texture(texture_size, rng_seed + 3 * i + 1, roughness=rough,
                                                      metallic=metallic))
        nt = w.add_texture(normal_texture(texture_size, rng_seed + 3 * i + 2, strength=0.25 + 0.02 * i))
        mats.append(w.add_material(base_color=(1, 1, 1, 1), metallic=1.0, roughness=1.0,
                                   base_tex=(bt, tex_sampler), mr_tex=(mr, tex_sampler), normal_tex=(nt, tex_sampler)))

    # 1.2859375 calibrates detail=1 to 262 426 triangles (SURVEY §8d asks for 262 144 +-1 %)
    d = detail * 1.2859375
    L, W, H = 24.0, 10.0, 9.0  # atrium interior: x in [-12,12], z in [-5,5], y in [0,9]

    def seg(n):
        return max(2, int(round(n * math.sqrt(d))))

    # --- model 0: floor + ceiling + 4 walls, heavily tessellated with gentle relief (6 primitives)
    def bumps(gx, gz):
        return 0.01 * np.sin(gx * 2.1) * np.cos(gz * 1.7)

    floor = grid(seg(160), seg(80), L, W, bumps, uv_scale=12.0)
    # the roof is open over the central half of the atrium (sun and sky enter there): two strips
    ceiling = merge([
        transform_mesh(grid(seg(96), seg(24), L, W * 0.25, None, uv_scale=8.0),
                       translate((0, H, zc)) @ rotate_x(math.pi))
        for zc in (-W * 0.375, W * 0.375)])
    wall_l = transform_mesh(grid(seg(128), seg(56), L, H, None, 8.0),
                            translate((0, H / 2, -W / 2)) @ rotate_x(math.pi / 2))
    wall_r = transform_mesh(grid(seg(128), seg(56), L, H, None, 8.0),
                            translate((0, H / 2, W / 2)) @ rotate_x(-math.pi / 2))
    wall_b = transform_mesh(grid(seg(56), seg(56), W, H, None, 4.0),
                            translate((-L / 2, H / 2, 0)) @ rotate_z(-math.pi / 2) @ rotate_y(math.pi / 2))
    wall_f = transform_mesh(grid(seg(56), seg(56), W, H, None, 4.0),
                            translate((L / 2, H / 2, 0)) @ rotate_z(math.pi / 2) @ rotate_y(math.pi / 2))
    shell = [(_add(w, floor, mats[0]), mats[0]), (_add(w, ceiling, mats[1]), mats[1]),
             (_add(w, wall_l, mats[2]), mats[2]), (_add(w, wall_r, mats[2]), mats[2]),
             (_add(w, wall_b, mats[3]), mats[3]), (_add(w, wall_f, mats[3]), mats[3])]
    w.add_instance(w.add_model(shell))

    # --- model 1: column = base box + shaft + capital (3 primitives), 16 instances
    shaft = cylinder(0.28, 3.2, seg(48), seg(24), uv_scale=2.0)
    base = box((-0.4, 0.0, -0.4), (0.4, 0.3, 0.4))
    capital = transform_mesh(box((-0.45, 0.0, -0.45), (0.45, 0.3, 0.45)), translate((0, 3.5, 0)))
    shaft = transform_mesh(shaft, translate((0, 0.3, 0)))
    column = w.add_model([(_add(w, base, mats[4]), mats[4]), (_add(w, shaft, mats[5]), mats[5]),
                          (_add(w, capital, mats[4]), mats[4])])
    # --- model 2: upper-storey column, thinner (3 primitives, u32 indices on the shaft), 8 instances
    shaft2 = transform_mesh(cylinder(0.2, 2.6, seg(40), seg(20), uv_scale=2.0), translate((0, 0.2, 0)))
    base2 = box((-0.3, 0.0, -0.3), (0.3, 0.2, 0.3))
    capital2 = transform_mesh(box((-0.33, 0.0, -0.33), (0.33, 0.2, 0.33)), translate((0, 2.8, 0)))
    column2 = w.add_model([(_add(w, base2, mats[6]), mats[6]),
                           (_add(w, shaft2, mats[7], force_u32_indices=True), mats[7]),
                           (_add(w, capital2, mats[6]), mats[6])])
    xs_cols = np.linspace(-L / 2 + 1.5, L / 2 - 1.5, 8)
    for x in xs_cols:
        w.add_instance(column, translate((x, 0.0, -3.4)))
        w.add_instance(column, translate((x, 0.0, 3.4)))
    for x in xs_cols[::2]:
        w.add_instance(column2, translate((x, 4.4, -3.4)))
        w.add_instance(column2, translate((x, 4.4, 3.4)))

    # --- model 3: arch between columns (half-cylinder soffit + 2 faces = 3 primitives), 2 instances of a row model
    def arch_row(z, mat_a, mat_b):
        parts_soffit, parts_face = [], []
        for x0, x1 in zip(xs_cols[:-1], xs_cols[1:]):
            cx, r = 0.5 * (x0 + x1), 0.5 * (x1 - x0) - 0.3
            soffit = cylinder(r, 0.8, seg(24), seg(4), uv_scale=1.0, arc=math.pi, arc_start=0.0)
            soffit = transform_mesh(soffit, translate((cx, 3.8, z - 0.4)) @ rotate_x(math.pi / 2))
            parts_soffit.append(soffit)
            face = grid(seg(12), seg(4), x1 - x0, 0.6, None, 1.0)
            parts_face.append(transform_mesh(face, translate((cx, 4.4, z)) @ rotate_x(math.pi / 2)))
        return [(_add(w, merge(parts_soffit), mat_a), mat_a), (_add(w, merge(parts_face), mat_b), mat_b)]

    w.add_instance(w.add_model(arch_row(-3.4, mats[8], mats[9])))
    w.add_instance(w.add_model(arch_row(3.4, mats[8], mats[9])))

    # --- model 5: gallery floor slabs + balustrade (4 primitives), 2 instances (mirrored)
    slab = transform_mesh(grid(seg(96), seg(8), L - 1.0, 1.6, None, 10.0), translate((0, 4.4, -4.2)))
    slab_under = transform_mesh(grid(seg(96), seg(8), L - 1.0, 1.6, None, 10.0),
                                translate((0, 4.3, -4.2)) @ rotate_x(math.pi))
    rail = transform_mesh(box((-(L - 1.0) / 2, 0.0, -0.05), ((L - 1.0) / 2, 0.12, 0.05)), translate((0, 5.3, -3.45)))
    balusters = merge([transform_mesh(cylinder(0.05, 0.9, seg(10), 2), translate((x, 4.4, -3.45)))
                       for x in np.linspace(-(L - 1.4) / 2, (L - 1.4) / 2, max(8, int(60 * d)))])
    gallery = w.add_model([(_add(w, slab, mats[10]), mats[10]), (_add(w, slab_under, mats[11]), mats[11]),
                           (_add(w, rail, mats[12]), mats[12]), (_add(w, balusters, mats[13]), mats[13])])
    w.add_instance(gallery)
    w.add_instance(gallery, scale((1.0, 1.0, -1.0)))  # mirrored instance (negative determinant)

    # --- model 6: roof beams (2 primitives) x 6 instances
    beam = box((-0.15, 0.0, -W / 2), (0.15, 0.4, W / 2), uv_scale=4.0)
    bracket = merge([transform_mesh(box((-0.12, -0.5, -0.12), (0.12, 0.0, 0.12)), translate((0, 0, z)))
                     for z in (-W / 2 + 0.4, W / 2 - 0.4)])
    beam_model = w.add_model([(_add(w, beam, mats[14]), mats[14]), (_add(w, bracket, mats[15]), mats[15])])
    for x in np.linspace(-L / 2 + 2.0, L / 2 - 2.0, 6):
        w.add_instance(beam_model, translate((x, H - 0.45, 0.0)))

    # --- model 7: hanging cloth (wavy grids, 3 colours = 3 primitives), 3 instances
    def cloth(seed_phase):
        def waves(gx, gz):
            return 0.12 * np.sin(gx * 3.0 + seed_phase) + 0.05 * np.sin(gz * 5.0 + 2.0 * seed_phase)
        g = grid(seg(72), seg(72), 2.2, 3.0, waves, uv_scale=2.0)
        return transform_mesh(g, rotate_x(math.pi / 2))
    cloth_model = w.add_model([(_add(w, cloth(0.3), mats[16]), mats[16]), ])
    cloth_model2 = w.add_model([(_add(w, cloth(1.1), mats[17]), mats[17])])
    cloth_model3 = w.add_model([(_add(w, cloth(2.3), mats[18]), mats[18])])
    w.add_instance(cloth_model, translate((-6.0, 6.2, 0.0)) @ rotate_y(math.pi / 2))
    w.add_instance(cloth_model2, translate((0.0, 6.2, 0.0)) @ rotate_y(math.pi / 2))
    w.add_instance(cloth_model3, translate((6.0, 6.2, 0.0)) @ rotate_y(math.pi / 2))

    # --- model 10: statues/urns: displaced spheres-of-revolution (5 primitives), 4 instances
    def urn(profile_phase):
        segs, rings = seg(64), seg(48)
        th = np.linspace(0.0, 2 * math.pi, segs + 1)
        ys = np.linspace(0.0, 1.4, rings + 1)
        gt, gy = np.meshgrid(th, ys, indexing="xy")
        r = 0.18 + 0.22 * np.sin(gy / 1.4 * math.pi) ** 2 + 0.03 * np.sin(8 * gt + profile_phase) * np.sin(gy * 6.0)
        pos = np.stack([r * np.cos(gt), gy, r * np.sin(gt)], axis=-1).reshape(-1, 3)
        nrm = np.stack([np.cos(gt), np.zeros_like(gt), np.sin(gt)], axis=-1).reshape(-1, 3)
        tan = np.stack([-np.sin(gt), np.zeros_like(gt), np.cos(gt), np.ones_like(gt)], axis=-1).reshape(-1, 4)
        uv = np.stack([gt / (2 * math.pi) * 3.0, gy], axis=-1).reshape(-1, 2)
        i = (np.arange(rings)[:, None] * (segs + 1) + np.arange(segs)[None, :]).reshape(-1)
        idx = np.stack([i, i + segs + 1, i + 1, i + 1, i + segs + 1, i + segs + 2], axis=1).reshape(-1)
        return pos, nrm, tan, uv, idx.astype(np.uint32)

    plinth = box((-0.45, 0.0, -0.45), (0.45, 0.5, 0.45))
    urn_parts = [(_add(w, plinth, mats[19]), mats[19])]
    for k, ph in enumerate((0.0, 1.0, 2.0, 3.0)):
        urn_parts.append((_add(w, transform_mesh(urn(ph), translate((0, 0.5, 0))), mats[20 + k], buffer_index=1),
                          mats[20 + k]))
    # each urn instance uses one plinth + one of the four bodies => four 2-primitive models
    for k in range(4):
        m = w.add_model([urn_parts[0], urn_parts[1 + k]])
        x = (-9.0, -3.0, 3.0, 9.0)[k]
        w.add_instance(m, translate((x, 0.0, 0.0)) @ rotate_y(0.7 * k))

    # --- metallic lamp spheres (material 24), 1 model x 1 instance: chains of small urns
    lamp = w.add_model([(_add(w, transform_mesh(urn(0.5), scale((0.5, 0.4, 0.5))), mats[24], buffer_index=1), mats[24])])
    w.add_instance(lamp, translate((0.0, 7.8, 2.0)))

    if foliage:
        _add_foliage(w, 20000)
    if lights:
        # lights=True: the 512 + 512 of BASELINE C4; lights=(n_point, n_spot): the first n of the same sequences
        n_point, n_spot = (512, 512) if lights is True else lights
        _add_lights(w, n_point, n_spot, lo=(-L / 2 + 0.5, 0.5, -W / 2 + 0.5), hi=(L / 2 - 0.5, H - 0.5, W / 2 - 0.5))

    w.set_directional_light((1.0, 1.0, 1.0), 2.0, (-1.0, -1.0, -1.0)) if not lights else None
    if lights:
        # keep the sun too ("IBL flag on, sun (2,2,2) dir (-1,-1,-1)")
        w.set_directional_light((1.0, 1.0, 1.0), 2.0, (-1.0, -1.0, -1.0))
    w.skybox = sky_cube(sky_size)
    w.camera = dict(eye=(-10.5, 2.2, 0.6), target=(2.0, 3.2, -0.4), up=(0.0, 1.0, 0.0),
                    fov=math.radians(59.0), zN=0.1, zF=100.0)
    return w


def _add_foliage(w, count):
    """20k alpha quads: half MASK (leaf-shaped alpha, cutoff 0.5), half BLEND (alpha in [0.2, 0.9])."""
    mask_tex = w.add_texture(leaf_alpha_texture(128, 0x1EAF, blend=False))
    blend_tex = w.add_texture(leaf_alpha_texture(128, 0x2EAF, blend=True))
    smp = w.add_sampler(S.FILTER_LINEAR, S.FILTER_LINEAR, S.WRAP_CLAMP_TO_EDGE, S.WRAP_CLAMP_TO_EDGE)
    mask_mat = w.add_material(base_color=(1, 1, 1, 1), metallic=0.0, roughness=0.8, alpha_cutoff=0.5,
                              alpha_mode=S.ALPHA_MODE_MASK, base_tex=(mask_tex, smp))
    blend_mat = w.add_material(base_color=(1, 1, 1, 1), metallic=0.0, roughness=0.8,
                               alpha_mode=S.ALPHA_MODE_BLEND, base_tex=(blend_tex, smp))
    half = count // 2
    for kind, (mat, n) in enumerate(((mask_mat, half), (blend_mat, count - half))):
        r = _rand01(np.arange(n), 100 + kind, 0xF011A6E)
        r2 = _rand01(np.arange(n), 200 + kind, 0xF011A6E)
        # clustered around the urns / along the galleries
        cx = np.array([-9.0, -3.0, 3.0, 9.0])[(r[:, 0] * 4).astype(int) % 4]
        centre = np.stack([cx + (r[:, 1] - 0.5) * 2.4, 1.9 + r[:, 2] * 1.6, (r2[:, 0] - 0.5) * 2.4], axis=1)
        yaw = r2[:, 1] * 2 * math.pi
        pitch = (r2[:, 2] - 0.5) * 1.2
        size = 0.12 + 0.1 * r[:, 2]
        ux = np.stack([np.cos(yaw), np.zeros(n), np.sin(yaw)], axis=1)
        vy = np.stack([-np.sin(yaw) * np.sin(pitch), np.cos(pitch), np.cos(yaw) * np.sin(pitch)], axis=1)
        p0 = centre - ux * size[:, None] - vy * size[:, None]
        p1 = centre + ux * size[:, None] - vy * size[:, None]
        p2 = centre + ux * size[:, None] + vy * size[:, None]
        p3 = centre - ux * size[:, None] + vy * size[:, None]
        pos = np.stack([p0, p1, p2, p3], axis=1).reshape(-1, 3)
        nrm = np.repeat(np.cross(ux, vy), 4, axis=0)
        tan = np.repeat(np.concatenate([ux, np.ones((n, 1))], axis=1), 4, axis=0)
        uv = np.tile(np.array([[0, 0], [1, 0], [1, 1], [0, 1]], np.float64), (n, 1))
        base = (np.arange(n) * 4)[:, None]
        idx = (base + np.array([0, 1, 2, 0, 2, 3])[None, :]).reshape(-1).astype(np.uint32)
        mesh = w.add_mesh(pos, idx, mat, normals=nrm, tangents=tan, uvs=uv)
        w.add_instance(w.add_model([(mesh, mat)]))


def _add_lights(w, n_point, n_spot, lo, hi):
    """512 point + 512 spot lights placed by pcg3d(uvec3(i, kind, 0x00C0FFEE)) in the scene AABB,
    power 1-5 W => radius sqrt(lum/0.01) (WorldData.cpp:1478-1500)."""
    lo, hi = np.asarray(lo), np.asarray(hi)
    r = _rand01(np.arange(n_point), 0, 0x00C0FFEE)
    c = _rand01(np.arange(n_point), 2, 0x00C0FFEE)
    for i in range(n_point):
        pos = lo + r[i] * (hi - lo)
        color = 0.4 + 0.6 * c[i]
        w.add_point_light(color, 1.0 + 4.0 * c[i, 0], pos)
    r = _rand01(np.arange(n_spot), 1, 0x00C0FFEE)
    c = _rand01(np.arange(n_spot), 3, 0x00C0FFEE)
    dd = _rand01(np.arange(n_spot), 4, 0x00C0FFEE)
    for i in range(n_spot):
        pos = lo + r[i] * (hi - lo)
        direction = np.array([dd[i, 0] - 0.5, -0.3 - dd[i, 1], dd[i, 2] - 0.5])
        direction /= np.linalg.norm(direction)
        inner = math.radians(10.0 + 15.0 * c[i, 1])
        outer = inner + math.radians(10.0 + 15.0 * c[i, 2])
        w.add_spot_light(0.4 + 0.6 * c[i], 2.0 + 6.0 * c[i, 0], pos, direction, inner, outer)


def tiny_triangles():
    """A <=12-triangle scene with analytically known hits for the traversal-semantics tests."""
    w = World()
    opaque = w.add_material(base_color=(0.8, 0.8, 0.8, 1.0), metallic=0.0, roughness=1.0)
    blend = w.add_material(base_color=(1.0, 1.0, 1.0, 0.5), metallic=0.0, roughness=1.0, alpha_mode=S.ALPHA_MODE_BLEND)
    zero = w.add_material(base_color=(1.0, 1.0, 1.0, 0.0), metallic=0.0, roughness=1.0, alpha_mode=S.ALPHA_MODE_BLEND)
    # three stacked unit quads facing +z at z = 0 (opaque), z = 1 (blend), z = 2 (alpha 0: always ignored)
    q0 = _add(w, quad((-1, -1, 0), (1, -1, 0), (1, 1, 0), (-1, 1, 0)), opaque)
    q1 = _add(w, quad((-1, -1, 1), (1, -1, 1), (1, 1, 1), (-1, 1, 1)), blend)
    q2 = _add(w, quad((-1, -1, 2), (1, -1, 2), (1, 1, 2), (-1, 1, 2)), zero)
    # two coincident opaque quads at z = -1 in different draw instances (tie-break test)
    q3 = _add(w, quad((-1, -1, -1), (1, -1, -1), (1, 1, -1), (-1, 1, -1)), opaque)
    for q, m in ((q0, opaque), (q1, blend), (q2, zero), (q3, opaque), (q3, opaque)):
        w.add_instance(w.add_model([(q, m)]))
    w.camera = dict(eye=(0.0, 0.0, 5.0), target=(0.0, 0.0, 0.0), up=(0.0, 1.0, 0.0),
                    fov=math.radians(45.0), zN=0.1, zF=100.0)
    return w


def texture_wall():
    """Quads covering every wrap mode x filter with power-of-two and odd texture sizes and UVs far
    outside [0, 1] (including negative ones): the texel-addressing test scene."""
    w = World()
    rng = np.random.default_rng(0x7E47)
    sizes = [(8, 4), (5, 7), (13, 6), (1, 1), (16, 16), (3, 2)]
    textures = [w.add_texture(rng.integers(0, 256, size=(h, ww, 4), dtype=np.uint8)) for ww, h in sizes]
    wraps = [S.WRAP_REPEAT, S.WRAP_MIRRORED_REPEAT, S.WRAP_CLAMP_TO_EDGE]
    cells = []
    for flt in (S.FILTER_LINEAR, S.FILTER_NEAREST):
        for ws in wraps:
            for wt in wraps:
                cells.append(w.add_sampler(flt, flt, ws, wt))
    cols = 6
    for k, smp in enumerate(cells):
        tex = textures[k % len(textures)]
        mat = w.add_material(base_color=(1.0, 1.0, 1.0, 1.0), metallic=0.0, roughness=1.0, base_tex=(tex, smp),
                             mr_tex=(textures[(k + 1) % len(textures)], smp),
                             normal_tex=(textures[(k + 2) % len(textures)], smp))
        cx, cy = (k % cols) - cols / 2.0, (k // cols) - 1.5
        p, n, t, uv, idx = quad((cx, cy, 0.0), (cx + 0.95, cy, 0.0), (cx + 0.95, cy + 0.95, 0.0), (cx, cy + 0.95, 0.0))
        uv = uv * (5.3 + 0.37 * k) - (2.6 + 0.21 * k)  # spans several periods on both sides of 0
        q = _add(w, (p, n, t, uv, idx), mat)
        w.add_instance(w.add_model([(q, mat)]))
    w.add_point_light((1.0, 1.0, 1.0), 60.0, (0.0, 0.0, 3.0))
    w.camera = dict(eye=(0.0, 0.0, 6.5), target=(0.0, 0.0, 0.0), up=(0.0, 1.0, 0.0),
                    fov=math.radians(50.0), zN=0.1, zF=100.0)
    return w


def transform_zoo():
    """One normal-mapped, textured box instanced under rotations, non-uniform scales, a shear and mirrors
    (negative determinant): the inverse-transpose normal path, tangent signs and instance transforms
    (scene/instances.glsl:36-53, World.cpp:405-414)."""
    w = World()
    base = w.add_texture(noise_texture(32, 11, base=(0.6, 0.5, 0.4), amplitude=0.3, cells=6))
    nrm = w.add_texture(normal_texture(32, 12, strength=0.6, cells=5))
    mat = w.add_material(base_color=(1.0, 1.0, 1.0, 1.0), metallic=0.3, roughness=0.45, base_tex=(base, 0),
                         normal_tex=(nrm, 0))
    cube = _add(w, box(uv_scale=2.0), mat)
    model = w.add_model([(cube, mat)])
    shear = np.eye(4)
    shear[0, 1] = 0.6
    xs = [
        np.eye(4),
        rotate_y(0.7) @ rotate_x(0.4),
        scale((1.8, 0.4, 0.7)),
        rotate_z(0.5) @ scale((0.3, 1.6, 1.1)) @ rotate_y(1.1),
        scale((-1.0, 1.0, 1.0)),
        rotate_y(-0.6) @ scale((0.8, -1.3, 0.9)),
        shear,
        scale((-0.7, -0.9, -1.2)) @ rotate_x(0.9),
    ]
    for k, m in enumerate(xs):
        w.add_instance(model, translate(((k % 4) * 2.2 - 3.3, (k // 4) * 2.4 - 1.6, 0.0)) @ m)
    w.add_point_light((1.0, 0.95, 0.9), 400.0, (0.0, 0.5, 6.0))
    w.add_spot_light((0.8, 0.9, 1.0), 600.0, (-4.0, 3.0, 5.0), (0.6, -0.4, -0.7), math.radians(25.0), math.radians(45.0))
    w.camera = dict(eye=(0.0, 0.3, 9.0), target=(0.0, 0.2, 0.0), up=(0.0, 1.0, 0.0), fov=math.radians(50.0), zN=0.1, zF=100.0)
    return w


def alpha_wall():
    """MASK and BLEND quads in front of an opaque, lit wall: every wrap mode x filter, power-of-two and odd alpha
    textures (hard-edged blobs, smooth ramps, noise, all-0 and all-255 regions), UVs several periods either side of 0,
    baseColorFactor.a in {0, 0.5, 1, 1.7}, cutoffs in {0, 0.3, 0.5, 1}: the any-hit test scene (rt/scene.rahit:18-39,
    materials.glsl:121-147) - in particular for the alpha bounds that settle most candidates without a texel fetch."""
    w = World()
    rng = np.random.default_rng(0xA1FA)

    def alpha_texture(ww, h, kind):
        ys, xs = np.mgrid[0:h, 0:ww]
        u, v = (xs + 0.5) / ww, (ys + 0.5) / h
        if kind == 0:    # hard-edged blob: 0 / 255 with a one-texel rim of in-between values
            r = np.hypot(u - 0.5, v - 0.45)
            a = np.clip((0.36 - r) * max(ww, h) * 0.7 + 0.5, 0.0, 1.0) * 255.0
        elif kind == 1:  # smooth ramp with a zero band
            a = np.clip(u * 1.3 - 0.15, 0.0, 1.0) * np.where((v > 0.4) & (v < 0.55), 0.0, 1.0) * 255.0
        elif kind == 2:  # noise: nothing for the bounds to settle
            a = rng.integers(0, 256, size=(h, ww)).astype(np.float64)
        else:            # stripes of exact 0 / 255 / mid grey
            a = np.choose((xs * 3 // max(1, ww)) % 3, [0.0, 255.0, 128.0])
        t = rng.integers(0, 256, size=(h, ww, 4), dtype=np.uint8)
        t[..., 3] = np.round(a).astype(np.uint8)
        return t

    sizes = [(64, 64), (37, 21), (128, 32), (5, 3), (1, 1), (16, 48)]
    wraps = [S.WRAP_REPEAT, S.WRAP_MIRRORED_REPEAT, S.WRAP_CLAMP_TO_EDGE]
    samplers = [w.add_sampler(f, f, ws, wt) for f in (S.FILTER_LINEAR, S.FILTER_NEAREST) for ws in wraps for wt in wraps]
    back = w.add_material(base_color=(0.7, 0.7, 0.7, 1.0), metallic=0.0, roughness=0.9)
    wall = _add(w, quad((-4.2, -2.4, -0.6), (4.2, -2.4, -0.6), (4.2, 2.4, -0.6), (-4.2, 2.4, -0.6)), back)
    w.add_instance(w.add_model([(wall, back)]))
    cols, k = 8, 0
    factors, cutoffs = [1.0, 0.5, 1.7, 0.0, 1.0, 0.9], [0.5, 0.3, 1.0, 0.0, 0.5, 0.7]
    for row in range(5):
        for col in range(cols):
            ww, h = sizes[k % len(sizes)]
            tex = w.add_texture(alpha_texture(ww, h, k % 4))
            mode = S.ALPHA_MODE_MASK if (k // 2) % 2 == 0 else S.ALPHA_MODE_BLEND
            mat = w.add_material(base_color=(0.9, 0.6 + 0.05 * (k % 5), 0.3, factors[k % len(factors)]), metallic=0.0,
                                 roughness=0.7, alpha_cutoff=cutoffs[(k // 3) % len(cutoffs)], alpha_mode=mode,
                                 base_tex=(tex, samplers[k % len(samplers)]))
            cx, cy = col - cols / 2.0, row - 2.5
            z = 0.15 * (k % 3)  # three layers: shadow and bounce rays meet several candidates
            p, n, t, uv, idx = quad((cx, cy, z), (cx + 1.05, cy, z), (cx + 1.05, cy + 1.05, z), (cx, cy + 1.05, z))
            if k % 5 != 0:
                uv = uv * (1.9 + 0.23 * (k % 7)) - (0.7 + 0.11 * (k % 9))  # several periods, negative too
            q = _add(w, (p, n, t, uv, idx), mat)
            w.add_instance(w.add_model([(q, mat)]))
            k += 1
    w.add_point_light((1.0, 1.0, 1.0), 80.0, (0.5, 0.3, 3.0))
    w.add_spot_light((0.9, 0.9, 1.0), 200.0, (-2.0, 1.5, 3.5), (0.4, -0.3, -0.85), math.radians(30.0), math.radians(50.0))
    w.camera = dict(eye=(0.2, 0.1, 6.2), target=(0.0, 0.0, 0.0), up=(0.0, 1.0, 0.0), fov=math.radians(48.0), zN=0.1, zF=100.0)
    return w
