#!/usr/bin/env python3
"""Generates tests/golden/kat.npz: known-answer vectors for the pure functions of the path.

The reference ships no tests, golden vectors or fixtures for this path and
cannot run here, so these vectors are an INDEPENDENT NumPy evaluation (float64 / uint32) of the
formulas in the reference's shader text and in the sources the shaders cite (Jarzynski-Olano PCG,
Heitz VNDF, Duff ONB, Wächter-Binder), seeded and committed.  They pin the oracle; the oracle
then pins the HIP kernels bit for bit.

    python tests/golden/make_kat.py      # rewrites tests/golden/kat.npz
"""
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
M32 = np.uint64(0xFFFFFFFF)


def u32(x):
    return np.asarray(x, dtype=np.uint64) & M32


def pcg(v):
    """res/shader/common/random.glsl:7-12"""
    v = u32(v)
    state = u32(v * np.uint64(747796405) + np.uint64(2891336453))
    word = u32(((state >> ((state >> np.uint64(28)) + np.uint64(4))) ^ state) * np.uint64(277803737))
    return u32((word >> np.uint64(22)) ^ word)


def pcg3d(v):
    """res/shader/common/random.glsl:17-28"""
    v = u32(v).copy()
    v = u32(v * np.uint64(1664525) + np.uint64(1013904223))
    x, y, z = v[..., 0].copy(), v[..., 1].copy(), v[..., 2].copy()
    x = u32(x + y * z)
    y = u32(y + z * x)
    z = u32(z + x * y)
    x ^= x >> np.uint64(16)
    y ^= y >> np.uint64(16)
    z ^= z >> np.uint64(16)
    x = u32(x + y * z)
    y = u32(y + z * x)
    z = u32(z + x * y)
    return np.stack([x, y, z], axis=-1)


def unit(v):
    return v / np.linalg.norm(v, axis=-1, keepdims=True)


def onb(n):
    """res/shader/common/sampling.glsl:37-47 -> rows b1, b2, n"""
    s = np.sign(n[:, 2])
    a = -1.0 / (s + n[:, 2])
    b = n[:, 0] * n[:, 1] * a
    b1 = np.stack([1.0 + s * n[:, 0] * n[:, 0] * a, s * b, -s * n[:, 0]], axis=1)
    b2 = np.stack([b, s + n[:, 1] * n[:, 1] * a, -n[:, 1]], axis=1)
    return b1, b2, n


PI = 3.14159265


def cosine_sample(n, u):
    """sampling.glsl:18-33"""
    a = (1.0 - 2.0 * u[:, 0]) * 0.99999
    b = np.sqrt(1.0 - a * a) * 0.99999
    phi = 2.0 * PI * u[:, 1]
    return unit(n + np.stack([b * np.cos(phi), b * np.sin(phi), a], axis=1))


def vndf_sample(ve, alpha, u):
    """sampling.glsl:53-79"""
    vh = unit(np.stack([alpha * ve[:, 0], alpha * ve[:, 1], ve[:, 2]], axis=1))
    lensq = vh[:, 0] ** 2 + vh[:, 1] ** 2
    t1v = np.where(lensq[:, None] > 0,
                   np.stack([-vh[:, 1], vh[:, 0], np.zeros_like(lensq)], axis=1) / np.sqrt(np.maximum(lensq, 1e-300))[:, None],
                   np.array([1.0, 0.0, 0.0]))
    t2v = np.cross(vh, t1v)
    r = np.sqrt(u[:, 0])
    phi = 2.0 * PI * u[:, 1]
    t1 = r * np.cos(phi)
    t2 = r * np.sin(phi)
    s = 0.5 * (1.0 + vh[:, 2])
    t2 = (1.0 - s) * np.sqrt(1.0 - t1 * t1) + s * t2
    nh = t1[:, None] * t1v + t2[:, None] * t2v + np.sqrt(np.maximum(0.0, 1.0 - t1 * t1 - t2 * t2))[:, None] * vh
    ne = unit(np.stack([alpha * nh[:, 0], alpha * nh[:, 1], np.maximum(0.0, nh[:, 2])], axis=1))
    i = -ve
    return i - 2.0 * np.sum(ne * i, axis=1, keepdims=True) * ne


def trowbridge_reitz(noh, alpha):
    a2 = alpha * alpha
    denom = noh * noh * (a2 - 1.0) + 1.0
    return a2 / (PI * denom * denom)


def schlick_tr(nol, nov, alpha):
    k = np.maximum(alpha * 0.5, 0.0001)
    return (nol / (nol * (1.0 - k) + k)) * (nov / (nov * (1.0 - k) + k))


def vndf_pdf(ve, le, alpha):
    """sampling.glsl:81-93"""
    ne = unit(ve + le)
    sat = lambda x: np.clip(x, 0.0, 1.0)
    nov, nol, noh = sat(ve[:, 2]), sat(le[:, 2]), sat(ne[:, 2])
    vndf = schlick_tr(nol, nov, alpha) * nov * trowbridge_reitz(noh, alpha) / ve[:, 2]
    return vndf / (4 * nov)


def eval_brdf(l, n, v, albedo, rough, metal):
    """brdf.glsl:67-87"""
    sat = lambda x: np.clip(x, 0.0, 1.0)
    h = unit(v + l)
    nol = sat(np.sum(n * l, axis=1))
    noh = sat(np.sum(n * h, axis=1))
    voh = sat(np.sum(v * h, axis=1))
    nov = sat(np.sum(n * v, axis=1))
    f0 = 0.04 * (1 - metal)[:, None] + albedo * metal[:, None]
    cdiff = albedo * 0.96 * (1 - metal)[:, None]
    alpha = rough * rough
    d = trowbridge_reitz(noh, alpha)
    f = f0 + (1.0 - f0) * ((1.0 - voh) ** 5)[:, None]
    g = schlick_tr(nol, nov, alpha)
    spec = d[:, None] * f * g[:, None] / (4.0 * nol * nov + 0.0001)[:, None]
    return (cdiff / PI + spec) * nol[:, None]


def offset_ray(p, n):
    """rt/ray.glsl:83-103 (float32 / int32 bit arithmetic)"""
    p = p.astype(np.float32)
    n = n.astype(np.float32)
    of_i = (np.float32(256.0) * n).astype(np.int32)  # truncation
    bits = p.view(np.int32)
    moved = (bits.astype(np.int64) + np.where(p < 0, -of_i, of_i).astype(np.int64))
    moved = ((moved + 2**31) % 2**32 - 2**31).astype(np.int32)
    p_i = moved.view(np.float32)
    near = np.abs(p) < np.float32(1.0 / 32.0)
    return np.where(near, p + np.float32(1.0 / 65536.0) * n, p_i)


def srgb_to_linear(x):
    """scene/materials.glsl:26-29"""
    return np.where(x <= 0.04045, x / 12.92, ((x + 0.055) / 1.055) ** 2.4)


def unpack_snorm(bits):
    """scene/geometry.glsl:95-103,116-127"""
    b = bits.astype(np.int64)

    def field(shift):
        v = (b >> shift) & 0x3FF
        return np.where(v >= 512, v - 1024, v).astype(np.float64)

    v = np.stack([field(0), field(10), field(20)], axis=1)
    v = np.maximum(v / 511.0, -1.0)
    w = (b >> 30) & 0x3
    w = np.where(w >= 2, w - 4, w).astype(np.float64)
    return unit(v), w


def point_light(pos, radiance, radius, surf):
    """scene/lighting.glsl:15-37"""
    to = pos - surf
    d2 = np.sum(to * to, axis=1)
    d = np.sqrt(d2)
    l = to / d[:, None]
    att = np.clip(1.0 - (d / radius) ** 4, 0.0, 1.0)
    return l, d, radiance * att[:, None] / d2[:, None]


def spot_light(pos, offset, rad, scale, direction, surf):
    """scene/lighting.glsl:39-56"""
    to = pos - surf
    d2 = np.sum(to * to, axis=1)
    d = np.sqrt(d2)
    l = to / d[:, None]
    cd = np.sum(-direction * l, axis=1)
    att = np.clip(cd * scale + offset, 0.0, 1.0) ** 2
    return l, d, att[:, None] * rad / d2[:, None]


def moller_trumbore(o, d, v0, v1, v2):
    """Independent ray/triangle reference (float64): returns hit, t, u (v1 weight), v (v2 weight), margin."""
    e1, e2 = v1 - v0, v2 - v0
    p = np.cross(d, e2)
    det = np.sum(e1 * p, axis=1)
    inv = 1.0 / np.where(det == 0, 1.0, det)
    s = o - v0
    u = np.sum(s * p, axis=1) * inv
    q = np.cross(s, e1)
    v = np.sum(d * q, axis=1) * inv
    t = np.sum(e2 * q, axis=1) * inv
    hit = (det != 0) & (u >= 0) & (v >= 0) & (u + v <= 1) & (t > 0)
    margin = np.minimum(np.minimum(u, v), 1.0 - u - v)  # distance from the nearest edge, barycentric
    return hit, t, u, v, margin


def main():
    rng = np.random.default_rng(0x9E3779B9)
    n = 256
    out = {}

    # integers
    seeds = np.concatenate([np.array([0, 1, 0xFFFFFFFF, 12345], dtype=np.uint64), rng.integers(0, 2**32, n, dtype=np.uint64)])
    out["pcg_in"] = seeds.astype(np.uint32)
    out["pcg_out"] = pcg(seeds).astype(np.uint32)
    v3 = np.concatenate([np.array([[0, 0, 1], [1, 2, 3], [1919, 1079, 8]], dtype=np.uint64),
                         rng.integers(0, 2**32, (n, 3), dtype=np.uint64)])
    out["pcg3d_in"] = v3.astype(np.uint32)
    out["pcg3d_out"] = pcg3d(v3).astype(np.uint32)

    # fp16 / snorm packing
    h_in = np.concatenate([np.array([1.0, 0.1, -2.5, 0.0, 65504.0, 1e-8, 6.1e-5], np.float32),
                           (rng.standard_normal(n) * 10).astype(np.float32)])
    out["half_in"] = h_in
    out["half_bits"] = h_in.astype(np.float16).view(np.uint16)
    bits = rng.integers(0, 2**32, n, dtype=np.uint64).astype(np.uint32)
    out["snorm_bits"] = bits
    sv, sw = unpack_snorm(bits)
    out["snorm_xyz"], out["snorm_w"] = sv, sw

    u2 = rng.random((n, 2))
    nrm = unit(rng.standard_normal((n, 3)))
    out["onb_in"] = nrm
    out["onb_b1"], out["onb_b2"], _ = onb(nrm)
    out["cos_n"], out["cos_u"] = nrm, u2
    out["cos_out"] = cosine_sample(nrm, u2)

    ve = unit(rng.standard_normal((n, 3)))
    ve[:, 2] = np.abs(ve[:, 2]) * 0.9 + 0.1
    ve = unit(ve)
    alpha = rng.random(n) * 0.9 + 0.01
    out["vndf_ve"], out["vndf_alpha"], out["vndf_u"] = ve, alpha, u2
    out["vndf_out"] = vndf_sample(ve, alpha, u2)
    le = unit(rng.standard_normal((n, 3)))
    le[:, 2] = np.abs(le[:, 2]) * 0.9 + 0.1
    le = unit(le)
    out["pdf_le"] = le
    out["pdf_out"] = vndf_pdf(ve, le, alpha)

    l = unit(nrm + 0.7 * rng.standard_normal((n, 3)))
    v = unit(nrm + 0.7 * rng.standard_normal((n, 3)))
    albedo, rough, metal = rng.random((n, 3)), rng.random(n) * 0.95 + 0.05, rng.random(n)
    out["brdf_l"], out["brdf_n"], out["brdf_v"] = l, nrm, v
    out["brdf_albedo"], out["brdf_rough"], out["brdf_metal"] = albedo, rough, metal
    out["brdf_out"] = eval_brdf(l, nrm, v, albedo, rough, metal)

    p = (rng.standard_normal((n, 3)) * 4).astype(np.float32)
    p[: n // 4] *= np.float32(0.003)
    nn = unit(rng.standard_normal((n, 3))).astype(np.float32)
    out["off_p"], out["off_n"] = p, nn
    out["off_out"] = offset_ray(p, nn)

    x = np.concatenate([np.arange(256) / 255.0, rng.random(n)])
    out["srgb_in"], out["srgb_out"] = x, srgb_to_linear(x)
    ang = np.concatenate([rng.random(n) * 2 * np.pi, rng.standard_normal(n) * 40])
    out["sincos_in"], out["sin_out"], out["cos_out2"] = ang, np.sin(ang.astype(np.float32).astype(np.float64)), np.cos(ang.astype(np.float32).astype(np.float64))
    pb, pe = rng.random(n) * 3 + 0.01, rng.standard_normal(n) * 2
    out["pow_b"], out["pow_e"] = pb, pe
    out["pow_out"] = pb.astype(np.float32).astype(np.float64) ** pe.astype(np.float32).astype(np.float64)

    lp, ls = rng.standard_normal((n, 3)) * 3, rng.standard_normal((n, 3)) * 3
    lr, lrad = rng.random((n, 3)) * 3, rng.random(n) * 15 + 1
    out["pl_pos"], out["pl_surf"], out["pl_rad"], out["pl_radius"] = lp, ls, lr, lrad
    out["pl_l"], out["pl_d"], out["pl_irr"] = point_light(lp, lr, lrad, ls)
    so, sscale, sdir = rng.standard_normal(n) * 0.5, rng.random(n) * 6, unit(rng.standard_normal((n, 3)))
    out["sl_off"], out["sl_scale"], out["sl_dir"] = so, sscale, sdir
    out["sl_l"], out["sl_d"], out["sl_irr"] = spot_light(lp, so, lr, sscale, sdir, ls)

    o = rng.standard_normal((n, 3)) * 3
    tgt = rng.standard_normal((n, 3))
    d = unit(tgt - o)
    v0, v1, v2 = tgt + rng.standard_normal((n, 3)), tgt + rng.standard_normal((n, 3)), tgt + rng.standard_normal((n, 3))
    out["tri_o"], out["tri_d"], out["tri_v0"], out["tri_v1"], out["tri_v2"] = o, d, v0, v1, v2
    hit, t, bu, bv, margin = moller_trumbore(o, d, v0, v1, v2)
    out["tri_hit"], out["tri_t"], out["tri_u"], out["tri_v"], out["tri_margin"] = hit, t, bu, bv, margin

    np.savez_compressed(os.path.join(HERE, "kat.npz"), **out)
    print("wrote kat.npz with %d arrays" % len(out))


if __name__ == "__main__":
    main()
