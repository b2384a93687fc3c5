"""Pins the oracle: hand-derived known answers + the independent NumPy
evaluation committed in tests/golden/kat.npz (generator: tests/golden/make_kat.py)."""
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def kat():
    return np.load(os.path.join(HERE, "golden", "kat.npz"))


def f32(*cols):
    return np.concatenate([np.asarray(c, np.float32).reshape(len(c), -1) for c in cols], axis=1)


def test_pcg_hand_derived_answers(oracle):
    # SURVEY Appendix A, derived from the formula text of common/random.glsl:7-28
    L = oracle.lib()
    assert L.ora_pcg(0) == 0x07BB2FE2
    assert L.ora_pcg(1) == 0xA8BEEA3C
    assert L.ora_pcg(0xFFFFFFFF) == 0xE62A4902
    assert L.ora_pcg(12345) == 0xF45EAD0E
    import ctypes as C
    for v, want in (((0, 0, 1), (0x7F5DB8D2, 0x68E0A5AC, 0x3B10C274)), ((1, 2, 3), (0xFA9F79A6, 0x48F2F44C, 0x596F5AB1)),
                    ((1919, 1079, 8), (0x53B41AB2, 0xD68ED4E7, 0xAC595768))):
        a = (C.c_uint32 * 3)(*v)
        L.ora_pcg3d(a)
        assert tuple(a) == want


def test_pcg_and_pcg3d_match_numpy(oracle, kat):
    import ctypes as C
    L = oracle.lib()
    for i, o in zip(kat["pcg_in"], kat["pcg_out"]):
        assert L.ora_pcg(int(i)) == int(o)
    for i, o in zip(kat["pcg3d_in"], kat["pcg3d_out"]):
        a = (C.c_uint32 * 3)(*[int(x) for x in i])
        L.ora_pcg3d(a)
        assert tuple(a) == tuple(int(x) for x in o)


def test_packing_hand_derived_answers(oracle):
    import ctypes as C
    L = oracle.lib()
    assert L.ora_pack_half(1.0) == 0x3C00 and L.ora_pack_half(0.1) == 0x2E66 and L.ora_pack_half(-2.5) == 0xC100
    assert L.ora_pack_snorm3x10_1x2((C.c_float * 4)(0, 0, 1, 0)) == 0x1FF00000
    assert L.ora_pack_snorm3x10_1x2((C.c_float * 4)(-1.0, 256 / 511.0, 0.0, -1.0)) == 0xC0040201


def test_half_conversion_exhaustive(oracle, kat):
    L = oracle.lib()
    # every binary16 value decodes like numpy and round-trips
    allh = np.arange(65536, dtype=np.uint16)
    want = allh.view(np.float16).astype(np.float32)
    got = np.array([L.ora_unpack_half(int(h)) for h in allh], np.float32)
    assert ((got.view(np.uint32) == want.view(np.uint32)) | (np.isnan(got) & np.isnan(want))).all()
    for x, bits in zip(kat["half_in"], kat["half_bits"]):
        assert L.ora_pack_half(float(x)) == int(bits)
    rng = np.random.default_rng(5)
    xs = np.concatenate([rng.standard_normal(4000) * 1000, rng.standard_normal(4000) * 1e-5, [65519.9, 65520.0, 5.96e-8, 2.98e-8, 2.99e-8]]).astype(np.float32)
    with np.errstate(over="ignore"):  # 65520.0 -> inf is the point of that sample
        want16 = xs.astype(np.float16).view(np.uint16)
    for x, w16 in zip(xs, want16):
        assert L.ora_pack_half(float(x)) == int(w16), float(x)


def test_snorm_unpack(oracle, kat):
    out = oracle.eval_fn("UNPACK_SNORM", kat["snorm_bits"].view(np.float32))
    np.testing.assert_allclose(out[:, :3], kat["snorm_xyz"], rtol=0, atol=2e-7)
    assert (out[:, 3] == kat["snorm_w"]).all()


def test_transcendental_kernels(oracle, kat):
    sc = oracle.eval_fn("SINCOS", kat["sincos_in"])
    np.testing.assert_allclose(sc[:, 0], kat["sin_out"], rtol=0, atol=3e-7)
    np.testing.assert_allclose(sc[:, 1], kat["cos_out2"], rtol=0, atol=3e-7)
    pw = oracle.eval_fn("POW", f32(kat["pow_b"], kat["pow_e"]))[:, 0]
    np.testing.assert_allclose(pw, kat["pow_out"], rtol=4e-6)
    sr = oracle.eval_fn("SRGB_TO_LINEAR", kat["srgb_in"])[:, 0]
    np.testing.assert_allclose(sr, kat["srgb_out"], rtol=3e-6, atol=1e-9)


def test_sampling_functions(oracle, kat):
    onb = oracle.eval_fn("ONB", kat["onb_in"])
    np.testing.assert_allclose(onb[:, 0:3], kat["onb_b1"], atol=2e-6)
    np.testing.assert_allclose(onb[:, 3:6], kat["onb_b2"], atol=2e-6)
    cs = oracle.eval_fn("COSINE_SAMPLE", f32(kat["cos_n"], kat["cos_u"]))
    np.testing.assert_allclose(cs, kat["cos_out"], atol=3e-6)
    vs = oracle.eval_fn("VNDF_SAMPLE", f32(kat["vndf_ve"], kat["vndf_alpha"], kat["vndf_u"]))
    np.testing.assert_allclose(vs, kat["vndf_out"], atol=2e-5)
    pdf = oracle.eval_fn("VNDF_PDF", f32(kat["vndf_ve"], kat["pdf_le"], kat["vndf_alpha"]))[:, 0]
    np.testing.assert_allclose(pdf, kat["pdf_out"], rtol=2e-4)


def test_brdf_lights_offset(oracle, kat):
    brdf = oracle.eval_fn("EVAL_BRDF", f32(kat["brdf_l"], kat["brdf_n"], kat["brdf_v"], kat["brdf_albedo"],
                                           kat["brdf_rough"], kat["brdf_metal"]))
    np.testing.assert_allclose(brdf, kat["brdf_out"], rtol=3e-4, atol=1e-6)
    off = oracle.eval_fn("OFFSET_RAY", f32(kat["off_p"], kat["off_n"]))
    assert (off.view(np.uint32) == kat["off_out"].astype(np.float32).view(np.uint32)).all()
    pl = oracle.eval_fn("POINT_LIGHT", f32(kat["pl_pos"], kat["pl_rad"], kat["pl_radius"], kat["pl_surf"]))
    np.testing.assert_allclose(pl[:, 0:3], kat["pl_l"], atol=2e-6)
    np.testing.assert_allclose(pl[:, 3], kat["pl_d"], rtol=2e-6)
    np.testing.assert_allclose(pl[:, 4:7], kat["pl_irr"], rtol=2e-5, atol=1e-7)
    sl = oracle.eval_fn("SPOT_LIGHT", f32(kat["pl_pos"], kat["sl_off"], kat["pl_rad"], kat["sl_scale"], kat["sl_dir"],
                                          kat["pl_surf"]))
    np.testing.assert_allclose(sl[:, 0:3], kat["sl_l"], atol=2e-6)
    np.testing.assert_allclose(sl[:, 4:7], kat["sl_irr"], rtol=1e-4, atol=1e-6)


def test_triangle_intersection_agrees_with_moller_trumbore(oracle, kat):
    n = len(kat["tri_o"])
    x = f32(kat["tri_o"], kat["tri_d"], kat["tri_v0"], kat["tri_v1"], kat["tri_v2"], np.zeros(n), np.full(n, np.inf))
    out = oracle.eval_fn("TRIANGLE", x)
    clear = np.abs(kat["tri_margin"]) > 1e-4  # away from edges both tests must agree
    assert ((out[:, 0] > 0) == kat["tri_hit"])[clear].all()
    hit = kat["tri_hit"] & clear
    assert hit.sum() > 10
    np.testing.assert_allclose(out[hit, 1], kat["tri_t"][hit], rtol=2e-4)
    np.testing.assert_allclose(out[hit, 2], kat["tri_u"][hit], atol=2e-4)
    np.testing.assert_allclose(out[hit, 3], kat["tri_v"][hit], atol=2e-4)


def test_rng_stream_matches_pcg3d(oracle, kat):
    v = kat["pcg3d_in"][:64]
    out = oracle.eval_fn("RNG", v.view(np.float32))
    s1 = kat["pcg3d_out"][:64].astype(np.uint64)
    np.testing.assert_array_equal(out[:, 0], (s1[:, 0].astype(np.float32) / np.float32(4294967296.0)))
    assert (out[:, 0] <= 1.0).all() and (out[:, 0] >= 0.0).all()
