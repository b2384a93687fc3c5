"""Tone map, the step after the path: oracle.c ora_tone_map against an independent NumPy float64
evaluation of res/shader/tone_map.comp, the DDS reader on the reference's LUT (CPU container only), and - with
-m gpu - the HIP kernel against the oracle through the C-ABI, bit for bit."""
import os

import numpy as np
import pytest

from prosper_amd import dds, scenes, structs as S

LUT_PATH = "/root/reference/res/texture/tony_mc_mapface.dds"


def synthetic_lut(dim=48, seed=3):
    """A smooth, Tony-McMapface-like LUT (compressive, slightly desaturating) plus noise, encoded as R9G9B9E5."""
    rng = np.random.default_rng(seed)
    g = (np.arange(dim) + 0.0) / (dim - 1)
    b, gg, r = np.meshgrid(g, g, g, indexing="ij")  # z = blue, y = green, x = red
    enc = np.stack([r, gg, b], axis=-1)
    lum = enc @ np.array([0.2126, 0.7152, 0.0722])
    rgb = 0.85 * enc + 0.15 * lum[..., None] + rng.random(enc.shape) * 0.01
    return dds.encode_r9g9b9e5(rgb)


def reference_tone_map(hdr, lut_u32, exposure, contrast):
    """tone_map.comp:17-60 in float64 (no contract arithmetic: an independent evaluation)."""
    lut = dds.decode_r9g9b9e5(lut_u32).astype(np.float64)
    n = lut.shape[0]
    c = hdr[..., :3].astype(np.float16).astype(np.float64) * exposure
    v = c.max(-1)
    mn = c.min(-1)
    ch = v - mn
    r, g, b = c[..., 0], c[..., 1], c[..., 2]
    safe = np.where(ch == 0, 1.0, ch)
    hue = np.where(ch == 0, 0.0, np.where(v == r, np.mod((g - b) / safe, 6.0),
                                          np.where(v == g, (b - r) / safe + 2.0, (r - g) / safe + 4.0)))
    sat = np.where(v == 0, 0.0, ch / np.where(v == 0, 1.0, v))
    v = np.where(v > 0, np.power(np.maximum(v, 1e-300), contrast), 0.0)
    chroma = v * sat
    x = chroma * (1.0 - np.abs(np.mod(hue, 2.0) - 1.0))
    z = np.zeros_like(x)
    sel = [np.stack(t, -1) for t in ((chroma, x, z), (x, chroma, z), (z, chroma, x), (z, x, chroma), (x, z, chroma), (chroma, z, x))]
    k = np.clip(np.floor(hue).astype(int), 0, 5)
    rgb = np.choose(k[..., None], sel) + (v - chroma)[..., None]
    enc = rgb / (rgb + 1.0)
    uvw = enc * ((n - 1.0) / n) + 0.5 / n
    t = uvw * n - 0.5
    f = np.floor(t)
    a = t - f
    i0 = np.clip(f.astype(int), 0, n - 1)
    i1 = np.clip(f.astype(int) + 1, 0, n - 1)
    out = np.zeros_like(rgb)
    for dz, wz in ((0, 1 - a[..., 2]), (1, a[..., 2])):
        for dy, wy in ((0, 1 - a[..., 1]), (1, a[..., 1])):
            for dx, wx in ((0, 1 - a[..., 0]), (1, a[..., 0])):
                ix = (i1 if dx else i0)[..., 0]
                iy = (i1 if dy else i0)[..., 1]
                iz = (i1 if dz else i0)[..., 2]
                out += (wx * wy * wz)[..., None] * lut[iz, iy, ix]
    out = np.power(np.maximum(out, 0.0), 1.0 / 2.2)
    return np.clip(out, 0.0, 1.0) * 255.0


def test_r9g9b9e5_round_trip():
    rng = np.random.default_rng(0)
    rgb = rng.random((1000, 3)) * np.array([4.0, 0.5, 0.01])
    back = dds.decode_r9g9b9e5(dds.encode_r9g9b9e5(rgb))
    assert np.all(np.abs(back - rgb) <= rgb.max(axis=1, keepdims=True) / 256.0)
    assert np.array_equal(dds.decode_r9g9b9e5(np.array([0x780001FF], np.uint32))[0], [511.0 * 2.0 ** -9, 0.0, 0.0])


def _hdr_image(rng, h=64, w=96):
    hdr = np.empty((h, w, 4), np.float32)
    hdr[..., :3] = rng.gamma(0.7, 0.6, size=(h, w, 3)).astype(np.float32)
    hdr[: h // 8] = 0.0                      # black rows: value == 0, chroma == 0
    hdr[h // 8: h // 4, :, 1] = hdr[h // 8: h // 4, :, 0]  # two equal channels (hue branches on ==)
    hdr[-4:, :, :3] *= 500.0                 # far beyond the LUT's knee
    hdr[..., 3] = 8.0
    return hdr


@pytest.mark.parametrize("exposure,contrast", [(1.0, 1.0), (2.5, 1.3), (0.3, 0.8)])
def test_oracle_tone_map_matches_float64_evaluation(oracle, exposure, contrast):
    rng = np.random.default_rng(5)
    hdr = _hdr_image(rng)
    lut = synthetic_lut()
    got = oracle.tone_map(hdr, lut, exposure, contrast)
    want = reference_tone_map(hdr, lut, exposure, contrast)
    assert (got[..., 3] == 255).all()
    # fp32 contract arithmetic vs float64: at most one code value apart, and rarely
    diff = np.abs(got[..., :3].astype(np.float64) - np.rint(want))
    assert diff.max() <= 1.0 and (diff > 0).mean() < 0.02


@pytest.mark.skipif(not os.path.exists(LUT_PATH), reason="the reference's LUT is only mounted in the CPU container")
def test_reads_the_reference_lut(oracle):
    lut = dds.read_lut(LUT_PATH)
    assert lut.shape == (48, 48, 48)
    rgb = dds.decode_r9g9b9e5(lut)
    assert rgb[0, 0, 0].max() < 0.02 and 0.9 < rgb[-1, -1, -1].min() <= 1.01  # black stays black, white ~ white
    grey = rgb[np.arange(48), np.arange(48), np.arange(48)]
    assert (np.diff(grey[:, 1]) > -1e-3).all()  # the neutral axis is monotone
    ramp = np.zeros((1, 256, 4), np.float32)
    ramp[0, :, :3] = (np.linspace(0.0, 8.0, 256, dtype=np.float32) ** 2)[:, None]
    out = oracle.tone_map(ramp, lut, 1.0, 1.0)[0, :, 1].astype(int)
    assert out[0] < 10 and out[-1] >= 250 and (np.diff(out) >= -1).all()  # small toe at black; 9-bit mantissas wiggle by one code


@pytest.mark.gpu
@pytest.mark.parametrize("exposure,contrast", [(1.0, 1.0), (2.5, 1.3)])
def test_gpu_tone_map_bit_exact(gpu_ctx, oracle, cornell_world, exposure, contrast):
    from conftest import default_pc
    w, h = 256, 160
    c = cornell_world.camera
    cam, fl = oracle.camera_uniforms(c["eye"], c["target"], c["up"], c["fov"], c["zN"], c["zF"], w, h)
    gpu_ctx.upload_scene(cornell_world)
    gpu_ctx.render(default_pc(S, fl, max_bounces=4), cam, w, h, frames=4)
    hdr = gpu_ctx.read_hdr()
    lut = synthetic_lut()
    gpu_ctx.set_tone_map_lut(lut)
    got = gpu_ctx.tone_map(exposure, contrast)
    want = oracle.tone_map(hdr, lut, exposure, contrast)
    assert got.shape == want.shape == (h, w, 4)
    assert np.array_equal(got, want), "%d texels differ" % (got != want).any(axis=2).sum()
    assert len(np.unique(got[..., :3].reshape(-1, 3), axis=0)) > 500


@pytest.mark.gpu
def test_gpu_tone_map_edge_values_and_errors(gpu_ctx, oracle, cornell_world):
    """Synthetic HDR through a caller-owned buffer: zeros, equal channels, huge values, NaN/inf; small odd LUT."""
    import torch
    from prosper_amd import capi
    rng = np.random.default_rng(11)
    hdr = _hdr_image(rng, 40, 56)
    hdr[5, 5, 0] = np.nan
    hdr[6, 6, 1] = np.inf
    ctx = capi.Context(0)
    try:
        with pytest.raises(capi.ProsperPtError):
            ctx.tone_map()  # nothing rendered
        t = torch.from_numpy(hdr).cuda()
        ctx.upload_scene(cornell_world)
        ctx.set_output_buffer(t.data_ptr(), t.numel() * 4)
        # a zero-sample render leaves the buffer untouched: use the smallest real render, then overwrite it
        c = cornell_world.camera
        cam, fl = oracle.camera_uniforms(c["eye"], c["target"], c["up"], c["fov"], c["zN"], c["zF"], 56, 40)
        from conftest import default_pc
        ctx.render(default_pc(S, fl, max_bounces=1), cam, 56, 40)
        torch.cuda.synchronize()
        t.copy_(torch.from_numpy(hdr))
        with pytest.raises(capi.ProsperPtError):
            ctx.tone_map()  # no LUT yet
        for dim in (48, 7):
            lut = synthetic_lut(dim)
            ctx.set_tone_map_lut(lut)
            got = ctx.tone_map(1.7, 1.1)
            assert np.array_equal(got, oracle.tone_map(hdr, lut, 1.7, 1.1)), dim
    finally:
        ctx.close()


def test_dds_lut_write_read_round_trip(tmp_path):
    lut = synthetic_lut(12)
    p = tmp_path / "lut.dds"
    dds.write_lut(str(p), lut)
    assert np.array_equal(dds.read_lut(str(p)), lut)
    if os.path.exists(LUT_PATH):  # same header fields as the reference's file
        assert open(LUT_PATH, "rb").read(148)[76:92] == p.read_bytes()[76:92]


@pytest.mark.gpu
def test_host_tone_map_class(gpu_ctx, oracle, cornell_world, tmp_path):
    """render::ToneMap (C++ host layer): init from a DDS file == init from texels == the C-ABI call; drawUi clamps."""
    import torch
    from conftest import default_pc
    from prosper_amd import capi
    from prosper_amd.rt_reference import ToneMap
    w, h = 128, 96
    c = cornell_world.camera
    cam, fl = oracle.camera_uniforms(c["eye"], c["target"], c["up"], c["fov"], c["zN"], c["zF"], w, h)
    gpu_ctx.upload_scene(cornell_world)
    gpu_ctx.render(default_pc(S, fl, max_bounces=3), cam, w, h, frames=2)
    hdr = gpu_ctx.read_hdr()
    lut = synthetic_lut(16)
    path = tmp_path / "lut.dds"
    dds.write_lut(str(path), lut)
    out = torch.zeros((h, w), dtype=torch.int32, device="cuda")
    for tm in (ToneMap(gpu_ctx, lut_path=str(path)), ToneMap(gpu_ctx, lut_texels=lut)):
        tm.draw_ui(2.0, 1.2)
        out.zero_()
        tm.record(out.data_ptr(), out.numel() * 4)
        torch.cuda.synchronize()
        got = out.cpu().numpy().view(np.uint8).reshape(h, w, 4)
        assert np.array_equal(got, oracle.tone_map(hdr, lut, 2.0, 1.2))
        tm.draw_ui(1e9, 0.0)  # sliders clamp to [0.001, 10000] (ToneMap.cpp:58-59)
        tm.record(out.data_ptr(), out.numel() * 4)
        torch.cuda.synchronize()
        got = out.cpu().numpy().view(np.uint8).reshape(h, w, 4)
        assert np.array_equal(got, oracle.tone_map(hdr, lut, 10000.0, 0.001))
        tm.close()
    with pytest.raises(capi.ProsperPtError):
        ToneMap(gpu_ctx, lut_path=str(tmp_path / "missing.dds"))
    (tmp_path / "bad.dds").write_bytes(b"DDS " + bytes(200))
    with pytest.raises(capi.ProsperPtError):
        ToneMap(gpu_ctx, lut_path=str(tmp_path / "bad.dds"))
