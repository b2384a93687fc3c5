#!/usr/bin/env python3
"""glTF/GLB -> path-traced PNG on one MI355X, end to end through the C-ABI:

    python scripts/render_gltf.py scene.gltf --spp 64 --size 1280x720 --out out.png \\
        [--lut res/texture/tony_mc_mapface.dds] [--env env/sky.ktx] [--eye x,y,z --target x,y,z]

glTF ingest (prosper_amd/gltf.py) -> prosper_pt_upload_scene -> prosper_pt_render_frames -> prosper_pt_tone_map
(Tony McMapface LUT when given, otherwise an identity LUT, i.e. plain x/(x+1) + gamma) -> PNG.
"""
import argparse
import math
import os
import struct
import sys
import zlib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402


def write_png(path, rgba):
    h, w, _ = rgba.shape
    raw = b"".join(b"\x00" + rgba[y].tobytes() for y in range(h))

    def chunk(kind, data):
        return struct.pack(">I", len(data)) + kind + data + struct.pack(">I", zlib.crc32(kind + data) & 0xFFFFFFFF)
    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 6, 0, 0, 0)) +
                chunk(b"IDAT", zlib.compress(raw, 6)) + chunk(b"IEND", b""))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("gltf")
    ap.add_argument("--out", default="out.png")
    ap.add_argument("--spp", type=int, default=64)
    ap.add_argument("--size", default="1280x720")
    ap.add_argument("--bounces", type=int, default=4)
    ap.add_argument("--lut", default=None, help="tony_mc_mapface.dds (3-D R9G9B9E5 DDS)")
    ap.add_argument("--env", default=None, help="RGBA16F cube map (.ktx, KTX 1.1): enables IBL")
    ap.add_argument("--eye", default=None)
    ap.add_argument("--target", default=None)
    ap.add_argument("--exposure", type=float, default=1.0)
    args = ap.parse_args()
    from prosper_amd import capi, dds, gltf, ktx, structs as S
    from prosper_amd.rt_reference import Camera
    w, h = (int(v) for v in args.size.lower().split("x"))
    world = gltf.load_gltf(args.gltf, bc7_on_gpu=True)  # prosper_cache BC7 files are decoded by the library at upload
    if world.missing_images:
        print("missing images replaced by white: %s" % ", ".join(world.missing_images), file=sys.stderr)
    if args.env:
        world.skybox = ktx.read_cube(args.env)
    if args.eye:
        world.camera["eye"] = tuple(float(v) for v in args.eye.split(","))
    if args.target:
        world.camera["target"] = tuple(float(v) for v in args.target.split(","))
    cam, focal = Camera.from_world(world, w, h).update_buffer()
    ctx = capi.Context(0)
    ctx.upload_scene(world)
    st = ctx.scene_stats()
    flags = S.PC_FLAG_ACCUMULATE | S.PC_FLAG_CLAMP_INDIRECT | S.PC_FLAG_SKIP_HISTORY | (S.PC_FLAG_IBL if args.env else 0)
    pc = S.ReferencePC(0, flags, 1, 1e-5, 1.0, focal, 3, min(args.bounces, 6))
    ctx.set_kernel_timing(True)
    ctx.render(pc, cam, w, h, frames=args.spp)
    ms, _ = ctx.last_render_timing()
    if args.lut:
        lut = dds.read_lut(args.lut)
    else:
        g = np.linspace(0.0, 1.0, 48)
        b, gg, r = np.meshgrid(g, g, g, indexing="ij")
        lut = dds.encode_r9g9b9e5(np.stack([r, gg, b], axis=-1))
    ctx.set_tone_map_lut(lut)
    write_png(args.out, ctx.tone_map(args.exposure, 1.0))
    print("%s: %d triangles, %dx%d x %d spp in %.1f ms (%.0f Mpaths/s) -> %s" % (
        os.path.basename(args.gltf), st.triangleCount, w, h, args.spp, ms, w * h * args.spp / ms / 1e3, args.out))
    ctx.close()


if __name__ == "__main__":
    main()
