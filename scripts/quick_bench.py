#!/usr/bin/env python3
"""Torch-free timing loop for kernel iteration: hipEvent time of prosper_pt_render_frames.

    python scripts/quick_bench.py [--config c2] [--steps 10] [--megakernel] [--lib path/to/variant.so]
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="c2")
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--megakernel", action="store_true")
    ap.add_argument("--persistent", action="store_true")
    ap.add_argument("--single-chain", action="store_true", help="PROSPER_PT_CREATE_SINGLE_CHAIN (A/B)")
    ap.add_argument("--stats", action="store_true")
    ap.add_argument("--counters", action="store_true", help="one counted render: per-stage work counters")
    ap.add_argument("--spp", type=int, default=0)
    ap.add_argument("--width", type=int, default=0)
    ap.add_argument("--height", type=int, default=0)
    ap.add_argument("--lib", default=None)
    ap.add_argument("--small-textures", action="store_true")
    ap.add_argument("--ranks", type=int, default=1, help="render only rank 0's stripes of an N-rank job (strong-scaling share)")
    args = ap.parse_args()
    from prosper_amd import capi
    if args.lib:
        capi.LIB_PATH = os.path.abspath(args.lib)
    from prosper_amd import scenes, structs as S
    from prosper_amd.rt_reference import Camera
    tex = 128 if args.small_textures else 1024
    cfg = {
        "c2": (lambda: scenes.cornell(), 1920, 1080, 8, 4, False),
        "c3": (lambda: scenes.sponza_class(texture_size=tex), 1920, 1080, 8, 4, True),
        "c4": (lambda: scenes.sponza_class(lights=True, foliage=True, texture_size=tex), 1920, 1080, 8, 4, True),
        "helmet": (lambda: __import__("prosper_amd.flight_helmet", fromlist=["x"]).load_fixture(), 1920, 1080, 8, 4, True),
        "helmet2k": (lambda: __import__("prosper_amd.flight_helmet", fromlist=["x"]).load_fixture(texture_size=2048), 1920, 1080, 8, 4, True),
    }[args.config]
    builder, w, h, spp, mb, ibl = cfg
    spp = args.spp or spp
    w = args.width or w
    h = args.height or h
    world = builder()
    cam, focal = Camera.from_world(world, w, h).update_buffer()
    ctx = capi.Context(0, S.CREATE_MEGAKERNEL if args.megakernel else (
        S.CREATE_PERSISTENT if args.persistent else (S.CREATE_SINGLE_CHAIN if args.single_chain else 0)))
    ctx.upload_scene(world)
    if args.stats:
        st = ctx.scene_stats()
        print('triangles %d nodes %d maxDepth %d build %.3fs deviceMB %.1f' % (st.triangleCount, st.nodeCount, st.maxDepth, st.buildSeconds, st.deviceBytes / 1e6))
    flags = S.PC_FLAG_ACCUMULATE | S.PC_FLAG_CLAMP_INDIRECT | S.PC_FLAG_SKIP_HISTORY | (S.PC_FLAG_IBL if ibl else 0)
    pc = S.ReferencePC(0, flags, 1, 1e-5, 1.0, focal, 3, mb)
    tile = None
    if args.ranks > 1:
        from prosper_amd import tiling
        tile = tiling.tile_for_rank(0, args.ranks)
    if args.counters:
        ctx.reset_counters()
        ctx.render(pc, cam, w, h, frames=spp, flags=S.RENDER_COUNT_WORK, tile=tile)
        for stage, name in enumerate(("generate_extend", "shade", "trace", "accumulate")):
            c = ctx.stage_counters(stage)
            d = {n: int(getattr(c, n)) for n, _ in c._fields_ if n != "reserved"}
            print("  %-16s %s" % (name, " ".join("%s=%d" % kv for kv in d.items() if kv[1])))
    ctx.set_kernel_timing(True)
    times = []
    for i in range(args.steps + 2):
        ctx.render(pc, cam, w, h, frames=spp, tile=tile)
        ms, per = ctx.last_render_timing()
        if i >= 2:
            times.append(ms)
    times.sort()
    med = times[len(times) // 2]
    print("  " + "  ".join("%s %dx%.0fus" % (k.replace("wf_", ""), v[1], v[0] * 1e3 / max(1, v[1])) for k, v in per.items() if v[1]))
    img = ctx.read_hdr()
    print("%s %s: median %.3f ms  min %.3f  => %.1f Mpaths/s  (checksum %.6f)" % (
        args.config, "megakernel" if args.megakernel else ("persistent" if args.persistent else "wavefront"), med, times[0], w * h * spp / args.ranks / med / 1e3,
        float(img[..., :3].astype("float64").mean())))


if __name__ == "__main__":
    main()
