#!/usr/bin/env python3
"""Renders one configuration and prints a SHA-256 of the HDR image plus the schedule-independent counters.

Run it under different hierarchies (PROSPER_PT_DEBUG=1 PROSPER_PT_DEBUG_OPTIONS="boxPad=<coefficient >= 1.6e-5>,ldsStackEntries=16,..."):
the hit contract says the digest must not change.
"""
import argparse
import hashlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="c3")
    ap.add_argument("--spp", type=int, default=8)
    args = ap.parse_args()
    from prosper_amd import capi, scenes, structs as S
    from prosper_amd.rt_reference import Camera
    world = {"c2": lambda: scenes.cornell(),
             "c3": lambda: scenes.sponza_class(texture_size=128),
             "c4": lambda: scenes.sponza_class(lights=True, foliage=True, texture_size=128)}[args.config]()
    w, h = 1920, 1080
    cam, focal = Camera.from_world(world, w, h).update_buffer()
    ctx = capi.Context(0)
    ctx.upload_scene(world)
    flags = S.PC_FLAG_ACCUMULATE | S.PC_FLAG_CLAMP_INDIRECT | S.PC_FLAG_SKIP_HISTORY | (0 if args.config == "c2" else S.PC_FLAG_IBL)
    pc = S.ReferencePC(0, flags, 1, 1e-5, 1.0, focal, 3, 4)
    ctx.reset_counters()
    ctx.render(pc, cam, w, h, frames=args.spp, flags=S.RENDER_COUNT_WORK)
    img = ctx.read_hdr()
    c = ctx.counters()
    st = ctx.scene_stats()
    print("%s pad=%s nodes=%d sha256=%s closestHits=%d shadowRays=%d anyHitCalls=%d nodeVisits=%d triangleTests=%d" % (
        args.config, ("%g" % ctx.debug_options().boxPad) if ctx.debug_options().boxPad else "default", st.nodeCount,
        hashlib.sha256(img.tobytes()).hexdigest()[:16], c.closestHits, c.shadowRays, c.anyHitCalls, c.nodeVisits,
        c.triangleTests))


if __name__ == "__main__":
    main()
