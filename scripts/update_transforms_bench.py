#!/usr/bin/env python3
"""How long does a moved instance cost?  Uploads S-sponza-class, moves 1 / 4 / all model instances a few times and times
prosper_pt_update_transforms_async (the GPU refit: host time of the call, device time between hipEvents on the stream)
and prosper_pt_rebuild_hierarchy (the host-side re-split + re-assembly).  Tooling (profiles/r03_moved_instances.txt)."""
import ctypes
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from prosper_amd import capi, scenes, structs as S  # noqa: E402
from prosper_amd.rt_reference import Camera  # noqa: E402
from prosper_amd.world import translate  # noqa: E402


def main():
    hip = ctypes.CDLL("libamdhip64.so")
    full = len(sys.argv) > 1 and sys.argv[1] == "full"
    world = scenes.sponza_class(lights=full, foliage=full) if full else scenes.sponza_class(texture_size=64, sky_size=32)
    ctx = capi.Context(0)
    t0 = time.perf_counter()
    ctx.upload_scene(world)
    st = ctx.scene_stats()
    hs = ctx.hierarchy_state()
    print("upload: %.1f ms wall, hierarchy build %.1f ms, %d triangles, %d nodes in %d levels, %d instances" % (
        (time.perf_counter() - t0) * 1e3, st.bvhBuildSeconds * 1e3, st.triangleCount, hs.nodeCount, hs.levels,
        len(world.model_instances)), flush=True)
    w, h = 1920, 1080
    cam, focal = Camera.from_world(world, w, h).update_buffer()
    pc = S.ReferencePC(0, S.PC_FLAG_ACCUMULATE | S.PC_FLAG_CLAMP_INDIRECT | S.PC_FLAG_SKIP_HISTORY | S.PC_FLAG_IBL, 1, 1e-5, 1.0, focal, 3, 4)
    start, stop, stream = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_void_p()
    hip.hipEventCreate(ctypes.byref(start))
    hip.hipEventCreate(ctypes.byref(stop))
    hip.hipStreamCreate(ctypes.byref(stream))  # a non-null stream: the refit is enqueued by the call itself, not deferred
    n = len(world.model_instances)
    for moved in (1, 4, n):
        host, device = [], []
        for rep in range(6):
            for i in range(moved):
                k = (rep * 7 + i * 3) % n
                model, m = world.model_instances[k]
                world.model_instances[k] = (model, translate((0.01 * (rep + 1), 0.0, 0.0)) @ m)
            world._frozen = None
            f = world.freeze()
            hip.hipDeviceSynchronize()
            hip.hipEventRecord(start, stream)
            t0 = time.perf_counter()
            capi._check(capi.lib().prosper_pt_update_transforms_async(ctx._h, ctypes.cast(f["transforms"], ctypes.c_void_p), n, 1, stream))
            host.append((time.perf_counter() - t0) * 1e3)
            hip.hipEventRecord(stop, stream)
            hip.hipEventSynchronize(stop)
            ms = ctypes.c_float()
            hip.hipEventElapsedTime(ctypes.byref(ms), start, stop)
            device.append(ms.value)
        print("moved %2d of %d instances: refit %.3f ms on the host (call returns), %.3f ms on the device; tree measure x%.3f" % (
            moved, n, sorted(host)[len(host) // 2], sorted(device)[len(device) // 2], ctx.hierarchy_state().costRatio), flush=True)
    # frames while instances move: update (staged, run by the frame's own chain into the next scene version) + 1-spp
    # frame, pipelined, against the same frames without updates; the world is frozen (Python) outside the timed loop
    poses = []
    for fr in range(60):
        model, m = world.model_instances[3]
        world.model_instances[3] = (model, translate((0.002, 0.0, 0.0)) @ m)
        world._frozen = None
        f = world.freeze()
        poses.append(bytes(ctypes.string_at(ctypes.addressof(f["transforms"]), ctypes.sizeof(f["transforms"]))))
    for label, update in (("no updates", False), ("one instance moves every frame", True)):
        for _ in range(6):
            ctx.render(pc, cam, w, h, frames=1, flags=S.RENDER_PIPELINED)
        hip.hipDeviceSynchronize()
        frames = 60
        t0 = time.perf_counter()
        for fr in range(frames):
            if update:
                capi._check(capi.lib().prosper_pt_update_transforms(ctx._h, poses[fr], n))
            ctx.render(pc, cam, w, h, frames=1, flags=S.RENDER_PIPELINED)
        hip.hipDeviceSynchronize()
        print("%-32s %.3f ms per 1-spp 1920x1080 frame, three frames in flight" % (label + ":", (time.perf_counter() - t0) * 1e3 / frames), flush=True)
    hs = ctx.hierarchy_state()
    print("after all that: %d refits, %d rebuilds, tree measure x%.3f" % (hs.refits, hs.rebuilds, hs.costRatio), flush=True)
    t0 = time.perf_counter()
    ctx.rebuild_hierarchy()
    st = ctx.scene_stats()
    print("prosper_pt_rebuild_hierarchy: %.1f ms wall, host build %.1f ms; tree measure x%.3f" % (
        (time.perf_counter() - t0) * 1e3, st.bvhBuildSeconds * 1e3, ctx.hierarchy_state().costRatio), flush=True)


if __name__ == "__main__":
    main()
