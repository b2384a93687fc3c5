#!/usr/bin/env python3
"""The frame loop while instances drift far enough for the hierarchy to be re-split (S-sponza-class, three instances carried
out of the hall, moved EVERY frame; 1-spp frames, three in flight, host paced like a swapchain): frame times around the
re-split, which runs on the context's worker thread since round 4 (it was a synchronous 60-75 ms rebuild inside
prosper_pt_update_transforms).  Tooling (profiles/r04_drift_rebuild.txt); run on the GPU box.

    python scripts/drift_rebuild_bench.py [sync]        sync: debug option alwaysRebuild on the frame that would trigger"""
import copy
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
    base = scenes.sponza_class()
    w, h = 1920, 1080
    cam, focal = Camera.from_world(base, w, h).update_buffer()
    flags = S.PC_FLAG_ACCUMULATE | S.PC_FLAG_CLAMP_INDIRECT | S.PC_FLAG_IBL | S.PC_FLAG_SKIP_HISTORY
    ctx = capi.Context(0)
    ctx.upload_scene(base)
    steps = 240
    poses = []
    for k in range(steps):
        world = copy.copy(base)
        world._frozen = None
        world.model_instances = list(base.model_instances)
        for i in (3, 5, 7):
            model, m = world.model_instances[i]
            world.model_instances[i] = (model, translate((0.12 * k, 0.016 * k, 0.032 * k)) @ m)
        world.freeze()
        poses.append(world)
    # the host never runs more than three frames ahead (it waits for the device every third frame; a pipelined render does
    # not touch the caller's stream, so there is nothing finer to wait on from outside)
    times, update_ms = [], []
    t0 = time.perf_counter()
    for k, world in enumerate(poses):
        t1 = time.perf_counter()
        ctx.update_transforms(world)
        update_ms.append((time.perf_counter() - t1) * 1e3)
        ctx.render(S.ReferencePC(0, flags, 1 + k, 1e-5, 1.0, focal, 3, 4), cam, w, h, frames=1, flags=S.RENDER_PIPELINED)
        if k % 3 == 2:
            hip.hipDeviceSynchronize()
            times.append((time.perf_counter() - t0) * 1e3 / 3)
            t0 = time.perf_counter()
    hip.hipDeviceSynchronize()
    ctx.finish_mesh_updates()
    hs = ctx.hierarchy_state()
    med = sorted(times)[len(times) // 2]
    worst = max(range(len(times)), key=lambda i: times[i])
    print("%d frames, every one with three instances moved: %.3f ms per frame (median of groups of three), worst group %.3f ms per frame (group %d);"
          " prosper_pt_update_transforms %.3f ms median, worst %.3f ms; %d refits, %d re-splits on the worker thread, measure x%.2f at the end" % (
              steps, med, times[worst], worst, sorted(update_ms)[len(update_ms) // 2], max(update_ms), hs.refits, hs.rebuilds, hs.costRatio))
    print("groups over 1.5 x the median: %s" % ", ".join("%d: %.2f ms" % (i, t) for i, t in enumerate(times) if t > 1.5 * med))


if __name__ == "__main__":
    main()
