#!/usr/bin/env python3
"""prosper's whole loading sequence on the reference's bundled asset (FlightHelmet, every image at its own 2048 x 2048 texels)
with the frame loop running: first the meshes, two per frame, under placeholder materials; then the images, three per frame,
each material switching from its placeholder when its images are there (WorldData::handleDeferredLoading,
WorldData.cpp:588-647).  1-spp frames, three in flight, the host never more than three frames ahead.  Against uploading the
loaded asset in one call.  Tooling (profiles/r04_loading_sequence.txt); run on the GPU box.

    python scripts/loading_sequence_bench.py [texture_size]"""
import ctypes
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from prosper_amd import capi, flight_helmet, structs as S  # noqa: E402
from prosper_amd.rt_reference import Camera  # noqa: E402
from test_adoption import streamed_state  # noqa: E402


def main():
    size = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
    hip = ctypes.CDLL("libamdhip64.so")
    full = flight_helmet.load_fixture(texture_size=size)
    meshes, images = len(full.metadatas), len(full.textures) - 1
    w, h = 1920, 1080
    cam, focal = Camera.from_world(full, w, h).update_buffer()
    flags = S.PC_FLAG_ACCUMULATE | S.PC_FLAG_CLAMP_INDIRECT | S.PC_FLAG_IBL
    ctx = capi.Context(0)
    for _ in range(2):
        t0 = time.perf_counter()
        ctx.upload_scene(full)
        whole = (time.perf_counter() - t0) * 1e3
    frame = [0]

    detail = []

    def frames(n):
        for _ in range(n):
            pc = S.ReferencePC(0, flags | (S.PC_FLAG_SKIP_HISTORY if frame[0] == 0 else 0), 1 + frame[0], 1e-5, 1.0, focal, 3, 4)
            t0 = time.perf_counter()
            ctx.render(pc, cam, w, h, frames=1, flags=S.RENDER_PIPELINED)
            t1 = time.perf_counter()
            frame[0] += 1
            if frame[0] % 3 == 0:
                hip.hipDeviceSynchronize()
            detail.append(((t1 - t0) * 1e3, (time.perf_counter() - t1) * 1e3))
    frames(12)
    hip.hipDeviceSynchronize()
    t0 = time.perf_counter()
    frames(30)
    hip.hipDeviceSynchronize()
    loaded_ms = (time.perf_counter() - t0) / 30 * 1e3
    for attempt in (1, 2):  # (the second pass: no first-use cost of the process left)
        placeholders = streamed_state(full, 0)
        t0 = time.perf_counter()
        ctx.upload_scene(placeholders.with_meshes_loaded([]))
        first = (time.perf_counter() - t0) * 1e3
        frame[0] = 0
        groups, calls, phase = [], [], []
        t_all = time.perf_counter()

        def step(work):
            t0 = time.perf_counter()
            work()
            calls.append((time.perf_counter() - t0) * 1e3)
            t0 = time.perf_counter()
            frames(3)
            groups.append((time.perf_counter() - t0) * 1e3 / 3)
            phase.append(what[0])
        what = ["meshes arriving"]
        loaded = 0
        while loaded < meshes:
            n = min(2, meshes - loaded)
            step(lambda: ctx.update_meshes(placeholders, list(range(loaded, loaded + n)), wait=False))
            loaded += n
        what[0] = "waiting for the last mesh build"
        while ctx.hierarchy_state().geometryBuildRunning:
            step(lambda: None)
        meshes_ms = (time.perf_counter() - t_all) * 1e3
        what[0] = "images arriving"
        arrived = 0
        while arrived < images:
            n = min(3, images - arrived)
            state = streamed_state(full, arrived + n)

            def adopt():
                ctx.update_textures(full.textures[arrived + 1:arrived + 1 + n], arrived + 1)
                ctx.update_materials(state.materials, 0)
            step(adopt)
            arrived += n
        hip.hipDeviceSynchronize()
        total = (time.perf_counter() - t_all) * 1e3
        med = sorted(groups)[len(groups) // 2]
        worst = max(range(len(groups)), key=lambda i: groups[i])
        print("groups of three frames over 1 ms per frame: " + ", ".join("%d (%s): %.2f ms" % (i, phase[i], g) for i, g in enumerate(groups) if g > 1.0))
        print("the slowest: group %d of %d, %s" % (worst, len(groups), phase[worst]))
        slow = max(range(len(detail)), key=lambda i: detail[i][0] + detail[i][1])
        print("slowest single render: call %.2f ms on the host, then %.2f ms waiting for the device; the ones around it: %s" % (
            detail[slow][0], detail[slow][1], " ".join("%.2f+%.2f" % d for d in detail[max(0, slow - 3):slow + 4])))
        print("pass %d:" % attempt)
        print("FlightHelmet, %d meshes, %d images of %d x %d: uploading the loaded asset in one call %.1f ms; its 1-spp frames %.3f ms" % (
            meshes, images, size, size, whole, loaded_ms))
        print("loading it with the frame loop running (first upload with placeholders only: %.1f ms): meshes in the scene after %.1f ms, everything"
              " after %.1f ms and %d frames; the adoption calls %.2f ms median / %.2f ms worst on the host per frame; frames %.3f ms median / %.3f ms"
              " worst per frame (groups of three)" % (first, meshes_ms, total, frame[0], sorted(calls)[len(calls) // 2], max(calls), med, max(groups)))


if __name__ == "__main__":
    main()
