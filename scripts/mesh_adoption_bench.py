#!/usr/bin/env python3
"""What adopting streamed-in meshes costs (prosper_pt_update_meshes) against uploading the scene again: S-sponza-class
(262 k triangles, 31 meshes in 15 models, 43 instances) arriving a few meshes per call, 1-spp frames in flight between the
calls.  Tooling (profiles/r04_mesh_adoption.txt); run on the GPU box.

    python scripts/mesh_adoption_bench.py [meshes_per_call]"""
import ctypes
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from prosper_amd import capi, scenes, structs as S  # noqa: E402
from prosper_amd.rt_reference import Camera  # noqa: E402


def main():
    per_call = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    pause = float(sys.argv[2]) * 1e-3 if len(sys.argv) > 2 else 0.0  # host-side gap between frames, ms
    hip = ctypes.CDLL("libamdhip64.so")
    full = scenes.sponza_class()
    meshes = len(full.metadatas)
    w, h = 1920, 1080
    cam, focal = Camera.from_world(full, w, h).update_buffer()
    flags = S.PC_FLAG_ACCUMULATE | S.PC_FLAG_CLAMP_INDIRECT | S.PC_FLAG_IBL
    ctx = capi.Context(0)
    for _ in range(2):
        t0 = time.perf_counter()
        ctx.upload_scene(full)
        whole = (time.perf_counter() - t0) * 1e3
    st = ctx.scene_stats()
    print("whole upload of the loaded scene (%d triangles): %.1f ms (texture side %.1f ms, hierarchy build %.1f ms)" % (
        st.triangleCount, whole, st.textureSeconds * 1e3, st.bvhBuildSeconds * 1e3), flush=True)
    ctx.upload_scene(full.with_meshes_loaded([]))
    frame = [0]
    # The host is paced: it never runs more than three frames ahead of the GPU (every third frame it waits for the device -
    # a pipelined render does not touch the caller's stream, so there is no per-frame fence to wait on from outside; a
    # swapchain paces prosper the same way).  Unpaced, a host enqueues a thousand frames while a build runs.
    def frames(n):
        for _ in range(n):
            pc = S.ReferencePC(0, flags | (S.PC_FLAG_SKIP_HISTORY if frame[0] == 0 else 0), 1 + frame[0], 1e-5, 1.0, focal, 3, 4)
            ctx.render(pc, cam, w, h, frames=1, flags=S.RENDER_PIPELINED)
            frame[0] += 1
            if frame[0] % 3 == 0:
                hip.hipDeviceSynchronize()
            if pause:
                time.sleep(pause)
    frames(3)
    plain = None
    hip.hipDeviceSynchronize()
    t0 = time.perf_counter()
    frames(30)
    hip.hipDeviceSynchronize()
    plain = (time.perf_counter() - t0) / 30 * 1e3
    print("1-spp frames of the empty scene, three in flight: %.3f ms per frame" % plain)
    print("%d meshes per call, one call per frame, the frame loop never waits (prosper_pt_update_meshes returns once the bytes are"
          " on the device; a worker thread builds the geometry; the first render after it is done switches):" % per_call)
    loaded = 0
    calls = []
    t_all = time.perf_counter()
    n_frames = 0
    while loaded < meshes:
        n = min(per_call, meshes - loaded)
        t0 = time.perf_counter()
        ctx.update_meshes(full, list(range(loaded, loaded + n)), wait=False)
        calls.append((time.perf_counter() - t0) * 1e3)
        loaded += n
        frames(1)
        n_frames += 1
    handed = (time.perf_counter() - t_all) * 1e3
    while ctx.hierarchy_state().geometryBuildRunning:
        frames(1)
        n_frames += 1
    hip.hipDeviceSynchronize()
    total = (time.perf_counter() - t_all) * 1e3
    st = ctx.hierarchy_state()
    print("  prosper_pt_update_meshes: %.2f ms per call on the host (median; the calls: %s)" % (
        sorted(calls)[len(calls) // 2], " ".join("%.2f" % c for c in calls)))
    print("  all %d meshes handed over after %.1f ms, in the scene after %.1f ms and %d frames (%.2f ms per frame); %d builds for %d calls" % (
        meshes, handed, total, n_frames, total / n_frames, st.geometryInstalls, st.meshUpdates))
    print("  against %.1f ms for ONE upload of the finished scene" % whole)
    # the same, waiting for every call's build (what a synchronous adoption costs)
    ctx.upload_scene(full.with_meshes_loaded([]))
    frame[0] = 0
    frames(3)
    loaded = 0
    sync_ms = []
    while loaded < meshes:
        n = min(per_call, meshes - loaded)
        t0 = time.perf_counter()
        ctx.update_meshes(full, list(range(loaded, loaded + n)), wait=True)
        sync_ms.append((time.perf_counter() - t0) * 1e3)
        loaded += n
        st = ctx.scene_stats()
        print("  waited for: meshes %2d..%2d %6.1f ms (hierarchy %5.1f ms) -> %7d triangles" % (loaded - n, loaded - 1, sync_ms[-1], st.bvhBuildSeconds * 1e3, st.triangleCount), flush=True)
        frames(2)
    print("  waiting for each build: %.1f ms in all" % sum(sync_ms))
    # the frame loop of a scene that is half there while the other half arrives: frame times with and without a build
    # under way (host time from one render call to the next, paced by the frame of three renders ago)
    half = list(range(0, meshes, 2))
    rest = [i for i in range(meshes) if i not in half]
    ctx.upload_scene(full.with_meshes_loaded(half))
    frame[0] = 0
    frames(12)

    def frame_times(n):  # per group of three frames (the pacing unit), as ms per frame
        out = []
        for _ in range(n):
            t0 = time.perf_counter()
            frames(3)
            out.append((time.perf_counter() - t0) * 1e3 / 3)
        return out
    quiet = frame_times(60)
    busy = []
    t0 = time.perf_counter()
    for k in range(0, len(rest), per_call):
        ctx.update_meshes(full, rest[k:k + per_call], wait=False)
        busy += frame_times(1)
    while ctx.hierarchy_state().geometryBuildRunning:
        busy += frame_times(1)
    took = (time.perf_counter() - t0) * 1e3
    after = frame_times(60)
    med = lambda v: sorted(v)[len(v) // 2]
    print("half of the scene (%d meshes) rendering while the other half arrives %d per three frames: %.3f ms per frame (median of groups"
          " of three) before, %.3f ms median / %.3f ms worst of the %d groups while the builds ran (%.1f ms), %.3f ms after" % (
              len(half), per_call, med(quiet), med(busy), max(busy), len(busy), took, med(after)))
    # the hierarchy an adoption leaves behind against a fresh build: frame time
    def timed(n=30):
        frames(6)
        hip.hipDeviceSynchronize()
        t0 = time.perf_counter()
        frames(n)
        hip.hipDeviceSynchronize()
        return (time.perf_counter() - t0) / n * 1e3
    adopted = timed()
    ctx.upload_scene(full)
    frame[0] = 0
    print("1-spp frames, three in flight: %.3f ms on the adopted scene, %.3f ms on a fresh upload" % (adopted, timed()))


if __name__ == "__main__":
    main()
