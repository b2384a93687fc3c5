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

    def frames(n):
        for _ in range(n):
            pc = S.ReferencePC(0, flags | (S.PC_FLAG_SKIP_HISTORY if frame[0] == 0 else 0), 1 + frame[0], 1e-5, 1.0, focal, 3, 4)
            ctx.render(pc, cam, w, h, frames=1, flags=S.RENDER_PIPELINED)
            frame[0] += 1
    frames(3)
    print("%d meshes per call; per call: host time of prosper_pt_update_meshes (device idle when it returns), of which the"
          " hierarchy (subtrees of the completed model instances + assembly), triangles afterwards" % per_call)
    loaded = 0
    total = 0.0
    while loaded < meshes:
        n = min(per_call, meshes - loaded)
        t0 = time.perf_counter()
        ctx.update_meshes(full, list(range(loaded, loaded + n)))
        ms = (time.perf_counter() - t0) * 1e3
        total += ms
        loaded += n
        st = ctx.scene_stats()
        print("  meshes %2d..%2d: %6.1f ms (hierarchy %5.1f ms)  -> %7d triangles, %6d nodes" % (
            loaded - n, loaded - 1, ms, st.bvhBuildSeconds * 1e3, st.triangleCount, st.nodeCount), flush=True)
        frames(2)
    hip.hipDeviceSynchronize()
    print("all %d meshes adopted in %d calls: %.1f ms in all, against %.1f ms for ONE upload of the finished scene and"
          " %d uploads of the growing one otherwise" % (meshes, (meshes + per_call - 1) // per_call, total, whole, (meshes + per_call - 1) // per_call))
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
