#!/usr/bin/env python3
"""What adopting streamed-in textures and materials costs (prosper_pt_update_textures / _materials) against uploading the
scene again: FlightHelmet with every image at the asset's own 2048 x 2048 texels (16 MB each), three images per frame,
1-spp frames with three in flight.  Tooling (profiles/r04_adoption.txt); run on the GPU box.

    python scripts/adoption_bench.py [texture_size]"""
import copy
import ctypes
import os
import sys
import time

import numpy as np

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
    images = len(full.textures) - 1
    w, h = 1920, 1080
    cam, focal = Camera.from_world(full, w, h).update_buffer()
    flags = S.PC_FLAG_ACCUMULATE | S.PC_FLAG_CLAMP_INDIRECT | S.PC_FLAG_IBL
    ctx = capi.Context(0)
    t0 = time.perf_counter()
    ctx.upload_scene(full)
    print("whole upload of the loaded scene: %.1f ms (texture side %.1f ms, hierarchy %.1f ms)" % (
        (time.perf_counter() - t0) * 1e3, ctx.scene_stats().textureSeconds * 1e3, ctx.scene_stats().bvhBuildSeconds * 1e3), flush=True)
    ctx.upload_scene(streamed_state(full, 0))

    def frames(n, first):
        for f in range(n):
            pc = S.ReferencePC(0, flags | (S.PC_FLAG_SKIP_HISTORY if first + f == 0 else 0), 1 + first + f, 1e-5, 1.0, focal, 3, 4)
            ctx.render(pc, cam, w, h, frames=1, flags=S.RENDER_PIPELINED)
    frames(12, 0)
    hip.hipDeviceSynchronize()
    t0 = time.perf_counter()
    frames(30, 12)
    hip.hipDeviceSynchronize()
    plain = (time.perf_counter() - t0) / 30 * 1e3
    print("1-spp frames, three in flight, nothing adopted: %.3f ms per frame" % plain, flush=True)
    per_frame = 3
    tex_ms, mat_ms = [], []
    t_all = time.perf_counter()
    loaded, frame = 0, 42
    while loaded < images:
        n = min(per_frame, images - loaded)
        state = streamed_state(full, loaded + n)
        mats = state.materials
        t0 = time.perf_counter()
        ctx.update_textures(full.textures[loaded + 1:loaded + 1 + n], loaded + 1)
        t1 = time.perf_counter()
        ctx.update_materials(mats, 0)
        t2 = time.perf_counter()
        tex_ms.append((t1 - t0) * 1e3 / n)
        mat_ms.append((t2 - t1) * 1e3)
        frames(1, frame)
        frame += 1
        loaded += n
    hip.hipDeviceSynchronize()
    total = (time.perf_counter() - t_all) * 1e3
    steps = len(tex_ms)
    print("adopting %d images of %d x %d (%.0f MB each) %d per frame + the material table every frame:" % (images, size, size, size * size * 4 / 1e6, per_frame))
    print("  prosper_pt_update_textures: %.2f ms per image on the host (median; the call waits for its own copies only)" % float(np.median(tex_ms)))
    print("  prosper_pt_update_materials: %.3f ms per call (median; %d materials)" % (float(np.median(mat_ms)), len(full.materials)))
    print("  %d frames with adoption: %.2f ms per frame against %.3f without" % (steps, total / steps, plain))
    frames(12, frame)
    hip.hipDeviceSynchronize()
    t0 = time.perf_counter()
    frames(30, frame + 12)
    hip.hipDeviceSynchronize()
    print("1-spp frames afterwards (everything adopted): %.3f ms per frame" % ((time.perf_counter() - t0) / 30 * 1e3))
    ctx.close()


if __name__ == "__main__":
    main()
