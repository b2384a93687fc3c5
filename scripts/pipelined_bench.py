#!/usr/bin/env python3
"""Throughput of back-to-back renders, in order vs PROSPER_PT_RENDER_PIPELINED (frames in flight), at the full
frame and at the per-rank shares of an N-rank job.

    python scripts/pipelined_bench.py c2 [c3 c4]        # SPP=1 in the environment: single-sample frames
"""
import ctypes
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from prosper_amd import capi, scenes, structs as S, tiling  # noqa: E402
from prosper_amd.rt_reference import Camera  # noqa: E402

SPP = int(os.environ.get("SPP", "8"))
if os.environ.get("LIB"):  # A/B against another build of the library
    capi.LIB_PATH = os.path.abspath(os.environ["LIB"])
CONFIGS = {
    "c2": (scenes.cornell, False),
    "c3": (lambda: scenes.sponza_class(), True),
    "c4": (lambda: scenes.sponza_class(lights=True, foliage=True), True),
    "helmet": (lambda: __import__("prosper_amd.flight_helmet", fromlist=["x"]).load_fixture(), True),
    "helmet2k": (lambda: __import__("prosper_amd.flight_helmet", fromlist=["x"]).load_fixture(texture_size=2048), True),
}


def main():
    hip = ctypes.CDLL("libamdhip64.so")
    width, height = 1920, 1080
    for name in sys.argv[1:] or ["c2"]:
        builder, ibl = CONFIGS[name]
        world = builder()
        cam, focal = Camera.from_world(world, width, height).update_buffer()
        flags = S.PC_FLAG_ACCUMULATE | S.PC_FLAG_CLAMP_INDIRECT | S.PC_FLAG_SKIP_HISTORY | (S.PC_FLAG_IBL if ibl else 0)
        pc = S.ReferencePC(0, flags, 1, 1e-5, 1.0, focal, 3, 4)
        ctx = capi.Context(0)
        ctx.upload_scene(world)
        for ranks in ((8, 4, 2, 1) if name == "c2" else (1,)):
            tile = tiling.tile_for_rank(0, ranks) if ranks > 1 else None
            for render_flags, label in ((0, "in order"), (S.RENDER_PIPELINED, "pipelined")):
                for _ in range(int(os.environ.get("WARM", "4"))):
                    ctx.render(pc, cam, width, height, frames=SPP, tile=tile, flags=render_flags)
                hip.hipDeviceSynchronize()
                steps = 30 if name == "c2" else 6
                t0 = time.perf_counter()
                for _ in range(steps):
                    ctx.render(pc, cam, width, height, frames=SPP, tile=tile, flags=render_flags)
                host_ms = (time.perf_counter() - t0) / steps * 1e3  # until the calls have returned: the host's share
                hip.hipDeviceSynchronize()
                ms = (time.perf_counter() - t0) / steps * 1e3
                print("%s ranks %d %-9s: %.3f ms/frame  %.0f Mpaths/s  (host %.3f ms/frame)" % (
                    name, ranks, label, ms, width * height * SPP / ranks / ms / 1e3, host_ms), flush=True)
        ctx.close()


if __name__ == "__main__":
    main()
