#!/usr/bin/env python3
"""The four bench configurations at their own size - 1920x1080, 8 spp, maxBounces 4 - exactly as bench.py renders them
(ONE batched call, three frames in flight) against the CPU oracle accumulating the same eight frames: every float of the
RGBA32F image, bit for bit.  (The -m gpu tests hold the 1-2 spp versions; this is the whole step.)  Tooling.

    python scripts/full_size_parity.py [c2 c3 c4 helmet]      -> profiles/r03_full_size_parity.txt"""
import hashlib
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402


def main():
    from conftest import default_pc, same_bits
    from oracle import binding as O
    from prosper_amd import capi, scenes, structs as S
    builders = {
        "c2": (lambda: scenes.cornell(), False),
        "c3": (lambda: scenes.sponza_class(), True),
        "c4": (lambda: scenes.sponza_class(lights=True, foliage=True), True),
        "helmet": (lambda: __import__("prosper_amd.flight_helmet", fromlist=["x"]).load_fixture(), True),
        "helmet2k": (lambda: __import__("prosper_amd.flight_helmet", fromlist=["x"]).load_fixture(texture_size=2048), True),
    }
    w, h, spp = 1920, 1080, 8
    ctx = capi.Context(0)
    bad = 0
    for name in sys.argv[1:] or ["c2", "c3", "c4", "helmet"]:
        builder, ibl = builders[name]
        world = builder()
        c = world.camera
        cam, fl = O.camera_uniforms(c["eye"], c["target"], c["up"], c["fov"], c["zN"], c["zF"], w, h)
        ctx.upload_scene(world)
        ctx.render(default_pc(S, fl, max_bounces=4, ibl=ibl), cam, w, h, frames=spp, flags=S.RENDER_PIPELINED)
        got = ctx.read_hdr()
        osc = O.OracleScene(world, brute_force=(name == "c2"))
        t0 = time.time()
        want = None
        for frame in range(1, spp + 1):
            want, _ = osc.render(default_pc(S, fl, frame_index=frame, max_bounces=4, ibl=ibl, skip_history=(frame == 1)), cam, w, h,
                                 history=want)
        osc.close()
        ok = same_bits(got, want).all(axis=2)
        bad += int((~ok).sum())
        print("%-6s %d triangles: %d x %d x %d spp = %d paths, %d of %d pixels differ from the oracle (oracle %.0f s); sha256 of the image %s; mean radiance %.6f" % (
            name, world.triangle_count(), w, h, spp, w * h * spp, int((~ok).sum()), ok.size, time.time() - t0,
            hashlib.sha256(np.ascontiguousarray(got).tobytes()).hexdigest()[:16], float(got[..., :3].astype(np.float64).mean())), flush=True)
    ctx.close()
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
