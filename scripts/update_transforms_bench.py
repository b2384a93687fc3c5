#!/usr/bin/env python3
"""How long does a moved instance cost?  Uploads S-sponza-class, then moves 1 / 4 / all model instances a few times and
prints the library's timers (prosper_pt_scene_stats: whole update, hierarchy rebuild).  Tooling (profiles/r02_bvh_instancing.txt)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from prosper_amd import capi, scenes  # noqa: E402
from prosper_amd.world import translate  # noqa: E402


def main():
    world = scenes.sponza_class(texture_size=64, sky_size=32)
    ctx = capi.Context(0)
    t0 = time.perf_counter()
    ctx.upload_scene(world)
    st = ctx.scene_stats()
    print("upload: %.1f ms wall, hierarchy %.1f ms, %d triangles, %d instances" % (
        (time.perf_counter() - t0) * 1e3, st.bvhBuildSeconds * 1e3, st.triangleCount, len(world.model_instances)), flush=True)
    n = len(world.model_instances)
    for moved in (1, 4, n):
        times = []
        for rep in range(4):
            for i in range(moved):
                k = (rep * 7 + i * 3) % n
                model, m = world.model_instances[k]
                world.model_instances[k] = (model, translate((0.01 * (rep + 1), 0.0, 0.0)) @ m)
            t0 = time.perf_counter()
            ctx.update_transforms(world)
            wall = (time.perf_counter() - t0) * 1e3
            st = ctx.scene_stats()
            times.append((wall, st.buildSeconds * 1e3, st.bvhBuildSeconds * 1e3))
        w, b, h = (sorted(x)[len(x) // 2] for x in zip(*times))
        print("moved %2d of %d instances: %.1f ms wall (python freeze included), library %.1f ms, hierarchy rebuild %.1f ms" % (
            moved, n, w, b, h), flush=True)


if __name__ == "__main__":
    main()
