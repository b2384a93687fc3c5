#!/usr/bin/env python3
"""World-space triangles of a synthetic scene -> scripts/sbvh_experiment.cpp (would spatial splits pay?).  CPU only.

    python scripts/sbvh_experiment.py [c3|c4|helmet]  > profiles/r03_sbvh_experiment.txt"""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from prosper_amd import scenes  # noqa: E402


def world_triangles(world):
    """(n, 9) float32: fp16-decoded positions x modelToWorld, (drawInstance, primitive) order (World.cpp:480-513)."""
    bufs = [np.concatenate(b) if len(b) else np.zeros(0, np.uint32) for b in world._buffers]
    out = []
    for model, m in world.model_instances:
        for mesh, _ in world.models[model]:
            md, info = world.metadatas[mesh], world.mesh_infos[mesh]
            buf = bufs[md.bufferIndex]
            n_idx = info.indexCount
            if md.usesShortIndices:
                idx = buf.view(np.uint16)[md.indicesOffset: md.indicesOffset + n_idx].astype(np.int64)
            else:
                idx = buf[md.indicesOffset: md.indicesOffset + n_idx].astype(np.int64)
            pos = buf[md.positionsOffset: md.positionsOffset + 2 * info.vertexCount].view(np.float16).reshape(-1, 4)[:, :3].astype(np.float64)
            p = (np.concatenate([pos, np.ones((len(pos), 1))], axis=1) @ np.asarray(m, np.float64).T)[:, :3]
            out.append(p[idx].reshape(-1, 9).astype(np.float32))
    return np.concatenate(out)


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "c3"
    world = {"c3": lambda: scenes.sponza_class(texture_size=8, sky_size=8),
             "c4": lambda: scenes.sponza_class(lights=True, foliage=True, texture_size=8, sky_size=8),
             "helmet": lambda: __import__("prosper_amd.flight_helmet", fromlist=["x"]).load_fixture()}[name]()
    tris = world_triangles(world)
    assert len(tris) == world.triangle_count()
    path = "/tmp/sbvh_%s.bin" % name
    tris.tofile(path)
    exe = "/tmp/sbvh_experiment"
    subprocess.check_call(["g++", "-O3", "-std=c++17", os.path.join(ROOT, "scripts", "sbvh_experiment.cpp"), "-o", exe])
    ext = tris.reshape(-1, 3, 3).max(axis=1) - tris.reshape(-1, 3, 3).min(axis=1)
    scene = tris.reshape(-1, 3).max(axis=0) - tris.reshape(-1, 3).min(axis=0)
    print("%s: %d triangles; median / 99th-percentile / largest triangle box edge %.3f / %.3f / %.3f of a %.1f x %.1f x %.1f scene" % (
        name, len(tris), np.median(ext.max(axis=1)), np.percentile(ext.max(axis=1), 99), ext.max(), *scene), flush=True)
    subprocess.check_call([exe, path])


if __name__ == "__main__":
    main()
