#!/usr/bin/env python3
"""World-space triangles of a synthetic scene -> scripts/wide_bvh_experiment.cpp (would an 8-wide node pay?).  CPU only.

    python scripts/wide_bvh_experiment.py [c3|c4|helmet] [rays]"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "scripts"))
from prosper_amd import scenes  # noqa: E402
from sbvh_experiment import world_triangles  # noqa: E402


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "c3"
    rays = sys.argv[2] if len(sys.argv) > 2 else "200000"
    world = {"c3": lambda: scenes.sponza_class(texture_size=8, sky_size=8),
             "c4": lambda: scenes.sponza_class(lights=True, foliage=True, texture_size=8, sky_size=8),
             "helmet": lambda: __import__("prosper_amd.flight_helmet", fromlist=["x"]).load_fixture()}[name]()
    tris = world_triangles(world)
    path = "/tmp/wide_%s.bin" % name
    tris.tofile(path)
    exe = "/tmp/wide_bvh_experiment"
    subprocess.check_call(["g++", "-O3", "-std=c++17", os.path.join(ROOT, "scripts", "wide_bvh_experiment.cpp"), "-o", exe])
    print("%s: %d triangles" % (name, len(tris)), flush=True)
    subprocess.check_call([exe, path, rays])


if __name__ == "__main__":
    main()
