#!/usr/bin/env python3
"""Writes tests/golden/flight_helmet.npz from /root/reference/res/glTF/FlightHelmet (see prosper_amd/flight_helmet.py).

    python tests/golden/make_flight_helmet.py

Data only: packed vertex / index streams, struct tables and down-filtered texels - no text of the reference."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from prosper_amd import gltf, world_io  # noqa: E402

SRC = "/root/reference/res/glTF/FlightHelmet/glTF/FlightHelmet.gltf"
TEX = 64


def box_filter(img, size):
    """[H, W, 4] uint8 -> [size, size, 4] uint8: mean over (H/size) x (W/size) blocks, round half to even."""
    h, w, c = img.shape
    if h <= size or w <= size:
        return img
    fy, fx = h // size, w // size
    blocks = img[: fy * size, : fx * size].astype(np.float64).reshape(size, fy, size, fx, c).mean(axis=(1, 3))
    return np.rint(blocks).astype(np.uint8)


def main():
    # the PNG texels, not the (absent) BC7 cache
    world = gltf.load_gltf(SRC, use_texture_cache=False)
    assert world.triangle_count() == 94722
    sizes = [t.shape[:2] for t in world.textures]
    world.textures = [box_filter(t, TEX) for t in world.textures]
    out = os.path.join(HERE, "flight_helmet.npz")
    world_io.save_world(out, world, extra={"source_texture_sizes": np.array(sizes, np.uint32),
                                          "missing_images": np.array(sorted(world.missing_images))})
    print("wrote %s: %.2f MB, %d triangles, %d textures (%d missing in the mount -> 1x1)" % (
        out, os.path.getsize(out) / 1e6, world.triangle_count(), len(world.textures), len(world.missing_images)))


if __name__ == "__main__":
    main()
