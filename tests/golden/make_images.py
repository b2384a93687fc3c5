#!/usr/bin/env python3
"""Generates tests/golden/cornell_48.npz: whole-image goldens of S-cornell at 48x48 for every
DrawType and for Default radiance (2 accumulated frames, maxBounces 4, IBL on).

These are produced by the build's OWN CPU oracle: they freeze the oracle
against regressions and pin the HIP kernels to it; they are not Vulkan outputs (the reference
cannot run here: "parity unpinned", oracle/oracle.h).

    python tests/golden/make_images.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from oracle import binding as oracle  # noqa: E402
from prosper_amd import scenes, structs as S  # noqa: E402

W = H = 48


def render_all(render_fn):
    """render_fn(pc, history) -> image; shared by the generator and the tests."""
    out = {}
    for name in S.DRAW_TYPES:
        if name in ("Default", "MeshletID"):
            continue
        pc = S.ReferencePC(S.DrawType[name], S.PC_FLAG_SKIP_HISTORY | S.PC_FLAG_ACCUMULATE, 1, 1e-5, 1.0, 0.0, 3, 1)
        out[name] = render_fn(pc, None)
    img = None
    for frame in (1, 2):
        flags = S.PC_FLAG_ACCUMULATE | S.PC_FLAG_CLAMP_INDIRECT | S.PC_FLAG_IBL | (S.PC_FLAG_SKIP_HISTORY if frame == 1 else 0)
        pc = S.ReferencePC(0, flags, frame, 1e-5, 1.0, 0.0, 3, 4)
        img = render_fn(pc, img)
    out["Default"] = img
    return out


def main():
    world = scenes.cornell(with_skybox=True)
    c = world.camera
    cam, _ = oracle.camera_uniforms(c["eye"], c["target"], c["up"], c["fov"], c["zN"], c["zF"], W, H)
    osc = oracle.OracleScene(world, brute_force=True)
    images = render_all(lambda pc, hist: osc.render(pc, cam, W, H, history=hist)[0])
    np.savez_compressed(os.path.join(HERE, "cornell_48.npz"), camera=np.frombuffer(cam, dtype=np.uint8), **images)
    print("wrote cornell_48.npz:", {k: float(v[..., :3].mean()) for k, v in images.items()})


if __name__ == "__main__":
    main()
