"""The reference's one bundled asset as a packed fixture: res/glTF/FlightHelmet (prosper's default scene,
src/main.cpp:32-33; 94 722 triangles, 6 materials, one of them BLEND - the lenses).

tests/golden/flight_helmet.npz is written by tests/golden/make_flight_helmet.py from the reference's files through
prosper_amd.gltf (index conventions, LIFO node order, TRS rules of WorldData.cpp:756-1543): the geometry buffer is the
packed blob as prosper lays it out (u16 indices, fp16x4 positions, snorm10 normals / tangents, fp16x2 uv;
DeferredLoadingContext.cpp:442-490,775-784), byte for byte.  Textures: the ten images the mount holds are box-filtered
to 64 x 64 (the 2048^2 originals are 25 MB; fixture budget 2 MB), the five it lacks (.MISSING_LARGE_BLOBS) are 1x1 white
as in the ingest.  Lights: none in the asset, so the default sun stays (lights.h:9,18-19); load_fixture adds the
procedural sky of the synthetic scenes for IBL (res/env/storm.ktx is not in the mount either)."""
import os

from . import world_io

FIXTURE = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "flight_helmet.npz")


def load_fixture(path=FIXTURE, sky_size=64, texture_size=None):
    """texture_size: every texture of the fixture (64 x 64 box-filtered images, 1 x 1 stand-ins) blown up to that many texels
    a side by replication - texture_size=2048 gives the asset the texel FOOTPRINT prosper loads it with (15 x 2048^2 RGBA8 =
    252 MB at level 0), which is what the texel fetches of the shade kernel see; the values are the fixture's."""
    from . import scenes
    w = world_io.load_world(path)
    if texture_size:
        import numpy as np
        for i, t in enumerate(w.textures):
            if i == 0:
                continue  # the default texture stays what it is
            t = np.asarray(t)
            fy, fx = max(1, texture_size // t.shape[0]), max(1, texture_size // t.shape[1])
            w.textures[i] = np.ascontiguousarray(np.repeat(np.repeat(t, fy, axis=0), fx, axis=1))
        w._frozen = None
    if sky_size:
        w.skybox = scenes.sky_cube(sky_size)
    # a view that fills the frame with the helmet (the asset is ~0.7 units tall around the origin); prosper's default
    # camera (Camera.hpp:22-48: eye (1, 0.5, 1) -> origin, 59 deg) sees it from three times as far
    w.camera = dict(eye=(0.30, 0.12, 0.42), target=(0.0, 0.0, 0.0), up=(0.0, 1.0, 0.0), fov=w.camera["fov"],
                    zN=w.camera["zN"], zF=w.camera["zF"])
    return w
