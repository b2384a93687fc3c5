"""prosper_amd/prosper_headless (csrc/host/headless_main.cpp): a plain C++ program that drives scene::World, scene::Camera and
render::TiledRtReference the way prosper's App drives its passes (App.cpp:516-578, Renderer.cpp:406-428), packing its own scene in the
reference's geometry formats.  The same scene built through the Python handles and rendered by the oracle must give the file it writes,
bit for bit: the C++ surface is a drop-in by itself, not only through ctypes."""
import os
import subprocess

import numpy as np
import pytest

from conftest import same_bits
from prosper_amd import scenes, structs as S
from prosper_amd.world import World

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BINARY = os.path.join(ROOT, "prosper_amd", "prosper_headless")
W, H, FRAMES = 200, 112, 3


def python_twin():
    """headless_main.cpp's scene through prosper_amd.world (the quads' normals / tangents / uvs are what scenes.quad derives)."""
    w = World()
    m1 = w.add_material(base_color=(0.75, 0.75, 0.75, 1.0), metallic=0.0, roughness=0.875)
    m2 = w.add_material(base_color=(0.25, 0.5, 0.75, 1.0), metallic=0.0, roughness=0.5)
    m3 = w.add_material(base_color=(0.875, 0.75, 0.5, 1.0), metallic=1.0, roughness=0.25)
    quads = [((-4, 0, 4), (4, 0, 4), (4, 0, -4), (-4, 0, -4)), ((-4, 0, -4), (4, 0, -4), (4, 4, -4), (-4, 4, -4)),
             ((-1, 0.5, 0), (1, 0.5, 0), (1, 2.5, -1), (-1, 2.5, -1))]
    for q, m in zip(quads, (m1, m2, m3)):
        mesh = scenes._add(w, scenes.quad(*q), m)
        w.add_instance(w.add_model([(mesh, m)]))
    w.set_directional_light((1.0, 1.0, 1.0), 2.0, (-1.0, -1.0, -1.0))
    w.point_lights.lights[0].radianceAndRadius = S.Vec4(4.0, 3.0, 2.0, 16.0)
    w.point_lights.lights[0].position = S.Vec4(1.5, 3.0, 1.0, 1.0)
    w.point_lights.count = 1
    return w


def test_headless_binary_is_built_and_links_the_library():
    assert os.path.exists(BINARY), "make -C prosper_amd/csrc builds prosper_amd/prosper_headless"
    out = subprocess.run(["ldd", BINARY], capture_output=True, text=True).stdout
    assert "libprosper_pt.so" in out and "not found" not in out


def test_the_twin_scene_packs_to_the_bytes_the_cpp_program_packs(tmp_path):
    """headless_main.cpp packs its quads by hand (pack_half / pack_snorm); prosper_amd.world packs the twin: the geometry
    buffers must agree word for word (4 vertices + 6 u16 indices + 4 streams per quad = 23 words each)."""
    w = python_twin()
    buf = np.concatenate(w._buffers[0])
    assert buf.size == 3 * 23
    out = str(tmp_path / "geometry.bin")
    run = subprocess.run([BINARY, "1", "1", "0", out], capture_output=True, text=True, timeout=60)  # frames = 0: no GPU involved
    assert run.returncode == 0, run.stderr
    assert np.array_equal(np.fromfile(out, np.uint32), buf)
    md = w.metadatas[2]
    assert (md.indicesOffset, md.positionsOffset, md.normalsOffset, md.tangentsOffset, md.texCoord0sOffset) == (92, 49, 57, 61, 65)
    # the tilted panel's normal (0, 1, 2) / sqrt(5) in snorm10: x 0, y 229, z 457
    assert int(buf[57]) == (229 << 10) | (457 << 20)
    assert int(buf[61]) == 511 | (1 << 30)  # tangent (1, 0, 0), sign +1


@pytest.mark.gpu
def test_cpp_headless_program_matches_the_oracle_bitwise(tmp_path, oracle):
    from prosper_amd.rt_reference import Camera
    out = str(tmp_path / "frame.rgba32f")
    run = subprocess.run([BINARY, str(W), str(H), str(FRAMES), out], capture_output=True, text=True, timeout=300)
    assert run.returncode == 0, run.stderr[-2000:]
    assert "samples per pixel %d" % FRAMES in run.stdout
    got = np.fromfile(out, np.float32).reshape(H, W, 4)
    world = python_twin()
    cam = Camera()  # the C++ scene::Camera with its default parameters, as the program uses it
    cam.look_at((0.0, 2.0, 5.0), (0.0, 1.0, 0.0))
    cam.update_resolution(W, H)
    uniforms, focal = cam.update_buffer()
    osc = oracle.OracleScene(world, brute_force=True)
    want = None
    for frame in range(1, FRAMES + 1):  # RtReference::record: frameIndex pre-incremented, first frame skips history
        flags = S.PC_FLAG_ACCUMULATE | S.PC_FLAG_CLAMP_INDIRECT | (S.PC_FLAG_SKIP_HISTORY if frame == 1 else 0)
        pc = S.ReferencePC(0, flags, frame, 1e-5, 1.0, focal, 3, 4)
        want, _ = osc.render(pc, uniforms, W, H, history=want)
    ok = same_bits(got, want).all(axis=2)
    assert ok.all(), "%d of %d pixels differ" % ((~ok).sum(), ok.size)
