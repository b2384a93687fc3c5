"""The host-side hierarchy builder (prosper_amd/csrc/bvh_build.cpp: per-instance subtrees built on the host's threads, the
re-braided top level, the 4-wide emitter, rebuilds after a moved instance, subtrees kept while meshes stream in) under AddressSanitizer + UBSan and under
ThreadSanitizer, through scripts/bvh_bench.cpp on a small synthetic scene.  CPU only: sanitizers cannot run on the GPU box."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("sanitizer", ["address,undefined", "thread"])
def test_bvh_builder_under_sanitizers(tmp_path, sanitizer):
    gxx = shutil.which("g++")
    if gxx is None:
        pytest.skip("no g++")
    exe = str(tmp_path / "bvh_bench")
    cmd = [gxx, "-O1", "-g", "-std=c++17", "-fsanitize=" + sanitizer, "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer",
           "-I", os.path.join(ROOT, "prosper_amd", "csrc"), "-I", os.path.join(ROOT, "include"), "-I", "/opt/rocm/include",
           "-D__HIP_PLATFORM_AMD__", os.path.join(ROOT, "scripts", "bvh_bench.cpp"),
           os.path.join(ROOT, "prosper_amd", "csrc", "bvh_build.cpp"), "-lpthread", "-o", exe]
    subprocess.check_call(cmd)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", TSAN_OPTIONS="halt_on_error=1")
    digests = {}
    for threads in ("1", "4"):
        out = subprocess.run([exe, "12", threads], env=env, capture_output=True, text=True, timeout=600)
        text = out.stdout + out.stderr
        assert out.returncode == 0, text[-2000:]
        assert "runtime error" not in text and "Sanitizer" not in text, text[-2000:]
        assert "flat build" in text
        assert text.count("equal to a fresh build") == 2  # InstancedBvh::adopt (streamed-in meshes)
        digests[threads] = [line.split("digest")[1].strip() for line in text.splitlines() if "digest" in line]
    # the trees (nodes + permutation) do not depend on how many threads built them: worker pool, deferred subranges,
    # the one-thread-per-instance and all-threads-per-instance paths
    assert digests["1"] == digests["4"] and len(digests["1"]) >= 5


_ORACLE_RENDER = r"""
import sys, hashlib
sys.path.insert(0, {root!r})
import oracle.binding as ob
if len(sys.argv) > 1:
    ob._LIB_PATH = sys.argv[1]
oracle = ob
from prosper_amd import scenes, structs as S
from prosper_amd.rt_reference import Camera
h = hashlib.sha256()
for world, ibl in ((scenes.cornell(), False),
                   (scenes.sponza_class(lights=(8, 8), foliage=True, texture_size=32, sky_size=16, detail=0.25), True)):
    w, hgt = 64, 36
    cam, focal = Camera.from_world(world, w, hgt).update_buffer()
    osc = oracle.OracleScene(world)
    img = None
    for frame in (1, 2):
        flags = S.PC_FLAG_ACCUMULATE | S.PC_FLAG_CLAMP_INDIRECT | (S.PC_FLAG_IBL if ibl else 0) | (S.PC_FLAG_SKIP_HISTORY if frame == 1 else 0)
        pc = S.ReferencePC(0, flags, frame, 1e-5, 1.0, focal, 3, 3)
        img, _ = osc.render(pc, cam, w, hgt, history=img)
    h.update(img.tobytes())
print("digest", h.hexdigest())
"""


def test_oracle_under_sanitizers(tmp_path):
    """The checker itself: oracle/oracle.c built with AddressSanitizer + UBSan renders a lit, alpha-tested, textured
    scene and the Cornell box without a report, and to the bits of the optimised build."""
    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("no gcc")
    asan = subprocess.run([gcc, "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not os.path.isabs(asan) or not os.path.exists(asan):
        pytest.skip("no libasan")
    lib = str(tmp_path / "liboracle_san.so")
    subprocess.check_call([gcc, "-O1", "-g", "-std=c11", "-fPIC", "-fopenmp", "-ffp-contract=off", "-fno-fast-math",
                           "-march=x86-64-v3", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-shared",
                           "-o", lib, os.path.join(ROOT, "oracle", "oracle.c"), "-lm"])
    script = str(tmp_path / "render.py")
    with open(script, "w") as f:
        f.write(_ORACLE_RENDER.format(root=ROOT))
    import sys
    plain = subprocess.run([sys.executable, script], capture_output=True, text=True, timeout=900)
    assert plain.returncode == 0, plain.stderr[-2000:]
    env = dict(os.environ, LD_PRELOAD=asan, ASAN_OPTIONS="detect_leaks=0", OMP_NUM_THREADS="4")
    san = subprocess.run([sys.executable, script, lib], env=env, capture_output=True, text=True, timeout=900)
    text = san.stdout + san.stderr
    assert san.returncode == 0, text[-3000:]
    assert "runtime error" not in text and "AddressSanitizer" not in text, text[-3000:]
    assert [l for l in san.stdout.splitlines() if l.startswith("digest")] == [l for l in plain.stdout.splitlines() if l.startswith("digest")]
