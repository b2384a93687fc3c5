"""ctypes binding of oracle/libprosper_oracle.so — TEST INFRASTRUCTURE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the
product package (prosper_amd) never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

from prosper_amd import structs as S

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libprosper_oracle.so")


class OraCounters(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in (
        "paths", "closestRays", "shadowRays", "closestHits", "lightSamples", "spotLightSamples", "skyLookups",
        "pixelsWritten", "historyReads")]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


FN = dict(SINCOS=0, POW=1, SRGB_TO_LINEAR=2, NORMALIZE=3, UNPACK_SNORM=4, ONB=5, COSINE_SAMPLE=6, VNDF_SAMPLE=7,
          VNDF_PDF=8, EVAL_BRDF=9, OFFSET_RAY=10, POINT_LIGHT=11, SPOT_LIGHT=12, TRIANGLE=13, HALF=14, RNG=15)
# (in_stride, out_stride) per function id
FN_SHAPES = {0: (1, 2), 1: (2, 1), 2: (1, 1), 3: (3, 3), 4: (1, 4), 5: (3, 9), 6: (5, 3), 7: (6, 3), 8: (7, 1),
             9: (14, 3), 10: (6, 3), 11: (10, 7), 12: (14, 7), 13: (17, 4), 14: (1, 2), 15: (3, 4)}


def build(force=False):
    """Compile the oracle with gcc (oracle/Makefile)."""
    if force or not os.path.exists(_LIB_PATH):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        L.ora_scene_create.restype = C.c_void_p
        L.ora_scene_create.argtypes = [C.POINTER(S.SceneView), C.c_int]
        L.ora_scene_destroy.argtypes = [C.c_void_p]
        L.ora_scene_triangle_count.restype = C.c_uint64
        L.ora_scene_triangle_count.argtypes = [C.c_void_p]
        L.ora_scene_set_literal_glsl.restype = None
        L.ora_scene_set_literal_glsl.argtypes = [C.c_void_p, C.c_int]
        L.ora_render.argtypes = [
            C.c_void_p, C.POINTER(S.ReferencePC), C.POINTER(S.CameraUniforms), C.c_uint32, C.c_uint32,
            C.POINTER(S.TileDesc), C.c_void_p, C.c_int, C.POINTER(OraCounters)]
        L.ora_trace_closest.restype = C.c_int
        L.ora_trace_closest.argtypes = [
            C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_float, C.c_float, C.c_uint32,
            C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_float)]
        L.ora_trace_shadow.restype = C.c_int
        L.ora_trace_shadow.argtypes = [
            C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_float, C.c_float, C.c_uint32]
        L.ora_pcg.restype = C.c_uint32
        L.ora_pcg.argtypes = [C.c_uint32]
        L.ora_pcg3d.argtypes = [C.POINTER(C.c_uint32)]
        L.ora_pack_half.restype = C.c_uint16
        L.ora_pack_half.argtypes = [C.c_float]
        L.ora_unpack_half.restype = C.c_float
        L.ora_unpack_half.argtypes = [C.c_uint16]
        L.ora_pack_snorm3x10_1x2.restype = C.c_uint32
        L.ora_pack_snorm3x10_1x2.argtypes = [C.POINTER(C.c_float)]
        L.ora_restir_di_trace.restype = None
        L.ora_restir_di_trace.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(S.CameraUniforms), C.c_uint32, C.c_uint32,
                                          C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        L.ora_tone_map.restype = None
        L.ora_tone_map.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_float, C.c_float, C.c_void_p, C.c_uint64]
        L.ora_eval_fn.restype = C.c_int
        L.ora_eval_fn.argtypes = [C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_uint32]
        L.ora_camera_uniforms.argtypes = [
            C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_float, C.c_float, C.c_float,
            C.c_uint32, C.c_uint32, C.POINTER(S.CameraUniforms), C.POINTER(C.c_float)]
        L.ora_pack_mesh.argtypes = [C.c_void_p] * 4 + [C.c_uint32] + [C.c_void_p] * 4
        _lib = L
    return _lib


class OracleScene:
    def __init__(self, world, brute_force=False):
        self.world = world  # keeps every borrowed array alive
        self._view = world.view()
        self._h = lib().ora_scene_create(C.byref(self._view), 1 if brute_force else 0)
        if not self._h:
            raise MemoryError("ora_scene_create failed")

    def close(self):
        if self._h:
            lib().ora_scene_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_literal_glsl(self, on):
        """main.rgen:241-283 as written: no zero-throughput rule (oracle.h ora_scene_set_literal_glsl)."""
        lib().ora_scene_set_literal_glsl(self._h, 1 if on else 0)

    @property
    def triangle_count(self):
        return int(lib().ora_scene_triangle_count(self._h))

    def render(self, pc, camera, width, height, history=None, tile=None, threads=0):
        """One accumulated frame; returns (rgba float32 [h, localW, 4], counters)."""
        if history is None:
            lw = local_width(width, tile)
            history = np.zeros((height, lw, 4), np.float32)
        img = np.ascontiguousarray(history, dtype=np.float32)
        counters = OraCounters()
        lib().ora_render(self._h, C.byref(pc), C.byref(camera), width, height,
                         C.byref(tile) if tile is not None else None, img.ctypes.data, threads, C.byref(counters))
        return img, counters

    def restir_di_trace(self, pc, camera, albedo_roughness, normal_metallic, depth, reservoirs, history=None, threads=0):
        """oracle.c ora_restir_di_trace; pc = (drawType, frameIndex, flags); returns rgba float32 [h, w, 4]."""
        ar = np.ascontiguousarray(albedo_roughness, np.float32)
        nm = np.ascontiguousarray(normal_metallic, np.float32)
        dp = np.ascontiguousarray(depth, np.float32)
        rs = np.ascontiguousarray(reservoirs, np.float32)
        h, w = dp.shape
        img = np.zeros((h, w, 4), np.float32) if history is None else np.ascontiguousarray(history, np.float32).copy()
        cpc = (C.c_uint32 * 3)(*pc)
        lib().ora_restir_di_trace(self._h, cpc, C.byref(camera), w, h, ar.ctypes.data, nm.ctypes.data, dp.ctypes.data,
                                  rs.ctypes.data, img.ctypes.data, threads)
        return img

    def trace_closest(self, origin, direction, t_min=0.0, t_max=float("inf"), seed=0):
        o = (C.c_float * 3)(*origin)
        d = (C.c_float * 3)(*direction)
        di, prim = C.c_uint32(), C.c_uint32()
        bary = (C.c_float * 2)()
        hit = lib().ora_trace_closest(self._h, o, d, t_min, t_max, seed, C.byref(di), C.byref(prim), bary)
        return bool(hit), di.value, prim.value, (bary[0], bary[1])

    def trace_shadow(self, origin, direction, t_min, t_max, seed=0):
        o = (C.c_float * 3)(*origin)
        d = (C.c_float * 3)(*direction)
        return bool(lib().ora_trace_shadow(self._h, o, d, t_min, t_max, seed))


def local_width(width, tile):
    if tile is None or tile.stripeCount <= 1 or tile.stripeWidth == 0:
        return width
    n = 0
    for x in range(0, width, tile.stripeWidth):
        if (x // tile.stripeWidth) % tile.stripeCount == tile.stripeIndex:
            n += min(tile.stripeWidth, width - x)
    return n


def eval_fn(fn, inputs):
    """Evaluate oracle function `fn` (name or id) over a float32 [n, in_stride] array."""
    fid = FN[fn] if isinstance(fn, str) else fn
    in_stride, out_stride = FN_SHAPES[fid]
    a = np.ascontiguousarray(inputs, dtype=np.float32).reshape(-1, in_stride)
    out = np.zeros((a.shape[0], out_stride), np.float32)
    rc = lib().ora_eval_fn(fid, a.ctypes.data, in_stride, out.ctypes.data, out_stride, a.shape[0])
    if rc != 0:
        raise ValueError("unknown oracle fn %r" % (fn,))
    return out


def tone_map(hdr, lut_r9g9b9e5, exposure=1.0, contrast=1.0):
    """oracle.c ora_tone_map: RGBA32F [h, w, 4] + uint32 [dim, dim, dim] R9G9B9E5 LUT -> uint8 [h, w, 4]."""
    hdr = np.ascontiguousarray(hdr, dtype=np.float32)
    lut = np.ascontiguousarray(lut_r9g9b9e5, dtype=np.uint32)
    assert hdr.shape[-1] == 4 and lut.ndim == 3 and lut.shape[0] == lut.shape[1] == lut.shape[2]
    out = np.empty(hdr.shape, np.uint8)
    lib().ora_tone_map(hdr.ctypes.data, lut.ctypes.data, lut.shape[0], exposure, contrast, out.ctypes.data, hdr.size // 4)
    return out


def camera_uniforms(eye, target, up, fov, zN, zF, width, height):
    cam = S.CameraUniforms()
    fl = C.c_float()
    lib().ora_camera_uniforms((C.c_float * 3)(*eye), (C.c_float * 3)(*target), (C.c_float * 3)(*up), fov, zN, zF,
                              width, height, C.byref(cam), C.byref(fl))
    return cam, fl.value
