"""Python handles onto the C++ host layer (prosper_amd/csrc/host): `Camera` and `RtReference`.

Same names, argument meaning and error behaviour as prosper's `scene::Camera`
(src/scene/Camera.hpp) and `render::RtReference` (src/render/RtReference.hpp:32-60); every call
goes through libprosper_pt.so (no Python re-implementation of the pass).
"""
import ctypes as C
import os

import numpy as np

from . import structs as S
from .capi import Context, ProsperPtError, RecordOptions, lib


class Camera:
    def __init__(self):
        self._h = C.c_void_p(lib().prosper_host_camera_create())
        if not self._h:
            raise MemoryError("prosper_host_camera_create failed")

    def close(self):
        if getattr(self, "_h", None):
            lib().prosper_host_camera_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def look_at(self, eye, target, up=(0.0, 1.0, 0.0)):
        f3 = C.c_float * 3
        lib().prosper_host_camera_look_at(self._h, f3(*eye), f3(*target), f3(*up))

    def set_parameters(self, fov, zN=0.1, zF=100.0, aperture_diameter=0.00001, focus_distance=1.0):
        lib().prosper_host_camera_set_parameters(self._h, fov, zN, zF, aperture_diameter, focus_distance)

    def update_resolution(self, width, height):
        lib().prosper_host_camera_update_resolution(self._h, width, height)

    def update_buffer(self):
        """Camera::updateBuffer -> (CameraUniforms, focalLength)"""
        u = S.CameraUniforms()
        fl = C.c_float()
        lib().prosper_host_camera_update_buffer(self._h, C.byref(u), C.byref(fl))
        return u, fl.value

    def changed_this_frame(self):
        return bool(lib().prosper_host_camera_changed_this_frame(self._h))

    def end_frame(self):
        lib().prosper_host_camera_end_frame(self._h)

    @classmethod
    def from_world(cls, world, width, height):
        cam = cls()
        c = world.camera
        cam.set_parameters(c["fov"], c["zN"], c["zF"])
        cam.look_at(c["eye"], c["target"], c["up"])
        cam.update_resolution(width, height)
        return cam


class RtReference:
    """render::RtReference: init / recompileShaders / drawUi / record / releasePreserved."""

    sMaxBounces = S.RT_MAX_BOUNCES

    class Options:
        def __init__(self, depthOfField=False, ibl=False, colorDirty=False, drawType="Default"):
            self.depthOfField = depthOfField
            self.ibl = ibl
            self.colorDirty = colorDirty
            self.drawType = drawType

    def __init__(self):
        self._h = None
        self._ctx = None
        self._world = None

    def init(self, device=0, flags=0):
        h = C.c_void_p()
        rc = lib().prosper_host_rt_reference_create(device, flags, C.byref(h))
        if rc != 0:
            raise ProsperPtError(rc, lib().prosper_host_last_error().decode())
        self._h = h
        self._ctx = Context(_borrowed=lib().prosper_host_rt_reference_context(h))

    def close(self):
        if self._h:
            lib().prosper_host_rt_reference_destroy(self._h)
            self._h = None
            self._ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def context(self):
        return self._ctx

    def set_world(self, world):
        """World::buildAccelerationStructures for this pass's GPU (App.cpp:573-578)."""
        view = world.view()
        rc = lib().prosper_host_rt_reference_set_scene(self._h, C.byref(view))
        if rc != 0:
            raise ProsperPtError(rc, lib().prosper_host_last_error().decode())
        self._world = world

    def recompile_shaders(self):
        lib().prosper_host_rt_reference_recompile_shaders(self._h)

    def draw_ui(self, accumulate=True, clampIndirect=True, rouletteStartBounce=3, maxBounces=S.RT_MAX_BOUNCES):
        lib().prosper_host_rt_reference_draw_ui(self._h, int(accumulate), int(clampIndirect), rouletteStartBounce,
                                                maxBounces)

    def record(self, camera, width, height, options=None, frame_count=1, tile=None, render_flags=0, stream=None):
        """Camera::updateBuffer + RtReference::record; returns the ReferencePC that was pushed."""
        options = options or RtReference.Options()
        o = RecordOptions(int(options.depthOfField), int(options.ibl), int(options.colorDirty),
                          S.DrawType[options.drawType] if isinstance(options.drawType, str) else int(options.drawType))
        pc = S.ReferencePC()
        rc = lib().prosper_host_rt_reference_record(
            self._h, camera._h, width, height, C.byref(o), frame_count, C.byref(tile) if tile is not None else None,
            render_flags, C.c_void_p(stream), C.byref(pc))
        if rc != 0:
            raise ProsperPtError(rc, lib().prosper_host_last_error().decode())
        return pc

    def release_preserved(self):
        lib().prosper_host_rt_reference_release_preserved(self._h)


class TiledRtReference:
    """render::TiledRtReference (csrc/host/tiled_rt_reference.hpp): one rank of a multi-GPU job.  record() renders the
    rank's stripes and enqueues the RCCL gather + de-interleave to the root; read_gathered() there returns the image."""

    @staticmethod
    def create_comm_id():
        return Context.comm_unique_id()

    def __init__(self, device, rank, ranks, comm_id=None, root=0, flags=0):
        h = C.c_void_p()
        buf = (C.c_uint8 * 128).from_buffer_copy(bytes(comm_id) if comm_id is not None else bytes(128))
        rc = lib().prosper_host_tiled_rt_reference_create(device, rank, ranks, buf, root, flags, C.byref(h))
        if rc != 0:
            raise ProsperPtError(rc, lib().prosper_host_last_error().decode())
        self._h = h
        self._ctx = Context(_borrowed=lib().prosper_host_tiled_rt_reference_context(h))
        self.rank, self.ranks, self.root = rank, ranks, root
        self._world = None

    @property
    def context(self):
        return self._ctx

    def set_world(self, world):
        view = world.view()
        rc = lib().prosper_host_tiled_rt_reference_set_scene(self._h, C.byref(view))
        if rc != 0:
            raise ProsperPtError(rc, lib().prosper_host_last_error().decode())
        self._world = world

    def record(self, camera, width, height, options=None, frame_count=1, render_flags=0, stream=None):
        """-> device pointer of the gathered image on the root (None elsewhere)."""
        options = options or RtReference.Options()
        o = RecordOptions(int(options.depthOfField), int(options.ibl), int(options.colorDirty),
                          S.DrawType[options.drawType] if isinstance(options.drawType, str) else int(options.drawType))
        out = C.POINTER(C.c_float)()
        rc = lib().prosper_host_tiled_rt_reference_record(
            self._h, camera._h, width, height, C.byref(o), frame_count, render_flags, C.c_void_p(stream), C.byref(out))
        if rc != 0:
            raise ProsperPtError(rc, lib().prosper_host_last_error().decode())
        return C.cast(out, C.c_void_p).value

    def wait_for_gather(self, stream=None):
        rc = lib().prosper_host_tiled_rt_reference_wait_for_gather(self._h, C.c_void_p(stream))
        if rc != 0:
            raise ProsperPtError(rc, lib().prosper_host_last_error().decode())

    def close(self):
        if self._h:
            lib().prosper_host_tiled_rt_reference_destroy(self._h)
            self._h = None
            self._ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class ToneMap:
    """Handle on the C++ render::ToneMap (csrc/host/tone_map.hpp; reference src/render/ToneMap.hpp)."""

    def __init__(self, ctx, lut_path=None, lut_texels=None):
        h = C.c_void_p()
        if lut_path is not None:
            rc = lib().prosper_host_tone_map_create(ctx._h, os.fsencode(lut_path), C.byref(h))
        else:
            lut = np.ascontiguousarray(lut_texels, dtype=np.uint32)
            rc = lib().prosper_host_tone_map_create_from_texels(ctx._h, lut.ctypes.data, lut.shape[0], C.byref(h))
        if rc != 0:
            raise ProsperPtError(rc, lib().prosper_host_last_error().decode())
        self._h = h

    def draw_ui(self, exposure, contrast):
        lib().prosper_host_tone_map_draw_ui(self._h, exposure, contrast)

    def record(self, device_ptr, byte_size, stream=None):
        rc = lib().prosper_host_tone_map_record(self._h, C.c_void_p(stream), C.c_void_p(device_ptr), byte_size)
        if rc != 0:
            raise ProsperPtError(rc, lib().prosper_host_last_error().decode())

    def close(self):
        if self._h:
            lib().prosper_host_tone_map_destroy(self._h)
            self._h = None
