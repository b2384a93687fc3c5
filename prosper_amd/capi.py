"""ctypes binding of libprosper_pt.so (include/prosper_pt/prosper_pt.h, prosper_host.h).

There is no Python or CPU fallback behind these calls: if the HIP library is missing, or no
gfx950 device is visible, they raise.
"""
import ctypes as C
import os
import subprocess

import numpy as np

from . import structs as S

_HERE = os.path.dirname(os.path.abspath(__file__))
# PROSPER_PT_LIB (read by this BINDING, i.e. by tests and measurement scripts - the library itself never looks): another
# build of the library, e.g. libprosper_pt_experiments.so (make -C prosper_amd/csrc EXPERIMENTS=1)
LIB_PATH = os.environ.get("PROSPER_PT_LIB") or os.path.join(_HERE, "libprosper_pt.so")


class ProsperPtError(RuntimeError):
    def __init__(self, code, message):
        super().__init__("prosper_pt error %d: %s" % (code, message))
        self.code = code


def build(force=False):
    """Compile the HIP library for gfx950 in-tree (prosper_amd/csrc/Makefile)."""
    args = ["make", "-C", os.path.join(_HERE, "csrc"), "-s"]
    if force:
        args.append("-B")
    subprocess.check_call(args)
    return LIB_PATH


_lib = None


def lib():
    """Loads libprosper_pt.so; raises if it has not been built (never falls back)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ProsperPtError(-2, "HIP extension %s is missing: run prosper_amd.capi.build()" % LIB_PATH)
    L = C.CDLL(LIB_PATH)
    vp, u32, i32 = C.c_void_p, C.c_uint32, C.c_int32
    L.prosper_pt_last_error.restype = C.c_char_p
    L.prosper_pt_abi_version.restype = u32
    L.prosper_pt_has_experiments.restype = u32
    L.prosper_pt_debug_options_default.argtypes = [C.POINTER(S.DebugOptions)]
    L.prosper_pt_debug_options_default.restype = None
    L.prosper_pt_set_debug_options.argtypes = [vp, C.POINTER(S.DebugOptions)]
    L.prosper_pt_get_debug_options.argtypes = [vp, C.POINTER(S.DebugOptions)]
    L.prosper_pt_create.argtypes = [C.POINTER(S.DeviceDesc), C.POINTER(vp)]
    L.prosper_pt_destroy.argtypes = [vp]
    L.prosper_pt_destroy.restype = None
    L.prosper_pt_upload_scene.argtypes = [vp, C.POINTER(S.SceneView)]
    L.prosper_pt_update_lights.argtypes = [
        vp, C.POINTER(S.DirectionalLightParameters), C.POINTER(S.PointLightsBuffer), C.POINTER(S.SpotLightsBuffer)]
    L.prosper_pt_get_scene_stats.argtypes = [vp, C.POINTER(S.SceneStats)]
    L.prosper_pt_update_textures.argtypes = [vp, vp, u32, u32]
    L.prosper_pt_update_materials.argtypes = [vp, vp, u32, u32]
    L.prosper_pt_update_meshes.argtypes = [vp, vp, u32]
    L.prosper_pt_finish_mesh_updates.argtypes = [vp]
    L.prosper_pt_update_transforms.argtypes = [vp, vp, u32]
    L.prosper_pt_update_transforms_async.argtypes = [vp, vp, u32, u32, vp]
    L.prosper_pt_rebuild_hierarchy.argtypes = [vp]
    L.prosper_pt_get_hierarchy_state.argtypes = [vp, C.POINTER(S.HierarchyState)]
    L.prosper_pt_debug_read_nodes.argtypes = [vp, vp, C.c_size_t]
    L.prosper_pt_set_output_buffer.argtypes = [vp, vp, C.c_size_t]
    L.prosper_pt_render.argtypes = [
        vp, C.POINTER(S.ReferencePC), C.POINTER(S.CameraUniforms), u32, u32, C.POINTER(S.TileDesc), u32, vp]
    L.prosper_pt_render_frames.argtypes = [
        vp, C.POINTER(S.ReferencePC), C.POINTER(S.CameraUniforms), u32, u32, C.POINTER(S.TileDesc), u32, u32, vp]
    L.prosper_pt_get_local_extent.argtypes = [vp, C.POINTER(u32), C.POINTER(u32)]
    L.prosper_pt_get_hdr_device_ptr.argtypes = [vp, C.POINTER(vp), C.POINTER(C.c_size_t)]
    L.prosper_pt_read_hdr.argtypes = [vp, vp, C.c_size_t, vp]
    L.prosper_pt_blit_rgba16f.argtypes = [vp, vp, C.c_size_t, vp]
    L.prosper_pt_restir_di_trace.argtypes = [
        vp, C.POINTER(S.RestirTracePC), C.POINTER(S.CameraUniforms), u32, u32, C.POINTER(S.RestirInputs), vp]
    L.prosper_pt_set_tone_map_lut.argtypes = [vp, vp, u32]
    L.prosper_pt_tone_map.argtypes = [vp, C.c_float, C.c_float, vp, vp, C.c_size_t, vp]
    L.prosper_pt_get_counters.argtypes = [vp, C.POINTER(S.Counters), vp]
    L.prosper_pt_reset_counters.argtypes = [vp, vp]
    L.prosper_pt_get_stage_counters.argtypes = [vp, u32, C.POINTER(S.Counters), vp]
    L.prosper_pt_get_last_render_timing.argtypes = [vp, C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(u32)]
    L.prosper_pt_get_last_render_ms.argtypes = [vp, C.POINTER(C.c_float), C.POINTER(C.c_float)]
    L.prosper_pt_kernel_name.argtypes = [u32]
    L.prosper_pt_kernel_name.restype = C.c_char_p
    L.prosper_pt_set_kernel_timing.argtypes = [vp, C.c_int]
    L.prosper_pt_eval_device_fn.argtypes = [vp, u32, vp, u32, vp, u32, u32]
    L.prosper_pt_debug_srgb_monotonicity.argtypes = [vp, u32, u32, C.POINTER(C.c_float), C.POINTER(C.c_uint64)]
    # multi-GPU: stripes + RCCL gather + de-interleave
    L.prosper_pt_comm_get_unique_id.argtypes = [vp]
    L.prosper_pt_comm_init.argtypes = [vp, vp, u32, u32]
    L.prosper_pt_comm_adopt.argtypes = [vp, vp, u32, u32]
    L.prosper_pt_comm_destroy.argtypes = [vp]
    L.prosper_pt_gather_tiles.argtypes = [vp, u32, vp, C.c_size_t, u32, vp]
    L.prosper_pt_gather_wait.argtypes = [vp, vp]
    L.prosper_pt_comm_query.argtypes = [vp, C.POINTER(S.CommInfo)]
    L.prosper_pt_get_gathered_device_ptr.argtypes = [vp, C.POINTER(vp), C.POINTER(u32), C.POINTER(u32)]
    L.prosper_pt_read_gathered.argtypes = [vp, vp, C.c_size_t, vp]
    L.prosper_pt_deinterleave_tiles.argtypes = [vp, vp, u32, u32, u32, u32, vp, C.c_size_t, vp]
    # host layer
    L.prosper_host_last_error.restype = C.c_char_p
    L.prosper_host_camera_create.restype = vp
    L.prosper_host_camera_destroy.argtypes = [vp]
    L.prosper_host_camera_destroy.restype = None
    f3 = C.POINTER(C.c_float)
    L.prosper_host_camera_look_at.argtypes = [vp, f3, f3, f3]
    L.prosper_host_camera_look_at.restype = None
    L.prosper_host_camera_set_parameters.argtypes = [vp] + [C.c_float] * 5
    L.prosper_host_camera_set_parameters.restype = None
    L.prosper_host_camera_update_resolution.argtypes = [vp, u32, u32]
    L.prosper_host_camera_update_resolution.restype = None
    L.prosper_host_camera_update_buffer.argtypes = [vp, C.POINTER(S.CameraUniforms), C.POINTER(C.c_float)]
    L.prosper_host_camera_update_buffer.restype = None
    L.prosper_host_camera_changed_this_frame.argtypes = [vp]
    L.prosper_host_camera_end_frame.argtypes = [vp]
    L.prosper_host_camera_end_frame.restype = None
    L.prosper_host_rt_reference_create.argtypes = [i32, u32, C.POINTER(vp)]
    L.prosper_host_rt_reference_destroy.argtypes = [vp]
    L.prosper_host_rt_reference_destroy.restype = None
    L.prosper_host_rt_reference_context.argtypes = [vp]
    L.prosper_host_rt_reference_context.restype = vp
    L.prosper_host_rt_reference_set_scene.argtypes = [vp, C.POINTER(S.SceneView)]
    L.prosper_host_rt_reference_draw_ui.argtypes = [vp, C.c_int, C.c_int, u32, u32]
    L.prosper_host_rt_reference_draw_ui.restype = None
    L.prosper_host_rt_reference_recompile_shaders.argtypes = [vp]
    L.prosper_host_rt_reference_recompile_shaders.restype = None
    L.prosper_host_rt_reference_release_preserved.argtypes = [vp]
    L.prosper_host_rt_reference_release_preserved.restype = None
    L.prosper_host_rt_reference_record.argtypes = [
        vp, vp, u32, u32, C.POINTER(RecordOptions), u32, C.POINTER(S.TileDesc), u32, vp, C.POINTER(S.ReferencePC)]
    L.prosper_host_tiled_rt_reference_create.argtypes = [i32, u32, u32, vp, u32, u32, C.POINTER(vp)]
    L.prosper_host_tiled_rt_reference_destroy.argtypes = [vp]
    L.prosper_host_tiled_rt_reference_destroy.restype = None
    L.prosper_host_tiled_rt_reference_context.argtypes = [vp]
    L.prosper_host_tiled_rt_reference_context.restype = vp
    L.prosper_host_tiled_rt_reference_set_scene.argtypes = [vp, C.POINTER(S.SceneView)]
    L.prosper_host_tiled_rt_reference_record.argtypes = [
        vp, vp, u32, u32, C.POINTER(RecordOptions), u32, u32, vp, C.POINTER(C.POINTER(C.c_float))]
    L.prosper_host_tiled_rt_reference_wait_for_gather.argtypes = [vp, vp]
    L.prosper_host_tone_map_create.argtypes = [vp, C.c_char_p, C.POINTER(vp)]
    L.prosper_host_tone_map_create_from_texels.argtypes = [vp, vp, u32, C.POINTER(vp)]
    L.prosper_host_tone_map_destroy.argtypes = [vp]
    L.prosper_host_tone_map_destroy.restype = None
    L.prosper_host_tone_map_draw_ui.argtypes = [vp, C.c_float, C.c_float]
    L.prosper_host_tone_map_draw_ui.restype = None
    L.prosper_host_tone_map_record.argtypes = [vp, vp, vp, C.c_size_t]
    _lib = L
    return L


class RecordOptions(C.Structure):
    """prosper_host_record_options == RtReference::Options (src/render/RtReference.hpp:44-50)"""

    _fields_ = [("depthOfField", C.c_uint32), ("ibl", C.c_uint32), ("colorDirty", C.c_uint32),
                ("drawType", C.c_uint32)]


def _check(rc):
    if rc != 0:
        raise ProsperPtError(rc, lib().prosper_pt_last_error().decode())


def _tile_ref(tile):
    return C.byref(tile) if tile is not None else None


def has_experiments():
    """True when the loaded library was built with -DPPT_EXPERIMENTS (the measured-slower variants are compiled in)."""
    return bool(lib().prosper_pt_has_experiments())


# Process-wide debug options of THIS BINDING (tests, sweeps): every Context applies them - on top of the library's defaults
# and under its own set_debug() - before its next upload, update or render.  The library keeps options per context and
# never reads the environment; this dict is what monkeypatch.setenv used to be for the tests.
_debug_defaults = {}
_debug_version = 0


def debug(**options):
    """capi.debug(ldsStackEntries=16) sets, capi.debug(ldsStackEntries=None) clears a process-wide option; capi.debug()
    with no argument clears them all."""
    global _debug_version
    if not options:
        _debug_defaults.clear()
    for k, v in options.items():
        if k not in dict(S.DebugOptions._fields_):
            raise KeyError("unknown debug option %r" % k)
        if v is None:
            _debug_defaults.pop(k, None)
        else:
            _debug_defaults[k] = v
    _debug_version += 1


class Context:
    """One prosper_pt context = one GPU (prosper_pt_create .. prosper_pt_destroy)."""

    def __init__(self, device=0, flags=0, _borrowed=None):
        self._owned = _borrowed is None
        self._own_debug = {}
        self._base_debug = None
        self._debug_seen = -1
        self._world = None
        if _borrowed is not None:
            self._h = C.c_void_p(_borrowed)
            return
        desc = S.DeviceDesc(C.sizeof(S.DeviceDesc), device, flags, 0)
        h = C.c_void_p()
        _check(lib().prosper_pt_create(C.byref(desc), C.byref(h)))
        self._h = h

    def set_debug(self, **options):
        """This context's debug options (prosper_pt_set_debug_options): ctx.set_debug(segments=2560); None clears one,
        no argument clears all.  Applied with the process-wide capi.debug() options before the next call."""
        if not options:
            self._own_debug = {}
        for k, v in options.items():
            if k not in dict(S.DebugOptions._fields_):
                raise KeyError("unknown debug option %r" % k)
            if v is None:
                self._own_debug.pop(k, None)
            else:
                self._own_debug[k] = v
        self._debug_seen = -1
        self._sync_debug()

    def debug_options(self):
        o = S.DebugOptions()
        _check(lib().prosper_pt_get_debug_options(self._h, C.byref(o)))
        return o

    def _sync_debug(self):
        if self._debug_seen == _debug_version:
            return
        if self._base_debug is None:
            # what the context was created with: the library's defaults, or PROSPER_PT_DEBUG_OPTIONS under PROSPER_PT_DEBUG=1
            self._base_debug = self.debug_options()
        o = S.DebugOptions.from_buffer_copy(bytes(self._base_debug))
        for k, v in list(_debug_defaults.items()) + list(self._own_debug.items()):
            setattr(o, k, v)
        _check(lib().prosper_pt_set_debug_options(self._h, C.byref(o)))
        self._debug_seen = _debug_version

    def close(self):
        if getattr(self, "_h", None) and self._owned:
            lib().prosper_pt_destroy(self._h)
        self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def upload_scene(self, world):
        self._sync_debug()
        view = world.view()
        _check(lib().prosper_pt_upload_scene(self._h, C.byref(view)))
        self._world = world

    def update_lights(self, world):
        world.freeze()  # "honor scene lighting": a scene with punctual lights and no sun of its own has none (WorldData.cpp:1537-1542)
        _check(lib().prosper_pt_update_lights(self._h, C.byref(world.directional), C.byref(world.point_lights),
                                              C.byref(world.spot_lights)))

    def update_transforms(self, world, stream=None, now=None):
        """New ModelInstanceTransforms of the uploaded scene from `world` (same scene, moved instances).  The table is
        staged and the refit runs at the head of the next render's own chain of launches; with `now` (default: whenever a
        stream is given) it is enqueued on `stream` by this call instead (PROSPER_PT_UPDATE_NOW; stream None / 0 = the
        null stream)."""
        self._sync_debug()
        world._frozen = None
        f = world.freeze()
        t = f["transforms"]
        if now is None:
            now = stream is not None
        _check(lib().prosper_pt_update_transforms_async(self._h, C.cast(t, C.c_void_p), len(world.model_instances),
                                                         S.UPDATE_NOW if now else 0, C.c_void_p(stream)))
        self._world = world

    def update_textures(self, textures, first):
        """Replaces materialTextures[first .. first + len(textures)): numpy [h, w, 4] uint8 arrays or world.Bc7Texture."""
        self._sync_debug()
        descs = (S.TextureDesc * len(textures))()
        keep = []
        for i, t in enumerate(textures):
            if hasattr(t, "blocks"):
                descs[i].texels, descs[i].width, descs[i].height, descs[i].format = t.blocks.ctypes.data, t.width, t.height, S.FORMAT_BC7_UNORM
                continue
            a = np.ascontiguousarray(t, np.uint8)
            keep.append(a)
            descs[i].texels, descs[i].width, descs[i].height, descs[i].format = a.ctypes.data, a.shape[1], a.shape[0], S.FORMAT_RGBA8_UNORM
        _check(lib().prosper_pt_update_textures(self._h, C.cast(descs, C.c_void_p), first, len(textures)))

    def update_materials(self, materials, first):
        """Replaces MaterialData[first .. first + len(materials)) (structs.MaterialData)."""
        self._sync_debug()
        arr = (S.MaterialData * len(materials))(*materials)
        _check(lib().prosper_pt_update_materials(self._h, C.cast(arr, C.c_void_p), first, len(materials)))

    def update_meshes(self, world, mesh_indices, wait=True):
        """Hands over meshes of `world` (a World that holds them) that the uploaded scene marked as not loaded
        (World.with_meshes_loaded): metadata, MeshInfo and the mesh's bytes of its geometry buffer.  wait: also
        prosper_pt_finish_mesh_updates - the next render shows them (otherwise the first render after the context's worker
        thread has built their geometry does)."""
        self._sync_debug()
        f = world.freeze()
        ups = (S.MeshUpdate * max(1, len(mesh_indices)))()
        for u, i in zip(ups, mesh_indices):
            buffer_index, first_word, words = world.mesh_ranges[i]
            buf = f["geometry_buffers"][buffer_index]
            u.meshIndex = i
            u.metadata = world.metadatas[i]
            u.info = world.mesh_infos[i]
            u.bytes = buf.ctypes.data + 4 * first_word
            u.byteOffset, u.byteCount, u.bufferByteSize = 4 * first_word, 4 * words, buf.nbytes
        _check(lib().prosper_pt_update_meshes(self._h, C.cast(ups, C.c_void_p), len(mesh_indices)))
        if wait:
            self.finish_mesh_updates()

    def finish_mesh_updates(self):
        _check(lib().prosper_pt_finish_mesh_updates(self._h))

    def rebuild_hierarchy(self):
        self._sync_debug()
        _check(lib().prosper_pt_rebuild_hierarchy(self._h))

    def hierarchy_state(self):
        st = S.HierarchyState()
        _check(lib().prosper_pt_get_hierarchy_state(self._h, C.byref(st)))
        return st

    def read_nodes(self):
        """The node array as the device holds it: (nodeCount, 20) uint32 (80-byte nodes)."""
        n = int(self.scene_stats().nodeCount)
        out = np.zeros((n, 20), np.uint32)
        _check(lib().prosper_pt_debug_read_nodes(self._h, out.ctypes.data, out.nbytes))
        return out

    def scene_stats(self):
        self._sync_debug()
        st = S.SceneStats()
        _check(lib().prosper_pt_get_scene_stats(self._h, C.byref(st)))
        return st

    def set_output_buffer(self, device_ptr, byte_size):
        _check(lib().prosper_pt_set_output_buffer(self._h, C.c_void_p(device_ptr), byte_size))

    def render(self, pc, camera, width, height, tile=None, frames=1, flags=0, stream=None):
        self._sync_debug()
        _check(lib().prosper_pt_render_frames(self._h, C.byref(pc), C.byref(camera), width, height, _tile_ref(tile),
                                              frames, flags, C.c_void_p(stream)))

    def local_extent(self):
        lw, h = C.c_uint32(), C.c_uint32()
        _check(lib().prosper_pt_get_local_extent(self._h, C.byref(lw), C.byref(h)))
        return lw.value, h.value

    def hdr_device_ptr(self):
        p, n = C.c_void_p(), C.c_size_t()
        _check(lib().prosper_pt_get_hdr_device_ptr(self._h, C.byref(p), C.byref(n)))
        return p.value, n.value

    def read_hdr(self, stream=None):
        lw, h = self.local_extent()
        out = np.empty((h, lw, 4), np.float32)
        _check(lib().prosper_pt_read_hdr(self._h, out.ctypes.data, out.nbytes, C.c_void_p(stream)))
        return out

    def blit_rgba16f(self, stream=None):
        lw, h = self.local_extent()
        out = np.empty((h, lw, 4), np.float16)
        _check(lib().prosper_pt_blit_rgba16f(self._h, out.ctypes.data, out.nbytes, C.c_void_p(stream)))
        return out

    def restir_di_trace(self, pc, camera, albedo_roughness, normal_metallic, depth, reservoirs, stream=None):
        """ReSTIR-DI trace over host G-buffer arrays ([h, w, 4], [h, w, 4], [h, w], [h, w, 2] float32)."""
        ar = np.ascontiguousarray(albedo_roughness, np.float32)
        nm = np.ascontiguousarray(normal_metallic, np.float32)
        dp = np.ascontiguousarray(depth, np.float32)
        rs = np.ascontiguousarray(reservoirs, np.float32)
        h, w = dp.shape
        assert ar.shape == (h, w, 4) and nm.shape == (h, w, 4) and rs.shape == (h, w, 2)
        inp = S.RestirInputs(ar.ctypes.data, nm.ctypes.data, dp.ctypes.data, rs.ctypes.data, 0, 0)
        _check(lib().prosper_pt_restir_di_trace(self._h, C.byref(pc), C.byref(camera), w, h, C.byref(inp), C.c_void_p(stream)))

    def restir_di_trace_device(self, pc, camera, width, height, ar_ptr, nm_ptr, depth_ptr, res_ptr, stream=None):
        """Same with device-resident inputs (raw device pointers, e.g. torch tensors' data_ptr())."""
        inp = S.RestirInputs(ar_ptr, nm_ptr, depth_ptr, res_ptr, 1, 0)
        _check(lib().prosper_pt_restir_di_trace(self._h, C.byref(pc), C.byref(camera), width, height, C.byref(inp),
                                                C.c_void_p(stream)))

    def set_tone_map_lut(self, lut_r9g9b9e5):
        """lut: uint32 [dim, dim, dim] (z, y, x) R9G9B9E5 texels, e.g. from prosper_amd.dds.read_lut."""
        lut = np.ascontiguousarray(lut_r9g9b9e5, dtype=np.uint32)
        assert lut.ndim == 3 and lut.shape[0] == lut.shape[1] == lut.shape[2]
        _check(lib().prosper_pt_set_tone_map_lut(self._h, lut.ctypes.data, lut.shape[0]))

    def tone_map(self, exposure=1.0, contrast=1.0, device_ptr=None, to_host=True, stream=None):
        """tone_map.comp over the current HDR tile -> uint8 [h, localWidth, 4] (or None with to_host=False)."""
        lw, h = self.local_extent()
        out = np.empty((h, lw, 4), np.uint8) if to_host else None
        _check(lib().prosper_pt_tone_map(self._h, exposure, contrast, C.c_void_p(device_ptr),
                                         out.ctypes.data if to_host else None, lw * h * 4, C.c_void_p(stream)))
        return out

    def counters(self, stream=None):
        c = S.Counters()
        _check(lib().prosper_pt_get_counters(self._h, C.byref(c), C.c_void_p(stream)))
        return c

    def stage_counters(self, stage, stream=None):
        c = S.Counters()
        _check(lib().prosper_pt_get_stage_counters(self._h, stage, C.byref(c), C.c_void_p(stream)))
        return c

    def last_render_timing(self):
        """-> (total_ms, {kernel name: (sum_ms, launches)}) of the last render, from hipEvents."""
        total = C.c_float()
        per = (C.c_float * S.MAX_KERNELS)()
        launches = (C.c_uint32 * S.MAX_KERNELS)()
        _check(lib().prosper_pt_get_last_render_timing(self._h, C.byref(total), per, launches))
        names = [lib().prosper_pt_kernel_name(i).decode() for i in range(S.MAX_KERNELS)]
        return total.value, {n: (per[i], launches[i]) for i, n in enumerate(names) if n}

    def reset_counters(self, stream=None):
        _check(lib().prosper_pt_reset_counters(self._h, C.c_void_p(stream)))

    def set_kernel_timing(self, enabled):
        _check(lib().prosper_pt_set_kernel_timing(self._h, 1 if enabled else 0))

    def last_render_ms(self):
        total = C.c_float()
        per = (C.c_float * S.MAX_KERNELS)()
        _check(lib().prosper_pt_get_last_render_ms(self._h, C.byref(total), per))
        names = [lib().prosper_pt_kernel_name(i).decode() for i in range(S.MAX_KERNELS)]
        return total.value, {n: per[i] for i, n in enumerate(names) if n}

    # ---- multi-GPU (include/prosper_pt/prosper_pt.h, "multi-GPU") ----
    @staticmethod
    def comm_unique_id():
        """ncclGetUniqueId: 128 bytes to hand to every rank's comm_init."""
        buf = (C.c_uint8 * 128)()
        _check(lib().prosper_pt_comm_get_unique_id(buf))
        return bytes(buf)

    def comm_init(self, unique_id, rank, ranks):
        buf = (C.c_uint8 * 128).from_buffer_copy(bytes(unique_id))
        _check(lib().prosper_pt_comm_init(self._h, buf, rank, ranks))

    def comm_destroy(self):
        _check(lib().prosper_pt_comm_destroy(self._h))

    def gather_tiles(self, root=0, device_ptr=None, byte_size=0, flags=0, stream=None):
        """RCCL gather of the ranks' tiles to `root` + de-interleave there (enqueue only)."""
        _check(lib().prosper_pt_gather_tiles(self._h, root, C.c_void_p(device_ptr), byte_size, flags, C.c_void_p(stream)))

    def gather_wait(self, stream=None):
        _check(lib().prosper_pt_gather_wait(self._h, C.c_void_p(stream)))

    def comm_info(self, stream=None):
        """The communicator's own view (ncclCommCount, rank, device) and the last gather's device time (waits for it)."""
        info = S.CommInfo()
        _check(lib().prosper_pt_comm_query(self._h, C.byref(info)))
        return info

    def read_gathered(self, stream=None):
        """Root only: the gathered [height, width, 4] float32 image (synchronises)."""
        p, w, h = C.c_void_p(), C.c_uint32(), C.c_uint32()
        _check(lib().prosper_pt_get_gathered_device_ptr(self._h, C.byref(p), C.byref(w), C.byref(h)))
        out = np.empty((h.value, w.value, 4), np.float32)
        _check(lib().prosper_pt_read_gathered(self._h, out.ctypes.data, out.nbytes, C.c_void_p(stream)))
        return out

    def deinterleave_tiles(self, tiles_ptr, ranks, stripe_width, width, height, full_ptr, stream=None):
        """The root's kernel alone, on raw device pointers."""
        _check(lib().prosper_pt_deinterleave_tiles(self._h, C.c_void_p(tiles_ptr), ranks, stripe_width, width, height,
                                                   C.c_void_p(full_ptr), width * height * 16, C.c_void_p(stream)))

    def srgb_monotonicity(self, first_bits, last_bits):
        """(max defect, decreasing adjacent pairs) of the device's sRGBtoLinear over every float in the bit range."""
        defect, decreases = C.c_float(0.0), C.c_uint64(0)
        _check(lib().prosper_pt_debug_srgb_monotonicity(self._h, first_bits, last_bits, C.byref(defect), C.byref(decreases)))
        return float(defect.value), int(decreases.value)

    def eval_device_fn(self, fn, inputs, in_stride, out_stride):
        a = np.ascontiguousarray(inputs, dtype=np.float32).reshape(-1, in_stride)
        out = np.zeros((a.shape[0], out_stride), np.float32)
        _check(lib().prosper_pt_eval_device_fn(self._h, fn, a.ctypes.data, in_stride, out.ctypes.data, out_stride,
                                               a.shape[0]))
        return out
