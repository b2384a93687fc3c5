"""World <-> .npz: a packed scene (the T3-T9 tables prosper's scene::World hands to the pass) as one file
of plain arrays, so that an asset prepared where its source files are (prosper_amd.gltf) can be rendered where they
are not.  Geometry stays in the reference's packed blob format (DeferredLoadingContext.cpp:775-784): no re-packing on
load, the bytes the kernels read are the bytes in the file."""
import ctypes as C

import numpy as np

from . import structs as S
from .world import Bc7Texture, World


def _struct_bytes(items, ctype):
    arr = (ctype * max(1, len(items)))(*items)
    return np.frombuffer(bytes(arr), np.uint8)[: C.sizeof(ctype) * len(items)].copy()


def save_world(path, world, extra=None):
    world.freeze()
    d = {
        "format": np.array([1], np.uint32),
        "buffer_count": np.array([len(world._buffers)], np.uint32),
        "metadatas": _struct_bytes(world.metadatas, S.GeometryMetadata),
        "mesh_infos": _struct_bytes(world.mesh_infos, S.MeshInfo),
        "materials": _struct_bytes(world.materials, S.MaterialData),
        "samplers": np.array(world.samplers, np.uint32).reshape(-1, 4),
        "model_sizes": np.array([len(m) for m in world.models], np.uint32),
        "sub_models": np.array([sm for m in world.models for sm in m], np.uint32).reshape(-1, 2),
        "instance_models": np.array([mi for mi, _ in world.model_instances], np.uint32),
        "instance_transforms": np.array([m for _, m in world.model_instances], np.float64).reshape(-1, 4, 4),
        "directional": np.frombuffer(bytes(world.directional), np.uint8).copy(),
        "directional_found": np.array([1 if world._directional_found else 0], np.uint32),
        "point_lights": np.frombuffer(bytes(world.point_lights), np.uint8)[: 32 * world.point_lights.count].copy(),
        "spot_lights": np.frombuffer(bytes(world.spot_lights), np.uint8)[: 48 * world.spot_lights.count].copy(),
        "camera": np.array(list(world.camera["eye"]) + list(world.camera["target"]) + list(world.camera["up"]) +
                           [world.camera["fov"], world.camera["zN"], world.camera["zF"]], np.float64),
        "texture_count": np.array([len(world.textures)], np.uint32),
    }
    for i, parts in enumerate(world._buffers):
        d["buffer_%d" % i] = np.concatenate(parts) if parts else np.zeros(0, np.uint32)
    for i, t in enumerate(world.textures):
        if isinstance(t, Bc7Texture):
            d["texture_%d_bc7" % i] = t.blocks
            d["texture_%d_size" % i] = np.array([t.width, t.height], np.uint32)
        else:
            d["texture_%d" % i] = t
    if world.skybox is not None:
        d["skybox"] = np.ascontiguousarray(world.skybox, np.float16)
    for k, v in (extra or {}).items():
        d["extra_" + k] = np.asarray(v)
    np.savez_compressed(path, **d)


def _structs(raw, ctype):
    n = raw.size // C.sizeof(ctype)
    return [ctype.from_buffer_copy(raw[i * C.sizeof(ctype):(i + 1) * C.sizeof(ctype)].tobytes()) for i in range(n)]


def load_world(path):
    z = np.load(path)
    assert int(z["format"][0]) == 1
    w = World()
    nb = int(z["buffer_count"][0])
    w._buffers = [[np.ascontiguousarray(z["buffer_%d" % i], np.uint32)] for i in range(nb)]
    w._buffer_words = [int(b[0].size) for b in w._buffers]
    w.metadatas = _structs(z["metadatas"], S.GeometryMetadata)
    w.mesh_infos = _structs(z["mesh_infos"], S.MeshInfo)
    w.materials = _structs(z["materials"], S.MaterialData)
    w.samplers = [tuple(int(x) for x in row) for row in z["samplers"]]
    subs = [tuple(int(x) for x in row) for row in z["sub_models"]]
    w.models, k = [], 0
    for n in z["model_sizes"]:
        w.models.append(subs[k:k + int(n)])
        k += int(n)
    w.model_instances = [(int(mi), np.array(m, np.float64)) for mi, m in zip(z["instance_models"], z["instance_transforms"])]
    w.directional = S.DirectionalLightParameters.from_buffer_copy(z["directional"].tobytes())
    w._directional_found = bool(z["directional_found"][0])
    pl, sl = z["point_lights"].tobytes(), z["spot_lights"].tobytes()
    w.point_lights = S.PointLightsBuffer()
    C.memmove(C.byref(w.point_lights), pl, len(pl))
    w.point_lights.count = len(pl) // 32
    w.spot_lights = S.SpotLightsBuffer()
    C.memmove(C.byref(w.spot_lights), sl, len(sl))
    w.spot_lights.count = len(sl) // 48
    c = z["camera"]
    w.camera = dict(eye=tuple(c[0:3]), target=tuple(c[3:6]), up=tuple(c[6:9]), fov=float(c[9]), zN=float(c[10]), zF=float(c[11]))
    w.textures = []
    for i in range(int(z["texture_count"][0])):
        if "texture_%d_bc7" % i in z.files:
            wd, ht = (int(x) for x in z["texture_%d_size" % i])
            w.textures.append(Bc7Texture(np.ascontiguousarray(z["texture_%d_bc7" % i], np.uint8), wd, ht))
        else:
            w.textures.append(np.ascontiguousarray(z["texture_%d" % i], np.uint8))
    if "skybox" in z.files:
        w.skybox = np.ascontiguousarray(z["skybox"], np.float16)
    w.extra = {k[6:]: z[k] for k in z.files if k.startswith("extra_")}
    return w
