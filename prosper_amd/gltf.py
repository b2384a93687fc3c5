"""glTF 2.0 -> World: the asset-side half of the path, following prosper's conventions.

What the reference does with cgltf in `src/scene/WorldData.cpp:681-1543` and
`src/scene/DeferredLoadingContext.cpp:111-156` is restated here on top of `World` (which already restates
`packMeshData` and the blob layout): a minimal JSON + bin (or .glb) reader that produces the same
T3-T9 tables prosper would upload for the file -

  * sampler i -> index i + 1 (0 = repeat/linear default), filters/wraps per `getVkFilterMode` /
    `getVkAddressMode` (WorldData.cpp:183-217);
  * image i -> texture index i + 1 (0 = `empty.png`), glTF *texture* t -> the pair (image + 1, sampler + 1)
    a material's `Texture2DSampler` packs (WorldData.cpp:740-754);
  * material m -> index m + 1 (0 = default material), factors/alpha as `loadMaterials` copies them
    (WorldData.cpp:756-828);
  * every primitive is one mesh (running index over meshes x primitives), a glTF mesh is a Model of
    sub-models (WorldData.cpp:830-915); `usesShortIndices` comes from the vertex count (World.add_mesh);
  * node TRS with the "close to identity" components dropped (threshold 1e-3, WorldData.cpp:1178-1211), the
    scene flattened depth-first with a LIFO stack so that model instances - hence DrawInstances - come out in
    the reference's order (WorldData.cpp:1364-1456, World.cpp:468-513);
  * KHR_lights_punctual: W -> radiance / (4 pi) for point and spot lights, W/m^2 kept for the sun, range or
    sqrt(luminance / 0.01) as radius, the spot's angle scale/offset, -Z as the light axis, and no default sun
    in a scene that only has punctual lights (WorldData.cpp:1458-1542, World.cpp:428-456);
  * the node that carries glTF camera 0 (the reference's current camera), if perspective, sets
    eye/target/up/fov/zN/zF (WorldData.cpp:1148-1173, World.cpp:414-426).

Not restated (third-party arithmetic the survey lists as result-invariant or unpinned, §8c): meshoptimizer's
vertex/index reordering (changes PrimitiveID colours only), mikktspace (a primitive without TANGENT keeps no
tangents here, i.e. its normal map is ignored), BC7 compression of the textures (texels are the decoded
PNG/JPEG bytes), glm's fp32 `inverse`/`decompose` (float64 here, rounded once).
"""
import base64
import json
import math
import os
import struct
import zlib

import numpy as np

from . import dds, structs as S
from .world import World

_COMPONENT = {5120: np.int8, 5121: np.uint8, 5122: np.int16, 5123: np.uint16, 5125: np.uint32, 5126: np.float32}
_WIDTH = {"SCALAR": 1, "VEC2": 2, "VEC3": 3, "VEC4": 4, "MAT2": 4, "MAT3": 9, "MAT4": 16}
_GLB_MAGIC, _CHUNK_JSON, _CHUNK_BIN = 0x46546C67, 0x4E4F534A, 0x004E4942


class GltfError(ValueError):
    pass


# ---------------------------------------------------------------------------------------------
# container
# ---------------------------------------------------------------------------------------------

def _read_container(path):
    with open(path, "rb") as f:
        blob = f.read()
    if len(blob) >= 12 and struct.unpack_from("<I", blob, 0)[0] == _GLB_MAGIC:
        _, version, length = struct.unpack_from("<III", blob, 0)
        if version != 2:
            raise GltfError("only GLB version 2 is supported")
        off, doc, bin_chunk = 12, None, None
        while off + 8 <= min(length, len(blob)):
            size, kind = struct.unpack_from("<II", blob, off)
            data = blob[off + 8: off + 8 + size]
            if kind == _CHUNK_JSON:
                doc = json.loads(data.decode("utf-8"))
            elif kind == _CHUNK_BIN and bin_chunk is None:
                bin_chunk = data
            off += 8 + size + (-size % 4)
        if doc is None:
            raise GltfError("GLB without a JSON chunk")
        return doc, bin_chunk
    return json.loads(blob.decode("utf-8")), None


def _load_uri(uri, base_dir):
    if uri.startswith("data:"):
        header, _, payload = uri.partition(",")
        return base64.b64decode(payload) if header.endswith(";base64") else payload.encode("latin-1")
    with open(os.path.join(base_dir, uri.replace("%20", " ")), "rb") as f:
        return f.read()


# ---------------------------------------------------------------------------------------------
# accessors (cgltf_accessor_unpack_floats / _indices)
# ---------------------------------------------------------------------------------------------

class _Document:
    def __init__(self, path):
        self.doc, glb_bin = _read_container(path)
        self.base = os.path.dirname(os.path.abspath(path))
        self.buffers = []
        for i, b in enumerate(self.doc.get("buffers", [])):
            if "uri" in b:
                self.buffers.append(_load_uri(b["uri"], self.base))
            elif i == 0 and glb_bin is not None:
                self.buffers.append(glb_bin)
            else:
                raise GltfError("buffer %d has no data" % i)

    def _raw(self, accessor):
        a = self.doc["accessors"][accessor]
        if "sparse" in a:
            raise GltfError("sparse accessors are not supported")
        if "bufferView" not in a:
            raise GltfError("accessor %d has no bufferView" % accessor)
        view = self.doc["bufferViews"][a["bufferView"]]
        dtype = np.dtype(_COMPONENT[a["componentType"]])
        width = _WIDTH[a["type"]]
        count = a["count"]
        start = view.get("byteOffset", 0) + a.get("byteOffset", 0)
        elem = dtype.itemsize * width
        stride = view.get("byteStride", 0) or elem
        buf = self.buffers[view["buffer"]]
        if count and start + stride * (count - 1) + elem > len(buf):
            raise GltfError("accessor %d reads past its buffer" % accessor)
        if stride == elem:
            data = np.frombuffer(buf, dtype=dtype, count=count * width, offset=start).reshape(count, width)
        else:
            rows = np.lib.stride_tricks.as_strided(
                np.frombuffer(buf, dtype=np.uint8, offset=start), shape=(count, elem), strides=(stride, 1))
            data = np.ascontiguousarray(rows).view(dtype).reshape(count, width)
        return data, bool(a.get("normalized", False))

    def floats(self, accessor):
        data, normalized = self._raw(accessor)
        out = data.astype(np.float32)
        if normalized and data.dtype != np.float32:
            info = np.iinfo(data.dtype)
            out = out / np.float32(info.max)
            if info.min < 0:
                out = np.maximum(out, np.float32(-1.0))
        return out

    def indices(self, accessor):
        data, _ = self._raw(accessor)
        return data.reshape(-1).astype(np.uint32)


# ---------------------------------------------------------------------------------------------
# images: Pillow when it is there, otherwise 8-bit non-interlaced PNG by hand
# ---------------------------------------------------------------------------------------------

def _decode_png(blob):
    if blob[:8] != b"\x89PNG\r\n\x1a\n":
        raise GltfError("not a PNG (JPEG needs Pillow)")
    off, idat, header, palette, trns = 8, [], None, None, None
    while off + 8 <= len(blob):
        size, kind = struct.unpack_from(">I4s", blob, off)
        data = blob[off + 8: off + 8 + size]
        if kind == b"IHDR":
            header = struct.unpack(">IIBBBBB", data)
        elif kind == b"PLTE":
            palette = np.frombuffer(data, np.uint8).reshape(-1, 3)
        elif kind == b"tRNS":
            trns = np.frombuffer(data, np.uint8)
        elif kind == b"IDAT":
            idat.append(data)
        elif kind == b"IEND":
            break
        off += 12 + size
    w, h, depth, colour, _, _, interlace = header
    channels = {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}[colour]
    if depth != 8 or interlace:
        raise GltfError("PNG fallback decoder handles 8-bit non-interlaced images only")
    raw = np.frombuffer(zlib.decompress(b"".join(idat)), np.uint8)
    row_bytes = w * channels
    raw = raw.reshape(h, row_bytes + 1)
    out = np.zeros((h, row_bytes), np.uint8)
    prev = np.zeros(row_bytes, np.int32)
    for y in range(h):
        kind, line = int(raw[y, 0]), raw[y, 1:].astype(np.int32)
        if kind == 0:
            cur = line
        elif kind == 2:
            cur = (line + prev) & 255
        else:
            cur = np.zeros(row_bytes, np.int32)
            for x in range(row_bytes):
                a = cur[x - channels] if x >= channels else 0
                b = prev[x]
                c = prev[x - channels] if x >= channels else 0
                if kind == 1:
                    pred = a
                elif kind == 3:
                    pred = (a + b) >> 1
                else:
                    p = a + b - c
                    pa, pb, pc = abs(p - a), abs(p - b), abs(p - c)
                    pred = a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)
                cur[x] = (line[x] + pred) & 255
        out[y] = cur
        prev = cur
    px = out.reshape(h, w, channels)
    rgba = np.full((h, w, 4), 255, np.uint8)
    if colour == 0:
        rgba[..., :3] = px
    elif colour == 2:
        rgba[..., :3] = px
    elif colour == 3:
        rgba[..., :3] = palette[px[..., 0]]
        if trns is not None:
            alpha = np.full(256, 255, np.uint8)
            alpha[: trns.size] = trns
            rgba[..., 3] = alpha[px[..., 0]]
    elif colour == 4:
        rgba[..., :3] = px[..., :1]
        rgba[..., 3] = px[..., 1]
    else:
        rgba[...] = px
    return rgba


def decode_image(blob):
    """Encoded PNG/JPEG bytes -> RGBA8 rows (what stb_image hands prosper, Texture.cpp:385)."""
    try:
        import io

        from PIL import Image
        return np.ascontiguousarray(np.asarray(Image.open(io.BytesIO(blob)).convert("RGBA"), dtype=np.uint8))
    except ImportError:
        return _decode_png(blob)


# ---------------------------------------------------------------------------------------------
# node transforms (glm conventions, column vectors)
# ---------------------------------------------------------------------------------------------

def _quat_to_mat4(q):
    """glm::mat4_cast of (x, y, z, w) as glTF stores it."""
    x, y, z, w = (float(c) for c in q)
    m = np.eye(4)
    m[0, 0] = 1 - 2 * (y * y + z * z)
    m[1, 0] = 2 * (x * y + w * z)
    m[2, 0] = 2 * (x * z - w * y)
    m[0, 1] = 2 * (x * y - w * z)
    m[1, 1] = 1 - 2 * (x * x + z * z)
    m[2, 1] = 2 * (y * z + w * x)
    m[0, 2] = 2 * (x * z + w * y)
    m[1, 2] = 2 * (y * z - w * x)
    m[2, 2] = 1 - 2 * (x * x + y * y)
    return m


def _euler_angles(q):
    """glm::eulerAngles (pitch, yaw, roll) of (x, y, z, w)."""
    x, y, z, w = (float(c) for c in q)
    pitch = math.atan2(2.0 * (y * z + w * x), w * w - x * x - y * y + z * z)
    yaw = math.asin(min(1.0, max(-1.0, -2.0 * (x * z - w * y))))
    roll = math.atan2(2.0 * (x * y + w * z), w * w + x * x - y * y - z * z)
    return pitch, yaw, roll


def _decompose(m):
    """T * R * S factors of a column-major 4x4 (the job glm::decompose does, WorldData.cpp:1182-1189)."""
    m = np.asarray(m, np.float64).reshape(4, 4).T  # glTF stores columns
    t = m[:3, 3].copy()
    cols = m[:3, :3].copy()
    s = np.linalg.norm(cols, axis=0)
    if np.linalg.det(cols) < 0:
        s = -s
    r = cols / np.where(s == 0, 1.0, s)
    # rotation matrix -> quaternion (x, y, z, w)
    tr = r[0, 0] + r[1, 1] + r[2, 2]
    if tr > 0:
        k = math.sqrt(tr + 1.0) * 2
        q = ((r[2, 1] - r[1, 2]) / k, (r[0, 2] - r[2, 0]) / k, (r[1, 0] - r[0, 1]) / k, 0.25 * k)
    elif r[0, 0] > r[1, 1] and r[0, 0] > r[2, 2]:
        k = math.sqrt(1.0 + r[0, 0] - r[1, 1] - r[2, 2]) * 2
        q = (0.25 * k, (r[0, 1] + r[1, 0]) / k, (r[0, 2] + r[2, 0]) / k, (r[2, 1] - r[1, 2]) / k)
    elif r[1, 1] > r[2, 2]:
        k = math.sqrt(1.0 + r[1, 1] - r[0, 0] - r[2, 2]) * 2
        q = ((r[0, 1] + r[1, 0]) / k, 0.25 * k, (r[1, 2] + r[2, 1]) / k, (r[0, 2] - r[2, 0]) / k)
    else:
        k = math.sqrt(1.0 + r[2, 2] - r[0, 0] - r[1, 1]) * 2
        q = ((r[0, 2] + r[2, 0]) / k, (r[1, 2] + r[2, 1]) / k, 0.25 * k, (r[1, 0] - r[0, 1]) / k)
    return t, q, s


_SRT_THRESHOLD = 0.001  # WorldData.cpp:1197


def _node_local_matrix(node):
    """The node's T * R * S with components within 1e-3 of identity dropped, as the reference keeps them."""
    translation, rotation, scale = np.zeros(3), (0.0, 0.0, 0.0, 1.0), np.ones(3)
    if "matrix" in node:
        translation, rotation, scale = _decompose(node["matrix"])
    if "translation" in node:
        translation = np.asarray(node["translation"], np.float64)
    if "rotation" in node:
        rotation = tuple(node["rotation"])
    if "scale" in node:
        scale = np.asarray(node["scale"], np.float64)
    m = np.eye(4)
    if np.any(np.abs(np.asarray(translation, np.float32)) > _SRT_THRESHOLD):
        t = np.eye(4)
        t[:3, 3] = np.asarray(translation, np.float32)
        m = m @ t
    if any(abs(a) > _SRT_THRESHOLD for a in _euler_angles(np.asarray(rotation, np.float32))):
        m = m @ _quat_to_mat4(np.asarray(rotation, np.float32))
    sc = np.asarray(scale, np.float32)
    if np.any(sc < 1.0 - _SRT_THRESHOLD) or np.any(sc > 1.0 + _SRT_THRESHOLD):
        m = m @ np.diag([float(sc[0]), float(sc[1]), float(sc[2]), 1.0])
    return m


# ---------------------------------------------------------------------------------------------
# the loader
# ---------------------------------------------------------------------------------------------

_FILTERS = {9728: S.FILTER_NEAREST, 9984: S.FILTER_NEAREST, 9986: S.FILTER_NEAREST,
            9729: S.FILTER_LINEAR, 9985: S.FILTER_LINEAR, 9987: S.FILTER_LINEAR}
_WRAPS = {33071: S.WRAP_CLAMP_TO_EDGE, 33648: S.WRAP_MIRRORED_REPEAT, 10497: S.WRAP_REPEAT}


def load_gltf(path, load_images=True, scene=None, use_texture_cache=True, bc7_on_gpu=False):
    """Reads `path` (.gltf or .glb) into a World laid out the way prosper lays the same file out.

    With `use_texture_cache`, an image file whose `prosper_cache/<name>.dds` exists (prosper's BC7 / RGBA8 cache,
    src/scene/Texture.cpp:38-47,213-296) AND is valid - its `.prosper_cache_tag` carries the current cache version and
    the source file's write time (Texture.cpp:124-160; `use_texture_cache="always"` skips that check, for caches that
    were copied between machines) - is read from there: that is what prosper samples, the PNG next to it is only the
    encoder's input.  A stale or untagged cache is ignored (prosper would re-encode it from the source).
    With `bc7_on_gpu` a BC7 cache file is handed to the library undecoded (`World.add_texture_bc7`: decoded by the
    HIP kernel at upload); otherwise `prosper_amd.bc7` decodes it here - same texels either way (tested).

    `world.missing_images` lists the image URIs that could not be read; they become 1x1 white textures
    (what prosper's texture slot 0, `empty.png`, is)."""
    d = _Document(path)
    doc = d.doc
    w = World()
    w.missing_images = []
    w.cached_images = []
    w.stale_cached_images = []

    # samplers: index + 1 (WorldData.cpp:699-719)
    for smp in doc.get("samplers", []):
        w.add_sampler(_FILTERS.get(smp.get("magFilter"), S.FILTER_LINEAR), _FILTERS.get(smp.get("minFilter"), S.FILTER_LINEAR),
                      _WRAPS.get(smp.get("wrapS", 10497), S.WRAP_CLAMP_TO_EDGE),
                      _WRAPS.get(smp.get("wrapT", 10497), S.WRAP_CLAMP_TO_EDGE))
    # images: texture index = image index + 1
    for img in doc.get("images", []):
        rgba = None
        if load_images:
            try:
                cached = None
                if use_texture_cache and "uri" in img and not img["uri"].startswith("data:"):
                    source = os.path.join(d.base, img["uri"].replace("%20", " "))
                    cached = dds.cache_path(source)
                    # Texture2D::init (Texture.cpp:377-415) reads the cache only when its tag names this source file's
                    # write time and the current cache version; otherwise it re-encodes from the source
                    if use_texture_cache != "always" and not dds.cache_valid(cached, source):
                        if os.path.exists(cached):
                            w.stale_cached_images.append(cached)
                        cached = None
                if cached is not None and os.path.exists(cached):
                    w.cached_images.append(cached)
                    fmt, tw, th, payload = dds.read_texture_raw(cached)
                    if bc7_on_gpu and fmt == dds.DXGI_FORMAT_BC7_UNORM:
                        w.add_texture_bc7(payload, tw, th)
                        continue
                    rgba = dds.read_texture(cached, levels=1)[0]
                elif "uri" in img:
                    rgba = decode_image(_load_uri(img["uri"], d.base))
                elif "bufferView" in img:
                    view = doc["bufferViews"][img["bufferView"]]
                    start = view.get("byteOffset", 0)
                    rgba = decode_image(d.buffers[view["buffer"]][start: start + view["byteLength"]])
            except (OSError, GltfError, dds.DdsError):
                rgba = None
        if rgba is None:
            w.missing_images.append(img.get("uri", "<bufferView>"))
            rgba = np.full((1, 1, 4), 255, np.uint8)
        w.add_texture(rgba)
    # glTF textures: (image + 1, sampler + 1) (WorldData.cpp:740-754)
    texture_pairs = [(0, 0)]
    for t in doc.get("textures", []):
        if "source" not in t:
            raise GltfError("texture without an image source")
        texture_pairs.append((t["source"] + 1, t["sampler"] + 1 if "sampler" in t else 0))

    def pair(info):
        return texture_pairs[info["index"] + 1] if info is not None else (0, 0)

    # materials: index + 1 (WorldData.cpp:756-828)
    for m in doc.get("materials", []):
        if "pbrMetallicRoughness" not in m:
            # the reference skips such a material WITHOUT reserving its slot, which shifts every later
            # index; files that rely on that are rejected instead of silently mis-assigned
            raise GltfError("material '%s' has no pbrMetallicRoughness block" % m.get("name", "?"))
        pbr = m["pbrMetallicRoughness"]
        mode = {"OPAQUE": S.ALPHA_MODE_OPAQUE, "MASK": S.ALPHA_MODE_MASK, "BLEND": S.ALPHA_MODE_BLEND}.get(
            m.get("alphaMode", "OPAQUE"), S.ALPHA_MODE_OPAQUE)
        w.add_material(base_color=pbr.get("baseColorFactor", (1.0, 1.0, 1.0, 1.0)),
                       metallic=float(pbr.get("metallicFactor", 1.0)), roughness=float(pbr.get("roughnessFactor", 1.0)),
                       alpha_cutoff=float(m.get("alphaCutoff", 0.5)), alpha_mode=mode,
                       base_tex=pair(pbr.get("baseColorTexture")), mr_tex=pair(pbr.get("metallicRoughnessTexture")),
                       normal_tex=pair(m.get("normalTexture")))

    # meshes: one Model per glTF mesh, one mesh per primitive (WorldData.cpp:830-915)
    for mesh in doc.get("meshes", []):
        subs = []
        for prim in mesh["primitives"]:
            if prim.get("mode", 4) != 4:
                raise GltfError("only triangle lists are supported")
            if "indices" not in prim:
                raise GltfError("non-indexed primitives are not supported (the reference asserts on them)")
            attrs = prim["attributes"]
            if "POSITION" not in attrs or "NORMAL" not in attrs:
                raise GltfError("POSITION and NORMAL are required (the reference asserts on them)")
            material = prim["material"] + 1 if "material" in prim else 0
            mi = w.add_mesh(
                d.floats(attrs["POSITION"])[:, :3], d.indices(prim["indices"]), material,
                normals=d.floats(attrs["NORMAL"])[:, :3],
                tangents=d.floats(attrs["TANGENT"])[:, :4] if "TANGENT" in attrs else None,
                uvs=d.floats(attrs["TEXCOORD_0"])[:, :2] if "TEXCOORD_0" in attrs else None)
            subs.append((mi, material))
        w.add_model(subs)

    # scene: depth-first with a LIFO stack (WorldData.cpp:1364-1456), transforms as World.cpp:359-466
    scenes = doc.get("scenes", [])
    nodes = doc.get("nodes", [])
    lights = doc.get("extensions", {}).get("KHR_lights_punctual", {}).get("lights", [])
    if scenes:
        scene_index = doc.get("scene", 0) if scene is None else scene
        roots = scenes[scene_index].get("nodes", [])
    else:
        roots = []
    camera_set = False
    sun_found = False
    for root in roots:
        stack = [(root, np.eye(4))]
        while stack:
            index, parent = stack.pop()
            node = nodes[index]
            m4 = (parent.astype(np.float32) @ _node_local_matrix(node).astype(np.float32)).astype(np.float64)
            for child in node.get("children", []):
                stack.append((child, m4))
            if "mesh" in node:
                w.add_instance(node["mesh"], m4)
            if node.get("camera") == 0 and not camera_set:  # m_currentCamera starts at camera 0
                cam = doc["cameras"][node["camera"]]
                if cam.get("type") == "perspective":
                    p = cam["perspective"]
                    w.camera = dict(
                        eye=tuple(float(v) for v in (m4 @ np.array([0.0, 0.0, 0.0, 1.0]))[:3]),
                        target=tuple(float(v) for v in (m4 @ np.array([0.0, 0.0, -1.0, 1.0]))[:3]),
                        up=tuple(float(v) for v in m4[:3, :3] @ np.array([0.0, 1.0, 0.0])),
                        fov=float(p["yfov"]), zN=float(p["znear"]), zF=float(p.get("zfar", 100.0)))
                    camera_set = True
            light_index = node.get("extensions", {}).get("KHR_lights_punctual", {}).get("light")
            if light_index is not None:
                light = lights[light_index]
                color = light.get("color", (1.0, 1.0, 1.0))
                intensity = float(light.get("intensity", 1.0))
                position = (m4 @ np.array([0.0, 0.0, 0.0, 1.0]))[:3]
                direction = m4[:3, :3] @ np.array([0.0, 0.0, -1.0])
                if light["type"] == "directional":
                    # a second sun is logged as ignored but still overwrites the first (WorldData.cpp:1462-1480,
                    # World.cpp:428-433): the last one in traversal order wins
                    w.set_directional_light(color, intensity, direction)
                    sun_found = True
                elif light["type"] == "point":
                    w.add_point_light(color, intensity, position, float(light.get("range", 0.0)))
                elif light["type"] == "spot":
                    spot = light.get("spot", {})
                    w.add_spot_light(color, intensity, position, direction, float(spot.get("innerConeAngle", 0.0)),
                                     float(spot.get("outerConeAngle", math.pi / 4.0)))
    return w
