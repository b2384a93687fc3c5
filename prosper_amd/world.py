"""Host-side scene container: the data contract prosper's `scene::World` hands to the RT pass.

Mirrors (in NumPy, because it is asset preparation, not the hot path):
  * packMeshData                  src/scene/DeferredLoadingContext.cpp:442-490
  * geometry blob order / offsets src/scene/DeferredLoadingContext.cpp:609-802,1192-1269
  * DrawInstance / TLAS ordering  src/scene/World.cpp:468-536,878-928
  * node transforms               src/scene/World.cpp:396-414
  * glTF light conversion         src/scene/WorldData.cpp:1455-1543, World.cpp:428-456
  * material / texture / sampler index conventions  src/scene/WorldData.cpp:681-828
"""
import ctypes as C
import math

import numpy as np

from . import structs as S


def pack_half(x):
    """glm::packHalf*: IEEE binary16, round-to-nearest-even."""
    return np.asarray(x, dtype=np.float32).astype(np.float16).view(np.uint16)


def _round_half_away(x):
    # glm::round = trunc(x + copysign(0.5, x))
    return np.trunc(x + np.copysign(np.float32(0.5), x))


def pack_snorm3x10_1x2(v):
    """glm::packSnorm3x10_1x2: round(clamp(v,-1,1) * (511,511,511,1)) into 10/10/10/2 bits."""
    v = np.asarray(v, dtype=np.float32).reshape(-1, 4)
    q = _round_half_away(np.clip(v, -1.0, 1.0) * np.array([511, 511, 511, 1], dtype=np.float32)).astype(np.int32)
    u = q.astype(np.uint32)
    return (u[:, 0] & 0x3FF) | ((u[:, 1] & 0x3FF) << 10) | ((u[:, 2] & 0x3FF) << 20) | ((u[:, 3] & 0x3) << 30)


def pack_mesh_data(positions, normals=None, tangents=None, uvs=None):
    """packMeshData (DeferredLoadingContext.cpp:442-490): fp32 attributes -> packed streams."""
    positions = np.asarray(positions, dtype=np.float32).reshape(-1, 3)
    n = positions.shape[0]
    p4 = np.concatenate([positions, np.ones((n, 1), np.float32)], axis=1)
    out = {"positions": pack_half(p4).reshape(n, 4).copy().view(np.uint32).reshape(n, 2)}
    if normals is not None:
        nn = np.concatenate([np.asarray(normals, np.float32).reshape(n, 3), np.zeros((n, 1), np.float32)], axis=1)
        out["normals"] = pack_snorm3x10_1x2(nn)
    if tangents is not None:
        out["tangents"] = pack_snorm3x10_1x2(np.asarray(tangents, np.float32).reshape(n, 4))
    if uvs is not None:
        out["uvs"] = pack_half(np.asarray(uvs, np.float32).reshape(n, 2)).copy().view(np.uint32).reshape(n)
    return out


def translate(t):
    m = np.eye(4, dtype=np.float64)
    m[:3, 3] = t
    return m


def scale(s):
    s = np.broadcast_to(np.asarray(s, dtype=np.float64), (3,))
    return np.diag([s[0], s[1], s[2], 1.0])


def rotate_y(angle):
    c, s = math.cos(angle), math.sin(angle)
    return np.array([[c, 0, s, 0], [0, 1, 0, 0], [-s, 0, c, 0], [0, 0, 0, 1]], dtype=np.float64)


def rotate_x(angle):
    c, s = math.cos(angle), math.sin(angle)
    return np.array([[1, 0, 0, 0], [0, c, -s, 0], [0, s, c, 0], [0, 0, 0, 1]], dtype=np.float64)


def rotate_z(angle):
    c, s = math.cos(angle), math.sin(angle)
    return np.array([[c, -s, 0, 0], [s, c, 0, 0], [0, 0, 1, 0], [0, 0, 0, 1]], dtype=np.float64)


class Bc7Texture:
    """BC7 blocks of one texture level, [N, 16] uint8, row-major."""

    def __init__(self, blocks, width, height):
        self.blocks, self.width, self.height = blocks, width, height


class World:
    """Builder + container for everything `prosper_pt_scene_view` points at."""

    GEOMETRY_BUFFER_BYTES = 64 * 1024 * 1024  # sGeometryBufferSize, DeferredLoadingContext.cpp:22

    def __init__(self):
        self._buffers = [[]]  # list of lists of uint32 arrays
        self._buffer_words = [0]
        self.metadatas = []
        self.mesh_infos = []
        self.models = []           # list of [(meshIndex, materialIndex)]
        self.model_instances = []  # list of (modelIndex, M4)
        # material 0 is the default material, texture 0 / sampler 0 the defaults
        self.materials = [self._material_struct()]
        self.textures = [np.full((1, 1, 4), 255, np.uint8)]
        self.samplers = [(S.FILTER_LINEAR, S.FILTER_LINEAR, S.WRAP_REPEAT, S.WRAP_REPEAT)]
        self.directional = S.DirectionalLightParameters()
        self.directional.irradiance = S.Vec4(2.0, 2.0, 2.0, 2.0)
        self.directional.direction = S.Vec4(-1.0, -1.0, -1.0, 1.0)
        self._directional_found = False
        self.point_lights = S.PointLightsBuffer()
        self.spot_lights = S.SpotLightsBuffer()
        self.skybox = None  # np.float16 [6, N, N, 4]
        self.camera = dict(eye=(1.0, 0.5, 1.0), target=(0.0, 0.0, 0.0), up=(0.0, 1.0, 0.0),
                           fov=math.radians(59.0), zN=0.1, zF=100.0)
        self._frozen = None

    # ---- textures / samplers / materials (WorldData.cpp:681-828) ----
    def add_texture(self, rgba8):
        rgba8 = np.ascontiguousarray(rgba8, dtype=np.uint8)
        assert rgba8.ndim == 3 and rgba8.shape[2] == 4
        self.textures.append(rgba8)
        return len(self.textures) - 1

    def add_texture_bc7(self, blocks, width, height):
        """Level 0 of a BC7 texture as prosper's cache stores it (`dds.read_texture_level0_raw`): the library
        decodes it on the GPU at upload (PROSPER_PT_FORMAT_BC7_UNORM)."""
        blocks = np.ascontiguousarray(np.frombuffer(bytes(blocks), np.uint8) if not isinstance(blocks, np.ndarray) else blocks,
                                      dtype=np.uint8).reshape(-1, 16)
        assert width % 4 == 0 and height % 4 == 0 and blocks.shape[0] == (width // 4) * (height // 4)
        self.textures.append(Bc7Texture(blocks, width, height))
        return len(self.textures) - 1

    def add_sampler(self, mag=S.FILTER_LINEAR, min_=S.FILTER_LINEAR, wrap_s=S.WRAP_REPEAT, wrap_t=S.WRAP_REPEAT):
        self.samplers.append((mag, min_, wrap_s, wrap_t))
        return len(self.samplers) - 1

    @staticmethod
    def _material_struct(base_color=(1, 1, 1, 1), metallic=1.0, roughness=1.0, alpha_cutoff=0.5,
                         alpha_mode=S.ALPHA_MODE_OPAQUE, base_tex=(0, 0), mr_tex=(0, 0), normal_tex=(0, 0)):
        m = S.MaterialData()
        m.baseColorFactor = S.Vec4(*[float(c) for c in base_color])
        m.metallicFactor = metallic
        m.roughnessFactor = roughness
        m.alphaCutoff = alpha_cutoff
        m.alphaMode = alpha_mode
        m.baseColorTextureSampler = (base_tex[1] << 24) | base_tex[0]
        m.metallicRoughnessTextureSampler = (mr_tex[1] << 24) | mr_tex[0]
        m.normalTextureSampler = (normal_tex[1] << 24) | normal_tex[0]
        return m

    def add_material(self, **kw):
        self.materials.append(self._material_struct(**kw))
        return len(self.materials) - 1

    # ---- meshes (DeferredLoadingContext.cpp:609-802, 1192-1269) ----
    def add_mesh(self, positions, indices, material_index, normals=None, tangents=None, uvs=None,
                 force_u32_indices=False, buffer_index=None):
        positions = np.asarray(positions, np.float32).reshape(-1, 3)
        indices = np.asarray(indices, np.uint32).reshape(-1)
        vertex_count = positions.shape[0]
        assert indices.size % 3 == 0 and (indices.size == 0 or int(indices.max()) < vertex_count)
        packed = pack_mesh_data(positions, normals, tangents, uvs)
        # usesShortIndices = vertexCount <= 0xFFFF (DeferredLoadingContext.cpp:627)
        short = vertex_count <= 0xFFFF and not force_u32_indices
        if short:
            idx16 = indices.astype(np.uint16)
            if idx16.size % 2:
                idx16 = np.concatenate([idx16, np.zeros(1, np.uint16)])
            idx_words = idx16.view(np.uint32)
        else:
            idx_words = indices

        words_needed = idx_words.size + packed["positions"].size + sum(
            packed[k].size for k in ("normals", "tangents", "uvs") if k in packed)
        if buffer_index is None:
            buffer_index = len(self._buffers) - 1
            if (self._buffer_words[buffer_index] + words_needed) * 4 > self.GEOMETRY_BUFFER_BYTES:
                self._buffers.append([])
                self._buffer_words.append(0)
                buffer_index += 1
        while buffer_index >= len(self._buffers):
            self._buffers.append([])
            self._buffer_words.append(0)

        md = S.GeometryMetadata(*([S.ABSENT] * 10), 1 if short else 0)
        md.bufferIndex = buffer_index

        def push(words):
            off = self._buffer_words[buffer_index]
            self._buffers[buffer_index].append(np.ascontiguousarray(words, np.uint32).reshape(-1))
            self._buffer_words[buffer_index] += int(words.size)
            return off

        off = push(idx_words)
        md.indicesOffset = off * 2 if short else off  # u16 units when short
        md.positionsOffset = push(packed["positions"])
        if "normals" in packed:
            md.normalsOffset = push(packed["normals"])
        if "tangents" in packed:
            md.tangentsOffset = push(packed["tangents"])
        if "uvs" in packed:
            md.texCoord0sOffset = push(packed["uvs"])
        self.metadatas.append(md)
        self.mesh_infos.append(S.MeshInfo(vertex_count, int(indices.size), 0, material_index))
        return len(self.metadatas) - 1

    @property
    def mesh_ranges(self):
        """Per mesh: (buffer index, first word, words) of the bytes the path reads of it in its geometry buffer - indices,
        positions and the attribute streams (UploadedGeometryData's range, DeferredLoadingContext.cpp:1192-1269)."""
        out = []
        for md, info in zip(self.metadatas, self.mesh_infos):
            short = md.usesShortIndices == 1
            lo = md.indicesOffset // 2 if short else md.indicesOffset
            hi = (md.indicesOffset + info.indexCount + 1) // 2 if short else md.indicesOffset + info.indexCount
            hi = max(hi, md.positionsOffset + 2 * info.vertexCount)
            for off in (md.normalsOffset, md.tangentsOffset, md.texCoord0sOffset):
                if off != S.ABSENT:
                    hi = max(hi, off + info.vertexCount)
                    lo = min(lo, off)
            lo = min(lo, md.positionsOffset)
            out.append((md.bufferIndex, lo, hi - lo))
        return out

    def with_meshes_loaded(self, loaded, buffers=None):
        """The scene while it streams in (WorldData::pollMeshWorker, WorldData.cpp:2003-2110): a copy in which only the
        meshes in `loaded` have arrived - the others keep a default-constructed GeometryMetadata (bufferIndex 0xFFFFFFFF),
        an empty MeshInfo and zeros where their bytes will be; buffers, draw instances and everything else are the
        finished scene's.  `buffers`: only the first so many geometry buffers exist yet (the loader creates the next one
        when a mesh no longer fits, DeferredLoadingContext.cpp:1192-1269)."""
        import copy
        loaded = set(loaded)
        keep = buffers
        w = copy.copy(self)
        w._frozen = None
        w.metadatas = [m if i in loaded else S.GeometryMetadata(*([S.ABSENT] * 10), 0) for i, m in enumerate(self.metadatas)]
        w.mesh_infos = [m if i in loaded else S.MeshInfo(0, 0, 0, 0) for i, m in enumerate(self.mesh_infos)]
        f = self.freeze()
        buffers = [b.copy() for b in f["geometry_buffers"]]
        for i, (b, first, words) in enumerate(self.mesh_ranges):
            if i not in loaded:
                buffers[b][first:first + words] = 0
        if keep is not None:
            assert all(self.mesh_ranges[i][0] < keep for i in loaded)
            buffers = buffers[:keep]
        w._buffers = [[b] for b in buffers]
        w._buffer_words = [int(b.size) for b in buffers]
        return w

    def add_model(self, sub_models):
        """sub_models: list of (meshIndex, materialIndex) — scene::Model::SubModel."""
        self.models.append(list(sub_models))
        return len(self.models) - 1

    def add_instance(self, model_index, transform=None):
        m = np.eye(4) if transform is None else np.asarray(transform, np.float64)
        self.model_instances.append((model_index, m))
        return len(self.model_instances) - 1

    # ---- lights (WorldData.cpp:1455-1543; World.cpp:428-456) ----
    def set_directional_light(self, color, intensity, direction):
        self.directional.irradiance = S.Vec4(*(float(c) * intensity for c in color), 0.0)
        self.directional.direction = S.Vec4(*[float(d) for d in direction], 0.0)
        self._directional_found = True

    def add_point_light(self, color, intensity_w, position, light_range=0.0):
        radiance = np.asarray(color, np.float32) * np.float32(intensity_w) / np.float32(4.0 * math.pi)
        luminance = float(np.dot(radiance, np.array([0.2126, 0.7152, 0.0722], np.float32)))
        radius = light_range if light_range > 0.0 else math.sqrt(luminance / 0.01)
        i = self.point_lights.count
        assert i < S.MAX_POINT_LIGHT_COUNT
        self.point_lights.lights[i].radianceAndRadius = S.Vec4(*[float(r) for r in radiance], radius)
        self.point_lights.lights[i].position = S.Vec4(*[float(p) for p in position], 1.0)
        self.point_lights.count = i + 1

    def add_spot_light(self, color, intensity_w, position, direction, inner_cone, outer_cone):
        angle_scale = 1.0 / max(0.001, math.cos(inner_cone) - math.cos(outer_cone))
        angle_offset = -math.cos(outer_cone) * angle_scale
        rad = np.asarray(color, np.float32) * np.float32(intensity_w) / np.float32(4.0 * math.pi)
        i = self.spot_lights.count
        assert i < S.MAX_SPOT_LIGHT_COUNT
        L = self.spot_lights.lights[i]
        L.radianceAndAngleScale = S.Vec4(*[float(r) for r in rad], angle_scale)
        L.positionAndAngleOffset = S.Vec4(*[float(p) for p in position], angle_offset)
        L.direction = S.Vec4(*[float(d) for d in direction], 0.0)
        self.spot_lights.count = i + 1

    # ---- freeze into C arrays ----
    def freeze(self):
        """Flatten models x instances into DrawInstance[] / ModelInstanceTransforms[] and pin all arrays."""
        if self._frozen is not None:
            return self._frozen
        # "Honor scene lighting": no sun if the scene has punctual lights only (WorldData.cpp:1537-1542)
        if not self._directional_found and (self.point_lights.count or self.spot_lights.count):
            self.directional.irradiance = S.Vec4(0.0, 0.0, 0.0, 0.0)

        f = {}
        f["geometry_buffers"] = [
            np.concatenate(parts) if parts else np.zeros(1, np.uint32) for parts in self._buffers
        ]
        f["metadatas"] = (S.GeometryMetadata * max(1, len(self.metadatas)))(*self.metadatas)
        f["mesh_infos"] = (S.MeshInfo * max(1, len(self.mesh_infos)))(*self.mesh_infos)

        draw_instances = []
        transforms = (S.ModelInstanceTransforms * max(1, len(self.model_instances)))()
        for mi, (model_index, m4) in enumerate(self.model_instances):
            m32 = m4.astype(np.float32)
            inv = np.linalg.inv(m4).astype(np.float32)
            t = transforms[mi]
            for r in range(3):
                # modelToWorld = transpose(M4): column r of the mat3x4 is row r of the affine
                t.modelToWorld.col[r] = S.Vec4(*[float(x) for x in m32[r, :]])
                # normalToWorld = mat3x4(inverse(M4)): column r of inverse(M4)
                t.normalToWorld.col[r] = S.Vec4(*[float(x) for x in inv[:, r]])
            for mesh_index, material_index in self.models[model_index]:
                draw_instances.append(S.DrawInstance(mi, mesh_index, material_index))
        f["transforms"] = transforms
        f["draw_instances"] = (S.DrawInstance * max(1, len(draw_instances)))(*draw_instances)
        f["draw_instance_count"] = len(draw_instances)
        f["materials"] = (S.MaterialData * len(self.materials))(*self.materials)
        tex = (S.TextureDesc * len(self.textures))()
        for i, t in enumerate(self.textures):
            if isinstance(t, Bc7Texture):
                tex[i].texels = t.blocks.ctypes.data
                tex[i].height, tex[i].width = t.height, t.width
                tex[i].format = S.FORMAT_BC7_UNORM
                continue
            tex[i].texels = t.ctypes.data
            tex[i].height, tex[i].width = t.shape[0], t.shape[1]
            tex[i].format = S.FORMAT_RGBA8_UNORM
        f["textures"] = tex
        smp = (S.SamplerDesc * len(self.samplers))()
        for i, s in enumerate(self.samplers):
            smp[i].magFilter, smp[i].minFilter, smp[i].wrapS, smp[i].wrapT = s
        f["samplers"] = smp
        self._frozen = f
        return f

    def view(self):
        """Returns a prosper_pt_scene_view; the World must outlive its use."""
        f = self.freeze()
        v = S.SceneView()
        v.struct_size = C.sizeof(S.SceneView)
        nb = len(f["geometry_buffers"])
        ptrs = (C.c_void_p * nb)(*[b.ctypes.data for b in f["geometry_buffers"]])
        sizes = (C.c_uint64 * nb)(*[b.nbytes for b in f["geometry_buffers"]])
        f["_ptrs"], f["_sizes"] = ptrs, sizes
        v.geometryBuffers = ptrs
        v.geometryBufferByteSizes = sizes
        v.geometryBufferCount = nb
        v.meshCount = len(self.metadatas)
        v.geometryMetadatas = f["metadatas"]
        v.meshInfos = f["mesh_infos"]
        v.drawInstances = f["draw_instances"]
        v.drawInstanceCount = f["draw_instance_count"]
        v.modelInstanceCount = len(self.model_instances)
        v.modelInstanceTransforms = f["transforms"]
        v.materials = f["materials"]
        v.materialCount = len(self.materials)
        v.textureCount = len(self.textures)
        v.textures = f["textures"]
        v.samplers = f["samplers"]
        v.samplerCount = len(self.samplers)
        v.directionalLight = C.pointer(self.directional)
        v.pointLights = C.pointer(self.point_lights)
        v.spotLights = C.pointer(self.spot_lights)
        if self.skybox is not None:
            sky = np.ascontiguousarray(self.skybox, dtype=np.float16)
            assert sky.ndim == 4 and sky.shape[0] == 6 and sky.shape[1] == sky.shape[2] and sky.shape[3] == 4
            f["_sky"] = sky
            v.skybox.texels = sky.ctypes.data
            v.skybox.faceSize = sky.shape[1]
        return v

    def triangle_count(self):
        self.freeze()
        total = 0
        for model_index, _ in self.model_instances:
            for mesh_index, _ in self.models[model_index]:
                total += self.mesh_infos[mesh_index].indexCount // 3
        return total
