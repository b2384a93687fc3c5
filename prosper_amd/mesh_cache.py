"""`.prosper_mesh` v4 cache blobs: the per-mesh files prosper's loader writes after packing a glTF
primitive (src/scene/DeferredLoadingContext.cpp:609-802, header struct DeferredLoadingContext.hpp:59-78) and
reads back instead of re-packing (:505-607).  Layout: u64 magic 'PRSPRMSH', u32 version (4), the header
(source write time as the platform's 8-byte file_time_type, then 13 u32: indexCount, vertexCount, meshletCount,
eight offsets, usesShortIndices, blobByteCount) and the blob: indices (u16 padded to 4 B, or u32) . positions
(half4) . normals (snorm 10:10:10:2) . tangents . texCoord0 (half2) . meshlets . meshletBounds . meshletVertices .
meshletTriangles.  Attribute offsets count u32 words from the start of the blob; 0xFFFFFFFF = absent.

The path tracer needs indices, positions, normals, tangents and uvs; meshlet sections are carried along
untouched (the RT pass never reads them)."""
import struct

import numpy as np

from . import structs as S

MAGIC = 0x48534D5250535250  # "PRSPRMSH"
VERSION = 4
ABSENT = 0xFFFFFFFF
_HEADER = struct.Struct("<QIq13I")
_FIELDS = ("indexCount", "vertexCount", "meshletCount", "positionsOffset", "normalsOffset", "tangentsOffset",
           "texCoord0sOffset", "meshletsOffset", "meshletBoundsOffset", "meshletVerticesOffset",
           "meshletTrianglesByteOffset", "usesShortIndices", "blobByteCount")


class MeshCacheError(ValueError):
    pass


def read_mesh_cache(path):
    """-> (header dict, blob as uint32 words)"""
    with open(path, "rb") as f:
        data = f.read()
    if len(data) < _HEADER.size:
        raise MeshCacheError("truncated mesh cache")
    magic, version, write_time, *fields = _HEADER.unpack_from(data, 0)
    if magic != MAGIC:
        raise MeshCacheError("expected a valid mesh cache (bad magic)")
    if version != VERSION:
        raise MeshCacheError("mesh cache version %d, expected %d" % (version, VERSION))
    header = dict(zip(_FIELDS, fields), sourceWriteTime=write_time)
    n = header["blobByteCount"]
    if n % 4 or len(data) < _HEADER.size + n:
        raise MeshCacheError("blob size does not match the header")
    if header["usesShortIndices"] != (1 if header["vertexCount"] <= 0xFFFF else 0):
        raise MeshCacheError("usesShortIndices disagrees with the vertex count")
    return header, np.frombuffer(data, dtype="<u4", count=n // 4, offset=_HEADER.size).copy()


def write_mesh_cache(path, header, blob_words, source_write_time=0):
    blob = np.ascontiguousarray(blob_words, dtype="<u4")
    h = dict(header, blobByteCount=blob.size * 4)
    with open(path, "wb") as f:
        f.write(_HEADER.pack(MAGIC, VERSION, source_write_time, *[h[k] for k in _FIELDS]))
        f.write(blob.tobytes())


def pack_cache(positions, indices, normals, tangents=None, uvs=None):
    """What writeCache lays out for a primitive without meshlets: (header, blob words)."""
    from .world import pack_mesh_data
    positions = np.asarray(positions, np.float32).reshape(-1, 3)
    indices = np.asarray(indices, np.uint32).reshape(-1)
    packed = pack_mesh_data(positions, normals, tangents, uvs)
    short = positions.shape[0] <= 0xFFFF
    if short:
        idx16 = indices.astype(np.uint16)
        if idx16.size % 2:
            idx16 = np.concatenate([idx16, np.zeros(1, np.uint16)])
        parts = [idx16.view(np.uint32)]
    else:
        parts = [indices]
    header = dict(indexCount=int(indices.size), vertexCount=int(positions.shape[0]), meshletCount=0,
                  usesShortIndices=1 if short else 0)
    words = parts[0].size
    for key, name in (("positionsOffset", "positions"), ("normalsOffset", "normals"), ("tangentsOffset", "tangents"),
                      ("texCoord0sOffset", "uvs")):
        if name in packed:
            header[key] = words
            parts.append(np.ascontiguousarray(packed[name], np.uint32).reshape(-1))
            words += parts[-1].size
        else:
            header[key] = ABSENT
    # empty meshlet sections: every offset points at the end of the vertex data, like computeOffset does
    header["meshletsOffset"] = header["meshletBoundsOffset"] = words
    header["meshletVerticesOffset"] = words * (2 if short else 1)
    header["meshletTrianglesByteOffset"] = words * 4
    return header, np.concatenate(parts).astype(np.uint32)


def add_cached_mesh(world, header, blob_words, material_index):
    """Appends a cached blob to the world's geometry buffer the way uploadGeometryData does
    (DeferredLoadingContext.cpp:1192-1269): the blob goes in as is, the metadata offsets are the header's
    plus where the blob landed."""
    blob = np.ascontiguousarray(blob_words, dtype=np.uint32)
    b = len(world._buffers) - 1
    if (world._buffer_words[b] + blob.size) * 4 > world.GEOMETRY_BUFFER_BYTES:
        world._buffers.append([])
        world._buffer_words.append(0)
        b += 1
    base = world._buffer_words[b]
    world._buffers[b].append(blob)
    world._buffer_words[b] += int(blob.size)
    short = header["usesShortIndices"] == 1
    md = S.GeometryMetadata(*([S.ABSENT] * 10), 1 if short else 0)
    md.bufferIndex = b
    md.indicesOffset = base * 2 if short else base

    def off(key):
        return S.ABSENT if header[key] == ABSENT else base + header[key]
    md.positionsOffset = off("positionsOffset")
    md.normalsOffset = off("normalsOffset")
    md.tangentsOffset = off("tangentsOffset")
    md.texCoord0sOffset = off("texCoord0sOffset")
    world.metadatas.append(md)
    world.mesh_infos.append(S.MeshInfo(header["vertexCount"], header["indexCount"], header["meshletCount"], material_index))
    world._frozen = None
    return len(world.metadatas) - 1
