"""KTX 1.1 reader for the one kind of file the path consumes: the RGBA16F environment cube
prosper samples on a miss (`env/*.ktx`, src/scene/Texture.cpp:589-636), parsed the way the reference's
reader does (src/utils/Ktx.cpp:50-166): 12-byte identifier, 13-word header, little-endian only, GL_HALF_FLOAT /
GL_RGBA / GL_RGBA16F only, key/value data skipped, per mip a u32 imageSize that for a non-array cube is the
size of ONE face, faces tightly packed in +X,-X,+Y,-Y,+Z,-Z order.  Only mip 0 is used (RT stages sample LOD 0)."""
import struct

import numpy as np

_IDENTIFIER_10 = bytes([0xAB, 0x4B, 0x54, 0x58, 0x20, 0x31, 0x31, 0xBB, 0x0D, 0x0A, 0x1A, 0x0A])
_IDENTIFIER_20 = bytes([0xAB, 0x4B, 0x54, 0x58, 0x20, 0x32, 0x30, 0xBB, 0x0D, 0x0A, 0x1A, 0x0A])
_GL_HALF_FLOAT, _GL_RGBA, _GL_RGBA16F = 0x140B, 0x1908, 0x881A


class KtxError(ValueError):
    pass


def read_cube(path):
    """-> float16 [6, N, N, 4]: mip 0 of an RGBA16F cube map (what World.skybox expects)."""
    with open(path, "rb") as f:
        blob = f.read()
    if blob[:12] == _IDENTIFIER_20:
        raise KtxError("KTX 2.0 is not supported")
    if blob[:12] != _IDENTIFIER_10 or len(blob) < 64:
        raise KtxError("not a KTX 1.1 file")
    (endianness, gl_type, _type_size, gl_format, gl_internal, gl_base, width, height, _depth, _layers, faces, _mips,
     kv_bytes) = struct.unpack_from("<13I", blob, 12)
    if endianness != 0x04030201:
        raise KtxError("KTX and program endianness don't match")
    if gl_type != _GL_HALF_FLOAT or gl_format != _GL_RGBA or gl_internal != _GL_RGBA16F or gl_base != gl_format:
        raise KtxError("only RGBA16F is supported")
    if faces != 6 or width == 0 or height != width:
        raise KtxError("expected a square cube map with 6 faces")
    off = 64 + kv_bytes
    (face_bytes,) = struct.unpack_from("<I", blob, off)
    if face_bytes != width * width * 8:
        raise KtxError("unexpected imageSize for mip 0 (faces must be tightly packed)")
    off += 4
    if len(blob) < off + 6 * face_bytes:
        raise KtxError("truncated KTX payload")
    return np.frombuffer(blob, dtype="<f2", count=6 * width * width * 4, offset=off).reshape(6, width, width, 4).copy()


def write_cube(path, cube, mip_levels=1):
    """Writes float16 [6, N, N, 4] as a KTX 1.1 RGBA16F cube (box-filtered mips when mip_levels > 1): the
    fixture writer for the reader above."""
    cube = np.ascontiguousarray(cube, dtype=np.float16)
    assert cube.ndim == 4 and cube.shape[0] == 6 and cube.shape[1] == cube.shape[2] and cube.shape[3] == 4
    n = cube.shape[1]
    key_value = b"KTXorientation\x00S=r,T=d\x00"
    entry = struct.pack("<I", len(key_value)) + key_value + b"\x00" * (-len(key_value) % 4)
    header = _IDENTIFIER_10 + struct.pack("<13I", 0x04030201, _GL_HALF_FLOAT, 2, _GL_RGBA, _GL_RGBA16F, _GL_RGBA, n, n, 0, 0,
                                          6, mip_levels, len(entry))
    out = [header, entry]
    level = cube.astype(np.float32)
    for _ in range(mip_levels):
        data = level.astype(np.float16)
        out.append(struct.pack("<I", data.shape[1] * data.shape[2] * 8))
        out.append(data.tobytes())
        if level.shape[1] > 1:
            level = 0.25 * (level[:, 0::2, 0::2] + level[:, 1::2, 0::2] + level[:, 0::2, 1::2] + level[:, 1::2, 1::2])
    with open(path, "wb") as f:
        f.write(b"".join(out))
