"""DDS reader for the one asset the tone-map step needs: a DX10-header DDS holding a 3-D
R9G9B9E5_SHAREDEXP texture, i.e. prosper's `res/texture/tony_mc_mapface.dds` (48^3), as the reference's own
reader accepts it (src/utils/Dds.cpp: magic, 124-byte header, 'DX10' four-cc, DXGI format, TEXTURE3D)."""
import struct

import numpy as np

DXGI_FORMAT_R8G8B8A8_UNORM = 28
DXGI_FORMAT_R9G9B9E5_SHAREDEXP = 67
_DIMENSION_TEXTURE3D = 4


class DdsError(ValueError):
    pass


def read_lut(path):
    """-> uint32 [depth, height, width] R9G9B9E5 texels of a 3-D DDS."""
    with open(path, "rb") as f:
        blob = f.read()
    if len(blob) < 148 or blob[:4] != b"DDS ":
        raise DdsError("not a DDS file")
    header = struct.unpack_from("<31I", blob, 4)
    size, _, height, width, _, depth, mips = header[:7]
    four_cc = header[20]
    if size != 124 or four_cc != 0x30315844:  # 'DX10'
        raise DdsError("only DX10-header DDS files are supported (as in the reference)")
    fmt, dimension, _, array_size, _ = struct.unpack_from("<5I", blob, 128)
    if fmt != DXGI_FORMAT_R9G9B9E5_SHAREDEXP or dimension != _DIMENSION_TEXTURE3D or array_size != 1 or mips > 1:
        raise DdsError("expected a single-mip 3-D R9G9B9E5_SHAREDEXP texture")
    count = width * height * depth
    if len(blob) < 148 + 4 * count:
        raise DdsError("truncated DDS payload")
    return np.frombuffer(blob, dtype="<u4", count=count, offset=148).reshape(depth, height, width).copy()


def decode_r9g9b9e5(texels):
    """uint32 R9G9B9E5 -> float32 [..., 3] (exact: mantissa * 2^(e - 24))."""
    t = np.asarray(texels, np.uint32)
    scale = np.ldexp(np.float32(1.0), (t >> 27).astype(np.int32) - 24).astype(np.float32)
    return np.stack([(t & 0x1FF).astype(np.float32) * scale, ((t >> 9) & 0x1FF).astype(np.float32) * scale,
                     ((t >> 18) & 0x1FF).astype(np.float32) * scale], axis=-1)


def encode_r9g9b9e5(rgb):
    """float [..., 3] >= 0 -> uint32 R9G9B9E5 (shared exponent from the largest channel; for test LUTs)."""
    rgb = np.clip(np.asarray(rgb, np.float64), 0.0, 65408.0)
    mx = np.maximum(rgb.max(axis=-1), 1e-30)
    e = np.clip(np.floor(np.log2(mx)).astype(np.int64) + 1 + 15, 0, 31)
    scale = np.ldexp(1.0, 24 - e)[..., None]
    m = np.rint(rgb * scale).astype(np.int64)
    over = (m.max(axis=-1) > 511) & (e < 31)
    e = np.where(over, e + 1, e)
    m = np.clip(np.rint(rgb * np.ldexp(1.0, 24 - e)[..., None]).astype(np.int64), 0, 511)
    return (m[..., 0] | (m[..., 1] << 9) | (m[..., 2] << 18) | (e << 27)).astype(np.uint32)


def write_lut(path, texels):
    """uint32 [d, h, w] R9G9B9E5 texels -> a DX10-header DDS like tony_mc_mapface.dds (fixture writer)."""
    t = np.ascontiguousarray(texels, dtype="<u4")
    d, h, w = t.shape
    header = [0] * 31
    header[0], header[1], header[2], header[3], header[4], header[5], header[6] = 124, 0x80100F, h, w, w * 4, d, 1
    header[18], header[19], header[20] = 32, 0x4, 0x30315844  # pixel format: size, DDPF_FOURCC, 'DX10'
    header[26], header[27] = 0x1008, 0x200000                # caps: texture | complex, caps2: volume
    with open(path, "wb") as f:
        f.write(b"DDS " + struct.pack("<31I", *header) +
                struct.pack("<5I", DXGI_FORMAT_R9G9B9E5_SHAREDEXP, _DIMENSION_TEXTURE3D, 0, 1, 0) + t.tobytes())
