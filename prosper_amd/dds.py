"""DDS reader for the assets on either side of the path, DX10-header files only, like the
reference's own reader (src/utils/Dds.cpp: magic, 124-byte header, 'DX10' four-cc, DXGI format):
  * the tone-map LUT, a 3-D R9G9B9E5_SHAREDEXP texture (`res/texture/tony_mc_mapface.dds`, 48^3);
  * prosper's texture cache, 2-D R8G8B8A8_UNORM / BC7_UNORM with mips (`prosper_cache/<name>.dds`)."""
import struct

import numpy as np

DXGI_FORMAT_R8G8B8A8_UNORM = 28
DXGI_FORMAT_R9G9B9E5_SHAREDEXP = 67
DXGI_FORMAT_BC7_UNORM = 98
_DIMENSION_TEXTURE2D = 3
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


def _level_sizes(width, height, fmt, mips):
    """(width, height, byte size) of every mip level, as the reference lays them out (src/utils/Dds.cpp:100-135:
    levels back to back, `max(extent >> i, 1)`, BC7 levels in whole 16-byte 4x4 blocks)."""
    out = []
    for i in range(mips):
        w, h = max(width >> i, 1), max(height >> i, 1)
        if fmt == DXGI_FORMAT_R8G8B8A8_UNORM:
            size = w * h * 4
        elif fmt == DXGI_FORMAT_BC7_UNORM:
            if w % 4 or h % 4:
                raise DdsError("BC7 mip %d is %dx%d: levels must divide evenly by 4x4" % (i, w, h))
            size = (w // 4) * (h // 4) * 16
        else:
            raise DdsError("unsupported DXGI format %d" % fmt)
        out.append((w, h, size))
    return out


def _texture_header(blob):
    if len(blob) < 148 or blob[:4] != b"DDS ":
        raise DdsError("not a DDS file")
    header = struct.unpack_from("<31I", blob, 4)
    size, _, height, width, _, depth, mips = header[:7]
    if size != 124 or header[20] != 0x30315844:
        raise DdsError("only DX10-header DDS files are supported (as in the reference)")
    fmt, dimension, _, array_size, _ = struct.unpack_from("<5I", blob, 128)
    if dimension != _DIMENSION_TEXTURE2D or array_size != 1 or depth > 1:
        raise DdsError("expected a single 2-D texture")
    if fmt not in (DXGI_FORMAT_R8G8B8A8_UNORM, DXGI_FORMAT_BC7_UNORM):
        raise DdsError("only R8G8B8A8_UNORM and BC7_UNORM 2-D textures are supported (src/utils/Dds.cpp:299-302)")
    return fmt, width, height, _level_sizes(width, height, fmt, max(mips, 1))


def read_texture_raw(path):
    """-> (DXGI format, width, height, level-0 payload bytes) of a texture-cache DDS, undecoded: BC7 blocks can go
    to the library as they are (PROSPER_PT_FORMAT_BC7_UNORM, decoded on the GPU at upload)."""
    with open(path, "rb") as f:
        blob = f.read()
    fmt, width, height, sizes = _texture_header(blob)
    if len(blob) < 148 + sizes[0][2]:
        raise DdsError("truncated DDS payload")
    return fmt, width, height, blob[148: 148 + sizes[0][2]]


def read_texture(path, levels=1):
    """A 2-D texture of prosper's texture cache (`prosper_cache/<name>.dds`, src/scene/Texture.cpp:38-47, 213-296):
    DX10-header DDS, R8G8B8A8_UNORM or BC7_UNORM, mip chain down to 4x4.  -> list of uint8 [h, w, 4] RGBA levels
    (the first `levels` of them; None = all).  BC7 levels are decoded by prosper_amd.bc7 - the texels the GPU's
    sampler sees in prosper.  The path tracer samples LOD 0 only (no derivatives in ray-tracing stages)."""
    from . import bc7
    with open(path, "rb") as f:
        blob = f.read()
    fmt, width, height, sizes = _texture_header(blob)
    out, offset = [], 148
    for i, (w, h, nbytes) in enumerate(sizes):
        if levels is not None and i >= levels:
            break
        if len(blob) < offset + nbytes:
            raise DdsError("truncated DDS payload")
        data = blob[offset: offset + nbytes]
        if fmt == DXGI_FORMAT_BC7_UNORM:
            out.append(bc7.decode_image(data, w, h))
        else:
            out.append(np.frombuffer(data, np.uint8).reshape(h, w, 4).copy())
        offset += nbytes
    return out


def write_texture(path, fmt, width, height, level_payloads):
    """Fixture writer: a DX10-header 2-D DDS from ready-made level payloads (bytes: RGBA8 texels or BC7 blocks),
    with the header fields the reference writes (src/utils/Dds.cpp:160-230)."""
    sizes = _level_sizes(width, height, fmt, len(level_payloads))
    for (w, h, nbytes), payload in zip(sizes, level_payloads):
        if len(payload) != nbytes:
            raise DdsError("level payload of %d bytes, expected %d" % (len(payload), nbytes))
    compressed = fmt == DXGI_FORMAT_BC7_UNORM
    mips = len(level_payloads)
    header = [0] * 31
    # caps | height | width | pixelformat (+ linearsize / pitch) (+ mipmapcount)
    header[0], header[1] = 124, 0x1007 | (0x80000 if compressed else 0x8) | (0x20000 if mips > 1 else 0)
    header[2], header[3] = height, width
    header[4] = sizes[0][2] if compressed else width * 4
    header[5], header[6] = 1, mips
    header[18], header[19], header[20] = 32, 0x4, 0x30315844
    header[26] = 0x1000 | (0x400008 if mips > 1 else 0)
    with open(path, "wb") as f:
        f.write(b"DDS " + struct.pack("<31I", *header) + struct.pack("<5I", fmt, _DIMENSION_TEXTURE2D, 0, 1, 0))
        for payload in level_payloads:
            f.write(payload)


def cache_path(source):
    """Where prosper caches the compressed copy of texture file `source` (src/scene/Texture.cpp:38-47)."""
    import os
    folder = os.path.join(os.path.dirname(source), "prosper_cache")
    return os.path.join(folder, os.path.splitext(os.path.basename(source))[0] + ".dds")


# ---- the texture cache's validity tag (src/scene/Texture.cpp:27-29,49-160) ----
TEXTURE_CACHE_MAGIC = 0x5845545250535250  # "PRSPRTEX"
TEXTURE_CACHE_VERSION = 5
# std::filesystem::file_time_type as libstdc++ stores it (the tag is not meant to be portable, Texture.cpp:72-74):
# int64 nanoseconds on chrono::file_clock, whose epoch is 2174-01-01T00:00:00Z = 6 437 664 000 s after the Unix epoch
_FILE_CLOCK_EPOCH_DIFF_NS = 6437664000 * 10**9


def cache_tag_path(cache_file):
    """cacheTagPath: the cache file with the extension replaced (Texture.cpp:49-54)."""
    import os
    return os.path.splitext(cache_file)[0] + ".prosper_cache_tag"


def source_write_time(source):
    """std::filesystem::last_write_time(source) in the tag's representation."""
    import os
    return os.stat(source).st_mtime_ns - _FILE_CLOCK_EPOCH_DIFF_NS


def read_cache_tag(cache_file):
    """readCacheTag (Texture.cpp:63-95) -> (version, sourceWriteTime); a missing tag or another version gives
    (0xFFFFFFFF or that version, None); a wrong magic raises DdsError like the original's runtime_error."""
    import os
    path = cache_tag_path(cache_file)
    if not os.path.exists(path):
        return 0xFFFFFFFF, None
    with open(path, "rb") as f:
        blob = f.read()
    if len(blob) < 4:
        return 0xFFFFFFFF, None
    version = struct.unpack_from("<I", blob, 0)[0]
    if version != TEXTURE_CACHE_VERSION:
        return version, None
    if len(blob) < 20 or struct.unpack_from("<Q", blob, 4)[0] != TEXTURE_CACHE_MAGIC:
        raise DdsError("expected a valid texture cache tag in file '%s'" % path)
    return version, struct.unpack_from("<q", blob, 12)[0]


def write_cache_tag(cache_file, source):
    """writeCacheTag (Texture.cpp:97-122): version, magic, the source's write time; through a temporary file."""
    import os
    path = cache_tag_path(cache_file)
    tmp = os.path.splitext(path)[0] + ".prosper_cache_tag_TMP"
    with open(tmp, "wb") as f:
        f.write(struct.pack("<IQq", TEXTURE_CACHE_VERSION, TEXTURE_CACHE_MAGIC, source_write_time(source)))
    os.replace(tmp, path)


def cache_valid(cache_file, source):
    """cacheValid (Texture.cpp:124-160): the cache file exists, its tag has the current version and the write time of
    the source file it was made from; any error reading the tag means invalid."""
    import os
    try:
        if not os.path.exists(cache_file):
            return False
        version, write_time = read_cache_tag(cache_file)
        return version == TEXTURE_CACHE_VERSION and write_time == source_write_time(source)
    except (OSError, DdsError, struct.error):
        return False

