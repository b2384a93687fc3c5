"""BC7 (BPTC UNORM) block decoder, numpy, bit-exact per the format definition (Khronos Data Format Specification,
"BPTC compressed texture image formats"; the same rules D3D11's BC7_UNORM and VK_FORMAT_BC7_UNORM_BLOCK decode by).

Why it is here: prosper does not sample the PNG texels of a glTF - it compresses every texture whose
mip chain divides by 4 to BC7 with a third-party encoder and caches the result as `prosper_cache/<name>.dds`
(src/scene/Texture.cpp:213-296, 377-415); the GPU then decodes those blocks in hardware.  Texel parity with prosper
on real assets therefore means reading that cache and decoding it exactly; the encoder itself (lossy heuristics of
the ISPC texture compressor) is not part of the path.

Pinned against an independent implementation: tests/test_bc7.py decodes random blocks of all eight modes (and the
reserved one) with Pillow's "bcn" decoder and compares every byte; the partition / anchor tables below were
recovered from that decoder by tests/golden/make_bc7_tables.py (the reference holds neither tables nor test data).
"""
import numpy as np

# subset of each of the 16 pixels (row-major) for the 64 two-subset and 64 three-subset partitions
PARTITION2 = (
    '0011001100110011', '0001000100010001', '0111011101110111', '0001001100110111',
    '0000000100010011', '0011011101111111', '0001001101111111', '0000000100110111',
    '0000000000010011', '0011011111111111', '0000000101111111', '0000000000010111',
    '0001011111111111', '0000000011111111', '0000111111111111', '0000000000001111',
    '0000100011101111', '0111000100000000', '0000000010001110', '0111001100010000',
    '0011000100000000', '0000100011001110', '0000000010001100', '0111001100110001',
    '0011000100010000', '0000100010001100', '0110011001100110', '0011011001101100',
    '0001011111101000', '0000111111110000', '0111000110001110', '0011100110011100',
    '0101010101010101', '0000111100001111', '0101101001011010', '0011001111001100',
    '0011110000111100', '0101010110101010', '0110100101101001', '0101101010100101',
    '0111001111001110', '0001001111001000', '0011001001001100', '0011101111011100',
    '0110100110010110', '0011110011000011', '0110011010011001', '0000011001100000',
    '0100111001000000', '0010011100100000', '0000001001110010', '0000010011100100',
    '0110110010010011', '0011011011001001', '0110001110011100', '0011100111000110',
    '0110110011001001', '0110001100111001', '0111111010000001', '0001100011100111',
    '0000111100110011', '0011001111110000', '0010001011101110', '0100010001110111',
)
PARTITION3 = (
    '0011001102212222', '0001001122112221', '0000200122112211', '0222002200110111',
    '0000000011221122', '0011001100220022', '0022002211111111', '0011001122112211',
    '0000000011112222', '0000111111112222', '0000111122222222', '0012001200120012',
    '0112011201120112', '0122012201220122', '0011011211221222', '0011200122002220',
    '0001001101121122', '0111001120012200', '0000112211221122', '0022002200221111',
    '0111011102220222', '0001000122212221', '0000001101220122', '0000110022102210',
    '0122012200110000', '0012001211222222', '0110122112210110', '0000011012211221',
    '0022110211020022', '0110011020022222', '0011012201220011', '0000200022112221',
    '0000000211221222', '0222002200120011', '0011001200220222', '0120012001200120',
    '0000111122220000', '0120120120120120', '0120201212010120', '0011220011220011',
    '0011112222000011', '0101010122222222', '0000000021212121', '0022112200221122',
    '0022001100220011', '0220122102201221', '0101222222220101', '0000212121212121',
    '0101010101012222', '0222011102220111', '0002111200021112', '0000211221122112',
    '0222011101110222', '0002111211120002', '0110011001102222', '0000000021122112',
    '0110011022222222', '0022001100110022', '0022112211220022', '0000000000002112',
    '0002000100020001', '0222122202221222', '0101222222222222', '0111201122012220',
)
# anchor pixel (its index drops the top bit) of subset 1 of a two-subset partition, of subsets 1 and 2 of a
# three-subset one; the anchor of subset 0 is always pixel 0
ANCHOR2 = (
    15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15,
    15, 2, 8, 2, 2, 8, 8, 15, 2, 8, 2, 2, 8, 8, 2, 2,
    15, 15, 6, 8, 2, 8, 15, 15, 2, 8, 2, 2, 2, 15, 15, 6,
    6, 2, 6, 8, 15, 15, 2, 2, 15, 15, 15, 15, 15, 2, 2, 15,
)
ANCHOR3A = (
    3, 3, 15, 15, 8, 3, 15, 15, 8, 8, 6, 6, 6, 5, 3, 3,
    3, 3, 8, 15, 3, 3, 6, 10, 5, 8, 8, 6, 8, 5, 15, 15,
    8, 15, 3, 5, 6, 10, 8, 15, 15, 3, 15, 5, 15, 15, 15, 15,
    3, 15, 5, 5, 5, 8, 5, 10, 5, 10, 8, 13, 15, 12, 3, 3,
)
ANCHOR3B = (
    15, 8, 8, 3, 15, 15, 3, 8, 15, 15, 15, 15, 15, 15, 15, 8,
    15, 8, 15, 3, 15, 8, 15, 8, 3, 15, 6, 10, 15, 15, 10, 8,
    15, 3, 15, 10, 10, 8, 9, 10, 6, 15, 8, 15, 3, 6, 6, 8,
    15, 3, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 3, 15, 15, 8,
)

# mode: subsets, partition bits, rotation bits, index-selection bits, colour bits, alpha bits,
#       per-endpoint p-bits, shared (per-subset) p-bits, index bits, secondary index bits
MODES = (
    (3, 4, 0, 0, 4, 0, 1, 0, 3, 0),
    (2, 6, 0, 0, 6, 0, 0, 1, 3, 0),
    (3, 6, 0, 0, 5, 0, 0, 0, 2, 0),
    (2, 6, 0, 0, 7, 0, 1, 0, 2, 0),
    (1, 0, 2, 1, 5, 6, 0, 0, 2, 3),
    (1, 0, 2, 0, 7, 8, 0, 0, 2, 2),
    (1, 0, 0, 0, 7, 7, 1, 0, 4, 0),
    (2, 6, 0, 0, 5, 5, 1, 0, 2, 0),
)
WEIGHTS = {2: (0, 21, 43, 64), 3: (0, 9, 18, 27, 37, 46, 55, 64),
           4: (0, 4, 9, 13, 17, 21, 26, 30, 34, 38, 43, 47, 51, 55, 60, 64)}

_P2 = np.array([[int(c) for c in row] for row in PARTITION2], np.int64)
_P3 = np.array([[int(c) for c in row] for row in PARTITION3], np.int64)
_A2 = np.array(ANCHOR2, np.int64)
_A3A = np.array(ANCHOR3A, np.int64)
_A3B = np.array(ANCHOR3B, np.int64)


def _field(bits, pos, n):
    """bits [N, 128] (bit k of the block in column k) -> the n-bit little-endian field at fixed position pos."""
    w = (1 << np.arange(n, dtype=np.int64))
    return bits[:, pos:pos + n].astype(np.int64) @ w


def _indices(bits, start, width, anchors):
    """16 indices of `width` bits starting at bit `start`, pixel order; pixels listed in anchors [N, k] store
    width - 1 bits.  -> [N, 16]"""
    n = bits.shape[0]
    pixel = np.arange(16, dtype=np.int64)[None, :]
    is_anchor = (pixel[:, :, None] == anchors[:, None, :]).any(axis=2)            # [N, 16]
    before = (anchors[:, None, :] < pixel[:, :, None]).sum(axis=2)                 # anchors strictly before the pixel
    pos = start + pixel * width - before                                           # [N, 16]
    out = np.zeros((n, 16), np.int64)
    for k in range(width):
        have = ~is_anchor | (k < width - 1)
        at = np.minimum(pos + k, 127)
        out |= (np.take_along_axis(bits, at, axis=1).astype(np.int64) & have) << k
    return out


def _interpolate(e0, e1, idx, width):
    w = np.array(WEIGHTS[width], np.int64)[idx]
    return ((64 - w) * e0 + w * e1 + 32) >> 6


def _decode_mode(bits, mode):
    ns, pb, rb, isb, cb, ab, epb, spb, ib, ib2 = MODES[mode]
    n = bits.shape[0]
    pos = mode + 1
    partition = _field(bits, pos, pb) if pb else np.zeros(n, np.int64)
    pos += pb
    rotation = _field(bits, pos, rb) if rb else np.zeros(n, np.int64)
    pos += rb
    index_selection = _field(bits, pos, isb) if isb else np.zeros(n, np.int64)
    pos += isb
    # endpoints: channel-major, then endpoint (subset 0 e0, subset 0 e1, subset 1 e0, ...)
    ends = np.zeros((n, 2 * ns, 4), np.int64)
    for channel in range(3):
        for e in range(2 * ns):
            ends[:, e, channel] = _field(bits, pos, cb)
            pos += cb
    if ab:
        for e in range(2 * ns):
            ends[:, e, 3] = _field(bits, pos, ab)
            pos += ab
    channels = 4 if ab else 3
    cbits = np.array([cb, cb, cb, ab], np.int64)
    if epb:
        for e in range(2 * ns):
            p = _field(bits, pos, 1)
            pos += 1
            ends[:, e, :channels] = (ends[:, e, :channels] << 1) | p[:, None]
        cbits = cbits + 1
    if spb:
        for s in range(ns):
            p = _field(bits, pos, 1)
            pos += 1
            for e in (2 * s, 2 * s + 1):
                ends[:, e, :channels] = (ends[:, e, :channels] << 1) | p[:, None]
        cbits = cbits + 1
    for channel in range(channels):
        b = int(cbits[channel])
        v = ends[:, :, channel] << (8 - b)
        ends[:, :, channel] = v | (v >> b)
    if not ab:
        ends[:, :, 3] = 255

    if ns == 1:
        subset = np.zeros((n, 16), np.int64)
        anchors = np.zeros((n, 1), np.int64)
    elif ns == 2:
        subset = _P2[partition]
        anchors = np.stack([np.zeros(n, np.int64), _A2[partition]], axis=1)
    else:
        subset = _P3[partition]
        anchors = np.stack([np.zeros(n, np.int64), _A3A[partition], _A3B[partition]], axis=1)
    idx = _indices(bits, pos, ib, anchors)
    pos += 16 * ib - ns
    e0 = np.take_along_axis(ends, (2 * subset)[:, :, None].repeat(4, axis=2), axis=1)       # [N, 16, 4]
    e1 = np.take_along_axis(ends, (2 * subset + 1)[:, :, None].repeat(4, axis=2), axis=1)
    out = np.empty((n, 16, 4), np.int64)
    if ib2:
        # two index sets (modes 4, 5): colour from the primary and alpha from the secondary set, swapped when the
        # index-selection bit of mode 4 is set
        idx2 = _indices(bits, pos, ib2, anchors)
        swap = (index_selection == 1)[:, None]
        colour_a = _interpolate(e0[:, :, :3], e1[:, :, :3], idx[:, :, None], ib)
        colour_b = _interpolate(e0[:, :, :3], e1[:, :, :3], idx2[:, :, None], ib2)
        alpha_a = _interpolate(e0[:, :, 3], e1[:, :, 3], idx, ib)
        alpha_b = _interpolate(e0[:, :, 3], e1[:, :, 3], idx2, ib2)
        out[:, :, :3] = np.where(swap[:, :, None], colour_b, colour_a)
        out[:, :, 3] = np.where(swap, alpha_a, alpha_b)
    else:
        out[:] = _interpolate(e0, e1, idx[:, :, None], ib)
    if rb:
        for r, channel in ((1, 0), (2, 1), (3, 2)):
            m = rotation == r
            if m.any():
                a = out[m, :, 3].copy()
                out[m, :, 3] = out[m, :, channel]
                out[m, :, channel] = a
    return out.astype(np.uint8)


def decode_blocks(blocks):
    """uint8 [N, 16] BC7 blocks -> uint8 [N, 16, 4] RGBA texels (pixel order row-major inside the 4x4 block).
    A block whose mode byte is 0 (reserved) decodes to transparent black, as the format specifies."""
    blocks = np.ascontiguousarray(blocks, np.uint8).reshape(-1, 16)
    n = blocks.shape[0]
    out = np.zeros((n, 16, 4), np.uint8)
    first = blocks[:, 0].astype(np.int64)
    low = first & -first                                  # lowest set bit of the mode byte
    mode = np.where(first == 0, 8, np.log2(np.maximum(low, 1)).astype(np.int64))
    for m in range(8):
        sel = np.nonzero(mode == m)[0]
        for c0 in range(0, len(sel), 1 << 16):            # bounded working set
            chunk = sel[c0:c0 + (1 << 16)]
            bits = np.unpackbits(blocks[chunk], axis=1, bitorder="little")
            out[chunk] = _decode_mode(bits, m)
    return out


def decode_image(data, width, height):
    """BC7 payload of one mip level (blocks row-major, 16 B each) -> uint8 [height, width, 4]."""
    if width % 4 or height % 4:
        raise ValueError("BC7 levels are whole 4x4 blocks (the reference falls back to RGBA8 otherwise)")
    bx, by = width // 4, height // 4
    blocks = np.frombuffer(data, np.uint8, count=bx * by * 16).reshape(-1, 16)
    texels = decode_blocks(blocks).reshape(by, bx, 4, 4, 4)
    return np.ascontiguousarray(texels.transpose(0, 2, 1, 3, 4).reshape(height, width, 4))
