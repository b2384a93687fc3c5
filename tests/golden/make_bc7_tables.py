#!/usr/bin/env python3
"""Recovers the BC7 partition and anchor-index tables by probing an independent decoder (Pillow's "bcn" raw
decoder) with hand-built blocks, and prints them as Python literals (the tables in prosper_amd/bc7.py).

The tables are constants of the format (Khronos Data Format Specification, BPTC section: 64 two-subset and 64
three-subset partitions of the 4x4 block, and the anchor pixel of the second / third subset of each); no copy of
them exists in this image or in the reference, which only links a third-party BC7 *encoder*.  Probes:
  * subset of every pixel: both endpoints of subset k set to a colour unique to k, all indices 0;
  * anchors: endpoints 0 and max for every subset, every index bit 1 - an anchor pixel stores one bit less, so it
    decodes to a smaller weight than all the others.
tests/test_bc7.py re-runs the probe and compares with the committed tables.
"""
import sys

import numpy as np
from PIL import Image


class Bits:
    def __init__(self):
        self.value = 0
        self.pos = 0

    def put(self, v, n):
        assert 0 <= v < (1 << n)
        self.value |= v << self.pos
        self.pos += n

    def block(self):
        assert self.pos <= 128, self.pos
        return self.value.to_bytes(16, "little")


def decode(block):
    return np.asarray(Image.frombytes("RGBA", (4, 4), block, "bcn", (7,))).reshape(16, 4)


def mode1_block(partition, reds, index_bits_value):
    """mode 1: 2 subsets, 6-bit partition, 6-bit RGB, one shared p-bit per subset, 3-bit indices."""
    b = Bits()
    b.put(0b10, 2)
    b.put(partition, 6)
    for channel in range(3):
        for e in range(4):  # s0e0 s0e1 s1e0 s1e1
            b.put(reds[e] if channel == 0 else 0, 6)
    b.put(0, 2)
    b.put(index_bits_value, 46)
    return b.block()


def mode2_block(partition, reds, index_bits_value):
    """mode 2: 3 subsets, 6-bit partition, 5-bit RGB, no p-bits, 2-bit indices."""
    b = Bits()
    b.put(0b100, 3)
    b.put(partition, 6)
    for channel in range(3):
        for e in range(6):
            b.put(reds[e] if channel == 0 else 0, 5)
    b.put(index_bits_value, 29)
    return b.block()


def probe():
    p2, p3, a2, a3a, a3b = [], [], [], [], []
    for partition in range(64):
        red = decode(mode1_block(partition, [0, 0, 63, 63], 0))[:, 0]
        subset = (red > 127).astype(int)
        assert subset[0] == 0
        p2.append(subset.tolist())
        red = decode(mode1_block(partition, [0, 63, 0, 63], (1 << 46) - 1))[:, 0]
        anchors = np.nonzero(red < 200)[0]
        assert len(anchors) == 2 and anchors[0] == 0 and subset[anchors[1]] == 1, (partition, anchors)
        a2.append(int(anchors[1]))

        red = decode(mode2_block(partition, [0, 0, 15, 15, 31, 31], 0))[:, 0]
        subset = np.where(red < 60, 0, np.where(red < 200, 1, 2))
        assert subset[0] == 0
        p3.append(subset.tolist())
        red = decode(mode2_block(partition, [0, 31, 0, 31, 0, 31], (1 << 29) - 1))[:, 0]
        anchors = np.nonzero(red < 200)[0]
        assert len(anchors) == 3 and anchors[0] == 0, (partition, anchors)
        by_subset = {int(subset[a]): int(a) for a in anchors}
        assert sorted(by_subset) == [0, 1, 2]
        a3a.append(by_subset[1])
        a3b.append(by_subset[2])
    return p2, p3, a2, a3a, a3b


def main():
    p2, p3, a2, a3a, a3b = probe()

    def packed(rows):
        return ["".join(str(v) for v in r) for r in rows]
    out = sys.stdout
    out.write("PARTITION2 = %r\n" % (packed(p2),))
    out.write("PARTITION3 = %r\n" % (packed(p3),))
    out.write("ANCHOR2 = %r\n" % (a2,))
    out.write("ANCHOR3A = %r\n" % (a3a,))
    out.write("ANCHOR3B = %r\n" % (a3b,))


if __name__ == "__main__":
    main()
