"""Image partition for multi-GPU rendering.

The image is cut into vertical stripes `stripe_width` pixels wide; rank r renders stripes s with
s % world == r (prosper_pt_tile_desc).  Each rank's HDR tile is [height, local_width, 4] with its
stripes in ascending order; one gather to rank 0 plus `deinterleave` rebuilds the full image.
RNG seeds use absolute pixel coordinates (rt/reference/main.rgen:227-229), so the result is
bit-identical to a single-GPU render.
"""
from . import structs as S

STRIPE_WIDTH = 16


def tile_for_rank(rank, world, stripe_width=STRIPE_WIDTH):
    return S.TileDesc(stripe_width, rank, world) if world > 1 else None


def local_width(width, rank, world, stripe_width=STRIPE_WIDTH):
    if world <= 1:
        return width
    n = 0
    for x in range(0, width, stripe_width):
        if (x // stripe_width) % world == rank:
            n += min(stripe_width, width - x)
    return n


def check_divisible(width, world, stripe_width=STRIPE_WIDTH):
    """A single-count gather needs equal tiles: the stripe count must divide over the ranks."""
    return width % stripe_width == 0 and (width // stripe_width) % world == 0


def deinterleave(tiles, width, stripe_width=STRIPE_WIDTH):
    """tiles: list (one per rank, rank order) of [height, local_width, C] torch tensors or numpy
    arrays of equal shape -> [height, width, C]."""
    world = len(tiles)
    if world == 1:
        return tiles[0]
    height, lw, ch = tiles[0].shape
    k = lw // stripe_width
    assert k * stripe_width == lw and k * world * stripe_width == width
    if hasattr(tiles[0], "view") and not hasattr(tiles[0], "ctypes"):  # torch
        import torch
        parts = [t.view(height, k, stripe_width, ch) for t in tiles]
        return torch.stack(parts, dim=2).reshape(height, width, ch)
    import numpy as np
    parts = [t.reshape(height, k, stripe_width, ch) for t in tiles]
    return np.stack(parts, axis=2).reshape(height, width, ch)
