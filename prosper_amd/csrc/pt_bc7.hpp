// pt_bc7.hpp - BC7 (BPTC UNORM) block decode on the device.
//
// prosper keeps its material textures BC7-compressed (src/scene/Texture.cpp:213-296: every texture whose mip
// chain divides by 4 is compressed into `prosper_cache/<name>.dds`) and lets the GPU's sampler decode them.  This
// build samples RGBA8 tiles (pt_scene.hpp DeviceTexture), so a BC7 texture handed over through the C-ABI
// (PROSPER_PT_FORMAT_BC7_UNORM) is decoded ONCE at upload, one thread per 4x4 block, into the same tiled layout:
// the texels the hardware decoder would return, bit for bit (format definition: Khronos Data Format Specification,
// BPTC).  prosper_amd/bc7.py is the same decoder in numpy (pinned against Pillow) and the parity check of this one.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "pt_bc7_tables.hpp"

namespace ppt
{

struct Bc7Bits
{
    uint64_t lo, hi;
    uint32_t pos;
    // next n (1..8) bits, least significant first
    __device__ __forceinline__ uint32_t get(uint32_t n)
    {
        uint64_t v;
        if (pos >= 64u)
            v = hi >> (pos - 64u);
        else
        {
            v = lo >> pos;
            if (pos + n > 64u) v |= hi << (64u - pos);
        }
        pos += n;
        return (uint32_t)v & ((1u << n) - 1u);
    }
};

__device__ __forceinline__ uint32_t bc7_weight(uint32_t bits, uint32_t index)
{
    // 2-bit: 0 21 43 64; 3-bit: 0 9 18 27 37 46 55 64; 4-bit: 0 4 9 13 17 21 26 30 34 38 43 47 51 55 60 64
    const uint64_t w2 = 0x402b1500ull;
    const uint64_t w3 = 0x40372e251b120900ull;
    const uint64_t w4lo = 0x1e1a15110d090400ull, w4hi = 0x403c37332f2b2622ull;
    if (bits == 2u) return (uint32_t)(w2 >> (8u * index)) & 0xFFu;
    if (bits == 3u) return (uint32_t)(w3 >> (8u * index)) & 0xFFu;
    return (uint32_t)((index < 8u ? w4lo >> (8u * index) : w4hi >> (8u * (index - 8u)))) & 0xFFu;
}

__device__ __forceinline__ uint32_t bc7_interpolate(uint32_t e0, uint32_t e1, uint32_t w)
{
    return ((64u - w) * e0 + w * e1 + 32u) >> 6;
}

// block: the 16 bytes as four little-endian words; texel[i] = R | G << 8 | B << 16 | A << 24 of pixel i
// (row-major in the 4x4 block).  A reserved block (mode byte 0) decodes to 0.
__device__ inline void bc7_decode_block(const uint32_t block[4], uint32_t texel[16])
{
    const uint32_t first = block[0] & 0xFFu;
    if (first == 0u)
    {
        for (int i = 0; i < 16; ++i) texel[i] = 0u;
        return;
    }
    const uint32_t mode = (uint32_t)__builtin_ctz(first);
    // subsets, partition bits, rotation bits, index-selection bits, colour bits, alpha bits, per-endpoint p-bits,
    // shared (per-subset) p-bits, index bits, secondary index bits
    uint32_t ns, pb, rb, isb, cb, ab, epb, spb, ib, ib2;
    switch (mode)
    {
    case 0: ns = 3; pb = 4; rb = 0; isb = 0; cb = 4; ab = 0; epb = 1; spb = 0; ib = 3; ib2 = 0; break;
    case 1: ns = 2; pb = 6; rb = 0; isb = 0; cb = 6; ab = 0; epb = 0; spb = 1; ib = 3; ib2 = 0; break;
    case 2: ns = 3; pb = 6; rb = 0; isb = 0; cb = 5; ab = 0; epb = 0; spb = 0; ib = 2; ib2 = 0; break;
    case 3: ns = 2; pb = 6; rb = 0; isb = 0; cb = 7; ab = 0; epb = 1; spb = 0; ib = 2; ib2 = 0; break;
    case 4: ns = 1; pb = 0; rb = 2; isb = 1; cb = 5; ab = 6; epb = 0; spb = 0; ib = 2; ib2 = 3; break;
    case 5: ns = 1; pb = 0; rb = 2; isb = 0; cb = 7; ab = 8; epb = 0; spb = 0; ib = 2; ib2 = 2; break;
    case 6: ns = 1; pb = 0; rb = 0; isb = 0; cb = 7; ab = 7; epb = 1; spb = 0; ib = 4; ib2 = 0; break;
    default: ns = 2; pb = 6; rb = 0; isb = 0; cb = 5; ab = 5; epb = 1; spb = 0; ib = 2; ib2 = 0; break;
    }
    Bc7Bits bits{(uint64_t)block[0] | ((uint64_t)block[1] << 32), (uint64_t)block[2] | ((uint64_t)block[3] << 32), mode + 1u};
    const uint32_t partition = pb ? bits.get(pb) : 0u;
    const uint32_t rotation = rb ? bits.get(rb) : 0u;
    const uint32_t indexSelection = isb ? bits.get(isb) : 0u;

    // endpoints, channel-major: [endpoint][channel]
    uint32_t ends[6][4];
    for (uint32_t c = 0; c < 3u; ++c)
        for (uint32_t e = 0; e < 2u * ns; ++e) ends[e][c] = bits.get(cb);
    for (uint32_t e = 0; e < 2u * ns; ++e) ends[e][3] = ab ? bits.get(ab) : 0u;
    const uint32_t channels = ab ? 4u : 3u;
    uint32_t colourBits = cb, alphaBits = ab;
    if (epb)
    {
        for (uint32_t e = 0; e < 2u * ns; ++e)
        {
            const uint32_t p = bits.get(1);
            for (uint32_t c = 0; c < channels; ++c) ends[e][c] = (ends[e][c] << 1) | p;
        }
        ++colourBits;
        ++alphaBits;
    }
    if (spb)
    {
        for (uint32_t s = 0; s < ns; ++s)
        {
            const uint32_t p = bits.get(1);
            for (uint32_t e = 2u * s; e < 2u * s + 2u; ++e)
                for (uint32_t c = 0; c < channels; ++c) ends[e][c] = (ends[e][c] << 1) | p;
        }
        ++colourBits;
        ++alphaBits;
    }
    for (uint32_t e = 0; e < 2u * ns; ++e)
    {
        for (uint32_t c = 0; c < 3u; ++c)
        {
            const uint32_t v = ends[e][c] << (8u - colourBits);
            ends[e][c] = v | (v >> colourBits);
        }
        if (ab)
        {
            const uint32_t v = ends[e][3] << (8u - alphaBits);
            ends[e][3] = v | (v >> alphaBits);
        }
        else
            ends[e][3] = 255u;
    }

    uint32_t subsets = 0u, anchor1 = 16u, anchor2 = 16u; // 2 bits per pixel; 16 = no such anchor
    if (ns == 2u)
    {
        subsets = kBc7Partition2[partition];
        anchor1 = kBc7Anchor2[partition];
    }
    else if (ns == 3u)
    {
        subsets = kBc7Partition3[partition];
        anchor1 = kBc7Anchor3a[partition];
        anchor2 = kBc7Anchor3b[partition];
    }
    uint32_t idx[16], idx2[16];
    for (uint32_t i = 0; i < 16u; ++i)
    {
        const bool isAnchor = i == 0u || i == anchor1 || i == anchor2;
        idx[i] = bits.get(isAnchor ? ib - 1u : ib);
    }
    for (uint32_t i = 0; i < 16u; ++i) idx2[i] = ib2 ? bits.get(i == 0u ? ib2 - 1u : ib2) : 0u;

    for (uint32_t i = 0; i < 16u; ++i)
    {
        const uint32_t s = (subsets >> (2u * i)) & 3u;
        const uint32_t *e0 = ends[2u * s], *e1 = ends[2u * s + 1u];
        uint32_t wc, wa;
        if (ib2)
        {
            // two index sets: colour from the primary, alpha from the secondary one, swapped by the selection bit
            const uint32_t wp = bc7_weight(ib, idx[i]), ws = bc7_weight(ib2, idx2[i]);
            wc = indexSelection ? ws : wp;
            wa = indexSelection ? wp : ws;
        }
        else
            wc = wa = bc7_weight(ib, idx[i]);
        uint32_t r = bc7_interpolate(e0[0], e1[0], wc), g = bc7_interpolate(e0[1], e1[1], wc),
                 b = bc7_interpolate(e0[2], e1[2], wc), a = bc7_interpolate(e0[3], e1[3], wa);
        if (rotation == 1u)
        {
            const uint32_t t = a;
            a = r;
            r = t;
        }
        else if (rotation == 2u)
        {
            const uint32_t t = a;
            a = g;
            g = t;
        }
        else if (rotation == 3u)
        {
            const uint32_t t = a;
            a = b;
            b = t;
        }
        texel[i] = r | (g << 8) | (b << 16) | (a << 24);
    }
}

} // namespace ppt
