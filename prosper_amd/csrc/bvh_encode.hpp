// bvh_encode.hpp — how a 4-wide node's child boxes are padded and stored (pt_scene.hpp BvhNode), shared by the host
// builder (bvh_build.cpp: the emitter) and the device refit (pt_kernels.hip: refit of moved instances), so that a refit
// writes exactly the bytes the emitter would have written for the same tree.  Integer and IEEE fp32 arithmetic only.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "pt_scene.hpp"

namespace ppt
{

#define PPT_ENC __host__ __device__ inline

struct EncBox
{
    float lo[3], hi[3];
};

PPT_ENC uint32_t enc_bits(float f) { return __builtin_bit_cast(uint32_t, f); }
PPT_ENC float enc_float(uint32_t u) { return __builtin_bit_cast(float, u); }
PPT_ENC float enc_min(float a, float b) { return b < a ? b : a; }
PPT_ENC float enc_max(float a, float b) { return b > a ? b : a; }
PPT_ENC float enc_abs(float a) { return enc_float(enc_bits(a) & 0x7FFFFFFFu); }

// the next representable float towards +inf / -inf (finite inputs; +-0 -> the smallest subnormal of that sign)
PPT_ENC float enc_next_up(float f)
{
    const uint32_t b = enc_bits(f);
    if ((b & 0x7FFFFFFFu) == 0u) return enc_float(0x00000001u);
    if (b >= 0x7F800000u && !(b & 0x80000000u)) return f; // +inf, NaN
    return enc_float((b & 0x80000000u) ? b - 1u : b + 1u);
}
PPT_ENC float enc_next_down(float f)
{
    const uint32_t b = enc_bits(f);
    if ((b & 0x7FFFFFFFu) == 0u) return enc_float(0x80000001u);
    if ((b & 0x7FFFFFFFu) >= 0x7F800000u && (b & 0x80000000u)) return f; // -inf, NaN
    return enc_float((b & 0x80000000u) ? b + 1u : b - 1u);
}

// binary32 -> binary16, round to nearest even
PPT_ENC uint16_t half_rne(float f)
{
    const uint32_t x = enc_bits(f);
    const uint32_t sign = (x >> 16) & 0x8000u;
    const uint32_t ax = x & 0x7FFFFFFFu;
    if (ax >= 0x7F800000u) return (uint16_t)(sign | 0x7C00u | ((ax > 0x7F800000u) ? 0x200u : 0u));
    if (ax >= 0x477FF000u) return (uint16_t)(sign | 0x7C00u);
    if (ax < 0x33000001u) return (uint16_t)sign;
    const int32_t e = (int32_t)(ax >> 23) - 127;
    const uint32_t m = (ax & 0x007FFFFFu) | 0x00800000u;
    const uint32_t shift = e < -14 ? (uint32_t)(13 + (-14 - e)) : 13u;
    const uint32_t he = e < -14 ? 0u : (uint32_t)(e + 15);
    uint32_t hm = m >> shift;
    const uint32_t rem = m & ((1u << shift) - 1u);
    const uint32_t halfway = 1u << (shift - 1);
    if (rem > halfway || (rem == halfway && (hm & 1u))) hm += 1;
    return (uint16_t)(sign | ((he == 0) ? hm : (((he - 1u) << 10) + hm)));
}
PPT_ENC float half_value(uint16_t h)
{
    const uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
    const uint32_t exp = (h >> 10) & 0x1Fu, man = h & 0x3FFu;
    uint32_t bits;
    if (exp == 0)
        bits = enc_bits((float)man * 5.9604644775390625e-08f) | sign;
    else if (exp == 31)
        bits = sign | 0x7F800000u | (man << 13);
    else
        bits = sign | ((exp + 112u) << 23) | (man << 13);
    return enc_float(bits);
}
// next representable half towards +inf / -inf (as ordered values, handling the sign-magnitude encoding)
PPT_ENC uint16_t half_next_up(uint16_t h)
{
    if ((h & 0x7FFFu) == 0) return 0x0001u;          // +-0 -> smallest positive
    if (h & 0x8000u) return (uint16_t)(h - 1u);       // negative: magnitude shrinks
    return (h == 0x7C00u) ? h : (uint16_t)(h + 1u);   // positive: magnitude grows (stop at +inf)
}
PPT_ENC uint16_t half_next_down(uint16_t h)
{
    if ((h & 0x7FFFu) == 0) return 0x8001u;
    if (h & 0x8000u) return (h == 0xFC00u) ? h : (uint16_t)(h + 1u);
    return (uint16_t)(h - 1u);
}
// binary32 -> binary16 rounded toward -inf / +inf: child boxes are stored as halfs and must stay conservative (they may
// only grow)
PPT_ENC uint16_t half_floor(float f)
{
    uint16_t h = half_rne(f);
    if (half_value(h) > f) h = half_next_down(h);
    return h;
}
PPT_ENC uint16_t half_ceil(float f)
{
    uint16_t h = half_rne(f);
    if (half_value(h) < f) h = half_next_up(h);
    return h;
}

// Conservative padding.  A valid hit (pt_device.hpp box_guard) lies in the ray's interval through the
// triangle's bounds grown by 2^-16 of its largest |coordinate|; a node box must contain those guard
// boxes (1.6e-5 > 2^-16 of the box's own largest |coordinate| does, and the term is monotone up the
// tree).  On top of that the node test works on fl(o - nodeOrigin), off by up to 2^-24 of the distance
// between the ray origin and the node: `slack` = 2e-6 * scene diagonal (32 * 2^-24) covers ray origins
// up to ~16 scene diagonals away; the relative term covers the fp32 subtractions that form the offsets.
PPT_ENC void enc_padded(const EncBox &b, float coeff, float slack, float lo[3], float hi[3])
{
    float mall = 0.0f;
    for (int k = 0; k < 3; ++k) mall = enc_max(mall, enc_max(enc_abs(b.lo[k]), enc_abs(b.hi[k])));
    for (int k = 0; k < 3; ++k)
    {
        const float pad = coeff * mall + 1e-6f * (b.hi[k] - b.lo[k]) + slack;
        lo[k] = b.lo[k] - pad;
        hi[k] = b.hi[k] + pad;
    }
}

// slack of a tree whose root box is `scene`
PPT_ENC float enc_slack(const EncBox &scene)
{
    const float dx = scene.hi[0] - scene.lo[0], dy = scene.hi[1] - scene.lo[1], dz = scene.hi[2] - scene.lo[2];
    // (__builtin_sqrtf is the correctly rounded square root on both sides; HIP's __fsqrt_rn is the native approximation)
    return 2e-6f * __builtin_sqrtf(dx * dx + dy * dy + dz * dz) + 1e-30f;
}

// Origin and the k child boxes of a node (boxes[c] = the exact bounds of child c's triangles) -> node.origin, node.lo,
// node.hi; the slots from k on get lo = hi = +inf (half 0x7C00), which no ray can enter.  node.child is the caller's.
PPT_ENC void enc_node_boxes(const EncBox boxes[4], uint32_t k, float padCoeff, float slack, BvhNode &node)
{
    node.reserved = k; // children in use (a leaf reference of triangle 0 alone reads like the unused marker ~0)
    for (int a = 0; a < 3; ++a)
        for (int c = 0; c < 4; ++c)
        {
            node.lo[a][c] = 0x7C00u;
            node.hi[a][c] = 0x7C00u;
        }
    float lo[4][3], hi[4][3];
    for (int a = 0; a < 3; ++a) node.origin[a] = enc_float(0x7F800000u);
    for (uint32_t c = 0; c < k; ++c)
    {
        enc_padded(boxes[c], padCoeff, slack, lo[c], hi[c]);
        for (int a = 0; a < 3; ++a) node.origin[a] = enc_min(node.origin[a], lo[c][a]);
    }
    for (uint32_t c = 0; c < k; ++c)
        for (int a = 0; a < 3; ++a)
        {
            // offsets from the node origin; the fp32 subtraction is pushed one ulp outward before the outward half
            // rounding, so origin + offset never lies inside the padded box
            const float offLo = enc_next_down(lo[c][a] - node.origin[a]);
            const float offHi = enc_next_up(hi[c][a] - node.origin[a]);
            node.lo[a][c] = half_floor(enc_max(offLo, 0.0f));
            node.hi[a][c] = half_ceil(offHi);
        }
}

} // namespace ppt
