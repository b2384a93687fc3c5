// pt_math.hpp — fp32 arithmetic of the HIP path tracer (gfx950).
//
// GLSL does not pin the bits of normalize/sin/cos/pow/dot, and prosper has no golden vectors, so
// this build fixes one arithmetic contract (DESIGN.md "Arithmetic contract") that the kernels and
// the CPU oracle both implement independently and that the parity tests check bit for bit:
//   * IEEE binary32, round-to-nearest-even, denormals kept, no contraction (-ffp-contract=off);
//     fmaf only where written;
//   * a/b and sqrt are correctly rounded (hipcc default -fhip-fp32-correctly-rounded-divide-sqrt);
//   * dot(a,b) = fma(a.z,b.z, fma(a.y,b.y, a.x*b.x)); cross is NOT fused (exact antisymmetry);
//     normalize(v) = v * (1/sqrt(dot(v,v))); vector / scalar = vector * (1 / scalar); division by a
//     constant multiplies by its fp32 reciprocal;
//   * sin/cos/exp2/log2 are the fixed polynomial kernels below (no v_sin_f32/v_exp_f32, whose
//     results are not reproducible off-GPU);
//   * min/max are IEEE minNum/maxNum (a NaN operand loses).
// Reference text: res/shader/common/math.glsl:4-13.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#define PPT_HD __host__ __device__ __forceinline__

namespace ppt
{

struct f2 { float x, y; };
struct f3 { float x, y, z; };
struct f4 { float x, y, z, w; };

constexpr float kPi = 3.14159265f; // math.glsl:4
constexpr float kInvPi = 1.0f / 3.14159265f;
constexpr float kTwoPi = 6.2831853f; // the GLSL front end folds (2.0 * PI)
constexpr float kInf = __builtin_huge_valf();

PPT_HD uint32_t f2u(float f) { return __builtin_bit_cast(uint32_t, f); }
PPT_HD float u2f(uint32_t u) { return __builtin_bit_cast(float, u); }

PPT_HD float fmax_(float a, float b)
{
    if (a != a) return b;
    if (b != b) return a;
    return a < b ? b : a;
}
PPT_HD float fmin_(float a, float b)
{
    if (a != a) return b;
    if (b != b) return a;
    return b < a ? b : a;
}
PPT_HD float clamp_(float x, float lo, float hi) { return fmin_(fmax_(x, lo), hi); }
PPT_HD float saturate(float x) { return clamp_(x, 0.0f, 1.0f); }
PPT_HD float fabs_(float x) { return u2f(f2u(x) & 0x7FFFFFFFu); }
PPT_HD float sign_(float x) { return x > 0.0f ? 1.0f : (x < 0.0f ? -1.0f : 0.0f); }

// float -> int/uint with saturation and NaN -> 0 (what v_cvt_i32_f32 / v_cvt_u32_f32 do; spelled
// out so the host build of this header agrees)
PPT_HD int32_t f2i(float x)
{
    if (x != x) return 0;
    if (x >= 2147483648.0f) return 2147483647;
    if (x <= -2147483648.0f) return (int32_t)0x80000000u;
    return (int32_t)x;
}
PPT_HD uint32_t f2uint(float x)
{
    if (x != x) return 0u;
    if (x >= 4294967296.0f) return 0xFFFFFFFFu;
    if (x <= 0.0f) return 0u;
    return (uint32_t)x;
}

PPT_HD f3 make3(float x, float y, float z) { return f3{x, y, z}; }
PPT_HD f3 operator+(f3 a, f3 b) { return f3{a.x + b.x, a.y + b.y, a.z + b.z}; }
PPT_HD f3 operator-(f3 a, f3 b) { return f3{a.x - b.x, a.y - b.y, a.z - b.z}; }
PPT_HD f3 operator*(f3 a, f3 b) { return f3{a.x * b.x, a.y * b.y, a.z * b.z}; }
PPT_HD f3 operator*(f3 a, float s) { return f3{a.x * s, a.y * s, a.z * s}; }
// vector / scalar = vector * (1 / scalar): one division, three multiplications
PPT_HD f3 operator/(f3 a, float s)
{
    const float inv = 1.0f / s;
    return f3{a.x * inv, a.y * inv, a.z * inv};
}
PPT_HD f3 operator-(f3 a) { return f3{-a.x, -a.y, -a.z}; }
PPT_HD float dot(f3 a, f3 b) { return __builtin_fmaf(a.z, b.z, __builtin_fmaf(a.y, b.y, a.x * b.x)); }
PPT_HD f3 cross(f3 a, f3 b)
{
    return f3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
PPT_HD float sqrt_(float x) { return __builtin_sqrtf(x); }
PPT_HD float length(f3 a) { return sqrt_(dot(a, a)); }
PPT_HD f3 normalize(f3 a)
{
    const float inv = 1.0f / sqrt_(dot(a, a));
    return a * inv;
}
PPT_HD float mix(float a, float b, float t) { return __builtin_fmaf(b, t, a * (1.0f - t)); }
PPT_HD f3 reflect(f3 i, f3 n)
{
    const float k = 2.0f * dot(n, i);
    return f3{__builtin_fmaf(-k, n.x, i.x), __builtin_fmaf(-k, n.y, i.y), __builtin_fmaf(-k, n.z, i.z)};
}
PPT_HD float max3(f3 v) { return fmax_(fmax_(v.x, v.y), v.z); }

// sin and cos of x, |x| < 2^15: three-step Cody-Waite reduction by pi/2, Cephes sinf/cosf kernels.
PPT_HD void sincos_(float x, float &s, float &c)
{
#if defined(PPT_EXPERIMENT_NATIVE_TRANSCENDENTALS) && defined(__HIP_DEVICE_COMPILE__)
    // TIMING ONLY (scripts/build_variant.sh native "-DPPT_EXPERIMENT_NATIVE_TRANSCENDENTALS"; profiles/r04_transcendentals.txt):
    // v_sin_f32 / v_cos_f32 - what the contract's polynomial kernels cost against the hardware's; breaks bit parity
    s = __builtin_amdgcn_sinf(x * 0.15915494309189532f);
    c = __builtin_amdgcn_cosf(x * 0.15915494309189532f);
    return;
#endif
    const float k = __builtin_rintf(x * 0.636619772f);
    float r = __builtin_fmaf(-k, 1.57079625f, x);
    r = __builtin_fmaf(-k, 7.54978942e-08f, r);
    r = __builtin_fmaf(-k, 5.39030253e-15f, r);
    const float z = r * r;
    float ps = __builtin_fmaf(-1.9515295891e-4f, z, 8.3321608736e-3f);
    ps = __builtin_fmaf(ps, z, -1.6666654611e-1f);
    const float sr = __builtin_fmaf(ps * z, r, r);
    float pc = __builtin_fmaf(2.443315711809948e-5f, z, -1.388731625493765e-3f);
    pc = __builtin_fmaf(pc, z, 4.166664568298827e-2f);
    const float cr = __builtin_fmaf(pc * z, z, __builtin_fmaf(-0.5f, z, 1.0f));
    const int32_t q = f2i(k) & 3;
    const float ss = (q & 1) ? cr : sr;
    const float cc = (q & 1) ? sr : cr;
    s = (q & 2) ? -ss : ss;
    c = ((q + 1) & 2) ? -cc : cc;
}

// log2 of a normal x > 0: mantissa folded to [sqrt(.5), sqrt(2)), atanh series in (m-1)/(m+1)
PPT_HD float log2_(float x)
{
#if defined(PPT_EXPERIMENT_NATIVE_TRANSCENDENTALS) && defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_logf(x); // v_log_f32 (timing only)
#endif
    const uint32_t bits = f2u(x);
    int32_t e = (int32_t)(bits >> 23) - 127;
    float m = u2f((bits & 0x007FFFFFu) | 0x3F800000u);
    if (m > 1.41421354f)
    {
        m = m * 0.5f;
        e += 1;
    }
    const float s = (m - 1.0f) / (m + 1.0f);
    const float z = s * s;
    float p = __builtin_fmaf(0.222222222f, z, 0.285714286f);
    p = __builtin_fmaf(p, z, 0.4f);
    p = __builtin_fmaf(p, z, 0.666666667f);
    p = __builtin_fmaf(p, z, 2.0f);
    const float ln = p * s;
    return __builtin_fmaf(ln, 1.44269504f, (float)e);
}

// exp2 on [-126, 127]: integer part by exponent bits, degree-7 Taylor in ln2 on [-.5, .5]
PPT_HD float exp2_(float x)
{
#if defined(PPT_EXPERIMENT_NATIVE_TRANSCENDENTALS) && defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_exp2f(x); // v_exp_f32 (timing only)
#endif
    x = clamp_(x, -126.0f, 127.0f);
    const float n = __builtin_rintf(x);
    const float f = x - n;
    float p = __builtin_fmaf(1.52527338e-5f, f, 1.54035304e-4f);
    p = __builtin_fmaf(p, f, 1.33335581e-3f);
    p = __builtin_fmaf(p, f, 9.61812911e-3f);
    p = __builtin_fmaf(p, f, 5.55041087e-2f);
    p = __builtin_fmaf(p, f, 2.40226507e-1f);
    p = __builtin_fmaf(p, f, 6.93147181e-1f);
    p = __builtin_fmaf(p, f, 1.0f);
    const float scale = u2f((uint32_t)(f2i(n) + 127) << 23);
    return p * scale;
}

PPT_HD float pow_(float x, float y)
{
    if (!(x > 0.0f)) return 0.0f;
    return exp2_(y * log2_(x));
}
// pow(x, 5.0), brdf.glsl:23
PPT_HD float pow5(float x)
{
    const float x2 = x * x;
    return (x2 * x2) * x;
}

// binary16 -> binary32, exact (unpackHalf2x16)
PPT_HD float half_to_float(uint32_t h)
{
#if defined(__HIP_DEVICE_COMPILE__)
    // v_cvt_f32_f16: binary16 -> binary32 is exact for every input (subnormals included: f16
    // denormals are enabled in the default float mode), i.e. the same function as the bit recipe below
    return (float)__builtin_bit_cast(_Float16, (uint16_t)h);
#endif
    const uint32_t sign = (h & 0x8000u) << 16;
    const uint32_t exp = (h >> 10) & 0x1Fu;
    const uint32_t man = h & 0x3FFu;
    if (exp == 0)
    {
        if (man == 0) return u2f(sign);
        const float v = (float)man * 5.9604644775390625e-08f; // man * 2^-24, exact
        return u2f(f2u(v) | sign);
    }
    if (exp == 31) return u2f(sign | 0x7F800000u | (man << 13));
    return u2f(sign | ((exp + 112u) << 23) | (man << 13));
}

// binary32 -> binary16, round-to-nearest-even (the RGBA32F -> RGBA16F blit)
PPT_HD uint32_t float_to_half(float f)
{
    const uint32_t x = f2u(f);
    const uint32_t sign = (x >> 16) & 0x8000u;
    const uint32_t ax = x & 0x7FFFFFFFu;
    if (ax >= 0x7F800000u) return sign | 0x7C00u | ((ax > 0x7F800000u) ? 0x200u : 0u);
    if (ax >= 0x477FF000u) return sign | 0x7C00u;
    if (ax < 0x33000001u) return sign;
    const int32_t e = (int32_t)(ax >> 23) - 127;
    const uint32_t m = (ax & 0x007FFFFFu) | 0x00800000u;
    const uint32_t shift = e < -14 ? (uint32_t)(13 + (-14 - e)) : 13u;
    const uint32_t he = e < -14 ? 0u : (uint32_t)(e + 15);
    uint32_t hm = m >> shift;
    const uint32_t rem = m & ((1u << shift) - 1u);
    const uint32_t halfway = 1u << (shift - 1);
    if (rem > halfway || (rem == halfway && (hm & 1u))) hm += 1;
    const uint32_t out = (he == 0) ? hm : (((he - 1u) << 10) + hm);
    return sign | out;
}

} // namespace ppt
