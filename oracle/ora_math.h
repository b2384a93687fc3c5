/*
 * oracle/ora_math.h — TEST INFRASTRUCTURE, not product code.
 *
 * Scalar fp32 arithmetic of the CPU restatement of prosper's path-tracing reference pass.
 * GLSL leaves the bit-level result of normalize/sin/cos/pow/dot open (Vulkan only bounds their
 * error), and the reference has no golden vectors, so the oracle fixes one
 * "arithmetic contract" and the HIP kernels are required to reproduce it bit for bit:
 *
 *   - every operation is IEEE-754 binary32, round-to-nearest-even, denormals kept, NO contraction
 *     (build with -ffp-contract=off); fmaf() is used only where written out below;
 *   - a/b and sqrtf are the correctly rounded operations;
 *   - dot(a,b) = fmaf(a.z,b.z, fmaf(a.y,b.y, a.x*b.x)); cross is NOT fused (a.y*b.z - a.z*b.y with
 *     both products rounded) so that cross(a,b) == -cross(b,a) exactly;
 *   - normalize(v) = v * (1 / sqrtf(dot(v,v))); vector / scalar = vector * (1 / scalar);
 *   - division by a constant (255, 511, 1023, 12.92, 1.055, PI) multiplies by the fp32 reciprocal;
 *   - sin/cos/exp2/log2/pow are the fixed polynomial kernels in this file (a few ulp, far inside
 *     Vulkan's precision bounds for GLSL.std.450 Sin/Cos/Pow);
 *   - min/max follow IEEE minNum/maxNum (a NaN operand loses), as GPU hardware does; GLSL leaves
 *     the NaN case undefined (GLSL 4.60 §8.3);
 *   - (integrator, oracle.c trace_path) a path whose throughput is exactly (0,0,0) after
 *     importanceSampleBounce ends: its later terms are 0 * X, and the ray the GLSL would go on to trace
 *     usually has a NaN direction (undefined per Vulkan); DESIGN.md section 3.
 *
 * Reference text restated here: res/shader/common/math.glsl:4-13 (PI, saturate, max3).
 */
#ifndef ORA_MATH_H
#define ORA_MATH_H

#include <math.h>
#include <stdint.h>
#include <string.h>

typedef struct ora_v2 { float x, y; } ora_v2;
typedef struct ora_v3 { float x, y, z; } ora_v3;
typedef struct ora_v4 { float x, y, z, w; } ora_v4;

/* math.glsl:4 — `#define PI 3.14159265`, rounded to binary32 when used in float expressions */
#define ORA_PI 3.14159265f
#define ORA_INV_PI (1.0f / 3.14159265f)

static inline uint32_t ora_f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float ora_u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

static inline float ora_max(float a, float b)
{
    if (a != a) return b;
    if (b != b) return a;
    return a < b ? b : a;
}
static inline float ora_min(float a, float b)
{
    if (a != a) return b;
    if (b != b) return a;
    return b < a ? b : a;
}
/* GLSL clamp(x, lo, hi) = min(max(x, lo), hi) */
static inline float ora_clamp(float x, float lo, float hi) { return ora_min(ora_max(x, lo), hi); }
/* math.glsl:5 */
static inline float ora_saturate(float x) { return ora_clamp(x, 0.0f, 1.0f); }
static inline float ora_abs(float x) { return ora_u2f(ora_f2u(x) & 0x7FFFFFFFu); }
/* GLSL sign(): 1, 0 (for +-0), -1; NaN -> 0 by the comparisons below */
static inline float ora_sign(float x) { return x > 0.0f ? 1.0f : (x < 0.0f ? -1.0f : 0.0f); }

/* float -> int conversion with the GPU's saturating behaviour (NaN -> 0), so that degenerate
 * inputs convert identically on both sides (C leaves out-of-range conversions undefined). */
static inline int32_t ora_f2i(float x)
{
    if (x != x) return 0;
    if (x >= 2147483648.0f) return 2147483647;
    if (x <= -2147483648.0f) return (int32_t)0x80000000u;
    return (int32_t)x;
}
static inline uint32_t ora_f2uint(float x)
{
    if (x != x) return 0u;
    if (x >= 4294967296.0f) return 0xFFFFFFFFu;
    if (x <= 0.0f) return 0u;
    return (uint32_t)x;
}

static inline ora_v3 ora_v3_make(float x, float y, float z) { ora_v3 r = {x, y, z}; return r; }
static inline ora_v3 ora_add(ora_v3 a, ora_v3 b) { return ora_v3_make(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline ora_v3 ora_sub(ora_v3 a, ora_v3 b) { return ora_v3_make(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline ora_v3 ora_mul(ora_v3 a, ora_v3 b) { return ora_v3_make(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline ora_v3 ora_scale(ora_v3 a, float s) { return ora_v3_make(a.x * s, a.y * s, a.z * s); }
/* vector / scalar = vector * (1 / scalar): one division, three multiplications */
static inline ora_v3 ora_divs(ora_v3 a, float s)
{
    const float inv = 1.0f / s;
    return ora_v3_make(a.x * inv, a.y * inv, a.z * inv);
}
static inline ora_v3 ora_neg(ora_v3 a) { return ora_v3_make(-a.x, -a.y, -a.z); }
static inline float ora_dot(ora_v3 a, ora_v3 b) { return fmaf(a.z, b.z, fmaf(a.y, b.y, a.x * b.x)); }
static inline ora_v3 ora_cross(ora_v3 a, ora_v3 b)
{
    return ora_v3_make(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
static inline float ora_length(ora_v3 a) { return sqrtf(ora_dot(a, a)); }
static inline ora_v3 ora_normalize(ora_v3 a)
{
    const float inv = 1.0f / sqrtf(ora_dot(a, a));
    return ora_scale(a, inv);
}
/* GLSL mix(a, b, t) = a*(1-t) + b*t */
static inline float ora_mix(float a, float b, float t) { return fmaf(b, t, a * (1.0f - t)); }
/* GLSL reflect(I, N) = I - 2*dot(N,I)*N */
static inline ora_v3 ora_reflect(ora_v3 i, ora_v3 n)
{
    const float k = 2.0f * ora_dot(n, i);
    return ora_v3_make(fmaf(-k, n.x, i.x), fmaf(-k, n.y, i.y), fmaf(-k, n.z, i.z));
}
/* math.glsl:8 */
static inline float ora_max3(ora_v3 v) { return ora_max(ora_max(v.x, v.y), v.z); }

/* ---- transcendental kernels (the contract; the HIP side has its own copy of the recipe) ---- */

/* sin and cos of x, |x| < 2^15: Cody-Waite reduction by pi/2 in three fmaf steps, then the
 * Cephes single-precision minimax polynomials on [-pi/4, pi/4]. */
static inline void ora_sincos(float x, float *s, float *c)
{
    const float k = rintf(x * 0.636619772f);
    float r = fmaf(-k, 1.57079625f, x);
    r = fmaf(-k, 7.54978942e-08f, r);
    r = fmaf(-k, 5.39030253e-15f, r);
    const float z = r * r;
    float ps = fmaf(-1.9515295891e-4f, z, 8.3321608736e-3f);
    ps = fmaf(ps, z, -1.6666654611e-1f);
    const float sr = fmaf(ps * z, r, r);
    float pc = fmaf(2.443315711809948e-5f, z, -1.388731625493765e-3f);
    pc = fmaf(pc, z, 4.166664568298827e-2f);
    const float cr = fmaf(pc * z, z, fmaf(-0.5f, z, 1.0f));
    const int32_t q = ora_f2i(k) & 3;
    const float ss = (q & 1) ? cr : sr;
    const float cc = (q & 1) ? sr : cr;
    *s = (q & 2) ? -ss : ss;
    *c = ((q + 1) & 2) ? -cc : cc;
}

/* log2(x) for normal x > 0: split exponent/mantissa into [sqrt(.5), sqrt(2)), atanh series. */
static inline float ora_log2(float x)
{
    uint32_t bits = ora_f2u(x);
    int32_t e = (int32_t)(bits >> 23) - 127;
    float m = ora_u2f((bits & 0x007FFFFFu) | 0x3F800000u);
    if (m > 1.41421354f)
    {
        m = m * 0.5f;
        e += 1;
    }
    const float s = (m - 1.0f) / (m + 1.0f);
    const float z = s * s;
    float p = fmaf(0.222222222f, z, 0.285714286f);
    p = fmaf(p, z, 0.4f);
    p = fmaf(p, z, 0.666666667f);
    p = fmaf(p, z, 2.0f);
    const float ln = p * s;
    return fmaf(ln, 1.44269504f, (float)e);
}

/* exp2(x) for x in [-126, 127]: split integer part, degree-7 Taylor in ln2 on [-.5, .5]. */
static inline float ora_exp2(float x)
{
    x = ora_clamp(x, -126.0f, 127.0f);
    const float n = rintf(x);
    const float f = x - n;
    float p = fmaf(1.52527338e-5f, f, 1.54035304e-4f);
    p = fmaf(p, f, 1.33335581e-3f);
    p = fmaf(p, f, 9.61812911e-3f);
    p = fmaf(p, f, 5.55041087e-2f);
    p = fmaf(p, f, 2.40226507e-1f);
    p = fmaf(p, f, 6.93147181e-1f);
    p = fmaf(p, f, 1.0f);
    const float scale = ora_u2f((uint32_t)(ora_f2i(n) + 127) << 23);
    return p * scale;
}

/* pow(x, y) for x >= 0 as exp2(y*log2(x)); x <= 0 gives 0 (GLSL: undefined for x < 0). */
static inline float ora_pow(float x, float y)
{
    if (!(x > 0.0f)) return 0.0f;
    return ora_exp2(y * ora_log2(x));
}
/* pow(x, 5.0) in brdf.glsl:23 by repeated multiplication */
static inline float ora_pow5(float x)
{
    const float x2 = x * x;
    return (x2 * x2) * x;
}

/* IEEE binary16 -> binary32, exact (GLSL unpackHalf2x16). */
static inline float ora_half_to_float(uint16_t h)
{
    const uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
    const uint32_t exp = (h >> 10) & 0x1Fu;
    const uint32_t man = h & 0x3FFu;
    if (exp == 0)
    {
        if (man == 0) return ora_u2f(sign);
        /* subnormal half: value = man * 2^-24, exact in binary32 */
        const float v = (float)man * 5.9604644775390625e-08f;
        return ora_u2f(ora_f2u(v) | sign);
    }
    if (exp == 31) return ora_u2f(sign | 0x7F800000u | (man << 13));
    return ora_u2f(sign | ((exp + 112u) << 23) | (man << 13));
}

/* binary32 -> binary16, round-to-nearest-even (glm::packHalf*, vkCmdBlitImage to RGBA16F). */
static inline uint16_t ora_float_to_half(float f)
{
    const uint32_t x = ora_f2u(f);
    const uint32_t sign = (x >> 16) & 0x8000u;
    const uint32_t ax = x & 0x7FFFFFFFu;
    if (ax >= 0x7F800000u) return (uint16_t)(sign | 0x7C00u | ((ax > 0x7F800000u) ? 0x200u : 0u));
    if (ax >= 0x477FF000u) return (uint16_t)(sign | 0x7C00u); /* rounds to >= 65520 -> inf */
    if (ax < 0x33000001u) return (uint16_t)sign;              /* <= 2^-25 rounds to zero */
    int32_t e = (int32_t)(ax >> 23) - 127;
    uint32_t m = (ax & 0x007FFFFFu) | 0x00800000u;
    uint32_t shift;
    uint32_t he;
    if (e < -14)
    {
        shift = (uint32_t)(13 + (-14 - e));
        he = 0;
    }
    else
    {
        shift = 13;
        he = (uint32_t)(e + 15);
    }
    uint32_t hm = m >> shift;
    const uint32_t rem = m & ((1u << shift) - 1u);
    const uint32_t half = 1u << (shift - 1);
    if (rem > half || (rem == half && (hm & 1u))) hm += 1;
    /* hm still carries the implicit bit for normals; adding it to (he-1)<<10 handles carries */
    uint32_t out = (he == 0) ? hm : (((he - 1u) << 10) + hm);
    return (uint16_t)(sign | out);
}

#endif /* ORA_MATH_H */
