/*
 * oracle/detmath.h -- TEST INFRASTRUCTURE (CPU oracle), never linked into the product.
 *
 * Deterministic fp32 math used by the CPU restatement of the reference shaders.
 *
 * Why it exists: the reference's RNG is frac(sin(dot(seed,k)) * 43758.5453)
 * (Assets/Shaders/random.h:8-12).  A 1-ulp difference in sin() moves the random
 * number by ~2.6e-3, which flips light / lobe choices.  The DX11 driver's
 * sin/cos/pow/rcp/rsqrt are vendor-specific and not available here ("parity
 * unpinned"), so BOTH sides of the parity check (this oracle and
 * the HIP kernels) implement the same, fully specified sequence of IEEE-754
 * binary32 operations: + - * / sqrt floor, no FMA contraction, round-to-nearest.
 * The HIP side carries its own copy of this specification
 * (gmu-path-tracer_amd/csrc/detmath.hpp); the two files are independent sources
 * that must agree bit for bit -- tests/test_detmath_gpu.py checks exactly that.
 *
 * Compile with -ffp-contract=off (see oracle/Makefile).
 */
#ifndef GMUPT_ORACLE_DETMATH_H
#define GMUPT_ORACLE_DETMATH_H

#include <math.h>
#include <stdint.h>
#include <string.h>

static inline uint32_t o_asuint(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float o_asfloat(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

/* HLSL min/max: return the non-NaN operand (D3D11 functional spec) */
static inline float o_min(float a, float b) { return (a < b || b != b) ? a : b; }
static inline float o_max(float a, float b) { return (a > b || b != b) ? a : b; }
static inline float o_saturate(float x) { return o_min(o_max(x, 0.0f), 1.0f); }
static inline float o_frac(float x) { return x - floorf(x); }

/* Cody-Waite split of pi/2 (three terms, the first has 8 significant bits) */
#define O_TWO_OVER_PI 0.636619772f
#define O_PIO2_A 1.5703125f
#define O_PIO2_B 4.837512969970703125e-4f
#define O_PIO2_C 7.54978995489188216e-8f

static inline void o_sincos_core(float x, float* sp, float* cp, float* mq)
{
    float q = floorf(x * O_TWO_OVER_PI + 0.5f);
    float r = ((x - q * O_PIO2_A) - q * O_PIO2_B) - q * O_PIO2_C;
    float z = r * r;
    /* minimax polynomials on [-pi/4, pi/4] (Cephes single precision coefficients) */
    float s = ((-1.9515295891e-4f * z + 8.3321608736e-3f) * z - 1.6666654611e-1f) * z * r + r;
    float c = ((2.443315711809948e-5f * z - 1.388731625493765e-3f) * z + 4.166664568298827e-2f) * z * z;
    c = c - 0.5f * z;
    c = c + 1.0f;
    *sp = s; *cp = c;
    *mq = q - 4.0f * floorf(q * 0.25f); /* quadrant 0..3 as float */
}

static inline float o_sin(float x)
{
    float s, c, m;
    o_sincos_core(x, &s, &c, &m);
    if (m == 0.0f) return s;
    if (m == 1.0f) return c;
    if (m == 2.0f) return -s;
    if (m == 3.0f) return -c;
    return x - x; /* inf/NaN input -> NaN */
}

static inline float o_cos(float x)
{
    float s, c, m;
    o_sincos_core(x, &s, &c, &m);
    if (m == 0.0f) return c;
    if (m == 1.0f) return -s;
    if (m == 2.0f) return -c;
    if (m == 3.0f) return s;
    return x - x;
}

/* log2 for x > 0 (finite) */
static inline float o_log2(float x)
{
    uint32_t u = o_asuint(x);
    int e = 0;
    if (u < 0x00800000u) { x = x * 16777216.0f; u = o_asuint(x); e = -24; }
    e += (int)(u >> 23) - 127;
    float m = o_asfloat((u & 0x007FFFFFu) | 0x3F800000u);
    if (m > 1.41421354f) { m = m * 0.5f; e += 1; }
    float f = m - 1.0f;
    float s = f / (2.0f + f);
    float z = s * s;
    float p = z * (0.333333343f + z * (0.2f + z * (0.142857149f + z * 0.111111112f)));
    float ln = 2.0f * s + 2.0f * s * p;
    return (float)e + ln * 1.44269502f;
}

static inline float o_exp2(float y)
{
    if (!(y > -150.0f)) return (y != y) ? y : 0.0f;
    if (y > 128.0f) return INFINITY;
    float n = floorf(y + 0.5f);
    float f = y - n;
    float p = 1.535336188319500e-4f;
    p = p * f + 1.339887440266574e-3f;
    p = p * f + 9.618437357674640e-3f;
    p = p * f + 5.550332471162809e-2f;
    p = p * f + 2.402264791363012e-1f;
    p = p * f + 6.931472028550421e-1f;
    p = p * f + 1.0f;
    int ni = (int)n;
    int n1 = ni / 2, n2 = ni - n1;
    float s1 = o_asfloat((uint32_t)(n1 + 127) << 23);
    float s2 = o_asfloat((uint32_t)(n2 + 127) << 23);
    return p * s1 * s2;
}

/* HLSL pow(x,y) = exp2(y*log2(x)); x <= 0 -> 0 (exp2(-inf)) for the y > 0 used here */
static inline float o_pow(float x, float y)
{
    if (x <= 0.0f) return 0.0f;
    return o_exp2(y * o_log2(x));
}

#endif
