// Deterministic fp32 math for the HIP kernels (gfx950).
//
// The reference's RNG is frac(sin(dot(seed, k)) * 43758.5453) (Assets/Shaders/random.h:8-12): one ulp of
// difference in sin() changes which light / BSDF lobe a path picks.  The vendor sin/cos/pow of the DX11
// driver are not specified, so this build fixes them: every function below is a stated sequence of IEEE-754
// binary32 operations (+ - * / sqrt floor, round-to-nearest, NO fused multiply-add) that gives bit-identical
// results on the host and on gfx950.  This file is compiled with
//     -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-gpu-flush-denormals-to-zero
// (see gmu-path-tracer_amd/build.py).  The specification is written out in DESIGN.md ("Deterministic math").
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace gmupt {

#define GM_HD __host__ __device__ __forceinline__

GM_HD uint32_t f2u(float f) { return __builtin_bit_cast(uint32_t, f); }
GM_HD float u2f(uint32_t u) { return __builtin_bit_cast(float, u); }

// HLSL min/max: the non-NaN operand wins (D3D11 functional spec)
GM_HD float hmin(float a, float b) { return (a < b || b != b) ? a : b; }
GM_HD float hmax(float a, float b) { return (a > b || b != b) ? a : b; }
GM_HD float hsaturate(float x) { return hmin(hmax(x, 0.0f), 1.0f); }
GM_HD float dfloor(float x) { return __builtin_floorf(x); }
GM_HD float dfrac(float x) { return x - dfloor(x); }
GM_HD float dsqrt(float x) { return __builtin_sqrtf(x); } // correctly rounded (compile flag)
GM_HD float dabs(float x) { return __builtin_fabsf(x); }

constexpr float kTwoOverPi = 0.636619772f;
constexpr float kPio2A = 1.5703125f;
constexpr float kPio2B = 4.837512969970703125e-4f;
constexpr float kPio2C = 7.54978995489188216e-8f;

struct SinCos { float s, c, q; };

GM_HD SinCos sincos_core(float x)
{
    SinCos r;
    float q = dfloor(x * kTwoOverPi + 0.5f);
    float t = ((x - q * kPio2A) - q * kPio2B) - q * kPio2C;
    float z = t * t;
    r.s = ((-1.9515295891e-4f * z + 8.3321608736e-3f) * z - 1.6666654611e-1f) * z * t + t;
    float c = ((2.443315711809948e-5f * z - 1.388731625493765e-3f) * z + 4.166664568298827e-2f) * z * z;
    c = c - 0.5f * z;
    c = c + 1.0f;
    r.c = c;
    r.q = q - 4.0f * dfloor(q * 0.25f);
    return r;
}

GM_HD float dsin(float x)
{
    SinCos r = sincos_core(x);
    if (r.q == 0.0f) return r.s;
    if (r.q == 1.0f) return r.c;
    if (r.q == 2.0f) return -r.s;
    if (r.q == 3.0f) return -r.c;
    return x - x;
}

GM_HD float dcos(float x)
{
    SinCos r = sincos_core(x);
    if (r.q == 0.0f) return r.c;
    if (r.q == 1.0f) return -r.s;
    if (r.q == 2.0f) return -r.c;
    if (r.q == 3.0f) return r.s;
    return x - x;
}

GM_HD float dlog2(float x)
{
    uint32_t u = f2u(x);
    int e = 0;
    if (u < 0x00800000u) { x = x * 16777216.0f; u = f2u(x); e = -24; }
    e += (int)(u >> 23) - 127;
    float m = u2f((u & 0x007FFFFFu) | 0x3F800000u);
    if (m > 1.41421354f) { m = m * 0.5f; e += 1; }
    float f = m - 1.0f;
    float s = f / (2.0f + f);
    float z = s * s;
    float p = z * (0.333333343f + z * (0.2f + z * (0.142857149f + z * 0.111111112f)));
    float ln = 2.0f * s + 2.0f * s * p;
    return (float)e + ln * 1.44269502f;
}

GM_HD float dexp2(float y)
{
    if (!(y > -150.0f)) return (y != y) ? y : 0.0f;
    if (y > 128.0f) return __builtin_inff();
    float n = dfloor(y + 0.5f);
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
    float s1 = u2f((uint32_t)(n1 + 127) << 23);
    float s2 = u2f((uint32_t)(n2 + 127) << 23);
    return p * s1 * s2;
}

GM_HD float dpow(float x, float y)
{
    if (x <= 0.0f) return 0.0f;
    return dexp2(y * dlog2(x));
}

// ---- float3 helpers with a fixed operation order ----
struct f3 { float x, y, z; };
GM_HD f3 mk3(float x, float y, float z) { f3 r; r.x = x; r.y = y; r.z = z; return r; }
GM_HD f3 operator+(f3 a, f3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
GM_HD f3 operator-(f3 a, f3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
GM_HD f3 operator*(f3 a, f3 b) { return mk3(a.x * b.x, a.y * b.y, a.z * b.z); }
GM_HD f3 operator/(f3 a, f3 b) { return mk3(a.x / b.x, a.y / b.y, a.z / b.z); }
GM_HD f3 operator*(f3 a, float s) { return mk3(a.x * s, a.y * s, a.z * s); }
GM_HD f3 neg3(f3 a) { return mk3(-a.x, -a.y, -a.z); }
GM_HD float dot3(f3 a, f3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
GM_HD f3 cross3(f3 a, f3 b) { return mk3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
GM_HD float length3(f3 a) { return dsqrt(dot3(a, a)); }
GM_HD f3 normalize3(f3 a) { float inv = 1.0f / length3(a); return a * inv; }

constexpr float kPi = 3.14159274f;     // structs.h:14 as binary32
constexpr float kInvPi = 0.318309873f; // structs.h:15 as binary32
constexpr float kEpsilon = 1e-8f;      // structs.h:10
constexpr float kEpsilonOffset = 1e-3f; // structs.h:11
constexpr float kFltMax = 3.402823466e+38f; // structs.h:9

// ---- RNG (Assets/Shaders/random.h:6-12) ----
struct Rng {
    float sx, sy, rsx, rsy;
    GM_HD void seed(uint32_t i, float rs0, float rs1)
    {
        float fi = (float)i;
        sx = dfrac(fi * kInvPi);
        sy = dfrac(fi * kPi);
        rsx = rs0; rsy = rs1;
    }
    GM_HD float next()
    {
        sx = sx - rsx;
        sy = sy - rsy;
        float d = sx * 12.9898f + sy * 78.233f;
        return dfrac(dsin(d) * 43758.5453f);
    }
};

} // namespace gmupt
