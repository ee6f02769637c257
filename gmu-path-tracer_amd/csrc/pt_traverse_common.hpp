// Pieces shared by the shipped ray-cast kernels (pt_traverse.hip) and the A/B variants (pt_traverse_variants.hip).
#pragma once
#include "pt_device.hpp"
#include "detmath.hpp"
#include "pt_kernel_util.hpp"

namespace gmupt {

constexpr int kTravBlock = 256;
constexpr int kMaxStack = 64;   // SBVH depth limit (Include/Nvidia-SBVH/SplitBVHBuilder.h:38)
// child descriptor of the packed tree: >= 0 inner node index; < 0 leaf starting at triangle record ~desc; kDone = nothing left
constexpr int kDone = (int)0x80000000;

struct TravCount { uint32_t inner, leaves, tris; };

// slab test on the packed node: same arithmetic as ray_aabb; v_min/v_max differ from hmin/hmax only in the sign of a zero
// result, which none of the comparisons below can observe
__device__ __forceinline__ float ray_box(float mnx, float mny, float mnz, float mxx, float mxy, float mxz, f3 o, f3 invdir)
{
    const float fx = (mxx - o.x) * invdir.x, fy = (mxy - o.y) * invdir.y, fz = (mxz - o.z) * invdir.z;
    const float nx = (mnx - o.x) * invdir.x, ny = (mny - o.y) * invdir.y, nz = (mnz - o.z) * invdir.z;
    const float t1 = __builtin_fminf(__builtin_fmaxf(fx, nx), __builtin_fminf(__builtin_fmaxf(fy, ny), __builtin_fmaxf(fz, nz)));
    const float t0 = __builtin_fmaxf(__builtin_fminf(fx, nx), __builtin_fmaxf(__builtin_fminf(fy, ny), __builtin_fminf(fz, nz)));
    return (t1 >= t0) ? (t0 > 0.0f ? t0 : t1) : -1.0f;
}

// as ray_box, but also returns the entry distance max(t0, 0) (0 when the origin is inside the box)
__device__ __forceinline__ float ray_box_entry(float mnx, float mny, float mnz, float mxx, float mxy, float mxz, f3 o, f3 invdir, float& entry)
{
    const float fx = (mxx - o.x) * invdir.x, fy = (mxy - o.y) * invdir.y, fz = (mxz - o.z) * invdir.z;
    const float nx = (mnx - o.x) * invdir.x, ny = (mny - o.y) * invdir.y, nz = (mnz - o.z) * invdir.z;
    const float t1 = __builtin_fminf(__builtin_fmaxf(fx, nx), __builtin_fminf(__builtin_fmaxf(fy, ny), __builtin_fmaxf(fz, nz)));
    const float t0 = __builtin_fmaxf(__builtin_fminf(fx, nx), __builtin_fmaxf(__builtin_fminf(fy, ny), __builtin_fminf(fz, nz)));
    entry = __builtin_fmaxf(t0, 0.0f);
    return (t1 >= t0) ? (t0 > 0.0f ? t0 : t1) : -1.0f;
}

// Moeller-Trumbore on a packed record (extensionRayCast.hlsl:38-62 == shadowRayCast.hlsl:16-40); returns false on a miss.
// Split into the fetch and the arithmetic so that a kernel can issue the fetch ahead of other work.
__device__ __forceinline__ void tri_fetch(const Tri48* tris, int i, float4& r0, float4& r1, float4& r2)
{
    const float4* r = reinterpret_cast<const float4*>(tris + i);
    r0 = r[0]; r1 = r[1]; r2 = r[2];
}

__device__ __forceinline__ bool tri_compute(const float4 r0, const float4 r1, const float4 r2, f3 o, f3 d, float& t, float& u, float& v, bool& last)
{
    last = __builtin_bit_cast(uint32_t, r2.y) != 0u;
    const f3 v0 = mk3(r0.x, r0.y, r0.z), e1 = mk3(r0.w, r1.x, r1.y), e2 = mk3(r1.z, r1.w, r2.x);
    const f3 pvec = cross3(d, e2);
    const float det = dot3(e1, pvec);
    if (det > -kEpsilon && det < kEpsilon) return false;
    const float invDet = 1.0f / det;
    const f3 tvec = o - v0;
    u = dot3(tvec, pvec) * invDet;
    if (u < 0.0f || u > 1.0f) return false;
    const f3 qvec = cross3(tvec, e1);
    v = dot3(d, qvec) * invDet;
    if (v < 0.0f || u + v > 1.0f) return false;
    t = dot3(e2, qvec) * invDet;
    return true;
}

// the same test without early exits: every lane does all the arithmetic and the rejections become one predicate (identical results:
// the same operations in the same order, the same comparisons; a lane that would have left early just carries values nobody reads)
__device__ __forceinline__ bool tri_compute_flat(const float4 r0, const float4 r1, const float4 r2, f3 o, f3 d, float& t, float& u, float& v, bool& last)
{
    last = __builtin_bit_cast(uint32_t, r2.y) != 0u;
    const f3 v0 = mk3(r0.x, r0.y, r0.z), e1 = mk3(r0.w, r1.x, r1.y), e2 = mk3(r1.z, r1.w, r2.x);
    const f3 pvec = cross3(d, e2);
    const float det = dot3(e1, pvec);
    const float invDet = 1.0f / det;
    const f3 tvec = o - v0;
    u = dot3(tvec, pvec) * invDet;
    const f3 qvec = cross3(tvec, e1);
    v = dot3(d, qvec) * invDet;
    t = dot3(e2, qvec) * invDet;
    const bool rejected = (det > -kEpsilon && det < kEpsilon) | (u < 0.0f) | (u > 1.0f) | (v < 0.0f) | (u + v > 1.0f);
    return !rejected;
}

__device__ __forceinline__ bool tri_test(const Tri48* tris, int i, f3 o, f3 d, float& t, float& u, float& v, bool& last)
{
    float4 r0, r1, r2;
    tri_fetch(tris, i, r0, r1, r2);
    return tri_compute(r0, r1, r2, o, d, t, u, v, last);
}

__device__ __forceinline__ void flush_counts(DevStats* st, const TravCount& tc, uint32_t rays, bool ext)
{
    // wave reduction, one atomic per wave and counter
    uint32_t a = tc.inner, b = tc.leaves, c = tc.tris, r = rays;
    for (int off = 32; off > 0; off >>= 1) { a += __shfl_down(a, off); b += __shfl_down(b, off); c += __shfl_down(c, off); r += __shfl_down(r, off); }
    if ((threadIdx.x & 63) == 0) {
        if (ext) { atomicAdd(&st->extInner, (unsigned long long)a); atomicAdd(&st->extLeaves, (unsigned long long)b); atomicAdd(&st->extTris, (unsigned long long)c); atomicAdd(&st->extRays, (unsigned long long)r); }
        else { atomicAdd(&st->shInner, (unsigned long long)a); atomicAdd(&st->shLeaves, (unsigned long long)b); atomicAdd(&st->shTris, (unsigned long long)c); atomicAdd(&st->shRays, (unsigned long long)r); }
    }
}

__device__ __forceinline__ void flush_sum(unsigned long long* dst, uint32_t v)
{
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
    if ((threadIdx.x & 63) == 0) atomicAdd(dst, (unsigned long long)v);
}

__device__ __forceinline__ void flush_wave_iters(DevStats* st, uint32_t wIn, uint32_t wTr, bool ext)
{
    uint32_t a = wIn, b = wTr;
    for (int off = 32; off > 0; off >>= 1) { a += __shfl_down(a, off); b += __shfl_down(b, off); }
    if ((threadIdx.x & 63) == 0) {
        if (ext) { atomicAdd(&st->extWaveInner, (unsigned long long)a); atomicAdd(&st->extWaveTris, (unsigned long long)b); }
        else { atomicAdd(&st->shWaveInner, (unsigned long long)a); atomicAdd(&st->shWaveTris, (unsigned long long)b); }
    }
}


} // namespace gmupt
