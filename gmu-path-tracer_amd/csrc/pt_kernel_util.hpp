// Small device helpers shared by the shading kernels (pt_kernels.hip) and the ray-cast kernels (pt_traverse.hip).
#pragma once
#include "pt_device.hpp"
#include "detmath.hpp"

namespace gmupt {

__device__ __forceinline__ float ldf(const RenderParams& p, uint32_t f, uint32_t i) { return p.state[(size_t)f * p.PS + i]; }
__device__ __forceinline__ uint32_t ldu(const RenderParams& p, uint32_t f, uint32_t i) { return __builtin_bit_cast(uint32_t, p.state[(size_t)f * p.PS + i]); }
__device__ __forceinline__ void stf(const RenderParams& p, uint32_t f, uint32_t i, float v) { p.state[(size_t)f * p.PS + i] = v; }
__device__ __forceinline__ void stu(const RenderParams& p, uint32_t f, uint32_t i, uint32_t v) { p.state[(size_t)f * p.PS + i] = __builtin_bit_cast(float, v); }
__device__ __forceinline__ f3 ld3(const RenderParams& p, uint32_t f, uint32_t i) { return mk3(ldf(p, f, i), ldf(p, f + 1, i), ldf(p, f + 2, i)); }
__device__ __forceinline__ void st3(const RenderParams& p, uint32_t f, uint32_t i, f3 v) { stf(p, f, i, v.x); stf(p, f + 1, i, v.y); stf(p, f + 2, i, v.z); }

__device__ __forceinline__ uint32_t lane_id() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }
// number of set bits of a 64-bit wave mask below this lane (the reference's NvWaveMultiPrefixExclusiveAdd(1, ballot))
__device__ __forceinline__ uint32_t prefix_rank(unsigned long long mask)
{
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}

__device__ __forceinline__ uint32_t cam_width(const gmupt_camera_buffer& c) { return (uint32_t)(1.0f / c.pixelSize[0]); }  // newPath.hlsl:30
__device__ __forceinline__ uint32_t cam_height(const gmupt_camera_buffer& c) { return (uint32_t)(1.0f / c.pixelSize[1]); } // newPath.hlsl:31

// local pixel index of a global screen coordinate, or kListEnd when outside the accumulation target
__device__ __forceinline__ uint32_t pixel_index(const RenderParams& p, uint32_t cx, uint32_t cy)
{
    uint32_t lx = cx - (p.tileEnabled ? p.tileX0 : 0u), ly = cy - (p.tileEnabled ? p.tileY0 : 0u);
    if (lx >= p.fbW || ly >= p.fbH) return kListEnd;
    return ly * p.fbW + lx;
}


} // namespace gmupt
