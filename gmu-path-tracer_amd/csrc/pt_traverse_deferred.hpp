// Building blocks of the deferred-leaf ray-cast kernels (pt_traverse.hip: the shipped kernels; pt_traverse_variants.hip: the A/B rungs).
#pragma once
#include "pt_traverse_common.hpp"

namespace gmupt {

#ifndef GMUPT_DEF_STACK
#define GMUPT_DEF_STACK 24
#endif
#ifndef GMUPT_DEF_FIFO
#define GMUPT_DEF_FIFO 4
#endif
// ------------------------------------------------------------------------------------------------ deferred-leaf variants
// Counter evidence on MI355X (profiles/r01_*): the ray casts are bound by VALU issue at low lane utilisation (one wave64
// instruction stream serves the inner-node lanes and the triangle lanes in turn: 44 % / 17 % of the lanes active), not by
// HBM, L2 or the vector-memory pipeline.  The reference's traversal has a property that removes the mix: boxes are never
// pruned against the current hit (extensionRayCast.hlsl:79-94,132-159), so the walk through the tree does not depend on
// any triangle test.  A lane therefore keeps walking and only QUEUES the leaves it reaches (per-lane FIFO in LDS); the wave
// runs its triangle tests in bursts when most lanes have leaves pending.  Each ray's leaves are still tested in visit order
// (FIFO) with the strict `t < distance` rule, so ties resolve exactly as in the reference; the shadow ray (any hit) may
// walk a little further than needed before its occluder is found, which cannot change its boolean result.
#ifndef GMUPT_DEF_BLOCK
#define GMUPT_DEF_BLOCK 1024
#endif
constexpr int kDefBlock = GMUPT_DEF_BLOCK;    // the waves of a workgroup share one LDS copy of the top of the tree
constexpr int kDefStack = GMUPT_DEF_STACK;     // LDS stack entries per lane incl. the sentinel; trees deeper than kDefStack - 2 use the overflow-checked instantiation
constexpr int kFifo = GMUPT_DEF_FIFO;          // pending leaves per lane
// One workgroup per CU owns the whole LDS: 24 + 4 words per lane for 1024 lanes (112 KB) and the 768 hottest nodes of the tree (48 KB).
// A larger top is worth more than a second workgroup's copy of a smaller one: 60 -> 68 % of the node visits of the bench scene and
// 46 -> 59 % of those of the 10 M-triangle scene are served from LDS (ray cast -1 % / -13 %).
static_assert((kDefStack + kFifo) * kDefBlock * 4 + kTopTreeNodes * 64 <= 160 * 1024, "LDS of one CU");
// Trees deeper than kDefStack - 2 run the instantiations with a bounds-checked global overflow (OVF).  They pay for the check anyway, so
// their LDS holds only kDeepStack entries per lane and the freed 32 KB hold more of the tree (10 M-triangle scene: 59 -> 63 % of the node
// visits from LDS, ray cast -3 %; on a shallow tree the check alone costs +4 %, so the plain layout keeps its 24 entries).
#ifndef GMUPT_DEEP_STACK
#define GMUPT_DEEP_STACK 16
#endif
constexpr int kDeepStack = GMUPT_DEEP_STACK;
template <bool OVF> constexpr int kDefLdsStack = OVF ? kDeepStack : kDefStack;
template <bool OVF> constexpr int kDefLdsTop = OVF ? kDeepTopTreeNodes : kTopTreeNodes;
template <bool OVF> __device__ __forceinline__ uint32_t top_count(const TravScene& ts) { return OVF ? ts.topCountDeep : ts.topCount; }
static_assert((kDeepStack + kFifo) * kDefBlock * 4 + kDeepTopTreeNodes * 64 <= 160 * 1024, "LDS of one CU (deep-tree layout)");
static_assert(kDeepStack <= kDefStack && kDeepTopTreeNodes >= kTopTreeNodes, "the two layouts share one node numbering and one overflow buffer");
static_assert((kFifo & (kFifo - 1)) == 0, "the leaf FIFO is indexed modulo its size");

template <bool OVF>
struct DefStack {
    int* lds; int* ovf; uint32_t ovfStride; uint32_t ptr; // lds[0] holds kDone for good: popping an empty stack ends the walk
    __device__ __forceinline__ void reset() { ptr = 1; }
    __device__ __forceinline__ void push(int v, DevStats* st)
    {
        if (!OVF || ptr < kDefLdsStack<OVF>) lds[ptr * kDefBlock] = v;
        else if (ptr < kMaxStack + 1) ovf[(size_t)(ptr - kDefLdsStack<OVF>) * ovfStride] = v;
        else atomicOr(&st->stackOverflow, 1u);   // (bit 1 is the watchdog's: a plain store would erase it)
        ptr++;
    }
    __device__ __forceinline__ int pop()
    {
        --ptr;
        if (!OVF || ptr < kDefLdsStack<OVF>) return lds[ptr * kDefBlock];
        if (ptr < kMaxStack + 1) return ovf[(size_t)(ptr - kDefLdsStack<OVF>) * ovfStride];
        return kDone;
    }
};

// Node fetch with EXPLICIT address spaces.  Written with generic pointers, the compiler merges the two branches into one FLAT load
// from a selected address (shared aperture or global): correct, but a FLAT access to LDS goes through the texture addresser like a
// global one, and that unit is what bounds this kernel.  Typed pointers keep an LDS read a ds_read_b128 and a global read a
// global_load_dwordx4, each under its own exec mask.
typedef float vec4f __attribute__((ext_vector_type(4)));
typedef int vec2i __attribute__((ext_vector_type(2)));
#define GMUPT_AS_LDS __attribute__((address_space(3)))
#define GMUPT_AS_GLOBAL __attribute__((address_space(1)))

template <bool TOP>
__device__ __forceinline__ void load_node(const TravScene& ts, const float4* s_top, uint32_t topCount, int cur, vec4f& a, vec4f& b, vec4f& c, vec2i& d)
{
    // LDS lanes first: the global lanes then only wait for the (short) LDS reads before their loads may target the same registers,
    // and nothing waits for the global loads before they are used
    const bool inTop = TOP && (uint32_t)cur < topCount;
    if (inTop) {
        const GMUPT_AS_LDS vec4f* n = (const GMUPT_AS_LDS vec4f*)(s_top) + cur * 4;
        a = n[0]; b = n[1]; c = n[2]; d = *(const GMUPT_AS_LDS vec2i*)(n + 3);
    }
    asm volatile("" ::: "memory"); // keeps the two regions apart and in this order (the optimiser would fold them into if / else, global first)
    if (!inTop) {
        const GMUPT_AS_GLOBAL vec4f* n = (const GMUPT_AS_GLOBAL vec4f*)(ts.nodes) + (size_t)cur * 4;
        a = n[0]; b = n[1]; c = n[2];
        const unsigned long long links = *(const GMUPT_AS_GLOBAL unsigned long long*)(n + 3);  // exactly 8 bytes: no spare destination registers
        d.x = (int)(uint32_t)links; d.y = (int)(uint32_t)(links >> 32);
    }
}

// The same fetches as raw buffer loads: exactly 3 x 16 + 8 bytes per node and 2 x 16 + 8 per triangle record, one request each, no
// re-grouping by the optimiser (which turns the triangle record into four overlapping loads and the 8-byte link pair into 16 bytes,
// i.e. spare destination registers that later instructions have to wait for), and hardware bounds checking for free.
typedef uint32_t vec4u __attribute__((ext_vector_type(4)));
typedef uint32_t vec2u __attribute__((ext_vector_type(2)));
constexpr int kBufferRsrcFlags = 0x00020000; // raw buffer, 32-bit data format (gfx9 family)

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, uint32_t bytes)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, kBufferRsrcFlags);
}

template <bool TOP>
__device__ __forceinline__ void load_node_buf(__amdgpu_buffer_rsrc_t nodes, const float4* s_top, uint32_t topCount, int cur, vec4f& a, vec4f& b, vec4f& c, vec2i& d)
{
    const bool inTop = TOP && (uint32_t)cur < topCount;
    if (inTop) {
        const GMUPT_AS_LDS vec4f* n = (const GMUPT_AS_LDS vec4f*)(s_top) + cur * 4;
        a = n[0]; b = n[1]; c = n[2]; d = *(const GMUPT_AS_LDS vec2i*)(n + 3);
    }
    asm volatile("" ::: "memory"); // LDS lanes first, see load_node
    if (!inTop) {
        const int off = cur * 64;
        a = __builtin_bit_cast(vec4f, __builtin_amdgcn_raw_buffer_load_b128(nodes, off, 0, 0));
        b = __builtin_bit_cast(vec4f, __builtin_amdgcn_raw_buffer_load_b128(nodes, off + 16, 0, 0));
        c = __builtin_bit_cast(vec4f, __builtin_amdgcn_raw_buffer_load_b128(nodes, off + 32, 0, 0));
        d = __builtin_bit_cast(vec2i, __builtin_amdgcn_raw_buffer_load_b64(nodes, off + 48, 0, 0));   // (16 bytes here instead of 8: +-0, profiles/r02_config3/bench_wide_links.json)
    }
}

typedef float vec2f __attribute__((ext_vector_type(2)));
// (whole-vector bit casts only: __builtin_bit_cast of a vector ELEMENT lvalue reads element 0 with this compiler)
__device__ __forceinline__ void tri_fetch_buf(__amdgpu_buffer_rsrc_t tris, int i, vec4f& r0, vec4f& r1, vec2f& r2)
{
    const int off = i * 48;
    r0 = __builtin_bit_cast(vec4f, __builtin_amdgcn_raw_buffer_load_b128(tris, off, 0, 0));
    r1 = __builtin_bit_cast(vec4f, __builtin_amdgcn_raw_buffer_load_b128(tris, off + 16, 0, 0));
    r2 = __builtin_bit_cast(vec2f, __builtin_amdgcn_raw_buffer_load_b64(tris, off + 32, 0, 0));
}

// both slab tests of a fetched node, then the reference's choice (extensionRayCast.hlsl:132-159)
template <bool OVF>
__device__ __forceinline__ int inner_compute(const vec4f a, const vec4f b, const vec4f c, const vec2i d, f3 o, f3 invdir, DefStack<OVF>& stk, DevStats* dst)
{
    const float leftHit = ray_box(a.x, a.y, a.z, a.w, b.x, b.y, o, invdir);
    const float rightHit = ray_box(b.z, b.w, c.x, c.y, c.z, c.w, o, invdir);
    const bool l = leftHit > 0.0f, r = rightHit > 0.0f;
    const bool swap = leftHit > rightHit;            // extensionRayCast.hlsl:136: nearer child first, the other one deferred
    if (l && r) { stk.push(swap ? d.x : d.y, dst); return swap ? d.y : d.x; }
    if (l | r) return l ? d.x : d.y;
    return stk.pop();
}

// inner_compute with the choice written as selects around the two stack operations (same tests, same order of visits)
template <bool OVF>
__device__ __forceinline__ int inner_compute_flat(const vec4f a, const vec4f b, const vec4f c, const vec2i d, f3 o, f3 invdir, DefStack<OVF>& stk, DevStats* dst)
{
    const float leftHit = ray_box(a.x, a.y, a.z, a.w, b.x, b.y, o, invdir);
    const float rightHit = ray_box(b.z, b.w, c.x, c.y, c.z, c.w, o, invdir);
    const bool l = leftHit > 0.0f, r = rightHit > 0.0f;
    const bool swap = leftHit > rightHit;            // extensionRayCast.hlsl:136: nearer child first, the other one deferred
    const bool both = l && r;
    const int nearChild = (both && swap) || (!l) ? d.y : d.x;   // both: nearer first; one: the hit one
    const int farChild = swap ? d.x : d.y;
    if (both) stk.push(farChild, dst);
    int next = nearChild;
    if (!(l | r)) next = stk.pop();
    return next;
}

template <bool OVF, bool TOP>
__device__ __forceinline__ int inner_step_d(const TravScene& ts, const float4* s_top, int cur, f3 o, f3 invdir, DefStack<OVF>& stk, DevStats* dst)
{
    vec4f a, b, c; vec2i d;
    load_node<TOP>(ts, s_top, top_count<OVF>(ts), cur, a, b, c, d); // the hottest nodes of the tree live in LDS (68 % of all inner-node visits on the bench scene)
    return inner_compute<OVF>(a, b, c, d, o, invdir, stk, dst);
}

// OPT-IN inner step with distance pruning (GMUPT_EXTEND_PRUNE=1 / GMUPT_SHADOW_PRUNE=1; both default to 0).
// In exact arithmetic a child box that the ray ENTERS beyond `limitT` (the current closest hit, or the light for a shadow ray; both
// with a relative margin) cannot hold a triangle test that changes the result: every accepted hit point lies inside some leaf box of
// its triangle (clipped boxes of spatial splits included), the visit order of the remaining nodes is unchanged, and the strict
// `t < distance` rule makes ties irrelevant.  In binary32 the reference's own Moeller-Trumbore test is noisy for rays within ~1e-7
// rad of a large triangle's plane (|det| just above the 1e-8 cut-off is rounding noise), and such a test can return a `t` far from
// the geometry -- the un-pruned reference then "finds" a hit that a pruned walk never tests.  Measured on the bench scene: bit-
// identical path state and framebuffer over 3000 full-size iterations (6.3 G rays, tools/prune_check.py) with both prunings on,
// k_extend 0.93 -> 0.78 ms and k_shadow 0.60 -> 0.49 ms; but it is not provable, so the default keeps the reference's
// no-pruning rule (quirk Q14) and parity claims are made for the default only.
template <bool OVF, bool TOP>
__device__ __forceinline__ int inner_step_pruned(const TravScene& ts, const float4* s_top, int cur, f3 o, f3 invdir, float limitT, DefStack<OVF>& stk, DevStats* dst)
{
    vec4f a, b, c; vec2i d;
    load_node<TOP>(ts, s_top, top_count<OVF>(ts), cur, a, b, c, d);
    float le, re;
    const float leftHit = ray_box_entry(a.x, a.y, a.z, a.w, b.x, b.y, o, invdir, le);
    const float rightHit = ray_box_entry(b.z, b.w, c.x, c.y, c.z, c.w, o, invdir, re);
    const bool l = leftHit > 0.0f && le <= limitT, r = rightHit > 0.0f && re <= limitT;
    const bool swap = leftHit > rightHit;
    if (l && r) { stk.push(swap ? d.x : d.y, dst); return swap ? d.y : d.x; }
    if (l | r) return l ? d.x : d.y;
    return stk.pop();
}

// LDS of one workgroup of the deferred-leaf kernels: traversal stacks, leaf FIFOs, top of the tree
#define GMUPT_DEF_LDS(TOP) \
    __shared__ int s_stack[kDefLdsStack<OVF> * kDefBlock]; \
    __shared__ int s_fifo[kFifo * kDefBlock]; \
    __shared__ float4 s_top[TOP ? kDefLdsTop<OVF> * 4 : 4]; \
    if (TOP) { \
        const float4* src = reinterpret_cast<const float4*>(p.trav.nodes); \
        for (uint32_t k = threadIdx.x; k < top_count<OVF>(p.trav) * 4u; k += kDefBlock) s_top[k] = src[k]; \
        __syncthreads(); \
    }

// the end of an extension ray: hit record of the closest triangle, then the light spheres (extensionRayCast.hlsl:168-194,218-232)
__device__ __forceinline__ void finish_extension_ray(const RenderParams& p, uint32_t index, f3 o, f3 d, float distance, float hu, float hv, int hitRef)
{
    if (distance < kFltMax) {
        st3(p, F_SP_X, index, o + d * distance);
        st3(p, F_BARY_X, index, mk3(1.0f - hu - hv, hu, hv));
        const int4 T = *reinterpret_cast<const int4*>(&p.scene.tris[hitRef]);
        stu(p, F_TRI_0, index, (uint32_t)T.x); stu(p, F_TRI_1, index, (uint32_t)T.y);
        stu(p, F_TRI_2, index, (uint32_t)T.z); stu(p, F_TRI_MAT, index, (uint32_t)T.w);
    }
    uint32_t lightIndex = 0;
    const uint32_t lc = p.cam.lightCount < GMUPT_MAX_LIGHTS ? p.cam.lightCount : GMUPT_MAX_LIGHTS;
    for (uint32_t li = 0; li < lc; li++) {
        const gmupt_light L = p.scene.lights[li];
        const f3 position = mk3(L.position[0], L.position[1], L.position[2]) - o;
        const float radius2 = L.radius * L.radius;
        const float tca = dot3(position, d);
        const float d2 = dot3(position, position) - tca * tca;
        if (d2 > radius2) continue;
        const float thc = dsqrt(radius2 - d2);
        float t0 = tca - thc;
        const float t1 = tca + thc;
        if (t0 < 0.0f) t0 = t1;
        if (t0 > 0.0f && t0 < distance) { distance = t0; lightIndex = li + 1; }
    }
    stu(p, F_IS_EMITTER, index, lightIndex);
    stf(p, F_HIT_DIST, index, distance);
}

// shadowRayCast.hlsl:144-148: QC[0..3] = (0, QC1 + QC0, 0, 0).  No ray-cast kernel reads those words (they read QC[6] and QC[7]).
__device__ __forceinline__ void shadow_counter_epilogue(const RenderParams& p)
{
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        const uint32_t q0 = p.qc[QC_NEWPATH], q1 = p.qc[QC_LASTPATHCNT];
        p.qc[QC_NEWPATH] = 0; p.qc[QC_LASTPATHCNT] = q0 + q1; p.qc[QC_MATUE4] = 0; p.qc[QC_MATGLASS] = 0;
    }
}

// ---- host launchers
#define GMUPT_DEF_LAUNCH(KERNEL, TOP, REPS, BURST) \
    do { if (ovf) { if (stats) hipLaunchKernelGGL((KERNEL<true, true, TOP, REPS, BURST>), dim3(pb), dim3(kDefBlock), 0, s, p); \
                    else hipLaunchKernelGGL((KERNEL<false, true, TOP, REPS, BURST>), dim3(pb), dim3(kDefBlock), 0, s, p); } \
         else { if (stats) hipLaunchKernelGGL((KERNEL<true, false, TOP, REPS, BURST>), dim3(pb), dim3(kDefBlock), 0, s, p); \
                else hipLaunchKernelGGL((KERNEL<false, false, TOP, REPS, BURST>), dim3(pb), dim3(kDefBlock), 0, s, p); } } while (0)

} // namespace gmupt
