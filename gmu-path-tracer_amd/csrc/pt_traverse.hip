// Ray-cast kernels for gfx950 (MI355X): extension (closest hit + light spheres) and shadow (any hit) -- the shipped versions.
// The rungs that led here (reference-layout, packed static, persistent while-while, interleaved, top-of-tree, cooperative LDS-DMA)
// are kept for A/B timing in pt_traverse_variants.hip (GMUPT_TRAVERSAL=ref|static|whilewhile|ififN|top|coop).
//
// What must not change (it decides results): the slab test arithmetic and its "hit iff result > 0" rule with no pruning
// against the current closest hit (extensionRayCast.hlsl:79-94,132-159), the near-child-first visit order (closest-hit ties
// are resolved by visit order, `t < distance` strict, :64-74), the Moeller-Trumbore operation order (:38-77), and the
// shadow acceptance rule t in (1e-8, 1e8), |d t| < lightDistance (shadowRayCast.hlsl:16-47,88-91).
// What is free: memory layout, loop structure, scheduling of the triangle tests, and -- for the any-hit shadow ray -- the visit order.
#include "pt_traverse_common.hpp"

namespace gmupt {

void launch_extend_variant(const RenderParams& p, uint32_t blocks, bool stats, int mode, hipStream_t s);
void launch_shadow_variant(const RenderParams& p, uint32_t blocks, bool stats, int mode, hipStream_t s);
uint32_t variant_overflow_entries();

#ifndef GMUPT_DEF_STACK
#define GMUPT_DEF_STACK 24
#endif
#ifndef GMUPT_DEF_FIFO
#define GMUPT_DEF_FIFO 8
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
#define GMUPT_DEF_BLOCK 512
#endif
constexpr int kDefBlock = GMUPT_DEF_BLOCK;    // the waves of a workgroup share one LDS copy of the top of the tree
constexpr int kDefStack = GMUPT_DEF_STACK;     // LDS stack entries per lane incl. the sentinel; trees deeper than kDefStack - 2 use the overflow-checked instantiation
constexpr int kFifo = GMUPT_DEF_FIFO;          // pending leaves per lane

template <bool OVF>
struct DefStack {
    int* lds; int* ovf; uint32_t ovfStride; uint32_t ptr; // lds[0] holds kDone for good: popping an empty stack ends the walk
    __device__ __forceinline__ void reset() { ptr = 1; }
    __device__ __forceinline__ void push(int v, DevStats* st)
    {
        if (!OVF || ptr < kDefStack) lds[ptr * kDefBlock] = v;
        else if (ptr < kMaxStack + 1) ovf[(size_t)(ptr - kDefStack) * ovfStride] = v;
        else st->stackOverflow = 1u;
        ptr++;
    }
    __device__ __forceinline__ int pop()
    {
        --ptr;
        if (!OVF || ptr < kDefStack) return lds[ptr * kDefBlock];
        if (ptr < kMaxStack + 1) return ovf[(size_t)(ptr - kDefStack) * ovfStride];
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
__device__ __forceinline__ void load_node(const TravScene& ts, const float4* s_top, int cur, vec4f& a, vec4f& b, vec4f& c, vec2i& d)
{
    // LDS lanes first: the global lanes then only wait for the (short) LDS reads before their loads may target the same registers,
    // and nothing waits for the global loads before they are used
    const bool inTop = TOP && (uint32_t)cur < ts.topCount;
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
        d = __builtin_bit_cast(vec2i, __builtin_amdgcn_raw_buffer_load_b64(nodes, off + 48, 0, 0));
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
    load_node<TOP>(ts, s_top, cur, a, b, c, d); // the first levels of the tree live in LDS (48 % of all inner-node visits on the bench scene)
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
    load_node<TOP>(ts, s_top, cur, a, b, c, d);
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
    __shared__ int s_stack[kDefStack * kDefBlock]; \
    __shared__ int s_fifo[kFifo * kDefBlock]; \
    __shared__ float4 s_top[TOP ? kTopTreeNodes * 4 : 4]; \
    if (TOP) { \
        const float4* src = reinterpret_cast<const float4*>(p.trav.nodes); \
        for (uint32_t k = threadIdx.x; k < p.trav.topCount * 4u; k += kDefBlock) s_top[k] = src[k]; \
        __syncthreads(); \
    }

template <bool STATS, bool OVF, bool TOP, int REPS, int BURST>
__device__ __forceinline__ void extend_body_d(const RenderParams& p, int* s_stack, int* s_fifo, const float4* s_top)
{
    constexpr int WORK_COUNTER = 0;
    const uint32_t gtid = blockIdx.x * kDefBlock + threadIdx.x;
    DefStack<OVF> stk; stk.lds = s_stack + threadIdx.x; stk.ovf = p.ovfStack + gtid; stk.ovfStride = p.ovfStride; stk.ptr = 1;
    s_stack[threadIdx.x] = kDone;
    int* fifo = s_fifo + threadIdx.x;
    const float sceneEps = 1.0e-4f * (dabs(p.trav.rootMax[0] - p.trav.rootMin[0]) + dabs(p.trav.rootMax[1] - p.trav.rootMin[1]) + dabs(p.trav.rootMax[2] - p.trav.rootMin[2]));
    TravCount tc = { 0, 0, 0 }; uint32_t rays = 0, wIn = 0, wTr = 0;
    const TravScene& ts = p.trav;
    const uint32_t count = p.qc[QC_EXT_COUNT];
    const uint32_t* qExt = p.queues + (size_t)Q_EXT_RAY * p.P;
    // work distribution: a persistent grid; every wave takes chunks of p.raysPerWave queue entries from a device counter
    // (one atomic per chunk; the counter is zeroed by k_material) -- no wave waits for another one
    uint32_t next = 0, end = 0;
    bool drained = false;

    bool haveRay = false;
    uint32_t index = 0;
    f3 o = mk3(0, 0, 0), d = mk3(0, 0, 1), invdir = mk3(0, 0, 0);
    float distance = kFltMax, hu = 0.0f, hv = 0.0f;
    int hitRef = -1;
    int cur = kDone;          // walk state: >= 0 inner node, kDone finished walking, otherwise a leaf waiting for a FIFO slot
    uint32_t qHead = 0, qCount = 0;
    int ti = -1;              // next triangle record of the leaf being tested, -1: none

    for (;;) {
        const bool idle = (cur == kDone) && (qCount == 0) && (ti < 0);
        const unsigned long long idleMask = __ballot(idle);
        const int nIdle = __popcll(idleMask);
        if (nIdle == 64 || (nIdle >= (int)p.tuneRefill && !drained)) { // wave-uniform
            if (next >= end && !drained) {
                uint32_t base = 0;
                if ((threadIdx.x & 63) == 0) base = atomicAdd(&p.travCounters[WORK_COUNTER], p.raysPerWave);
                base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
                next = base; end = (base + p.raysPerWave < count) ? base + p.raysPerWave : count;
                if (base >= count) { drained = true; next = end = 0; }
            }
            if (idle) {
                if (haveRay) {
                    // finish the ray: extensionRayCast.hlsl:218-232
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
                    haveRay = false;
                }
                const uint32_t my = next + prefix_rank(idleMask);
                if (my < end) {
                    index = qExt[my];
                    if (index != kQueueHole) {
                        haveRay = true;
                        if (STATS) rays++;
                        o = ld3(p, F_RAY_OX, index); d = ld3(p, F_RAY_DX, index);
                        invdir = mk3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
                        distance = kFltMax; hitRef = -1; hu = 0.0f; hv = 0.0f;
                        stk.reset(); qHead = 0; qCount = 0; ti = -1;
                        cur = (ray_box(ts.rootMin[0], ts.rootMin[1], ts.rootMin[2], ts.rootMax[0], ts.rootMax[1], ts.rootMax[2], o, invdir) > 0.0f) ? ts.rootDesc : kDone;
                    }
                }
            }
            if (nIdle == 64 && drained) break; // nothing in flight and the queue is exhausted (wave-uniform)
            next = (next + (uint32_t)nIdle < end) ? next + (uint32_t)nIdle : end;
        }

        // ---- walk: REPS inner steps per lane; a reached leaf is queued and the walk goes on with the popped node
#pragma unroll
        for (int rep = 0; rep < REPS; rep++) {
            if (cur >= 0) {
                if (STATS) { tc.inner++; if (prefix_rank(__ballot(1)) == 0) wIn++; }
                cur = p.extendPrune ? inner_step_pruned<OVF, TOP>(ts, s_top, cur, o, invdir, distance * 1.0001f + sceneEps, stk, p.stats)
                                    : inner_step_d<OVF, TOP>(ts, s_top, cur, o, invdir, stk, p.stats);
            }
            if (cur < 0 && cur != kDone && qCount < (uint32_t)kFifo) {
                if (STATS) tc.leaves++;
                fifo[((qHead + qCount) & (kFifo - 1)) * kDefBlock] = ~cur;
                qCount++;
                cur = stk.pop();
            }
        }

        // ---- triangle burst when enough lanes have leaves pending, or when nobody can walk any further
        const bool pending = (qCount > 0) || (ti >= 0);
        const int nPending = __popcll(__ballot(pending));
        const int nWalking = __popcll(__ballot(cur >= 0));
        if (nPending >= (int)p.tuneTriThresh || (nWalking == 0 && nPending > 0)) { // wave-uniform
#pragma unroll
            for (int k = 0; k < BURST; k++) {
                if (ti < 0 && qCount > 0) { ti = fifo[(qHead & (kFifo - 1)) * kDefBlock]; qHead++; qCount--; }
                if (ti >= 0) {
                    if (STATS) { tc.tris++; if (prefix_rank(__ballot(1)) == 0) wTr++; }
                    float t = 0.0f, u = 0.0f, v = 0.0f; bool last;
                    if (tri_test(ts.tris, ti, o, d, t, u, v, last)) {
                        if (t >= 0.0f && t < distance) { distance = t; hitRef = ti; hu = u; hv = v; } // extensionRayCast.hlsl:64-74
                    }
                    ti = last ? -1 : ti + 1;
                }
            }
        }
    }
    if (STATS) { flush_counts(p.stats, tc, rays, true); flush_wave_iters(p.stats, wIn, wTr, true); }
}

template <bool STATS, bool OVF, bool TOP, int REPS, int BURST>
__global__ __launch_bounds__(kDefBlock) void k_extend_d(RenderParams p)
{
    GMUPT_DEF_LDS(TOP)
    extend_body_d<STATS, OVF, TOP, REPS, BURST>(p, s_stack, s_fifo, s_top);
}

// shadowRayCast.hlsl:144-148: QC[0..3] = (0, QC1 + QC0, 0, 0).  No ray-cast kernel reads those words (they read QC[6] and QC[7]).
__device__ __forceinline__ void shadow_counter_epilogue(const RenderParams& p)
{
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        const uint32_t q0 = p.qc[QC_NEWPATH], q1 = p.qc[QC_LASTPATHCNT];
        p.qc[QC_NEWPATH] = 0; p.qc[QC_LASTPATHCNT] = q0 + q1; p.qc[QC_MATUE4] = 0; p.qc[QC_MATGLASS] = 0;
    }
}

template <bool STATS, bool OVF, bool TOP, int REPS, int BURST>
__device__ __forceinline__ void shadow_body_d(const RenderParams& p, int* s_stack, int* s_fifo, const float4* s_top)
{
    constexpr int WORK_COUNTER = 1;
    const uint32_t gtid = blockIdx.x * kDefBlock + threadIdx.x;
    const uint32_t count = p.qc[QC_SHADOWRAY];                               // shadowRayCast.hlsl:151
    DefStack<OVF> stk; stk.lds = s_stack + threadIdx.x; stk.ovf = p.ovfStack + gtid; stk.ovfStride = p.ovfStride; stk.ptr = 1;
    s_stack[threadIdx.x] = kDone;
    int* fifo = s_fifo + threadIdx.x;
    const float sceneEps = 1.0e-4f * (dabs(p.trav.rootMax[0] - p.trav.rootMin[0]) + dabs(p.trav.rootMax[1] - p.trav.rootMin[1]) + dabs(p.trav.rootMax[2] - p.trav.rootMin[2]));
    TravCount tc = { 0, 0, 0 }; uint32_t rays = 0, wIn = 0, wTr = 0;
    const TravScene& ts = p.trav;
    const uint32_t* qSh = p.queues + (size_t)Q_SHADOW_RAY * p.P;
    uint32_t next = 0, end = 0;
    bool drained = false;

    bool haveRay = false, occluded = false;
    uint32_t index = 0;
    f3 o = mk3(0, 0, 0), d = mk3(0, 0, 1), invdir = mk3(0, 0, 0);
    float lightDistance = 0.0f, limitT = 0.0f;
    int cur = kDone;
    uint32_t qHead = 0, qCount = 0;
    int ti = -1;

    for (;;) {
        const bool idle = (cur == kDone) && (qCount == 0) && (ti < 0);
        const unsigned long long idleMask = __ballot(idle);
        const int nIdle = __popcll(idleMask);
        if (nIdle == 64 || (nIdle >= (int)p.tuneRefill && !drained)) {
            if (next >= end && !drained) {
                uint32_t base = 0;
                if ((threadIdx.x & 63) == 0) base = atomicAdd(&p.travCounters[WORK_COUNTER], p.raysPerWave);
                base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
                next = base; end = (base + p.raysPerWave < count) ? base + p.raysPerWave : count;
                if (base >= count) { drained = true; next = end = 0; }
            }
            if (idle) {
                if (haveRay) { stu(p, F_IN_SHADOW, index, occluded ? 1u : 0u); haveRay = false; } // :167
                const uint32_t my = next + prefix_rank(idleMask);
                if (my < end) {
                    index = qSh[my];                                         // :159
                    haveRay = true; occluded = false;
                    if (STATS) rays++;
                    o = ld3(p, F_SH_OX, index); d = ld3(p, F_SH_DX, index);  // :162-163
                    lightDistance = ldf(p, F_LIGHT_DIST, index);             // :164
                    invdir = mk3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
                    limitT = (lightDistance / length3(d)) * 1.0001f + sceneEps; // parametric distance of the light, with margin
                    stk.reset(); qHead = 0; qCount = 0; ti = -1;
                    cur = (ray_box(ts.rootMin[0], ts.rootMin[1], ts.rootMin[2], ts.rootMax[0], ts.rootMax[1], ts.rootMax[2], o, invdir) > 0.0f) ? ts.rootDesc : kDone;
                }
            }
            if (nIdle == 64 && drained) break;
            next = (next + (uint32_t)nIdle < end) ? next + (uint32_t)nIdle : end;
        }
#pragma unroll
        for (int rep = 0; rep < REPS; rep++) {
            if (cur >= 0) {
                if (STATS) { tc.inner++; if (prefix_rank(__ballot(1)) == 0) wIn++; }
                cur = p.shadowPrune ? inner_step_pruned<OVF, TOP>(ts, s_top, cur, o, invdir, limitT, stk, p.stats) : inner_step_d<OVF, TOP>(ts, s_top, cur, o, invdir, stk, p.stats);
            }
            if (cur < 0 && cur != kDone && qCount < (uint32_t)kFifo) {
                if (STATS) tc.leaves++;
                fifo[((qHead + qCount) & (kFifo - 1)) * kDefBlock] = ~cur;
                qCount++;
                cur = stk.pop();
            }
        }
        const bool pending = (qCount > 0) || (ti >= 0);
        const int nPending = __popcll(__ballot(pending));
        const int nWalking = __popcll(__ballot(cur >= 0));
        if (nPending >= (int)p.tuneTriThresh || (nWalking == 0 && nPending > 0)) {
#pragma unroll
            for (int k = 0; k < BURST; k++) {
                if (ti < 0 && qCount > 0) { ti = fifo[(qHead & (kFifo - 1)) * kDefBlock]; qHead++; qCount--; }
                if (ti >= 0) {
                    if (STATS) { tc.tris++; if (prefix_rank(__ballot(1)) == 0) wTr++; }
                    float t = 0.0f, u = 0.0f, v = 0.0f; bool last;
                    if (tri_test(ts.tris, ti, o, d, t, u, v, last)) {
                        // shadowRayCast.hlsl:41-45,89: t in (1e-8, 1e8) and |d t| < lightDistance => occluded: the ray is decided
                        if (t > kEpsilon && t < 1.0f / kEpsilon && length3(d * t) < lightDistance) { occluded = true; last = true; qCount = 0; cur = kDone; }
                    }
                    ti = last ? -1 : ti + 1;
                }
            }
        }
    }
    if (STATS) { flush_counts(p.stats, tc, rays, false); flush_wave_iters(p.stats, wIn, wTr, false); }
}

template <bool STATS, bool OVF, bool TOP, int REPS, int BURST>
__global__ __launch_bounds__(kDefBlock) void k_shadow_d(RenderParams p)
{
    GMUPT_DEF_LDS(TOP)
    shadow_counter_epilogue(p);
    shadow_body_d<STATS, OVF, TOP, REPS, BURST>(p, s_stack, s_fifo, s_top);
}

// Both ray casts in one persistent launch: a wave that finds the extension queue exhausted goes on with shadow-ray chunks instead of
// leaving the chip, so the tail of the extension cast (a wave works for ~0.4 ms on one 128-ray chunk, the last waves run alone) is
// filled with shadow rays.  The two casts touch disjoint path-state fields (hit record / inShadow), so their order is free.
template <bool STATS, bool OVF, bool TOP, int REPS, int BURST>
__global__ __launch_bounds__(kDefBlock) void k_cast_d(RenderParams p)
{
    GMUPT_DEF_LDS(TOP)
    shadow_counter_epilogue(p);
    extend_body_d<STATS, OVF, TOP, REPS, BURST>(p, s_stack, s_fifo, s_top);
    shadow_body_d<STATS, OVF, TOP, REPS, BURST>(p, s_stack, s_fifo, s_top);
}

// Both ray casts in one persistent launch AND in one loop: a lane carries either an extension ray or a shadow ray (`kind`), so the
// lanes that run out of extension rays take shadow rays while their neighbours are still walking -- no wave waits for its longest
// extension ray before it starts on the shadow queue.  The walk is identical for both kinds; only the hit rule of the triangle burst
// and the finish differ, and those run divergent only in the few iterations in which a wave holds both kinds.
// `phase` (wave-uniform): 0 extension queue, 1 shadow queue, 2 both exhausted.  The opt-in prunings are not offered here.
template <bool STATS, bool OVF, bool TOP, int REPS, int BURST>
__global__ __launch_bounds__(kDefBlock) void k_cast_m(RenderParams p)
{
    GMUPT_DEF_LDS(TOP)
    shadow_counter_epilogue(p);
    const uint32_t gtid = blockIdx.x * kDefBlock + threadIdx.x;
    DefStack<OVF> stk; stk.lds = s_stack + threadIdx.x; stk.ovf = p.ovfStack + gtid; stk.ovfStride = p.ovfStride; stk.ptr = 1;
    s_stack[threadIdx.x] = kDone;
    int* fifo = s_fifo + threadIdx.x;
    TravCount tcE = { 0, 0, 0 }, tcS = { 0, 0, 0 }; uint32_t raysE = 0, raysS = 0, wInE = 0, wTrE = 0, wInS = 0, wTrS = 0;
    const TravScene& ts = p.trav;
    const uint32_t countExt = p.qc[QC_EXT_COUNT], countSh = p.qc[QC_SHADOWRAY];  // extensionRayCast.hlsl:205, shadowRayCast.hlsl:151
    const uint32_t* qExt = p.queues + (size_t)Q_EXT_RAY * p.P;
    const uint32_t* qSh = p.queues + (size_t)Q_SHADOW_RAY * p.P;
    uint32_t next = 0, end = 0;
    int phase = 0;
    const unsigned long long tStart = STATS ? wall_clock64() : 0ull;
    uint32_t rayInner = 0, census0 = 0, census1 = 0, census2 = 0, census3 = 0;

    bool haveRay = false;
    int kind = 0;                 // 0: extension ray, 1: shadow ray
    uint32_t index = 0;
    f3 o = mk3(0, 0, 0), d = mk3(0, 0, 1), invdir = mk3(0, 0, 0);
    float distance = kFltMax;     // extension: closest hit so far; shadow: distance of the light
    float hu = 0.0f, hv = 0.0f;
    int hitRef = -1;              // extension: triangle record of the closest hit; shadow: >= 0 when occluded
    int cur = kDone;
    uint32_t qHead = 0, qCount = 0;
    int ti = -1;

    for (;;) {
        const bool idle = (cur == kDone) && (qCount == 0) && (ti < 0);
        const unsigned long long idleMask = __ballot(idle);
        const int nIdle = __popcll(idleMask);
        if (nIdle == 64 || (nIdle >= (int)p.tuneRefill && phase < 2)) { // wave-uniform
            while (next >= end && phase < 2) { // next chunk of the current queue, or the first one of the next queue
                uint32_t base = 0;
                if ((threadIdx.x & 63) == 0) base = atomicAdd(&p.travCounters[phase], p.raysPerWave);
                base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
                const uint32_t count = phase == 0 ? countExt : countSh;
                if (base < count) { next = base; end = (base + p.raysPerWave < count) ? base + p.raysPerWave : count; }
                else { phase++; next = end = 0; }
            }
            if (idle) {
                if (haveRay) {
                    if (STATS && kind == 0) { atomicAdd(&p.stats->rayInnerHist[rayInner / 16u < 31u ? rayInner / 16u : 31u], 1ull); rayInner = 0; }
                    if (kind == 0) {
                        // finish the extension ray: extensionRayCast.hlsl:218-232
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
                    } else {
                        stu(p, F_IN_SHADOW, index, hitRef >= 0 ? 1u : 0u);   // shadowRayCast.hlsl:167
                    }
                    haveRay = false;
                }
                const uint32_t my = next + prefix_rank(idleMask);
                if (my < end) {
                    if (phase == 0) { // wave-uniform
                        index = qExt[my];                                    // extensionRayCast.hlsl:210
                        if (index != kQueueHole) {
                            haveRay = true; kind = 0;
                            if (STATS) raysE++;
                            o = ld3(p, F_RAY_OX, index); d = ld3(p, F_RAY_DX, index); // :213-214
                            distance = kFltMax;
                        }
                    } else {
                        index = qSh[my];                                     // shadowRayCast.hlsl:159
                        haveRay = true; kind = 1;
                        if (STATS) raysS++;
                        o = ld3(p, F_SH_OX, index); d = ld3(p, F_SH_DX, index); // :162-163
                        distance = ldf(p, F_LIGHT_DIST, index);              // :164
                    }
                    if (haveRay) {
                        invdir = mk3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
                        hitRef = -1; hu = 0.0f; hv = 0.0f;
                        stk.reset(); qHead = 0; qCount = 0; ti = -1;
                        cur = (ray_box(ts.rootMin[0], ts.rootMin[1], ts.rootMin[2], ts.rootMax[0], ts.rootMax[1], ts.rootMax[2], o, invdir) > 0.0f) ? ts.rootDesc : kDone;
                    }
                }
            }
            if (nIdle == 64 && phase == 2) break; // nothing in flight and both queues are exhausted (wave-uniform)
            next = (next + (uint32_t)nIdle < end) ? next + (uint32_t)nIdle : end;
        }

        if (STATS) { // lane census: where do the 64 lanes of a wave spend the loop iterations?
            const bool pendingNow = (qCount > 0) || (ti >= 0);
            census0 += __popcll(__ballot(cur == kDone && !pendingNow)); census1 += __popcll(__ballot(cur >= 0));
            census2 += __popcll(__ballot(cur < 0 && cur != kDone)); census3 += __popcll(__ballot(cur == kDone && pendingNow));
        }
        // ---- walk: REPS inner steps per lane; a reached leaf is queued and the walk goes on with the popped node
#pragma unroll
        for (int rep = 0; rep < REPS; rep++) {
            if (cur >= 0) {
                if (STATS) { if (kind == 0) { tcE.inner++; rayInner++; } else tcS.inner++; if (prefix_rank(__ballot(1)) == 0) { if (phase == 0) wInE++; else wInS++; } }
                cur = inner_step_d<OVF, TOP>(ts, s_top, cur, o, invdir, stk, p.stats);
            }
            if (cur < 0 && cur != kDone && qCount < (uint32_t)kFifo) {
                if (STATS) { if (kind == 0) tcE.leaves++; else tcS.leaves++; }
                fifo[((qHead + qCount) & (kFifo - 1)) * kDefBlock] = ~cur;
                qCount++;
                cur = stk.pop();
            }
        }

        // ---- triangle burst when enough lanes have leaves pending, or when nobody can walk any further
        const bool pending = (qCount > 0) || (ti >= 0);
        const int nPending = __popcll(__ballot(pending));
        const int nWalking = __popcll(__ballot(cur >= 0));
        if (nPending >= (int)p.tuneTriThresh || (nWalking == 0 && nPending > 0)) { // wave-uniform
#pragma unroll
            for (int k = 0; k < BURST; k++) {
                if (ti < 0 && qCount > 0) { ti = fifo[(qHead & (kFifo - 1)) * kDefBlock]; qHead++; qCount--; }
                if (ti >= 0) {
                    if (STATS) { if (kind == 0) tcE.tris++; else tcS.tris++; if (prefix_rank(__ballot(1)) == 0) { if (phase == 0) wTrE++; else wTrS++; } }
                    float t = 0.0f, u = 0.0f, v = 0.0f; bool last;
                    if (tri_test(ts.tris, ti, o, d, t, u, v, last)) {
                        if (kind == 0) {
                            if (t >= 0.0f && t < distance) { distance = t; hitRef = ti; hu = u; hv = v; } // extensionRayCast.hlsl:64-74
                        } else {
                            // shadowRayCast.hlsl:41-45,89: t in (1e-8, 1e8) and |d t| < lightDistance => occluded: the ray is decided
                            if (t > kEpsilon && t < 1.0f / kEpsilon && length3(d * t) < distance) { hitRef = ti; last = true; qCount = 0; cur = kDone; }
                        }
                    }
                    ti = last ? -1 : ti + 1;
                }
            }
        }
    }
    if (STATS) { flush_counts(p.stats, tcE, raysE, true); flush_wave_iters(p.stats, wInE, wTrE, true);
                 flush_counts(p.stats, tcS, raysS, false); flush_wave_iters(p.stats, wInS, wTrS, false);
                 if ((threadIdx.x & 63) == 0) {
                     const unsigned long long life = wall_clock64() - tStart;
                     atomicAdd(&p.stats->castWaves, 1ull); atomicAdd(&p.stats->castWaveClocks, life); atomicMax(&p.stats->castWaveClocksMax, life);
                     atomicAdd(&p.stats->castWaveEndHist[life / 5000ull < 31ull ? life / 5000ull : 31ull], 1ull);
                     atomicAdd(&p.stats->laneCensus[0], (unsigned long long)census0); atomicAdd(&p.stats->laneCensus[1], (unsigned long long)census1);
                     atomicAdd(&p.stats->laneCensus[2], (unsigned long long)census2); atomicAdd(&p.stats->laneCensus[3], (unsigned long long)census3);
                 } }
}

// k_cast_m with the two fetches of a loop step issued together: when the wave is in a triangle burst, every step first issues the node
// fetch of the walking lanes AND the triangle fetch of the lanes with a pending leaf, then does the slab tests and the triangle test.
// A burst therefore costs no memory round trips of its own (REPS per loop iteration instead of REPS + BURST); what is tested, and in which
// order per ray, is unchanged: leaves leave the per-lane FIFO in visit order.  BURST is unused here (one triangle record per step).
template <bool STATS, bool OVF, bool TOP, int REPS, int BURST>
__global__ __launch_bounds__(kDefBlock) void k_cast_f(RenderParams p)
{
    GMUPT_DEF_LDS(TOP)
    shadow_counter_epilogue(p);
    const uint32_t gtid = blockIdx.x * kDefBlock + threadIdx.x;
    DefStack<OVF> stk; stk.lds = s_stack + threadIdx.x; stk.ovf = p.ovfStack + gtid; stk.ovfStride = p.ovfStride; stk.ptr = 1;
    s_stack[threadIdx.x] = kDone;
    int* fifo = s_fifo + threadIdx.x;
    TravCount tcE = { 0, 0, 0 }, tcS = { 0, 0, 0 }; uint32_t raysE = 0, raysS = 0, wInE = 0, wTrE = 0, wInS = 0, wTrS = 0;
    const TravScene& ts = p.trav;
    const uint32_t countExt = p.qc[QC_EXT_COUNT], countSh = p.qc[QC_SHADOWRAY];  // extensionRayCast.hlsl:205, shadowRayCast.hlsl:151
    const uint32_t* qExt = p.queues + (size_t)Q_EXT_RAY * p.P;
    const uint32_t* qSh = p.queues + (size_t)Q_SHADOW_RAY * p.P;
    const __amdgpu_buffer_rsrc_t rNodes = make_rsrc(ts.nodes, ts.triBase * 64u);              // ts.triBase = number of packed nodes
    const __amdgpu_buffer_rsrc_t rTris = make_rsrc(ts.tris, (p.scene.numTris + 1u) * 48u);    // + the sentinel record
    uint32_t next = 0, end = 0;
    uint32_t chunkBase = 0, qe0 = kQueueHole, qe1 = kQueueHole;  // the current chunk of queue entries, lane l holds entries l and 64 + l
    int phase = 0;
    const unsigned long long tStart = STATS ? wall_clock64() : 0ull;
    unsigned long long tDrain = 0ull; uint32_t drainIters = 0, drainBusy = 0;
    uint32_t rayInner = 0, census0 = 0, census1 = 0, census2 = 0, census3 = 0;

    bool haveRay = false;
    int kind = 0;                 // 0: extension ray, 1: shadow ray
    uint32_t index = 0;
    f3 o = mk3(0, 0, 0), d = mk3(0, 0, 1), invdir = mk3(0, 0, 0);
    float distance = kFltMax;     // extension: closest hit so far; shadow: distance of the light
    float hu = 0.0f, hv = 0.0f;
    int hitRef = -1;              // extension: triangle record of the closest hit; shadow: >= 0 when occluded
    int cur = kDone;
    uint32_t qHead = 0, qCount = 0;
    int ti = -1;

    for (;;) {
        const bool idle = (cur == kDone) && (qCount == 0) && (ti < 0);
        const unsigned long long idleMask = __ballot(idle);
        const int nIdle = __popcll(idleMask);
        if (nIdle == 64 || (nIdle >= (int)p.tuneRefill && phase < 2)) { // wave-uniform
            while (next >= end && phase < 2) { // next chunk of the current queue, or the first one of the next queue
                uint32_t base = 0;
                if ((threadIdx.x & 63) == 0) base = atomicAdd(&p.travCounters[phase], p.raysPerWave);
                base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
                const uint32_t count = phase == 0 ? countExt : countSh;
                if (base < count) {
                    next = base; end = (base + p.raysPerWave < count) ? base + p.raysPerWave : count;
                    // the whole chunk of queue entries goes into two registers per lane (two coalesced loads): a refill then takes
                    // its entry from a neighbour's register instead of starting with a dependent queue read
                    chunkBase = base;
                    const uint32_t* q = phase == 0 ? qExt : qSh;
                    const uint32_t lane = threadIdx.x & 63u;
                    qe0 = (base + lane < end) ? q[base + lane] : kQueueHole;
                    qe1 = (base + 64u + lane < end) ? q[base + 64u + lane] : kQueueHole;
                }
                else { phase++; next = end = 0; }
            }
#ifndef GMUPT_KNOCKOUT
#define GMUPT_KNOCKOUT 0   // timing experiments only (tools/knockout.py; results are wrong): 1 no result stores, 3 no triangle tests
#endif
            // queue entry of every idle lane, out of the chunk registers (all lanes execute the two shuffles: a lane that sits out
            // would not lend its registers)
            const uint32_t my = next + prefix_rank(idleMask);
            const uint32_t entry = idle ? ((my - chunkBase) & 127u) : 0u;
            const uint32_t entryLo = (uint32_t)__shfl((int)qe0, (int)(entry & 63u)), entryHi = (uint32_t)__shfl((int)qe1, (int)(entry & 63u));
            if (idle) {
                // the next ray first: its loads are in flight while the finished ray is written back
                const bool take = my < end;
                const uint32_t newIndex = take ? (entry < 64u ? entryLo : entryHi) : kQueueHole;   // extensionRayCast.hlsl:210 / shadowRayCast.hlsl:159
                const bool newRay = take && (phase != 0 || newIndex != kQueueHole); // holes only exist in the extension queue
                f3 newO = mk3(0, 0, 0), newD = mk3(0, 0, 1); float newDist = kFltMax;
                if (newRay) {
                    if (phase == 0) { newO = ld3(p, F_RAY_OX, newIndex); newD = ld3(p, F_RAY_DX, newIndex); }                 // :213-214
                    else { newO = ld3(p, F_SH_OX, newIndex); newD = ld3(p, F_SH_DX, newIndex); newDist = ldf(p, F_LIGHT_DIST, newIndex); } // :162-164
                }
                if (haveRay && GMUPT_KNOCKOUT == 1) haveRay = false;
                if (haveRay) {
                    if (STATS && kind == 0) { atomicAdd(&p.stats->rayInnerHist[rayInner / 16u < 31u ? rayInner / 16u : 31u], 1ull); rayInner = 0; }
                    if (kind == 0) {
                        // finish the extension ray: extensionRayCast.hlsl:218-232
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
                    } else {
                        stu(p, F_IN_SHADOW, index, hitRef >= 0 ? 1u : 0u);   // shadowRayCast.hlsl:167
                    }
                    haveRay = false;
                }
                if (newRay) {
                    haveRay = true; kind = phase; index = newIndex;            // phase is 0 (extension) or 1 (shadow) here
                    if (STATS) { if (phase == 0) raysE++; else raysS++; }
                    o = newO; d = newD; distance = newDist;
                    invdir = mk3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
                    hitRef = -1; hu = 0.0f; hv = 0.0f;
                    stk.reset(); qHead = 0; qCount = 0; ti = -1;
                    cur = (ray_box(ts.rootMin[0], ts.rootMin[1], ts.rootMin[2], ts.rootMax[0], ts.rootMax[1], ts.rootMax[2], o, invdir) > 0.0f) ? ts.rootDesc : kDone;
                }
            }
            if (nIdle == 64 && phase == 2) break; // nothing in flight and both queues are exhausted (wave-uniform)
            next = (next + (uint32_t)nIdle < end) ? next + (uint32_t)nIdle : end;
        }

        if (STATS) { // lane census: where do the 64 lanes of a wave spend the loop iterations?
            const bool pendingNow = (qCount > 0) || (ti >= 0);
            if (phase == 2) { if (tDrain == 0ull) tDrain = wall_clock64(); drainIters++; drainBusy += __popcll(__ballot(haveRay && !(cur == kDone && !pendingNow))); }
            census0 += __popcll(__ballot(cur == kDone && !pendingNow)); census1 += __popcll(__ballot(cur >= 0));
            census2 += __popcll(__ballot(cur < 0 && cur != kDone)); census3 += __popcll(__ballot(cur == kDone && pendingNow));
        }
        // ---- REPS steps; a burst (wave-uniform, decided once per iteration) adds one triangle test per step to the lanes with leaves pending
        const bool pendingNow = (qCount > 0) || (ti >= 0);
        const int nPending = __popcll(__ballot(pendingNow));
        const int nWalking = __popcll(__ballot(cur >= 0));
        const bool burst = nPending >= (int)p.tuneTriThresh || (nWalking == 0 && nPending > 0); // wave-uniform
#pragma unroll
        for (int rep = 0; rep < REPS; rep++) {
            // fetch phase
            const bool doNode = cur >= 0;
            vec4f na, nb, nc; vec2i nd;                      // defined for the doNode lanes only
            if (doNode) load_node_buf<TOP>(rNodes, s_top, ts.topCount, cur, na, nb, nc, nd);
            if (burst && ti < 0 && qCount > 0) { ti = fifo[(qHead & (kFifo - 1)) * kDefBlock]; qHead++; qCount--; }
            const bool doTri = burst && ti >= 0;
            vec4f r0, r1; vec2f r2;                          // defined for the doTri lanes only
            if (doTri) tri_fetch_buf(rTris, ti, r0, r1, r2);
            // compute phase
            if (doNode) {
                if (STATS) { if (kind == 0) { tcE.inner++; rayInner++; } else tcS.inner++; if (prefix_rank(__ballot(1)) == 0) { if (phase == 0) wInE++; else wInS++; } }
                cur = inner_compute_flat<OVF>(na, nb, nc, nd, o, invdir, stk, p.stats);
            }
            if (GMUPT_KNOCKOUT == 3) { if (cur < 0 && cur != kDone) cur = stk.pop(); }
            else if (cur < 0 && cur != kDone && qCount < (uint32_t)kFifo) {
                if (STATS) { if (kind == 0) tcE.leaves++; else tcS.leaves++; }
                fifo[((qHead + qCount) & (kFifo - 1)) * kDefBlock] = ~cur;
                qCount++;
                cur = stk.pop();
            }
            if (doTri) {
                if (STATS) { if (kind == 0) tcE.tris++; else tcS.tris++; if (prefix_rank(__ballot(1)) == 0) { if (phase == 0) wTrE++; else wTrS++; } }
                float t = 0.0f, u = 0.0f, v = 0.0f; bool last;
                if (tri_compute_flat(make_float4(r0.x, r0.y, r0.z, r0.w), make_float4(r1.x, r1.y, r1.z, r1.w),
                                     make_float4(r2.x, r2.y, 0.0f, 0.0f), o, d, t, u, v, last)) {
                    if (kind == 0) {
                        if (t >= 0.0f && t < distance) { distance = t; hitRef = ti; hu = u; hv = v; } // extensionRayCast.hlsl:64-74
                    } else {
                        // shadowRayCast.hlsl:41-45,89: t in (1e-8, 1e8) and |d t| < lightDistance => occluded: the ray is decided
                        if (t > kEpsilon && t < 1.0f / kEpsilon && length3(d * t) < distance) { hitRef = ti; last = true; qCount = 0; cur = kDone; }
                    }
                }
                ti = last ? -1 : ti + 1;
            }
        }
    }
    if (STATS) { flush_counts(p.stats, tcE, raysE, true); flush_wave_iters(p.stats, wInE, wTrE, true);
                 flush_counts(p.stats, tcS, raysS, false); flush_wave_iters(p.stats, wInS, wTrS, false);
                 if ((threadIdx.x & 63) == 0) {
                     const unsigned long long tEnd = wall_clock64();
                     const unsigned long long life = tEnd - tStart;
                     atomicAdd(&p.stats->castDrainClocks, tDrain ? tEnd - tDrain : 0ull); atomicAdd(&p.stats->castDrainIters, (unsigned long long)drainIters);
                     atomicAdd(&p.stats->castDrainBusyLanes, (unsigned long long)drainBusy);
                     atomicAdd(&p.stats->castWaves, 1ull); atomicAdd(&p.stats->castWaveClocks, life); atomicMax(&p.stats->castWaveClocksMax, life);
                     atomicAdd(&p.stats->castWaveEndHist[life / 5000ull < 31ull ? life / 5000ull : 31ull], 1ull);
                     atomicAdd(&p.stats->laneCensus[0], (unsigned long long)census0); atomicAdd(&p.stats->laneCensus[1], (unsigned long long)census1);
                     atomicAdd(&p.stats->laneCensus[2], (unsigned long long)census2); atomicAdd(&p.stats->laneCensus[3], (unsigned long long)census3);
                 } }
}


// ------------------------------------------------------------------------------------------------ three-slot lane pipeline
// Lane census of the kernels above (collect_stats): 29 % of the lanes have no ray (they wait for the next batched refill, whose
// finish-and-fetch code and two dependent loads stall the whole wave), 14 % have finished walking and wait for their queued leaves.
// Here a lane owns three ray slots: PREFETCHED (queue entry, origin, direction, reciprocal direction and root test are fetched and
// computed ahead, in batches, two loop iterations before they are needed), ACTIVE (walking / testing) and RESULT (a finished ray whose
// stores are issued later, in batches).  A lane that finishes a ray moves it to its result slot and starts its prefetched ray in the
// same iteration with register moves only; nothing in the hot loop waits for a ray fetch.  Per-ray arithmetic and order: unchanged.
constexpr int kFinishBatch = 24;    // issue the result stores when this many lanes hold a finished ray
constexpr int kPrefetchBatch = 8;   // request new queue entries when this many lanes have an empty prefetch slot

template <bool STATS, bool OVF, bool TOP, int REPS, int BURST>
__global__ __launch_bounds__(kDefBlock) void k_extend_e(RenderParams p)
{
    __shared__ int s_stack[kDefStack * kDefBlock];
    __shared__ int s_fifo[kFifo * kDefBlock];
    __shared__ float4 s_top[TOP ? kTopTreeNodes * 4 : 4];
    if (TOP) {
        const float4* src = reinterpret_cast<const float4*>(p.trav.nodes);
        for (uint32_t k = threadIdx.x; k < p.trav.topCount * 4u; k += kDefBlock) s_top[k] = src[k];
        __syncthreads();
    }
    const uint32_t gtid = blockIdx.x * kDefBlock + threadIdx.x;
    DefStack<OVF> stk; stk.lds = s_stack + threadIdx.x; stk.ovf = p.ovfStack + gtid; stk.ovfStride = p.ovfStride; stk.ptr = 1;
    s_stack[threadIdx.x] = kDone;
    int* fifo = s_fifo + threadIdx.x;
    TravCount tc = { 0, 0, 0 }; uint32_t rays = 0, wIn = 0, wTr = 0;
    const TravScene& ts = p.trav;
    const uint32_t count = p.qc[QC_EXT_COUNT];
    const uint32_t* qExt = p.queues + (size_t)Q_EXT_RAY * p.P;
    uint32_t next = 0, end = 0;      // current chunk of the extension queue (wave-uniform)
    bool drained = false;            // the device work counter is exhausted (wave-uniform)

    // ACTIVE slot
    bool haveRay = false;
    uint32_t index = 0;
    f3 o = mk3(0, 0, 0), d = mk3(0, 0, 1), invdir = mk3(0, 0, 0);
    float distance = kFltMax, hu = 0.0f, hv = 0.0f;
    int hitRef = -1, cur = kDone;
    uint32_t qHead = 0, qCount = 0;
    int ti = -1;
    // RESULT slot
    bool resValid = false;
    uint32_t rIndex = 0; f3 rO = mk3(0, 0, 0), rD = mk3(0, 0, 1); float rDist = kFltMax, rHu = 0.0f, rHv = 0.0f; int rHit = -1;
    // PREFETCH slot: 0 empty, 1 queue entry requested, 2 ray requested, 3 ready
    int pf = 0;
    uint32_t pfIndex = 0; f3 pfO = mk3(0, 0, 0), pfD = mk3(0, 0, 1), pfInv = mk3(0, 0, 0); int pfCur = kDone;

    for (;;) {
        // (a) a finished ray moves to the result slot (register moves)
        if (haveRay && cur == kDone && qCount == 0 && ti < 0 && !resValid) {
            resValid = true; rIndex = index; rO = o; rD = d; rDist = distance; rHu = hu; rHv = hv; rHit = hitRef;
            haveRay = false;
        }
        // (b) an empty active slot takes the prefetched ray (register moves)
        if (!haveRay && pf == 3) {
            haveRay = true; index = pfIndex; o = pfO; d = pfD; invdir = pfInv; cur = pfCur;
            distance = kFltMax; hitRef = -1; hu = 0.0f; hv = 0.0f;
            stk.reset(); qHead = 0; qCount = 0; ti = -1;
            pf = 0;
            if (STATS) rays++;
        }
        const unsigned long long busyMask = __ballot(haveRay);
        // (c) finish rays in batches: extensionRayCast.hlsl:218-232
        {
            const int nRes = __popcll(__ballot(resValid));
            const bool blocked = haveRay && cur == kDone && qCount == 0 && ti < 0;    // finished, but its result slot is still occupied
            if (nRes >= kFinishBatch || __ballot(blocked) != 0ull || (busyMask == 0ull && nRes > 0)) { // wave-uniform
                if (resValid) {
                    float dist = rDist;
                    if (dist < kFltMax) {
                        st3(p, F_SP_X, rIndex, rO + rD * dist);
                        st3(p, F_BARY_X, rIndex, mk3(1.0f - rHu - rHv, rHu, rHv));
                        const int4 T = *reinterpret_cast<const int4*>(&p.scene.tris[rHit]);
                        stu(p, F_TRI_0, rIndex, (uint32_t)T.x); stu(p, F_TRI_1, rIndex, (uint32_t)T.y);
                        stu(p, F_TRI_2, rIndex, (uint32_t)T.z); stu(p, F_TRI_MAT, rIndex, (uint32_t)T.w);
                    }
                    uint32_t lightIndex = 0;
                    const uint32_t lc = p.cam.lightCount < GMUPT_MAX_LIGHTS ? p.cam.lightCount : GMUPT_MAX_LIGHTS;
                    for (uint32_t li = 0; li < lc; li++) {
                        const gmupt_light L = p.scene.lights[li];
                        const f3 position = mk3(L.position[0], L.position[1], L.position[2]) - rO;
                        const float radius2 = L.radius * L.radius;
                        const float tca = dot3(position, rD);
                        const float d2 = dot3(position, position) - tca * tca;
                        if (d2 > radius2) continue;
                        const float thc = dsqrt(radius2 - d2);
                        float t0 = tca - thc;
                        const float t1 = tca + thc;
                        if (t0 < 0.0f) t0 = t1;
                        if (t0 > 0.0f && t0 < dist) { dist = t0; lightIndex = li + 1; }
                    }
                    stu(p, F_IS_EMITTER, rIndex, lightIndex);
                    stf(p, F_HIT_DIST, rIndex, dist);
                    resValid = false;
                }
            }
        }
        // (d) prefetch pipeline, youngest stage last so that a slot advances one stage per iteration
        if (pf == 2) { // ray data has arrived: reciprocal direction and root test, once per ray (extensionRayCast.hlsl:81,103-105)
            pfInv = mk3(1.0f / pfD.x, 1.0f / pfD.y, 1.0f / pfD.z);
            pfCur = (ray_box(ts.rootMin[0], ts.rootMin[1], ts.rootMin[2], ts.rootMax[0], ts.rootMax[1], ts.rootMax[2], pfO, pfInv) > 0.0f) ? ts.rootDesc : kDone;
            pf = 3;
        }
        if (pf == 1) { // the queue entry has arrived: request the ray (:213-214), or drop a hole left by a retired slot
            if (pfIndex == kQueueHole) pf = 0;
            else { pfO = ld3(p, F_RAY_OX, pfIndex); pfD = ld3(p, F_RAY_DX, pfIndex); pf = 2; }
        }
        {
            const unsigned long long needMask = __ballot(pf == 0);
            const int nNeed = __popcll(needMask);
            if (!drained && (nNeed >= kPrefetchBatch || (busyMask == 0ull && nNeed > 0))) { // wave-uniform
                if (next >= end) {
                    uint32_t base = 0;
                    if ((threadIdx.x & 63) == 0) base = atomicAdd(&p.travCounters[0], p.raysPerWave);
                    base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
                    next = base; end = (base + p.raysPerWave < count) ? base + p.raysPerWave : count;
                    if (base >= count) { drained = true; next = end = 0; }
                }
                if (pf == 0) {
                    const uint32_t my = next + prefix_rank(needMask);
                    if (my < end) { pfIndex = qExt[my]; pf = 1; }                 // :210
                }
                next = (next + (uint32_t)nNeed < end) ? next + (uint32_t)nNeed : end;
            }
        }
        // (e) done when nothing is in flight anywhere in the wave and the queue is exhausted
        if (drained && busyMask == 0ull && __ballot(resValid || pf != 0) == 0ull) break;

        // (f) walk: REPS inner steps per lane; a reached leaf is queued and the walk goes on with the popped node
#pragma unroll
        for (int rep = 0; rep < REPS; rep++) {
            if (cur >= 0) {
                if (STATS) { tc.inner++; if (prefix_rank(__ballot(1)) == 0) wIn++; }
                cur = inner_step_d<OVF, TOP>(ts, s_top, cur, o, invdir, stk, p.stats);
            }
            if (cur < 0 && cur != kDone && qCount < (uint32_t)kFifo) {
                if (STATS) tc.leaves++;
                fifo[((qHead + qCount) & (kFifo - 1)) * kDefBlock] = ~cur;
                qCount++;
                cur = stk.pop();
            }
        }
        // (g) triangle burst when enough lanes have leaves pending, or when nobody can walk any further
        const bool pending = (qCount > 0) || (ti >= 0);
        const int nPending = __popcll(__ballot(pending));
        const int nWalking = __popcll(__ballot(cur >= 0));
        if (nPending >= (int)p.tuneTriThresh || (nWalking == 0 && nPending > 0)) { // wave-uniform
#pragma unroll
            for (int k = 0; k < BURST; k++) {
                if (ti < 0 && qCount > 0) { ti = fifo[(qHead & (kFifo - 1)) * kDefBlock]; qHead++; qCount--; }
                if (ti >= 0) {
                    if (STATS) { tc.tris++; if (prefix_rank(__ballot(1)) == 0) wTr++; }
                    float t = 0.0f, u = 0.0f, v = 0.0f; bool last;
                    if (tri_test(ts.tris, ti, o, d, t, u, v, last)) {
                        if (t >= 0.0f && t < distance) { distance = t; hitRef = ti; hu = u; hv = v; } // extensionRayCast.hlsl:64-74
                    }
                    ti = last ? -1 : ti + 1;
                }
            }
        }
    }
    if (STATS) { flush_counts(p.stats, tc, rays, true); flush_wave_iters(p.stats, wIn, wTr, true); }
}

template <bool STATS, bool OVF, bool TOP, int REPS, int BURST>
__global__ __launch_bounds__(kDefBlock) void k_shadow_e(RenderParams p)
{
    __shared__ int s_stack[kDefStack * kDefBlock];
    __shared__ int s_fifo[kFifo * kDefBlock];
    __shared__ float4 s_top[TOP ? kTopTreeNodes * 4 : 4];
    if (TOP) {
        const float4* src = reinterpret_cast<const float4*>(p.trav.nodes);
        for (uint32_t k = threadIdx.x; k < p.trav.topCount * 4u; k += kDefBlock) s_top[k] = src[k];
    }
    const uint32_t gtid = blockIdx.x * kDefBlock + threadIdx.x;
    const uint32_t count = p.qc[QC_SHADOWRAY];                               // shadowRayCast.hlsl:151
    __syncthreads();
    if (gtid == 0) {
        // :144-148: QC[0..3] = (0, QC1 + QC0, 0, 0).  Nothing else in this kernel reads those words.
        const uint32_t q0 = p.qc[QC_NEWPATH], q1 = p.qc[QC_LASTPATHCNT];
        p.qc[QC_NEWPATH] = 0; p.qc[QC_LASTPATHCNT] = q0 + q1; p.qc[QC_MATUE4] = 0; p.qc[QC_MATGLASS] = 0;
    }
    DefStack<OVF> stk; stk.lds = s_stack + threadIdx.x; stk.ovf = p.ovfStack + gtid; stk.ovfStride = p.ovfStride; stk.ptr = 1;
    s_stack[threadIdx.x] = kDone;
    int* fifo = s_fifo + threadIdx.x;
    TravCount tc = { 0, 0, 0 }; uint32_t rays = 0, wIn = 0, wTr = 0;
    const TravScene& ts = p.trav;
    const uint32_t* qSh = p.queues + (size_t)Q_SHADOW_RAY * p.P;
    uint32_t next = 0, end = 0;
    bool drained = false;

    bool haveRay = false, occluded = false;
    uint32_t index = 0;
    f3 o = mk3(0, 0, 0), d = mk3(0, 0, 1), invdir = mk3(0, 0, 0);
    float lightDistance = 0.0f;
    int cur = kDone;
    uint32_t qHead = 0, qCount = 0;
    int ti = -1;
    bool resValid = false; uint32_t rIndex = 0; bool rOccluded = false;
    int pf = 0;
    uint32_t pfIndex = 0; f3 pfO = mk3(0, 0, 0), pfD = mk3(0, 0, 1), pfInv = mk3(0, 0, 0); float pfLight = 0.0f; int pfCur = kDone;

    for (;;) {
        if (haveRay && cur == kDone && qCount == 0 && ti < 0 && !resValid) { resValid = true; rIndex = index; rOccluded = occluded; haveRay = false; }
        if (!haveRay && pf == 3) {
            haveRay = true; index = pfIndex; o = pfO; d = pfD; invdir = pfInv; lightDistance = pfLight; cur = pfCur; occluded = false;
            stk.reset(); qHead = 0; qCount = 0; ti = -1;
            pf = 0;
            if (STATS) rays++;
        }
        const unsigned long long busyMask = __ballot(haveRay);
        {
            const int nRes = __popcll(__ballot(resValid));
            const bool blocked = haveRay && cur == kDone && qCount == 0 && ti < 0;
            if (nRes >= kFinishBatch || __ballot(blocked) != 0ull || (busyMask == 0ull && nRes > 0)) {
                if (resValid) { stu(p, F_IN_SHADOW, rIndex, rOccluded ? 1u : 0u); resValid = false; }   // :167
            }
        }
        if (pf == 2) {
            pfInv = mk3(1.0f / pfD.x, 1.0f / pfD.y, 1.0f / pfD.z);
            pfCur = (ray_box(ts.rootMin[0], ts.rootMin[1], ts.rootMin[2], ts.rootMax[0], ts.rootMax[1], ts.rootMax[2], pfO, pfInv) > 0.0f) ? ts.rootDesc : kDone;
            pf = 3;
        }
        if (pf == 1) { pfO = ld3(p, F_SH_OX, pfIndex); pfD = ld3(p, F_SH_DX, pfIndex); pfLight = ldf(p, F_LIGHT_DIST, pfIndex); pf = 2; } // :162-164
        {
            const unsigned long long needMask = __ballot(pf == 0);
            const int nNeed = __popcll(needMask);
            if (!drained && (nNeed >= kPrefetchBatch || (busyMask == 0ull && nNeed > 0))) {
                if (next >= end) {
                    uint32_t base = 0;
                    if ((threadIdx.x & 63) == 0) base = atomicAdd(&p.travCounters[1], p.raysPerWave);
                    base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
                    next = base; end = (base + p.raysPerWave < count) ? base + p.raysPerWave : count;
                    if (base >= count) { drained = true; next = end = 0; }
                }
                if (pf == 0) {
                    const uint32_t my = next + prefix_rank(needMask);
                    if (my < end) { pfIndex = qSh[my]; pf = 1; }                  // :159
                }
                next = (next + (uint32_t)nNeed < end) ? next + (uint32_t)nNeed : end;
            }
        }
        if (drained && busyMask == 0ull && __ballot(resValid || pf != 0) == 0ull) break;

#pragma unroll
        for (int rep = 0; rep < REPS; rep++) {
            if (cur >= 0) {
                if (STATS) { tc.inner++; if (prefix_rank(__ballot(1)) == 0) wIn++; }
                cur = inner_step_d<OVF, TOP>(ts, s_top, cur, o, invdir, stk, p.stats);
            }
            if (cur < 0 && cur != kDone && qCount < (uint32_t)kFifo) {
                if (STATS) tc.leaves++;
                fifo[((qHead + qCount) & (kFifo - 1)) * kDefBlock] = ~cur;
                qCount++;
                cur = stk.pop();
            }
        }
        const bool pending = (qCount > 0) || (ti >= 0);
        const int nPending = __popcll(__ballot(pending));
        const int nWalking = __popcll(__ballot(cur >= 0));
        if (nPending >= (int)p.tuneTriThresh || (nWalking == 0 && nPending > 0)) {
#pragma unroll
            for (int k = 0; k < BURST; k++) {
                if (ti < 0 && qCount > 0) { ti = fifo[(qHead & (kFifo - 1)) * kDefBlock]; qHead++; qCount--; }
                if (ti >= 0) {
                    if (STATS) { tc.tris++; if (prefix_rank(__ballot(1)) == 0) wTr++; }
                    float t = 0.0f, u = 0.0f, v = 0.0f; bool last;
                    if (tri_test(ts.tris, ti, o, d, t, u, v, last)) {
                        // shadowRayCast.hlsl:41-45,89: t in (1e-8, 1e8) and |d t| < lightDistance => occluded: the ray is decided
                        if (t > kEpsilon && t < 1.0f / kEpsilon && length3(d * t) < lightDistance) { occluded = true; last = true; qCount = 0; cur = kDone; }
                    }
                    ti = last ? -1 : ti + 1;
                }
            }
        }
    }
    if (STATS) { flush_counts(p.stats, tc, rays, false); flush_wave_iters(p.stats, wIn, wTr, false); }
}

// ------------------------------------------------------------------------------------------------ host launchers
#define GMUPT_DEF_LAUNCH(KERNEL, TOP, REPS, BURST) \
    do { if (ovf) { if (stats) hipLaunchKernelGGL((KERNEL<true, true, TOP, REPS, BURST>), dim3(pb), dim3(kDefBlock), 0, s, p); \
                    else hipLaunchKernelGGL((KERNEL<false, true, TOP, REPS, BURST>), dim3(pb), dim3(kDefBlock), 0, s, p); } \
         else { if (stats) hipLaunchKernelGGL((KERNEL<true, false, TOP, REPS, BURST>), dim3(pb), dim3(kDefBlock), 0, s, p); \
                else hipLaunchKernelGGL((KERNEL<false, false, TOP, REPS, BURST>), dim3(pb), dim3(kDefBlock), 0, s, p); } } while (0)

// both casts in one launch (modes 60..); launch_extend/launch_shadow are not called for these modes
bool traversal_is_fused(int mode) { return mode >= 60; }
void launch_cast(const RenderParams& p, bool stats, int mode, hipStream_t s)
{
    const uint32_t pb = p.travGridBlocks;
    const bool ovf = p.trav.maxDepth + 2 > (uint32_t)kDefStack;
    const bool plain = !p.extendPrune && !p.shadowPrune;   // the opt-in prunings exist in the separate bodies only
    // k_cast_f addresses nodes and triangle records with 32-bit byte offsets into buffer resources (< 2 GiB each: 33 M nodes, 44 M records)
    const bool fits = (uint64_t)p.trav.triBase * 64ull < (1ull << 31) && ((uint64_t)p.scene.numTris + 1ull) * 48ull < (1ull << 31);
    if (mode == 60 && plain && fits) GMUPT_DEF_LAUNCH(k_cast_f, true, 6, 4);   // cast0 (default): mixed lanes, fused fetches
    else if ((mode == 62 || mode == 60 || mode == 63) && plain) { if (mode == 63 && fits) GMUPT_DEF_LAUNCH(k_cast_f, true, 4, 4); else GMUPT_DEF_LAUNCH(k_cast_m, true, 4, 4); } // cast2: mixed lanes, separate triangle bursts; cast3: cast0 with 4 steps per iteration
    else GMUPT_DEF_LAUNCH(k_cast_d, true, 4, 4);
}

void launch_extend(const RenderParams& p, uint32_t blocks, bool stats, int mode, hipStream_t s)
{
    if (mode >= 60) mode = 40;
    if (mode >= 50) {
        const uint32_t pb = p.travGridBlocks;
        const bool ovf = p.trav.maxDepth + 2 > (uint32_t)kDefStack;
        switch (mode - 50) {
        case 0: GMUPT_DEF_LAUNCH(k_extend_e, true, 4, 4); break;
        case 1: GMUPT_DEF_LAUNCH(k_extend_e, true, 3, 4); break;
        default: GMUPT_DEF_LAUNCH(k_extend_e, true, 2, 4); break;
        }
    } else if (mode >= 40) {
        const uint32_t pb = p.travGridBlocks;
        const bool ovf = p.trav.maxDepth + 2 > (uint32_t)kDefStack;
        switch (mode - 40) {
        case 0: GMUPT_DEF_LAUNCH(k_extend_d, true, 4, 4); break;
        case 1: GMUPT_DEF_LAUNCH(k_extend_d, false, 4, 4); break;
        case 2: GMUPT_DEF_LAUNCH(k_extend_d, true, 3, 4); break;
        case 3: GMUPT_DEF_LAUNCH(k_extend_d, true, 2, 4); break;
        default: GMUPT_DEF_LAUNCH(k_extend_d, true, 4, 2); break;
        }
    } else launch_extend_variant(p, blocks, stats, mode, s);
}
void launch_shadow(const RenderParams& p, uint32_t blocks, bool stats, int mode, hipStream_t s)
{
    if (mode >= 60) mode = 40;
    if (mode >= 50) {
        const uint32_t pb = p.travGridBlocks;
        const bool ovf = p.trav.maxDepth + 2 > (uint32_t)kDefStack;
        switch (mode - 50) {
        case 0: GMUPT_DEF_LAUNCH(k_shadow_e, true, 4, 4); break;
        case 1: GMUPT_DEF_LAUNCH(k_shadow_e, true, 3, 4); break;
        default: GMUPT_DEF_LAUNCH(k_shadow_e, true, 2, 4); break;
        }
    } else if (mode >= 40) {
        const uint32_t pb = p.travGridBlocks;
        const bool ovf = p.trav.maxDepth + 2 > (uint32_t)kDefStack;
        switch (mode - 40) {
        case 0: GMUPT_DEF_LAUNCH(k_shadow_d, true, 4, 4); break;
        case 1: GMUPT_DEF_LAUNCH(k_shadow_d, false, 4, 4); break;
        case 2: GMUPT_DEF_LAUNCH(k_shadow_d, true, 3, 4); break;
        case 3: GMUPT_DEF_LAUNCH(k_shadow_d, true, 2, 4); break;
        default: GMUPT_DEF_LAUNCH(k_shadow_d, true, 4, 2); break;
        }
    } else launch_shadow_variant(p, blocks, stats, mode, s);
}
uint32_t traversal_block_threads() { return kTravBlock; }
uint32_t deferred_block_threads() { return kDefBlock; }
uint32_t traversal_overflow_entries() { const uint32_t a = (uint32_t)(kMaxStack + 1 - kDefStack), b = variant_overflow_entries(); return a > b ? a : b; }

} // namespace gmupt
