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

template <bool OVF, bool TOP>
__device__ __forceinline__ int inner_step_d(const TravScene& ts, const float4* s_top, int cur, f3 o, f3 invdir, DefStack<OVF>& stk, DevStats* dst)
{
    float4 a, b, c; int4 d;
    if (TOP && (uint32_t)cur < ts.topCount) { // the first levels of the tree live in LDS (48 % of all inner-node visits on the bench scene)
        const float4* n = s_top + cur * 4;
        a = n[0]; b = n[1]; c = n[2]; d = *reinterpret_cast<const int4*>(n + 3);
    } else {
        const float4* n = reinterpret_cast<const float4*>(ts.nodes + cur);
        a = n[0]; b = n[1]; c = n[2]; d = *reinterpret_cast<const int4*>(n + 3);
    }
    const float leftHit = ray_box(a.x, a.y, a.z, a.w, b.x, b.y, o, invdir);
    const float rightHit = ray_box(b.z, b.w, c.x, c.y, c.z, c.w, o, invdir);
    const bool l = leftHit > 0.0f, r = rightHit > 0.0f;
    const bool swap = leftHit > rightHit;            // extensionRayCast.hlsl:136: nearer child first, the other one deferred
    if (l && r) { stk.push(swap ? d.x : d.y, dst); return swap ? d.y : d.x; }
    if (l | r) return l ? d.x : d.y;
    return stk.pop();
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
    float4 a, b, c; int4 d;
    if (TOP && (uint32_t)cur < ts.topCount) {
        const float4* n = s_top + cur * 4;
        a = n[0]; b = n[1]; c = n[2]; d = *reinterpret_cast<const int4*>(n + 3);
    } else {
        const float4* n = reinterpret_cast<const float4*>(ts.nodes + cur);
        a = n[0]; b = n[1]; c = n[2]; d = *reinterpret_cast<const int4*>(n + 3);
    }
    float le, re;
    const float leftHit = ray_box_entry(a.x, a.y, a.z, a.w, b.x, b.y, o, invdir, le);
    const float rightHit = ray_box_entry(b.z, b.w, c.x, c.y, c.z, c.w, o, invdir, re);
    const bool l = leftHit > 0.0f && le <= limitT, r = rightHit > 0.0f && re <= limitT;
    const bool swap = leftHit > rightHit;
    if (l && r) { stk.push(swap ? d.x : d.y, dst); return swap ? d.y : d.x; }
    if (l | r) return l ? d.x : d.y;
    return stk.pop();
}

template <bool STATS, bool OVF, bool TOP, int REPS, int BURST>
__global__ __launch_bounds__(kDefBlock) void k_extend_d(RenderParams p)
{
    constexpr int WORK_COUNTER = 0;
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
    const float sceneEps = 1.0e-4f * (dabs(p.trav.rootMax[0] - p.trav.rootMin[0]) + dabs(p.trav.rootMax[1] - p.trav.rootMin[1]) + dabs(p.trav.rootMax[2] - p.trav.rootMin[2]));
    TravCount tc = { 0, 0, 0 }; uint32_t rays = 0, wIn = 0, wTr = 0;
    const TravScene& ts = p.trav;
    const uint32_t count = p.qc[QC_EXT_COUNT];
    const uint32_t* qExt = p.queues + (size_t)Q_EXT_RAY * p.P;
    // work distribution: a persistent grid; every wave takes chunks of p.raysPerWave queue entries from a device counter
    // (one atomic per chunk; the counter is zeroed by k_scan) -- no wave waits for another one
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

        if (STATS) { // lane census (collect_stats only): where do the 64 lanes of a wave spend the loop iterations?
            const bool pendingNow = (qCount > 0) || (ti >= 0);
            const uint32_t nI = __popcll(__ballot(cur == kDone && !pendingNow)), nW = __popcll(__ballot(cur >= 0));
            const uint32_t nS = __popcll(__ballot(cur < 0 && cur != kDone)), nP = __popcll(__ballot(cur == kDone && pendingNow));
            if ((threadIdx.x & 63) == 0) { atomicAdd(&p.stats->extDepthHist[28], (unsigned long long)nI); atomicAdd(&p.stats->extDepthHist[29], (unsigned long long)nW);
                                           atomicAdd(&p.stats->extDepthHist[30], (unsigned long long)nS); atomicAdd(&p.stats->extDepthHist[31], (unsigned long long)nP); }
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
__global__ __launch_bounds__(kDefBlock) void k_shadow_d(RenderParams p)
{
    constexpr int WORK_COUNTER = 1;
    __shared__ int s_stack[kDefStack * kDefBlock];
    __shared__ int s_fifo[kFifo * kDefBlock];
    __shared__ float4 s_top[TOP ? kTopTreeNodes * 4 : 4];
    if (TOP) {
        const float4* src = reinterpret_cast<const float4*>(p.trav.nodes);
        for (uint32_t k = threadIdx.x; k < p.trav.topCount * 4u; k += kDefBlock) s_top[k] = src[k];
        __syncthreads();
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

// ------------------------------------------------------------------------------------------------ host launchers
#define GMUPT_DEF_LAUNCH(KERNEL, TOP, REPS, BURST) \
    do { if (ovf) { if (stats) hipLaunchKernelGGL((KERNEL<true, true, TOP, REPS, BURST>), dim3(pb), dim3(kDefBlock), 0, s, p); \
                    else hipLaunchKernelGGL((KERNEL<false, true, TOP, REPS, BURST>), dim3(pb), dim3(kDefBlock), 0, s, p); } \
         else { if (stats) hipLaunchKernelGGL((KERNEL<true, false, TOP, REPS, BURST>), dim3(pb), dim3(kDefBlock), 0, s, p); \
                else hipLaunchKernelGGL((KERNEL<false, false, TOP, REPS, BURST>), dim3(pb), dim3(kDefBlock), 0, s, p); } } while (0)

void launch_extend(const RenderParams& p, uint32_t blocks, bool stats, int mode, hipStream_t s)
{
    if (mode >= 40) {
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
    if (mode >= 40) {
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
