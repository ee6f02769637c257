// A/B VARIANTS of the ray-cast kernels (selected with GMUPT_TRAVERSAL=...): the rungs of the optimisation ladder of DESIGN.md section 4.
// The shipped kernels live in pt_traverse.hip.  Every variant passes the same bitwise parity suite.
//
//   k_extend / k_shadow          packed traversal data (default): 64-byte two-child nodes, 48-byte pre-gathered triangles,
//                                while-while loop structure, per-lane LDS stacks.
//   k_extend_ref / k_shadow_ref  the same algorithm straight on the reference's buffers (48-B nodes, 16-B triangle records,
//                                12-B vertices).  Kept for A/B timing in one process (GMUPT_TRAVERSAL=ref) and as the
//                                executable statement of what the packed kernels must reproduce bit for bit.
//
// What must not change (it decides results): the slab test arithmetic and its "hit iff result > 0" rule with no pruning
// against the current closest hit (extensionRayCast.hlsl:79-94,132-159), the near-child-first visit order (closest-hit ties
// are resolved by visit order, `t < distance` strict, :64-74), the Moeller-Trumbore operation order (:38-77), and the
// shadow acceptance rule t in (1e-8, 1e8), |d t| < lightDistance (shadowRayCast.hlsl:16-47,88-91).
// What is free: memory layout, loop structure, and -- for the any-hit shadow ray only -- the visit order.
#ifdef GMUPT_VARIANTS   // the whole file: A/B rungs, not part of the shipped library
#include "pt_traverse_deferred.hpp"

namespace gmupt {

// rayAABBIntersection: extensionRayCast.hlsl:79-94 == shadowRayCast.hlsl:49-63
__device__ __forceinline__ float ray_aabb(float4 mn, float4 mx, f3 o, f3 invdir)
{
    const float fx = (mx.x - o.x) * invdir.x, fy = (mx.y - o.y) * invdir.y, fz = (mx.z - o.z) * invdir.z;
    const float nx = (mn.x - o.x) * invdir.x, ny = (mn.y - o.y) * invdir.y, nz = (mn.z - o.z) * invdir.z;
    const float tmaxx = hmax(fx, nx), tmaxy = hmax(fy, ny), tmaxz = hmax(fz, nz);
    const float tminx = hmin(fx, nx), tminy = hmin(fy, ny), tminz = hmin(fz, nz);
    const float t1 = hmin(tmaxx, hmin(tmaxy, tmaxz));
    const float t0 = hmax(tminx, hmax(tminy, tminz));
    return (t1 >= t0) ? (t0 > 0.0f ? t0 : t1) : -1.0f;
}

constexpr int kLdsStack = 24;   // entries per lane kept in LDS; deeper entries spill to a global overflow array

struct TravStack {
    int* lds;       // s_stack + threadIdx.x, stride kTravBlock
    int* ovf;       // global overflow + global thread id, stride ovfStride
    uint32_t ovfStride;
    uint32_t ptr;
    __device__ __forceinline__ void push(int v, DevStats* st)
    {
        if (ptr < kLdsStack) lds[ptr * kTravBlock] = v;
        else if (ptr < kMaxStack) ovf[(size_t)(ptr - kLdsStack) * ovfStride] = v;
        else atomicOr(&st->stackOverflow, 1u);
        ptr++;
    }
    __device__ __forceinline__ int pop()
    {
        if (ptr == 0) return -1;   // the reference's sentinel stack[0] = -1 (extensionRayCast.hlsl:100)
        --ptr;
        if (ptr < kLdsStack) return lds[ptr * kTravBlock];
        if (ptr < kMaxStack) return ovf[(size_t)(ptr - kLdsStack) * ovfStride];
        return -1;
    }
};


struct ExtHit { f3 hitPoint, bary; int4 tri; };

__device__ __forceinline__ f3 load_vertex(const float* verts, int idx) { const float* v = verts + 3 * (size_t)idx; return mk3(v[0], v[1], v[2]); }

// extensionRayCast.hlsl:96-166 + rayTriangleIntersection :38-77
template <bool STATS>
__device__ __forceinline__ float bvh_closest(const SceneView& sc, f3 o, f3 d, ExtHit& hit, TravStack& stk, DevStats* dst, TravCount& tc)
{
    float distance = kFltMax;
    const f3 invdir = mk3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
    stk.ptr = 0;
    const DNode root = sc.nodes[0];
    if (!(ray_aabb(root.mn, root.mx, o, invdir) > 0.0f)) return distance;
    int4 link = root.link;
    for (;;) {
        if (link.z) { // leaf
            if (STATS) tc.leaves++;
            for (int i = link.x; i < link.y; i++) {
                const int4 T = *reinterpret_cast<const int4*>(&sc.tris[i]);
                const f3 v0 = load_vertex(sc.verts, T.x), v1 = load_vertex(sc.verts, T.y), v2 = load_vertex(sc.verts, T.z);
                if (STATS) tc.tris++;
                const f3 e1 = v1 - v0, e2 = v2 - v0;
                const f3 pvec = cross3(d, e2);
                const float det = dot3(e1, pvec);
                if (det > -kEpsilon && det < kEpsilon) continue;
                const float invDet = 1.0f / det;
                const f3 tvec = o - v0;
                const float u = dot3(tvec, pvec) * invDet;
                if (u < 0.0f || u > 1.0f) continue;
                const f3 qvec = cross3(tvec, e1);
                const float v = dot3(d, qvec) * invDet;
                if (v < 0.0f || u + v > 1.0f) continue;
                const float t = dot3(e2, qvec) * invDet;
                if (t >= 0.0f && t < distance) {
                    distance = t;
                    hit.hitPoint = o + d * t;
                    hit.bary = mk3(1.0f - u - v, u, v);
                    hit.tri = T;
                }
            }
        } else {
            if (STATS) tc.inner++;
            const DNode left = sc.nodes[link.x];
            const DNode right = sc.nodes[link.y];
            const float leftHit = ray_aabb(left.mn, left.mx, o, invdir);
            const float rightHit = ray_aabb(right.mn, right.mx, o, invdir);
            if (leftHit > 0.0f && rightHit > 0.0f) {
                if (leftHit > rightHit) { stk.push(link.x, dst); link = right.link; }
                else { stk.push(link.y, dst); link = left.link; }
                continue;
            } else if (leftHit > 0.0f) { link = left.link; continue; }
            else if (rightHit > 0.0f) { link = right.link; continue; }
        }
        const int idx = stk.pop();
        if (idx < 0) break;
        link = sc.nodes[idx].link;
    }
    return distance;
}

// shadowRayCast.hlsl:65-136 + rayTriangleIntersection :16-47
template <bool STATS>
__device__ __forceinline__ bool bvh_any(const SceneView& sc, f3 o, f3 d, float lightDistance, TravStack& stk, DevStats* dst, TravCount& tc)
{
    const f3 invdir = mk3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
    stk.ptr = 0;
    const DNode root = sc.nodes[0];
    if (!(ray_aabb(root.mn, root.mx, o, invdir) > 0.0f)) return false;
    int4 link = root.link;
    for (;;) {
        if (link.z) {
            if (STATS) tc.leaves++;
            for (int i = link.x; i < link.y; i++) {
                const int4 T = *reinterpret_cast<const int4*>(&sc.tris[i]);
                const f3 v0 = load_vertex(sc.verts, T.x), v1 = load_vertex(sc.verts, T.y), v2 = load_vertex(sc.verts, T.z);
                if (STATS) tc.tris++;
                const f3 e1 = v1 - v0, e2 = v2 - v0;
                const f3 pvec = cross3(d, e2);
                const float det = dot3(e1, pvec);
                if (det > -kEpsilon && det < kEpsilon) continue;
                const float invDet = 1.0f / det;
                const f3 tvec = o - v0;
                const float u = dot3(tvec, pvec) * invDet;
                if (u < 0.0f || u > 1.0f) continue;
                const f3 qvec = cross3(tvec, e1);
                const float v = dot3(d, qvec) * invDet;
                if (v < 0.0f || u + v > 1.0f) continue;
                const float t = dot3(e2, qvec) * invDet;
                if (t > kEpsilon && t < 1.0f / kEpsilon) {
                    const float dist = length3(d * t);
                    if (dist < lightDistance) return true;
                }
            }
        } else {
            if (STATS) tc.inner++;
            const DNode left = sc.nodes[link.x];
            const DNode right = sc.nodes[link.y];
            const float leftHit = ray_aabb(left.mn, left.mx, o, invdir);
            const float rightHit = ray_aabb(right.mn, right.mx, o, invdir);
            if (leftHit > 0.0f && rightHit > 0.0f) {
                if (leftHit > rightHit) { stk.push(link.x, dst); link = right.link; }
                else { stk.push(link.y, dst); link = left.link; }
                continue;
            } else if (leftHit > 0.0f) { link = left.link; continue; }
            else if (rightHit > 0.0f) { link = right.link; continue; }
        }
        const int idx = stk.pop();
        if (idx < 0) break;
        link = sc.nodes[idx].link;
    }
    return false;
}

template <bool STATS>
__global__ __launch_bounds__(kTravBlock) void k_extend_ref(RenderParams p)
{
    __shared__ int s_stack[kLdsStack * kTravBlock];
    const uint32_t gtid = blockIdx.x * kTravBlock + threadIdx.x;
    const uint32_t stride = gridDim.x * kTravBlock;
    TravStack stk; stk.lds = s_stack + threadIdx.x; stk.ovf = p.ovfStack + gtid; stk.ovfStride = p.ovfStride; stk.ptr = 0;
    TravCount tc = { 0, 0, 0 }; uint32_t rays = 0;
    const uint32_t count = p.qc[QC_EXT_COUNT];
    const uint32_t* qExt = p.queues + (size_t)Q_EXT_RAY * p.P;
    for (uint32_t q = gtid; q < count; q += stride) {
        const uint32_t index = qExt[q];                                      // extensionRayCast.hlsl:210
        if (index == kQueueHole) continue;
        const f3 o = ld3(p, F_RAY_OX, index), d = ld3(p, F_RAY_DX, index);   // :213-214
        ExtHit hit; hit.hitPoint = mk3(0, 0, 0); hit.bary = mk3(0, 0, 0); hit.tri = make_int4(0, 0, 0, 0);
        float distance = bvh_closest<STATS>(p.scene, o, d, hit, stk, p.stats, tc); // :216
        if (STATS) rays++;
        if (distance < kFltMax) {                                            // :218-225
            st3(p, F_SP_X, index, hit.hitPoint);
            st3(p, F_BARY_X, index, hit.bary);
            stu(p, F_TRI_0, index, (uint32_t)hit.tri.x); stu(p, F_TRI_1, index, (uint32_t)hit.tri.y);
            stu(p, F_TRI_2, index, (uint32_t)hit.tri.z); stu(p, F_TRI_MAT, index, (uint32_t)hit.tri.w);
        }
        // rayLightIntersection :168-194
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
        stu(p, F_IS_EMITTER, index, lightIndex);                             // :231
        stf(p, F_HIT_DIST, index, distance);                                 // :232
    }
    if (STATS) flush_counts(p.stats, tc, rays, true);
}

template <bool STATS>
__global__ __launch_bounds__(kTravBlock) void k_shadow_ref(RenderParams p)
{
    __shared__ int s_stack[kLdsStack * kTravBlock];
    const uint32_t gtid = blockIdx.x * kTravBlock + threadIdx.x;
    const uint32_t stride = gridDim.x * kTravBlock;
    const uint32_t count = p.qc[QC_SHADOWRAY];                               // shadowRayCast.hlsl:151
    __syncthreads();
    if (gtid == 0) {
        // :144-148: QC[0..3] = (0, QC1 + QC0, 0, 0).  Nothing else in this kernel reads those words.
        const uint32_t q0 = p.qc[QC_NEWPATH], q1 = p.qc[QC_LASTPATHCNT];
        p.qc[QC_NEWPATH] = 0; p.qc[QC_LASTPATHCNT] = q0 + q1; p.qc[QC_MATUE4] = 0; p.qc[QC_MATGLASS] = 0;
    }
    TravStack stk; stk.lds = s_stack + threadIdx.x; stk.ovf = p.ovfStack + gtid; stk.ovfStride = p.ovfStride; stk.ptr = 0;
    TravCount tc = { 0, 0, 0 }; uint32_t rays = 0;
    const uint32_t* qSh = p.queues + (size_t)Q_SHADOW_RAY * p.P;
    for (uint32_t q = gtid; q < count; q += stride) {
        const uint32_t index = qSh[q];                                       // :159
        const f3 o = ld3(p, F_SH_OX, index), d = ld3(p, F_SH_DX, index);     // :162-163
        const float lightDistance = ldf(p, F_LIGHT_DIST, index);             // :164
        const bool inShadow = bvh_any<STATS>(p.scene, o, d, lightDistance, stk, p.stats, tc); // :166
        if (STATS) rays++;
        stu(p, F_IN_SHADOW, index, inShadow ? 1u : 0u);                      // :167
    }
    if (STATS) flush_counts(p.stats, tc, rays, false);
}


// ------------------------------------------------------------------------------------------------ packed traversal data

struct PackedStack {
    int* lds; int* ovf; uint32_t ovfStride; uint32_t ptr;
    __device__ __forceinline__ void push(int v, DevStats* st)
    {
        if (ptr < kLdsStack) lds[ptr * kTravBlock] = v;
        else if (ptr < kMaxStack) ovf[(size_t)(ptr - kLdsStack) * ovfStride] = v;
        else atomicOr(&st->stackOverflow, 1u);
        ptr++;
    }
    __device__ __forceinline__ int pop()
    {
        if (ptr == 0) return kDone;
        --ptr;
        if (ptr < kLdsStack) return lds[ptr * kTravBlock];
        if (ptr < kMaxStack) return ovf[(size_t)(ptr - kLdsStack) * ovfStride];
        return kDone;
    }
};

// one inner step: test both children, descend into the nearer hit child, defer the other (extensionRayCast.hlsl:126-159)
__device__ __forceinline__ int inner_step(const TravScene& ts, int cur, f3 o, f3 invdir, PackedStack& stk, DevStats* dst)
{
    const float4* n = reinterpret_cast<const float4*>(ts.nodes + cur);
    const float4 a = n[0], b = n[1], c = n[2];
    const int4 d = *reinterpret_cast<const int4*>(n + 3);
#ifdef GMUPT_EXPERIMENT_EXTRA_LOAD
    { const volatile float4* vn = reinterpret_cast<const volatile float4*>(n); float ex = vn[0].x; float ey = vn[2].y; asm volatile("" :: "v"(ex), "v"(ey)); }
#endif
    const float leftHit = ray_box(a.x, a.y, a.z, a.w, b.x, b.y, o, invdir);
    const float rightHit = ray_box(b.z, b.w, c.x, c.y, c.z, c.w, o, invdir);
    if (leftHit > 0.0f && rightHit > 0.0f) {
        if (leftHit > rightHit) { stk.push(d.x, dst); return d.y; }
        stk.push(d.y, dst); return d.x;
    }
    if (leftHit > 0.0f) return d.x;
    if (rightHit > 0.0f) return d.y;
    return stk.pop();
}

template <bool STATS>
__device__ __forceinline__ float packed_closest(const TravScene& ts, f3 o, f3 d, int& hitRef, float& hitU, float& hitV, PackedStack& stk, DevStats* dst, TravCount& tc)
{
    float distance = kFltMax;
    hitRef = -1; hitU = 0.0f; hitV = 0.0f;
    const f3 invdir = mk3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
    stk.ptr = 0;
    if (!(ray_box(ts.rootMin[0], ts.rootMin[1], ts.rootMin[2], ts.rootMax[0], ts.rootMax[1], ts.rootMax[2], o, invdir) > 0.0f)) return distance;
    int cur = ts.rootDesc;
    while (cur != kDone) {
        while (cur >= 0) {
            if (STATS) tc.inner++;
            cur = inner_step(ts, cur, o, invdir, stk, dst);
        }
        if (cur == kDone) break;
        if (STATS) tc.leaves++;
        int i = ~cur;
        bool last;
        do {
            const float4* r = reinterpret_cast<const float4*>(ts.tris + i);
            const float4 r0 = r[0], r1 = r[1], r2 = r[2];
            last = __builtin_bit_cast(uint32_t, r2.y) != 0u;
            if (STATS) tc.tris++;
            const f3 v0 = mk3(r0.x, r0.y, r0.z), e1 = mk3(r0.w, r1.x, r1.y), e2 = mk3(r1.z, r1.w, r2.x);
            const f3 pvec = cross3(d, e2);
            const float det = dot3(e1, pvec);
            if (!(det > -kEpsilon && det < kEpsilon)) {
                const float invDet = 1.0f / det;
                const f3 tvec = o - v0;
                const float u = dot3(tvec, pvec) * invDet;
                if (!(u < 0.0f || u > 1.0f)) {
                    const f3 qvec = cross3(tvec, e1);
                    const float v = dot3(d, qvec) * invDet;
                    if (!(v < 0.0f || u + v > 1.0f)) {
                        const float t = dot3(e2, qvec) * invDet;
                        if (t >= 0.0f && t < distance) { distance = t; hitRef = i; hitU = u; hitV = v; }
                    }
                }
            }
            i++;
        } while (!last);
        cur = stk.pop();
    }
    return distance;
}

template <bool STATS>
__device__ __forceinline__ bool packed_any(const TravScene& ts, f3 o, f3 d, float lightDistance, PackedStack& stk, DevStats* dst, TravCount& tc)
{
    const f3 invdir = mk3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
    stk.ptr = 0;
    if (!(ray_box(ts.rootMin[0], ts.rootMin[1], ts.rootMin[2], ts.rootMax[0], ts.rootMax[1], ts.rootMax[2], o, invdir) > 0.0f)) return false;
    int cur = ts.rootDesc;
    bool occluded = false;
    while (cur != kDone) {
        while (cur >= 0) {
            if (STATS) tc.inner++;
            cur = inner_step(ts, cur, o, invdir, stk, dst);
        }
        if (cur == kDone) break;
        if (STATS) tc.leaves++;
        int i = ~cur;
        bool last;
        do {
            const float4* r = reinterpret_cast<const float4*>(ts.tris + i);
            const float4 r0 = r[0], r1 = r[1], r2 = r[2];
            last = __builtin_bit_cast(uint32_t, r2.y) != 0u;
            if (STATS) tc.tris++;
            const f3 v0 = mk3(r0.x, r0.y, r0.z), e1 = mk3(r0.w, r1.x, r1.y), e2 = mk3(r1.z, r1.w, r2.x);
            const f3 pvec = cross3(d, e2);
            const float det = dot3(e1, pvec);
            if (!(det > -kEpsilon && det < kEpsilon)) {
                const float invDet = 1.0f / det;
                const f3 tvec = o - v0;
                const float u = dot3(tvec, pvec) * invDet;
                if (!(u < 0.0f || u > 1.0f)) {
                    const f3 qvec = cross3(tvec, e1);
                    const float v = dot3(d, qvec) * invDet;
                    if (!(v < 0.0f || u + v > 1.0f)) {
                        const float t = dot3(e2, qvec) * invDet;
                        if (t > kEpsilon && t < 1.0f / kEpsilon) {
                            if (length3(d * t) < lightDistance) { occluded = true; last = true; }
                        }
                    }
                }
            }
            i++;
        } while (!last);
        cur = occluded ? kDone : stk.pop();
    }
    return occluded;
}

template <bool STATS>
__global__ __launch_bounds__(kTravBlock) void k_extend(RenderParams p)
{
    __shared__ int s_stack[kLdsStack * kTravBlock];
    const uint32_t gtid = blockIdx.x * kTravBlock + threadIdx.x;
    const uint32_t stride = gridDim.x * kTravBlock;
    PackedStack stk; stk.lds = s_stack + threadIdx.x; stk.ovf = p.ovfStack + gtid; stk.ovfStride = p.ovfStride; stk.ptr = 0;
    TravCount tc = { 0, 0, 0 }; uint32_t rays = 0;
    const uint32_t count = p.qc[QC_EXT_COUNT];
    const uint32_t* qExt = p.queues + (size_t)Q_EXT_RAY * p.P;
    for (uint32_t q = gtid; q < count; q += stride) {
        const uint32_t index = qExt[q];                                      // extensionRayCast.hlsl:210
        if (index == kQueueHole) continue;
        const f3 o = ld3(p, F_RAY_OX, index), d = ld3(p, F_RAY_DX, index);   // :213-214
        int hitRef; float hu, hv;
        float distance = packed_closest<STATS>(p.trav, o, d, hitRef, hu, hv, stk, p.stats, tc); // :216
        if (STATS) rays++;
        if (distance < kFltMax) {                                            // :218-225
            st3(p, F_SP_X, index, o + d * distance);                         // :66,71 hitPoint = origin + direction * t
            st3(p, F_BARY_X, index, mk3(1.0f - hu - hv, hu, hv));            // :72
            const int4 T = *reinterpret_cast<const int4*>(&p.scene.tris[hitRef]); // :121 state.tri = indices[i]
            stu(p, F_TRI_0, index, (uint32_t)T.x); stu(p, F_TRI_1, index, (uint32_t)T.y);
            stu(p, F_TRI_2, index, (uint32_t)T.z); stu(p, F_TRI_MAT, index, (uint32_t)T.w);
        }
        uint32_t lightIndex = 0;                                             // rayLightIntersection :168-194
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
        stu(p, F_IS_EMITTER, index, lightIndex);                             // :231
        stf(p, F_HIT_DIST, index, distance);                                 // :232
    }
    if (STATS) flush_counts(p.stats, tc, rays, true);
}

template <bool STATS>
__global__ __launch_bounds__(kTravBlock) void k_shadow(RenderParams p)
{
    __shared__ int s_stack[kLdsStack * kTravBlock];
    const uint32_t gtid = blockIdx.x * kTravBlock + threadIdx.x;
    const uint32_t stride = gridDim.x * kTravBlock;
    const uint32_t count = p.qc[QC_SHADOWRAY];                               // shadowRayCast.hlsl:151
    __syncthreads();
    if (gtid == 0) {
        // :144-148: QC[0..3] = (0, QC1 + QC0, 0, 0).  Nothing else in this kernel reads those words.
        const uint32_t q0 = p.qc[QC_NEWPATH], q1 = p.qc[QC_LASTPATHCNT];
        p.qc[QC_NEWPATH] = 0; p.qc[QC_LASTPATHCNT] = q0 + q1; p.qc[QC_MATUE4] = 0; p.qc[QC_MATGLASS] = 0;
    }
    PackedStack stk; stk.lds = s_stack + threadIdx.x; stk.ovf = p.ovfStack + gtid; stk.ovfStride = p.ovfStride; stk.ptr = 0;
    TravCount tc = { 0, 0, 0 }; uint32_t rays = 0;
    const uint32_t* qSh = p.queues + (size_t)Q_SHADOW_RAY * p.P;
    for (uint32_t q = gtid; q < count; q += stride) {
        const uint32_t index = qSh[q];                                       // :159
        const f3 o = ld3(p, F_SH_OX, index), d = ld3(p, F_SH_DX, index);     // :162-163
        const float lightDistance = ldf(p, F_LIGHT_DIST, index);             // :164
        const bool inShadow = packed_any<STATS>(p.trav, o, d, lightDistance, stk, p.stats, tc); // :166
        if (STATS) rays++;
        stu(p, F_IN_SHADOW, index, inShadow ? 1u : 0u);                      // :167
    }
    if (STATS) flush_counts(p.stats, tc, rays, false);
}


// ------------------------------------------------------------------------------------------------ persistent-lane variants
// Rays differ a lot in length (no pruning: a ray visits every box its whole line pierces), so with one ray per lane a wave64
// spends most of its time waiting for its longest ray (measured VALU lane utilisation of the kernels above: 12 %).
// Here a wave owns a contiguous chunk of the queue and hands a new ray to a lane as soon as enough lanes are idle
// (Aila & Laine style lane refill, wave64 ballot + mbcnt ranks, no atomics: the chunk is private to the wave).
// Per-ray arithmetic and visit order are unchanged, so results are identical.
constexpr uint32_t kRaysPerWave = 256;   // queue entries owned by one wave
constexpr int kRefillThreshold = 20;     // refill when at least this many lanes are idle

template <bool STATS>
__global__ __launch_bounds__(kTravBlock) void k_extend_p(RenderParams p)
{
    __shared__ int s_stack[kLdsStack * kTravBlock];
    const uint32_t gtid = blockIdx.x * kTravBlock + threadIdx.x;
    PackedStack stk; stk.lds = s_stack + threadIdx.x; stk.ovf = p.ovfStack + gtid; stk.ovfStride = p.ovfStride; stk.ptr = 0;
    TravCount tc = { 0, 0, 0 }; uint32_t rays = 0; uint32_t wIn = 0, wTr = 0;
    const TravScene& ts = p.trav;
    const uint32_t count = p.qc[QC_EXT_COUNT];
    const uint32_t* qExt = p.queues + (size_t)Q_EXT_RAY * p.P;
    uint32_t next = (gtid >> 6) * kRaysPerWave;
    const uint32_t end = (next + kRaysPerWave < count) ? next + kRaysPerWave : count;
    if (next >= end) return; // wave-uniform

    bool haveRay = false;
    uint32_t index = 0;
    f3 o = mk3(0, 0, 0), d = mk3(0, 0, 1), invdir = mk3(0, 0, 0);
    float distance = kFltMax, hu = 0.0f, hv = 0.0f;
    int hitRef = -1;
    int cur = kDone;

    for (;;) {
        const bool idle = (cur == kDone);
        const unsigned long long idleMask = __ballot(idle);
        const int nIdle = __popcll(idleMask);
        if (nIdle == 64 || (nIdle >= kRefillThreshold && next < end)) { // wave-uniform
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
                        stk.ptr = 0;
                        cur = (ray_box(ts.rootMin[0], ts.rootMin[1], ts.rootMin[2], ts.rootMax[0], ts.rootMax[1], ts.rootMax[2], o, invdir) > 0.0f) ? ts.rootDesc : kDone;
                    }
                }
            }
            if (nIdle == 64 && next >= end) break; // nothing in flight and the chunk is exhausted (wave-uniform)
            next = (next + (uint32_t)nIdle < end) ? next + (uint32_t)nIdle : end;
        }
        while (cur >= 0) {
            if (STATS) { tc.inner++; if (prefix_rank(__ballot(1)) == 0) wIn++; }
            cur = inner_step(ts, cur, o, invdir, stk, p.stats);
        }
        if (cur != kDone) {
            if (STATS) tc.leaves++;
            int i = ~cur;
            bool last;
            do {
                const float4* r = reinterpret_cast<const float4*>(ts.tris + i);
                const float4 r0 = r[0], r1 = r[1], r2 = r[2];
                last = __builtin_bit_cast(uint32_t, r2.y) != 0u;
                if (STATS) { tc.tris++; if (prefix_rank(__ballot(1)) == 0) wTr++; }
                const f3 v0 = mk3(r0.x, r0.y, r0.z), e1 = mk3(r0.w, r1.x, r1.y), e2 = mk3(r1.z, r1.w, r2.x);
                const f3 pvec = cross3(d, e2);
                const float det = dot3(e1, pvec);
                if (!(det > -kEpsilon && det < kEpsilon)) {
                    const float invDet = 1.0f / det;
                    const f3 tvec = o - v0;
                    const float u = dot3(tvec, pvec) * invDet;
                    if (!(u < 0.0f || u > 1.0f)) {
                        const f3 qvec = cross3(tvec, e1);
                        const float v = dot3(d, qvec) * invDet;
                        if (!(v < 0.0f || u + v > 1.0f)) {
                            const float t = dot3(e2, qvec) * invDet;
                            if (t >= 0.0f && t < distance) { distance = t; hitRef = i; hu = u; hv = v; }
                        }
                    }
                }
                i++;
            } while (!last);
            cur = stk.pop();
        }
    }
    if (STATS) { flush_counts(p.stats, tc, rays, true); flush_wave_iters(p.stats, wIn, wTr, true); }
}

template <bool STATS>
__global__ __launch_bounds__(kTravBlock) void k_shadow_p(RenderParams p)
{
    __shared__ int s_stack[kLdsStack * kTravBlock];
    const uint32_t gtid = blockIdx.x * kTravBlock + threadIdx.x;
    const uint32_t count = p.qc[QC_SHADOWRAY];                               // shadowRayCast.hlsl:151
    __syncthreads();
    if (gtid == 0) {
        // :144-148: QC[0..3] = (0, QC1 + QC0, 0, 0).  Nothing else in this kernel reads those words.
        const uint32_t q0 = p.qc[QC_NEWPATH], q1 = p.qc[QC_LASTPATHCNT];
        p.qc[QC_NEWPATH] = 0; p.qc[QC_LASTPATHCNT] = q0 + q1; p.qc[QC_MATUE4] = 0; p.qc[QC_MATGLASS] = 0;
    }
    PackedStack stk; stk.lds = s_stack + threadIdx.x; stk.ovf = p.ovfStack + gtid; stk.ovfStride = p.ovfStride; stk.ptr = 0;
    TravCount tc = { 0, 0, 0 }; uint32_t rays = 0; uint32_t wIn = 0, wTr = 0;
    const TravScene& ts = p.trav;
    const uint32_t* qSh = p.queues + (size_t)Q_SHADOW_RAY * p.P;
    uint32_t next = (gtid >> 6) * kRaysPerWave;
    const uint32_t end = (next + kRaysPerWave < count) ? next + kRaysPerWave : count;
    if (next >= end) return;

    bool haveRay = false, occluded = false;
    uint32_t index = 0;
    f3 o = mk3(0, 0, 0), d = mk3(0, 0, 1), invdir = mk3(0, 0, 0);
    float lightDistance = 0.0f;
    int cur = kDone;

    for (;;) {
        const bool idle = (cur == kDone);
        const unsigned long long idleMask = __ballot(idle);
        const int nIdle = __popcll(idleMask);
        if (nIdle == 64 || (nIdle >= kRefillThreshold && next < end)) {
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
                    stk.ptr = 0;
                    cur = (ray_box(ts.rootMin[0], ts.rootMin[1], ts.rootMin[2], ts.rootMax[0], ts.rootMax[1], ts.rootMax[2], o, invdir) > 0.0f) ? ts.rootDesc : kDone;
                }
            }
            if (nIdle == 64 && next >= end) break;
            next = (next + (uint32_t)nIdle < end) ? next + (uint32_t)nIdle : end;
        }
        while (cur >= 0) {
            if (STATS) { tc.inner++; if (prefix_rank(__ballot(1)) == 0) wIn++; }
            cur = inner_step(ts, cur, o, invdir, stk, p.stats);
        }
        if (cur != kDone) {
            if (STATS) tc.leaves++;
            int i = ~cur;
            bool last;
            do {
                const float4* r = reinterpret_cast<const float4*>(ts.tris + i);
                const float4 r0 = r[0], r1 = r[1], r2 = r[2];
                last = __builtin_bit_cast(uint32_t, r2.y) != 0u;
                if (STATS) { tc.tris++; if (prefix_rank(__ballot(1)) == 0) wTr++; }
                const f3 v0 = mk3(r0.x, r0.y, r0.z), e1 = mk3(r0.w, r1.x, r1.y), e2 = mk3(r1.z, r1.w, r2.x);
                const f3 pvec = cross3(d, e2);
                const float det = dot3(e1, pvec);
                if (!(det > -kEpsilon && det < kEpsilon)) {
                    const float invDet = 1.0f / det;
                    const f3 tvec = o - v0;
                    const float u = dot3(tvec, pvec) * invDet;
                    if (!(u < 0.0f || u > 1.0f)) {
                        const f3 qvec = cross3(tvec, e1);
                        const float v = dot3(d, qvec) * invDet;
                        if (!(v < 0.0f || u + v > 1.0f)) {
                            const float t = dot3(e2, qvec) * invDet;
                            if (t > kEpsilon && t < 1.0f / kEpsilon) {
                                if (length3(d * t) < lightDistance) { occluded = true; last = true; }
                            }
                        }
                    }
                }
                i++;
            } while (!last);
            cur = occluded ? kDone : stk.pop();
        }
    }
    if (STATS) { flush_counts(p.stats, tc, rays, false); flush_wave_iters(p.stats, wIn, wTr, false); }
}

// ------------------------------------------------------------------------------------------------ interleaved (if-if) variants
// One unit of work per lane and loop iteration: an inner step OR one triangle test.  The while-while form above keeps lanes
// waiting at their leaf until the slowest lane of the wave has finished descending (measured lane utilisation 18 % in the
// inner loop, 32 % in the triangle loop); here a lane with a ray always has something to do.  `cur` alone carries the state:
// >= 0 inner node, kDone nothing, otherwise ~index of the NEXT triangle record of the current leaf.

template <bool STATS, int REPS, int REFILL>
__global__ __launch_bounds__(kTravBlock) void k_extend_i(RenderParams p)
{
    constexpr bool EXT_KERNEL = true;
    __shared__ int s_stack[kLdsStack * kTravBlock];
    const uint32_t gtid = blockIdx.x * kTravBlock + threadIdx.x;
    PackedStack stk; stk.lds = s_stack + threadIdx.x; stk.ovf = p.ovfStack + gtid; stk.ovfStride = p.ovfStride; stk.ptr = 0;
    TravCount tc = { 0, 0, 0 }; uint32_t rays = 0, wIn = 0, wTr = 0;
    const TravScene& ts = p.trav;
    const uint32_t count = p.qc[QC_EXT_COUNT];
    const uint32_t* qExt = p.queues + (size_t)Q_EXT_RAY * p.P;
    uint32_t next = (gtid >> 6) * p.raysPerWave;
    const uint32_t end = (next + p.raysPerWave < count) ? next + p.raysPerWave : count;
    if (next >= end) return; // wave-uniform

    bool haveRay = false;
    uint32_t index = 0;
    f3 o = mk3(0, 0, 0), d = mk3(0, 0, 1), invdir = mk3(0, 0, 0);
    float distance = kFltMax, hu = 0.0f, hv = 0.0f;
    int hitRef = -1;
    int cur = kDone;

    for (;;) {
        const bool idle = (cur == kDone);
        const unsigned long long idleMask = __ballot(idle);
        const int nIdle = __popcll(idleMask);
        if (nIdle == 64 || (nIdle >= REFILL && next < end)) { // wave-uniform
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
                        stk.ptr = 0;
                        cur = (ray_box(ts.rootMin[0], ts.rootMin[1], ts.rootMin[2], ts.rootMax[0], ts.rootMax[1], ts.rootMax[2], o, invdir) > 0.0f) ? ts.rootDesc : kDone;
                        if (STATS) { if (cur < 0 && cur != kDone) tc.leaves++; }
                    }
                }
            }
            if (nIdle == 64 && next >= end) break; // nothing in flight and the chunk is exhausted (wave-uniform)
            next = (next + (uint32_t)nIdle < end) ? next + (uint32_t)nIdle : end;
        }
        bool stepped = false;
#pragma unroll
        for (int rep = 0; rep < REPS; rep++) {
            if (cur >= 0) {
                if (STATS) { tc.inner++; if (prefix_rank(__ballot(1)) == 0) wIn++; if (EXT_KERNEL) atomicAdd(&p.stats->extDepthHist[ts.nodes[cur].d[2] & 31], 1ull); }
                cur = inner_step(ts, cur, o, invdir, stk, p.stats);
                if (STATS) { if (cur < 0 && cur != kDone) tc.leaves++; }
                stepped = true;
            }
        }
        if (!stepped && cur != kDone) {
            if (STATS) { tc.tris++; if (prefix_rank(__ballot(1)) == 0) wTr++; }
            const int i = ~cur;
            float t = 0.0f, u = 0.0f, v = 0.0f; bool last;
            if (tri_test(ts.tris, i, o, d, t, u, v, last)) {
                if (t >= 0.0f && t < distance) { distance = t; hitRef = i; hu = u; hv = v; } // extensionRayCast.hlsl:64-74
            }
            cur = last ? stk.pop() : ~(i + 1);
            if (STATS) { if (last && cur < 0 && cur != kDone) tc.leaves++; }
        }
    }
    if (STATS) { flush_counts(p.stats, tc, rays, true); flush_wave_iters(p.stats, wIn, wTr, true); }
}

template <bool STATS, int REPS, int REFILL>
__global__ __launch_bounds__(kTravBlock) void k_shadow_i(RenderParams p)
{
    constexpr bool EXT_KERNEL = false;
    __shared__ int s_stack[kLdsStack * kTravBlock];
    const uint32_t gtid = blockIdx.x * kTravBlock + threadIdx.x;
    const uint32_t count = p.qc[QC_SHADOWRAY];                               // shadowRayCast.hlsl:151
    __syncthreads();
    if (gtid == 0) {
        // :144-148: QC[0..3] = (0, QC1 + QC0, 0, 0).  Nothing else in this kernel reads those words.
        const uint32_t q0 = p.qc[QC_NEWPATH], q1 = p.qc[QC_LASTPATHCNT];
        p.qc[QC_NEWPATH] = 0; p.qc[QC_LASTPATHCNT] = q0 + q1; p.qc[QC_MATUE4] = 0; p.qc[QC_MATGLASS] = 0;
    }
    PackedStack stk; stk.lds = s_stack + threadIdx.x; stk.ovf = p.ovfStack + gtid; stk.ovfStride = p.ovfStride; stk.ptr = 0;
    TravCount tc = { 0, 0, 0 }; uint32_t rays = 0, wIn = 0, wTr = 0;
    const TravScene& ts = p.trav;
    const uint32_t* qSh = p.queues + (size_t)Q_SHADOW_RAY * p.P;
    uint32_t next = (gtid >> 6) * p.raysPerWave;
    const uint32_t end = (next + p.raysPerWave < count) ? next + p.raysPerWave : count;
    if (next >= end) return;

    bool haveRay = false, occluded = false;
    uint32_t index = 0;
    f3 o = mk3(0, 0, 0), d = mk3(0, 0, 1), invdir = mk3(0, 0, 0);
    float lightDistance = 0.0f;
    int cur = kDone;

    for (;;) {
        const bool idle = (cur == kDone);
        const unsigned long long idleMask = __ballot(idle);
        const int nIdle = __popcll(idleMask);
        if (nIdle == 64 || (nIdle >= REFILL && next < end)) {
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
                    stk.ptr = 0;
                    cur = (ray_box(ts.rootMin[0], ts.rootMin[1], ts.rootMin[2], ts.rootMax[0], ts.rootMax[1], ts.rootMax[2], o, invdir) > 0.0f) ? ts.rootDesc : kDone;
                    if (STATS) { if (cur < 0 && cur != kDone) tc.leaves++; }
                }
            }
            if (nIdle == 64 && next >= end) break;
            next = (next + (uint32_t)nIdle < end) ? next + (uint32_t)nIdle : end;
        }
        bool stepped = false;
#pragma unroll
        for (int rep = 0; rep < REPS; rep++) {
            if (cur >= 0) {
                if (STATS) { tc.inner++; if (prefix_rank(__ballot(1)) == 0) wIn++; if (EXT_KERNEL) atomicAdd(&p.stats->extDepthHist[ts.nodes[cur].d[2] & 31], 1ull); }
                cur = inner_step(ts, cur, o, invdir, stk, p.stats);
                if (STATS) { if (cur < 0 && cur != kDone) tc.leaves++; }
                stepped = true;
            }
        }
        if (!stepped && cur != kDone) {
            if (STATS) { tc.tris++; if (prefix_rank(__ballot(1)) == 0) wTr++; }
            const int i = ~cur;
            float t = 0.0f, u = 0.0f, v = 0.0f; bool last;
            if (tri_test(ts.tris, i, o, d, t, u, v, last)) {
                // shadowRayCast.hlsl:41-45,89: t in (1e-8, 1e8) and |d t| < lightDistance => occluded, stop
                if (t > kEpsilon && t < 1.0f / kEpsilon && length3(d * t) < lightDistance) occluded = true;
            }
            cur = occluded ? kDone : (last ? stk.pop() : ~(i + 1));
            if (STATS) { if (!occluded && last && cur < 0 && cur != kDone) tc.leaves++; }
        }
    }
    if (STATS) { flush_counts(p.stats, tc, rays, false); flush_wave_iters(p.stats, wIn, wTr, false); }
}

// ------------------------------------------------------------------------------------------------ top-of-tree-in-LDS variants
// Measured (tools/micro/gather64.hip): MI355X fetches ~105 G random 64-byte records/s from a 25 MB table when every lane
// wants a different record, and 3x that when the four lanes of a quad want the same one; the interleaved kernels above
// already run at ~70 G records/s, i.e. they are bound by the vector-memory pipeline, not by HBM or VALU.  48 % of all
// inner-node visits go to the first 8 levels of the tree (235 nodes on the bench scene), so those nodes are kept in LDS
// (16 KB per workgroup, read with ds_read_b128): half of the node fetches leave the vector-memory pipeline.
constexpr int kTopStack = 16; // LDS stack entries per lane (deeper entries: global overflow array)

struct TopStack {
    int* lds; int* ovf; uint32_t ovfStride; uint32_t ptr;
    __device__ __forceinline__ void push(int v, DevStats* st)
    {
        if (ptr < kTopStack) lds[ptr * kTravBlock] = v;
        else if (ptr < kMaxStack) ovf[(size_t)(ptr - kTopStack) * ovfStride] = v;
        else atomicOr(&st->stackOverflow, 1u);
        ptr++;
    }
    __device__ __forceinline__ int pop()
    {
        if (ptr == 0) return kDone;
        --ptr;
        if (ptr < kTopStack) return lds[ptr * kTravBlock];
        if (ptr < kMaxStack) return ovf[(size_t)(ptr - kTopStack) * ovfStride];
        return kDone;
    }
};

__device__ __forceinline__ void load_top_tree(const TravScene& ts, float4* s_top)
{
    const float4* src = reinterpret_cast<const float4*>(ts.nodes);
    for (uint32_t k = threadIdx.x; k < ts.topCount * 4u; k += kTravBlock) s_top[k] = src[k];
    __syncthreads();
}

__device__ __forceinline__ int inner_step_top(const TravScene& ts, const float4* s_top, int cur, f3 o, f3 invdir, TopStack& stk, DevStats* dst)
{
    float4 a, b, c; int4 d;
    if ((uint32_t)cur < ts.topCount) {
        const float4* n = s_top + cur * 4;
        a = n[0]; b = n[1]; c = n[2]; d = *reinterpret_cast<const int4*>(n + 3);
    } else {
        const float4* n = reinterpret_cast<const float4*>(ts.nodes + cur);
        a = n[0]; b = n[1]; c = n[2]; d = *reinterpret_cast<const int4*>(n + 3);
    }
    const float leftHit = ray_box(a.x, a.y, a.z, a.w, b.x, b.y, o, invdir);
    const float rightHit = ray_box(b.z, b.w, c.x, c.y, c.z, c.w, o, invdir);
    if (leftHit > 0.0f && rightHit > 0.0f) {
        if (leftHit > rightHit) { stk.push(d.x, dst); return d.y; }
        stk.push(d.y, dst); return d.x;
    }
    if (leftHit > 0.0f) return d.x;
    if (rightHit > 0.0f) return d.y;
    return stk.pop();
}

template <bool STATS, int REPS, int REFILL>
__global__ __launch_bounds__(kTravBlock) void k_extend_t(RenderParams p)
{
    constexpr bool EXT_KERNEL = true;
    __shared__ int s_stack[kTopStack * kTravBlock];
    __shared__ float4 s_top[kTopTreeNodes * 4];
    load_top_tree(p.trav, s_top);
    const uint32_t gtid = blockIdx.x * kTravBlock + threadIdx.x;
    TopStack stk; stk.lds = s_stack + threadIdx.x; stk.ovf = p.ovfStack + gtid; stk.ovfStride = p.ovfStride; stk.ptr = 0;
    TravCount tc = { 0, 0, 0 }; uint32_t rays = 0, wIn = 0, wTr = 0;
    const TravScene& ts = p.trav;
    const uint32_t count = p.qc[QC_EXT_COUNT];
    const uint32_t* qExt = p.queues + (size_t)Q_EXT_RAY * p.P;
    uint32_t next = (gtid >> 6) * p.raysPerWave;
    const uint32_t end = (next + p.raysPerWave < count) ? next + p.raysPerWave : count;
    if (next >= end) return; // wave-uniform

    bool haveRay = false;
    uint32_t index = 0;
    f3 o = mk3(0, 0, 0), d = mk3(0, 0, 1), invdir = mk3(0, 0, 0);
    float distance = kFltMax, hu = 0.0f, hv = 0.0f;
    int hitRef = -1;
    int cur = kDone;

    for (;;) {
        const bool idle = (cur == kDone);
        const unsigned long long idleMask = __ballot(idle);
        const int nIdle = __popcll(idleMask);
        if (nIdle == 64 || (nIdle >= REFILL && next < end)) { // wave-uniform
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
                        stk.ptr = 0;
                        cur = (ray_box(ts.rootMin[0], ts.rootMin[1], ts.rootMin[2], ts.rootMax[0], ts.rootMax[1], ts.rootMax[2], o, invdir) > 0.0f) ? ts.rootDesc : kDone;
                        if (STATS) { if (cur < 0 && cur != kDone) tc.leaves++; }
                    }
                }
            }
            if (nIdle == 64 && next >= end) break; // nothing in flight and the chunk is exhausted (wave-uniform)
            next = (next + (uint32_t)nIdle < end) ? next + (uint32_t)nIdle : end;
        }
        bool stepped = false;
#pragma unroll
        for (int rep = 0; rep < REPS; rep++) {
            if (cur >= 0) {
                if (STATS) { tc.inner++; if (prefix_rank(__ballot(1)) == 0) wIn++; if (EXT_KERNEL) atomicAdd(&p.stats->extDepthHist[ts.nodes[cur].d[2] & 31], 1ull); }
                cur = inner_step_top(ts, s_top, cur, o, invdir, stk, p.stats);
                if (STATS) { if (cur < 0 && cur != kDone) tc.leaves++; }
                stepped = true;
            }
        }
        if (!stepped && cur != kDone) {
            if (STATS) { tc.tris++; if (prefix_rank(__ballot(1)) == 0) wTr++; }
            const int i = ~cur;
            float t = 0.0f, u = 0.0f, v = 0.0f; bool last;
            if (tri_test(ts.tris, i, o, d, t, u, v, last)) {
                if (t >= 0.0f && t < distance) { distance = t; hitRef = i; hu = u; hv = v; } // extensionRayCast.hlsl:64-74
            }
            cur = last ? stk.pop() : ~(i + 1);
            if (STATS) { if (last && cur < 0 && cur != kDone) tc.leaves++; }
        }
    }
    if (STATS) { flush_counts(p.stats, tc, rays, true); flush_wave_iters(p.stats, wIn, wTr, true); }
}

template <bool STATS, int REPS, int REFILL>
__global__ __launch_bounds__(kTravBlock) void k_shadow_t(RenderParams p)
{
    constexpr bool EXT_KERNEL = false;
    __shared__ int s_stack[kTopStack * kTravBlock];
    __shared__ float4 s_top[kTopTreeNodes * 4];
    load_top_tree(p.trav, s_top);
    const uint32_t gtid = blockIdx.x * kTravBlock + threadIdx.x;
    const uint32_t count = p.qc[QC_SHADOWRAY];                               // shadowRayCast.hlsl:151
    __syncthreads();
    if (gtid == 0) {
        // :144-148: QC[0..3] = (0, QC1 + QC0, 0, 0).  Nothing else in this kernel reads those words.
        const uint32_t q0 = p.qc[QC_NEWPATH], q1 = p.qc[QC_LASTPATHCNT];
        p.qc[QC_NEWPATH] = 0; p.qc[QC_LASTPATHCNT] = q0 + q1; p.qc[QC_MATUE4] = 0; p.qc[QC_MATGLASS] = 0;
    }
    TopStack stk; stk.lds = s_stack + threadIdx.x; stk.ovf = p.ovfStack + gtid; stk.ovfStride = p.ovfStride; stk.ptr = 0;
    TravCount tc = { 0, 0, 0 }; uint32_t rays = 0, wIn = 0, wTr = 0;
    const TravScene& ts = p.trav;
    const uint32_t* qSh = p.queues + (size_t)Q_SHADOW_RAY * p.P;
    uint32_t next = (gtid >> 6) * p.raysPerWave;
    const uint32_t end = (next + p.raysPerWave < count) ? next + p.raysPerWave : count;
    if (next >= end) return;

    bool haveRay = false, occluded = false;
    uint32_t index = 0;
    f3 o = mk3(0, 0, 0), d = mk3(0, 0, 1), invdir = mk3(0, 0, 0);
    float lightDistance = 0.0f;
    int cur = kDone;

    for (;;) {
        const bool idle = (cur == kDone);
        const unsigned long long idleMask = __ballot(idle);
        const int nIdle = __popcll(idleMask);
        if (nIdle == 64 || (nIdle >= REFILL && next < end)) {
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
                    stk.ptr = 0;
                    cur = (ray_box(ts.rootMin[0], ts.rootMin[1], ts.rootMin[2], ts.rootMax[0], ts.rootMax[1], ts.rootMax[2], o, invdir) > 0.0f) ? ts.rootDesc : kDone;
                    if (STATS) { if (cur < 0 && cur != kDone) tc.leaves++; }
                }
            }
            if (nIdle == 64 && next >= end) break;
            next = (next + (uint32_t)nIdle < end) ? next + (uint32_t)nIdle : end;
        }
        bool stepped = false;
#pragma unroll
        for (int rep = 0; rep < REPS; rep++) {
            if (cur >= 0) {
                if (STATS) { tc.inner++; if (prefix_rank(__ballot(1)) == 0) wIn++; if (EXT_KERNEL) atomicAdd(&p.stats->extDepthHist[ts.nodes[cur].d[2] & 31], 1ull); }
                cur = inner_step_top(ts, s_top, cur, o, invdir, stk, p.stats);
                if (STATS) { if (cur < 0 && cur != kDone) tc.leaves++; }
                stepped = true;
            }
        }
        if (!stepped && cur != kDone) {
            if (STATS) { tc.tris++; if (prefix_rank(__ballot(1)) == 0) wTr++; }
            const int i = ~cur;
            float t = 0.0f, u = 0.0f, v = 0.0f; bool last;
            if (tri_test(ts.tris, i, o, d, t, u, v, last)) {
                // shadowRayCast.hlsl:41-45,89: t in (1e-8, 1e8) and |d t| < lightDistance => occluded, stop
                if (t > kEpsilon && t < 1.0f / kEpsilon && length3(d * t) < lightDistance) occluded = true;
            }
            cur = occluded ? kDone : (last ? stk.pop() : ~(i + 1));
            if (STATS) { if (!occluded && last && cur < 0 && cur != kDone) tc.leaves++; }
        }
    }
    if (STATS) { flush_counts(p.stats, tc, rays, false); flush_wave_iters(p.stats, wIn, wTr, false); }
}

// ------------------------------------------------------------------------------------------------ cooperative-fetch variants
// Measured on MI355X: the kernels above are bound by the number of per-lane vector-memory accesses (two redundant L1-hit
// dword loads per inner step cost +39 % time): a 64-byte node fetched by one lane is four 16-byte accesses to four different
// places of the texture-address pipeline.  Here every lane needs exactly ONE 64-byte record per step (inner node or triangle,
// same array), and the four lanes of a quad fetch the four quarters of one record with a single `global_load_lds_dwordx4`
// (LDS-DMA, per-lane source address): one instruction = 16 whole records instead of 64 quarter records.  Four such
// instructions bring the records of all 64 lanes into a per-wave LDS staging area, from which each lane reads its own
// record with four ds_read_b128.  Arithmetic and visit order are unchanged.
constexpr int kCoopBlock = 128;                 // 2 waves per workgroup (LDS: stage + stacks per wave)
constexpr int kCoopStack = 16;                  // LDS stack entries per lane; deeper entries go to the global overflow array
constexpr int kStageRegion = 1024 + 64;         // bytes per staging region (+64: the four regions start in different bank quarters)
constexpr int kStageBytes = 4 * kStageRegion;

struct CoopStack {
    int* lds; int* ovf; uint32_t ovfStride; uint32_t ptr;
    __device__ __forceinline__ void push(int v, DevStats* st)
    {
        if (ptr < kCoopStack) lds[ptr * kCoopBlock] = v;
        else if (ptr < kMaxStack) ovf[(size_t)(ptr - kCoopStack) * ovfStride] = v;
        else atomicOr(&st->stackOverflow, 1u);
        ptr++;
    }
    __device__ __forceinline__ int pop()
    {
        if (ptr == 0) return kDone;
        --ptr;
        if (ptr < kCoopStack) return lds[ptr * kCoopBlock];
        if (ptr < kMaxStack) return ovf[(size_t)(ptr - kCoopStack) * ovfStride];
        return kDone;
    }
};

template <int J>
__device__ __forceinline__ int quad_bcast(int v) { return __builtin_amdgcn_update_dpp(0, v, J * 0x55, 0xF, 0xF, true); } // quad_perm:[J,J,J,J]

// all 64 lanes call this with full EXEC: fetches recs[recIndex(lane)] of every lane into `stage` and returns the lane's record
__device__ __forceinline__ void coop_fetch(const TravScene& ts, int cur, char* stage, float4& q0, float4& q1, float4& q2, float4& q3)
{
    typedef __attribute__((address_space(3))) void lds_void;
    typedef const __attribute__((address_space(1))) void glb_void;
    const uint32_t lane = threadIdx.x & 63u;
    // record index: inner node cur, triangle record triBase + ~cur, idle lanes fetch record 0 (any valid address)
    const uint32_t rec = (cur >= 0) ? (uint32_t)cur : (cur == kDone ? 0u : ts.triBase + (uint32_t)~cur);
    const char* base = reinterpret_cast<const char*>(ts.recs) + 16u * (lane & 3u);
    const uint32_t r0 = (uint32_t)quad_bcast<0>((int)rec), r1 = (uint32_t)quad_bcast<1>((int)rec), r2 = (uint32_t)quad_bcast<2>((int)rec), r3 = (uint32_t)quad_bcast<3>((int)rec);
    __builtin_amdgcn_global_load_lds((glb_void*)(base + (size_t)r0 * 64u), (lds_void*)(stage + 0 * kStageRegion), 16, 0, 0);
    __builtin_amdgcn_global_load_lds((glb_void*)(base + (size_t)r1 * 64u), (lds_void*)(stage + 1 * kStageRegion), 16, 0, 0);
    __builtin_amdgcn_global_load_lds((glb_void*)(base + (size_t)r2 * 64u), (lds_void*)(stage + 2 * kStageRegion), 16, 0, 0);
    __builtin_amdgcn_global_load_lds((glb_void*)(base + (size_t)r3 * 64u), (lds_void*)(stage + 3 * kStageRegion), 16, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    // instruction j brought the record of lane 4g+j to region j, bytes [64g, 64g+64): lane 4g+i reads region i
    const float4* mine = reinterpret_cast<const float4*>(stage + (lane & 3u) * kStageRegion + (lane >> 2) * 64u);
    q0 = mine[0]; q1 = mine[1]; q2 = mine[2]; q3 = mine[3];
}

__device__ __forceinline__ int inner_step_regs(float4 a, float4 b, float4 c, float4 dq, f3 o, f3 invdir, CoopStack& stk, DevStats* dst)
{
    const int dl = __builtin_bit_cast(int, dq.x), dr = __builtin_bit_cast(int, dq.y);
    const float leftHit = ray_box(a.x, a.y, a.z, a.w, b.x, b.y, o, invdir);
    const float rightHit = ray_box(b.z, b.w, c.x, c.y, c.z, c.w, o, invdir);
    if (leftHit > 0.0f && rightHit > 0.0f) {
        if (leftHit > rightHit) { stk.push(dl, dst); return dr; }
        stk.push(dr, dst); return dl;
    }
    if (leftHit > 0.0f) return dl;
    if (rightHit > 0.0f) return dr;
    return stk.pop();
}

__device__ __forceinline__ bool tri_test_regs(float4 r0, float4 r1, float4 r2, f3 o, f3 d, float& t, float& u, float& v)
{
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

template <bool STATS, int REFILL>
__global__ __launch_bounds__(kCoopBlock) void k_extend_c(RenderParams p)
{
    __shared__ int s_stack[kCoopStack * kCoopBlock];
    __shared__ __attribute__((aligned(64))) char s_stage[(kCoopBlock / 64) * kStageBytes];
    const uint32_t gtid = blockIdx.x * kCoopBlock + threadIdx.x;
    CoopStack stk; stk.lds = s_stack + threadIdx.x; stk.ovf = p.ovfStack + gtid; stk.ovfStride = p.ovfStride; stk.ptr = 0;
    char* stage = s_stage + (threadIdx.x >> 6) * kStageBytes;
    TravCount tc = { 0, 0, 0 }; uint32_t rays = 0, wIn = 0, wTr = 0;
    const TravScene& ts = p.trav;
    const uint32_t count = p.qc[QC_EXT_COUNT];
    const uint32_t* qExt = p.queues + (size_t)Q_EXT_RAY * p.P;
    uint32_t next = (gtid >> 6) * p.raysPerWave;
    const uint32_t end = (next + p.raysPerWave < count) ? next + p.raysPerWave : count;
    if (next >= end) return; // wave-uniform

    bool haveRay = false;
    uint32_t index = 0;
    f3 o = mk3(0, 0, 0), d = mk3(0, 0, 1), invdir = mk3(0, 0, 0);
    float distance = kFltMax, hu = 0.0f, hv = 0.0f;
    int hitRef = -1;
    int cur = kDone;

    for (;;) {
        const bool idle = (cur == kDone);
        const unsigned long long idleMask = __ballot(idle);
        const int nIdle = __popcll(idleMask);
        if (nIdle == 64 || (nIdle >= REFILL && next < end)) { // wave-uniform
            if (idle) {
                if (haveRay) {
                    // finish the ray: extensionRayCast.hlsl:218-232
                    if (distance < kFltMax) {
                        st3(p, F_SP_X, index, o + d * distance);
                        st3(p, F_BARY_X, index, mk3(1.0f - hu - hv, hu, hv));
                        const int4 T = *reinterpret_cast<const int4*>(&ts.recs[ts.triBase + (uint32_t)hitRef].q[12]);
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
                        stk.ptr = 0;
                        cur = (ray_box(ts.rootMin[0], ts.rootMin[1], ts.rootMin[2], ts.rootMax[0], ts.rootMax[1], ts.rootMax[2], o, invdir) > 0.0f) ? ts.rootDesc : kDone;
                        if (STATS) { if (cur < 0 && cur != kDone) tc.leaves++; }
                    }
                }
            }
            if (nIdle == 64 && next >= end) break; // nothing in flight and the chunk is exhausted (wave-uniform)
            next = (next + (uint32_t)nIdle < end) ? next + (uint32_t)nIdle : end;
        }
        float4 q0, q1, q2, q3;
        coop_fetch(ts, cur, stage, q0, q1, q2, q3);
        if (cur >= 0) {
            if (STATS) { tc.inner++; if (prefix_rank(__ballot(1)) == 0) wIn++; }
            cur = inner_step_regs(q0, q1, q2, q3, o, invdir, stk, p.stats);
            if (STATS) { if (cur < 0 && cur != kDone) tc.leaves++; }
        } else if (cur != kDone) {
            if (STATS) { tc.tris++; if (prefix_rank(__ballot(1)) == 0) wTr++; }
            const int i = ~cur;
            const bool last = __builtin_bit_cast(uint32_t, q2.y) != 0u;
            float t = 0.0f, u = 0.0f, v = 0.0f;
            if (tri_test_regs(q0, q1, q2, o, d, t, u, v)) {
                if (t >= 0.0f && t < distance) { distance = t; hitRef = i; hu = u; hv = v; } // extensionRayCast.hlsl:64-74
            }
            cur = last ? stk.pop() : ~(i + 1);
            if (STATS) { if (last && cur < 0 && cur != kDone) tc.leaves++; }
        }
    }
    if (STATS) { flush_counts(p.stats, tc, rays, true); flush_wave_iters(p.stats, wIn, wTr, true); }
}

template <bool STATS, int REFILL>
__global__ __launch_bounds__(kCoopBlock) void k_shadow_c(RenderParams p)
{
    __shared__ int s_stack[kCoopStack * kCoopBlock];
    __shared__ __attribute__((aligned(64))) char s_stage[(kCoopBlock / 64) * kStageBytes];
    const uint32_t gtid = blockIdx.x * kCoopBlock + threadIdx.x;
    const uint32_t count = p.qc[QC_SHADOWRAY];                               // shadowRayCast.hlsl:151
    __syncthreads();
    if (gtid == 0) {
        // :144-148: QC[0..3] = (0, QC1 + QC0, 0, 0).  Nothing else in this kernel reads those words.
        const uint32_t q0c = p.qc[QC_NEWPATH], q1c = p.qc[QC_LASTPATHCNT];
        p.qc[QC_NEWPATH] = 0; p.qc[QC_LASTPATHCNT] = q0c + q1c; p.qc[QC_MATUE4] = 0; p.qc[QC_MATGLASS] = 0;
    }
    CoopStack stk; stk.lds = s_stack + threadIdx.x; stk.ovf = p.ovfStack + gtid; stk.ovfStride = p.ovfStride; stk.ptr = 0;
    char* stage = s_stage + (threadIdx.x >> 6) * kStageBytes;
    TravCount tc = { 0, 0, 0 }; uint32_t rays = 0, wIn = 0, wTr = 0;
    const TravScene& ts = p.trav;
    const uint32_t* qSh = p.queues + (size_t)Q_SHADOW_RAY * p.P;
    uint32_t next = (gtid >> 6) * p.raysPerWave;
    const uint32_t end = (next + p.raysPerWave < count) ? next + p.raysPerWave : count;
    if (next >= end) return;

    bool haveRay = false, occluded = false;
    uint32_t index = 0;
    f3 o = mk3(0, 0, 0), d = mk3(0, 0, 1), invdir = mk3(0, 0, 0);
    float lightDistance = 0.0f;
    int cur = kDone;

    for (;;) {
        const bool idle = (cur == kDone);
        const unsigned long long idleMask = __ballot(idle);
        const int nIdle = __popcll(idleMask);
        if (nIdle == 64 || (nIdle >= REFILL && next < end)) {
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
                    stk.ptr = 0;
                    cur = (ray_box(ts.rootMin[0], ts.rootMin[1], ts.rootMin[2], ts.rootMax[0], ts.rootMax[1], ts.rootMax[2], o, invdir) > 0.0f) ? ts.rootDesc : kDone;
                    if (STATS) { if (cur < 0 && cur != kDone) tc.leaves++; }
                }
            }
            if (nIdle == 64 && next >= end) break;
            next = (next + (uint32_t)nIdle < end) ? next + (uint32_t)nIdle : end;
        }
        float4 q0, q1, q2, q3;
        coop_fetch(ts, cur, stage, q0, q1, q2, q3);
        if (cur >= 0) {
            if (STATS) { tc.inner++; if (prefix_rank(__ballot(1)) == 0) wIn++; }
            cur = inner_step_regs(q0, q1, q2, q3, o, invdir, stk, p.stats);
            if (STATS) { if (cur < 0 && cur != kDone) tc.leaves++; }
        } else if (cur != kDone) {
            if (STATS) { tc.tris++; if (prefix_rank(__ballot(1)) == 0) wTr++; }
            const int i = ~cur;
            const bool last = __builtin_bit_cast(uint32_t, q2.y) != 0u;
            float t = 0.0f, u = 0.0f, v = 0.0f;
            if (tri_test_regs(q0, q1, q2, o, d, t, u, v)) {
                // shadowRayCast.hlsl:41-45,89: t in (1e-8, 1e8) and |d t| < lightDistance => occluded, stop
                if (t > kEpsilon && t < 1.0f / kEpsilon && length3(d * t) < lightDistance) occluded = true;
            }
            cur = occluded ? kDone : (last ? stk.pop() : ~(i + 1));
            if (STATS) { if (!occluded && last && cur < 0 && cur != kDone) tc.leaves++; }
        }
    }
    if (STATS) { flush_counts(p.stats, tc, rays, false); flush_wave_iters(p.stats, wIn, wTr, false); }
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
                        finish_extension_ray(p, index, o, d, distance, hu, hv, hitRef);
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
    __shared__ int s_stack[kDefLdsStack<OVF> * kDefBlock];
    __shared__ int s_fifo[kFifo * kDefBlock];
    __shared__ float4 s_top[TOP ? kDefLdsTop<OVF> * 4 : 4];
    if (TOP) {
        const float4* src = reinterpret_cast<const float4*>(p.trav.nodes);
        for (uint32_t k = threadIdx.x; k < top_count<OVF>(p.trav) * 4u; k += kDefBlock) s_top[k] = src[k];
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
                    finish_extension_ray(p, rIndex, rO, rD, dist, rHu, rHv, rHit);
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
    __shared__ int s_stack[kDefLdsStack<OVF> * kDefBlock];
    __shared__ int s_fifo[kFifo * kDefBlock];
    __shared__ float4 s_top[TOP ? kDefLdsTop<OVF> * 4 : 4];
    if (TOP) {
        const float4* src = reinterpret_cast<const float4*>(p.trav.nodes);
        for (uint32_t k = threadIdx.x; k < top_count<OVF>(p.trav) * 4u; k += kDefBlock) s_top[k] = src[k];
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
void launch_extend_variant(const RenderParams& p, uint32_t blocks, bool stats, int mode, hipStream_t s)
{
    if (mode >= 50) {   // pipeN: three ray slots per lane
        const uint32_t pb = p.travGridBlocks;
        const bool ovf = p.trav.maxDepth + 2 > (uint32_t)kDefStack;
        switch (mode - 50) {
        case 0: GMUPT_DEF_LAUNCH(k_extend_e, true, 4, 4); break;
        case 1: GMUPT_DEF_LAUNCH(k_extend_e, true, 3, 4); break;
        default: GMUPT_DEF_LAUNCH(k_extend_e, true, 2, 4); break;
        }
    } else if (mode == 30) {
        const uint32_t pb = (p.L + p.raysPerWave * (kTravBlock / 64) - 1) / (p.raysPerWave * (kTravBlock / 64));
        if (stats) hipLaunchKernelGGL((k_extend_t<true, 2, 20>), dim3(pb), dim3(kTravBlock), 0, s, p);
        else hipLaunchKernelGGL((k_extend_t<false, 2, 20>), dim3(pb), dim3(kTravBlock), 0, s, p);
    } else if (mode == 20) {
        const uint32_t pb = (p.L + p.raysPerWave * (kCoopBlock / 64) - 1) / (p.raysPerWave * (kCoopBlock / 64));
        if (stats) hipLaunchKernelGGL((k_extend_c<true, 16>), dim3(pb), dim3(kCoopBlock), 0, s, p);
        else hipLaunchKernelGGL((k_extend_c<false, 16>), dim3(pb), dim3(kCoopBlock), 0, s, p);
    } else if (mode >= 3) {
        const uint32_t pb = (p.L + p.raysPerWave * (kTravBlock / 64) - 1) / (p.raysPerWave * (kTravBlock / 64));
        const int v = mode - 3;
#define GMUPT_EXT_CASE(N, REPS, REFILL) case N: if (stats) hipLaunchKernelGGL((k_extend_i<true, REPS, REFILL>), dim3(pb), dim3(kTravBlock), 0, s, p); \
                                                else hipLaunchKernelGGL((k_extend_i<false, REPS, REFILL>), dim3(pb), dim3(kTravBlock), 0, s, p); break;
        switch (v) { GMUPT_EXT_CASE(0, 1, 20) GMUPT_EXT_CASE(1, 2, 20) GMUPT_EXT_CASE(2, 3, 20) GMUPT_EXT_CASE(3, 1, 8) GMUPT_EXT_CASE(4, 2, 8) GMUPT_EXT_CASE(5, 2, 12) default: break; }
    } else if (mode == 0) {
        const uint32_t pb = (p.L + kRaysPerWave * (kTravBlock / 64) - 1) / (kRaysPerWave * (kTravBlock / 64));
        if (stats) hipLaunchKernelGGL(k_extend_p<true>, dim3(pb), dim3(kTravBlock), 0, s, p);
        else hipLaunchKernelGGL(k_extend_p<false>, dim3(pb), dim3(kTravBlock), 0, s, p);
    } else if (mode == 1) {
        if (stats) hipLaunchKernelGGL(k_extend_ref<true>, dim3(blocks), dim3(kTravBlock), 0, s, p);
        else hipLaunchKernelGGL(k_extend_ref<false>, dim3(blocks), dim3(kTravBlock), 0, s, p);
    } else {
        if (stats) hipLaunchKernelGGL(k_extend<true>, dim3(blocks), dim3(kTravBlock), 0, s, p);
        else hipLaunchKernelGGL(k_extend<false>, dim3(blocks), dim3(kTravBlock), 0, s, p);
    }
}
void launch_shadow_variant(const RenderParams& p, uint32_t blocks, bool stats, int mode, hipStream_t s)
{
    if (mode >= 50) {   // pipeN: three ray slots per lane
        const uint32_t pb = p.travGridBlocks;
        const bool ovf = p.trav.maxDepth + 2 > (uint32_t)kDefStack;
        switch (mode - 50) {
        case 0: GMUPT_DEF_LAUNCH(k_shadow_e, true, 4, 4); break;
        case 1: GMUPT_DEF_LAUNCH(k_shadow_e, true, 3, 4); break;
        default: GMUPT_DEF_LAUNCH(k_shadow_e, true, 2, 4); break;
        }
    } else if (mode == 30) {
        const uint32_t pb = (p.L + p.raysPerWave * (kTravBlock / 64) - 1) / (p.raysPerWave * (kTravBlock / 64));
        if (stats) hipLaunchKernelGGL((k_shadow_t<true, 2, 20>), dim3(pb), dim3(kTravBlock), 0, s, p);
        else hipLaunchKernelGGL((k_shadow_t<false, 2, 20>), dim3(pb), dim3(kTravBlock), 0, s, p);
    } else if (mode == 20) {
        const uint32_t pb = (p.L + p.raysPerWave * (kCoopBlock / 64) - 1) / (p.raysPerWave * (kCoopBlock / 64));
        if (stats) hipLaunchKernelGGL((k_shadow_c<true, 16>), dim3(pb), dim3(kCoopBlock), 0, s, p);
        else hipLaunchKernelGGL((k_shadow_c<false, 16>), dim3(pb), dim3(kCoopBlock), 0, s, p);
    } else if (mode >= 3) {
        const uint32_t pb = (p.L + p.raysPerWave * (kTravBlock / 64) - 1) / (p.raysPerWave * (kTravBlock / 64));
        const int v = mode - 3;
#define GMUPT_SH_CASE(N, REPS, REFILL) case N: if (stats) hipLaunchKernelGGL((k_shadow_i<true, REPS, REFILL>), dim3(pb), dim3(kTravBlock), 0, s, p); \
                                               else hipLaunchKernelGGL((k_shadow_i<false, REPS, REFILL>), dim3(pb), dim3(kTravBlock), 0, s, p); break;
        switch (v) { GMUPT_SH_CASE(0, 1, 20) GMUPT_SH_CASE(1, 2, 20) GMUPT_SH_CASE(2, 3, 20) GMUPT_SH_CASE(3, 1, 8) GMUPT_SH_CASE(4, 2, 8) GMUPT_SH_CASE(5, 2, 12) default: break; }
    } else if (mode == 0) {
        const uint32_t pb = (p.L + kRaysPerWave * (kTravBlock / 64) - 1) / (kRaysPerWave * (kTravBlock / 64));
        if (stats) hipLaunchKernelGGL(k_shadow_p<true>, dim3(pb), dim3(kTravBlock), 0, s, p);
        else hipLaunchKernelGGL(k_shadow_p<false>, dim3(pb), dim3(kTravBlock), 0, s, p);
    } else if (mode == 1) {
        if (stats) hipLaunchKernelGGL(k_shadow_ref<true>, dim3(blocks), dim3(kTravBlock), 0, s, p);
        else hipLaunchKernelGGL(k_shadow_ref<false>, dim3(blocks), dim3(kTravBlock), 0, s, p);
    } else {
        if (stats) hipLaunchKernelGGL(k_shadow<true>, dim3(blocks), dim3(kTravBlock), 0, s, p);
        else hipLaunchKernelGGL(k_shadow<false>, dim3(blocks), dim3(kTravBlock), 0, s, p);
    }
}
uint32_t variant_overflow_entries() { return kMaxStack + 1 - kCoopStack; } // sized for the variant with the smallest LDS stack


bool launch_cast_variant(const RenderParams& p, bool stats, int mode, hipStream_t s)
{
    (void)mode;
    const uint32_t pb = p.travGridBlocks;
    const bool ovf = p.trav.maxDepth + 2 > (uint32_t)kDefStack;
    GMUPT_DEF_LAUNCH(k_cast_m, true, 4, 4);
    return true;
}

} // namespace gmupt
#endif // GMUPT_VARIANTS
