// Ray-cast kernels for gfx950 (MI355X): extension (closest hit + light spheres) and shadow (any hit) -- the shipped versions.
// The rungs that led here (reference-layout, packed static, persistent while-while, interleaved, top-of-tree, cooperative LDS-DMA)
// live in pt_traverse_variants.hip, which is only part of the -DGMUPT_VARIANTS build (GMUPT_TRAVERSAL=ref|static|whilewhile|ififN|top|coop|pipeN|cast1|cast2).
//
// What must not change (it decides results): the slab test arithmetic and its "hit iff result > 0" rule with no pruning
// against the current closest hit (extensionRayCast.hlsl:79-94,132-159), the near-child-first visit order (closest-hit ties
// are resolved by visit order, `t < distance` strict, :64-74), the Moeller-Trumbore operation order (:38-77), and the
// shadow acceptance rule t in (1e-8, 1e8), |d t| < lightDistance (shadowRayCast.hlsl:16-47,88-91).
// What is free: memory layout, loop structure, scheduling of the triangle tests, and -- for the any-hit shadow ray -- the visit order.
#include "pt_traverse_deferred.hpp"

namespace gmupt {

#ifdef GMUPT_VARIANTS   // the A/B rungs of the traversal ladder are only part of the -DGMUPT_VARIANTS build (tests, tools/sweep*.sh)
void launch_extend_variant(const RenderParams& p, uint32_t blocks, bool stats, int mode, hipStream_t s);
void launch_shadow_variant(const RenderParams& p, uint32_t blocks, bool stats, int mode, hipStream_t s);
bool launch_cast_variant(const RenderParams& p, bool stats, int mode, hipStream_t s);
uint32_t variant_overflow_entries();
#endif

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
                    finish_extension_ray(p, index, o, d, distance, hu, hv, hitRef);
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

// k_cast_m with the two fetches of a loop step issued together: when the wave is in a triangle burst, every step first issues the node
// fetch of the walking lanes AND the triangle fetch of the lanes with a pending leaf, then does the slab tests and the triangle test.
// A burst therefore costs no memory round trips of its own (REPS per loop iteration instead of REPS + BURST); what is tested, and in which
// order per ray, is unchanged: leaves leave the per-lane FIFO in visit order.  BURST is unused here (one triangle record per step).
#ifdef GMUPT_CAST_WAVES_PER_EU
#define GMUPT_CAST_OCCUPANCY __attribute__((amdgpu_waves_per_eu(GMUPT_CAST_WAVES_PER_EU, GMUPT_CAST_WAVES_PER_EU)))
#else
#define GMUPT_CAST_OCCUPANCY
#endif
template <bool STATS, bool OVF, bool TOP, int REPS, int BURST>
__global__ __launch_bounds__(kDefBlock) GMUPT_CAST_OCCUPANCY void k_cast_f(RenderParams p)
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
    uint32_t next = 0, end = 0, lastBase = 0;
    uint32_t chunkBase = 0, qe0 = kQueueHole, qe1 = kQueueHole;  // the current chunk of queue entries, lane l holds entries l and 64 + l
    int phase = 0;
#ifndef GMUPT_DRAIN_TIMING
#define GMUPT_DRAIN_TIMING 0   // 1 (diagnostic build only): the plain kernel stamps its start, the moment a wave finds both queues empty, and its exit
#endif
    const unsigned long long tStart = (STATS || GMUPT_DRAIN_TIMING) ? wall_clock64() : 0ull;
    unsigned long long tDrain = 0ull; uint32_t drainIters = 0, drainBusy = 0;
    uint32_t rayInner = 0, census0 = 0, census1 = 0, census2 = 0, census3 = 0, topE = 0, topS = 0, helped = 0, nested = 0;

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
    // ---- the drain (both queues exhausted): a wave is as slow as its longest ray, so the lanes that have finished HELP the ones still
    // walking.  Boxes are never pruned, so a deferred subtree can be walked by another lane; the BOTTOM entry of a ray's stack is the
    // subtree the ray would visit last, hence its hits only count if they are strictly closer than everything the owner finds itself
    // (ties go to the earlier test, extensionRayCast.hlsl:64) -- and among helpers the later donated subtree is visited earlier.
    int owner = -1;               // >= 0: this lane walks a subtree donated by lane `owner` (same ray, own stack, own leaf FIFO)
    uint32_t bottom = 1;          // lowest live entry of this lane's stack (entries below it were donated; the slot under it holds the sentinel)
    uint32_t outstanding = 0;     // owner: helpers that have not reported yet
    uint32_t donations = 0;       // subtrees this lane has donated so far (a helper donates parts of its subtree in turn)
    uint32_t dno = 0;             // helper: the number of its donation at its owner
    float tT = kFltMax, uT = 0.0f, vT = 0.0f; int refT = -1; uint32_t dT = 0;   // owner: best helper result so far (by t, then by LATER donation)
    const uint32_t lane = threadIdx.x & 63u;

    // Watchdog: a persistent kernel must end whatever happens.  A wave of the bench scene runs ~150 loop iterations per launch, one of the
    // 10 M-triangle scene ~600; a wave that reaches the limit has met a bug in the hand-over protocol of the drain: it flags the launch
    // (GMUPT_STAT_CAST_ABORTED: results invalid) and leaves.  -DGMUPT_LOOP_DUMP also records the state of its lanes (debugging aid).
    uint32_t loopCount = 0;
    for (;;) {
        if (++loopCount > p.castLoopCap) {   // (2^20 by default; GMUPT_CAST_LOOP_CAP)
            if (lane == 0u) atomicOr(&p.stats->stackOverflow, 2u);
#ifdef GMUPT_LOOP_DUMP
            const unsigned long long mHave = __ballot(haveRay), mHelper = __ballot(haveRay && owner >= 0), mOut = __ballot(outstanding != 0u),
                                     mWalk = __ballot(cur >= 0), mLeaf = __ballot(cur < 0 && cur != kDone), mPend = __ballot(qCount > 0 || ti >= 0);
            if (lane == 0u && atomicAdd(&p.stats->castWaveEndHist[31], 1ull) == 0ull) {
                p.stats->castWaveEndHist[0] = mHave; p.stats->castWaveEndHist[1] = mHelper; p.stats->castWaveEndHist[2] = mOut;
                p.stats->castWaveEndHist[3] = mWalk; p.stats->castWaveEndHist[4] = mLeaf; p.stats->castWaveEndHist[5] = mPend;
                p.stats->castWaveEndHist[6] = (unsigned long long)phase; p.stats->castWaveEndHist[7] = (unsigned long long)blockIdx.x * 100ull + (threadIdx.x >> 6);
            }
            if (mHave >> lane & 1ull) { p.stats->castWaveEndHist[8 + (lane & 15u)] = ((unsigned long long)(uint32_t)owner << 32) | (outstanding << 16) | (stk.ptr << 8) | bottom; }
#endif
            break;
        }
        if (phase == 2) { // wave-uniform: drain service
            // (a) helpers that have finished their subtree report to their owner
            const bool reports = haveRay && owner >= 0 && outstanding == 0u && cur == kDone && qCount == 0 && ti < 0;
            if (reports) {   // what its own helpers found in the parts it gave away: later in visit order than its own walk, so only if strictly closer
                if (kind == 0) { if (refT >= 0 && tT < distance) { distance = tT; hu = uT; hv = vT; hitRef = refT; } }
                else if (refT >= 0) hitRef = refT;
            }
            unsigned long long fin = __ballot(reports);
            while (fin) {
                const int hl = __builtin_ctzll(fin); fin &= fin - 1ull;
                const int ol = __builtin_amdgcn_readlane(owner, hl);
                const float th = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, distance), hl));
                const float uh = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, hu), hl));
                const float vh = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, hv), hl));
                const int rh = __builtin_amdgcn_readlane(hitRef, hl);
                const uint32_t dh = (uint32_t)__builtin_amdgcn_readlane((int)dno, hl);
                if ((int)lane == ol) {
                    outstanding--;
                    if (rh >= 0) {
                        if (kind == 1) refT = rh;                                                       // any occluder decides a shadow ray
                        else if (th < tT || (th == tT && dh > dT)) { tT = th; uT = uh; vT = vh; refT = rh; dT = dh; }
                    }
                }
                if ((int)lane == hl) { haveRay = false; owner = -1; tT = kFltMax; refT = -1; dT = 0; }
            }
            // (b) owners that are done and have heard from all their helpers are written back now: their lanes become free
            const bool done = haveRay && owner < 0 && outstanding == 0u && cur == kDone && qCount == 0 && ti < 0;
            // (a lane whose walk has ended never gives: an occluded shadow ray leaves its stack behind, and those subtrees no longer matter)
            const unsigned long long wantHelp = __ballot(haveRay && cur != kDone && stk.ptr > bottom && bottom + 1u < (uint32_t)kDefLdsStack<OVF> && donations < 12u);
            if (wantHelp != 0ull && done) {
                if (kind == 0) {
                    if (refT >= 0 && tT < distance) { distance = tT; hu = uT; hv = vT; hitRef = refT; }
                    finish_extension_ray(p, index, o, d, distance, hu, hv, hitRef);
                } else stu(p, F_IN_SHADOW, index, (hitRef >= 0 || refT >= 0) ? 1u : 0u);
                haveRay = false; tT = kFltMax; refT = -1; dT = 0;
            }
            // (c) free lanes take the bottom stack entry of lanes that still have deferred subtrees: the k-th free lane pairs with the k-th
            //     donor, all pairs of the wave at once (the donors post their lane number to lane k, the takers fetch the ray from there)
            const unsigned long long freeLanes = __ballot(!haveRay);
            if (wantHelp != 0ull && freeLanes != 0ull) {
                const uint32_t nDonors = (uint32_t)__popcll(wantHelp), nFree = (uint32_t)__popcll(freeLanes);
                const uint32_t nPairs = nDonors < nFree ? nDonors : nFree;
                const bool isDonor = ((wantHelp >> lane) & 1ull) != 0ull;
                const uint32_t dRank = prefix_rank(wantHelp), fRank = prefix_rank(freeLanes);
                const bool gives = isDonor && dRank < nPairs, takes = !haveRay && fRank < nPairs;
                int node = 0;
                if (gives) {
                    int* slot = s_stack + bottom * kDefBlock + threadIdx.x;
                    node = *slot; *slot = kDone;           // the slot becomes the sentinel of what is left of the owner's stack
                    bottom++; outstanding++; donations++;
                    if (STATS) { helped++; if (owner >= 0) nested++; }   // nested: a helper gives a part of ITS subtree away
                }
                // lane k learns which lane the k-th donor is; a taker with rank k reads that number from lane k, then the ray from the donor
                const int donorOfRank = __builtin_amdgcn_ds_permute((int)((gives ? dRank : 63u) << 2), gives ? (int)lane : 0);
                const int src = __shfl(donorOfRank, takes ? (int)fRank : 0);
                const int sl = takes ? src : (int)lane;
                const int node2 = __shfl(node, sl), k2 = __shfl(kind, sl), i2 = __shfl((int)index, sl), n2 = __shfl((int)donations, sl);
                const float ox = __shfl(o.x, sl), oy = __shfl(o.y, sl), oz = __shfl(o.z, sl), dx = __shfl(d.x, sl), dy = __shfl(d.y, sl), dz = __shfl(d.z, sl);
                const float lim = __shfl(distance, sl);
                if (takes) {
                    haveRay = true; owner = src; kind = k2; index = (uint32_t)i2; dno = (uint32_t)n2; donations = 0;
                    o = mk3(ox, oy, oz); d = mk3(dx, dy, dz); invdir = mk3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
                    distance = k2 == 1 ? lim : kFltMax;     // shadow: the distance of the light; extension: no hit yet
                    hitRef = -1; hu = 0.0f; hv = 0.0f;
                    stk.reset(); bottom = 1; outstanding = 0; qHead = 0; qCount = 0; ti = -1;
                    tT = kFltMax; refT = -1; dT = 0;
                    cur = node2;
                }
            }
        }
        // (a helper, or a lane that still waits for its helpers, is not idle: it has a report to make or to receive)
        const bool idle = (cur == kDone) && (qCount == 0) && (ti < 0) && !(haveRay && (owner >= 0 || outstanding != 0u));
        const unsigned long long idleMask = __ballot(idle);
        const int nIdle = __popcll(idleMask);
        if (nIdle == 64 || (nIdle >= (int)p.tuneRefill && phase < 2)) { // wave-uniform
            while (next >= end && phase < 2) { // next chunk of the current queue, or the first one of the next queue
                uint32_t base = 0;
                const uint32_t count = phase == 0 ? countExt : countSh;
                // towards the end of the LAST queue the chunks shrink: the waves then run dry within half the time of each other
                // (the position is estimated from this wave's previous chunk: every other wave has taken about one since)
                const uint32_t gridWaves = gridDim.x * (uint32_t)(kDefBlock / 64);
#ifndef GMUPT_TAIL_CHUNKS
#define GMUPT_TAIL_CHUNKS 2
#endif
                const bool tail = GMUPT_TAIL_CHUNKS && phase == 1 && lastBase + (uint32_t)GMUPT_TAIL_CHUNKS * gridWaves * p.raysPerWave > count;
                const uint32_t req = tail ? (p.raysPerWave > 64u ? p.raysPerWave / 2u : p.raysPerWave) : p.raysPerWave;
                if ((threadIdx.x & 63) == 0) base = atomicAdd(&p.travCounters[phase], req);
                base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
                if (base < count) {
                    lastBase = base;
                    next = base; end = (base + req < count) ? base + req : count;
                    // the whole chunk of queue entries goes into two registers per lane (two coalesced loads): a refill then takes
                    // its entry from a neighbour's register instead of starting with a dependent queue read
                    chunkBase = base;
                    const uint32_t* q = phase == 0 ? qExt : qSh;
                    qe0 = (base + lane < end) ? q[base + lane] : kQueueHole;
                    qe1 = (base + 64u + lane < end) ? q[base + 64u + lane] : kQueueHole;
                }
                else { phase++; next = end = 0; lastBase = 0; }
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
                        // finish the extension ray: extensionRayCast.hlsl:218-232 (a helper's hit only counts if it is strictly closer)
                        if (refT >= 0 && tT < distance) { distance = tT; hu = uT; hv = vT; hitRef = refT; }
                        finish_extension_ray(p, index, o, d, distance, hu, hv, hitRef);
                    } else {
                        stu(p, F_IN_SHADOW, index, (hitRef >= 0 || refT >= 0) ? 1u : 0u);   // shadowRayCast.hlsl:167
                    }
                    haveRay = false; tT = kFltMax; refT = -1; dT = 0;
                }
                if (newRay) {
                    bottom = 1; donations = 0;
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

        if (GMUPT_DRAIN_TIMING && !STATS && phase == 2) { if (tDrain == 0ull) tDrain = wall_clock64(); drainIters++; }
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
            if (doNode) load_node_buf<TOP>(rNodes, s_top, top_count<OVF>(ts), cur, na, nb, nc, nd);
            if (burst && ti < 0 && qCount > 0) { ti = fifo[(qHead & (kFifo - 1)) * kDefBlock]; qHead++; qCount--; }
            const bool doTri = burst && ti >= 0;
            vec4f r0, r1; vec2f r2;                          // defined for the doTri lanes only
            if (doTri) tri_fetch_buf(rTris, ti, r0, r1, r2);
            // compute phase
            if (doNode) {
                if (STATS) { if (kind == 0) { tcE.inner++; rayInner++; } else tcS.inner++; if (prefix_rank(__ballot(1)) == 0) { if (phase == 0) wInE++; else wInS++; }
                             if (TOP && (uint32_t)cur < top_count<OVF>(ts)) { if (kind == 0) topE++; else topS++; } }
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
    if (GMUPT_DRAIN_TIMING && !STATS && (threadIdx.x & 63) == 0) {
        const unsigned long long tEnd = wall_clock64();
        atomicAdd(&p.stats->castDrainClocks, tDrain ? tEnd - tDrain : 0ull); atomicAdd(&p.stats->castDrainIters, (unsigned long long)drainIters);
        atomicAdd(&p.stats->castWaves, 1ull); atomicAdd(&p.stats->castWaveClocks, tEnd - tStart); atomicMax(&p.stats->castWaveClocksMax, tEnd - tStart);
        const unsigned long long life = tEnd - tStart, bucket = life / 4000ull;    // 40-us buckets of the wave lifetimes (all waves start together)
        atomicAdd(&p.stats->castWaveEndHist[bucket < 31ull ? bucket : 31ull], 1ull);
    }
    if (STATS) { flush_counts(p.stats, tcE, raysE, true); flush_wave_iters(p.stats, wInE, wTrE, true);
                 flush_counts(p.stats, tcS, raysS, false); flush_wave_iters(p.stats, wInS, wTrS, false);
                 flush_sum(&p.stats->extTopInner, topE); flush_sum(&p.stats->shTopInner, topS); flush_sum(&p.stats->castHelperSubtrees, helped); flush_sum(&p.stats->castNestedHelpers, nested);
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

#ifdef GMUPT_VARIANTS
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
#endif

// ------------------------------------------------------------------------------------------------ host launchers
bool traversal_is_fused(int mode) { return mode >= 60; }

// The rungs this build contains: cast0 (default) and def0 (separate launches); everything else needs -DGMUPT_VARIANTS
bool traversal_mode_available(int mode)
{
#ifdef GMUPT_VARIANTS
    (void)mode; return true;
#else
    return mode == 60 || mode == 40 || mode == 70;
#endif
}

// Both ray casts in one launch.  Returns the GMUPT_STAT_* bits of what was launched, or 0 when the fused kernel does not take this
// configuration (opt-in prunings; node / triangle arrays beyond the 32-bit offsets of its buffer resources): the caller then runs the
// two separate launches (launch_extend + launch_shadow).
uint32_t launch_cast_wide(const RenderParams& p, bool stats, hipStream_t s);   // pt_traverse_wide.hip

uint32_t launch_cast(const RenderParams& p, bool stats, int mode, hipStream_t s)
{
    if (mode == 70) {   // wide: the 4-wide collapse; a configuration it does not take (no collapse built, opt-in prunings) runs the default kernel
        const uint32_t launched = launch_cast_wide(p, stats, s);
        if (launched) return launched;
        mode = 60;
    }
    const uint32_t pb = p.travGridBlocks;
    const bool ovf = p.trav.maxDepth + 2 > (uint32_t)kDefStack;
    const bool plain = !p.extendPrune && !p.shadowPrune;   // the opt-in prunings exist in the separate bodies only
    // k_cast_f addresses nodes and triangle records with 32-bit byte offsets into buffer resources (< 2 GiB each: 33 M nodes, 44 M records)
    const bool fits = (uint64_t)p.trav.triBase * 64ull < (1ull << 31) && ((uint64_t)p.scene.numTris + 1ull) * 48ull < (1ull << 31);
    const uint32_t spill = ovf ? GMUPT_STAT_STACK_SPILL : 0u;
    if (mode == 60 && plain && fits) { GMUPT_DEF_LAUNCH(k_cast_f, true, 6, 4); return GMUPT_STAT_FUSED_CAST | GMUPT_STAT_CAST_FETCH | spill; } // cast0 (default): mixed lanes, fused fetches
#ifdef GMUPT_VARIANTS
    if (mode == 63 && plain && fits) { GMUPT_DEF_LAUNCH(k_cast_f, true, 4, 4); return GMUPT_STAT_FUSED_CAST | GMUPT_STAT_CAST_FETCH | spill; } // cast3: cast0 with 4 steps per iteration
    if ((mode == 62 || mode == 63) && plain) { if (launch_cast_variant(p, stats, mode, s)) return GMUPT_STAT_FUSED_CAST | spill; }              // cast2: mixed lanes, separate triangle bursts
    if (mode == 61 || !plain) { GMUPT_DEF_LAUNCH(k_cast_d, true, 4, 4); return GMUPT_STAT_FUSED_CAST | spill; }                                  // cast1: extension then shadow per wave
#endif
    return 0u;
}

uint32_t launch_extend(const RenderParams& p, uint32_t blocks, bool stats, int mode, hipStream_t s)
{
    const uint32_t pb = p.travGridBlocks;
    const bool ovf = p.trav.maxDepth + 2 > (uint32_t)kDefStack;
    if (mode >= 60) mode = 40;
#ifdef GMUPT_VARIANTS
    if (mode >= 50 || mode < 40) { launch_extend_variant(p, blocks, stats, mode, s); return 0u; }
    switch (mode - 40) {
    case 0: break;
    case 1: GMUPT_DEF_LAUNCH(k_extend_d, false, 4, 4); return 0u;
    case 2: GMUPT_DEF_LAUNCH(k_extend_d, true, 3, 4); return 0u;
    case 3: GMUPT_DEF_LAUNCH(k_extend_d, true, 2, 4); return 0u;
    default: GMUPT_DEF_LAUNCH(k_extend_d, true, 4, 2); return 0u;
    }
#else
    (void)blocks;
#endif
    GMUPT_DEF_LAUNCH(k_extend_d, true, 4, 4);   // def0
    return ovf ? GMUPT_STAT_STACK_SPILL : 0u;
}

uint32_t launch_shadow(const RenderParams& p, uint32_t blocks, bool stats, int mode, hipStream_t s)
{
    const uint32_t pb = p.travGridBlocks;
    const bool ovf = p.trav.maxDepth + 2 > (uint32_t)kDefStack;
    if (mode >= 60) mode = 40;
#ifdef GMUPT_VARIANTS
    if (mode >= 50 || mode < 40) { launch_shadow_variant(p, blocks, stats, mode, s); return 0u; }
    switch (mode - 40) {
    case 0: break;
    case 1: GMUPT_DEF_LAUNCH(k_shadow_d, false, 4, 4); return 0u;
    case 2: GMUPT_DEF_LAUNCH(k_shadow_d, true, 3, 4); return 0u;
    case 3: GMUPT_DEF_LAUNCH(k_shadow_d, true, 2, 4); return 0u;
    default: GMUPT_DEF_LAUNCH(k_shadow_d, true, 4, 2); return 0u;
    }
#else
    (void)blocks;
#endif
    GMUPT_DEF_LAUNCH(k_shadow_d, true, 4, 4);   // def0
    return ovf ? GMUPT_STAT_STACK_SPILL : 0u;
}

uint32_t traversal_block_threads() { return kTravBlock; }
uint32_t deferred_block_threads() { return kDefBlock; }
// nodes of the tree top the ray casts of a tree of this depth keep in LDS (the spilling-stack instantiations hold more, see kDefLdsTop)
uint32_t traversal_top_capacity(uint32_t maxDepth) { return maxDepth + 2 > (uint32_t)kDefStack ? (uint32_t)kDeepTopTreeNodes : (uint32_t)kTopTreeNodes; }
uint32_t traversal_wide_overflow_entries();   // pt_traverse_wide.hip
uint32_t traversal_overflow_entries()
{
    uint32_t a = (uint32_t)(kMaxStack + 1 - kDeepStack);
    if (traversal_wide_overflow_entries() > a) a = traversal_wide_overflow_entries();
#ifdef GMUPT_VARIANTS
    const uint32_t b = variant_overflow_entries(); if (b > a) a = b;
#endif
    return a;
}

} // namespace gmupt
