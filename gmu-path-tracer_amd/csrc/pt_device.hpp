// Device-side data layout of the wavefront path tracer (gfx950).
//
// Path state: the reference keeps 21 fields in one raw buffer, SoA by field with float3 in 16-byte slots
// (Assets/Shaders/structs.h:19-48, 248 B/path).  Here every scalar component is its own array of P words
// (50 words = 200 B/path, no padding): a wave reads one component of 64 consecutive slots as one 256-byte
// coalesced transaction.  gmupt_debug_{read,write}_path_state convert to/from the reference layout.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include "../../include/gmupt.h"

namespace gmupt {

enum Field : uint32_t {
    F_RAY_OX, F_RAY_OY, F_RAY_OZ, F_RAY_DX, F_RAY_DY, F_RAY_DZ,
    F_MAT_R, F_MAT_G, F_MAT_B, F_MAT_METALLIC, F_MAT_ROUGHNESS,
    F_NRM_X, F_NRM_Y, F_NRM_Z,
    F_SP_X, F_SP_Y, F_SP_Z,
    F_BARY_X, F_BARY_Y, F_BARY_Z,
    F_HIT_DIST,
    F_TRI_0, F_TRI_1, F_TRI_2, F_TRI_MAT,
    F_SH_OX, F_SH_OY, F_SH_OZ, F_SH_DX, F_SH_DY, F_SH_DZ,
    F_LIGHT_IDX, F_LIGHT_DIST, F_IN_SHADOW,
    F_RAD_R, F_RAD_G, F_RAD_B,
    F_THR_R, F_THR_G, F_THR_B,
    F_LTHR_R, F_LTHR_G, F_LTHR_B,
    F_DL_R, F_DL_G, F_DL_B,
    F_PATH_LEN, F_SCR_X, F_SCR_Y, F_IS_EMITTER,
    F_COUNT
};
static_assert(F_COUNT == 50, "50 words per path");

// per-slot class written by the logic kernel, consumed by the material kernel
enum SlotClass : uint8_t { CLS_UE4 = 0, CLS_GLASS = 1, CLS_ENDED = 2, CLS_RETIRED = 3, CLS_NONE = 4, CLS_MASK = 0x0F, CLS_SHADOW_BIT = 0x10 };
constexpr int kNumCounts = 4; // per-block counts: UE4, glass, ended, shadow-ray pushers
#ifndef GMUPT_SCAN_GROUP
#define GMUPT_SCAN_GROUP 64
#endif
constexpr uint32_t kScanGroup = GMUPT_SCAN_GROUP; // blocks per group of the two-level rank computation (a -D knob so that a test build can reach many groups with a small pool)

// queue counters, same indices as the reference (structs.h:62-68); [7] is this build's live extension-queue length
enum Counter : uint32_t { QC_NEWPATH = 0, QC_LASTPATHCNT = 1, QC_MATUE4 = 2, QC_MATGLASS = 3, QC_EXT_UE4_OFFSET = 4, QC_EXT_GLASS_OFFSET = 5, QC_SHADOWRAY = 6, QC_EXT_COUNT = 7 };
// queues, same order as the reference (structs.h:53-58): queue q lives at queues + q * P
enum Queue : uint32_t { Q_NEWPATH = 0, Q_MAT_UE4 = 1, Q_MAT_GLASS = 2, Q_EXT_RAY = 3, Q_SHADOW_RAY = 4 };

constexpr uint32_t kQueueHole = 0xFFFFFFFFu; // extension-queue entry of a slot retired by path_budget
constexpr uint32_t kListEnd = 0xFFFFFFFFu;
constexpr int kBlock = 256;

struct DevStats {
    unsigned long long pathsGenerated, pathsCompleted, segments;
    unsigned long long extRays, extInner, extLeaves, extTris;
    unsigned long long shRays, shInner, shLeaves, shTris;
    unsigned long long extWaveInner, extWaveTris, shWaveInner, shWaveTris; // wave-level loop iterations (SIMD efficiency = lane steps / (64 * wave iterations))
    unsigned long long extDepthHist[32]; // collect_stats: inner-node visits of the extension rays by node depth
    // collect_stats, fused ray cast (k_cast_m) only:
    unsigned long long laneCensus[4];     // lane-iterations without a ray / walking / holding a leaf for a full FIFO / walk finished, leaves pending
    unsigned long long castWaves, castWaveClocks, castWaveClocksMax; // per-wave lifetime in 100 MHz ticks: sum and maximum
    unsigned long long castDrainClocks, castDrainIters, castDrainBusyLanes; // after the wave found both queues empty: ticks, loop iterations, lanes with a ray summed over them
    unsigned long long castWaveEndHist[32]; // wave lifetimes in 50-us buckets (all waves of the persistent grid start together)
    unsigned long long rayInnerHist[32];    // extension rays by inner nodes visited, 16 per bucket
    unsigned long long extTopInner, shTopInner; // inner-node visits served by the LDS-resident top of the tree (no vector-memory request)
    unsigned long long castHelperSubtrees;      // subtrees handed to idle lanes in the drain of the fused ray cast
    unsigned long long castNestedHelpers;       // of those: subtrees a HELPER gave away in turn (collect_stats)
    unsigned long long castRedoRays;            // wide ray cast: rays walked again in the reference's binary order (closest hit an exact tie between two triangles; a full stack)
    unsigned long long widePairFetches;         // wide ray cast, collect_stats: 80-byte TriPair records fetched (a leaf of n references takes ceil(n / 2))
    unsigned long long wideBoxTests;            // wide ray cast, collect_stats: occupied box slots tested
    unsigned long long wideIters, wideGeneralIters; // wide ray cast, collect_stats: iterations of a wave (six steps each); of those: with the general slab test (a walking ray with an infinite / NaN 1 / d component)
    uint32_t activePaths;
    uint32_t stackOverflow; // bit 0: traversal needed more than the provisioned stack (results then differ from an unbounded stack); bit 1: a wave of the fused ray cast left at its iteration limit
};

struct alignas(16) DNode { float4 mn; float4 mx; int4 link; }; // gmupt_bvh_node, 48 B: link = (left, right, isLeaf, pad)
static_assert(sizeof(DNode) == sizeof(gmupt_bvh_node), "node layout");

struct SceneView {
    const DNode* nodes;
    const gmupt_triangle* tris;
    const float* verts;
    const gmupt_light* lights;
    const gmupt_tri_props* props;
    const gmupt_material* materials;
    uint32_t numNodes, numTris, numVerts, numMaterials;
    const uint8_t* tex[3];   // diffuse, metallicRoughness, normals: RGBA8 arrays of square layers (nullptr = unbound)
    uint32_t texSize[3], texLayers[3];
};

// Packed traversal copy of the scene, built at bind time (gmupt_renderer_bind_scene) from the reference-layout buffers.
// Node64: both children of one inner node -- boxes (12 floats) + child descriptors; inner nodes only, in the flatten order.
//   float4 a = (lmin.xyz, lmax.x)  b = (lmax.yz, rmin.xy)  c = (rmin.z, rmax.xyz)  int4 d = (leftDesc, rightDesc, 0, 0)
//   descriptor >= 0: index of an inner Node64; < 0: leaf whose first triangle record is ~descriptor
// Tri48: one SBVH reference -- v0, e1 = v1 - v0, e2 = v2 - v0 (the subtractions the shader does per test, done once in
//   binary32 on the host: bit-identical), then a flag word: != 0 on the last record of a leaf.
//   float4 r0 = (v0.xyz, e1.x)  r1 = (e1.yz, e2.xy)  r2 = (e2.z, lastFlag, 0, 0)
struct alignas(64) Node64 { float a[4], b[4], c[4]; int32_t d[4]; };
struct alignas(16) Tri48 { float r0[4], r1[4], r2[4]; };
static_assert(sizeof(Node64) == 64 && sizeof(Tri48) == 48, "packed traversal records");

// WNode: one node of the 4-wide collapse of the binary tree (pt_traverse_wide.hip) -- exactly one 128-byte line, the unit of every
// memory-side read of this chip.  Up to four descendants of one inner node of the binary tree (its two children, the inner ones among
// them replaced by THEIR children, largest surface area first, until four slots are taken), plane by plane so that the four slab tests
// of a step are 4-wide vector arithmetic:
//   p[0] = min.x of slots 0..3   p[1] = min.y   p[2] = min.z   p[3] = max.z   p[4] = max.y   p[5] = max.x   (an empty slot is all NaN: never hit;
//   the two planes of an axis are rows a and 5 - a: a ray fetches them in the order it meets them with ONE constant, row + other row = 5)
//   link[k] >= 0: index of the WNode of that descendant; < 0: leaf whose first triangle record is ~link[k]; kDone in an empty slot
//   aux[0] = depth of the node in the binary tree, aux[1] = number of occupied slots
struct alignas(128) WNode { float p[6][4]; int32_t link[4]; int32_t aux[4]; };
static_assert(sizeof(WNode) == 128, "one wide node per 128-byte line");

// TriPair: TWO references of one leaf, component by component, for the wide kernel's packed triangle test (two triangles per step on
// v_pk_* arithmetic): w[2c], w[2c + 1] = component c of the first / second triangle in the order v0.xyz, e1.xyz, e2.xyz (18 words), then
// w[18] != 0 on the last pair of the leaf, w[19] spare.  A leaf with an odd number of references ends with an all-zero second triangle
// (det = 0: rejected by the test itself).  pairRef[2 p + k] = the reference (index of the Tri48 / gmupt_triangle record) in slot k of pair p.
struct alignas(16) TriPair { float w[20]; };
static_assert(sizeof(TriPair) == 80, "packed triangle pairs");

// Rec64: the same data as 64-byte records in ONE array (inner nodes first, then one record per triangle reference) for the
// cooperative ray-cast kernels: a lane needs exactly one record per step, and four adjacent lanes fetch the four 16-byte
// quarters of one record with a single LDS-DMA instruction.  Triangle record: r0..r2 as Tri48, r3 = (v0, v1, v2, materialID).
struct alignas(64) Rec64 { float q[16]; };

#ifndef GMUPT_TOP_NODES
#define GMUPT_TOP_NODES 768
#endif
#ifndef GMUPT_DEEP_TOP_NODES
#define GMUPT_DEEP_TOP_NODES 1280
#endif
constexpr int kDeepTopTreeNodes = GMUPT_DEEP_TOP_NODES; // the same for trees whose depth needs the spilling stack anyway: fewer stack entries in LDS, more of the tree (80 KB)
constexpr int kTopTreeNodes = GMUPT_TOP_NODES; // inner nodes (breadth-first from the root) that the ray-cast kernels keep in LDS: 48 KB

struct TravScene {
    const Rec64* recs;
    uint32_t triBase;      // index of the first triangle record in recs (= number of inner nodes)
    const Node64* nodes;
    const Tri48* tris;
    int32_t rootDesc;
    uint32_t topCount;     // nodes[0 .. topCount) are the top of the tree (largest surface area first) that fits the LDS of the plain layout
    uint32_t topCountDeep; // the same for the layout of the spilling-stack instantiations (topCountDeep >= topCount: one numbering serves both)
    uint32_t maxDepth;     // depth of the deepest node (bounds the traversal stack)
    float rootMin[3], rootMax[3];
    // 4-wide collapse of the same tree (pt_traverse_wide.hip)
    const WNode* wnodes;
    uint32_t wideCount;      // records of wnodes
    uint32_t wideTopCount;   // wnodes[0 .. wideTopCount) live in LDS (largest surface area first)
    uint32_t wideStackBound; // entries the walk of this tree can have pending at most (all four slots hit on every level)
    int32_t wideRootDesc;    // 0 (the root's WNode) or the leaf descriptor of a one-leaf tree
    const TriPair* pairs;    // the leaves of the wide tree as triangle pairs (a leaf link of a WNode is ~first pair)
    const uint32_t* pairRef; // 2 per pair: the reference in that slot, 0xFFFFFFFF in a padding slot
    uint32_t numPairs;
};

struct RenderParams {
    float* state;          // F_COUNT * P words
    uint32_t P, L;         // pool / live slots
    uint8_t* cls;          // P
    uint32_t* listNext;    // P: per-pixel list of paths that ended this iteration
    uint32_t* listHead;    // fbW * fbH
    float* sample;         // 3 * P: tonemapped sample of an ended path
    uint32_t* blockCounts; // kNumCounts * nBlocks
    uint32_t nBlocks;
    uint32_t* groupTotals; // 2 * kNumCounts * nGroups: class counts summed over groups of kScanGroup blocks, double-buffered by iteration parity
    uint32_t nGroups;
    uint32_t groupParity;  // which half the current iteration uses
    uint32_t* queues;      // 5 * P
    uint32_t* qc;          // 8
    DevStats* stats;
    float4* fb;
    uint32_t fbW, fbH;
    uint32_t tileEnabled, tileX0, tileY0;
    uint32_t budget, maxDepth;
    int* ovfStack;         // global overflow of the traversal stacks
    uint32_t ovfStride;    // threads of the traversal grid
    uint32_t raysPerWave;  // queue entries owned by one wave of the persistent ray-cast kernels
    uint32_t* travCounters; // [0] extension, [1] shadow: next unassigned queue entry (zeroed by k_material every iteration); [4 + 8 phase + segment]: the same per queue segment (XCD experiment)
    uint32_t travGridBlocks; // persistent ray-cast grid
    uint32_t extendPrune;  // 1: the extension ray skips boxes it enters beyond its current closest hit (see pt_traverse.hip)
    uint32_t shadowPrune;  // 1: the shadow ray skips boxes it enters beyond the light (cannot change its boolean result)
    uint32_t PS;                        // words between two fields of the path state: P + a pad (fields that are a power of two apart share their HBM channel rotation)
    uint32_t tuneRefill, tuneTriThresh; // lane-refill / triangle-burst thresholds of the deferred-leaf kernels
    uint32_t tuneWideSteps;             // wide ray cast: steps per iteration (0: by table size; 6 or 8)
    uint32_t wideQuarterTail;           // wide ray cast (set at launch): the last half round of shadow-ray chunks in quarters
    uint32_t xcdBins;      // experiment builds only (-DGMUPT_WIDE_XCD_EXPERIMENT): the queues are eight equal segments, a wave serves the segment of its XCD first
    uint32_t castLoopCap;  // watchdog of the fused ray cast: loop iterations after which a wave gives up (GMUPT_STAT_CAST_ABORTED)
    gmupt_camera_buffer cam;
    SceneView scene;
    TravScene trav;
};

} // namespace gmupt
