/*
 * oracle/gmupt_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C) of the wavefront path-tracing hot path of
 * WildBitangent/GMU-Path-Tracer: the six HLSL compute stages dispatched by
 * Renderer::draw() (reference Source/Renderer.cpp:195-211) plus the host-side
 * camera constant buffer / seed stream (Source/Camera.cpp:13-88).
 *
 * PARITY UNPINNED: the reference ships no tests, no golden vectors, no captured
 * images and no CPU renderer, its GPU stages (HLSL cs_5_0 + NVAPI) cannot be
 * compiled or run here, and its output is schedule-nondeterministic
 * (SURVEY.md header facts 1, 4, 5; section 8c).  This oracle therefore follows
 * the shader sources line by line under the *canonical schedule* (work items of
 * every stage execute in ascending index order, framebuffer read-modify-writes
 * are applied sequentially in that order) and with the deterministic fp32 math
 * of oracle/detmath.h in place of the vendor sin/cos/pow.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use
 * this library.  The product (libgmupt.so) never links or loads it.
 */
#ifndef GMUPT_ORACLE_H
#define GMUPT_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_MAX_LIGHTS 128 /* Include/Constants.hpp:15 */
#define ORC_STATE_BYTES 248 /* Source/Renderer.cpp:71 */

/* Camera constant buffer, 112 bytes: Include/Camera.hpp:8-22 == Assets/Shaders/structs.h:163-180 */
typedef struct {
    float pos[4];
    float ulc[4];
    float horizontal[4];
    float vertical[4];
    float pixelSize[2];
    float randomSeed[2];
    float envColor[4];
    int32_t sampleCounter; /* iterationCounter on the host side */
    uint32_t lightCount;
    uint32_t sampleLights;
    uint32_t pad_;
} orc_camera_buffer;

typedef struct { float min[3]; float pad0; float max[3]; float pad1; int32_t left, right, isLeaf; float pad2; } orc_bvh_node; /* 48 B BVHWrapper.hpp:13-21 */
typedef struct { int32_t v[3]; uint32_t materialID; } orc_triangle;                                        /* 16 B BVHWrapper.hpp:23-27 */
typedef struct { float normal[3]; float pad0; float uv[2]; uint32_t materialID; float pad1; } orc_tri_props; /* 32 B BVHWrapper.hpp:29-34 */
typedef struct { float position[3]; float falloff; float emission[3]; float radius; } orc_light;            /* 32 B Scene.hpp:13-19 */
typedef struct { float color[4]; float metallic, roughness, refractIndex, transmittance; int32_t tex[3]; uint32_t type; } orc_material; /* 48 B Scene.hpp:43-68 */

typedef struct {
    const orc_bvh_node* nodes; uint32_t numNodes;
    const orc_triangle* tris; uint32_t numTris;
    const float* verts; uint32_t numVerts;          /* tightly packed float3 */
    const orc_tri_props* props;                     /* one per vertex */
    const orc_light* lights;                        /* ORC_MAX_LIGHTS entries */
    const orc_material* materials; uint32_t numMaterials;
    /* Texture2DArray t5..t7 (diffuse, metallicRoughness, normals): RGBA8 UNORM, square layers (Scene.cpp:247-303); NULL = unbound */
    const uint8_t* tex[3]; uint32_t texSize[3]; uint32_t texLayers[3];
} orc_scene;

typedef struct {
    uint32_t poolPaths;   /* PATHCOUNT (Constants.hpp:13) */
    uint32_t livePaths;   /* slots the fixed-size grid-stride loops reach: ITERATIONS*NUM_GROUPS*NUM_THREADS (quirk Q1) */
    uint32_t fbWidth, fbHeight; /* accumulation target = tile size */
    /* build-side extensions (documented in DESIGN.md); all zero = reference behaviour */
    uint32_t tileEnabled, tileX0, tileY0; /* map new paths over this tile only; coord is global */
    uint32_t pathBudget;  /* stop regenerating once this many paths were generated (0 = never) */
    uint32_t maxDepth;    /* terminate when pathLength >= maxDepth (0 = unbounded) */
    uint32_t stackSize;   /* traversal stack entries (reference: 16, unchecked). 0 -> 64 */
    uint32_t threads;     /* OpenMP threads for the two ray-cast stages (order-independent); 0/1 = serial */
} orc_config;

typedef struct {
    uint64_t extRays, extInner, extLeaves, extTris;
    uint64_t shRays, shInner, shLeaves, shTris;
    uint64_t pathsGenerated, pathsEnded, segments;
    uint32_t maxStack;
} orc_stats;

typedef struct orc_renderer orc_renderer;

orc_renderer* orc_create(const orc_scene* scene, const orc_config* cfg);
void orc_destroy(orc_renderer*);
void orc_set_camera(orc_renderer*, const orc_camera_buffer* cam);
/* one Renderer::draw(): logic, newPath, materialUE4, materialGlass, extensionRayCast, shadowRayCast */
void orc_iterate(orc_renderer*);
/* individual stages (for stage-level parity tests); order as in Renderer.cpp:195-211 */
void orc_stage_logic(orc_renderer*);
void orc_stage_new_path(orc_renderer*);
void orc_stage_material_ue4(orc_renderer*);
void orc_stage_material_glass(orc_renderer*);
void orc_stage_extension(orc_renderer*);
void orc_stage_shadow(orc_renderer*);

uint8_t* orc_path_state(orc_renderer*);   /* 248 * poolPaths bytes, reference layout structs.h:19-48 */
uint32_t* orc_queues(orc_renderer*);      /* 5 * poolPaths u32, structs.h:53-58 */
uint32_t* orc_counters(orc_renderer*);    /* 8 u32, structs.h:62-68 (+[7] = live extension count, extension) */
float* orc_framebuffer(orc_renderer*);    /* fbWidth*fbHeight RGBA32F, a = sample count bits */
void orc_get_stats(orc_renderer*, orc_stats*);
void orc_reset_stats(orc_renderer*);
uint32_t orc_active_paths(orc_renderer*); /* slots not retired by pathBudget */

/* host side: Camera.cpp restatement */
typedef struct {
    orc_camera_buffer cb;
    float front[3], up[3], left[3];
    float halfWidth, halfHeight;
    float pitch, yaw;
    int moveHysteresis;
    uint32_t randState; /* MSVC LCG state, starts at 1 (no srand in the reference) */
} orc_camera;

void orc_camera_init(orc_camera*, uint32_t width, uint32_t height); /* Camera::Camera + updateResolution */
void orc_camera_update_resolution(orc_camera*, uint32_t width, uint32_t height);
void orc_camera_set_pose(orc_camera*, float x, float y, float z, float pitch, float yaw);
void orc_camera_update(orc_camera*); /* Camera::update with no input (dt irrelevant) */
int orc_msvc_rand(uint32_t* state);

/* texture filter probe */
void orc_debug_sample(const orc_scene* sc, int which, float u, float v, int layer, float* out4);

/* detmath probes for the GPU-vs-oracle bit tests */
void orc_detmath_eval(int fn, const float* x, const float* y, float* out, uint32_t n);

#ifdef __cplusplus
}
#endif
#endif
