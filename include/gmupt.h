/*
 * include/gmupt.h -- C-ABI of the MI355X-native wavefront path tracer (libgmupt.so).
 *
 * This is the drop-in boundary for the hot path of WildBitangent/GMU-Path-Tracer:
 * it replaces the D3D11 / NVAPI calls inside the reference's Renderer and Scene
 * (there is no FFI layer in the reference; the boundary is the set of D3D11 calls
 * listed per entry point below).  All citations are relative to the reference tree.
 * Plain C, POD arguments, no HIP / torch types in any signature.
 *
 * Conventions
 *   - every function returns GMUPT_OK (0) or a negative gmupt_status;
 *     gmupt_last_error() returns a thread-local description of the last failure
 *     (reference: HRESULT != S_OK -> std::runtime_error(fmt::format(..)),
 *      Source/Renderer.cpp:286-297,424-429,484-497; the C++ wrappers in
 *      gmu-path-tracer_amd/host re-throw std::runtime_error);
 *   - creator owns, explicit *_destroy (reference: uni::UniqueHandle<T>::Release,
 *     Include/UniqueDX11.hpp:7-91); uploads copy, the caller may free at once;
 *   - *_create / upload functions are thread-safe per device (reference creates
 *     buffers from a BVH worker and 3 texture workers concurrently,
 *     Source/Scene.cpp:89,153-155); per-renderer functions are single-threaded;
 *   - gmupt_iterate() enqueues on the renderer's HIP stream and does not
 *     synchronise with the host (reference: Renderer::draw never reads back).
 */
#ifndef GMUPT_H
#define GMUPT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GMUPT_MAX_LIGHTS 128    /* Include/Constants.hpp:15 MAX_LIGHTS (also the material cbuffer size, logic.hlsl:8) */
#define GMUPT_PATHCOUNT (1u << 21) /* Include/Constants.hpp:13 */
#define GMUPT_REF_GRID_THREADS (34u * 8u * 256u) /* NUM_SM*8 groups x NUM_THREADS, Constants.hpp:12-16 */
#define GMUPT_STATE_BYTES 248u  /* bytes per path of the reference path-state buffer, Source/Renderer.cpp:71 */

typedef enum {
    GMUPT_OK = 0,
    GMUPT_ERR_INVALID_ARGUMENT = -1,
    GMUPT_ERR_HIP = -2,          /* a HIP runtime call failed (no GPU, out of memory, launch failure) */
    GMUPT_ERR_OUT_OF_MEMORY = -3,
    GMUPT_ERR_NOT_BOUND = -4,    /* renderer used before a scene / camera was bound */
    GMUPT_ERR_UNSUPPORTED = -5,
    GMUPT_ERR_IO = -6,
    GMUPT_ERR_CAST_FAULT = -7    /* a ray-cast launch flagged its own results as invalid (GMUPT_STAT_STACK_OVERFLOW or GMUPT_STAT_CAST_ABORTED): returned by every
                                    call that hands out or waits for a frame -- gmupt_synchronize, gmupt_read_framebuffer, gmupt_copy_framebuffer_to_device,
                                    gmupt_get_stats (the statistics are still filled in), gmupt_render_budget -- until gmupt_reset_stats; the C++ Renderer throws,
                                    like the reference on a failed device call (Source/Renderer.cpp:286-297) */
} gmupt_status;

/* ---- POD layouts shared with the reference (sizes are static_assert'ed in the implementation) ---- */

/* Include/BVHWrapper.hpp:13-21 == Assets/Shaders/structs.h:182-194; 48 bytes.
 * inner node: children at left/right with right == left + 1 (Source/BVHWrapper.cpp:87-91);
 * leaf: [left, right) is a range of the triangle array. */
typedef struct { float min[3]; float pad0; float max[3]; float pad1; int32_t left; int32_t right; int32_t isLeaf; float pad2; } gmupt_bvh_node;
/* Include/BVHWrapper.hpp:23-27 == structs.h:196-200; 16 bytes, one per SBVH reference */
typedef struct { int32_t v[3]; uint32_t materialID; } gmupt_triangle;
/* Include/BVHWrapper.hpp:29-34 == structs.h:202-210; 32 bytes, one per VERTEX */
typedef struct { float normal[3]; float pad0; float uv[2]; uint32_t materialID; float pad1; } gmupt_tri_props;
/* Include/Scene.hpp:13-19 == structs.h:155-161; 32 bytes */
typedef struct { float position[3]; float falloff; float emission[3]; float radius; } gmupt_light;
/* Include/Scene.hpp:43-68 == structs.h:219-233; 48 bytes */
typedef struct { float color[4]; float metallic; float roughness; float refractIndex; float transmittance; int32_t textureIndices[3]; uint32_t materialType; } gmupt_material;
/* Include/Camera.hpp:8-22 == structs.h:163-180; 112 bytes, uploaded every frame (Source/Renderer.cpp:161) */
typedef struct {
    float position[4]; float upperLeftCorner[4]; float horizontal[4]; float vertical[4];
    float pixelSize[2]; float randomSeed[2]; float envColor[4];
    int32_t iterationCounter; uint32_t lightCount; uint32_t sampleLights; uint32_t pad_;
} gmupt_camera_buffer;

enum { GMUPT_MATERIAL_UE4 = 0, GMUPT_MATERIAL_GLASS = 1 }; /* Scene.hpp:53-57 */

/* ---- device ---- */
typedef struct gmupt_device gmupt_device;
/* replaces Renderer::createDevice (Source/Renderer.cpp:252-301): selects a HIP device (one process per GPU) */
int gmupt_device_create(int hip_device, gmupt_device** out);
void gmupt_device_destroy(gmupt_device* dev);
const char* gmupt_last_error(void);
/* number of visible HIP devices, or a negative status (does not initialise a context) */
int gmupt_device_count(void);

/* ---- scene resources ---- */
typedef enum {
    GMUPT_BUFFER_BVH_NODES = 0,  /* 48 B elements; Scene.cpp:174 mBVHBuffer  (t0) */
    GMUPT_BUFFER_TRIANGLES = 1,  /* 16 B;          Scene.cpp:175 mIndexBuffer (t1) */
    GMUPT_BUFFER_VERTICES = 2,   /* 12 B float3;   Scene.cpp:176 mVertexBuffer (t2) */
    GMUPT_BUFFER_LIGHTS = 3,     /* 32 B x <=128;  Scene.cpp:305-329 mLightBuffer (t3); zero-padded to 128 entries */
    GMUPT_BUFFER_TRI_PROPS = 4,  /* 32 B;          Scene.cpp:177 mTriangleProperties (t4) */
    GMUPT_BUFFER_MATERIALS = 5,  /* 48 B x <=128;  Scene.cpp:194-207 mMaterialPropertyBuffer (b1) */
    GMUPT_BUFFER_TEXTURE_ARRAY = 6 /* R8G8B8A8_UNORM Texture2DArray, square layers; created with gmupt_texture_array_create */
} gmupt_buffer_kind;

typedef struct gmupt_buffer gmupt_buffer;
/* replaces createBuffer<T> (Include/Util.hpp:17-43) + ID3D11Device::CreateBuffer with initial data */
int gmupt_buffer_create(gmupt_device* dev, gmupt_buffer_kind kind, const void* data, size_t bytes, gmupt_buffer** out);
/* replaces the light-buffer re-upload of the GUI (Source/GUI.cpp:125-130): UpdateSubresource on an existing buffer.
 * After updating node / triangle / vertex buffers call gmupt_renderer_bind_scene again (it rebuilds the traversal copy). */
int gmupt_buffer_update(gmupt_buffer* buf, const void* data, size_t bytes);
void gmupt_buffer_destroy(gmupt_buffer* buf);
size_t gmupt_buffer_size(const gmupt_buffer* buf);
/* replaces Scene::createTextures (Source/Scene.cpp:247-303): `layers` square RGBA8 images of `size` x `size` texels, tightly packed
 * (the caller has already resized every layer to the common size, as the reference does with avir) */
int gmupt_texture_array_create(gmupt_device* dev, const uint8_t* rgba8, uint32_t size, uint32_t layers, gmupt_buffer** out);

/* Host-side image helpers for the texture path (no device work).
 * gmupt_image_decode_png replaces lodepng::decode(out, w, h, file) (Source/Scene.cpp:226): PNG file bytes -> tightly packed RGBA8;
 * *rgba is allocated by the library, release it with gmupt_image_free.
 * gmupt_image_resize_square replaces avir::CImageResizer<fpclass_float8_dil>(8)::resizeImage(src, n, n, 0, dst, m, m, 4, 0) on square RGBA8
 * layers (Scene.cpp:269-279): the same bytes as avir 2.4 (Include/avir/avir.h) for every size pair, pinned against avir compiled from the
 * reference tree (tests/golden/texture_ref.npz); equal sizes run avir's filters too, they are not a copy.  dst holds new_size * new_size * 4 bytes.
 * gmupt_texture_common_size is the reference's size rule (Scene.cpp:232-241): median of the DISTINCT layer byte sizes -> width. */
int gmupt_image_decode_png(const void* png, size_t bytes, uint32_t* width, uint32_t* height, uint8_t** rgba);
void gmupt_image_free(uint8_t* rgba);
int gmupt_image_resize_square(const uint8_t* rgba, uint32_t old_size, uint32_t new_size, uint8_t* dst);
uint32_t gmupt_texture_common_size(const size_t* layer_bytes, uint32_t layers);

/* ---- renderer ---- */
typedef struct {
    uint32_t width, height;      /* accumulation target (== tile size when tile_enabled); createRenderTexture, Renderer.cpp:110-142 */
    uint32_t pool_paths;         /* PATHCOUNT; 0 -> GMUPT_PATHCOUNT (Renderer.cpp:71,78) */
    uint32_t live_paths;         /* slots the stage loops reach; 0 -> pool_paths.
                                    Reference value: (pool/GMUPT_REF_GRID_THREADS)*GMUPT_REF_GRID_THREADS (quirk: ITERATIONS = 30) */
    /* extensions of this build; all zero = reference behaviour */
    uint32_t tile_enabled, tile_x0, tile_y0; /* multi-GPU: generate paths for this tile of the full frame only */
    uint32_t path_budget;        /* stop regenerating after this many paths (0 = progressive, never stops) */
    uint32_t max_depth;          /* terminate at this path length (0 = unbounded, the reference) */
    uint32_t collect_stats;      /* 1: traverse kernels also count inner-node visits / triangle tests (slower) */
} gmupt_renderer_desc;

typedef struct gmupt_renderer gmupt_renderer;
/* replaces Renderer::createBuffers + createRenderTexture (Renderer.cpp:58-142): path state, queues, counters, accumulation target (zero-filled) */
int gmupt_renderer_create(gmupt_device* dev, const gmupt_renderer_desc* desc, gmupt_renderer** out);
void gmupt_renderer_destroy(gmupt_renderer* r);
/* replaces CSSetShaderResources / CSSetConstantBuffers(b1) of Renderer::draw (Renderer.cpp:166-192).
 * The buffers must outlive the binding.  Builds the renderer's internal traversal copy of the BVH. */
int gmupt_renderer_bind_scene(gmupt_renderer* r, const gmupt_buffer* nodes, const gmupt_buffer* triangles, const gmupt_buffer* vertices,
                              const gmupt_buffer* lights, const gmupt_buffer* tri_props, const gmupt_buffer* materials);
/* replaces the t5..t7 / s0 slots of CSSetShaderResources + CSSetSamplers (Renderer.cpp:173-175,192): diffuse, metallicRoughness and
 * normal Texture2DArrays, each may be NULL (an unbound slot reads zero).  Sampling is bilinear + wrap at mip 0 on UNORM8 without sRGB
 * decode, as the reference's sampler (Scene.cpp:180-192); the filter arithmetic is stated in DESIGN.md. */
int gmupt_renderer_bind_textures(gmupt_renderer* r, const gmupt_buffer* diffuse, const gmupt_buffer* metallic_roughness, const gmupt_buffer* normals);
/* replaces UpdateSubresource(mCameraBuffer) (Renderer.cpp:161).  iterationCounter == 0 resets the accumulation on the next iterate. */
int gmupt_set_camera(gmupt_renderer* r, const gmupt_camera_buffer* cam);
/* replaces the six Dispatch(NUM_GROUPS,1,1) of Renderer::draw (Renderer.cpp:195-211): one wavefront iteration, asynchronous */
int gmupt_iterate(gmupt_renderer* r);
/* replaces Renderer::resize -> createRenderTexture (Renderer.cpp:408-413): new zeroed accumulation target */
int gmupt_resize(gmupt_renderer* r, uint32_t width, uint32_t height);
/* replaces captureScreen's staging copy + Map (Renderer.cpp:355-381): synchronous readback of width*height RGBA32F (a = sample count bits) */
int gmupt_read_framebuffer(gmupt_renderer* r, float* rgba, size_t bytes);
/* device-to-device copy of the accumulation target into caller-owned device memory (e.g. a torch tensor) on the renderer's stream,
 * followed by a stream synchronise -- used by the multi-GPU tile gather */
int gmupt_copy_framebuffer_to_device(gmupt_renderer* r, void* device_dst, size_t bytes);
/* the 8 queue counters (structs.h:62-68; [7] = live extension-queue entries), synchronous */
int gmupt_get_counters(gmupt_renderer* r, uint32_t out[8]);
int gmupt_synchronize(gmupt_renderer* r);

#define GMUPT_STAT_STACK_OVERFLOW 1u /* a traversal stack exceeded 64 entries (results invalid; never seen on a builder-made tree) */
#define GMUPT_STAT_FUSED_CAST 2u     /* both ray casts ran as one launch: ms_extend is the time of that launch, ms_shadow is 0 */
#define GMUPT_STAT_CAST_FETCH 4u     /* that launch was k_cast_f (the default kernel; it needs node / triangle arrays below 2 GiB each) */
#define GMUPT_STAT_CAST_ABORTED 16u  /* a wave of the fused ray cast left its loop at the iteration limit (2^20 loop iterations; a bench-scene wave runs
                                        ~150): a defect, results invalid -- the kernel ends whatever happens */
#define GMUPT_STAT_CAST_WIDE 32u     /* that launch was k_cast_w: the walk over the 4-wide collapse of the tree (GMUPT_TRAVERSAL=wide) */
#define GMUPT_STAT_STACK_SPILL 8u    /* the tree is deeper than the LDS part of the traversal stacks: the instantiation with the bounds-checked
                                        global spill ran (results are the same; GMUPT_STAT_STACK_OVERFLOW is the error flag) */
typedef struct {
    uint64_t iterations;
    uint64_t paths_generated;    /* new paths started (device counter) */
    uint64_t paths_completed;    /* paths accumulated into the framebuffer */
    uint64_t segments;           /* live-slot iterations (the reference overlay's "MP/s" unit, GUI.cpp:48) */
    uint32_t active_paths;       /* slots not retired by path_budget */
    uint32_t flags;              /* GMUPT_STAT_* bits; the ray-cast bits describe the launches since the last gmupt_reset_stats */
    /* collect_stats only */
    uint64_t ext_rays, ext_inner, ext_leaves, ext_tris;
    uint64_t sh_rays, sh_inner, sh_leaves, sh_tris;
    /* device time per stage group in ms, accumulated since the last reset (HIP events on the renderer's stream; timing must be enabled) */
    double ms_logic, ms_scan, ms_accumulate, ms_material, ms_extend, ms_shadow; /* ms_scan and ms_accumulate are always 0: the queue ranks
                                    are computed inside the logic / material kernels and the accumulation inside the material kernel */
    uint64_t timed_iterations;
    /* collect_stats only: wave-level loop iterations of the ray casts; SIMD efficiency = lane steps / (64 * wave iterations) */
    uint64_t ext_wave_inner, ext_wave_tris, sh_wave_inner, sh_wave_tris;
    uint64_t ext_depth_hist[32]; /* inner-node visits of the extension rays by node depth */
    /* collect_stats with the fused ray cast (GMUPT_STAT_FUSED_CAST) only */
    uint64_t lane_census[4];          /* lane-iterations: no ray / walking / holding a leaf for a full FIFO / walk done, leaves pending */
    uint64_t cast_waves, cast_wave_ticks, cast_wave_ticks_max; /* wave lifetimes in wall-clock ticks: count, sum, maximum */
    uint64_t cast_drain_ticks, cast_drain_iters, cast_drain_busy_lanes; /* after a wave found both queues empty: ticks, loop iterations, busy lanes summed over them */
    uint64_t cast_wave_end_hist[32];  /* wave lifetimes in 50-us buckets */
    uint64_t ray_inner_hist[32];      /* extension rays by inner nodes visited, 16 per bucket */
    uint64_t ext_top_inner, sh_top_inner; /* inner-node visits served by the LDS-resident top of the tree (k_cast_f, collect_stats) */
    uint64_t cast_helper_subtrees;        /* deferred subtrees walked by a finished lane for a lane still walking, in the drain of k_cast_f (collect_stats) */
    uint64_t cast_nested_helpers;         /* of those: subtrees a helper lane gave away in turn (collect_stats) */
    uint64_t cast_redo_rays;              /* wide ray cast: rays walked again in the reference's binary order (their closest hit was an exact tie in t between
                                             two triangles, or a traversal stack ran full) */
    uint64_t wide_nodes, wide_top_nodes, wide_stack_bound; /* the 4-wide collapse of the bound scene (GMUPT_TRAVERSAL=wide; 0 when none was built): 128-byte records,
                                             how many of them live in LDS, and the most entries the inner stack of a walk could hold (every slot hit on every level) */
    uint64_t wide_pairs, wide_pair_fetches; /* the leaves of that collapse as 80-byte triangle-pair records: how many there are; how many were fetched (collect_stats) */
    uint64_t wide_box_tests;              /* wide ray cast, collect_stats: occupied box slots tested */
    uint64_t wide_iterations, wide_general_iterations; /* wide ray cast, collect_stats: iterations of a wave (six steps each); of those: with the general slab test
                                             (a ray of the wave has an infinite or NaN 1 / d component: near / far plane not known from the sign) */
} gmupt_stats;
int gmupt_get_stats(gmupt_renderer* r, gmupt_stats* out); /* synchronises */
int gmupt_reset_stats(gmupt_renderer* r);
int gmupt_enable_timing(gmupt_renderer* r, int mode); /* hipEvent timing on the renderer's stream: 0 off (default), 1 every stage group, 2 only the extension ray cast */

/* render until path_budget paths have completed (desc.path_budget must be > 0).  Every frame does what the reference's
 * Window::loop does (Source/Window.cpp:86-87): Camera::update (new randomSeed pair, iterationCounter++), upload, iterate.
 * The drain check reads one device word every 8 iterations; the loop ends when no slot is active, or 512 iterations after the budget
 * ran out (2.5 x the ~205-iteration life of a healthy path; what is still alive then are the reference's NaN-throughput paths, which
 * end only when their ray happens to hit a light).  Returns the iterations run in *iters. */
typedef struct gmupt_camera gmupt_camera;
int gmupt_render_budget(gmupt_renderer* r, gmupt_camera* camera, uint32_t max_iterations, uint32_t* iters);

/* ---- test / debug access (reference path-state layout, structs.h:19-48) ---- */
int gmupt_debug_read_path_state(gmupt_renderer* r, void* dst, size_t bytes);        /* 248 * pool_paths */
int gmupt_debug_write_path_state(gmupt_renderer* r, const void* src, size_t bytes);
int gmupt_debug_read_queues(gmupt_renderer* r, uint32_t* dst, size_t bytes);        /* 5 * pool_paths u32, structs.h:53-58 */
int gmupt_debug_write_queues(gmupt_renderer* r, const uint32_t* src, size_t bytes);
int gmupt_debug_write_counters(gmupt_renderer* r, const uint32_t in[8]);
int gmupt_debug_write_framebuffer(gmupt_renderer* r, const float* rgba, size_t bytes);
typedef enum { GMUPT_STAGE_SHADE = 0 /* logic+newPath+materialUE4+materialGlass */, GMUPT_STAGE_EXTEND = 1, GMUPT_STAGE_SHADOW = 2,
               GMUPT_STAGE_RAYCASTS = 3 /* both ray casts as gmupt_iterate launches them (one fused launch by default) */ } gmupt_stage;
int gmupt_debug_run_stage(gmupt_renderer* r, gmupt_stage stage);
/* evaluates the device copy of the deterministic math (fn: 0 sin, 1 cos, 2 log2, 3 exp2, 4 pow(x,y), 5 frac, 6 rng probe) */
int gmupt_debug_detmath(gmupt_device* dev, int fn, const float* x, const float* y, float* out, uint32_t n);

/* ---- host side: SBVH build + flatten (replaces BVHWrapper::buildSBVH, Source/BVHWrapper.cpp:13-96, and the vendored Nvidia-SBVH builder) ---- */
typedef struct {
    float split_alpha;       /* BVH::BuildParams::splitAlpha = 1e-5 (Include/Nvidia-SBVH/BVH.h:77) */
    int32_t max_depth;       /* SplitBVHBuilder::MaxDepth = 64 */
    int32_t max_spatial_depth; /* MaxSpatialDepth = 48 */
    int32_t min_leaf_size;   /* Platform default 1 */
    int32_t max_leaf_size;   /* Platform default 0x7FFFFFF */
    float node_cost, tri_cost; /* Platform default 1, 1 */
} gmupt_sbvh_params;
void gmupt_sbvh_default_params(gmupt_sbvh_params* p);
typedef struct gmupt_sbvh gmupt_sbvh;
/* vertices: numVerts tightly packed float3; indices: numTris * 3 vertex indices */
int gmupt_sbvh_build(const float* vertices, uint32_t num_vertices, const int32_t* indices, uint32_t num_triangles,
                     const gmupt_sbvh_params* params, gmupt_sbvh** out);
uint32_t gmupt_sbvh_num_nodes(const gmupt_sbvh* h);
uint32_t gmupt_sbvh_num_references(const gmupt_sbvh* h);
float gmupt_sbvh_sah(const gmupt_sbvh* h);
uint32_t gmupt_sbvh_depth(const gmupt_sbvh* h);
/* flattened reference layout; vertex_material may be NULL (materialID 0). ref_triangle (optional) receives the source triangle of each reference */
int gmupt_sbvh_flatten(const gmupt_sbvh* h, const uint32_t* vertex_material, gmupt_bvh_node* nodes, gmupt_triangle* triangles, int32_t* ref_triangle);
void gmupt_sbvh_destroy(gmupt_sbvh* h);

/* ---- host side: camera (replaces Camera::updateResolution / update / setRotation, Source/Camera.cpp) ---- */
int gmupt_camera_create(uint32_t width, uint32_t height, gmupt_camera** out);
void gmupt_camera_destroy(gmupt_camera* c);
void gmupt_camera_update_resolution(gmupt_camera* c, uint32_t width, uint32_t height);
void gmupt_camera_set_pose(gmupt_camera* c, float x, float y, float z, float pitch, float yaw); /* Scene.cpp:95-97 */
void gmupt_camera_update(gmupt_camera* c, float dt); /* Camera::update without input devices */
void gmupt_camera_reset_accumulation(gmupt_camera* c); /* iterationCounter = -1 (Renderer.cpp:152) */
/* input for the next gmupt_camera_update calls: mouse delta in degrees (yaw += dx, pitch -= dy, consumed by one update; Camera.cpp:28-33) and the
 * W/S/A/D key states (bit 0..3, held until changed; 5 units/s, :47-57).  Any input restarts the accumulation by the reference's
 * hysteresis rule (:72-83): iterationCounter -> 0 when it is > 4, and once more when the motion has stopped. */
void gmupt_camera_set_input(gmupt_camera* c, float mouse_dx, float mouse_dy, uint32_t keys_wsad);
gmupt_camera_buffer* gmupt_camera_get_buffer(gmupt_camera* c);

const char* gmupt_version(void);

#ifdef __cplusplus
}
#endif
#endif
