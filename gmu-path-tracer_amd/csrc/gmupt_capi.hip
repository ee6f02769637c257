// C-ABI of libgmupt.so (see include/gmupt.h for the contract and the reference call sites each entry replaces).
// Host code only; the kernels live in pt_kernels.hip.
#include "pt_device.hpp"
#include "../host/sbvh_builder.hpp"
#include "../host/Camera.hpp"
#include "../host/TextureLoader.hpp"
#include "../host/png_reader.hpp"

#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <string>
#include <unordered_map>
#include <mutex>
#include <new>
#include <string>
#include <vector>

static_assert(sizeof(gmupt_bvh_node) == 48, "BVHNode is 48 bytes (Include/BVHWrapper.hpp:13-21)");
static_assert(sizeof(gmupt_triangle) == 16, "Triangle is 16 bytes (Include/BVHWrapper.hpp:23-27)");
static_assert(sizeof(gmupt_tri_props) == 32, "TriangleProperties is 32 bytes (Include/BVHWrapper.hpp:29-34)");
static_assert(sizeof(gmupt_light) == 32, "Light is 32 bytes (Include/Scene.hpp:13-19)");
static_assert(sizeof(gmupt_material) == 48, "MaterialProperty is 48 bytes (Include/Scene.hpp:43-68)");
static_assert(sizeof(gmupt_camera_buffer) == 112, "CameraBuffer is 112 bytes (Include/Camera.hpp:8-22)");
static_assert(offsetof(gmupt_camera_buffer, pixelSize) == 64 && offsetof(gmupt_camera_buffer, randomSeed) == 72 &&
              offsetof(gmupt_camera_buffer, envColor) == 80 && offsetof(gmupt_camera_buffer, iterationCounter) == 96 &&
              offsetof(gmupt_camera_buffer, lightCount) == 100 && offsetof(gmupt_camera_buffer, sampleLights) == 104, "Cam cbuffer offsets (structs.h:163-180)");
static_assert(offsetof(gmupt_bvh_node, max) == 16 && offsetof(gmupt_bvh_node, left) == 32 && offsetof(gmupt_bvh_node, isLeaf) == 40, "BVHNode offsets");
static_assert(offsetof(gmupt_material, metallic) == 16 && offsetof(gmupt_material, textureIndices) == 32 && offsetof(gmupt_material, materialType) == 44, "MaterialProperty offsets");

namespace gmupt {
void launch_clear(const RenderParams& p, hipStream_t s);
void launch_logic(const RenderParams& p, hipStream_t s);
void launch_material(const RenderParams& p, int clearFrame, hipStream_t s);
uint32_t launch_extend(const RenderParams& p, uint32_t blocks, bool stats, int mode, hipStream_t s);   // the launch_* of the ray casts return GMUPT_STAT_* bits of what they launched
uint32_t launch_shadow(const RenderParams& p, uint32_t blocks, bool stats, int mode, hipStream_t s);
uint32_t launch_cast(const RenderParams& p, bool stats, int mode, hipStream_t s);                      // 0: not launched, run the two separate casts
bool traversal_is_fused(int mode);
bool traversal_mode_available(int mode);
void launch_detmath(int fn, const float* x, const float* y, float* out, uint32_t n, hipStream_t s);
uint32_t traversal_block_threads();
uint32_t deferred_block_threads();
uint32_t traversal_overflow_entries();
uint32_t traversal_top_capacity(uint32_t maxDepth);
uint32_t traversal_wide_top_capacity();
}
using namespace gmupt;

// ------------------------------------------------------------------------------------------------ errors
static thread_local std::string g_lastError;

static int fail(int code, const char* fmt, ...)
{
    char buf[1024];
    va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof(buf), fmt, ap); va_end(ap);
    g_lastError = buf;
    return code;
}
#define HIP_TRY(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return fail(GMUPT_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); } while (0)

extern "C" const char* gmupt_last_error(void) { return g_lastError.c_str(); }
extern "C" const char* gmupt_version(void) { return "gmupt 0.1 (gfx950)"; }

// ------------------------------------------------------------------------------------------------ device
struct gmupt_device { int id; hipDeviceProp_t prop; };

extern "C" int gmupt_device_count(void)
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) return fail(GMUPT_ERR_HIP, "hipGetDeviceCount failed: %s", hipGetErrorString(e));
    return n;
}

extern "C" int gmupt_device_create(int hip_device, gmupt_device** out)
{
    if (!out) return fail(GMUPT_ERR_INVALID_ARGUMENT, "gmupt_device_create: out is null");
    *out = nullptr;
    int n = 0;
    HIP_TRY(hipGetDeviceCount(&n));
    if (hip_device < 0 || hip_device >= n) return fail(GMUPT_ERR_INVALID_ARGUMENT, "gmupt_device_create: device %d of %d", hip_device, n);
    gmupt_device* d = new (std::nothrow) gmupt_device();
    if (!d) return fail(GMUPT_ERR_OUT_OF_MEMORY, "gmupt_device_create: out of host memory");
    d->id = hip_device;
    hipError_t e = hipSetDevice(hip_device);
    if (e == hipSuccess) e = hipGetDeviceProperties(&d->prop, hip_device);
    if (e != hipSuccess) { delete d; return fail(GMUPT_ERR_HIP, "device %d: %s", hip_device, hipGetErrorString(e)); }
    *out = d;
    return GMUPT_OK;
}

extern "C" void gmupt_device_destroy(gmupt_device* dev) { delete dev; }

// ------------------------------------------------------------------------------------------------ buffers
struct gmupt_buffer { gmupt_device* dev; gmupt_buffer_kind kind; void* dptr; size_t bytes; size_t elems; uint32_t texSize = 0, texLayers = 0; };

static size_t kind_stride(gmupt_buffer_kind k)
{
    switch (k) {
    case GMUPT_BUFFER_BVH_NODES: return 48; case GMUPT_BUFFER_TRIANGLES: return 16; case GMUPT_BUFFER_VERTICES: return 12;
    case GMUPT_BUFFER_LIGHTS: return 32; case GMUPT_BUFFER_TRI_PROPS: return 32; case GMUPT_BUFFER_MATERIALS: return 48;
    case GMUPT_BUFFER_TEXTURE_ARRAY: return 4;
    }
    return 0;
}

extern "C" int gmupt_buffer_create(gmupt_device* dev, gmupt_buffer_kind kind, const void* data, size_t bytes, gmupt_buffer** out)
{
    if (!dev || !out) return fail(GMUPT_ERR_INVALID_ARGUMENT, "gmupt_buffer_create: null argument");
    *out = nullptr;
    const size_t stride = kind_stride(kind);
    if (!stride) return fail(GMUPT_ERR_INVALID_ARGUMENT, "gmupt_buffer_create: unknown kind %d", (int)kind);
    if (bytes % stride) return fail(GMUPT_ERR_INVALID_ARGUMENT, "gmupt_buffer_create: %zu bytes is not a multiple of the %zu-byte element", bytes, stride);
    if (bytes && !data) return fail(GMUPT_ERR_INVALID_ARGUMENT, "gmupt_buffer_create: data is null");
    size_t alloc = bytes;
    // lights / materials live in fixed 128-entry tables (Scene.hpp:116, logic.hlsl:8); entries past the data are zero
    if (kind == GMUPT_BUFFER_LIGHTS || kind == GMUPT_BUFFER_MATERIALS) {
        if (bytes > stride * GMUPT_MAX_LIGHTS) return fail(GMUPT_ERR_INVALID_ARGUMENT, "gmupt_buffer_create: more than %d entries", GMUPT_MAX_LIGHTS);
        alloc = stride * GMUPT_MAX_LIGHTS;
    }
    if (alloc == 0) alloc = stride;
    gmupt_buffer* b = new (std::nothrow) gmupt_buffer();
    if (!b) return fail(GMUPT_ERR_OUT_OF_MEMORY, "gmupt_buffer_create: out of host memory");
    b->dev = dev; b->kind = kind; b->bytes = alloc; b->elems = bytes / stride; b->dptr = nullptr;
    hipError_t e = hipSetDevice(dev->id);
    if (e == hipSuccess) e = hipMalloc(&b->dptr, alloc + 16); // +16: 12-byte vertices are read with in-bounds dword loads only, slack is for safety
    if (e == hipSuccess) e = hipMemset(b->dptr, 0, alloc + 16);
    if (e == hipSuccess && bytes) e = hipMemcpy(b->dptr, data, bytes, hipMemcpyHostToDevice);
    if (e != hipSuccess) { if (b->dptr) (void)hipFree(b->dptr); delete b; return fail(GMUPT_ERR_HIP, "gmupt_buffer_create(%zu bytes): %s", alloc, hipGetErrorString(e)); }
    *out = b;
    return GMUPT_OK;
}

extern "C" int gmupt_image_decode_png(const void* png, size_t bytes, uint32_t* width, uint32_t* height, uint8_t** rgba)
{
    if (!png || !width || !height || !rgba) return fail(GMUPT_ERR_INVALID_ARGUMENT, "gmupt_image_decode_png: null argument");
    *rgba = nullptr; *width = *height = 0;
    try {
        gmupt::png::Image img = gmupt::png::decode(static_cast<const uint8_t*>(png), bytes);
        uint8_t* mem = static_cast<uint8_t*>(std::malloc(img.rgba.size()));
        if (!mem) return fail(GMUPT_ERR_OUT_OF_MEMORY, "gmupt_image_decode_png: out of host memory");
        std::memcpy(mem, img.rgba.data(), img.rgba.size());
        *rgba = mem; *width = img.width; *height = img.height;
    } catch (const std::exception& e) { return fail(GMUPT_ERR_INVALID_ARGUMENT, "gmupt_image_decode_png: %s", e.what()); }
    return GMUPT_OK;
}

extern "C" void gmupt_image_free(uint8_t* rgba) { std::free(rgba); }

extern "C" int gmupt_image_resize_square(const uint8_t* rgba, uint32_t old_size, uint32_t new_size, uint8_t* dst)
{
    if (!rgba || !dst || old_size == 0 || new_size == 0) return fail(GMUPT_ERR_INVALID_ARGUMENT, "gmupt_image_resize_square: null or empty argument");
    try {
        const std::vector<uint8_t> out = gmupt::resizeSquare(rgba, old_size, new_size);
        std::memcpy(dst, out.data(), out.size());
    } catch (const std::exception& e) { return fail(GMUPT_ERR_INVALID_ARGUMENT, "gmupt_image_resize_square: %s", e.what()); }
    return GMUPT_OK;
}

extern "C" uint32_t gmupt_texture_common_size(const size_t* layer_bytes, uint32_t layers)
{
    if (!layer_bytes || layers == 0) return 0;
    return gmupt::commonDimension(std::vector<size_t>(layer_bytes, layer_bytes + layers));
}

extern "C" int gmupt_texture_array_create(gmupt_device* dev, const uint8_t* rgba8, uint32_t size, uint32_t layers, gmupt_buffer** out)
{
    if (!dev || !out || !rgba8) return fail(GMUPT_ERR_INVALID_ARGUMENT, "gmupt_texture_array_create: null argument");
    if (size == 0 || layers == 0 || size > 16384) return fail(GMUPT_ERR_INVALID_ARGUMENT, "gmupt_texture_array_create: %u layers of %ux%u", layers, size, size);
    int rc = gmupt_buffer_create(dev, GMUPT_BUFFER_TEXTURE_ARRAY, rgba8, (size_t)size * size * layers * 4, out);
    if (rc == GMUPT_OK) { (*out)->texSize = size; (*out)->texLayers = layers; }
    return rc;
}

extern "C" int gmupt_buffer_update(gmupt_buffer* buf, const void* data, size_t bytes)
{
    if (!buf || (!data && bytes)) return fail(GMUPT_ERR_INVALID_ARGUMENT, "gmupt_buffer_update: null argument");
    if (bytes > buf->bytes) return fail(GMUPT_ERR_INVALID_ARGUMENT, "gmupt_buffer_update: %zu bytes into a %zu-byte buffer", bytes, buf->bytes);
    HIP_TRY(hipSetDevice(buf->dev->id));
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(buf->dptr, data, bytes, hipMemcpyHostToDevice));
    return GMUPT_OK;
}

extern "C" void gmupt_buffer_destroy(gmupt_buffer* buf)
{
    if (!buf) return;
    (void)hipSetDevice(buf->dev->id);
    (void)hipFree(buf->dptr);
    delete buf;
}

extern "C" size_t gmupt_buffer_size(const gmupt_buffer* buf) { return buf ? buf->bytes : 0; }

// ------------------------------------------------------------------------------------------------ renderer
struct StageEvents { hipEvent_t e[5]; bool extOnly = false; };   // logic | material | ray cast (extension) | shadow

struct gmupt_renderer {
    gmupt_device* dev = nullptr;
    gmupt_renderer_desc desc{};
    hipStream_t stream = nullptr;
    RenderParams p{};
    bool sceneBound = false, cameraSet = false;
    uint64_t iterations = 0;
    uint32_t travBlocks = 0;
    // timing
    int timing = 0; // 0 off, 1 all stages, 2 only the extension ray cast (two events per iteration)
    std::vector<StageEvents> evPool; size_t evUsed = 0;
    double msStage[4] = { 0, 0, 0, 0 }; uint64_t timedIters = 0;
    std::vector<void*> allocs;
    // packed traversal copy of the bound scene
    void* travNodes = nullptr; void* travTris = nullptr; void* travRecs = nullptr; void* travWide = nullptr; void* travPairs = nullptr; void* travPairRef = nullptr;
    int travMode = 70; // GMUPT_TRAVERSAL: "wide" (default) both ray casts in one launch over the 4-wide collapse | "cast0" the same over the binary tree | "def0" separate launches; the other rungs of the ladder exist in -DGMUPT_VARIANTS builds only
    uint32_t castFlags = 0; // GMUPT_STAT_* bits of the ray-cast kernels launched since the last reset
};

static int dev_alloc(gmupt_renderer* r, void** ptr, size_t bytes, int fill)
{
    *ptr = nullptr;
    HIP_TRY(hipMalloc(ptr, bytes ? bytes : 16));
    r->allocs.push_back(*ptr);
    HIP_TRY(hipMemsetAsync(*ptr, fill, bytes ? bytes : 16, r->stream));
    return GMUPT_OK;
}

static int alloc_framebuffer(gmupt_renderer* r, uint32_t w, uint32_t h)
{
    void* fb = nullptr; void* head = nullptr;
    const size_t npix = (size_t)w * h;
    HIP_TRY(hipMalloc(&fb, npix * 16 + 16));
    HIP_TRY(hipMalloc(&head, npix * 4 + 16));
    HIP_TRY(hipMemsetAsync(fb, 0, npix * 16 + 16, r->stream));        // createRenderTexture: no initial data => zero
    HIP_TRY(hipMemsetAsync(head, 0xFF, npix * 4 + 16, r->stream));
    r->p.fb = (float4*)fb; r->p.listHead = (uint32_t*)head; r->p.fbW = w; r->p.fbH = h;
    return GMUPT_OK;
}

extern "C" void gmupt_renderer_destroy(gmupt_renderer* r)
{
    if (!r) return;
    (void)hipSetDevice(r->dev->id);
    if (r->stream) (void)hipStreamSynchronize(r->stream);
    for (auto& se : r->evPool) for (auto& e : se.e) (void)hipEventDestroy(e);
    for (void* a : r->allocs) (void)hipFree(a);
    if (r->p.fb) (void)hipFree(r->p.fb);
    if (r->p.listHead) (void)hipFree(r->p.listHead);
    if (r->travNodes) (void)hipFree(r->travNodes);
    if (r->travTris) (void)hipFree(r->travTris);
    if (r->travRecs) (void)hipFree(r->travRecs);
    if (r->travWide) (void)hipFree(r->travWide);
    if (r->travPairs) (void)hipFree(r->travPairs);
    if (r->travPairRef) (void)hipFree(r->travPairRef);
    if (r->stream) (void)hipStreamDestroy(r->stream);
    delete r;
}

// GMUPT_TRAVERSAL selects a rung of the traversal ladder; all rungs give identical results, the default is the fastest
static int parse_traversal_mode(const char* tv)
{
    constexpr int kDefault = 70;                                            // wide: both ray casts in one launch over the 4-wide collapse of the tree (cast0, the binary fused kernel, takes what it does not)
    if (!tv || !*tv) return kDefault;
    if (std::strcmp(tv, "whilewhile") == 0) return 0;
    if (std::strcmp(tv, "ref") == 0) return 1;
    if (std::strcmp(tv, "static") == 0) return 2;
    if (std::strncmp(tv, "ifif", 4) == 0) return 3 + std::atoi(tv + 4);
    if (std::strcmp(tv, "coop") == 0) return 20;
    if (std::strcmp(tv, "top") == 0) return 30;
    if (std::strncmp(tv, "def", 3) == 0) return 40 + std::atoi(tv + 3);     // separate deferred-leaf launches
    if (std::strncmp(tv, "pipe", 4) == 0) return 50 + std::atoi(tv + 4);    // three-slot lane pipeline
    if (std::strncmp(tv, "cast", 4) == 0) return 60 + std::atoi(tv + 4);    // cast0 mixed lanes + fused fetches, cast1 extension then shadow per wave, cast2 mixed lanes
    if (std::strcmp(tv, "wide") == 0) return 70;                            // both ray casts in one launch over the 4-wide collapse of the tree (pt_traverse_wide.hip)
    return kDefault;
}

extern "C" int gmupt_renderer_create(gmupt_device* dev, const gmupt_renderer_desc* desc, gmupt_renderer** out)
{
    if (!dev || !desc || !out) return fail(GMUPT_ERR_INVALID_ARGUMENT, "gmupt_renderer_create: null argument");
    *out = nullptr;
    if (desc->width == 0 || desc->height == 0) return fail(GMUPT_ERR_INVALID_ARGUMENT, "gmupt_renderer_create: empty accumulation target %ux%u", desc->width, desc->height);
    gmupt_renderer* r = new (std::nothrow) gmupt_renderer();
    if (!r) return fail(GMUPT_ERR_OUT_OF_MEMORY, "gmupt_renderer_create: out of host memory");
    r->dev = dev; r->desc = *desc;
    r->travMode = parse_traversal_mode(std::getenv("GMUPT_TRAVERSAL"));
    if (!traversal_mode_available(r->travMode)) {
        delete r;
        return fail(GMUPT_ERR_UNSUPPORTED, "gmupt_renderer_create: GMUPT_TRAVERSAL=%s is not part of this build (wide, cast0 and def0 are; the other rungs need -DGMUPT_VARIANTS)", std::getenv("GMUPT_TRAVERSAL"));
    }
    if (r->desc.pool_paths == 0) r->desc.pool_paths = GMUPT_PATHCOUNT;
    if (r->desc.live_paths == 0 || r->desc.live_paths > r->desc.pool_paths) r->desc.live_paths = r->desc.pool_paths;
    const uint32_t P = r->desc.pool_paths, L = r->desc.live_paths;
    if ((uint64_t)P * F_COUNT * 4ull > (200ull << 30)) { delete r; return fail(GMUPT_ERR_INVALID_ARGUMENT, "gmupt_renderer_create: pool of %u paths is too large", P); }

    hipError_t e = hipSetDevice(dev->id);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&r->stream, hipStreamNonBlocking);
    if (e != hipSuccess) { delete r; return fail(GMUPT_ERR_HIP, "gmupt_renderer_create: %s", hipGetErrorString(e)); }

    RenderParams& p = r->p;
    p.P = P; p.L = L;
    p.nBlocks = (L + kBlock - 1) / kBlock;
    p.tileEnabled = desc->tile_enabled; p.tileX0 = desc->tile_x0; p.tileY0 = desc->tile_y0;
    p.budget = desc->path_budget; p.maxDepth = desc->max_depth;
    const uint32_t tb = traversal_block_threads();
    r->travBlocks = (L + tb - 1) / tb;
    p.ovfStride = r->travBlocks * tb;
    if (p.ovfStride < deferred_block_threads()) p.ovfStride = deferred_block_threads();
    { const char* rw = std::getenv("GMUPT_RAYS_PER_WAVE"); p.raysPerWave = rw ? (uint32_t)std::atoi(rw) : 128u; if (p.raysPerWave < 64) p.raysPerWave = 64; if (r->travMode >= 60 && p.raysPerWave > 128) p.raysPerWave = 128; /* the fused kernel keeps a chunk in two registers per lane */ }
    { const char* wpc = std::getenv("GMUPT_WAVES_PER_CU"); const uint32_t w = wpc ? (uint32_t)std::atoi(wpc) : 16u; const uint32_t db = deferred_block_threads(); p.travGridBlocks = (uint32_t)dev->prop.multiProcessorCount * ((w * 64 + db - 1) / db); if (p.travGridBlocks * db > p.ovfStride) p.travGridBlocks = p.ovfStride / db; if (p.travGridBlocks == 0) p.travGridBlocks = 1; }
    { const char* ep = std::getenv("GMUPT_EXTEND_PRUNE"); p.extendPrune = ep ? (uint32_t)std::atoi(ep) : 0u; }
    { const char* sp = std::getenv("GMUPT_SHADOW_PRUNE"); p.shadowPrune = sp ? (uint32_t)std::atoi(sp) : 0u; }
    { const char* ws = std::getenv("GMUPT_WIDE_STEPS"); p.tuneWideSteps = ws ? (uint32_t)std::atoi(ws) : 0u; }
    { const char* xb = std::getenv("GMUPT_XCD_BINS"); p.xcdBins = xb ? (uint32_t)std::atoi(xb) : 0u; }
    { const char* lc = std::getenv("GMUPT_CAST_LOOP_CAP"); p.castLoopCap = lc ? (uint32_t)std::atoi(lc) : (1u << 20); if (p.castLoopCap == 0) p.castLoopCap = 1u << 20; }
    { const char* e1 = std::getenv("GMUPT_REFILL"); p.tuneRefill = e1 ? (uint32_t)std::atoi(e1) : 20u; const char* e2 = std::getenv("GMUPT_TRI_THRESH"); p.tuneTriThresh = e2 ? (uint32_t)std::atoi(e2) : ((r->travMode == 60 || r->travMode == 63 || r->travMode == 70) ? 24u : 32u); } // fused fetches make a burst cheaper

    int rc = GMUPT_OK;
    // Renderer::createBuffers creates the UAV buffers without initial data: D3D11 zero-initialises them
    // the fields of the path state 17 x 256 bytes further apart than the pool size: with P a power of two, the ~50 streams a stage reads and
    // writes would otherwise all be at the same point of the HBM channel rotation (k_logic / k_material: -2 to -3 % on config 3)
    { const char* pad = std::getenv("GMUPT_STATE_PAD"); p.PS = P + (pad ? (uint32_t)std::atoi(pad) & ~63u : 1088u); }
    if (rc == GMUPT_OK) rc = dev_alloc(r, (void**)&p.state, (size_t)F_COUNT * p.PS * 4, 0);
    if (rc == GMUPT_OK) rc = dev_alloc(r, (void**)&p.cls, (size_t)P, CLS_ENDED);
    if (rc == GMUPT_OK) rc = dev_alloc(r, (void**)&p.listNext, (size_t)P * 4, 0xFF);
    if (rc == GMUPT_OK) rc = dev_alloc(r, (void**)&p.sample, (size_t)P * 12, 0);
    if (rc == GMUPT_OK) rc = dev_alloc(r, (void**)&p.blockCounts, (size_t)p.nBlocks * 4 * kNumCounts, 0);
    p.nGroups = (p.nBlocks + kScanGroup - 1) / kScanGroup; p.groupParity = 0;
    if (rc == GMUPT_OK) rc = dev_alloc(r, (void**)&p.groupTotals, (size_t)2 * kNumCounts * p.nGroups * 4, 0);
    if (rc == GMUPT_OK) rc = dev_alloc(r, (void**)&p.queues, (size_t)P * 20, 0);
    if (rc == GMUPT_OK) rc = dev_alloc(r, (void**)&p.qc, 32, 0);
    if (rc == GMUPT_OK) rc = dev_alloc(r, (void**)&p.stats, sizeof(DevStats), 0);
    if (rc == GMUPT_OK) rc = dev_alloc(r, (void**)&p.travCounters, 128, 0);
    if (rc == GMUPT_OK) rc = dev_alloc(r, (void**)&p.ovfStack, (size_t)p.ovfStride * traversal_overflow_entries() * 4, 0);
    if (rc == GMUPT_OK) rc = alloc_framebuffer(r, desc->width, desc->height);
    if (rc == GMUPT_OK) {
        DevStats init{}; init.activePaths = L;
        e = hipMemcpyAsync(p.stats, &init, sizeof(init), hipMemcpyHostToDevice, r->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(r->stream);
        if (e != hipSuccess) rc = fail(GMUPT_ERR_HIP, "gmupt_renderer_create: %s", hipGetErrorString(e));
    }
    if (rc != GMUPT_OK) { std::string keep = g_lastError; gmupt_renderer_destroy(r); g_lastError = keep; return rc; }
    *out = r;
    return GMUPT_OK;
}

// Packs the reference-layout BVH into the traversal records of pt_device.hpp (Node64 / Tri48).  Host-side, once per bind.
static int build_traversal_copy(gmupt_renderer* r, const gmupt_buffer* nodesB, const gmupt_buffer* trisB, const gmupt_buffer* vertsB)
{
    HIP_TRY(hipSetDevice(r->dev->id));
    HIP_TRY(hipStreamSynchronize(r->stream));
    const size_t N = nodesB->elems, R = trisB->elems, V = vertsB->elems;
    std::vector<gmupt_bvh_node> nodes(N);
    std::vector<gmupt_triangle> tris(R ? R : 1);
    std::vector<float> verts(V ? V * 3 : 3);
    HIP_TRY(hipMemcpy(nodes.data(), nodesB->dptr, N * sizeof(gmupt_bvh_node), hipMemcpyDeviceToHost));
    if (R) HIP_TRY(hipMemcpy(tris.data(), trisB->dptr, R * sizeof(gmupt_triangle), hipMemcpyDeviceToHost));
    if (V) HIP_TRY(hipMemcpy(verts.data(), vertsB->dptr, V * 12, hipMemcpyDeviceToHost));

    // validate what the kernels will index with (a malformed tree must not become an out-of-bounds access on the GPU)
    std::vector<int32_t> innerIndex(N, -1);
    int32_t numInner = 0;
    for (size_t i = 0; i < N; i++) {
        const gmupt_bvh_node& n = nodes[i];
        if (n.isLeaf) {
            if (n.left < 0 || n.right < n.left || (size_t)n.right > R) return fail(GMUPT_ERR_INVALID_ARGUMENT, "bind_scene: leaf %zu has triangle range [%d, %d) outside [0, %zu)", i, n.left, n.right, R);
        } else {
            if (n.left <= (int32_t)i || n.right <= (int32_t)i || (size_t)n.left >= N || (size_t)n.right >= N) return fail(GMUPT_ERR_INVALID_ARGUMENT, "bind_scene: inner node %zu has children (%d, %d) outside (%zu, %zu)", i, n.left, n.right, i, N);
            numInner++;
        }
    }
    std::vector<int32_t> depth(N, 0);
    int32_t maxDepth = 0;
    for (size_t i = 0; i < N; i++) if (!nodes[i].isLeaf) { depth[(size_t)nodes[i].left] = depth[i] + 1; depth[(size_t)nodes[i].right] = depth[i] + 1; }
    for (size_t i = 0; i < N; i++) maxDepth = std::max(maxDepth, depth[i]);
    // packed numbering: first the inner nodes that the ray-cast kernels keep in LDS (the part of the tree every ray walks), then the
    // remaining inner nodes.  How many fit depends on the kernel instantiation this tree will run: a tree that needs the spilling stack
    // keeps fewer stack entries and more nodes in LDS (pt_traverse_deferred.hpp: kDefLdsStack / kDefLdsTop)
    const size_t topCapacity = traversal_top_capacity((uint32_t)maxDepth);
    {
        // the LDS-resident set grows from the root by always expanding the frontier node with the largest surface area (the usual
        // visit-probability estimate); GMUPT_TOP_ORDER=bfs selects plain breadth-first order (0.5 % slower on the bench scene)
        std::vector<int32_t> bfs; bfs.reserve(topCapacity);
        const char* order = std::getenv("GMUPT_TOP_ORDER");
        if (!(order && std::strcmp(order, "bfs") == 0)) {
            auto area = [&](int32_t i) { const gmupt_bvh_node& n = nodes[(size_t)i]; const double dx = (double)n.max[0] - n.min[0], dy = (double)n.max[1] - n.min[1], dz = (double)n.max[2] - n.min[2]; return dx * dy + dy * dz + dz * dx; };
            std::vector<std::pair<double, int32_t>> frontier;
            if (!nodes[0].isLeaf) frontier.push_back({ area(0), 0 });
            while (!frontier.empty() && bfs.size() < topCapacity) {
                size_t best = 0;
                for (size_t k = 1; k < frontier.size(); k++) if (frontier[k].first > frontier[best].first || (frontier[k].first == frontier[best].first && frontier[k].second < frontier[best].second)) best = k;
                const int32_t v = frontier[best].second;
                frontier.erase(frontier.begin() + (long)best);
                bfs.push_back(v);
                const gmupt_bvh_node& n = nodes[(size_t)v];
                if (!nodes[(size_t)n.left].isLeaf) frontier.push_back({ area(n.left), n.left });
                if (!nodes[(size_t)n.right].isLeaf) frontier.push_back({ area(n.right), n.right });
            }
        } else {
            if (!nodes[0].isLeaf) bfs.push_back(0);
            for (size_t h = 0; h < bfs.size() && bfs.size() < topCapacity; h++) {
                const gmupt_bvh_node& n = nodes[(size_t)bfs[h]];
                if (!nodes[(size_t)n.left].isLeaf && bfs.size() < topCapacity) bfs.push_back(n.left);
                if (!nodes[(size_t)n.right].isLeaf && bfs.size() < topCapacity) bfs.push_back(n.right);
            }
        }
        int32_t nextIdx = 0;
        for (int32_t v : bfs) innerIndex[(size_t)v] = nextIdx++;
        r->p.trav.topCountDeep = (uint32_t)bfs.size();                                        // what the spilling-stack instantiations keep in LDS
        r->p.trav.topCount = (uint32_t)std::min(bfs.size(), (size_t)kTopTreeNodes);             // what every other kernel keeps (a prefix of the same order)
        // The rest of the inner nodes.  Every memory-side read of the ray cast is a whole 128-byte line (TCC_EA0_RDREQ_128B is all of
        // TCC_EA0_RDREQ: profiles/r02_micro/fetch_size_calibration.txt), i.e. TWO 64-byte records.  A node therefore shares its line with the
        // inner child a ray is most likely to visit next (the one with the larger surface area): that visit then finds its record in the
        // cache.  Nodes without such a partner share a line with the next one of their kind in flatten order (usually a sibling or cousin).
        // GMUPT_NODE_PAIRING=0 keeps the plain flatten order (A/B timing; results do not depend on the numbering).
        const char* pairing = std::getenv("GMUPT_NODE_PAIRING");
        if (pairing && std::atoi(pairing) == 0) {
            for (size_t i = 0; i < N; i++) if (!nodes[i].isLeaf && innerIndex[i] < 0) innerIndex[i] = nextIdx++;
        } else {
            auto area = [&](int32_t i) { const gmupt_bvh_node& n = nodes[(size_t)i]; const double dx = (double)n.max[0] - n.min[0], dy = (double)n.max[1] - n.min[1], dz = (double)n.max[2] - n.min[2]; return dx * dy + dy * dz + dz * dx; };
            if (nextIdx & 1) nextIdx++;                                   // lines start at even records (an unused record keeps the parity)
            std::vector<int32_t> partner(N, -1), singles;
            std::vector<uint8_t> taken(N, 0);
            for (size_t i = 0; i < N; i++) {                              // parents come before their children in the reference numbering
                if (nodes[i].isLeaf || innerIndex[i] >= 0 || taken[i]) continue;
                const int32_t l = nodes[i].left, rr = nodes[i].right;
                const bool li = !nodes[(size_t)l].isLeaf && innerIndex[(size_t)l] < 0, ri = !nodes[(size_t)rr].isLeaf && innerIndex[(size_t)rr] < 0;
                int32_t c = -1;
                if (li && ri) c = area(l) >= area(rr) ? l : rr; else if (li) c = l; else if (ri) c = rr;
                if (c >= 0) { partner[i] = c; taken[(size_t)c] = 1; } else singles.push_back((int32_t)i);
            }
            for (size_t i = 0; i < N; i++) if (partner[i] >= 0) { innerIndex[i] = nextIdx++; innerIndex[(size_t)partner[i]] = nextIdx++; }
            for (int32_t v : singles) innerIndex[(size_t)v] = nextIdx++;
        }
        numInner = nextIdx;   // records of the packed array (one may be an unused filler)
    }
    for (size_t i = 0; i < R; i++) {
        for (int k = 0; k < 3; k++)
            if (tris[i].v[k] < 0 || (size_t)tris[i].v[k] >= V) return fail(GMUPT_ERR_INVALID_ARGUMENT, "bind_scene: triangle record %zu references vertex %d of %zu", i, tris[i].v[k], V);
        if (tris[i].materialID >= (uint32_t)GMUPT_MAX_LIGHTS) return fail(GMUPT_ERR_INVALID_ARGUMENT, "bind_scene: triangle record %zu has material %u (the material table holds %d entries, logic.hlsl:8)", i, tris[i].materialID, GMUPT_MAX_LIGHTS);
    }

    auto desc = [&](int32_t child) -> int32_t {
        const gmupt_bvh_node& c = nodes[(size_t)child];
        if (!c.isLeaf) return innerIndex[(size_t)child];
        // an empty leaf cannot be expressed by "first record + last flag": point it at a degenerate sentinel record
        return ~(c.right > c.left ? c.left : (int32_t)R);
    };
    std::vector<Node64> packed((size_t)numInner ? (size_t)numInner : 1);
    std::memset(packed.data(), 0, packed.size() * sizeof(Node64));
    for (size_t i = 0; i < N; i++) {
        if (nodes[i].isLeaf) continue;
        const gmupt_bvh_node& L = nodes[(size_t)nodes[i].left];
        const gmupt_bvh_node& Rn = nodes[(size_t)nodes[i].right];
        Node64& o = packed[(size_t)innerIndex[i]];
        o.a[0] = L.min[0]; o.a[1] = L.min[1]; o.a[2] = L.min[2]; o.a[3] = L.max[0];
        o.b[0] = L.max[1]; o.b[1] = L.max[2]; o.b[2] = Rn.min[0]; o.b[3] = Rn.min[1];
        o.c[0] = Rn.min[2]; o.c[1] = Rn.max[0]; o.c[2] = Rn.max[1]; o.c[3] = Rn.max[2];
        o.d[0] = desc(nodes[i].left); o.d[1] = desc(nodes[i].right); o.d[2] = depth[i]; o.d[3] = 0;
    }
    std::vector<Tri48> ptris(R + 1);
    std::memset(ptris.data(), 0, ptris.size() * sizeof(Tri48));
    for (size_t i = 0; i < R; i++) {
        const float* v0 = &verts[3 * (size_t)tris[i].v[0]]; const float* v1 = &verts[3 * (size_t)tris[i].v[1]]; const float* v2 = &verts[3 * (size_t)tris[i].v[2]];
        Tri48& t = ptris[i];
        t.r0[0] = v0[0]; t.r0[1] = v0[1]; t.r0[2] = v0[2];
        t.r0[3] = v1[0] - v0[0]; t.r1[0] = v1[1] - v0[1]; t.r1[1] = v1[2] - v0[2];   // e1 = v1 - v0 (extensionRayCast.hlsl:40)
        t.r1[2] = v2[0] - v0[0]; t.r1[3] = v2[1] - v0[1]; t.r2[0] = v2[2] - v0[2];   // e2 = v2 - v0 (:41)
    }
    const uint32_t one = 1u;
    for (size_t i = 0; i < N; i++)
        if (nodes[i].isLeaf && nodes[i].right > nodes[i].left) std::memcpy(&ptris[(size_t)nodes[i].right - 1].r2[1], &one, 4);
    std::memcpy(&ptris[R].r2[1], &one, 4); // sentinel: all-zero triangle (det = 0, rejected), last flag set

    // word 10 of a triangle record: the number of the first reference with the same (v0, v1, v2, material) -- duplicated references of one
    // triangle (spatial splits) produce identical hit records, so a tie in t between them is no tie (pt_traverse_wide.hip)
    const bool wantWide = r->travMode == 70;
    if (wantWide) {
        struct Key { int32_t v[3]; uint32_t mat; uint32_t idx; };
        std::vector<Key> keys(R);
        for (size_t i = 0; i < R; i++) keys[i] = { { tris[i].v[0], tris[i].v[1], tris[i].v[2] }, tris[i].materialID, (uint32_t)i };
        std::sort(keys.begin(), keys.end(), [](const Key& a, const Key& b) {
            if (a.v[0] != b.v[0]) return a.v[0] < b.v[0];
            if (a.v[1] != b.v[1]) return a.v[1] < b.v[1];
            if (a.v[2] != b.v[2]) return a.v[2] < b.v[2];
            if (a.mat != b.mat) return a.mat < b.mat;
            return a.idx < b.idx; });
        for (size_t i = 0; i < R;) {
            size_t j = i;
            while (j < R && keys[j].v[0] == keys[i].v[0] && keys[j].v[1] == keys[i].v[1] && keys[j].v[2] == keys[i].v[2] && keys[j].mat == keys[i].mat) {
                std::memcpy(&ptris[keys[j].idx].r2[2], &keys[i].idx, 4);
                j++;
            }
            i = j;
        }
        const uint32_t none = 0xFFFFFFFFu;
        std::memcpy(&ptris[R].r2[2], &none, 4);
    }

    // 4-wide collapse (WNode, pt_device.hpp): the two children of an inner node, the inner one with the largest surface area replaced by
    // ITS children until four slots are taken; every inner slot becomes a wide node in turn.  Only built when every child box lies inside
    // its parent's box (what a bounding-volume hierarchy is; the wide walk's equivalence to the binary one rests on it).
    // the leaves as triangle pairs (TriPair, pt_device.hpp): consecutive references of a leaf two by two
    std::vector<TriPair> pairs;
    std::vector<uint32_t> pairRef;
    std::vector<int32_t> leafPair(N, -1);       // first pair of every leaf node
    if (wantWide) {
        pairs.reserve(R / 2 + N / 2 + 2); pairRef.reserve(R + N + 4);
        auto put = [&](TriPair& pr, int slot, size_t ref) {
            const Tri48& t = ptris[ref];
            const float c[9] = { t.r0[0], t.r0[1], t.r0[2], t.r0[3], t.r1[0], t.r1[1], t.r1[2], t.r1[3], t.r2[0] };   // v0.xyz, e1.xyz, e2.xyz
            for (int k = 0; k < 9; k++) pr.w[2 * k + slot] = c[k];
        };
        const uint32_t one32 = 1u;
        for (size_t i = 0; i < N; i++) {
            if (!nodes[i].isLeaf) continue;
            leafPair[i] = (int32_t)pairs.size();
            const int32_t a = nodes[i].left, b = nodes[i].right;
            for (int32_t k = a; k < b || k == a; k += 2) {          // (an empty leaf gets one all-zero pair)
                TriPair pr; std::memset(&pr, 0, sizeof(pr));
                uint32_t r0 = 0xFFFFFFFFu, r1 = 0xFFFFFFFFu;
                if (k < b) { put(pr, 0, (size_t)k); r0 = (uint32_t)k; }
                if (k + 1 < b) { put(pr, 1, (size_t)k + 1); r1 = (uint32_t)k + 1; }
                if (k + 2 >= b) std::memcpy(&pr.w[18], &one32, 4);
                const uint32_t nrefs = (k < b ? 1u : 0u) + (k + 1 < b ? 1u : 0u); std::memcpy(&pr.w[19], &nrefs, 4);   // (statistics)
                pairs.push_back(pr); pairRef.push_back(r0); pairRef.push_back(r1);
            }
        }
    }

    std::vector<WNode> wide;
    uint32_t wideTop = 0, wideBound = 0;
    bool contained = true;
    for (size_t i = 0; i < N && contained && wantWide; i++) {
        if (nodes[i].isLeaf) continue;
        for (int32_t c : { nodes[i].left, nodes[i].right })
            for (int k = 0; k < 3; k++)
                if (!(nodes[(size_t)c].min[k] >= nodes[i].min[k] && nodes[(size_t)c].max[k] <= nodes[i].max[k] && nodes[(size_t)c].min[k] <= nodes[(size_t)c].max[k])) contained = false;
    }
    if (wantWide && contained && !nodes[0].isLeaf) {
        auto area = [&](int32_t i) { const gmupt_bvh_node& n = nodes[(size_t)i]; const double dx = (double)n.max[0] - n.min[0], dy = (double)n.max[1] - n.min[1], dz = (double)n.max[2] - n.min[2]; return dx * dy + dy * dz + dz * dx; };
        // Opening a slot drops ITS box test for the rays that reach its children.  "Child hit implies parent hit" holds for every ray
        // unless a child is flat on an axis on which the parent is not, in the plane of one of the parent's faces (a ray with d = 0 on that
        // axis that starts in this plane gets NaNs from the child's two planes -- no condition -- and +-inf from the parent's: a miss;
        // pt_traverse_wide.hip).  Such a node keeps its own slot.
        auto opens = [&](int32_t c) {
            const gmupt_bvh_node& pn = nodes[(size_t)c];
            for (int32_t x : { pn.left, pn.right }) {
                const gmupt_bvh_node& cn = nodes[(size_t)x];
                for (int k = 0; k < 3; k++)
                    if (cn.min[k] == cn.max[k] && pn.min[k] != pn.max[k] && (cn.min[k] == pn.min[k] || cn.max[k] == pn.max[k])) return false;
            }
            return true;
        };
        struct Slots { int32_t s[4]; int n; int32_t bin; };
        std::vector<Slots> created;                    // creation order: parents before children
        std::vector<int32_t> createdOf(N, -1);
        std::vector<int32_t> todo{ 0 };
        while (!todo.empty()) {
            const int32_t v = todo.back(); todo.pop_back();
            Slots w; w.bin = v; w.n = 2; w.s[0] = nodes[(size_t)v].left; w.s[1] = nodes[(size_t)v].right; w.s[2] = w.s[3] = -1;
            while (w.n < 4) {
                int best = -1;
                for (int k = 0; k < w.n; k++) if (!nodes[(size_t)w.s[k]].isLeaf && opens(w.s[k]) && (best < 0 || area(w.s[k]) > area(w.s[best]))) best = k;
                if (best < 0) break;
                const int32_t c = w.s[best];
                for (int k = w.n; k > best + 1; k--) w.s[k] = w.s[k - 1];
                w.s[best] = nodes[(size_t)c].left; w.s[best + 1] = nodes[(size_t)c].right; w.n++;
            }
            createdOf[(size_t)v] = (int32_t)created.size();
            created.push_back(w);
            for (int k = w.n - 1; k >= 0; k--) if (!nodes[(size_t)w.s[k]].isLeaf) todo.push_back(w.s[k]);
        }
        const size_t W = created.size();
        // numbering: the LDS-resident top first (grown from the root, largest surface area first), then creation order (depth-first)
        std::vector<int32_t> number(W, -1);
        int32_t nextW = 0;
        {
            std::vector<std::pair<double, int32_t>> frontier{ { area(0), 0 } };
            const size_t cap = traversal_wide_top_capacity();
            while (!frontier.empty() && (size_t)nextW < cap) {
                size_t best = 0;
                for (size_t k = 1; k < frontier.size(); k++) if (frontier[k].first > frontier[best].first || (frontier[k].first == frontier[best].first && frontier[k].second < frontier[best].second)) best = k;
                const int32_t c = frontier[best].second;
                frontier.erase(frontier.begin() + (long)best);
                number[(size_t)c] = nextW++;
                const Slots& w = created[(size_t)c];
                for (int k = 0; k < w.n; k++) if (!nodes[(size_t)w.s[k]].isLeaf) frontier.push_back({ area(w.s[k]), createdOf[(size_t)w.s[k]] });
            }
            wideTop = (uint32_t)nextW;
        }
        for (size_t c = 0; c < W; c++) if (number[c] < 0) number[c] = nextW++;
        wide.resize(W);
        const float qnan = std::numeric_limits<float>::quiet_NaN();
        for (size_t c = 0; c < W; c++) {
            const Slots& w = created[c];
            WNode& o = wide[(size_t)number[c]];
            for (int k = 0; k < 4; k++) {
                if (k < w.n) {
                    const gmupt_bvh_node& b = nodes[(size_t)w.s[k]];
                    for (int a = 0; a < 3; a++) { o.p[a][k] = b.min[a]; o.p[5 - a][k] = b.max[a]; }   // rows: min x, y, z, max z, y, x
                    o.link[k] = b.isLeaf ? ~leafPair[(size_t)w.s[k]] : number[(size_t)createdOf[(size_t)w.s[k]]];
                } else {
                    for (int a = 0; a < 6; a++) o.p[a][k] = qnan;      // never hit
                    o.link[k] = (int32_t)0x80000000;
                }
            }
            o.aux[0] = depth[(size_t)w.bin]; o.aux[1] = w.n; o.aux[2] = o.aux[3] = 0;
        }
        // most entries the inner stack of a walk can hold: every inner slot hit on every level, the deepest child visited last
        std::vector<uint32_t> occ(W, 0);
        for (size_t c = W; c-- > 0;) {
            const Slots& w = created[c];
            uint32_t inner = 0, deepest = 0;
            for (int k = 0; k < w.n; k++) if (!nodes[(size_t)w.s[k]].isLeaf) { inner++; deepest = std::max(deepest, occ[(size_t)createdOf[(size_t)w.s[k]]]); }
            occ[c] = inner ? inner - 1 + deepest : 0;
        }
        wideBound = occ[0];
    }
    if (r->travWide) { HIP_TRY(hipFree(r->travWide)); r->travWide = nullptr; }
    if (!wide.empty()) {
        HIP_TRY(hipMalloc(&r->travWide, wide.size() * sizeof(WNode)));
        HIP_TRY(hipMemcpy(r->travWide, wide.data(), wide.size() * sizeof(WNode), hipMemcpyHostToDevice));
    }
    if (r->travPairs) { HIP_TRY(hipFree(r->travPairs)); r->travPairs = nullptr; }
    if (r->travPairRef) { HIP_TRY(hipFree(r->travPairRef)); r->travPairRef = nullptr; }
    if (!wide.empty()) {
        HIP_TRY(hipMalloc(&r->travPairs, pairs.size() * sizeof(TriPair)));
        HIP_TRY(hipMemcpy(r->travPairs, pairs.data(), pairs.size() * sizeof(TriPair), hipMemcpyHostToDevice));
        HIP_TRY(hipMalloc(&r->travPairRef, pairRef.size() * 4));
        HIP_TRY(hipMemcpy(r->travPairRef, pairRef.data(), pairRef.size() * 4, hipMemcpyHostToDevice));
    }
    r->p.trav.pairs = (const TriPair*)r->travPairs; r->p.trav.pairRef = (const uint32_t*)r->travPairRef; r->p.trav.numPairs = (uint32_t)pairs.size();
    r->p.trav.wnodes = (const WNode*)r->travWide; r->p.trav.wideCount = (uint32_t)wide.size(); r->p.trav.wideTopCount = wideTop; r->p.trav.wideStackBound = wideBound;
    r->p.trav.wideRootDesc = 0;

    if (r->travNodes) { HIP_TRY(hipFree(r->travNodes)); r->travNodes = nullptr; }
    if (r->travTris) { HIP_TRY(hipFree(r->travTris)); r->travTris = nullptr; }
    HIP_TRY(hipMalloc(&r->travNodes, packed.size() * sizeof(Node64)));
    HIP_TRY(hipMalloc(&r->travTris, ptris.size() * sizeof(Tri48)));
    HIP_TRY(hipMemcpy(r->travNodes, packed.data(), packed.size() * sizeof(Node64), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(r->travTris, ptris.data(), ptris.size() * sizeof(Tri48), hipMemcpyHostToDevice));
#ifdef GMUPT_VARIANTS
    // unified 64-byte records for the cooperative kernels (rungs of the test build only)
    std::vector<Rec64> recs(packed.size() + ptris.size());
    std::memset(recs.data(), 0, recs.size() * sizeof(Rec64));
    for (size_t i = 0; i < packed.size(); i++) std::memcpy(&recs[i], &packed[i], 64);
    for (size_t i = 0; i < ptris.size(); i++) {
        std::memcpy(&recs[packed.size() + i], &ptris[i], 48);
        if (i < R) std::memcpy(&recs[packed.size() + i].q[12], &tris[i], 16); // (v0, v1, v2, materialID)
    }
    if (r->travRecs) { HIP_TRY(hipFree(r->travRecs)); r->travRecs = nullptr; }
    HIP_TRY(hipMalloc(&r->travRecs, recs.size() * sizeof(Rec64)));
    HIP_TRY(hipMemcpy(r->travRecs, recs.data(), recs.size() * sizeof(Rec64), hipMemcpyHostToDevice));
#endif
    TravScene& t = r->p.trav;
    t.recs = (const Rec64*)r->travRecs; t.triBase = (uint32_t)packed.size();
    t.maxDepth = (uint32_t)maxDepth;
    t.nodes = (const Node64*)r->travNodes; t.tris = (const Tri48*)r->travTris;
    t.rootDesc = nodes[0].isLeaf ? ~(nodes[0].right > nodes[0].left ? nodes[0].left : (int32_t)R) : innerIndex[0];
    for (int k = 0; k < 3; k++) { t.rootMin[k] = nodes[0].min[k]; t.rootMax[k] = nodes[0].max[k]; }
    return GMUPT_OK;
}

extern "C" int gmupt_renderer_bind_scene(gmupt_renderer* r, const gmupt_buffer* nodes, const gmupt_buffer* triangles, const gmupt_buffer* vertices,
                                         const gmupt_buffer* lights, const gmupt_buffer* tri_props, const gmupt_buffer* materials)
{
    if (!r || !nodes || !triangles || !vertices || !lights || !tri_props || !materials) return fail(GMUPT_ERR_INVALID_ARGUMENT, "gmupt_renderer_bind_scene: null argument");
    if (nodes->kind != GMUPT_BUFFER_BVH_NODES || triangles->kind != GMUPT_BUFFER_TRIANGLES || vertices->kind != GMUPT_BUFFER_VERTICES ||
        lights->kind != GMUPT_BUFFER_LIGHTS || tri_props->kind != GMUPT_BUFFER_TRI_PROPS || materials->kind != GMUPT_BUFFER_MATERIALS)
        return fail(GMUPT_ERR_INVALID_ARGUMENT, "gmupt_renderer_bind_scene: buffer bound to the wrong slot");
    if (nodes->elems == 0) return fail(GMUPT_ERR_INVALID_ARGUMENT, "gmupt_renderer_bind_scene: empty BVH");
    if (tri_props->elems < vertices->elems) return fail(GMUPT_ERR_INVALID_ARGUMENT, "gmupt_renderer_bind_scene: %zu vertex properties for %zu vertices", tri_props->elems, vertices->elems);
    SceneView& s = r->p.scene;
    s.nodes = (const DNode*)nodes->dptr; s.tris = (const gmupt_triangle*)triangles->dptr; s.verts = (const float*)vertices->dptr;
    s.lights = (const gmupt_light*)lights->dptr; s.props = (const gmupt_tri_props*)tri_props->dptr; s.materials = (const gmupt_material*)materials->dptr;
    s.numNodes = (uint32_t)nodes->elems; s.numTris = (uint32_t)triangles->elems; s.numVerts = (uint32_t)vertices->elems; s.numMaterials = (uint32_t)materials->elems;
    int rc = build_traversal_copy(r, nodes, triangles, vertices);
    if (rc != GMUPT_OK) return rc;
    r->sceneBound = true;
    return GMUPT_OK;
}

extern "C" int gmupt_renderer_bind_textures(gmupt_renderer* r, const gmupt_buffer* diffuse, const gmupt_buffer* metallic_roughness, const gmupt_buffer* normals)
{
    if (!r) return fail(GMUPT_ERR_INVALID_ARGUMENT, "gmupt_renderer_bind_textures: null renderer");
    const gmupt_buffer* t[3] = { diffuse, metallic_roughness, normals };
    for (int k = 0; k < 3; k++) {
        if (t[k] && t[k]->kind != GMUPT_BUFFER_TEXTURE_ARRAY) return fail(GMUPT_ERR_INVALID_ARGUMENT, "gmupt_renderer_bind_textures: slot %d is not a texture array", k);
        r->p.scene.tex[k] = t[k] ? (const uint8_t*)t[k]->dptr : nullptr;
        r->p.scene.texSize[k] = t[k] ? t[k]->texSize : 0u;
        r->p.scene.texLayers[k] = t[k] ? t[k]->texLayers : 0u;
    }
    return GMUPT_OK;
}

extern "C" int gmupt_set_camera(gmupt_renderer* r, const gmupt_camera_buffer* cam)
{
    if (!r || !cam) return fail(GMUPT_ERR_INVALID_ARGUMENT, "gmupt_set_camera: null argument");
    r->p.cam = *cam; // travels to the kernels as a launch argument: the per-frame 112-byte upload of Renderer.cpp:161
    r->cameraSet = true;
    return GMUPT_OK;
}

static int resolve_timing(gmupt_renderer* r)
{
    if (r->evUsed == 0) return GMUPT_OK;
    HIP_TRY(hipStreamSynchronize(r->stream));
    for (size_t k = 0; k < r->evUsed; k++) {
        for (int sidx = 0; sidx < 4; sidx++) {
            if (r->evPool[k].extOnly && sidx != 2) continue;
            float ms = 0.f;
            HIP_TRY(hipEventElapsedTime(&ms, r->evPool[k].e[sidx], r->evPool[k].e[sidx + 1]));
            r->msStage[sidx] += ms;
        }
        r->timedIters++;
    }
    r->evUsed = 0;
    return GMUPT_OK;
}

static int run_iteration(gmupt_renderer* r, bool doShade, bool doExtend, bool doShadow)
{
    if (!r->sceneBound) return fail(GMUPT_ERR_NOT_BOUND, "gmupt_iterate: no scene bound");
    if (!r->cameraSet) return fail(GMUPT_ERR_NOT_BOUND, "gmupt_iterate: no camera set");
    HIP_TRY(hipSetDevice(r->dev->id));
    const RenderParams& p = r->p;
    const int clearFrame = (p.cam.iterationCounter == 0) ? 1 : 0; // logic.hlsl:206
    const bool stats = r->desc.collect_stats != 0;
    StageEvents* ev = nullptr;
    const bool extOnly = r->timing == 2;
    if (r->timing && doShade && doExtend && doShadow) {
        if (r->evUsed == r->evPool.size()) {
            if (r->evPool.size() >= 4096) { int rc = resolve_timing(r); if (rc != GMUPT_OK) return rc; }
            else { StageEvents se; for (auto& e : se.e) HIP_TRY(hipEventCreate(&e)); r->evPool.push_back(se); }
        }
        ev = &r->evPool[r->evUsed++];
        ev->extOnly = extOnly;
        if (!extOnly) HIP_TRY(hipEventRecord(ev->e[0], r->stream));
    }
    if (doShade) {
        r->p.groupParity ^= 1u;    // p is a reference to r->p: the launches of this iteration see the flipped half of the group totals
        if (clearFrame) launch_clear(p, r->stream); else launch_logic(p, r->stream);
        if (ev && !extOnly) HIP_TRY(hipEventRecord(ev->e[1], r->stream));
        launch_material(p, clearFrame, r->stream);  // computes its own queue offsets (no scan launch)
        if (ev) HIP_TRY(hipEventRecord(ev->e[2], r->stream));
    }
    if (!doShade) HIP_TRY(hipMemsetAsync(p.travCounters, 0, 128, r->stream)); // k_material (block 0) zeroes the ray-cast work counters in a full iteration
    if (doExtend && doShadow && traversal_is_fused(r->travMode)) {
        // one launch for both ray casts; its time is reported as the extension stage, the shadow stage as zero
        const uint32_t launched = launch_cast(p, stats, r->travMode, r->stream);
        if (launched) {
            r->castFlags |= launched;
            if (ev) HIP_TRY(hipEventRecord(ev->e[3], r->stream));
            if (ev && !extOnly) HIP_TRY(hipEventRecord(ev->e[4], r->stream));
            HIP_TRY(hipGetLastError());
            return GMUPT_OK;
        }
    }
    if (doExtend) { r->castFlags |= launch_extend(p, r->travBlocks, stats, r->travMode, r->stream); if (ev) HIP_TRY(hipEventRecord(ev->e[3], r->stream)); }
    if (doShadow) { r->castFlags |= launch_shadow(p, r->travBlocks, stats, r->travMode, r->stream); if (ev && !extOnly) HIP_TRY(hipEventRecord(ev->e[4], r->stream)); }
    HIP_TRY(hipGetLastError());
    return GMUPT_OK;
}

// After a stream synchronise: did a ray-cast launch flag its own results as invalid (DevStats::stackOverflow, sticky until gmupt_reset_stats)?
static int check_cast_flags(gmupt_renderer* r, const char* who)
{
    uint32_t flags = 0;
    HIP_TRY(hipMemcpyAsync(&flags, &r->p.stats->stackOverflow, 4, hipMemcpyDeviceToHost, r->stream));
    HIP_TRY(hipStreamSynchronize(r->stream));
    if (flags & 2u) return fail(GMUPT_ERR_CAST_FAULT, "%s: a wave of the ray cast left its loop at the iteration limit (GMUPT_STAT_CAST_ABORTED): the frame is invalid", who);
    if (flags & 1u) return fail(GMUPT_ERR_CAST_FAULT, "%s: a traversal stack overflowed (GMUPT_STAT_STACK_OVERFLOW): the frame is invalid", who);
    return GMUPT_OK;
}

extern "C" int gmupt_iterate(gmupt_renderer* r)
{
    if (!r) return fail(GMUPT_ERR_INVALID_ARGUMENT, "gmupt_iterate: null renderer");
    int rc = run_iteration(r, true, true, true);
    if (rc == GMUPT_OK) r->iterations++;
    return rc;
}

extern "C" int gmupt_debug_run_stage(gmupt_renderer* r, gmupt_stage stage)
{
    if (!r) return fail(GMUPT_ERR_INVALID_ARGUMENT, "gmupt_debug_run_stage: null renderer");
    switch (stage) {
    case GMUPT_STAGE_SHADE: return run_iteration(r, true, false, false);
    case GMUPT_STAGE_EXTEND: return run_iteration(r, false, true, false);
    case GMUPT_STAGE_SHADOW: return run_iteration(r, false, false, true);
    case GMUPT_STAGE_RAYCASTS: return run_iteration(r, false, true, true);
    }
    return fail(GMUPT_ERR_INVALID_ARGUMENT, "gmupt_debug_run_stage: unknown stage %d", (int)stage);
}

extern "C" int gmupt_synchronize(gmupt_renderer* r)
{
    if (!r) return fail(GMUPT_ERR_INVALID_ARGUMENT, "gmupt_synchronize: null renderer");
    HIP_TRY(hipSetDevice(r->dev->id));
    HIP_TRY(hipStreamSynchronize(r->stream));
    return check_cast_flags(r, "gmupt_synchronize");
}

extern "C" int gmupt_resize(gmupt_renderer* r, uint32_t width, uint32_t height)
{
    if (!r || width == 0 || height == 0) return fail(GMUPT_ERR_INVALID_ARGUMENT, "gmupt_resize: bad argument");
    HIP_TRY(hipSetDevice(r->dev->id));
    HIP_TRY(hipStreamSynchronize(r->stream));
    // the new target first: a failed allocation leaves the renderer on its old, still valid target
    float4* oldFb = r->p.fb; uint32_t* oldHead = r->p.listHead; const uint32_t oldW = r->p.fbW, oldH = r->p.fbH;
    r->p.fb = nullptr; r->p.listHead = nullptr;
    const int rc = alloc_framebuffer(r, width, height);
    if (rc != GMUPT_OK) {
        if (r->p.fb) (void)hipFree(r->p.fb);
        if (r->p.listHead) (void)hipFree(r->p.listHead);
        r->p.fb = oldFb; r->p.listHead = oldHead; r->p.fbW = oldW; r->p.fbH = oldH;
        return rc;
    }
    HIP_TRY(hipStreamSynchronize(r->stream));
    (void)hipFree(oldFb); (void)hipFree(oldHead);
    r->desc.width = width; r->desc.height = height;
    return GMUPT_OK;
}

extern "C" int gmupt_read_framebuffer(gmupt_renderer* r, float* rgba, size_t bytes)
{
    if (!r || !rgba) return fail(GMUPT_ERR_INVALID_ARGUMENT, "gmupt_read_framebuffer: null argument");
    const size_t need = (size_t)r->p.fbW * r->p.fbH * 16;
    if (bytes < need) return fail(GMUPT_ERR_INVALID_ARGUMENT, "gmupt_read_framebuffer: %zu bytes given, %zu needed", bytes, need);
    HIP_TRY(hipSetDevice(r->dev->id));
    HIP_TRY(hipMemcpyAsync(rgba, r->p.fb, need, hipMemcpyDeviceToHost, r->stream));
    HIP_TRY(hipStreamSynchronize(r->stream));
    return check_cast_flags(r, "gmupt_read_framebuffer");
}

extern "C" int gmupt_copy_framebuffer_to_device(gmupt_renderer* r, void* device_dst, size_t bytes)
{
    if (!r || !device_dst) return fail(GMUPT_ERR_INVALID_ARGUMENT, "gmupt_copy_framebuffer_to_device: null argument");
    const size_t need = (size_t)r->p.fbW * r->p.fbH * 16;
    if (bytes < need) return fail(GMUPT_ERR_INVALID_ARGUMENT, "gmupt_copy_framebuffer_to_device: %zu bytes given, %zu needed", bytes, need);
    HIP_TRY(hipSetDevice(r->dev->id));
    HIP_TRY(hipMemcpyAsync(device_dst, r->p.fb, need, hipMemcpyDeviceToDevice, r->stream));
    HIP_TRY(hipStreamSynchronize(r->stream));
    return check_cast_flags(r, "gmupt_copy_framebuffer_to_device");
}

extern "C" int gmupt_get_counters(gmupt_renderer* r, uint32_t out[8])
{
    if (!r || !out) return fail(GMUPT_ERR_INVALID_ARGUMENT, "gmupt_get_counters: null argument");
    HIP_TRY(hipSetDevice(r->dev->id));
    HIP_TRY(hipMemcpyAsync(out, r->p.qc, 32, hipMemcpyDeviceToHost, r->stream));
    HIP_TRY(hipStreamSynchronize(r->stream));
    return GMUPT_OK;
}

extern "C" int gmupt_enable_timing(gmupt_renderer* r, int enabled)
{
    if (!r) return fail(GMUPT_ERR_INVALID_ARGUMENT, "gmupt_enable_timing: null renderer");
    if (!enabled) { int rc = resolve_timing(r); if (rc != GMUPT_OK) return rc; }
    r->timing = enabled < 0 ? 0 : enabled;
    return GMUPT_OK;
}

extern "C" int gmupt_get_stats(gmupt_renderer* r, gmupt_stats* out)
{
    if (!r || !out) return fail(GMUPT_ERR_INVALID_ARGUMENT, "gmupt_get_stats: null argument");
    HIP_TRY(hipSetDevice(r->dev->id));
    int rc = resolve_timing(r);
    if (rc != GMUPT_OK) return rc;
    DevStats ds;
    HIP_TRY(hipMemcpyAsync(&ds, r->p.stats, sizeof(ds), hipMemcpyDeviceToHost, r->stream));
    HIP_TRY(hipStreamSynchronize(r->stream));
    std::memset(out, 0, sizeof(*out));
    out->iterations = r->iterations;
    out->paths_generated = ds.pathsGenerated; out->paths_completed = ds.pathsCompleted; out->segments = ds.segments;
    out->active_paths = ds.activePaths; out->flags = ((ds.stackOverflow & 1u) ? GMUPT_STAT_STACK_OVERFLOW : 0u) | ((ds.stackOverflow & 2u) ? GMUPT_STAT_CAST_ABORTED : 0u) | r->castFlags;
    out->ext_rays = ds.extRays; out->ext_inner = ds.extInner; out->ext_leaves = ds.extLeaves; out->ext_tris = ds.extTris;
    out->sh_rays = ds.shRays; out->sh_inner = ds.shInner; out->sh_leaves = ds.shLeaves; out->sh_tris = ds.shTris;
    out->ms_logic = r->msStage[0]; out->ms_material = r->msStage[1];
    out->ms_scan = 0.0;       // no scan launch: the queue ranks are computed inside k_logic (group totals) and k_material (block prefixes)
    out->ms_accumulate = 0.0; // accumulation is fused into the material kernel
    out->ms_extend = r->msStage[2]; out->ms_shadow = r->msStage[3];
    out->timed_iterations = r->timedIters;
    for (int k = 0; k < 32; k++) { out->ext_depth_hist[k] = ds.extDepthHist[k]; out->cast_wave_end_hist[k] = ds.castWaveEndHist[k]; out->ray_inner_hist[k] = ds.rayInnerHist[k]; }
    for (int k = 0; k < 4; k++) out->lane_census[k] = ds.laneCensus[k];
    out->cast_waves = ds.castWaves; out->cast_wave_ticks = ds.castWaveClocks; out->cast_wave_ticks_max = ds.castWaveClocksMax;
    out->cast_drain_ticks = ds.castDrainClocks; out->cast_drain_iters = ds.castDrainIters; out->cast_drain_busy_lanes = ds.castDrainBusyLanes;
    out->ext_top_inner = ds.extTopInner; out->sh_top_inner = ds.shTopInner; out->cast_helper_subtrees = ds.castHelperSubtrees;
    out->cast_nested_helpers = ds.castNestedHelpers; out->cast_redo_rays = ds.castRedoRays; out->wide_nodes = r->p.trav.wideCount; out->wide_top_nodes = r->p.trav.wideTopCount; out->wide_stack_bound = r->p.trav.wideStackBound; out->wide_pairs = r->p.trav.numPairs; out->wide_pair_fetches = ds.widePairFetches; out->wide_box_tests = ds.wideBoxTests; out->wide_iterations = ds.wideIters; out->wide_general_iterations = ds.wideGeneralIters;
    out->ext_wave_inner = ds.extWaveInner; out->ext_wave_tris = ds.extWaveTris; out->sh_wave_inner = ds.shWaveInner; out->sh_wave_tris = ds.shWaveTris;
    if (ds.stackOverflow & 3u) return fail(GMUPT_ERR_CAST_FAULT, "gmupt_get_stats: a ray-cast launch flagged its results as invalid (flags %#x; the statistics are filled in)", out->flags);
    return GMUPT_OK;
}

extern "C" int gmupt_reset_stats(gmupt_renderer* r)
{
    if (!r) return fail(GMUPT_ERR_INVALID_ARGUMENT, "gmupt_reset_stats: null renderer");
    HIP_TRY(hipSetDevice(r->dev->id));
    int rc = resolve_timing(r);
    if (rc != GMUPT_OK) return rc;
    DevStats ds;
    HIP_TRY(hipMemcpyAsync(&ds, r->p.stats, sizeof(ds), hipMemcpyDeviceToHost, r->stream));
    HIP_TRY(hipStreamSynchronize(r->stream));
    const uint32_t active = ds.activePaths;
    std::memset(&ds, 0, sizeof(ds)); ds.activePaths = active;
    HIP_TRY(hipMemcpyAsync(r->p.stats, &ds, sizeof(ds), hipMemcpyHostToDevice, r->stream));
    HIP_TRY(hipStreamSynchronize(r->stream));
    for (double& m : r->msStage) m = 0.0;
    r->timedIters = 0; r->iterations = 0; r->castFlags = 0;
    return GMUPT_OK;
}

struct gmupt_camera { Camera cam; gmupt_camera(uint32_t w, uint32_t h) : cam(w, h) {} };

extern "C" int gmupt_render_budget(gmupt_renderer* r, gmupt_camera* cam, uint32_t max_iterations, uint32_t* iters)
{
    if (!r || !cam) return fail(GMUPT_ERR_INVALID_ARGUMENT, "gmupt_render_budget: null argument");
    if (!r->desc.path_budget) return fail(GMUPT_ERR_INVALID_ARGUMENT, "gmupt_render_budget: renderer was created without a path_budget");
    HIP_TRY(hipSetDevice(r->dev->id));
    static_assert(offsetof(DevStats, stackOverflow) == offsetof(DevStats, activePaths) + 4, "the drain check reads both words with one copy");
    uint32_t* hostActive = nullptr;   // [0] activePaths, [1] stackOverflow
    HIP_TRY(hipHostMalloc((void**)&hostActive, 8, hipHostMallocDefault));
    hostActive[0] = 1; hostActive[1] = 0;
    uint32_t k = 0;
    int rc = GMUPT_OK;
    // Drain: stop when no slot is active any more -- or kDrainHorizon iterations after the budget ran out.  A healthy path lives at most
    // ~205 iterations (Russian roulette after 200 bounces, logic.hlsl:248-255), but the reference has paths that do not end that way:
    // a throughput that overflowed to inf survives the roulette and becomes inf / inf = NaN (:251-254), a NaN throughput survives
    // `all(throughput <= 0)` (:237), and such a path only ends when its ray happens to hit a light or leave the scene -- in the closed
    // bench room their number halves every ~450 iterations (0.007 % of all paths; their sample is saturate(NaN) = 0).  In the
    // reference's progressive loop they just occupy pool slots; a bounded job must cut them off.
    constexpr uint32_t kDrainHorizon = 512;
    uint32_t drainStart = 0xFFFFFFFFu;
    for (; k < max_iterations; k++) {
        cam->cam.update(0.0f);                       // Renderer::update -> Scene::update -> Camera::update (Renderer.cpp:158)
        rc = gmupt_set_camera(r, cam->cam.getBuffer());
        if (rc == GMUPT_OK) rc = gmupt_iterate(r);   // Renderer::draw
        if (rc != GMUPT_OK) break;
        if ((k & 7u) == 7u) {                        // drain check without stalling every iteration
            hipError_t e = hipMemcpyAsync(hostActive, &r->p.stats->activePaths, 8, hipMemcpyDeviceToHost, r->stream);
            if (e == hipSuccess) e = hipStreamSynchronize(r->stream);
            if (e != hipSuccess) { rc = fail(GMUPT_ERR_HIP, "gmupt_render_budget: %s", hipGetErrorString(e)); break; }
            if (hostActive[1] & 3u) { rc = fail(GMUPT_ERR_CAST_FAULT, "gmupt_render_budget: a ray-cast launch flagged its results as invalid after %u iterations (flags %#x)", k + 1, hostActive[1]); k++; break; }
            if (*hostActive == 0) { k++; break; }
            if (*hostActive < r->p.L && drainStart == 0xFFFFFFFFu) drainStart = k;     // the first slots have retired: the budget is spent
            if (drainStart != 0xFFFFFFFFu && k - drainStart >= kDrainHorizon) { k++; break; }
        }
    }
    (void)hipHostFree(hostActive);
    if (rc == GMUPT_OK) { hipError_t e = hipStreamSynchronize(r->stream); if (e != hipSuccess) rc = fail(GMUPT_ERR_HIP, "gmupt_render_budget: %s", hipGetErrorString(e)); }
    if (rc == GMUPT_OK) rc = check_cast_flags(r, "gmupt_render_budget");
    if (iters) *iters = k;
    return rc;
}

// ------------------------------------------------------------------------------------------------ debug access (reference layout)
namespace {
struct FieldMap { uint32_t refOffset, slotBytes, comps, first; };
// Assets/Shaders/structs.h:19-48 (offset in bytes per path, x PATHCOUNT) -> first SoA component
const FieldMap kFieldMap[] = {
    { 0, 16, 3, F_RAY_OX }, { 16, 16, 3, F_RAY_DX }, { 32, 16, 3, F_MAT_R }, { 48, 8, 2, F_MAT_METALLIC }, { 56, 16, 3, F_NRM_X },
    { 72, 16, 3, F_SP_X }, { 88, 16, 3, F_BARY_X }, { 104, 4, 1, F_HIT_DIST }, { 108, 16, 4, F_TRI_0 }, { 124, 16, 3, F_SH_OX },
    { 140, 16, 3, F_SH_DX }, { 156, 4, 1, F_LIGHT_IDX }, { 160, 4, 1, F_LIGHT_DIST }, { 164, 4, 1, F_IN_SHADOW }, { 168, 16, 3, F_RAD_R },
    { 184, 16, 3, F_THR_R }, { 200, 16, 3, F_LTHR_R }, { 216, 16, 3, F_DL_R }, { 232, 4, 1, F_PATH_LEN }, { 236, 8, 2, F_SCR_X }, { 244, 4, 1, F_IS_EMITTER },
};
}

extern "C" int gmupt_debug_read_path_state(gmupt_renderer* r, void* dst, size_t bytes)
{
    if (!r || !dst) return fail(GMUPT_ERR_INVALID_ARGUMENT, "gmupt_debug_read_path_state: null argument");
    const size_t P = r->p.P;
    if (bytes < P * GMUPT_STATE_BYTES) return fail(GMUPT_ERR_INVALID_ARGUMENT, "gmupt_debug_read_path_state: %zu bytes given, %zu needed", bytes, P * (size_t)GMUPT_STATE_BYTES);
    HIP_TRY(hipSetDevice(r->dev->id));
    const size_t PS = r->p.PS;
    std::vector<uint32_t> soa((size_t)F_COUNT * PS);
    HIP_TRY(hipMemcpyAsync(soa.data(), r->p.state, soa.size() * 4, hipMemcpyDeviceToHost, r->stream));
    HIP_TRY(hipStreamSynchronize(r->stream));
    std::memset(dst, 0, P * GMUPT_STATE_BYTES);
    uint8_t* out = (uint8_t*)dst;
    for (const FieldMap& f : kFieldMap)
        for (size_t i = 0; i < P; i++) {
            uint32_t* o = (uint32_t*)(out + (size_t)f.refOffset * P + (size_t)f.slotBytes * i);
            for (uint32_t c = 0; c < f.comps; c++) o[c] = soa[(size_t)(f.first + c) * PS + i];
        }
    return GMUPT_OK;
}

extern "C" int gmupt_debug_write_path_state(gmupt_renderer* r, const void* src, size_t bytes)
{
    if (!r || !src) return fail(GMUPT_ERR_INVALID_ARGUMENT, "gmupt_debug_write_path_state: null argument");
    const size_t P = r->p.P;
    if (bytes < P * GMUPT_STATE_BYTES) return fail(GMUPT_ERR_INVALID_ARGUMENT, "gmupt_debug_write_path_state: %zu bytes given, %zu needed", bytes, P * (size_t)GMUPT_STATE_BYTES);
    HIP_TRY(hipSetDevice(r->dev->id));
    const size_t PS = r->p.PS;
    std::vector<uint32_t> soa((size_t)F_COUNT * PS);
    const uint8_t* in = (const uint8_t*)src;
    for (const FieldMap& f : kFieldMap)
        for (size_t i = 0; i < P; i++) {
            const uint32_t* o = (const uint32_t*)(in + (size_t)f.refOffset * P + (size_t)f.slotBytes * i);
            for (uint32_t c = 0; c < f.comps; c++) soa[(size_t)(f.first + c) * PS + i] = o[c];
        }
    HIP_TRY(hipMemcpyAsync(r->p.state, soa.data(), soa.size() * 4, hipMemcpyHostToDevice, r->stream));
    HIP_TRY(hipStreamSynchronize(r->stream));
    return GMUPT_OK;
}

extern "C" int gmupt_debug_read_queues(gmupt_renderer* r, uint32_t* dst, size_t bytes)
{
    if (!r || !dst) return fail(GMUPT_ERR_INVALID_ARGUMENT, "gmupt_debug_read_queues: null argument");
    const size_t need = (size_t)r->p.P * 20;
    if (bytes < need) return fail(GMUPT_ERR_INVALID_ARGUMENT, "gmupt_debug_read_queues: %zu bytes given, %zu needed", bytes, need);
    HIP_TRY(hipSetDevice(r->dev->id));
    HIP_TRY(hipMemcpyAsync(dst, r->p.queues, need, hipMemcpyDeviceToHost, r->stream));
    HIP_TRY(hipStreamSynchronize(r->stream));
    return GMUPT_OK;
}

extern "C" int gmupt_debug_write_queues(gmupt_renderer* r, const uint32_t* src, size_t bytes)
{
    if (!r || !src) return fail(GMUPT_ERR_INVALID_ARGUMENT, "gmupt_debug_write_queues: null argument");
    const size_t need = (size_t)r->p.P * 20;
    if (bytes < need) return fail(GMUPT_ERR_INVALID_ARGUMENT, "gmupt_debug_write_queues: %zu bytes given, %zu needed", bytes, need);
    HIP_TRY(hipSetDevice(r->dev->id));
    HIP_TRY(hipMemcpyAsync(r->p.queues, src, need, hipMemcpyHostToDevice, r->stream));
    HIP_TRY(hipStreamSynchronize(r->stream));
    return GMUPT_OK;
}

extern "C" int gmupt_debug_write_counters(gmupt_renderer* r, const uint32_t in[8])
{
    if (!r || !in) return fail(GMUPT_ERR_INVALID_ARGUMENT, "gmupt_debug_write_counters: null argument");
    HIP_TRY(hipSetDevice(r->dev->id));
    HIP_TRY(hipMemcpyAsync(r->p.qc, in, 32, hipMemcpyHostToDevice, r->stream));
    HIP_TRY(hipStreamSynchronize(r->stream));
    return GMUPT_OK;
}

extern "C" int gmupt_debug_write_framebuffer(gmupt_renderer* r, const float* rgba, size_t bytes)
{
    if (!r || !rgba) return fail(GMUPT_ERR_INVALID_ARGUMENT, "gmupt_debug_write_framebuffer: null argument");
    const size_t need = (size_t)r->p.fbW * r->p.fbH * 16;
    if (bytes < need) return fail(GMUPT_ERR_INVALID_ARGUMENT, "gmupt_debug_write_framebuffer: %zu bytes given, %zu needed", bytes, need);
    HIP_TRY(hipSetDevice(r->dev->id));
    HIP_TRY(hipMemcpyAsync(r->p.fb, rgba, need, hipMemcpyHostToDevice, r->stream));
    HIP_TRY(hipStreamSynchronize(r->stream));
    return GMUPT_OK;
}

extern "C" int gmupt_debug_detmath(gmupt_device* dev, int fn, const float* x, const float* y, float* out, uint32_t n)
{
    if (!dev || !x || !y || !out) return fail(GMUPT_ERR_INVALID_ARGUMENT, "gmupt_debug_detmath: null argument");
    if (n == 0) return GMUPT_OK;
    HIP_TRY(hipSetDevice(dev->id));
    float *dx = nullptr, *dy = nullptr, *dout = nullptr;
    HIP_TRY(hipMalloc((void**)&dx, (size_t)n * 4)); HIP_TRY(hipMalloc((void**)&dy, (size_t)n * 4)); HIP_TRY(hipMalloc((void**)&dout, (size_t)n * 4));
    HIP_TRY(hipMemcpy(dx, x, (size_t)n * 4, hipMemcpyHostToDevice)); HIP_TRY(hipMemcpy(dy, y, (size_t)n * 4, hipMemcpyHostToDevice));
    launch_detmath(fn, dx, dy, dout, n, nullptr);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(out, dout, (size_t)n * 4, hipMemcpyDeviceToHost));
    (void)hipFree(dx); (void)hipFree(dy); (void)hipFree(dout);
    return GMUPT_OK;
}

// ------------------------------------------------------------------------------------------------ host: SBVH
struct gmupt_sbvh { gmupt::SbvhBuilder* b; };

extern "C" void gmupt_sbvh_default_params(gmupt_sbvh_params* p)
{
    if (!p) return;
    p->split_alpha = 1.0e-5f; p->max_depth = 64; p->max_spatial_depth = 48; p->min_leaf_size = 1; p->max_leaf_size = 0x7FFFFFF;
    p->node_cost = 1.0f; p->tri_cost = 1.0f;
}

extern "C" int gmupt_sbvh_build(const float* vertices, uint32_t num_vertices, const int32_t* indices, uint32_t num_triangles,
                                const gmupt_sbvh_params* params, gmupt_sbvh** out)
{
    if (!out || (!vertices && num_vertices) || (!indices && num_triangles)) return fail(GMUPT_ERR_INVALID_ARGUMENT, "gmupt_sbvh_build: null argument");
    *out = nullptr;
    gmupt_sbvh_params prm; gmupt_sbvh_default_params(&prm);
    if (params) prm = *params;
    if (prm.max_depth < 1 || prm.max_depth > 64) return fail(GMUPT_ERR_INVALID_ARGUMENT, "gmupt_sbvh_build: max_depth %d outside [1, 64]", prm.max_depth);
    try {
        gmupt_sbvh* h = new gmupt_sbvh();
        h->b = new gmupt::SbvhBuilder(vertices, num_vertices, indices, num_triangles, prm);
        h->b->build();
        *out = h;
    } catch (const std::exception& e) {
        return fail(GMUPT_ERR_INVALID_ARGUMENT, "gmupt_sbvh_build: %s", e.what());
    }
    return GMUPT_OK;
}
extern "C" uint32_t gmupt_sbvh_num_nodes(const gmupt_sbvh* h) { return h ? h->b->numNodes() : 0; }
extern "C" uint32_t gmupt_sbvh_num_references(const gmupt_sbvh* h) { return h ? h->b->numReferences() : 0; }
extern "C" float gmupt_sbvh_sah(const gmupt_sbvh* h) { return h ? h->b->sah() : 0.0f; }
extern "C" uint32_t gmupt_sbvh_depth(const gmupt_sbvh* h) { return h ? h->b->depth() : 0; }
extern "C" int gmupt_sbvh_flatten(const gmupt_sbvh* h, const uint32_t* vertex_material, gmupt_bvh_node* nodes, gmupt_triangle* triangles, int32_t* ref_triangle)
{
    if (!h || !nodes) return fail(GMUPT_ERR_INVALID_ARGUMENT, "gmupt_sbvh_flatten: null argument");
    h->b->flatten(vertex_material, nodes, triangles, ref_triangle);
    return GMUPT_OK;
}
extern "C" void gmupt_sbvh_destroy(gmupt_sbvh* h) { if (h) { delete h->b; delete h; } }

// ------------------------------------------------------------------------------------------------ host: camera
extern "C" int gmupt_camera_create(uint32_t width, uint32_t height, gmupt_camera** out)
{
    if (!out || !width || !height) return fail(GMUPT_ERR_INVALID_ARGUMENT, "gmupt_camera_create: bad argument");
    *out = new (std::nothrow) gmupt_camera(width, height);
    return *out ? GMUPT_OK : fail(GMUPT_ERR_OUT_OF_MEMORY, "gmupt_camera_create: out of host memory");
}
extern "C" void gmupt_camera_destroy(gmupt_camera* c) { delete c; }
extern "C" void gmupt_camera_update_resolution(gmupt_camera* c, uint32_t width, uint32_t height) { if (c) c->cam.updateResolution(width, height); }
extern "C" void gmupt_camera_set_pose(gmupt_camera* c, float x, float y, float z, float pitch, float yaw) { if (c) { c->cam.setPosition(x, y, z); c->cam.setRotation(pitch, yaw); } }
extern "C" void gmupt_camera_update(gmupt_camera* c, float dt) { if (c) c->cam.update(dt); }
extern "C" void gmupt_camera_set_input(gmupt_camera* c, float mouse_dx, float mouse_dy, uint32_t keys_wsad)
{
    if (!c) return;
    c->cam.addMouseDelta(mouse_dx, mouse_dy);
    c->cam.setKeys((keys_wsad & 1u) != 0, (keys_wsad & 2u) != 0, (keys_wsad & 4u) != 0, (keys_wsad & 8u) != 0);
}

extern "C" void gmupt_camera_reset_accumulation(gmupt_camera* c) { if (c) c->cam.getBuffer()->iterationCounter = -1; }
extern "C" gmupt_camera_buffer* gmupt_camera_get_buffer(gmupt_camera* c) { return c ? c->cam.getBuffer() : nullptr; }
