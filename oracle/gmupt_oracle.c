/*
 * oracle/gmupt_oracle.c -- TEST INFRASTRUCTURE ONLY (see gmupt_oracle.h).
 *
 * Scalar C restatement of the reference's six wavefront stages under the
 * canonical schedule.  PARITY UNPINNED by the reference (no golden vectors exist,
 * SURVEY.md 8c); every function cites the reference file:line it follows.
 * All citations are relative to /root/reference/.
 *
 * Arithmetic conventions (shared with the HIP kernels, stated in DESIGN.md):
 *   - IEEE-754 binary32, round-to-nearest, no FMA contraction (-ffp-contract=off);
 *   - dot(a,b)   = (a.x*b.x + a.y*b.y) + a.z*b.z
 *   - cross(a,b) = (a.y*b.z - a.z*b.y, a.z*b.x - a.x*b.z, a.x*b.y - a.y*b.x)
 *   - length(v)  = sqrt(dot(v,v));  normalize(v) = v * (1 / length(v))
 *   - HLSL expressions are evaluated left to right exactly as written;
 *   - pow(x, 5) and pow(x, 4) with literal integer exponents are multiplication
 *     chains ((x*x)*(x*x))*x and (x*x)*(x*x); pow(x, 1/2.2) is exp2(y*log2 x);
 *   - sin, cos, exp2, log2 are the deterministic versions of detmath.h.
 */
#include "gmupt_oracle.h"
#include "detmath.h"

#include <stdlib.h>
#include <stdio.h>
#include <float.h>

#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------ vectors */
typedef struct { float x, y, z; } v3;

static inline v3 V(float x, float y, float z) { v3 r = { x, y, z }; return r; }
static inline v3 vadd(v3 a, v3 b) { return V(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 vsub(v3 a, v3 b) { return V(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 vmul(v3 a, v3 b) { return V(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline v3 vdiv(v3 a, v3 b) { return V(a.x / b.x, a.y / b.y, a.z / b.z); }
static inline v3 vscale(v3 a, float s) { return V(a.x * s, a.y * s, a.z * s); }
static inline v3 vneg(v3 a) { return V(-a.x, -a.y, -a.z); }
static inline float vdot(v3 a, v3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
static inline v3 vcross(v3 a, v3 b) { return V(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
static inline float vlength(v3 a) { return sqrtf(vdot(a, a)); }
static inline v3 vnormalize(v3 a) { float inv = 1.0f / vlength(a); return vscale(a, inv); }

#define O_PI 3.14159274f      /* structs.h:14 rounded to binary32 */
#define O_INVPI 0.318309873f  /* structs.h:15 rounded to binary32 */
#define O_EPSILON 1e-8f       /* structs.h:10 */
#define O_EPSILON_OFFSET 1e-3f /* structs.h:11 */

/* ------------------------------------------------------------------ state layout (structs.h:19-48) */
/* every field occupies stride*4 bytes per path, float3 in 16-byte slots; offsets are multiplied by PATHCOUNT */
enum {
    F_RAY_ORIGIN = 0, F_RAY_DIRECTION = 16, F_MAT_COLOR = 32, F_MAT_MR = 48, F_NORMAL = 56,
    F_SURFACEPOINT = 72, F_BARYCOORD = 88, F_HITDISTANCE = 104, F_TRIANGLE = 108,
    F_SHADOWRAY_ORIGIN = 124, F_SHADOWRAY_DIRECTION = 140, F_LIGHT_INDEX = 156, F_LIGHT_DISTANCE = 160,
    F_INSHADOW = 164, F_RADIANCE = 168, F_THROUGHPUT = 184, F_LIGHT_THROUGHPUT = 200, F_DIRECT_LIGHT = 216,
    F_PATH_LENGTH = 232, F_SCREEN_COORD = 236, F_ISEMITTER = 244
};
/* queues (structs.h:53-58) */
enum { Q_NEWPATH = 0, Q_MAT_UE4 = 1, Q_MAT_GLASS = 2, Q_EXT_RAY = 3, Q_SHADOW_RAY = 4 };
/* counters (structs.h:62-68); [7] is this build's extension: number of live entries in the extension queue */
enum { QC_NEWPATH = 0, QC_LASTPATHCNT = 1, QC_MATUE4 = 2, QC_MATGLASS = 3, QC_EXT_UE4_OFFSET = 4, QC_EXT_GLASS_OFFSET = 5, QC_SHADOWRAY = 6, QC_EXT_COUNT = 7 };

struct orc_renderer {
    orc_scene scene;
    orc_config cfg;
    orc_camera_buffer cam;
    uint8_t* state;
    uint32_t* queue;
    uint32_t qc[8];
    float* fb;
    uint8_t* retired; /* extension: slot retired by pathBudget */
    uint8_t* scratchKind; /* per slot / queue item outcome of the parallel first pass of logic and materialUE4 */
    float* scratchRad;    /* radiance of the paths ended in this iteration (3 per slot) */
    uint32_t activePaths;
    orc_stats stats;
};

static inline uint8_t* fld(orc_renderer* r, uint32_t off, uint32_t widthBytes, uint32_t index)
{
    /* GET(what,index,bytes) = OFFSET + 4*bytes*index with OFFSET = off*PATHCOUNT (structs.h:16-18,73) */
    return r->state + (size_t)off * r->cfg.poolPaths + (size_t)widthBytes * index;
}
static inline v3 ld3(orc_renderer* r, uint32_t off, uint32_t i) { float* p = (float*)fld(r, off, 16, i); return V(p[0], p[1], p[2]); }
static inline void st3(orc_renderer* r, uint32_t off, uint32_t i, v3 v) { float* p = (float*)fld(r, off, 16, i); p[0] = v.x; p[1] = v.y; p[2] = v.z; }
static inline float ld1f(orc_renderer* r, uint32_t off, uint32_t i) { return *(float*)fld(r, off, 4, i); }
static inline void st1f(orc_renderer* r, uint32_t off, uint32_t i, float v) { *(float*)fld(r, off, 4, i) = v; }
static inline uint32_t ld1u(orc_renderer* r, uint32_t off, uint32_t i) { return *(uint32_t*)fld(r, off, 4, i); }
static inline void st1u(orc_renderer* r, uint32_t off, uint32_t i, uint32_t v) { *(uint32_t*)fld(r, off, 4, i) = v; }

static inline uint32_t* qptr(orc_renderer* r, int q) { return r->queue + (size_t)q * r->cfg.poolPaths; }

/* ------------------------------------------------------------------ RNG (random.h:6-12) */
typedef struct { float sx, sy; float rsx, rsy; } rng_t;

static inline void rng_seed(rng_t* g, const orc_camera_buffer* cam, uint32_t i)
{
    /* seed = float2(frac(index * INVPI), frac(index * PI)); logic.hlsl:216, newPath.hlsl:27, materialUE4.hlsl:131, materialGlass.hlsl:61 */
    float fi = (float)i;
    g->sx = o_frac(fi * O_INVPI);
    g->sy = o_frac(fi * O_PI);
    g->rsx = cam->randomSeed[0];
    g->rsy = cam->randomSeed[1];
}
static inline float rng_next(rng_t* g)
{
    /* seed -= cam.randomSeed; return frac(sin(dot(seed, float2(12.9898, 78.233))) * 43758.5453); random.h:10-11 */
    g->sx = g->sx - g->rsx;
    g->sy = g->sy - g->rsy;
    float d = g->sx * 12.9898f + g->sy * 78.233f;
    return o_frac(o_sin(d) * 43758.5453f);
}

/* ------------------------------------------------------------------ bsdf.h */
static inline v3 schlickFresnel3(float r0, float theta) /* bsdf.h:1-6 (note the minus sign, quirk Q12) */
{
    float m = o_saturate(1.0f - theta);
    float m2 = m * m;
    float v = r0 - (1.0f - r0) * m2 * m2 * m;
    return V(v, v, v);
}
static inline float GGXTrowbridgeReitz(float XdotY, float alpha) /* bsdf.h:8-13 */
{
    float a2 = alpha * alpha;
    float x = XdotY * XdotY * (a2 - 1.0f) + 1.0f;
    return a2 / (O_PI * x * x);
}
static inline float smithSchlickGGX(float XdotY, float alpha) /* bsdf.h:15-20 */
{
    float a1 = alpha + 1.0f;
    float k = (a1 * a1) / 8.0f;
    return XdotY / (XdotY * (1.0f - k) + k);
}
static inline float lightFalloff(float distance, float radius) /* bsdf.h:22-26 */
{
    float q = distance / radius;
    float q2 = q * q;
    float n = o_saturate(1.0f - q2 * q2);
    return (n * n) / (distance * distance + 1.0f);
}
static inline float powerHeuristic(float rayPdf, float lightPdf) /* bsdf.h:28-32 */
{
    float t = rayPdf * rayPdf;
    return t / (lightPdf * lightPdf + t);
}

/* ------------------------------------------------------------------ helpers */
static inline uint32_t cam_width(const orc_camera_buffer* c) { return (uint32_t)(1.0f / c->pixelSize[0]); } /* newPath.hlsl:30 */
static inline uint32_t cam_height(const orc_camera_buffer* c) { return (uint32_t)(1.0f / c->pixelSize[1]); } /* newPath.hlsl:31 */

static void tile_dims(orc_renderer* r, uint32_t* w, uint32_t* h, uint32_t* x0, uint32_t* y0)
{
    if (r->cfg.tileEnabled) { *w = r->cfg.fbWidth; *h = r->cfg.fbHeight; *x0 = r->cfg.tileX0; *y0 = r->cfg.tileY0; }
    else { *w = cam_width(&r->cam); *h = cam_height(&r->cam); *x0 = 0; *y0 = 0; }
}

/* ------------------------------------------------------------------ stage 1: logic.hlsl */

/* logic.hlsl:165-190 clearTexture: zero every pixel's sample count (colour kept), newPath[i] = i, QC0 = PATHCOUNT */
static void clear_texture(orc_renderer* r)
{
    uint32_t w, h, x0, y0;
    tile_dims(r, &w, &h, &x0, &y0);
    uint32_t P = r->cfg.poolPaths;
    uint32_t* qn = qptr(r, Q_NEWPATH);
    for (uint32_t index = 0; index < w * h && index < r->cfg.fbWidth * r->cfg.fbHeight; index++)
        r->fb[(size_t)index * 4 + 3] = o_asfloat(0u);
    r->qc[QC_NEWPATH] = P; /* logic.hlsl:178 (quirk Q1: PATHCOUNT, not the live count) */
    for (uint32_t index = 0; index < P; index++) qn[index] = index; /* logic.hlsl:187-188 */
    /* extension bookkeeping: a clear revives every slot */
    for (uint32_t i = 0; i < P; i++) r->retired[i] = 0;
}

/* logic.hlsl:33-77 endPath (the queue push is done by the caller in canonical order) */
static void accumulate_sample(orc_renderer* r, v3 radiance, uint32_t index)
{
    radiance = V(o_saturate(radiance.x), o_saturate(radiance.y), o_saturate(radiance.z)); /* :49 */
    radiance = vdiv(radiance, vadd(radiance, V(1, 1, 1)));                                  /* :52 */
    const float g = 1.0f / 2.2f;
    radiance = V(o_pow(radiance.x, g), o_pow(radiance.y, g), o_pow(radiance.z, g));        /* :53 */

    uint32_t* sc = (uint32_t*)fld(r, F_SCREEN_COORD, 8, index);                             /* :55 */
    uint32_t cx = sc[0], cy = sc[1];
    uint32_t lx = cx - (r->cfg.tileEnabled ? r->cfg.tileX0 : 0u);
    uint32_t ly = cy - (r->cfg.tileEnabled ? r->cfg.tileY0 : 0u);
    if (lx >= r->cfg.fbWidth || ly >= r->cfg.fbHeight) return; /* out-of-bounds UAV writes are dropped by D3D11 */
    float* px = r->fb + ((size_t)ly * r->cfg.fbWidth + lx) * 4;
    v3 pixel = V(px[0], px[1], px[2]);                                                      /* :57 */
    uint32_t sampleCount = o_asuint(px[3]);                                                 /* :58 */
    /* :73  ((pixel * sampleCount++) + radiance) / sampleCount */
    float n0 = (float)sampleCount;
    sampleCount++;
    float n1 = (float)sampleCount;
    v3 out = vadd(vscale(pixel, n0), radiance);
    px[0] = out.x / n1; px[1] = out.y / n1; px[2] = out.z / n1;
    px[3] = o_asfloat(sampleCount);
}

/* SampleLevel(samplerState, float3(uv, layer), 0) with the reference's sampler: MIN_MAG_MIP_LINEAR, WRAP (Scene.cpp:180-192) on an
 * R8G8B8A8_UNORM array, no sRGB decode (Scene.cpp:260).  The hardware filter is not specified bit for bit (D3D11 only asks for
 * 8 fractional weight bits), so this build states it: texel centre convention x = u*size - 0.5, fp32 weights, lerp(a,b,t) = a + t*(b-a),
 * horizontal pairs first, UNORM decode c / 255.  PARITY UNPINNED against the DX11 sampler. */
static void sample_bilinear(const orc_scene* sc, int which, float u, float v, int layer, float out[4])
{
    out[0] = out[1] = out[2] = out[3] = 0.0f;
    const uint8_t* tex = sc->tex[which];
    const int n = (int)sc->texSize[which];
    if (!tex || n <= 0 || sc->texLayers[which] == 0) return;            /* unbound SRV reads zero */
    if (layer < 0) layer = 0;
    if (layer >= (int)sc->texLayers[which]) layer = (int)sc->texLayers[which] - 1;
    float x = u * (float)n - 0.5f, y = v * (float)n - 0.5f;
    if (!(fabsf(x) < 1.0e9f)) x = 0.0f;
    if (!(fabsf(y) < 1.0e9f)) y = 0.0f;
    const float x0 = floorf(x), y0 = floorf(y);
    const float fx = x - x0, fy = y - y0;
    int ix0 = (int)x0 % n; if (ix0 < 0) ix0 += n;
    int iy0 = (int)y0 % n; if (iy0 < 0) iy0 += n;
    const int ix1 = (ix0 + 1 == n) ? 0 : ix0 + 1, iy1 = (iy0 + 1 == n) ? 0 : iy0 + 1;
    const uint8_t* base = tex + (size_t)layer * n * n * 4;
    const uint8_t* c00 = base + ((size_t)iy0 * n + ix0) * 4; const uint8_t* c10 = base + ((size_t)iy0 * n + ix1) * 4;
    const uint8_t* c01 = base + ((size_t)iy1 * n + ix0) * 4; const uint8_t* c11 = base + ((size_t)iy1 * n + ix1) * 4;
    for (int k = 0; k < 4; k++) {
        const float a = (float)c00[k] / 255.0f, b = (float)c10[k] / 255.0f, c = (float)c01[k] / 255.0f, d = (float)c11[k] / 255.0f;
        const float r0 = a + fx * (b - a), r1 = c + fx * (d - c);
        out[k] = r0 + fy * (r1 - r0);
    }
}

/* logic.hlsl:79-133 setMaterialHitProperties */
static uint32_t set_material_hit_properties(orc_renderer* r, uint32_t index)
{
    uint32_t* tri = (uint32_t*)fld(r, F_TRIANGLE, 16, index);            /* :81 */
    /* :82 float3 indices = tri.xyz -- the index round-trips through float (quirk Q18) */
    uint32_t i0 = (uint32_t)(float)tri[0], i1 = (uint32_t)(float)tri[1], i2 = (uint32_t)(float)tri[2];
    v3 bary = ld3(r, F_BARYCOORD, index);                                /* :83 */
    const orc_tri_props* tp = r->scene.props;
    v3 n0 = V(tp[i0].normal[0], tp[i0].normal[1], tp[i0].normal[2]);     /* :89-91 */
    v3 n1 = V(tp[i1].normal[0], tp[i1].normal[1], tp[i1].normal[2]);
    v3 n2 = V(tp[i2].normal[0], tp[i2].normal[1], tp[i2].normal[2]);
    /* :93 texCoord = t0*b.x + t1*b.y + t2*b.z */
    float tu = (tp[i0].uv[0] * bary.x + tp[i1].uv[0] * bary.y) + tp[i2].uv[0] * bary.z;
    float tv = (tp[i0].uv[1] * bary.x + tp[i1].uv[1] * bary.y) + tp[i2].uv[1] * bary.z;
    /* :94 normal = n0*b.x + n1*b.y + n2*b.z */
    v3 normal = vadd(vadd(vscale(n0, bary.x), vscale(n1, bary.y)), vscale(n2, bary.z));

    orc_material m = r->scene.materials[tri[3] < ORC_MAX_LIGHTS ? tri[3] : 0]; /* :96 (cbuffer of MAX_LIGHTS entries, :8) */
    float t[4];
    if (m.tex[0] >= 0) {                                                 /* :99-100 diffuse: whole float4 */
        sample_bilinear(&r->scene, 0, tu, tv, m.tex[0], t);
        m.color[0] = t[0]; m.color[1] = t[1]; m.color[2] = t[2]; m.color[3] = t[3];
    }
    if (m.tex[1] >= 0) {                                                 /* :102-107 metallic = .x, roughness = .y (quirk Q11) */
        sample_bilinear(&r->scene, 1, tu, tv, m.tex[1], t);
        m.metallic = t[0]; m.roughness = t[1];
    }
    if (m.tex[2] >= 0) {                                                 /* :109-124 normal map */
        sample_bilinear(&r->scene, 2, tu, tv, m.tex[2], t);
        v3 data = V(t[0] * 2.0f - 1.0f, t[1] * 2.0f - 1.0f, t[2] * 2.0f - 1.0f); /* :112 */
        v3 rayDirection = ld3(r, F_RAY_DIRECTION, index);                /* :115 */
        v3 ortNormal = vdot(normal, rayDirection) <= 0.0f ? normal : vscale(normal, -1.0f); /* :116 */
        v3 up = fabsf(ortNormal.z) < 0.999f ? V(0, 0, 1) : V(1, 0, 0);   /* :119 */
        v3 tangent = vnormalize(vcross(up, ortNormal));                  /* :120 */
        v3 bitangent = vcross(ortNormal, tangent);                       /* :121 */
        normal = vadd(vadd(vscale(tangent, data.x), vscale(bitangent, data.y)), vscale(ortNormal, data.z)); /* :123 (not renormalised) */
    }
    float rough = o_max(0.014f, m.roughness);                            /* :126 */

    st3(r, F_MAT_COLOR, index, V(m.color[0], m.color[1], m.color[2]));   /* :128 */
    float* mr = (float*)fld(r, F_MAT_MR, 8, index);                      /* :129 */
    mr[0] = m.metallic; mr[1] = rough;
    st3(r, F_NORMAL, index, normal);                                     /* :130 */
    return m.type;                                                       /* :132 */
}

/* logic.hlsl:135-163 createShadowRay */
static void create_shadow_ray(orc_renderer* r, rng_t* g, uint32_t index)
{
    uint32_t lightIndex = (uint32_t)(rng_next(g) * (float)r->cam.lightCount); /* :137 */
    float z = 1.0f - 2.0f * rng_next(g);                                 /* :140 */
    float rr = sqrtf(o_max(0.0f, 1.0f - z * z));                         /* :141 */
    float phi = 2.0f * O_PI * rng_next(g);                               /* :142 */
    float x = rr * o_cos(phi);                                           /* :143 */
    float y = rr * o_sin(phi);                                           /* :144 */

    const orc_light* L = &r->scene.lights[lightIndex < ORC_MAX_LIGHTS ? lightIndex : ORC_MAX_LIGHTS - 1];
    v3 lightPosition = vadd(V(L->position[0], L->position[1], L->position[2]), vscale(V(x, y, z), L->radius)); /* :146 */

    v3 normal = ld3(r, F_NORMAL, index);                                 /* :149 */
    v3 surfacePos = vadd(ld3(r, F_SURFACEPOINT, index), vscale(normal, O_EPSILON_OFFSET)); /* :150 */
    v3 lightDir = vsub(lightPosition, surfacePos);                       /* :151 */
    float distance = vlength(lightDir);                                  /* :152 */
    lightDir = vnormalize(lightDir);                                     /* :154 */

    st1u(r, F_LIGHT_INDEX, index, lightIndex);                           /* :159 */
    st3(r, F_SHADOWRAY_ORIGIN, index, surfacePos);                       /* :160 */
    st3(r, F_SHADOWRAY_DIRECTION, index, lightDir);                      /* :161 */
    st1f(r, F_LIGHT_DISTANCE, index, distance - O_EPSILON_OFFSET);       /* :162 */
}

/* logic.hlsl:192-197 sampleLight */
static v3 sample_light(orc_renderer* r, uint32_t li)
{
    const orc_light* L = &r->scene.lights[li < ORC_MAX_LIGHTS ? li : ORC_MAX_LIGHTS - 1];
    v3 e = V(L->emission[0], L->emission[1], L->emission[2]);
    float emax = o_max(e.x, o_max(e.y, e.z));
    return V(e.x / emax, e.y / emax, e.z / emax);
}

/* logic.hlsl:200-302 main, canonical schedule: index ascending over the live slots */
void orc_stage_logic(orc_renderer* r)
{
    if (r->cam.sampleCounter == 0) { clear_texture(r); return; }         /* :206-209 */

    uint32_t* qNew = qptr(r, Q_NEWPATH);
    uint32_t* qUE4 = qptr(r, Q_MAT_UE4);
    uint32_t* qGlass = qptr(r, Q_MAT_GLASS);

    /* Two passes with identical results: the per-slot work (everything that only touches the slot's own state) runs on all host
     * cores; queue pushes and framebuffer updates follow serially in ascending slot order -- the canonical schedule. */
    enum { K_SKIP = 0, K_ENDED = 1, K_UE4 = 2, K_GLASS = 3, K_OTHER = 4 };
    const int nthreads = r->cfg.threads > 1 ? (int)r->cfg.threads : 1;
    const int64_t live = (int64_t)r->cfg.livePaths;
#pragma omp parallel for schedule(static) num_threads(nthreads)
    for (int64_t idx = 0; idx < live; idx++) {                           /* :212-214 */
        const uint32_t index = (uint32_t)idx;
        r->scratchKind[index] = K_SKIP;
        if (r->retired[index]) continue;                                 /* extension: pathBudget */
        rng_t g; rng_seed(&g, &r->cam, index);                           /* :216 */
        int pathEliminated = 0;                                          /* :217 */
        v3 throughput = ld3(r, F_THROUGHPUT, index);                     /* :219 */
        v3 radiance = ld3(r, F_RADIANCE, index);                         /* :220 */
        uint32_t isEmitter = ld1u(r, F_ISEMITTER, index);

        if (isEmitter > 0) {                                             /* :222-226 */
            radiance = vadd(radiance, vmul(sample_light(r, isEmitter - 1), throughput));
            pathEliminated = 1;
        } else {
            if (!ld1u(r, F_INSHADOW, index))                             /* :230-231 */
                radiance = vadd(radiance, vmul(ld3(r, F_DIRECT_LIGHT, index), throughput));
            throughput = vmul(throughput, ld3(r, F_LIGHT_THROUGHPUT, index)); /* :234 */
            if (throughput.x <= 0.0f && throughput.y <= 0.0f && throughput.z <= 0.0f) /* :237 */
                pathEliminated = 1;
            if (ld1f(r, F_HITDISTANCE, index) == FLT_MAX) {              /* :241-245 */
                radiance = vadd(radiance, vmul(throughput, V(r->cam.envColor[0], r->cam.envColor[1], r->cam.envColor[2])));
                pathEliminated = 1;
            }
            uint32_t pl = ld1u(r, F_PATH_LENGTH, index);
            if (pl > 200) {                                              /* :248-255 */
                float p = o_max(throughput.x, o_max(throughput.y, throughput.z));
                if (rng_next(&g) > p * 0.004f) pathEliminated = 1;
                throughput = vscale(throughput, 1.0f / p);
            }
            if (r->cfg.maxDepth && pl >= r->cfg.maxDepth) pathEliminated = 1; /* extension (config 5) */
        }

        if (pathEliminated) {                                            /* :259 endPath, applied in the second pass */
            r->scratchKind[index] = K_ENDED;
            r->scratchRad[3 * (size_t)index] = radiance.x; r->scratchRad[3 * (size_t)index + 1] = radiance.y; r->scratchRad[3 * (size_t)index + 2] = radiance.z;
            continue;
        }
        uint32_t materialType = set_material_hit_properties(r, index);   /* :262 */
        r->scratchKind[index] = materialType == 0 ? K_UE4 : (materialType == 1 ? K_GLASS : K_OTHER);

        create_shadow_ray(r, &g, index);                                 /* :291 */
        uint32_t pathLength = ld1u(r, F_PATH_LENGTH, index) + 1;         /* :293 */
        st3(r, F_RADIANCE, index, radiance);                             /* :295 */
        st3(r, F_THROUGHPUT, index, throughput);                         /* :296 */
        st1u(r, F_PATH_LENGTH, index, pathLength);                       /* :297 */
        st1u(r, F_INSHADOW, index, 1u);                                  /* :298 */
    }
    for (uint32_t index = 0; index < r->cfg.livePaths; index++) {        /* ascending slot order */
        const uint8_t k = r->scratchKind[index];
        if (k == K_SKIP) continue;
        r->stats.segments++;
        if (k == K_ENDED) {
            const float* rad = r->scratchRad + 3 * (size_t)index;
            accumulate_sample(r, V(rad[0], rad[1], rad[2]), index);      /* :49-73 */
            qNew[r->qc[QC_NEWPATH]++] = index;                           /* :75 */
            r->stats.pathsEnded++;
        }
        else if (k == K_UE4) qUE4[r->qc[QC_MATUE4]++] = index;           /* :282-283 */
        else if (k == K_GLASS) qGlass[r->qc[QC_MATGLASS]++] = index;     /* :284-285 */
    }
}

/* ------------------------------------------------------------------ stage 2: newPath.hlsl:14-61 */
void orc_stage_new_path(orc_renderer* r)
{
    uint32_t queueElementCount = r->qc[QC_NEWPATH];                      /* :18 */
    uint32_t lastPath = r->qc[QC_LASTPATHCNT];                           /* :19 */
    uint32_t* qNew = qptr(r, Q_NEWPATH);
    uint32_t* qExt = qptr(r, Q_EXT_RAY);
    uint32_t w, h, x0, y0;
    tile_dims(r, &w, &h, &x0, &y0);
    v3 ulc = V(r->cam.ulc[0], r->cam.ulc[1], r->cam.ulc[2]);
    v3 hor = V(r->cam.horizontal[0], r->cam.horizontal[1], r->cam.horizontal[2]);
    v3 ver = V(r->cam.vertical[0], r->cam.vertical[1], r->cam.vertical[2]);
    uint32_t generated = 0;

    for (uint32_t queueIndex = 0; queueIndex < r->cfg.livePaths; queueIndex++) { /* :21-25 */
        if (queueIndex >= queueElementCount) break;
        rng_t g; rng_seed(&g, &r->cam, queueIndex);                      /* :27 */
        uint32_t index = qNew[queueIndex];                               /* :29 */

        if (r->cfg.pathBudget && (uint32_t)(lastPath + queueIndex) >= r->cfg.pathBudget) {
            /* extension: budget exhausted -> retire the slot instead of regenerating it */
            r->retired[index] = 1;
            r->activePaths--;
            qExt[queueIndex] = 0xFFFFFFFFu; /* hole in the extension queue */
            continue;
        }
        uint32_t newIndex = (lastPath + queueIndex) % (w * h);           /* :33 */
        uint32_t cx = x0 + newIndex % w, cy = y0 + newIndex / w;         /* :34 */

        float jx = rng_next(&g) * 2.0f - 1.0f;                           /* :36 */
        float jy = rng_next(&g) * 2.0f - 1.0f;
        float u = ((float)cx + jx) * r->cam.pixelSize[0];                /* :37 */
        float v = ((float)cy + jy) * r->cam.pixelSize[1];
        /* :39 normalize(cam.ulc + uv.x * cam.horizontal - uv.y * cam.vertical) */
        v3 dir = vnormalize(vsub(vadd(ulc, vscale(hor, u)), vscale(ver, v)));

        st3(r, F_RAY_ORIGIN, index, V(r->cam.pos[0], r->cam.pos[1], r->cam.pos[2])); /* :41 */
        st3(r, F_RAY_DIRECTION, index, dir);                             /* :42 */
        uint32_t* sc = (uint32_t*)fld(r, F_SCREEN_COORD, 8, index); sc[0] = cx; sc[1] = cy; /* :43 */
        st3(r, F_RADIANCE, index, V(0, 0, 0));                           /* :44 */
        st3(r, F_THROUGHPUT, index, V(1, 1, 1));                         /* :45 */
        st3(r, F_LIGHT_THROUGHPUT, index, V(1, 1, 1));                   /* :46 */
        st1u(r, F_PATH_LENGTH, index, 0);                                /* :47 */
        st1u(r, F_INSHADOW, index, 1u);                                  /* :48 */
        qExt[queueIndex] = index;                                        /* :51 */
        generated++;
    }
    r->stats.pathsGenerated += generated;
    /* :55-60 thread 0: offsets for the UE4 / glass parts of the extension queue, shadow counter reset */
    r->qc[QC_EXT_UE4_OFFSET] = queueElementCount;
    r->qc[QC_EXT_GLASS_OFFSET] = queueElementCount + r->qc[QC_MATUE4];
    r->qc[QC_SHADOWRAY] = 0;
}

/* ------------------------------------------------------------------ stage 3: materialUE4.hlsl */
typedef struct { v3 rayDir; v3 baseColor; float metallic, roughness; v3 normal; } ue4_state;

static void make_basis(v3 N, v3* tangent, v3* bitangent) /* materialUE4.hlsl:35-37 */
{
    v3 up = fabsf(N.z) < 0.999f ? V(0, 0, 1) : V(1, 0, 0);
    *tangent = vnormalize(vcross(up, N));
    *bitangent = vcross(N, *tangent);
}

static v3 ue4_sample(const ue4_state* st, rng_t* g) /* materialUE4.hlsl:24-68 */
{
    v3 N = st->normal;
    v3 Vv = vneg(st->rayDir);
    float r0 = rng_next(g), r1 = rng_next(g);                            /* :29 */
    float diffuseRatio = 1.0f - st->metallic;                            /* :31 */
    v3 tangent, bitangent; make_basis(N, &tangent, &bitangent);
    v3 direction;
    if (rng_next(g) < diffuseRatio) {                                    /* :40 */
        float x = sqrtf(r0);                                             /* :42 */
        float phi = 2.0f * O_PI * r1;                                    /* :43 */
        float dx = x * o_cos(phi);                                       /* :44 */
        float dy = x * o_sin(phi);                                       /* :45 */
        float dz = sqrtf(o_max(0.0f, 1.0f - dx * dx - dy * dy));         /* :46 */
        direction = vadd(vadd(vscale(tangent, dx), vscale(bitangent, dy)), vscale(N, dz)); /* :48 */
    } else {
        float a = st->roughness * st->roughness;                         /* :52 */
        float phi = 2.0f * O_PI * r0;                                    /* :53 */
        float cosTheta = sqrtf((1.0f - r1) / (1.0f + (a * a - 1.0f) * r1)); /* :55 */
        float sinTheta = sqrtf(1.0f - cosTheta * cosTheta);              /* :56 */
        float hx = sinTheta * o_cos(phi);                                /* :59 */
        float hy = sinTheta * o_sin(phi);                                /* :60 */
        float hz = cosTheta;                                             /* :61 */
        direction = vadd(vadd(vscale(tangent, hx), vscale(bitangent, hy)), vscale(N, hz)); /* :63 */
        direction = vsub(vscale(direction, 2.0f * vdot(Vv, direction)), Vv); /* :64 */
    }
    return direction;
}

static float ue4_pdf(const ue4_state* st, v3 direction) /* materialUE4.hlsl:70-90 */
{
    v3 N = st->normal, Vv = vneg(st->rayDir), L = direction;
    float diffuseRatio = 1.0f - st->metallic;                            /* :76 */
    float specularRatio = 1.0f - diffuseRatio;                           /* :77 */
    v3 H = vnormalize(vadd(L, Vv));                                      /* :79 */
    float NdotH = fabsf(vdot(N, H));                                     /* :81 */
    float pdfGGXTR = GGXTrowbridgeReitz(NdotH, st->roughness * st->roughness) * NdotH; /* :82 */
    float pdfSpec = pdfGGXTR / (4.0f * fabsf(vdot(Vv, H)));              /* :85 */
    float pdfDiff = fabsf(vdot(L, N)) * (1.0f / O_PI);                   /* :86 */
    return diffuseRatio * pdfDiff + specularRatio * pdfSpec;             /* :89 */
}

static v3 ue4_evaluate(const ue4_state* st, v3 direction) /* materialUE4.hlsl:92-115 */
{
    v3 N = st->normal, Vv = vneg(st->rayDir), L = direction;
    float NdotL = vdot(N, L), NdotV = vdot(N, Vv);                       /* :98-99 */
    if (NdotL <= 0.0f || NdotV <= 0.0f) return V(0, 0, 0);               /* :100-101 */
    v3 H = vnormalize(vadd(L, Vv));                                      /* :103 */
    float NdotH = vdot(N, H), LdotH = vdot(L, H);                        /* :104-105 */
    float D = GGXTrowbridgeReitz(NdotH, st->roughness * st->roughness);  /* :107 */
    float G = smithSchlickGGX(NdotL, st->roughness) * smithSchlickGGX(NdotV, st->roughness); /* :108 */
    /* :110 lerp(0.037, base, metallic) = x + s*(y - x) */
    v3 sc = V(0.037f + st->metallic * (st->baseColor.x - 0.037f),
              0.037f + st->metallic * (st->baseColor.y - 0.037f),
              0.037f + st->metallic * (st->baseColor.z - 0.037f));
    float w = 1.0f - LdotH;                                              /* :111 pow(.,5) as a multiplication chain */
    float w2 = w * w;
    float fc = w2 * w2 * w;
    v3 F = V((1.0f - fc) * sc.x + fc, (1.0f - fc) * sc.y + fc, (1.0f - fc) * sc.z + fc); /* :112 */
    /* :114 (base / PI) * (1 - metallic) + (D * F * G) / (4 * NdotL * NdotV) */
    float den = 4.0f * NdotL * NdotV;
    float om = 1.0f - st->metallic;
    return V((st->baseColor.x / O_PI) * om + (D * F.x * G) / den,
             (st->baseColor.y / O_PI) * om + (D * F.y * G) / den,
             (st->baseColor.z / O_PI) * om + (D * F.z * G) / den);
}

void orc_stage_material_ue4(orc_renderer* r) /* materialUE4.hlsl:118-192 */
{
    uint32_t queueElementCount = r->qc[QC_MATUE4];                       /* :122 */
    uint32_t extQueueOffset = r->qc[QC_EXT_UE4_OFFSET];                  /* :123 */
    uint32_t* qUE4 = qptr(r, Q_MAT_UE4);
    uint32_t* qExt = qptr(r, Q_EXT_RAY);
    uint32_t* qSh = qptr(r, Q_SHADOW_RAY);

    /* per-item work on all host cores; the shadow queue is then filled serially in ascending queue order (canonical schedule) */
    const int nthreads = r->cfg.threads > 1 ? (int)r->cfg.threads : 1;
    const int64_t items = (int64_t)(queueElementCount < r->cfg.livePaths ? queueElementCount : r->cfg.livePaths); /* :125-129 */
#pragma omp parallel for schedule(static) num_threads(nthreads)
    for (int64_t qi = 0; qi < items; qi++) {
        const uint32_t queueIndex = (uint32_t)qi;
        rng_t g; rng_seed(&g, &r->cam, queueIndex);                      /* :131 */
        uint32_t index = qUE4[queueIndex];                               /* :132 */

        ue4_state st;
        st.rayDir = ld3(r, F_RAY_DIRECTION, index);                      /* :140 */
        st.normal = ld3(r, F_NORMAL, index);                             /* :141 */
        float* mr = (float*)fld(r, F_MAT_MR, 8, index);                  /* :143 */
        st.baseColor = ld3(r, F_MAT_COLOR, index);                       /* :144 */
        st.metallic = mr[0]; st.roughness = mr[1];                       /* :145-146 */

        v3 bsdfDir = ue4_sample(&st, &g);                                /* :148 */
        float pdf = ue4_pdf(&st, bsdfDir);                               /* :149 */
        v3 throughput = V(0, 0, 0);
        if (pdf > 0.0f) {                                                /* :151-152 */
            v3 e = ue4_evaluate(&st, bsdfDir);
            float an = fabsf(vdot(st.normal, bsdfDir));
            throughput = V(e.x * an / pdf, e.y * an / pdf, e.z * an / pdf);
        }
        st3(r, F_LIGHT_THROUGHPUT, index, throughput);                   /* :154 */

        v3 surfacePoint = ld3(r, F_SURFACEPOINT, index);                 /* :157 */
        st3(r, F_RAY_ORIGIN, index, vadd(surfacePoint, vscale(bsdfDir, O_EPSILON_OFFSET))); /* :158,160 */
        st3(r, F_RAY_DIRECTION, index, bsdfDir);                         /* :161 */
        qExt[extQueueOffset + queueIndex] = index;                       /* :162 */

        v3 lightDir = ld3(r, F_SHADOWRAY_DIRECTION, index);              /* :165 */
        int legitLight = vdot(lightDir, st.normal) > 0.0f;               /* :167 */
        r->scratchKind[queueIndex] = (uint8_t)legitLight;
        if (legitLight) {                                                /* :178-190 */
            uint32_t lightIndex = ld1u(r, F_LIGHT_INDEX, index);
            float distance = ld1f(r, F_LIGHT_DISTANCE, index);
            const orc_light* L = &r->scene.lights[lightIndex < ORC_MAX_LIGHTS ? lightIndex : ORC_MAX_LIGHTS - 1];
            float lightPdf = distance * distance / (4.0f * O_PI * L->radius * L->radius); /* :184 */
            float bsdfPdf = ue4_pdf(&st, lightDir);                      /* :185 */
            /* :187 powerHeuristic * eval * emission * lightCount * falloff, left to right */
            float ph = powerHeuristic(lightPdf, bsdfPdf);
            v3 e = ue4_evaluate(&st, lightDir);
            float lc = (float)r->cam.lightCount;
            float fo = lightFalloff(distance, L->falloff);
            v3 dl = V(ph * e.x * L->emission[0] * lc * fo, ph * e.y * L->emission[1] * lc * fo, ph * e.z * L->emission[2] * lc * fo);
            st3(r, F_DIRECT_LIGHT, index, dl);                           /* :188 */
        }
    }
    for (uint32_t queueIndex = 0; queueIndex < (uint32_t)items; queueIndex++)
        if (r->scratchKind[queueIndex]) qSh[r->qc[QC_SHADOWRAY]++] = qUE4[queueIndex]; /* :189 */
}

/* ------------------------------------------------------------------ stage 4: materialGlass.hlsl */
static v3 hlsl_reflect(v3 i, v3 n) { return vsub(i, vscale(n, 2.0f * vdot(n, i))); } /* i - 2*n*dot(i,n): HLSL intrinsic */
static v3 hlsl_refract(v3 i, v3 n, float eta)
{
    /* HLSL intrinsic: k = 1 - eta^2 (1 - dot(n,i)^2); k < 0 -> 0 ; eta*i - (eta*dot(n,i) + sqrt(k))*n */
    float d = vdot(n, i);
    float k = 1.0f - eta * eta * (1.0f - d * d);
    if (k < 0.0f) return V(0, 0, 0);
    return vsub(vscale(i, eta), vscale(n, eta * d + sqrtf(k)));
}

static v3 glass_sample(v3 stNormal, v3 rayDir, rng_t* g) /* materialGlass.hlsl:23-46 */
{
    v3 normal = vdot(stNormal, rayDir) <= 0.0f ? stNormal : vscale(stNormal, -1.0f); /* :25 */
    const float n1 = 1.0f, n2 = 1.458f;                                  /* :28-29 */
    float r0 = (n1 - n2) / (n1 + n2);                                    /* :31 */
    r0 = r0 * r0;                                                        /* :32 */
    float theta = vdot(vneg(rayDir), normal);                            /* :34 */
    float probability = schlickFresnel3(r0, theta).x;                    /* :36 */
    float refractFactor = vdot(stNormal, normal) > 0.0f ? (n1 / n2) : (n2 / n1); /* :38 */
    v3 transDirection = vnormalize(hlsl_refract(rayDir, normal, refractFactor)); /* :39 */
    float cos2t = 1.0f - refractFactor * refractFactor * (1.0f - theta * theta); /* :40 */
    /* :42 HLSL || does not short-circuit: rand() is always consumed */
    float rnd = rng_next(g);
    if (cos2t < 0.0f || rnd < probability)
        return vnormalize(hlsl_reflect(rayDir, normal));                 /* :43 */
    return transDirection;                                               /* :45 */
}

void orc_stage_material_glass(orc_renderer* r) /* materialGlass.hlsl:48-85 */
{
    uint32_t queueElementCount = r->qc[QC_MATGLASS];                     /* :52 */
    uint32_t extQueueOffset = r->qc[QC_EXT_GLASS_OFFSET];                /* :53 */
    uint32_t* qGlass = qptr(r, Q_MAT_GLASS);
    uint32_t* qExt = qptr(r, Q_EXT_RAY);
    for (uint32_t queueIndex = 0; queueIndex < r->cfg.livePaths; queueIndex++) { /* :55-59 */
        if (queueIndex >= queueElementCount) break;
        rng_t g; rng_seed(&g, &r->cam, queueIndex);                      /* :61 */
        uint32_t index = qGlass[queueIndex];                             /* :62 */
        v3 rayDir = ld3(r, F_RAY_DIRECTION, index);                      /* :69 */
        v3 normal = ld3(r, F_NORMAL, index);                             /* :70 */
        v3 baseColor = ld3(r, F_MAT_COLOR, index);                       /* :71 */
        v3 bsdfDir = glass_sample(normal, rayDir, &g);                   /* :73 */
        st3(r, F_LIGHT_THROUGHPUT, index, baseColor);                    /* :75 */
        v3 surfacePoint = ld3(r, F_SURFACEPOINT, index);                 /* :78 */
        st3(r, F_RAY_ORIGIN, index, vadd(surfacePoint, vscale(bsdfDir, O_EPSILON_OFFSET))); /* :79,81 */
        st3(r, F_RAY_DIRECTION, index, bsdfDir);                         /* :82 */
        qExt[extQueueOffset + queueIndex] = index;                       /* :83 */
    }
}

/* ------------------------------------------------------------------ stages 5/6: ray casts */
static inline float ray_aabb(const float* mn, const float* mx, v3 o, v3 invdir) /* extensionRayCast.hlsl:79-94 == shadowRayCast.hlsl:49-63 */
{
    float fx = (mx[0] - o.x) * invdir.x, fy = (mx[1] - o.y) * invdir.y, fz = (mx[2] - o.z) * invdir.z;
    float nx = (mn[0] - o.x) * invdir.x, ny = (mn[1] - o.y) * invdir.y, nz = (mn[2] - o.z) * invdir.z;
    float tmaxx = o_max(fx, nx), tmaxy = o_max(fy, ny), tmaxz = o_max(fz, nz);
    float tminx = o_min(fx, nx), tminy = o_min(fy, ny), tminz = o_min(fz, nz);
    float t1 = o_min(tmaxx, o_min(tmaxy, tmaxz));
    float t0 = o_max(tminx, o_max(tminy, tminz));
    return (t1 >= t0) ? (t0 > 0.0f ? t0 : t1) : -1.0f;
}

typedef struct { uint64_t inner, leaves, tris; uint32_t maxStack; } trav_stats;

typedef struct { v3 hitPoint, bary; orc_triangle tri; } ext_hit;

/* extensionRayCast.hlsl:38-77 */
static inline int ext_ray_triangle(v3 o, v3 d, v3 v0, v3 v1, v3 v2, float* distance, ext_hit* hit)
{
    v3 e1 = vsub(v1, v0), e2 = vsub(v2, v0);
    v3 pvec = vcross(d, e2);
    float det = vdot(e1, pvec);
    if (det > -O_EPSILON && det < O_EPSILON) return 0;
    float invDet = 1.0f / det;
    v3 tvec = vsub(o, v0);
    float u = vdot(tvec, pvec) * invDet;
    if (u < 0.0f || u > 1.0f) return 0;
    v3 qvec = vcross(tvec, e1);
    float v = vdot(d, qvec) * invDet;
    if (v < 0.0f || u + v > 1.0f) return 0;
    float t = vdot(e2, qvec) * invDet;
    if (t >= 0.0f) {
        v3 pp = vadd(o, vscale(d, t));
        if (t < *distance) {
            *distance = t;
            hit->hitPoint = pp;
            hit->bary = V(1.0f - u - v, u, v);
            return 1;
        }
    }
    return 0;
}

/* extensionRayCast.hlsl:96-166 */
static float ext_bvh(const orc_scene* sc, v3 o, v3 d, ext_hit* hit, uint32_t stackSize, trav_stats* ts)
{
    int stack[256];
    uint32_t ptr = 0;
    stack[ptr++] = -1;
    float distance = FLT_MAX;
    v3 invdir = V(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
    const orc_bvh_node* tree = sc->nodes;
    const float* vt = sc->verts;

    if (ray_aabb(tree[0].min, tree[0].max, o, invdir) > 0.0f) {
        for (int idx = 0; idx > -1;) {
            const orc_bvh_node* node = &tree[idx];
            if (node->isLeaf) {
                ts->leaves++;
                for (int i = node->left; i < node->right; i++) {
                    const orc_triangle* T = &sc->tris[i];
                    v3 v0 = V(vt[3 * T->v[0]], vt[3 * T->v[0] + 1], vt[3 * T->v[0] + 2]);
                    v3 v1 = V(vt[3 * T->v[1]], vt[3 * T->v[1] + 1], vt[3 * T->v[1] + 2]);
                    v3 v2 = V(vt[3 * T->v[2]], vt[3 * T->v[2] + 1], vt[3 * T->v[2] + 2]);
                    ts->tris++;
                    if (ext_ray_triangle(o, d, v0, v1, v2, &distance, hit)) hit->tri = *T;
                }
            } else {
                ts->inner++;
                const orc_bvh_node* left = &tree[node->left];
                const orc_bvh_node* right = &tree[node->right];
                float leftHit = ray_aabb(left->min, left->max, o, invdir);
                float rightHit = ray_aabb(right->min, right->max, o, invdir);
                if (leftHit > 0.0f && rightHit > 0.0f) {
                    int deferred;
                    if (leftHit > rightHit) { idx = node->right; deferred = node->left; }
                    else { idx = node->left; deferred = node->right; }
                    if (ptr < stackSize) stack[ptr] = deferred; /* the reference writes unchecked (quirk Q23) */
                    ptr++;
                    if (ptr > ts->maxStack) ts->maxStack = ptr;
                    continue;
                } else if (leftHit > 0.0f) { idx = node->left; continue; }
                else if (rightHit > 0.0f) { idx = node->right; continue; }
            }
            --ptr;
            idx = ptr < stackSize ? stack[ptr] : -1;
        }
    }
    return distance;
}

/* extensionRayCast.hlsl:168-194 */
static void ray_lights(const orc_scene* sc, uint32_t lightCount, v3 o, v3 d, uint32_t* lightIndex, float* distance)
{
    for (uint32_t i = 0; i < lightCount && i < ORC_MAX_LIGHTS; i++) {
        const orc_light* L = &sc->lights[i];
        v3 position = vsub(V(L->position[0], L->position[1], L->position[2]), o);
        float radius2 = L->radius * L->radius;
        float tca = vdot(position, d);
        float d2 = vdot(position, position) - tca * tca;
        if (d2 > radius2) continue;
        float thc = sqrtf(radius2 - d2);
        float t0 = tca - thc;
        float t1 = tca + thc;
        if (t0 < 0.0f) t0 = t1;
        if (t0 > 0.0f && t0 < *distance) { *distance = t0; *lightIndex = i + 1; }
    }
}

/* extensionRayCast.hlsl:197-234 -- "Always going through all paths" in extension-queue order */
void orc_stage_extension(orc_renderer* r)
{
    uint32_t* qExt = qptr(r, Q_EXT_RAY);
    uint32_t count = r->cfg.livePaths;
    if (r->cfg.pathBudget) {
        /* extension: with retired slots the queue holds only [new | UE4 | glass] live entries */
        count = r->qc[QC_EXT_GLASS_OFFSET] + r->qc[QC_MATGLASS];
        if (count > r->cfg.livePaths) count = r->cfg.livePaths;
    }
    r->qc[QC_EXT_COUNT] = count;
    uint32_t stackSize = r->cfg.stackSize ? r->cfg.stackSize : 64;
    uint64_t inner = 0, leaves = 0, tris = 0, rays = 0; uint32_t maxStack = r->stats.maxStack;
    int nthreads = r->cfg.threads > 1 ? (int)r->cfg.threads : 1;
    (void)nthreads;
#pragma omp parallel for schedule(dynamic, 256) num_threads(nthreads) reduction(+:inner,leaves,tris,rays) reduction(max:maxStack)
    for (uint32_t queueIndex = 0; queueIndex < count; queueIndex++) {
        uint32_t index = qExt[queueIndex];                               /* :210 */
        if (index == 0xFFFFFFFFu) continue;                              /* extension: hole left by a retired slot */
        rays++;
        v3 o = ld3(r, F_RAY_ORIGIN, index), d = ld3(r, F_RAY_DIRECTION, index); /* :213-214 */
        ext_hit hit; memset(&hit, 0, sizeof(hit)); trav_stats ts = { 0, 0, 0, 0 };
        float distance = ext_bvh(&r->scene, o, d, &hit, stackSize, &ts); /* :216 */
        if (distance < FLT_MAX) {                                        /* :218-225 */
            st3(r, F_SURFACEPOINT, index, hit.hitPoint);
            st3(r, F_BARYCOORD, index, hit.bary);
            uint32_t* tri = (uint32_t*)fld(r, F_TRIANGLE, 16, index);
            tri[0] = (uint32_t)hit.tri.v[0]; tri[1] = (uint32_t)hit.tri.v[1]; tri[2] = (uint32_t)hit.tri.v[2]; tri[3] = hit.tri.materialID;
        }
        uint32_t lightIndex = 0;                                         /* :228-229 */
        ray_lights(&r->scene, r->cam.lightCount, o, d, &lightIndex, &distance);
        st1u(r, F_ISEMITTER, index, lightIndex);                         /* :231 */
        st1f(r, F_HITDISTANCE, index, distance);                         /* :232 */
        inner += ts.inner; leaves += ts.leaves; tris += ts.tris;
        if (ts.maxStack > maxStack) maxStack = ts.maxStack;
    }
    r->stats.extRays += rays; r->stats.extInner += inner; r->stats.extLeaves += leaves; r->stats.extTris += tris;
    r->stats.maxStack = maxStack;
}

/* shadowRayCast.hlsl:16-47 */
static inline int sh_ray_triangle(v3 o, v3 d, v3 v0, v3 v1, v3 v2, float* distance)
{
    v3 e1 = vsub(v1, v0), e2 = vsub(v2, v0);
    v3 pvec = vcross(d, e2);
    float det = vdot(e1, pvec);
    if (det > -O_EPSILON && det < O_EPSILON) return 0;
    float invDet = 1.0f / det;
    v3 tvec = vsub(o, v0);
    float u = vdot(tvec, pvec) * invDet;
    if (u < 0.0f || u > 1.0f) return 0;
    v3 qvec = vcross(tvec, e1);
    float v = vdot(d, qvec) * invDet;
    if (v < 0.0f || u + v > 1.0f) return 0;
    float t = vdot(e2, qvec) * invDet;
    if (t > O_EPSILON && t < 1.0f / O_EPSILON) {
        *distance = vlength(vscale(d, t));
        return 1;
    }
    return 0;
}

/* shadowRayCast.hlsl:65-136 */
static int sh_bvh(const orc_scene* sc, v3 o, v3 d, float lightDistance, uint32_t stackSize, trav_stats* ts)
{
    int stack[256];
    uint32_t ptr = 0;
    stack[ptr++] = -1;
    v3 invdir = V(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
    const orc_bvh_node* tree = sc->nodes;
    const float* vt = sc->verts;
    if (ray_aabb(tree[0].min, tree[0].max, o, invdir) > 0.0f) {
        for (int idx = 0; idx > -1;) {
            const orc_bvh_node* node = &tree[idx];
            if (node->isLeaf) {
                ts->leaves++;
                for (int i = node->left; i < node->right; i++) {
                    const orc_triangle* T = &sc->tris[i];
                    v3 v0 = V(vt[3 * T->v[0]], vt[3 * T->v[0] + 1], vt[3 * T->v[0] + 2]);
                    v3 v1 = V(vt[3 * T->v[1]], vt[3 * T->v[1] + 1], vt[3 * T->v[1] + 2]);
                    v3 v2 = V(vt[3 * T->v[2]], vt[3 * T->v[2] + 1], vt[3 * T->v[2] + 2]);
                    float distance = FLT_MAX;
                    ts->tris++;
                    if (sh_ray_triangle(o, d, v0, v1, v2, &distance) && distance < lightDistance) return 1;
                }
            } else {
                ts->inner++;
                const orc_bvh_node* left = &tree[node->left];
                const orc_bvh_node* right = &tree[node->right];
                float leftHit = ray_aabb(left->min, left->max, o, invdir);
                float rightHit = ray_aabb(right->min, right->max, o, invdir);
                if (leftHit > 0.0f && rightHit > 0.0f) {
                    int deferred;
                    if (leftHit > rightHit) { idx = node->right; deferred = node->left; }
                    else { idx = node->left; deferred = node->right; }
                    if (ptr < stackSize) stack[ptr] = deferred;
                    ptr++;
                    if (ptr > ts->maxStack) ts->maxStack = ptr;
                    continue;
                } else if (leftHit > 0.0f) { idx = node->left; continue; }
                else if (rightHit > 0.0f) { idx = node->right; continue; }
            }
            --ptr;
            idx = ptr < stackSize ? stack[ptr] : -1;
        }
    }
    return 0;
}

/* shadowRayCast.hlsl:139-169 */
void orc_stage_shadow(orc_renderer* r)
{
    /* :144-148 thread 0: QC[0..3] = (0, QC1 + QC0, 0, 0) */
    uint32_t q0 = r->qc[QC_NEWPATH], q1 = r->qc[QC_LASTPATHCNT];
    r->qc[QC_NEWPATH] = 0; r->qc[QC_LASTPATHCNT] = q0 + q1; r->qc[QC_MATUE4] = 0; r->qc[QC_MATGLASS] = 0;

    uint32_t queueElementCount = r->qc[QC_SHADOWRAY];                    /* :151 */
    uint32_t* qSh = qptr(r, Q_SHADOW_RAY);
    uint32_t count = queueElementCount < r->cfg.livePaths ? queueElementCount : r->cfg.livePaths; /* :153-157 */
    uint32_t stackSize = r->cfg.stackSize ? r->cfg.stackSize : 64;
    uint64_t inner = 0, leaves = 0, tris = 0; uint32_t maxStack = r->stats.maxStack;
    int nthreads = r->cfg.threads > 1 ? (int)r->cfg.threads : 1;
    (void)nthreads;
#pragma omp parallel for schedule(dynamic, 256) num_threads(nthreads) reduction(+:inner,leaves,tris) reduction(max:maxStack)
    for (uint32_t queueIndex = 0; queueIndex < count; queueIndex++) {
        uint32_t index = qSh[queueIndex];                                /* :159 */
        v3 o = ld3(r, F_SHADOWRAY_ORIGIN, index), d = ld3(r, F_SHADOWRAY_DIRECTION, index); /* :162-163 */
        float lightDistance = ld1f(r, F_LIGHT_DISTANCE, index);          /* :164 */
        trav_stats ts = { 0, 0, 0, 0 };
        int inShadow = sh_bvh(&r->scene, o, d, lightDistance, stackSize, &ts); /* :166 */
        st1u(r, F_INSHADOW, index, (uint32_t)inShadow);                  /* :167 */
        inner += ts.inner; leaves += ts.leaves; tris += ts.tris;
        if (ts.maxStack > maxStack) maxStack = ts.maxStack;
    }
    r->stats.shRays += count; r->stats.shInner += inner; r->stats.shLeaves += leaves; r->stats.shTris += tris;
    r->stats.maxStack = maxStack;
}

/* ------------------------------------------------------------------ driver */
void orc_iterate(orc_renderer* r) /* Renderer.cpp:195-211 */
{
    orc_stage_logic(r);
    orc_stage_new_path(r);
    orc_stage_material_ue4(r);
    orc_stage_material_glass(r);
    orc_stage_extension(r);
    orc_stage_shadow(r);
}

orc_renderer* orc_create(const orc_scene* scene, const orc_config* cfg)
{
    orc_renderer* r = (orc_renderer*)calloc(1, sizeof(*r));
    if (!r) return NULL;
    r->scene = *scene;
    r->cfg = *cfg;
    if (r->cfg.livePaths == 0 || r->cfg.livePaths > r->cfg.poolPaths) r->cfg.livePaths = r->cfg.poolPaths;
    if (r->cfg.stackSize > 256) r->cfg.stackSize = 256;
    /* Renderer.cpp:58-108: buffers are created without initial data; D3D11 zero-fills them */
    r->state = (uint8_t*)calloc((size_t)r->cfg.poolPaths, ORC_STATE_BYTES);
    r->queue = (uint32_t*)calloc((size_t)r->cfg.poolPaths * 5, 4);
    r->fb = (float*)calloc((size_t)r->cfg.fbWidth * r->cfg.fbHeight * 4, 4);
    r->retired = (uint8_t*)calloc(r->cfg.poolPaths, 1);
    r->scratchKind = (uint8_t*)calloc(cfg->poolPaths ? cfg->poolPaths : 1, 1);
    r->scratchRad = (float*)calloc((size_t)(cfg->poolPaths ? cfg->poolPaths : 1) * 3, sizeof(float));
    r->activePaths = r->cfg.livePaths;
    if (!r->state || !r->queue || !r->fb || !r->retired) { orc_destroy(r); return NULL; }
    return r;
}

void orc_destroy(orc_renderer* r)
{
    if (!r) return;
    free(r->state); free(r->queue); free(r->fb); free(r->retired); free(r->scratchKind); free(r->scratchRad); free(r);
}

void orc_set_camera(orc_renderer* r, const orc_camera_buffer* cam)
{
    r->cam = *cam; /* Renderer.cpp:161 UpdateSubresource */
    if (cam->sampleCounter == 0) r->activePaths = r->cfg.livePaths;
}
uint8_t* orc_path_state(orc_renderer* r) { return r->state; }
uint32_t* orc_queues(orc_renderer* r) { return r->queue; }
uint32_t* orc_counters(orc_renderer* r) { return r->qc; }
float* orc_framebuffer(orc_renderer* r) { return r->fb; }
void orc_get_stats(orc_renderer* r, orc_stats* s) { *s = r->stats; }
void orc_reset_stats(orc_renderer* r) { memset(&r->stats, 0, sizeof(r->stats)); }
uint32_t orc_active_paths(orc_renderer* r) { return r->activePaths; }

/* ------------------------------------------------------------------ host camera (Source/Camera.cpp) */
int orc_msvc_rand(uint32_t* state)
{
    /* MSVC CRT rand(): x = x*214013 + 2531011; return (x >> 16) & 0x7FFF; no srand in the reference => seed 1 */
    *state = *state * 214013u + 2531011u;
    return (int)((*state >> 16) & 0x7FFFu);
}

static void n3(float* v) { float l = sqrtf((v[0] * v[0] + v[1] * v[1]) + v[2] * v[2]); v[0] /= l; v[1] /= l; v[2] /= l; } /* XMVector3Normalize */
static void c3(const float* a, const float* b, float* o) { o[0] = a[1] * b[2] - a[2] * b[1]; o[1] = a[2] * b[0] - a[0] * b[2]; o[2] = a[0] * b[1] - a[1] * b[0]; }
static float d3(const float* a, const float* b) { return (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2]; }

void orc_camera_update_resolution(orc_camera* c, uint32_t width, uint32_t height) /* Camera.cpp:13-23 */
{
    const float theta = 60 * 3.14f / 180;                                /* :15 (3.14, quirk Q21) */
    const float aspect = width / (float)height;                          /* :16 */
    c->halfHeight = tanf(theta / 2.f);                                   /* :18 */
    c->halfWidth = aspect * c->halfHeight;                               /* :19 */
    c->cb.pixelSize[0] = 1.f / width; c->cb.pixelSize[1] = 1.f / height; /* :21 */
    c->cb.sampleCounter = -1;                                            /* :22 */
}

void orc_camera_init(orc_camera* c, uint32_t width, uint32_t height) /* Camera.hpp:8-22,35-43 defaults */
{
    memset(c, 0, sizeof(*c));
    c->cb.pos[0] = 1.f; c->cb.pos[1] = 3.f; c->cb.pos[2] = 8.f;
    c->cb.envColor[0] = 0.0f; c->cb.envColor[1] = 0.0001f; c->cb.envColor[2] = 0.0001f;
    c->cb.sampleCounter = -1; c->cb.lightCount = 2; c->cb.sampleLights = 0;
    c->front[2] = 1.f; c->up[1] = 1.f;
    c->pitch = 0.f; c->yaw = 270.f;
    c->randState = 1;
    orc_camera_update_resolution(c, width, height);
}

void orc_camera_set_pose(orc_camera* c, float x, float y, float z, float pitch, float yaw) /* Scene.cpp:95-97 */
{
    c->cb.pos[0] = x; c->cb.pos[1] = y; c->cb.pos[2] = z; c->cb.pos[3] = 0.f;
    c->pitch = pitch; c->yaw = yaw;
}

void orc_camera_update(orc_camera* c) /* Camera.cpp:25-88 with no mouse / keyboard input */
{
    if (c->pitch > 89.0f) c->pitch = 89.0f;                              /* :32-33 */
    if (c->pitch < -89.0f) c->pitch = -89.0f;
    const float toRad = 3.141592654f / 180.0f;                           /* XMConvertToRadians */
    float ry = c->yaw * toRad, rp = c->pitch * toRad;
    c->front[0] = cosf(ry) * cosf(rp);                                   /* :35-39 */
    c->front[1] = sinf(rp);
    c->front[2] = sinf(ry) * cosf(rp);
    n3(c->front);                                                        /* :41 */
    const float yAxis[3] = { 0, 1, 0 };
    c3(yAxis, c->front, c->left); n3(c->left);                           /* :42 */
    c3(c->front, c->left, c->up); n3(c->up);                             /* :43 */

    /* :61 view = transpose(XMMatrixLookAtRH(pos, front + pos, up)) => rows (R0,D0),(R1,D1),(R2,D2) */
    float focus[3] = { c->front[0] + c->cb.pos[0], c->front[1] + c->cb.pos[1], c->front[2] + c->cb.pos[2] };
    float negEyeDir[3] = { c->cb.pos[0] - focus[0], c->cb.pos[1] - focus[1], c->cb.pos[2] - focus[2] };
    float R2[3] = { negEyeDir[0], negEyeDir[1], negEyeDir[2] }; n3(R2);
    float R0[3]; c3(c->up, R2, R0); n3(R0);
    float R1[3]; c3(R2, R0, R1);
    float negEye[3] = { -c->cb.pos[0], -c->cb.pos[1], -c->cb.pos[2] };
    float D0 = d3(R0, negEye), D1 = d3(R1, negEye), D2 = d3(R2, negEye);
    const float* left = R0; const float* up = R1; const float* w = R2;  /* :63-65 */
    float l4[4] = { left[0], left[1], left[2], D0 }, u4[4] = { up[0], up[1], up[2], D1 }, w4[4] = { w[0], w[1], w[2], D2 };
    for (int i = 0; i < 4; i++) {
        c->cb.ulc[i] = -c->halfWidth * l4[i] + c->halfHeight * u4[i] - w4[i]; /* :67 */
        c->cb.horizontal[i] = 2 * c->halfWidth * l4[i];                  /* :68 */
        c->cb.vertical[i] = 2 * c->halfHeight * u4[i];                   /* :69 */
    }
    c->cb.sampleCounter++;                                               /* :70 */
    if (c->moveHysteresis && c->cb.sampleCounter > 4) {                  /* :79-83 (no input => first branch never taken) */
        c->cb.sampleCounter = 0; c->moveHysteresis = 0;
    }
    float a = orc_msvc_rand(&c->randState) / (float)32767;               /* :85-87 */
    float b = orc_msvc_rand(&c->randState) / (float)32767;
    c->cb.randomSeed[0] = a; c->cb.randomSeed[1] = b;
}

void orc_debug_sample(const orc_scene* sc, int which, float u, float v, int layer, float* out4) { sample_bilinear(sc, which, u, v, layer, out4); }

/* ------------------------------------------------------------------ detmath probe */
void orc_detmath_eval(int fn, const float* x, const float* y, float* out, uint32_t n)
{
    for (uint32_t i = 0; i < n; i++) {
        switch (fn) {
        case 0: out[i] = o_sin(x[i]); break;
        case 1: out[i] = o_cos(x[i]); break;
        case 2: out[i] = o_log2(x[i]); break;
        case 3: out[i] = o_exp2(x[i]); break;
        case 4: out[i] = o_pow(x[i], y[i]); break;
        case 5: out[i] = o_frac(x[i]); break;
        case 6: { rng_t g; orc_camera_buffer cb; cb.randomSeed[0] = y[i]; cb.randomSeed[1] = y[i] * 0.5f; rng_seed(&g, &cb, (uint32_t)x[i]); rng_next(&g); out[i] = rng_next(&g); } break;
        default: out[i] = 0.0f;
        }
    }
}
