// Wavefront path-tracing kernels for gfx950 (MI355X).  Hand-written HIP; wave64 throughout.
//
// One wavefront iteration = the six dispatches of the reference's Renderer::draw()
// (Source/Renderer.cpp:195-211), regrouped for CDNA4:
//
//   k_logic      logic.hlsl:200-302 per slot: terminate / accumulate decision, material fetch, NEE set-up.
//                Writes a 1-byte class per slot and per-block class counts instead of pushing to queues
//                with per-warp atomics (the reference's NvBallot + InterlockedAdd idiom, logic.hlsl:36-44,
//                263-285): queue positions are then RANKS, independent of scheduling order.
//                The block counts are also summed per group of 64 blocks (integer atomics, order-free).
//   k_material   first its own queue offsets -- totals of the earlier groups + counts of the earlier blocks of its group: the result
//                of an exclusive scan without a scan launch; block 0 hands the seven counters over -- then the rank of every slot
//                inside its class by wave64 ballot + mbcnt + block offset, then, fused by
//                class: framebuffer accumulation (logic.hlsl:49-73) + newPath.hlsl:14-61, materialUE4.hlsl:118-192,
//                materialGlass.hlsl:48-85.  The RNG of those stages is seeded by the queue index
//                (newPath.hlsl:27, materialUE4.hlsl:131, materialGlass.hlsl:61), so ranks must be the
//                canonical ones (ascending slot order) -- which a scan gives and atomics do not.
//   k_extend     extensionRayCast.hlsl:197-234, closest hit + light spheres, per-wave LDS traversal stacks.
//   k_shadow     shadowRayCast.hlsl:139-169, any hit; thread 0 performs the counter hand-over (:144-148).
//
// Arithmetic: every float expression is evaluated in the reference's written order with IEEE binary32
// operations and no FMA contraction; sin/cos/exp2/log2 come from detmath.hpp.  The CPU oracle (oracle/) states
// the same sequence independently; tests compare the two bit for bit.
#include "pt_device.hpp"
#include "detmath.hpp"
#include "pt_kernel_util.hpp"

namespace gmupt {

// ------------------------------------------------------------------------------------------------ bsdf.h
__device__ __forceinline__ float schlickFresnel(float r0, float theta) // bsdf.h:1-6
{
    float m = hsaturate(1.0f - theta);
    float m2 = m * m;
    return r0 - (1.0f - r0) * m2 * m2 * m;
}
__device__ __forceinline__ float GGXTrowbridgeReitz(float XdotY, float alpha) // bsdf.h:8-13
{
    float a2 = alpha * alpha;
    float x = XdotY * XdotY * (a2 - 1.0f) + 1.0f;
    return a2 / (kPi * x * x);
}
__device__ __forceinline__ float smithSchlickGGX(float XdotY, float alpha) // bsdf.h:15-20
{
    float a1 = alpha + 1.0f;
    float k = (a1 * a1) / 8.0f;
    return XdotY / (XdotY * (1.0f - k) + k);
}
__device__ __forceinline__ float lightFalloff(float distance, float radius) // bsdf.h:22-26
{
    float q = distance / radius;
    float q2 = q * q;
    float n = hsaturate(1.0f - q2 * q2);
    return (n * n) / (distance * distance + 1.0f);
}
__device__ __forceinline__ float powerHeuristic(float rayPdf, float lightPdf) // bsdf.h:28-32
{
    float t = rayPdf * rayPdf;
    return t / (lightPdf * lightPdf + t);
}

// ------------------------------------------------------------------------------------------------ textures
// SampleLevel(linear, wrap) on an RGBA8 UNORM array (logic.hlsl:100,104,111; sampler Scene.cpp:180-192): texel centres at
// u*size - 0.5, fp32 weights, lerp(a,b,t) = a + t*(b-a), horizontal pairs first, c/255 decode.  Stated arithmetic.
__device__ __forceinline__ float4 sample_bilinear(const SceneView& sc, int which, float u, float v, int layer)
{
    float4 out = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    const uint8_t* tex = sc.tex[which];
    const int n = (int)sc.texSize[which];
    if (!tex || n <= 0 || sc.texLayers[which] == 0) return out;
    if (layer < 0) layer = 0;
    if (layer >= (int)sc.texLayers[which]) layer = (int)sc.texLayers[which] - 1;
    float x = u * (float)n - 0.5f, y = v * (float)n - 0.5f;
    if (!(dabs(x) < 1.0e9f)) x = 0.0f;
    if (!(dabs(y) < 1.0e9f)) y = 0.0f;
    const float x0 = dfloor(x), y0 = dfloor(y);
    const float fx = x - x0, fy = y - y0;
    int ix0 = (int)x0 % n; if (ix0 < 0) ix0 += n;
    int iy0 = (int)y0 % n; if (iy0 < 0) iy0 += n;
    const int ix1 = (ix0 + 1 == n) ? 0 : ix0 + 1, iy1 = (iy0 + 1 == n) ? 0 : iy0 + 1;
    const uchar4* base = reinterpret_cast<const uchar4*>(tex) + (size_t)layer * n * n;
    const uchar4 c00 = base[(size_t)iy0 * n + ix0], c10 = base[(size_t)iy0 * n + ix1], c01 = base[(size_t)iy1 * n + ix0], c11 = base[(size_t)iy1 * n + ix1];
#define GM_BILERP(ch) { const float a = (float)c00.ch / 255.0f, b = (float)c10.ch / 255.0f, c = (float)c01.ch / 255.0f, d = (float)c11.ch / 255.0f; \
                        const float r0 = a + fx * (b - a), r1 = c + fx * (d - c); out.ch = r0 + fy * (r1 - r0); }
    GM_BILERP(x) GM_BILERP(y) GM_BILERP(z) GM_BILERP(w)
#undef GM_BILERP
    return out;
}

// ------------------------------------------------------------------------------------------------ block class counts
// every thread of the block calls this once; writes blockCounts[k * nBlocks + blockIdx.x] for k = 0..3
// (k = 3: UE4 slots whose light lies in the upper hemisphere, i.e. the entries of the shadow queue)
__device__ __forceinline__ void publish_block_counts(const RenderParams& p, int c)
{
    __shared__ uint32_t s_cnt[kNumCounts][kBlock / 64];
    const uint32_t wave = threadIdx.x >> 6;
    unsigned long long b0 = __ballot((c & CLS_MASK) == CLS_UE4), b1 = __ballot(c == CLS_GLASS), b2 = __ballot(c == CLS_ENDED), b3 = __ballot(c == (CLS_UE4 | CLS_SHADOW_BIT));
    if ((threadIdx.x & 63) == 0) { s_cnt[0][wave] = __popcll(b0); s_cnt[1][wave] = __popcll(b1); s_cnt[2][wave] = __popcll(b2); s_cnt[3][wave] = __popcll(b3); }
    __syncthreads();
    if (threadIdx.x < kNumCounts) {
        uint32_t s = 0;
        for (int w = 0; w < kBlock / 64; w++) s += s_cnt[threadIdx.x][w];
        p.blockCounts[threadIdx.x * p.nBlocks + blockIdx.x] = s;
        // second level: totals per group of kScanGroup blocks (integer sums: the order of the atomics does not matter)
        if (s) atomicAdd(&p.groupTotals[((size_t)p.groupParity * kNumCounts + threadIdx.x) * p.nGroups + blockIdx.x / kScanGroup], s);
    }
}

// ------------------------------------------------------------------------------------------------ k_clear (logic.hlsl:165-190)
__global__ __launch_bounds__(kBlock) void k_clear(RenderParams p)
{
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    // zero the sample count of every pixel, keep the colour (logic.hlsl:180-181)
    uint32_t w = p.tileEnabled ? p.fbW : cam_width(p.cam), h = p.tileEnabled ? p.fbH : cam_height(p.cam);
    uint32_t npix = w * h; if (npix > p.fbW * p.fbH) npix = p.fbW * p.fbH;
    for (uint32_t k = i; k < npix; k += gridDim.x * kBlock) { p.fb[k].w = __builtin_bit_cast(float, 0u); p.listHead[k] = kListEnd; }
    // newPath[i] = i for the whole pool (logic.hlsl:187-188): every live slot becomes an ended path of rank i
    int c = CLS_NONE;
    if (i < p.L) { c = CLS_ENDED; p.cls[i] = CLS_ENDED; }
    publish_block_counts(p, c);
}

// ------------------------------------------------------------------------------------------------ k_logic (logic.hlsl:200-302)
__global__ __launch_bounds__(kBlock) void k_logic(RenderParams p)
{
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    int c = CLS_NONE;
    if (i < p.L) {
        c = p.cls[i];
        if (c != CLS_RETIRED) {
            Rng g; g.seed(i, p.cam.randomSeed[0], p.cam.randomSeed[1]);     // :216
            bool pathEliminated = false;                                   // :217
            f3 throughput = ld3(p, F_THR_R, i);                             // :219
            f3 radiance = ld3(p, F_RAD_R, i);                               // :220
            const uint32_t isEmitter = ldu(p, F_IS_EMITTER, i);
            uint32_t pl = 0;
            if (isEmitter > 0) {                                            // :222-226, sampleLight :192-197
                uint32_t li = isEmitter - 1; if (li >= GMUPT_MAX_LIGHTS) li = GMUPT_MAX_LIGHTS - 1;
                const gmupt_light L = p.scene.lights[li];
                float emax = hmax(L.emission[0], hmax(L.emission[1], L.emission[2]));
                f3 e = mk3(L.emission[0] / emax, L.emission[1] / emax, L.emission[2] / emax);
                radiance = radiance + e * throughput;
                pathEliminated = true;
            } else {
                if (!ldu(p, F_IN_SHADOW, i)) radiance = radiance + ld3(p, F_DL_R, i) * throughput; // :230-231
                throughput = throughput * ld3(p, F_LTHR_R, i);              // :234
                if (throughput.x <= 0.0f && throughput.y <= 0.0f && throughput.z <= 0.0f) pathEliminated = true; // :237
                if (ldf(p, F_HIT_DIST, i) == kFltMax) {                     // :241-245
                    radiance = radiance + throughput * mk3(p.cam.envColor[0], p.cam.envColor[1], p.cam.envColor[2]);
                    pathEliminated = true;
                }
                pl = ldu(p, F_PATH_LEN, i);
                if (pl > 200) {                                             // :248-255
                    float pr = hmax(throughput.x, hmax(throughput.y, throughput.z));
                    if (g.next() > pr * 0.004f) pathEliminated = true;
                    throughput = throughput * (1.0f / pr);
                }
                if (p.maxDepth && pl >= p.maxDepth) pathEliminated = true;  // extension
            }

            if (pathEliminated) {
                // endPath :49-53: per-sample tonemap; the running-mean update is applied in canonical order by k_material
                c = CLS_ENDED;
                f3 r = mk3(hsaturate(radiance.x), hsaturate(radiance.y), hsaturate(radiance.z));
                r = r / (r + mk3(1.0f, 1.0f, 1.0f));
                const float gm = 1.0f / 2.2f;
                r = mk3(dpow(r.x, gm), dpow(r.y, gm), dpow(r.z, gm));
                p.sample[i] = r.x; p.sample[(size_t)p.P + i] = r.y; p.sample[(size_t)2 * p.P + i] = r.z;
                const uint32_t pix = pixel_index(p, ldu(p, F_SCR_X, i), ldu(p, F_SCR_Y, i));
                uint32_t prev = kListEnd;
                if (pix != kListEnd) prev = atomicExch(&p.listHead[pix], i);
                p.listNext[i] = prev;
            } else {
                // setMaterialHitProperties :79-133
                const uint32_t t0 = ldu(p, F_TRI_0, i), t1 = ldu(p, F_TRI_1, i), t2 = ldu(p, F_TRI_2, i), tm = ldu(p, F_TRI_MAT, i);
                const uint32_t i0 = (uint32_t)(float)t0, i1 = (uint32_t)(float)t1, i2 = (uint32_t)(float)t2; // :82 float round trip
                const f3 bary = ld3(p, F_BARY_X, i);
                const gmupt_tri_props* tp = p.scene.props;
                f3 n0 = mk3(tp[i0].normal[0], tp[i0].normal[1], tp[i0].normal[2]);
                f3 n1 = mk3(tp[i1].normal[0], tp[i1].normal[1], tp[i1].normal[2]);
                f3 n2 = mk3(tp[i2].normal[0], tp[i2].normal[1], tp[i2].normal[2]);
                f3 normal = (n0 * bary.x + n1 * bary.y) + n2 * bary.z;      // :94
                gmupt_material m = p.scene.materials[tm < GMUPT_MAX_LIGHTS ? tm : 0u]; // :96
                if (m.textureIndices[0] >= 0 || m.textureIndices[1] >= 0 || m.textureIndices[2] >= 0) {
                    const float tu = (tp[i0].uv[0] * bary.x + tp[i1].uv[0] * bary.y) + tp[i2].uv[0] * bary.z; // :93
                    const float tv = (tp[i0].uv[1] * bary.x + tp[i1].uv[1] * bary.y) + tp[i2].uv[1] * bary.z;
                    if (m.textureIndices[0] >= 0) {                        // :99-100
                        const float4 t = sample_bilinear(p.scene, 0, tu, tv, m.textureIndices[0]);
                        m.color[0] = t.x; m.color[1] = t.y; m.color[2] = t.z; m.color[3] = t.w;
                    }
                    if (m.textureIndices[1] >= 0) {                        // :102-107 metallic = .x, roughness = .y
                        const float4 t = sample_bilinear(p.scene, 1, tu, tv, m.textureIndices[1]);
                        m.metallic = t.x; m.roughness = t.y;
                    }
                    if (m.textureIndices[2] >= 0) {                        // :109-124 normal map
                        const float4 t = sample_bilinear(p.scene, 2, tu, tv, m.textureIndices[2]);
                        const f3 data = mk3(t.x * 2.0f - 1.0f, t.y * 2.0f - 1.0f, t.z * 2.0f - 1.0f);
                        const f3 rayDirection = ld3(p, F_RAY_DX, i);
                        const f3 ortNormal = dot3(normal, rayDirection) <= 0.0f ? normal : normal * -1.0f;
                        const f3 up = dabs(ortNormal.z) < 0.999f ? mk3(0.0f, 0.0f, 1.0f) : mk3(1.0f, 0.0f, 0.0f);
                        const f3 tangent = normalize3(cross3(up, ortNormal));
                        const f3 bitangent = cross3(ortNormal, tangent);
                        normal = (tangent * data.x + bitangent * data.y) + ortNormal * data.z; // :123 (not renormalised)
                    }
                }
                const float rough = hmax(0.014f, m.roughness);             // :126
                st3(p, F_MAT_R, i, mk3(m.color[0], m.color[1], m.color[2])); // :128
                stf(p, F_MAT_METALLIC, i, m.metallic); stf(p, F_MAT_ROUGHNESS, i, rough); // :129
                st3(p, F_NRM_X, i, normal);                                 // :130
                c = (m.materialType == GMUPT_MATERIAL_UE4) ? CLS_UE4 : CLS_GLASS; // :262-285 (types other than 0/1 are not produced by Scene)

                // createShadowRay :135-163
                uint32_t lightIndex = (uint32_t)(g.next() * (float)p.cam.lightCount);
                float z = 1.0f - 2.0f * g.next();
                float rr = dsqrt(hmax(0.0f, 1.0f - z * z));
                float phi = 2.0f * kPi * g.next();
                float x = rr * dcos(phi);
                float y = rr * dsin(phi);
                const gmupt_light L = p.scene.lights[lightIndex < GMUPT_MAX_LIGHTS ? lightIndex : GMUPT_MAX_LIGHTS - 1];
                f3 lightPosition = mk3(L.position[0], L.position[1], L.position[2]) + mk3(x, y, z) * L.radius;
                f3 surfacePos = ld3(p, F_SP_X, i) + normal * kEpsilonOffset;
                f3 lightDir = lightPosition - surfacePos;
                float distance = length3(lightDir);
                lightDir = normalize3(lightDir);
                stu(p, F_LIGHT_IDX, i, lightIndex);
                st3(p, F_SH_OX, i, surfacePos);
                st3(p, F_SH_DX, i, lightDir);
                stf(p, F_LIGHT_DIST, i, distance - kEpsilonOffset);
                // materialUE4.hlsl:167: a UE4 slot pushes a shadow ray iff the light direction lies in the normal's hemisphere;
                // both operands are final here, so the shadow queue can be ranked by the same scan as the material queues
                if (c == CLS_UE4 && dot3(lightDir, normal) > 0.0f) c |= CLS_SHADOW_BIT;

                st3(p, F_RAD_R, i, radiance);                               // :295
                st3(p, F_THR_R, i, throughput);                             // :296
                stu(p, F_PATH_LEN, i, pl + 1u);                              // :293,297
                stu(p, F_IN_SHADOW, i, 1u);                                 // :298
            }
            p.cls[i] = (uint8_t)c;
        }
    }
    publish_block_counts(p, c);
}

// ------------------------------------------------------------------------------------------------ material stages
struct Ue4State { f3 rayDir; f3 baseColor; float metallic, roughness; f3 normal; };

__device__ __forceinline__ f3 ue4Sample(const Ue4State& st, Rng& g) // materialUE4.hlsl:24-68
{
    const f3 N = st.normal;
    const f3 V = neg3(st.rayDir);
    const float r0 = g.next(), r1 = g.next();                               // :29
    const float diffuseRatio = 1.0f - st.metallic;                          // :31
    const f3 up = dabs(N.z) < 0.999f ? mk3(0.0f, 0.0f, 1.0f) : mk3(1.0f, 0.0f, 0.0f); // :35
    const f3 tangent = normalize3(cross3(up, N));                           // :36
    const f3 bitangent = cross3(N, tangent);                                // :37
    f3 direction;
    if (g.next() < diffuseRatio) {                                          // :40
        float x = dsqrt(r0);
        float phi = 2.0f * kPi * r1;
        float dx = x * dcos(phi);
        float dy = x * dsin(phi);
        float dz = dsqrt(hmax(0.0f, 1.0f - dx * dx - dy * dy));
        direction = (tangent * dx + bitangent * dy) + N * dz;               // :48
    } else {
        float a = st.roughness * st.roughness;                              // :52
        float phi = 2.0f * kPi * r0;
        float cosTheta = dsqrt((1.0f - r1) / (1.0f + (a * a - 1.0f) * r1)); // :55
        float sinTheta = dsqrt(1.0f - cosTheta * cosTheta);
        float hx = sinTheta * dcos(phi);
        float hy = sinTheta * dsin(phi);
        float hz = cosTheta;
        direction = (tangent * hx + bitangent * hy) + N * hz;               // :63
        direction = direction * (2.0f * dot3(V, direction)) - V;            // :64
    }
    return direction;
}

__device__ __forceinline__ float ue4Pdf(const Ue4State& st, f3 direction) // materialUE4.hlsl:70-90
{
    const f3 N = st.normal, V = neg3(st.rayDir), L = direction;
    const float diffuseRatio = 1.0f - st.metallic;
    const float specularRatio = 1.0f - diffuseRatio;
    const f3 H = normalize3(L + V);
    const float NdotH = dabs(dot3(N, H));
    const float pdfGGXTR = GGXTrowbridgeReitz(NdotH, st.roughness * st.roughness) * NdotH;
    const float pdfSpec = pdfGGXTR / (4.0f * dabs(dot3(V, H)));
    const float pdfDiff = dabs(dot3(L, N)) * (1.0f / kPi);
    return diffuseRatio * pdfDiff + specularRatio * pdfSpec;
}

__device__ __forceinline__ f3 ue4Evaluate(const Ue4State& st, f3 direction) // materialUE4.hlsl:92-115
{
    const f3 N = st.normal, V = neg3(st.rayDir), L = direction;
    const float NdotL = dot3(N, L), NdotV = dot3(N, V);
    if (NdotL <= 0.0f || NdotV <= 0.0f) return mk3(0.0f, 0.0f, 0.0f);
    const f3 H = normalize3(L + V);
    const float NdotH = dot3(N, H), LdotH = dot3(L, H);
    const float D = GGXTrowbridgeReitz(NdotH, st.roughness * st.roughness);
    const float G = smithSchlickGGX(NdotL, st.roughness) * smithSchlickGGX(NdotV, st.roughness);
    const f3 sc = mk3(0.037f + st.metallic * (st.baseColor.x - 0.037f),
                      0.037f + st.metallic * (st.baseColor.y - 0.037f),
                      0.037f + st.metallic * (st.baseColor.z - 0.037f));    // :110 lerp
    const float w = 1.0f - LdotH;
    const float w2 = w * w;
    const float fc = w2 * w2 * w;                                            // :111 pow(1 - LdotH, 5)
    const f3 F = mk3((1.0f - fc) * sc.x + fc, (1.0f - fc) * sc.y + fc, (1.0f - fc) * sc.z + fc);
    const float den = 4.0f * NdotL * NdotV;
    const float om = 1.0f - st.metallic;
    return mk3((st.baseColor.x / kPi) * om + (D * F.x * G) / den,
               (st.baseColor.y / kPi) * om + (D * F.y * G) / den,
               (st.baseColor.z / kPi) * om + (D * F.z * G) / den);           // :114
}

__device__ __forceinline__ void stage_ue4(const RenderParams& p, uint32_t queueIndex, uint32_t index, uint32_t shadowRank, uint32_t extOffset) // materialUE4.hlsl:118-192
{
    Rng g; g.seed(queueIndex, p.cam.randomSeed[0], p.cam.randomSeed[1]);   // :131
    Ue4State st;
    st.rayDir = ld3(p, F_RAY_DX, index);
    st.normal = ld3(p, F_NRM_X, index);
    st.baseColor = ld3(p, F_MAT_R, index);
    st.metallic = ldf(p, F_MAT_METALLIC, index);
    st.roughness = ldf(p, F_MAT_ROUGHNESS, index);

    const f3 bsdfDir = ue4Sample(st, g);                                     // :148
    const float pdf = ue4Pdf(st, bsdfDir);                                   // :149
    f3 throughput = mk3(0.0f, 0.0f, 0.0f);
    if (pdf > 0.0f) {                                                        // :151-152
        const f3 e = ue4Evaluate(st, bsdfDir);
        const float an = dabs(dot3(st.normal, bsdfDir));
        throughput = mk3(e.x * an / pdf, e.y * an / pdf, e.z * an / pdf);
    }
    st3(p, F_LTHR_R, index, throughput);                                     // :154
    const f3 surfacePoint = ld3(p, F_SP_X, index);                           // :157
    st3(p, F_RAY_OX, index, surfacePoint + bsdfDir * kEpsilonOffset);        // :158,160
    st3(p, F_RAY_DX, index, bsdfDir);                                        // :161
    p.queues[(size_t)Q_EXT_RAY * p.P + extOffset + queueIndex] = index;       // :162 (extOffset = QC[4])

    const f3 lightDir = ld3(p, F_SH_DX, index);                              // :165
    if (dot3(lightDir, st.normal) > 0.0f) {                                  // :167,178
        const uint32_t lightIndex = ldu(p, F_LIGHT_IDX, index);
        const float distance = ldf(p, F_LIGHT_DIST, index);
        const gmupt_light L = p.scene.lights[lightIndex < GMUPT_MAX_LIGHTS ? lightIndex : GMUPT_MAX_LIGHTS - 1];
        const float lightPdf = distance * distance / (4.0f * kPi * L.radius * L.radius); // :184
        const float bsdfPdf = ue4Pdf(st, lightDir);                          // :185
        const float ph = powerHeuristic(lightPdf, bsdfPdf);
        const f3 e = ue4Evaluate(st, lightDir);
        const float lc = (float)p.cam.lightCount;
        const float fo = lightFalloff(distance, L.falloff);
        st3(p, F_DL_R, index, mk3(ph * e.x * L.emission[0] * lc * fo, ph * e.y * L.emission[1] * lc * fo, ph * e.z * L.emission[2] * lc * fo)); // :187-188
        // :168-176,189: shadow queue slot = rank among the UE4 slots that push one (canonical order, no atomics)
        const uint32_t pos = shadowRank;
        p.queues[(size_t)Q_SHADOW_RAY * p.P + pos] = index;                  // :189
    }
}

__device__ __forceinline__ void stage_glass(const RenderParams& p, uint32_t queueIndex, uint32_t index, uint32_t extOffset) // materialGlass.hlsl:23-85
{
    Rng g; g.seed(queueIndex, p.cam.randomSeed[0], p.cam.randomSeed[1]);   // :61
    const f3 rayDir = ld3(p, F_RAY_DX, index);
    const f3 stNormal = ld3(p, F_NRM_X, index);
    const f3 baseColor = ld3(p, F_MAT_R, index);

    const f3 normal = dot3(stNormal, rayDir) <= 0.0f ? stNormal : stNormal * -1.0f; // :25
    const float n1 = 1.0f, n2 = 1.458f;
    float r0 = (n1 - n2) / (n1 + n2);
    r0 = r0 * r0;
    const float theta = dot3(neg3(rayDir), normal);                          // :34
    const float probability = schlickFresnel(r0, theta);                     // :36
    const float eta = dot3(stNormal, normal) > 0.0f ? (n1 / n2) : (n2 / n1); // :38
    // refract(i, n, eta) (HLSL intrinsic)
    const float dn = dot3(normal, rayDir);
    const float k = 1.0f - eta * eta * (1.0f - dn * dn);
    f3 refr = mk3(0.0f, 0.0f, 0.0f);
    if (!(k < 0.0f)) refr = rayDir * eta - normal * (eta * dn + dsqrt(k));
    const f3 transDirection = normalize3(refr);                              // :39
    const float cos2t = 1.0f - eta * eta * (1.0f - theta * theta);           // :40
    const float rnd = g.next();                                              // :42 (|| does not short-circuit in HLSL)
    f3 bsdfDir = transDirection;
    if (cos2t < 0.0f || rnd < probability)
        bsdfDir = normalize3(rayDir - normal * (2.0f * dot3(normal, rayDir))); // :43 reflect

    st3(p, F_LTHR_R, index, baseColor);                                      // :75
    const f3 surfacePoint = ld3(p, F_SP_X, index);
    st3(p, F_RAY_OX, index, surfacePoint + bsdfDir * kEpsilonOffset);        // :79,81
    st3(p, F_RAY_DX, index, bsdfDir);                                        // :82
    p.queues[(size_t)Q_EXT_RAY * p.P + extOffset + queueIndex] = index;       // :83 (extOffset = QC[5])
}

// running-mean update of one pixel for all paths that ended on it this iteration, in ascending slot order
// (canonical schedule of logic.hlsl:57-73; the reference's unsynchronised read-modify-write loses samples instead)
__device__ __forceinline__ void accumulate_pixel(const RenderParams& p, uint32_t pix, uint32_t head)
{
    float4 px = p.fb[pix];
    uint32_t n = __builtin_bit_cast(uint32_t, px.w);
    long long last = -1;
    for (;;) {
        uint32_t best = kListEnd;
        for (uint32_t s = head; s != kListEnd; s = p.listNext[s])
            if ((long long)s > last && s < best) best = s;
        if (best == kListEnd) break;
        const float r = p.sample[best], gch = p.sample[(size_t)p.P + best], b = p.sample[(size_t)2 * p.P + best];
        const float n0 = (float)n;
        n++;
        const float n1 = (float)n;
        px.x = (px.x * n0 + r) / n1;                                         // logic.hlsl:73
        px.y = (px.y * n0 + gch) / n1;
        px.z = (px.z * n0 + b) / n1;
        last = (long long)best;
    }
    px.w = __builtin_bit_cast(float, n);
    p.fb[pix] = px;
    p.listHead[pix] = kListEnd;
}

__device__ __forceinline__ void stage_new_path(const RenderParams& p, uint32_t queueIndex, uint32_t index, int clearFrame) // newPath.hlsl:14-61
{
    if (!clearFrame) {
        const uint32_t pix = pixel_index(p, ldu(p, F_SCR_X, index), ldu(p, F_SCR_Y, index));
        if (pix != kListEnd && p.listHead[pix] == index) accumulate_pixel(p, pix, index);
    }
    const uint32_t lastPath = p.qc[QC_LASTPATHCNT];                          // :19
    p.queues[(size_t)Q_NEWPATH * p.P + queueIndex] = index;                  // logic.hlsl:75
    if (p.budget && (uint32_t)(lastPath + queueIndex) >= p.budget) {         // extension: retire instead of regenerating
        p.cls[index] = CLS_RETIRED;
        p.queues[(size_t)Q_EXT_RAY * p.P + queueIndex] = kQueueHole;
        return;
    }
    Rng g; g.seed(queueIndex, p.cam.randomSeed[0], p.cam.randomSeed[1]);   // :27
    const uint32_t w = p.tileEnabled ? p.fbW : cam_width(p.cam), h = p.tileEnabled ? p.fbH : cam_height(p.cam);
    const uint32_t newIndex = (lastPath + queueIndex) % (w * h);             // :33
    const uint32_t cx = (p.tileEnabled ? p.tileX0 : 0u) + newIndex % w, cy = (p.tileEnabled ? p.tileY0 : 0u) + newIndex / w; // :34
    const float jx = g.next() * 2.0f - 1.0f;                                 // :36
    const float jy = g.next() * 2.0f - 1.0f;
    const float u = ((float)cx + jx) * p.cam.pixelSize[0];                   // :37
    const float v = ((float)cy + jy) * p.cam.pixelSize[1];
    const f3 ulc = mk3(p.cam.upperLeftCorner[0], p.cam.upperLeftCorner[1], p.cam.upperLeftCorner[2]);
    const f3 hor = mk3(p.cam.horizontal[0], p.cam.horizontal[1], p.cam.horizontal[2]);
    const f3 ver = mk3(p.cam.vertical[0], p.cam.vertical[1], p.cam.vertical[2]);
    const f3 dir = normalize3((ulc + hor * u) - ver * v);                    // :39
    st3(p, F_RAY_OX, index, mk3(p.cam.position[0], p.cam.position[1], p.cam.position[2])); // :41
    st3(p, F_RAY_DX, index, dir);                                            // :42
    stu(p, F_SCR_X, index, cx); stu(p, F_SCR_Y, index, cy);                  // :43
    st3(p, F_RAD_R, index, mk3(0.0f, 0.0f, 0.0f));                           // :44
    st3(p, F_THR_R, index, mk3(1.0f, 1.0f, 1.0f));                           // :45
    st3(p, F_LTHR_R, index, mk3(1.0f, 1.0f, 1.0f));                          // :46
    stu(p, F_PATH_LEN, index, 0u);                                           // :47
    stu(p, F_IN_SHADOW, index, 1u);                                          // :48
    p.queues[(size_t)Q_EXT_RAY * p.P + queueIndex] = index;                  // :51
}

#ifndef GMUPT_MATERIAL_REGROUP
#define GMUPT_MATERIAL_REGROUP 1
#endif
__global__ __launch_bounds__(kBlock) void k_material(RenderParams p, int clearFrame)
{
    __shared__ uint32_t s_cnt[kNumCounts][kBlock / 64];
    __shared__ uint32_t s_pre[kNumCounts], s_tot[kNumCounts];
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    const uint32_t wave = threadIdx.x >> 6;
    const int cfull = (i < p.L) ? (int)p.cls[i] : (int)CLS_NONE;
    const int c = cfull & CLS_MASK;
    const bool shadow = cfull == (CLS_UE4 | CLS_SHADOW_BIT);
    const unsigned long long b0 = __ballot(c == CLS_UE4), b1 = __ballot(c == CLS_GLASS), b2 = __ballot(c == CLS_ENDED), b3 = __ballot(shadow);
    if ((threadIdx.x & 63) == 0) { s_cnt[0][wave] = __popcll(b0); s_cnt[1][wave] = __popcll(b1); s_cnt[2][wave] = __popcll(b2); s_cnt[3][wave] = __popcll(b3); }
    if (threadIdx.x < kNumCounts) { s_pre[threadIdx.x] = 0; s_tot[threadIdx.x] = 0; }
    __syncthreads();

    // ---- queue offsets of this block without a separate scan launch: (class counts of all earlier groups) + (of the earlier blocks of
    // this group), and the totals over all groups (the queue counters of the reference).  Every block reads at most nGroups + 63 values
    // per class; the sums are exact integers, so the ranks are those of a sequential scan.
    {
        const uint32_t* gt = p.groupTotals + (size_t)p.groupParity * kNumCounts * p.nGroups;
        const uint32_t myGroup = blockIdx.x / kScanGroup, groupStart = myGroup * kScanGroup;
        uint32_t pre[kNumCounts] = { 0, 0, 0, 0 }, tot[kNumCounts] = { 0, 0, 0, 0 };
        for (uint32_t g = threadIdx.x; g < p.nGroups; g += kBlock)
#pragma unroll
            for (int k = 0; k < kNumCounts; k++) { const uint32_t v = gt[(size_t)k * p.nGroups + g]; tot[k] += v; if (g < myGroup) pre[k] += v; }
        if (threadIdx.x < kScanGroup && groupStart + threadIdx.x < blockIdx.x)
#pragma unroll
            for (int k = 0; k < kNumCounts; k++) pre[k] += p.blockCounts[(size_t)k * p.nBlocks + groupStart + threadIdx.x];
#pragma unroll
        for (int k = 0; k < kNumCounts; k++) {
            uint32_t a = pre[k], t = tot[k];
            for (int off = 32; off > 0; off >>= 1) { a += __shfl_down(a, off); t += __shfl_down(t, off); }
            if ((threadIdx.x & 63) == 0) { if (a) atomicAdd(&s_pre[k], a); if (t) atomicAdd(&s_tot[k], t); }
        }
        __syncthreads();
    }
    const uint32_t nUE4 = s_tot[CLS_UE4], nGlass = s_tot[CLS_GLASS], nEnded = s_tot[CLS_ENDED], nShadow = s_tot[3];
    const uint32_t qc0 = clearFrame ? p.P : nEnded;                 // logic.hlsl:178 stores PATHCOUNT on a clear
    const uint32_t extUE4Offset = qc0, extGlassOffset = qc0 + nUE4; // newPath.hlsl:57-58

    if (blockIdx.x == 0) {
        // the other half of the group totals is the next iteration's: nobody reads or writes it during this launch
        uint32_t* other = p.groupTotals + (size_t)(p.groupParity ^ 1u) * kNumCounts * p.nGroups;
        for (uint32_t k = threadIdx.x; k < kNumCounts * p.nGroups; k += kBlock) other[k] = 0u;
        if (threadIdx.x == 0) {
            const uint32_t nNew = qc0 < p.L ? qc0 : p.L;           // newPath.hlsl:21-25 reaches at most the live slots
            const uint32_t lastPath = p.qc[QC_LASTPATHCNT];
            p.qc[QC_NEWPATH] = qc0;
            p.qc[QC_MATUE4] = nUE4;
            p.qc[QC_MATGLASS] = nGlass;
            p.qc[QC_EXT_UE4_OFFSET] = extUE4Offset;
            p.qc[QC_EXT_GLASS_OFFSET] = extGlassOffset;
            p.qc[QC_SHADOWRAY] = nShadow;                          // newPath.hlsl:59 resets it, materialUE4.hlsl:173 counts it up to this total
            p.qc[QC_EXT_COUNT] = nNew + nUE4 + nGlass;
            p.travCounters[0] = 0; p.travCounters[1] = 0;
            uint32_t gen = nNew;
            if (p.budget) { uint32_t remaining = p.budget > lastPath ? p.budget - lastPath : 0u; if (gen > remaining) gen = remaining; }
            DevStats* st = p.stats;
            if (clearFrame) st->activePaths = p.L;
            st->activePaths -= (nNew - gen);
            st->pathsGenerated += gen;
            if (!clearFrame) { st->pathsCompleted += nEnded; st->segments += (unsigned long long)nEnded + nUE4 + nGlass; }
        }
    }
#if GMUPT_MATERIAL_REGROUP
    // ---- the slots of the block regrouped by class before the stages run: thread j takes the j-th slot of the sequence (UE4 slots in slot order,
    // then glass, then ended).  The three stages are long and different (k_material is bound by VALU issue, not by memory): with the classes
    // mixed as the slots are, nearly every wave runs all three one after the other at 63 % of its lanes; regrouped, at most two waves of a block
    // hold more than one class.  Nothing a slot computes depends on the thread that computes it: rank and slot are handed over with it.
    __shared__ uint16_t s_item[kBlock], s_srank[kBlock];
    uint32_t nBlk[3];
#pragma unroll
    for (int k = 0; k < 3; k++) { uint32_t n = 0; for (uint32_t w = 0; w < kBlock / 64; w++) n += s_cnt[k][w]; nBlk[k] = n; }
    if (c <= CLS_ENDED) {
        // local rank = slots of the same class with a smaller index in this block: earlier waves + lower lanes
        const unsigned long long mine = (c == CLS_UE4) ? b0 : (c == CLS_GLASS) ? b1 : b2;
        uint32_t local = prefix_rank(mine);
        for (uint32_t w = 0; w < wave; w++) local += s_cnt[c][w];
        uint32_t slocal = 0;
        if (shadow) { slocal = prefix_rank(b3); for (uint32_t w = 0; w < wave; w++) slocal += s_cnt[3][w]; }
        const uint32_t j = (c == CLS_UE4 ? 0u : c == CLS_GLASS ? nBlk[CLS_UE4] : nBlk[CLS_UE4] + nBlk[CLS_GLASS]) + local;
        s_item[j] = (uint16_t)(threadIdx.x | ((uint32_t)c << 12) | (shadow ? 0x8000u : 0u));
        s_srank[j] = (uint16_t)slocal;
    }
    __syncthreads();
    if (threadIdx.x >= nBlk[0] + nBlk[1] + nBlk[2]) return;
    const uint32_t item = s_item[threadIdx.x];
    const int cj = (int)((item >> 12) & 7u);
    const uint32_t slot = blockIdx.x * kBlock + (item & 0xFFFu);
    // rank = slots of the same class with a smaller index: block offset + position inside the block's run of that class
    const uint32_t rank = s_pre[cj] + threadIdx.x - (cj == CLS_UE4 ? 0u : cj == CLS_GLASS ? nBlk[CLS_UE4] : nBlk[CLS_UE4] + nBlk[CLS_GLASS]);

    if (cj == CLS_ENDED) stage_new_path(p, rank, slot, clearFrame);
    else {
        p.queues[(size_t)(cj == CLS_UE4 ? Q_MAT_UE4 : Q_MAT_GLASS) * p.P + rank] = slot; // logic.hlsl:282-285
        if (cj == CLS_UE4) stage_ue4(p, rank, slot, (item & 0x8000u) ? s_pre[3] + s_srank[threadIdx.x] : 0u, extUE4Offset);
        else stage_glass(p, rank, slot, extGlassOffset);
    }
}
#else
    if (c > CLS_ENDED) return;
    // rank = slots of the same class with a smaller index: block offset + earlier waves + lower lanes
    const unsigned long long mine = (c == CLS_UE4) ? b0 : (c == CLS_GLASS) ? b1 : b2;
    uint32_t rank = s_pre[c] + prefix_rank(mine);
    for (uint32_t w = 0; w < wave; w++) rank += s_cnt[c][w];

    if (c == CLS_ENDED) stage_new_path(p, rank, i, clearFrame);
    else {
        p.queues[(size_t)(c == CLS_UE4 ? Q_MAT_UE4 : Q_MAT_GLASS) * p.P + rank] = i; // logic.hlsl:282-285
        if (c == CLS_UE4) {
            uint32_t srank = 0;
            if (shadow) {
                srank = s_pre[3] + prefix_rank(b3);
                for (uint32_t w = 0; w < wave; w++) srank += s_cnt[3][w];
            }
            stage_ue4(p, rank, i, srank, extUE4Offset);
        } else stage_glass(p, rank, i, extGlassOffset);
    }
}
#endif

// ------------------------------------------------------------------------------------------------ detmath probe
__global__ void k_detmath(int fn, const float* x, const float* y, float* out, uint32_t n)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float r = 0.0f;
    switch (fn) {
    case 0: r = dsin(x[i]); break;
    case 1: r = dcos(x[i]); break;
    case 2: r = dlog2(x[i]); break;
    case 3: r = dexp2(x[i]); break;
    case 4: r = dpow(x[i], y[i]); break;
    case 5: r = dfrac(x[i]); break;
    case 6: { Rng g; g.seed((uint32_t)x[i], y[i], y[i] * 0.5f); g.next(); r = g.next(); } break;
    default: break;
    }
    out[i] = r;
}

// ------------------------------------------------------------------------------------------------ host launchers
static inline uint32_t slot_blocks(const RenderParams& p) { return p.nBlocks; }

void launch_clear(const RenderParams& p, hipStream_t s) { hipLaunchKernelGGL(k_clear, dim3(slot_blocks(p)), dim3(kBlock), 0, s, p); }
void launch_logic(const RenderParams& p, hipStream_t s) { hipLaunchKernelGGL(k_logic, dim3(slot_blocks(p)), dim3(kBlock), 0, s, p); }
void launch_material(const RenderParams& p, int clearFrame, hipStream_t s) { hipLaunchKernelGGL(k_material, dim3(slot_blocks(p)), dim3(kBlock), 0, s, p, clearFrame); }
void launch_detmath(int fn, const float* x, const float* y, float* out, uint32_t n, hipStream_t s)
{
    hipLaunchKernelGGL(k_detmath, dim3((n + 255) / 256), dim3(256), 0, s, fn, x, y, out, n);
}
} // namespace gmupt
