// Wide ray cast for gfx950 (MI355X): both ray casts in one persistent launch over a 4-WIDE collapse of the binary SBVH, one 128-byte
// record -- one memory-side line -- per dependent step (GMUPT_TRAVERSAL=wide).
//
// Why the walk may be restructured at all.  The reference never prunes a box against the current hit: a child is visited iff its slab
// test returns > 0 (extensionRayCast.hlsl:79-94,132-159).  So the set of leaves a ray reaches is fixed -- every leaf whose box and all of
// whose ancestors' boxes are hit -- and so are the triangle tests.  A child's box lies inside its parent's box and the slab arithmetic
// is monotone in the plane coordinates (binary32 subtraction and multiplication by the same 1 / d are monotone; NaN-dropping min / max
// keep the child's interval inside the parent's), so "child hit" implies "parent hit" and testing only SOME of a leaf's ancestors
// reaches exactly the same leaves.  Two conditions, both checked on the host when the collapse is built (gmupt_capi.hip):
//   - every child box lies inside its parent's box (a tree that violates it is walked by the binary kernel);
//   - the one case where monotonicity fails: a ray with an infinite 1 / d component (d = 0 on an axis; the reference's 8-bit random
//     numbers make 0.3 % of the bounces off axis-aligned walls such rays) whose origin lies exactly in the plane of a box that is FLAT
//     on that axis.  0 x inf = NaN is dropped by min / max, so the flat box imposes no condition, while a non-flat parent that has
//     this plane as a FACE is missed (near = far = +-inf).  An inner node with a child that is flat on one of its faces is therefore
//     never opened by the collapse: its own box test is kept, exactly where the reference makes it.
// What the visit ORDER decides is only which of two accepted hits with bitwise equal t wins (`t < distance` is strict,
// extensionRayCast.hlsl:64-74).  This kernel walks in whatever order is cheapest, notices when the closest hit of an extension ray is
// such a tie between two DIFFERENT triangles (the duplicated references of one triangle give identical hit records either way) and
// then PARKS the ray in its lane; parked rays are walked again in the reference's binary near-first order on the packed binary copy
// (Node64), all parked lanes of the wave at once -- at the end of the wave's life, or as soon as a quarter of its lanes are parked.
// A handful of rays per launch on ordinary scenes, most rays on the adversarial grid meshes of the tests.  A shadow ray is any-hit:
// its result does not depend on the order at all (shadowRayCast.hlsl:88-91).
//
// Per lane: ONE array of kWideStack words in LDS ([entry][lane], conflict-free) holds two stacks growing towards each other -- inner
// nodes still to visit from the bottom up, leaves still to test from the top down.  A step fetches one WNode (seven 16-byte pieces:
// six planes x four slots, four links), does the four slab tests, keeps one hit inner slot as the next node and pushes the other
// hits.  Leaves are tested in wave-wide bursts, one triangle record per step, fetched together with the node of the walking lanes
// (as k_cast_f).  A lane whose inner stack alone fills its LDS share gives the wide walk of that ray up and parks it as well (the exact
// walk has a bounds-checked global overflow).  Drain: finished lanes take the BOTTOM inner entry of busy lanes (the largest subtree
// left) and report their best hit back; results merge by "smaller t wins, equal t is a tie".
#include "pt_traverse_deferred.hpp"

namespace gmupt {

#ifndef GMUPT_WIDE_STACK
#define GMUPT_WIDE_STACK 24
#endif
#ifndef GMUPT_WIDE_TOP
#define GMUPT_WIDE_TOP 512
#endif
#ifndef GMUPT_WIDE_REPS
#define GMUPT_WIDE_REPS 6
#endif
#ifndef GMUPT_WIDE_PARK
#define GMUPT_WIDE_PARK 16
#endif
#ifndef GMUPT_WIDE_QUADPK
#define GMUPT_WIDE_QUADPK 1     // the packed slab arithmetic one plane row per asm statement (four operations per wait state)
#endif
#ifndef GMUPT_WIDE_QUADTRI
#define GMUPT_WIDE_QUADTRI 1    // the same for the triangle-pair arithmetic (four statements)
#endif
#ifndef GMUPT_WIDE_SIGNED
#define GMUPT_WIDE_SIGNED 1     // planes fetched in ray-sign order, ordered slab tree unless a walking ray of the wave has a special 1 / d
#endif
constexpr int kWideStack = GMUPT_WIDE_STACK;   // LDS words per lane shared by the two stacks
constexpr int kWideTop = GMUPT_WIDE_TOP;       // WNodes of the tree top kept in LDS
constexpr int kWideRoom = 4;                   // a node step pushes at most four entries
constexpr int kWidePark = GMUPT_WIDE_PARK;     // parked lanes of a wave that trigger the exact walk before the wave has run dry
static_assert(kWideStack * kDefBlock * 4 + kWideTop * 128 <= 160 * 1024, "LDS of one CU (wide layout)");
constexpr int kWideOvf = kMaxStack + 2 > kWideStack ? kMaxStack + 2 - kWideStack : 0;   // the exact walk's stack: the lane's LDS share, then this many entries of the global overflow buffer

typedef int vec4i __attribute__((ext_vector_type(4)));

// One WNode into registers: from the LDS copy of the tree top (the top walk), or from global memory (everything below the top)
__device__ __forceinline__ void load_wnode_lds(const float4* s_top, int cur, vec4f& q0, vec4f& q1, vec4f& q2, vec4f& q3, vec4f& q4, vec4f& q5, vec4i& lk)
{
    const GMUPT_AS_LDS vec4f* n = (const GMUPT_AS_LDS vec4f*)(s_top) + cur * 8;
    q0 = n[0]; q1 = n[1]; q2 = n[2]; q3 = n[5]; q4 = n[4]; q5 = n[3]; lk = *(const GMUPT_AS_LDS vec4i*)(n + 6);   // rows: min x, y, z, max z, y, x
}
__device__ __forceinline__ void load_wnode_glb(__amdgpu_buffer_rsrc_t nodes, int cur, vec4f& q0, vec4f& q1, vec4f& q2, vec4f& q3, vec4f& q4, vec4f& q5, vec4i& lk)
{
    const int off = cur * 128;
    q0 = __builtin_bit_cast(vec4f, __builtin_amdgcn_raw_buffer_load_b128(nodes, off, 0, 0));
    q1 = __builtin_bit_cast(vec4f, __builtin_amdgcn_raw_buffer_load_b128(nodes, off + 16, 0, 0));
    q2 = __builtin_bit_cast(vec4f, __builtin_amdgcn_raw_buffer_load_b128(nodes, off + 32, 0, 0));
    q3 = __builtin_bit_cast(vec4f, __builtin_amdgcn_raw_buffer_load_b128(nodes, off + 80, 0, 0));
    q4 = __builtin_bit_cast(vec4f, __builtin_amdgcn_raw_buffer_load_b128(nodes, off + 64, 0, 0));
    q5 = __builtin_bit_cast(vec4f, __builtin_amdgcn_raw_buffer_load_b128(nodes, off + 48, 0, 0));
    lk = __builtin_bit_cast(vec4i, __builtin_amdgcn_raw_buffer_load_b128(nodes, off + 96, 0, 0));
}

// The planes of a WNode in RAY-SIGN order: q0 .. q2 = the plane of each axis the ray meets first (row `axis` for a positive, row 5 - axis for a
// negative 1 / d), q3 .. q5 = the other one.  Which row is a per-lane byte offset (RayPk's packed `sg`: bytes 0..2 = 0|80, 16|64, 32|48), added to
// the lane's record address with one SDWA add per axis; the far row of every axis is at (2 * record + 80) - near address.
__device__ __forceinline__ uint32_t add_byte0(uint32_t a, uint32_t b) { uint32_t r; asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ uint32_t add_byte1(uint32_t a, uint32_t b) { uint32_t r; asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ uint32_t add_byte2(uint32_t a, uint32_t b) { uint32_t r; asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2" : "=v"(r) : "v"(a), "v"(b)); return r; }
struct PlaneAddr { uint32_t nx, ny, nz, fx, fy, fz, rec; };
__device__ __forceinline__ PlaneAddr plane_addresses(int cur, uint32_t sg)
{
    PlaneAddr a; a.rec = (uint32_t)cur * 128u;
    const uint32_t k2 = a.rec * 2u + 80u;
    a.nx = add_byte0(a.rec, sg); a.ny = add_byte1(a.rec, sg); a.nz = add_byte2(a.rec, sg);
    a.fx = k2 - a.nx; a.fy = k2 - a.ny; a.fz = k2 - a.nz;
    return a;
}
__device__ __forceinline__ void load_wnode_lds_signed(const float4* s_top, const PlaneAddr a, vec4f& q0, vec4f& q1, vec4f& q2, vec4f& q3, vec4f& q4, vec4f& q5, vec4i& lk)
{
    const uint32_t base = (uint32_t)(uintptr_t)((const GMUPT_AS_LDS vec4f*)(s_top));     // 0: the top is the first thing in LDS
    q0 = *(const GMUPT_AS_LDS vec4f*)(uintptr_t)(base + a.nx); q1 = *(const GMUPT_AS_LDS vec4f*)(uintptr_t)(base + a.ny); q2 = *(const GMUPT_AS_LDS vec4f*)(uintptr_t)(base + a.nz);
    q3 = *(const GMUPT_AS_LDS vec4f*)(uintptr_t)(base + a.fx); q4 = *(const GMUPT_AS_LDS vec4f*)(uintptr_t)(base + a.fy); q5 = *(const GMUPT_AS_LDS vec4f*)(uintptr_t)(base + a.fz);
    lk = *(const GMUPT_AS_LDS vec4i*)(uintptr_t)(base + a.rec + 96u);
}
__device__ __forceinline__ void load_wnode_glb_signed(__amdgpu_buffer_rsrc_t nodes, const PlaneAddr a, vec4f& q0, vec4f& q1, vec4f& q2, vec4f& q3, vec4f& q4, vec4f& q5, vec4i& lk)
{
    q0 = __builtin_bit_cast(vec4f, __builtin_amdgcn_raw_buffer_load_b128(nodes, (int)a.nx, 0, 0));
    q1 = __builtin_bit_cast(vec4f, __builtin_amdgcn_raw_buffer_load_b128(nodes, (int)a.ny, 0, 0));
    q2 = __builtin_bit_cast(vec4f, __builtin_amdgcn_raw_buffer_load_b128(nodes, (int)a.nz, 0, 0));
    q3 = __builtin_bit_cast(vec4f, __builtin_amdgcn_raw_buffer_load_b128(nodes, (int)a.fx, 0, 0));
    q4 = __builtin_bit_cast(vec4f, __builtin_amdgcn_raw_buffer_load_b128(nodes, (int)a.fy, 0, 0));
    q5 = __builtin_bit_cast(vec4f, __builtin_amdgcn_raw_buffer_load_b128(nodes, (int)a.fz, 0, 0));
    lk = __builtin_bit_cast(vec4i, __builtin_amdgcn_raw_buffer_load_b128(nodes, (int)a.rec + 96, 0, 0));
}

// the slab test of ray_box (extensionRayCast.hlsl:79-94) on one slot; "hit" is `result > 0`, i.e. t1 >= t0 and (t0 > 0 ? t0 : t1) > 0,
// which is t1 >= t0 && t1 > 0 (t1 >= t0 > 0 implies t1 > 0; a NaN fails both forms)
__device__ __forceinline__ bool slab_hit(float mnx, float mny, float mnz, float mxx, float mxy, float mxz, f3 o, f3 invdir)
{
    const float fx = (mxx - o.x) * invdir.x, fy = (mxy - o.y) * invdir.y, fz = (mxz - o.z) * invdir.z;
    const float nx = (mnx - o.x) * invdir.x, ny = (mny - o.y) * invdir.y, nz = (mnz - o.z) * invdir.z;
    const float t1 = __builtin_fminf(__builtin_fmaxf(fx, nx), __builtin_fminf(__builtin_fmaxf(fy, ny), __builtin_fmaxf(fz, nz)));
    const float t0 = __builtin_fmaxf(__builtin_fminf(fx, nx), __builtin_fmaxf(__builtin_fminf(fy, ny), __builtin_fminf(fz, nz)));
    return (t1 >= t0) & (t1 > 0.0f);
}

// source-triangle number of a reference (word 10 of its Tri48 record): equal for the duplicated references of one triangle
__device__ __forceinline__ uint32_t tri_canon(__amdgpu_buffer_rsrc_t tris, int i)
{
    return (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(tris, i * 48 + 40, 0, 0);
}

// Packed binary32 arithmetic (v_pk_add_f32 / v_pk_mul_f32: the same IEEE operations as the scalar instructions, two at a time) as the
// machine instructions themselves.  Why not vector types and the compiler: (1) it puts a canonicalising `v_max x, x` in front of every
// fminf / fmaxf that consumes a packed product; (2) a scalar operand that both halves must see (a ray component) it either copies into
// a register pair of its own -- 18 registers this kernel does not have -- or re-aligns through SCRATCH memory in every step; the
// instruction itself can take either half of an aligned pair for both results (op_sel).  So the ray lives in five aligned pairs (RayPk)
// and each helper names the half it broadcasts.  Every packed instruction is followed by `s_nop 0`: on gfx950 its result may not be read
// by the very next instruction (the compiler pads its own packed code the same way; it cannot see into an asm statement).
// v_min / v_max / v_min3 / v_max3 return the non-NaN operand like the HLSL min / max, and no operand here can be a signalling NaN
// (products, and the quiet NaNs of empty slots).
#ifndef GMUPT_WIDE_PK
#define GMUPT_WIDE_PK 1
#endif
#define GMUPT_PK2(NAME, TEXT) __device__ __forceinline__ vec2f NAME(vec2f a, vec2f b) { vec2f r; asm(TEXT "\n\ts_nop 0" : "=v"(r) : "v"(a), "v"(b)); return r; }
GMUPT_PK2(pk_mul, "v_pk_mul_f32 %0, %1, %2")                                                        // a * b
GMUPT_PK2(pk_add, "v_pk_add_f32 %0, %1, %2")                                                        // a + b
GMUPT_PK2(pk_sub, "v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]")                              // a - b
GMUPT_PK2(pk_mul_lo, "v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]")                                     // a * (b.x, b.x)
GMUPT_PK2(pk_mul_hi, "v_pk_mul_f32 %0, %1, %2 op_sel:[0,1]")                                        // a * (b.y, b.y)
GMUPT_PK2(pk_sub_lo, "v_pk_add_f32 %0, %1, %2 op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]")           // a - (b.x, b.x)
GMUPT_PK2(pk_sub_hi, "v_pk_add_f32 %0, %1, %2 op_sel:[0,1] neg_lo:[0,1] neg_hi:[0,1]")              // a - (b.y, b.y)
GMUPT_PK2(pk_lo_sub, "v_pk_add_f32 %0, %2, %1 op_sel_hi:[0,1] neg_lo:[0,1] neg_hi:[0,1]")           // (b.x, b.x) - a
GMUPT_PK2(pk_hi_sub, "v_pk_add_f32 %0, %2, %1 op_sel:[1,0] neg_lo:[0,1] neg_hi:[0,1]")              // (b.y, b.y) - a
#undef GMUPT_PK2
__device__ __forceinline__ float v_min(float a, float b) { float r; asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float v_max(float a, float b) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float v_min3(float a, float b, float c) { float r; asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
__device__ __forceinline__ float v_max3(float a, float b, float c) { float r; asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }

// The lane's ray in five aligned register pairs: (o.x, o.y), (o.z, 1/d.z), (1/d.x, 1/d.y), (d.x, d.y), (d.z, sg).  sg (raw bits): byte k = the byte
// offset of the plane row the ray meets first on axis k (load_wnode_*_signed), byte 3 = 1 when a component of 1 / d is infinite or NaN
// (a direction component that is zero, denormal or NaN): for such a ray (plane - o) * (1 / d) can be NaN, and min / max decide which plane
// is the near one -- the wave then takes the general form of the test.
struct RayPk { vec2f oxy, ozi, ixy, dxy, dzz; };
__device__ __forceinline__ uint32_t ray_sg(const RayPk& r) { return __builtin_bit_cast(vec2u, r.dzz).y; }
__device__ __forceinline__ f3 ray_o(const RayPk& r) { return mk3(r.oxy.x, r.oxy.y, r.ozi.x); }
__device__ __forceinline__ f3 ray_inv(const RayPk& r) { return mk3(r.ixy.x, r.ixy.y, r.ozi.y); }
__device__ __forceinline__ f3 ray_d(const RayPk& r) { return mk3(r.dxy.x, r.dxy.y, r.dzz.x); }
__device__ __forceinline__ void ray_set(RayPk& r, f3 o, f3 d)
{
    r.oxy.x = o.x; r.oxy.y = o.y; r.ozi.x = o.z; r.ixy.x = 1.0f / d.x; r.ixy.y = 1.0f / d.y; r.ozi.y = 1.0f / d.z;
    r.dxy.x = d.x; r.dxy.y = d.y; r.dzz.x = d.z;
    const vec2u ib = __builtin_bit_cast(vec2u, r.ixy), zb = __builtin_bit_cast(vec2u, r.ozi);      // (whole-vector casts: a bit cast of a vector ELEMENT reads element 0)
    const uint32_t bx = ib.x, by = ib.y, bz = zb.y;
    const bool special = ((bx & 0x7F800000u) == 0x7F800000u) | ((by & 0x7F800000u) == 0x7F800000u) | ((bz & 0x7F800000u) == 0x7F800000u);
    const uint32_t sg = ((bx >> 31) ? 80u : 0u) | ((by >> 31) ? (64u << 8) : (16u << 8)) | ((bz >> 31) ? (48u << 16) : (32u << 16)) | (special ? (1u << 24) : 0u);
    vec2u dz = __builtin_bit_cast(vec2u, r.dzz); dz.y = sg; r.dzz = __builtin_bit_cast(vec2f, dz);
}

#define GMUPT_LO(Q) __builtin_shufflevector(Q, Q, 0, 1)
#define GMUPT_HI(Q) __builtin_shufflevector(Q, Q, 2, 3)
__device__ __forceinline__ bool slab_hit_pk(float nx, float ny, float nz, float fx, float fy, float fz)
{
    const float t1 = v_min3(v_max(fx, nx), v_max(fy, ny), v_max(fz, nz));
    const float t0 = v_max3(v_min(fx, nx), v_min(fy, ny), v_min(fz, nz));
    return (t1 >= t0) & (t1 > 0.0f);
}
// the same test when the near / far plane of every axis is known (rows in ray-sign order, 1 / d finite): (near - o) / d <= (far - o) / d by the
// monotonicity of the two rounded operations, so min(f, n) IS n and max(f, n) IS f -- six of the eight min / max of a slot go
__device__ __forceinline__ bool slab_hit_ordered(float nx, float ny, float nz, float fx, float fy, float fz)
{
    const float t1 = v_min3(fx, fy, fz);
    const float t0 = v_max3(nx, ny, nz);
    return (t1 >= t0) & (t1 > 0.0f);
}
// the four slab tests of a step: (plane - o) * (1 / d) for two slots per packed instruction, then the min / max tree of ray_box per slot
// (GENERAL: wave-uniform; the rows may be in either order for it)
__device__ __forceinline__ void slab_hits4(const vec4f q0, const vec4f q1, const vec4f q2, const vec4f q3, const vec4f q4, const vec4f q5, const RayPk& ray,
                                           bool& h0, bool& h1, bool& h2, bool& h3, const bool general = true)
{
#if GMUPT_WIDE_PK && GMUPT_WIDE_QUADPK
    // one plane row (four slots) per statement: two subtractions, two multiplications -- a product is read two instructions after its
    // difference was written, so one wait state at the end covers the statement (4 instead of 1 packed operations per `s_nop`)
    vec2f nxa, nxb, nya, nyb, nza, nzb, fxa, fxb, fya, fyb, fza, fzb;
#define GMUPT_ROW(A, B, Q, O, OSEL, INV, ISEL) \
    asm("v_pk_add_f32 %0, %2, %4 " OSEL " neg_lo:[0,1] neg_hi:[0,1]\n\tv_pk_add_f32 %1, %3, %4 " OSEL " neg_lo:[0,1] neg_hi:[0,1]\n\t" \
        "v_pk_mul_f32 %0, %0, %5 " ISEL "\n\tv_pk_mul_f32 %1, %1, %5 " ISEL "\n\ts_nop 0" \
        : "=&v"(A), "=&v"(B) : "v"(GMUPT_LO(Q)), "v"(GMUPT_HI(Q)), "v"(O), "v"(INV))
    GMUPT_ROW(nxa, nxb, q0, ray.oxy, "op_sel_hi:[1,0]", ray.ixy, "op_sel_hi:[1,0]");
    GMUPT_ROW(nya, nyb, q1, ray.oxy, "op_sel:[0,1]", ray.ixy, "op_sel:[0,1]");
    GMUPT_ROW(nza, nzb, q2, ray.ozi, "op_sel_hi:[1,0]", ray.ozi, "op_sel:[0,1]");
    GMUPT_ROW(fxa, fxb, q3, ray.oxy, "op_sel_hi:[1,0]", ray.ixy, "op_sel_hi:[1,0]");
    GMUPT_ROW(fya, fyb, q4, ray.oxy, "op_sel:[0,1]", ray.ixy, "op_sel:[0,1]");
    GMUPT_ROW(fza, fzb, q5, ray.ozi, "op_sel_hi:[1,0]", ray.ozi, "op_sel:[0,1]");
#undef GMUPT_ROW
#elif GMUPT_WIDE_PK
    const vec2f nxa = pk_mul_lo(pk_sub_lo(GMUPT_LO(q0), ray.oxy), ray.ixy), nxb = pk_mul_lo(pk_sub_lo(GMUPT_HI(q0), ray.oxy), ray.ixy);
    const vec2f nya = pk_mul_hi(pk_sub_hi(GMUPT_LO(q1), ray.oxy), ray.ixy), nyb = pk_mul_hi(pk_sub_hi(GMUPT_HI(q1), ray.oxy), ray.ixy);
    const vec2f nza = pk_mul_hi(pk_sub_lo(GMUPT_LO(q2), ray.ozi), ray.ozi), nzb = pk_mul_hi(pk_sub_lo(GMUPT_HI(q2), ray.ozi), ray.ozi);
    const vec2f fxa = pk_mul_lo(pk_sub_lo(GMUPT_LO(q3), ray.oxy), ray.ixy), fxb = pk_mul_lo(pk_sub_lo(GMUPT_HI(q3), ray.oxy), ray.ixy);
    const vec2f fya = pk_mul_hi(pk_sub_hi(GMUPT_LO(q4), ray.oxy), ray.ixy), fyb = pk_mul_hi(pk_sub_hi(GMUPT_HI(q4), ray.oxy), ray.ixy);
    const vec2f fza = pk_mul_hi(pk_sub_lo(GMUPT_LO(q5), ray.ozi), ray.ozi), fzb = pk_mul_hi(pk_sub_lo(GMUPT_HI(q5), ray.ozi), ray.ozi);
#endif
#if GMUPT_WIDE_PK
    if (general) {
        h0 = slab_hit_pk(nxa.x, nya.x, nza.x, fxa.x, fya.x, fza.x); h1 = slab_hit_pk(nxa.y, nya.y, nza.y, fxa.y, fya.y, fza.y);
        h2 = slab_hit_pk(nxb.x, nyb.x, nzb.x, fxb.x, fyb.x, fzb.x); h3 = slab_hit_pk(nxb.y, nyb.y, nzb.y, fxb.y, fyb.y, fzb.y);
    } else {
        h0 = slab_hit_ordered(nxa.x, nya.x, nza.x, fxa.x, fya.x, fza.x); h1 = slab_hit_ordered(nxa.y, nya.y, nza.y, fxa.y, fya.y, fza.y);
        h2 = slab_hit_ordered(nxb.x, nyb.x, nzb.x, fxb.x, fyb.x, fzb.x); h3 = slab_hit_ordered(nxb.y, nyb.y, nzb.y, fxb.y, fyb.y, fzb.y);
    }
#else
    const f3 o = ray_o(ray), invdir = ray_inv(ray);
    h0 = slab_hit(q0.x, q1.x, q2.x, q3.x, q4.x, q5.x, o, invdir); h1 = slab_hit(q0.y, q1.y, q2.y, q3.y, q4.y, q5.y, o, invdir);
    h2 = slab_hit(q0.z, q1.z, q2.z, q3.z, q4.z, q5.z, o, invdir); h3 = slab_hit(q0.w, q1.w, q2.w, q3.w, q4.w, q5.w, o, invdir);
#endif
}

// Moeller-Trumbore (extensionRayCast.hlsl:38-62 == shadowRayCast.hlsl:16-40) on the TWO triangles of a TriPair at once: every operation of
// tri_compute_flat, in its order, on a (first, second) pair of operands -- packed where the machine has the instruction (sub, mul, add),
// the scalar instruction twice where it has not (the correctly rounded division, the comparisons).  okA / okB: not rejected.
struct PairHit { vec2f t, u, v; bool okA, okB; bool last; };
__device__ __forceinline__ PairHit tri_pair_compute(const vec4f a0, const vec4f a1, const vec4f a2, const vec4f a3, const vec4f a4, const RayPk& ray)
{
    const vec2f v0x = GMUPT_LO(a0), v0y = GMUPT_HI(a0), v0z = GMUPT_LO(a1), e1x = GMUPT_HI(a1), e1y = GMUPT_LO(a2), e1z = GMUPT_HI(a2);
    const vec2f e2x = GMUPT_LO(a3), e2y = GMUPT_HI(a3), e2z = GMUPT_LO(a4);
    PairHit h;
#if GMUPT_WIDE_PK && GMUPT_WIDE_QUADPK && GMUPT_WIDE_QUADTRI
    // the same operations as the branch below, in four statements of independent instructions (one wait state per statement instead of
    // one per packed instruction: no result is read by the instruction that follows the one that wrote it)
    vec2f px, py, pz, tx, ty, tz, m1, m2, m3;
    asm("v_pk_mul_f32 %0, %11, %15 op_sel:[0,1]\n\t"                                    // pvec = cross3(d, e2): e2z d.y
        "v_pk_mul_f32 %6, %10, %16 op_sel_hi:[1,0]\n\t"                                 //   e2y d.z
        "v_pk_mul_f32 %1, %9, %16 op_sel_hi:[1,0]\n\t"                                  //   e2x d.z
        "v_pk_mul_f32 %7, %11, %15 op_sel_hi:[1,0]\n\t"                                 //   e2z d.x
        "v_pk_mul_f32 %2, %10, %15 op_sel_hi:[1,0]\n\t"                                 //   e2y d.x
        "v_pk_mul_f32 %8, %9, %15 op_sel:[0,1]\n\t"                                     //   e2x d.y
        "v_pk_add_f32 %3, %17, %12 op_sel_hi:[0,1] neg_lo:[0,1] neg_hi:[0,1]\n\t"       // tvec = o - v0
        "v_pk_add_f32 %4, %17, %13 op_sel:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n\t"
        "v_pk_add_f32 %5, %18, %14 op_sel_hi:[0,1] neg_lo:[0,1] neg_hi:[0,1]\n\t"
        "v_pk_add_f32 %0, %0, %6 neg_lo:[0,1] neg_hi:[0,1]\n\t"
        "v_pk_add_f32 %1, %1, %7 neg_lo:[0,1] neg_hi:[0,1]\n\t"
        "v_pk_add_f32 %2, %2, %8 neg_lo:[0,1] neg_hi:[0,1]\n\ts_nop 0"
        : "=&v"(px), "=&v"(py), "=&v"(pz), "=&v"(tx), "=&v"(ty), "=&v"(tz), "=&v"(m1), "=&v"(m2), "=&v"(m3)
        : "v"(e2x), "v"(e2y), "v"(e2z), "v"(v0x), "v"(v0y), "v"(v0z), "v"(ray.dxy), "v"(ray.dzz), "v"(ray.oxy), "v"(ray.ozi));
    vec2f det, uu, c1, c2, c3, c4;
    asm("v_pk_mul_f32 %0, %6, %9\n\tv_pk_mul_f32 %2, %7, %10\n\tv_pk_mul_f32 %3, %8, %11\n\t"        // dot3(e1, pvec)
        "v_pk_mul_f32 %1, %12, %9\n\tv_pk_mul_f32 %4, %13, %10\n\tv_pk_mul_f32 %5, %14, %11\n\t"     // dot3(tvec, pvec)
        "v_pk_add_f32 %0, %0, %2\n\tv_pk_add_f32 %1, %1, %4\n\tv_pk_add_f32 %0, %0, %3\n\tv_pk_add_f32 %1, %1, %5\n\ts_nop 0"
        : "=&v"(det), "=&v"(uu), "=&v"(c1), "=&v"(c2), "=&v"(c3), "=&v"(c4)
        : "v"(e1x), "v"(e1y), "v"(e1z), "v"(px), "v"(py), "v"(pz), "v"(tx), "v"(ty), "v"(tz));
    vec2f qx, qy, qz;
    asm("v_pk_mul_f32 %0, %7, %11\n\tv_pk_mul_f32 %3, %8, %10\n\tv_pk_mul_f32 %1, %8, %9\n\t"        // qvec = cross3(tvec, e1)
        "v_pk_mul_f32 %4, %6, %11\n\tv_pk_mul_f32 %2, %6, %10\n\tv_pk_mul_f32 %5, %7, %9\n\t"
        "v_pk_add_f32 %0, %0, %3 neg_lo:[0,1] neg_hi:[0,1]\n\tv_pk_add_f32 %1, %1, %4 neg_lo:[0,1] neg_hi:[0,1]\n\tv_pk_add_f32 %2, %2, %5 neg_lo:[0,1] neg_hi:[0,1]\n\ts_nop 0"
        : "=&v"(qx), "=&v"(qy), "=&v"(qz), "=&v"(m1), "=&v"(m2), "=&v"(m3)
        : "v"(tx), "v"(ty), "v"(tz), "v"(e1x), "v"(e1y), "v"(e1z));
    vec2f invDet; invDet.x = 1.0f / det.x; invDet.y = 1.0f / det.y;
    vec2f vv, tt;
    asm("v_pk_mul_f32 %1, %7, %10 op_sel_hi:[1,0]\n\tv_pk_mul_f32 %3, %8, %10 op_sel:[0,1]\n\tv_pk_mul_f32 %4, %9, %11 op_sel_hi:[1,0]\n\t"   // dot3(d, qvec)
        "v_pk_mul_f32 %2, %12, %7\n\tv_pk_mul_f32 %5, %13, %8\n\tv_pk_mul_f32 %6, %14, %9\n\t"                                                // dot3(e2, qvec)
        "v_pk_mul_f32 %0, %0, %15\n\t"                                                                                                          // u
        "v_pk_add_f32 %1, %1, %3\n\tv_pk_add_f32 %2, %2, %5\n\tv_pk_add_f32 %1, %1, %4\n\tv_pk_add_f32 %2, %2, %6\n\t"
        "v_pk_mul_f32 %1, %1, %15\n\tv_pk_mul_f32 %2, %2, %15\n\ts_nop 0"                                                                      // v, t
        : "+v"(uu), "=&v"(vv), "=&v"(tt), "=&v"(c1), "=&v"(c2), "=&v"(c3), "=&v"(c4)
        : "v"(qx), "v"(qy), "v"(qz), "v"(ray.dxy), "v"(ray.dzz), "v"(e2x), "v"(e2y), "v"(e2z), "v"(invDet));
    h.u = uu; h.v = vv; h.t = tt;
#elif GMUPT_WIDE_PK
    // pvec = cross3(d, e2) = (d.y e2.z - d.z e2.y, d.z e2.x - d.x e2.z, d.x e2.y - d.y e2.x)
    const vec2f px = pk_sub(pk_mul_hi(e2z, ray.dxy), pk_mul_lo(e2y, ray.dzz)), py = pk_sub(pk_mul_lo(e2x, ray.dzz), pk_mul_lo(e2z, ray.dxy)), pz = pk_sub(pk_mul_lo(e2y, ray.dxy), pk_mul_hi(e2x, ray.dxy));
    const vec2f det = pk_add(pk_add(pk_mul(e1x, px), pk_mul(e1y, py)), pk_mul(e1z, pz));                     // dot3(e1, pvec)
    vec2f invDet; invDet.x = 1.0f / det.x; invDet.y = 1.0f / det.y;
    const vec2f tx = pk_lo_sub(v0x, ray.oxy), ty = pk_hi_sub(v0y, ray.oxy), tz = pk_lo_sub(v0z, ray.ozi);   // tvec = o - v0
    h.u = pk_mul(pk_add(pk_add(pk_mul(tx, px), pk_mul(ty, py)), pk_mul(tz, pz)), invDet);                     // dot3(tvec, pvec) * invDet
    const vec2f qx = pk_sub(pk_mul(ty, e1z), pk_mul(tz, e1y)), qy = pk_sub(pk_mul(tz, e1x), pk_mul(tx, e1z)), qz = pk_sub(pk_mul(tx, e1y), pk_mul(ty, e1x));   // qvec = cross3(tvec, e1)
    h.v = pk_mul(pk_add(pk_add(pk_mul_lo(qx, ray.dxy), pk_mul_hi(qy, ray.dxy)), pk_mul_lo(qz, ray.dzz)), invDet);   // dot3(d, qvec) * invDet
    h.t = pk_mul(pk_add(pk_add(pk_mul(e2x, qx), pk_mul(e2y, qy)), pk_mul(e2z, qz)), invDet);                  // dot3(e2, qvec) * invDet
#else
    const f3 o = ray_o(ray), d = ray_d(ray);
    const vec2f dx = { d.x, d.x }, dy = { d.y, d.y }, dz = { d.z, d.z }, ox = { o.x, o.x }, oy = { o.y, o.y }, oz = { o.z, o.z };
    const vec2f px = dy * e2z - dz * e2y, py = dz * e2x - dx * e2z, pz = dx * e2y - dy * e2x;
    const vec2f det = (e1x * px + e1y * py) + e1z * pz;
    vec2f invDet; invDet.x = 1.0f / det.x; invDet.y = 1.0f / det.y;
    const vec2f tx = ox - v0x, ty = oy - v0y, tz = oz - v0z;
    h.u = ((tx * px + ty * py) + tz * pz) * invDet;
    const vec2f qx = ty * e1z - tz * e1y, qy = tz * e1x - tx * e1z, qz = tx * e1y - ty * e1x;
    h.v = ((dx * qx + dy * qy) + dz * qz) * invDet;
    h.t = ((e2x * qx + e2y * qy) + e2z * qz) * invDet;
#endif
    h.okA = !((det.x > -kEpsilon && det.x < kEpsilon) | (h.u.x < 0.0f) | (h.u.x > 1.0f) | (h.v.x < 0.0f) | (h.u.x + h.v.x > 1.0f));
    h.okB = !((det.y > -kEpsilon && det.y < kEpsilon) | (h.u.y < 0.0f) | (h.u.y > 1.0f) | (h.v.y < 0.0f) | (h.u.y + h.v.y > 1.0f));
    h.last = __builtin_bit_cast(vec4u, a4).z != 0u;   // (whole-vector bit cast: __builtin_bit_cast of a vector ELEMENT reads element 0 with this compiler)
    return h;
}
#undef GMUPT_LO
#undef GMUPT_HI

template <bool STATS, int REPS>
__global__ __launch_bounds__(kDefBlock) void k_cast_w(RenderParams p)
{
    constexpr int S = kWideStack;
    // one LDS object, the tree top first: a top node's LDS address is then its byte offset in the node table (cur * 128), and the plane
    // addresses of a step (load_wnode_*_signed) serve the LDS and the vector-memory fetch alike
    __shared__ float4 s_lds[kWideTop * 8 + S * kDefBlock / 4];
    float4* s_top = s_lds;
    int* s_stack = reinterpret_cast<int*>(s_lds + kWideTop * 8);
    {
        const float4* src = reinterpret_cast<const float4*>(p.trav.wnodes);
        for (uint32_t k = threadIdx.x; k < p.trav.wideTopCount * 8u; k += kDefBlock) s_top[k] = src[k];
        __syncthreads();
    }
    shadow_counter_epilogue(p);
    int* sl = s_stack + threadIdx.x;     // entry i of this lane: sl[i * kDefBlock]
    TravCount tcE = { 0, 0, 0 }, tcS = { 0, 0, 0 }; uint32_t raysE = 0, raysS = 0, wInE = 0, wTrE = 0, wInS = 0, wTrS = 0;
    const TravScene& ts = p.trav;
    const uint32_t countExt = p.qc[QC_EXT_COUNT], countSh = p.qc[QC_SHADOWRAY];  // extensionRayCast.hlsl:205, shadowRayCast.hlsl:151
    const uint32_t* qExt = p.queues + (size_t)Q_EXT_RAY * p.P;
    const uint32_t* qSh = p.queues + (size_t)Q_SHADOW_RAY * p.P;
    const __amdgpu_buffer_rsrc_t rNodes = make_rsrc(ts.wnodes, ts.wideCount * 128u);
    const __amdgpu_buffer_rsrc_t rTris = make_rsrc(ts.tris, (p.scene.numTris + 1u) * 48u);    // + the sentinel record (the exact walk; the source-triangle numbers)
    const __amdgpu_buffer_rsrc_t rPairs = make_rsrc(ts.pairs, ts.numPairs * 80u);
    const uint32_t topCount = ts.wideTopCount;
    uint32_t next = 0, end = 0, lastBase = 0;
    uint32_t chunkBase = 0, qe0 = kQueueHole, qe1 = kQueueHole;  // the current chunk of queue entries, lane l holds entries l and 64 + l
    int phase = 0;
    uint32_t census0 = 0, census1 = 0, census2 = 0, census3 = 0, topE = 0, topS = 0, helped = 0, nested = 0, boxes = 0, pairFetches = 0, itersAll = 0, itersGeneral = 0;

    bool haveRay = false;
    int kind = 0;                 // 0: extension ray, 1: shadow ray; 2 / 3: the same, PARKED for the exact walk
    uint32_t index = 0;
    RayPk ray; ray_set(ray, mk3(0, 0, 0), mk3(1, 1, 1));   // origin, direction and 1 / d of the lane's ray
    float distance = kFltMax;     // extension: closest hit so far; shadow: distance of the light
    float hu = 0.0f, hv = 0.0f;
    int hitRef = -1;              // extension: triangle record of the closest hit; shadow: >= 0 when occluded
    bool redo = false;            // the wide walk does not decide this ray: its closest hit ties with a hit of another triangle, or a stack ran full
    int cur = kDone;              // >= 0: WNode to visit; kDone: nothing in hand
    uint32_t pa = 0;              // inner stack: entries sl[bottom .. pa)
    uint32_t pb = S - 1;          // leaf stack: entries sl(pb .. S - 1], pushed downwards
    uint32_t bottom = 0;          // entries below were given to helper lanes
    int ti = -1;                  // next triangle record of the leaf being tested
    int owner = -1;               // >= 0: this lane walks a subtree given to it by lane `owner` (same ray)
    uint32_t outstanding = 0;     // helpers that have not reported yet
    uint32_t donations = 0;       // subtrees this lane has given away for its current ray
    float tT = kFltMax, uT = 0.0f, vT = 0.0f; int refT = -1; bool redoT = false;   // best result reported by this lane's helpers so far
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t gtid = blockIdx.x * kDefBlock + threadIdx.x;

    // the helpers' results into the lane's own, once all of them have reported (any order: the smaller t wins, equal t between two lanes
    // counts as a tie; an occluder found by anybody decides a shadow ray whatever else happened)
#define GMUPT_WIDE_MERGE_OWN() \
    do { if (kind == 0) { redo = redo || redoT || (refT >= 0 && tT == distance); if (refT >= 0 && tT < distance) { distance = tT; hu = uT; hv = vT; hitRef = refT; } } \
         else { if (refT >= 0) hitRef = refT; redo = (redo || redoT) && hitRef < 0; } \
         tT = kFltMax; refT = -1; redoT = false; } while (0)
    // the end of a decided ray in its lane
#define GMUPT_WIDE_FINISH() \
    do { if (kind == 0) finish_extension_ray(p, index, ray_o(ray), ray_d(ray), distance, hu, hv, hitRef >= 0 ? (int)ts.pairRef[hitRef] : -1);   /* extensionRayCast.hlsl:218-232; hitRef is a pair SLOT */ \
         else stu(p, F_IN_SHADOW, index, hitRef >= 0 ? 1u : 0u);                                    /* shadowRayCast.hlsl:167 */ \
         haveRay = false; redo = false; } while (0)

    // Watchdog: a persistent kernel must end whatever happens (see k_cast_f)
    uint32_t loopCount = 0;
    const unsigned long long tStart = STATS ? wall_clock64() : 0ull;   // (collect_stats: wave lifetimes, share of the drain)
    unsigned long long tDrain = 0ull; uint32_t drainIters = 0, drainBusy = 0;
    for (;;) {
        if (++loopCount > p.castLoopCap) { if (lane == 0u) atomicOr(&p.stats->stackOverflow, 2u); break; }
        const bool leavesEmpty = pb == (uint32_t)(S - 1);
        if (STATS && phase == 2) { if (tDrain == 0ull) tDrain = wall_clock64(); drainIters++; drainBusy += (uint32_t)__popcll(__ballot(haveRay)); }
        if (phase == 2) { // wave-uniform: drain service
            // (a) helpers that have finished their subtree (and have heard from their own helpers) report to their owner
            const bool reports = haveRay && owner >= 0 && outstanding == 0u && cur == kDone && leavesEmpty && ti < 0;
            if (reports) GMUPT_WIDE_MERGE_OWN();
            unsigned long long fin = __ballot(reports);
            while (fin) {
                const int hl = __builtin_ctzll(fin); fin &= fin - 1ull;
                const int ol = __builtin_amdgcn_readlane(owner, hl);
                const float th = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, distance), hl));
                const float uh = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, hu), hl));
                const float vh = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, hv), hl));
                const int rh = __builtin_amdgcn_readlane(hitRef, hl);
                const int redoh = __builtin_amdgcn_readlane((int)redo, hl);
                if ((int)lane == ol) {
                    outstanding--;
                    redoT = redoT || redoh != 0;
                    if (rh >= 0) {
                        if (kind == 1) refT = rh;                                                       // any occluder decides a shadow ray
                        else if (th < tT) { tT = th; uT = uh; vT = vh; refT = rh; }
                        else if (th == tT) redoT = true;
                    }
                }
                if ((int)lane == hl) { haveRay = false; owner = -1; tT = kFltMax; refT = -1; redoT = false; redo = false; }
            }
        }
        {   // a lane whose own walk has ended and whose helpers have all reported knows its result -- or that the wide walk did not decide
            // the ray: then the ray is PARKED in the lane (not idle, not free) until the exact walk below takes it
            const bool ended = haveRay && kind < 2 && owner < 0 && outstanding == 0u && cur == kDone && leavesEmpty && ti < 0;
            if (ended) { GMUPT_WIDE_MERGE_OWN(); if (redo) { kind += 2; redo = false; pa = 0; bottom = 0; } }
        }
        if (phase == 2) { // wave-uniform: drain service, continued
            // (b) owners that are done and have heard from all their helpers are written back now: their lanes become free
            const bool done = haveRay && kind < 2 && owner < 0 && outstanding == 0u && cur == kDone && leavesEmpty && ti < 0;
            // (a lane whose walk has ended never gives: an occluded shadow ray leaves its stack behind, and those subtrees no longer matter)
            const unsigned long long wantHelp = __ballot(haveRay && cur != kDone && pa > bottom && donations < 12u);
            if (wantHelp != 0ull && done) GMUPT_WIDE_FINISH();
            // (c) free lanes take the bottom inner entry of lanes that still have some: the k-th free lane pairs with the k-th donor
            const unsigned long long freeLanes = __ballot(!haveRay);
            if (wantHelp != 0ull && freeLanes != 0ull) {
                const uint32_t nDonors = (uint32_t)__popcll(wantHelp), nFree = (uint32_t)__popcll(freeLanes);
                const uint32_t nPairs = nDonors < nFree ? nDonors : nFree;
                const bool isDonor = ((wantHelp >> lane) & 1ull) != 0ull;
                const uint32_t dRank = prefix_rank(wantHelp), fRank = prefix_rank(freeLanes);
                const bool gives = isDonor && dRank < nPairs, takes = !haveRay && fRank < nPairs;
                int node = 0;
                if (gives) {
                    node = sl[bottom * kDefBlock];
                    bottom++; outstanding++; donations++;
                    if (STATS) { helped++; if (owner >= 0) nested++; }
                }
                const int donorOfRank = __builtin_amdgcn_ds_permute((int)((gives ? dRank : 63u) << 2), gives ? (int)lane : 0);
                const int src = __shfl(donorOfRank, takes ? (int)fRank : 0);
                const int srcLane = takes ? src : (int)lane;
                const int node2 = __shfl(node, srcLane), k2 = __shfl(kind, srcLane), i2 = __shfl((int)index, srcLane);
                const float ox = __shfl(ray.oxy.x, srcLane), oy = __shfl(ray.oxy.y, srcLane), oz = __shfl(ray.ozi.x, srcLane), dx = __shfl(ray.dxy.x, srcLane), dy = __shfl(ray.dxy.y, srcLane), dz = __shfl(ray.dzz.x, srcLane);
                const float lim = __shfl(distance, srcLane);
                if (takes) {
                    haveRay = true; owner = src; kind = k2; index = (uint32_t)i2; donations = 0;
                    ray_set(ray, mk3(ox, oy, oz), mk3(dx, dy, dz));
                    distance = k2 == 1 ? lim : kFltMax;     // shadow: the distance of the light; extension: no hit yet
                    hitRef = -1; hu = 0.0f; hv = 0.0f; redo = false;
                    pa = 0; bottom = 0; pb = S - 1; outstanding = 0; ti = -1;
                    tT = kFltMax; refT = -1; redoT = false;
                    cur = node2;
                }
            }
        }
        // (a helper, a lane that still waits for its helpers, or a parked lane is not idle)
        const bool parked = haveRay && kind >= 2;
        const bool idle = (cur == kDone) && (pb == (uint32_t)(S - 1)) && (ti < 0) && !(haveRay && (owner >= 0 || outstanding != 0u || kind >= 2));
        const unsigned long long idleMask = __ballot(idle);
        const int nIdle = __popcll(idleMask);
        const unsigned long long parkedMask = __ballot(parked);
        if (parkedMask != 0ull) {
            const int nParked = __popcll(parkedMask);
            if (nParked >= kWidePark || nParked + nIdle == 64) {   // wave-uniform
                // ---- the exact walk of the parked rays, as the reference walks them: binary tree, nearer child first, the farther one deferred,
                // triangles of a leaf in order, strict `t < distance` (extensionRayCast.hlsl:96-166); any occluder within the light's distance for a
                // shadow ray (shadowRayCast.hlsl:65-136).  Packed binary copy (Node64 / Tri48); stack = the lane's LDS share, then the global overflow.
                if (lane == 0u) atomicAdd(&p.stats->castRedoRays, (unsigned long long)nParked);
                if (parked) {
                    const bool shadowRay = kind == 3;
                    ray_set(ray, shadowRay ? ld3(p, F_SH_OX, index) : ld3(p, F_RAY_OX, index), shadowRay ? ld3(p, F_SH_DX, index) : ld3(p, F_RAY_DX, index));
                    distance = shadowRay ? ldf(p, F_LIGHT_DIST, index) : kFltMax;
                    const f3 o = ray_o(ray), invdir = ray_inv(ray), d = ray_d(ray);
                    hitRef = -1; hu = 0.0f; hv = 0.0f;
                    int* ovf = p.ovfStack + gtid;
                    uint32_t sp = 0;
#define GMUPT_EXACT_PUSH(V) do { if (sp < (uint32_t)S) sl[sp * kDefBlock] = (V); else if (sp - (uint32_t)S < (uint32_t)kWideOvf) ovf[(size_t)(sp - (uint32_t)S) * p.ovfStride] = (V); \
                                 else atomicOr(&p.stats->stackOverflow, 1u); sp++; } while (0)
#define GMUPT_EXACT_POP(DST) do { --sp; if (sp < (uint32_t)S) DST = sl[sp * kDefBlock]; else if (sp - (uint32_t)S < (uint32_t)kWideOvf) DST = ovf[(size_t)(sp - (uint32_t)S) * p.ovfStride]; else DST = kDone; } while (0)
                    GMUPT_EXACT_PUSH(kDone);
                    int c = (ray_box(ts.rootMin[0], ts.rootMin[1], ts.rootMin[2], ts.rootMax[0], ts.rootMax[1], ts.rootMax[2], o, invdir) > 0.0f) ? ts.rootDesc : kDone;
                    while (c != kDone) {
                        if (c >= 0) {
                            const Node64& nd = ts.nodes[c];
                            const float leftHit = ray_box(nd.a[0], nd.a[1], nd.a[2], nd.a[3], nd.b[0], nd.b[1], o, invdir);
                            const float rightHit = ray_box(nd.b[2], nd.b[3], nd.c[0], nd.c[1], nd.c[2], nd.c[3], o, invdir);
                            const bool l = leftHit > 0.0f, r = rightHit > 0.0f;
                            const bool swap = leftHit > rightHit;                 // extensionRayCast.hlsl:136
                            const int dl = nd.d[0], dr = nd.d[1];
                            if (l && r) { GMUPT_EXACT_PUSH(swap ? dl : dr); c = swap ? dr : dl; }
                            else if (l | r) c = l ? dl : dr;
                            else GMUPT_EXACT_POP(c);
                        } else {
                            int tix = ~c;
                            bool decided = false;
                            for (;;) {
                                float t = 0.0f, u = 0.0f, v = 0.0f; bool last;
                                if (tri_test(ts.tris, tix, o, d, t, u, v, last)) {
                                    if (shadowRay) { if (t > kEpsilon && t < 1.0f / kEpsilon && length3(d * t) < distance) { hitRef = tix; decided = true; break; } }   // shadowRayCast.hlsl:41-45,89
                                    else if (t >= 0.0f && t < distance) { distance = t; hitRef = tix; hu = u; hv = v; }                                              // extensionRayCast.hlsl:64-74
                                }
                                if (last) break;
                                tix++;
                            }
                            if (decided) c = kDone; else GMUPT_EXACT_POP(c);
                        }
                    }
#undef GMUPT_EXACT_PUSH
#undef GMUPT_EXACT_POP
                    if (shadowRay) stu(p, F_IN_SHADOW, index, hitRef >= 0 ? 1u : 0u);            // shadowRayCast.hlsl:167
                    else finish_extension_ray(p, index, o, d, distance, hu, hv, hitRef);         // extensionRayCast.hlsl:218-232
                    haveRay = false; kind = 0; redo = false; cur = kDone; pa = 0; bottom = 0; pb = S - 1; ti = -1;
                    tT = kFltMax; refT = -1; redoT = false; hitRef = -1;
                }
                continue;
            }
        }
        if (nIdle == 64 || (nIdle >= (int)p.tuneRefill && phase < 2)) { // wave-uniform
#ifndef GMUPT_WIDE_XCD_EXPERIMENT
#define GMUPT_WIDE_XCD_EXPERIMENT 0   // 1 (tools/xcd_experiment.py only): queues pre-binned into eight equal segments, one per XCD (does an L2 per subtree pay?)
#endif
#if GMUPT_WIDE_XCD_EXPERIMENT
            while (p.xcdBins && next >= end && phase < 2) {
                const uint32_t count = phase == 0 ? countExt : countSh, segLen = count >> 3;
                const uint32_t xcc = (uint32_t)__builtin_amdgcn_s_getreg((3 << 11) | 20) & 7u;     // HW_REG_XCC_ID: the XCD this workgroup runs on
                bool got = false;
                for (uint32_t t = 0; t < 8u && !got; t++) {
                    const uint32_t sgm = (xcc + t) & 7u;
                    uint32_t base = 0;
                    if (lane == 0u) base = atomicAdd(&p.travCounters[4u + 8u * (uint32_t)phase + sgm], p.raysPerWave);
                    base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
                    if (base < segLen) {
                        got = true;
                        const uint32_t gb = sgm * segLen + base, ge = (base + p.raysPerWave < segLen ? gb + p.raysPerWave : (sgm + 1u) * segLen);
                        next = gb; end = ge; chunkBase = gb;
                        const uint32_t* q = phase == 0 ? qExt : qSh;
                        qe0 = (gb + lane < ge) ? q[gb + lane] : kQueueHole;
                        qe1 = (gb + 64u + lane < ge) ? q[gb + 64u + lane] : kQueueHole;
                    }
                }
                if (!got) { phase++; next = end = 0; }
            }
#endif
            while (next >= end && phase < 2) { // next chunk of the current queue, or the first one of the next queue
                uint32_t base = 0;
                const uint32_t count = phase == 0 ? countExt : countSh;
                const uint32_t gridWaves = gridDim.x * (uint32_t)(kDefBlock / 64);
                const bool tail = phase == 1 && lastBase + 2u * gridWaves * p.raysPerWave > count;   // towards the end of the last queue the chunks shrink
                const bool tail2 = p.wideQuarterTail && phase == 1 && lastBase + gridWaves * p.raysPerWave / 2u > count;            // the last half round: quarter chunks (large tables only)
                const uint32_t req = tail2 ? (p.raysPerWave > 32u ? p.raysPerWave / 4u : p.raysPerWave) : tail ? (p.raysPerWave > 64u ? p.raysPerWave / 2u : p.raysPerWave) : p.raysPerWave;
                if (lane == 0u) base = atomicAdd(&p.travCounters[phase], req);
                base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
                if (base < count) {
                    lastBase = base;
                    next = base; end = (base + req < count) ? base + req : count;
                    chunkBase = base;
                    const uint32_t* q = phase == 0 ? qExt : qSh;
                    qe0 = (base + lane < end) ? q[base + lane] : kQueueHole;
                    qe1 = (base + 64u + lane < end) ? q[base + 64u + lane] : kQueueHole;
                }
                else { phase++; next = end = 0; lastBase = 0; }
            }
            const uint32_t my = next + prefix_rank(idleMask);
            const uint32_t entry = idle ? ((my - chunkBase) & 127u) : 0u;
            const uint32_t entryLo = (uint32_t)__shfl((int)qe0, (int)(entry & 63u)), entryHi = (uint32_t)__shfl((int)qe1, (int)(entry & 63u));
            if (idle) {
                // the next ray first: its loads are in flight while the finished ray is written back
                const bool take = my < end;
                const uint32_t newIndex = take ? (entry < 64u ? entryLo : entryHi) : kQueueHole;   // extensionRayCast.hlsl:210 / shadowRayCast.hlsl:159
                const bool newRay = take && ((phase != 0 && !GMUPT_WIDE_XCD_EXPERIMENT) || newIndex != kQueueHole); // holes only exist in the extension queue (and in the experiment's padded bins)
                f3 newO = mk3(0, 0, 0), newD = mk3(0, 0, 1); float newDist = kFltMax;
                if (newRay) {
                    if (phase == 0) { newO = ld3(p, F_RAY_OX, newIndex); newD = ld3(p, F_RAY_DX, newIndex); }                 // :213-214
                    else { newO = ld3(p, F_SH_OX, newIndex); newD = ld3(p, F_SH_DX, newIndex); newDist = ldf(p, F_LIGHT_DIST, newIndex); } // :162-164
                }
                if (haveRay) GMUPT_WIDE_FINISH();   // (decided: a ray that needs the exact walk was parked at the top of the loop and its lane is not idle)
                if (newRay) {
                    bottom = 0; donations = 0;
                    haveRay = true; kind = phase; index = newIndex;            // phase is 0 (extension) or 1 (shadow) here
                    if (STATS) { if (phase == 0) raysE++; else raysS++; }
                    distance = newDist;
                    ray_set(ray, newO, newD);
                    hitRef = -1; hu = 0.0f; hv = 0.0f; redo = false;
                    pa = 0; pb = S - 1; ti = -1;
                    cur = (ray_box(ts.rootMin[0], ts.rootMin[1], ts.rootMin[2], ts.rootMax[0], ts.rootMax[1], ts.rootMax[2], newO, ray_inv(ray)) > 0.0f) ? 0 : kDone;   // WNode 0 is the root's
                }
            }
            if (nIdle == 64 && phase == 2) break; // nothing in flight and both queues are exhausted (wave-uniform)
            next = (next + (uint32_t)nIdle < end) ? next + (uint32_t)nIdle : end;
        }

        // ---- REPS steps; a burst (wave-uniform, decided once per iteration) adds one triangle test per step to the lanes with leaves pending
        const bool pendingNow = (pb != (uint32_t)(S - 1)) || (ti >= 0);
        const bool roomNow = pb >= pa && pb - pa >= (uint32_t)(kWideRoom - 1);
        if (cur >= 0 && !roomNow && !pendingNow) {
            // the inner stack alone fills the LDS share of this lane: this ray's wide walk is given up, the exact walk (which spills) takes it
            redo = true; cur = kDone; pa = bottom;
        }
        const int nPending = __popcll(__ballot(pendingNow));
        const int nWalking = __popcll(__ballot(cur >= 0 && roomNow));
        const bool burst = nPending >= (int)p.tuneTriThresh || (nWalking == 0 && nPending > 0); // wave-uniform
        if (STATS) { // lane census: where do the 64 lanes of a wave spend the loop iterations?
            census0 += __popcll(__ballot(cur == kDone && !pendingNow)); census1 += __popcll(__ballot(cur >= 0 && roomNow));
            census2 += __popcll(__ballot(cur >= 0 && !roomNow)); census3 += __popcll(__ballot(cur == kDone && pendingNow));
        }
        // (wave-uniform, once per iteration: rays only change in the refill above) the ordered slab tree needs every walking ray's 1 / d finite
        const bool generalSlabs = !GMUPT_WIDE_SIGNED || __ballot(cur >= 0 && (ray_sg(ray) >> 24) != 0u) != 0ull;
        if (STATS && lane == 0u) { itersAll++; if (generalSlabs) itersGeneral++; }
#pragma unroll
        for (int rep = 0; rep < REPS; rep++) {
#if GMUPT_WIDE_SIGNED
#define GMUPT_WIDE_LOAD_LDS() load_wnode_lds_signed(s_top, pla, q0, q1, q2, q3, q4, q5, lk)
#define GMUPT_WIDE_LOAD_GLB() load_wnode_glb_signed(rNodes, pla, q0, q1, q2, q3, q4, q5, lk)
#else
#define GMUPT_WIDE_LOAD_LDS() load_wnode_lds(s_top, cur, q0, q1, q2, q3, q4, q5, lk)
#define GMUPT_WIDE_LOAD_GLB() load_wnode_glb(rNodes, cur, q0, q1, q2, q3, q4, q5, lk)
#endif
            // the four slab tests of the node in q0 .. lk, then: the first hit inner slot is the next node, the other hits are pushed (inner nodes from the
            // bottom, leaves from the top) -- ranks by prefix counts, one predicated LDS store per slot and stack (nested regions per slot: the same time)
#define GMUPT_WIDE_NODE_COMPUTE(INTOP) \
            { if (STATS) { if (kind == 0) tcE.inner++; else tcS.inner++; if (prefix_rank(__ballot(1)) == 0) { if (phase == 0) wInE++; else wInS++; } \
                           if (INTOP) { if (kind == 0) topE++; else topS++; } boxes += (uint32_t)ts.wnodes[cur].aux[1]; } \
              bool h0, h1, h2, h3; \
              slab_hits4(q0, q1, q2, q3, q4, q5, ray, h0, h1, h2, h3, generalSlabs); \
              const bool i0 = h0 & (lk.x >= 0), i1 = h1 & (lk.y >= 0), i2 = h2 & (lk.z >= 0), i3 = h3 & (lk.w >= 0); \
              const bool f0 = h0 & (lk.x < 0), f1 = h1 & (lk.y < 0), f2 = h2 & (lk.z < 0), f3_ = h3 & (lk.w < 0); \
              const uint32_t r1 = i0 ? 1u : 0u, r2 = r1 + (i1 ? 1u : 0u), r3 = r2 + (i2 ? 1u : 0u), cI = r3 + (i3 ? 1u : 0u);   /* inner hits before slot k */ \
              const uint32_t s1 = f0 ? 1u : 0u, s2 = s1 + (f1 ? 1u : 0u), s3 = s2 + (f2 ? 1u : 0u), cL = s3 + (f3_ ? 1u : 0u);  /* leaf hits before slot k */ \
              int nxt = i0 ? lk.x : (i1 ? lk.y : (i2 ? lk.z : (i3 ? lk.w : kDone)));                                            /* the first hit inner slot is the next node */ \
              if (i1 & (r1 != 0u)) sl[(pa + r1 - 1u) * kDefBlock] = lk.y; \
              if (i2 & (r2 != 0u)) sl[(pa + r2 - 1u) * kDefBlock] = lk.z; \
              if (i3 & (r3 != 0u)) sl[(pa + r3 - 1u) * kDefBlock] = lk.w; \
              if (f0) sl[pb * kDefBlock] = lk.x; \
              if (f1) sl[(pb - s1) * kDefBlock] = lk.y; \
              if (f2) sl[(pb - s2) * kDefBlock] = lk.z; \
              if (f3_) sl[(pb - s3) * kDefBlock] = lk.w; \
              pa += cI != 0u ? cI - 1u : 0u; pb -= cL; \
              if (STATS) { if (kind == 0) tcE.leaves += cL; else tcS.leaves += cL; } \
              if (nxt < 0) { if (pa > bottom) { pa--; nxt = sl[pa * kDefBlock]; } else { pa = 0; bottom = 0; } }   /* (an empty inner stack frees its dead entries) */ \
              cur = nxt; }
            vec4f q0, q1, q2, q3, q4, q5; vec4i lk;
            // fetch phase
            const bool doNode = cur >= 0 && pb >= pa && pb - pa >= (uint32_t)(kWideRoom - 1);
            const bool inTop = (uint32_t)cur < topCount;
            const PlaneAddr pla = plane_addresses(cur, ray_sg(ray)); (void)pla;    // (for every lane: the node-less ones load nothing)
            if (doNode && inTop) GMUPT_WIDE_LOAD_LDS();
            asm volatile("" ::: "memory");   // LDS lanes first, see load_node
            if (doNode && !inTop) GMUPT_WIDE_LOAD_GLB();
            if (burst && ti < 0 && pb != (uint32_t)(S - 1)) { pb++; ti = ~sl[pb * kDefBlock]; }
            const bool doTri = burst && ti >= 0;
            vec4f a0, a1, a2, a3, a4;                        // defined for the doTri lanes only: the TriPair of this step
            if (doTri) {
                const int off = ti * 80;
                a0 = __builtin_bit_cast(vec4f, __builtin_amdgcn_raw_buffer_load_b128(rPairs, off, 0, 0));
                a1 = __builtin_bit_cast(vec4f, __builtin_amdgcn_raw_buffer_load_b128(rPairs, off + 16, 0, 0));
                a2 = __builtin_bit_cast(vec4f, __builtin_amdgcn_raw_buffer_load_b128(rPairs, off + 32, 0, 0));
                a3 = __builtin_bit_cast(vec4f, __builtin_amdgcn_raw_buffer_load_b128(rPairs, off + 48, 0, 0));
                a4 = __builtin_bit_cast(vec4f, __builtin_amdgcn_raw_buffer_load_b128(rPairs, off + 64, 0, 0));
            }
            // compute phase
            if (doNode) GMUPT_WIDE_NODE_COMPUTE(inTop)
            if (doTri) {
                if (STATS) { pairFetches++; if (prefix_rank(__ballot(1)) == 0) { if (phase == 0) wTrE++; else wTrS++; } }
                const PairHit h = tri_pair_compute(a0, a1, a2, a3, a4, ray);
                bool last = h.last;
                if (STATS) { const uint32_t n = __builtin_bit_cast(vec4u, a4).w; if (kind == 0) tcE.tris += n; else tcS.tris += n; }   // references in this pair (1 or 2; 0 in the pair of an empty leaf)
                if (kind == 0) {
                    // extensionRayCast.hlsl:64-74 for the first, then for the second triangle (a closer hit ends a tie; equal t against another triangle is one)
#define GMUPT_WIDE_ACCEPT(OK, T, U, V, SLOT) \
                    if (OK) { if ((T) >= 0.0f && (T) < distance) { distance = (T); hitRef = (SLOT); hu = (U); hv = (V); redo = false; } \
                              else if ((T) == distance && hitRef >= 0) { if (tri_canon(rTris, (int)ts.pairRef[SLOT]) != tri_canon(rTris, (int)ts.pairRef[hitRef])) redo = true; } }
                    GMUPT_WIDE_ACCEPT(h.okA, h.t.x, h.u.x, h.v.x, 2 * ti)
                    GMUPT_WIDE_ACCEPT(h.okB, h.t.y, h.u.y, h.v.y, 2 * ti + 1)
#undef GMUPT_WIDE_ACCEPT
                } else {
                    // shadowRayCast.hlsl:41-45,89: t in (1e-8, 1e8) and |d t| < lightDistance => occluded: the ray is decided
                    const f3 d = ray_d(ray);
                    const bool occA = h.okA && h.t.x > kEpsilon && h.t.x < 1.0f / kEpsilon && length3(d * h.t.x) < distance;
                    const bool occB = h.okB && h.t.y > kEpsilon && h.t.y < 1.0f / kEpsilon && length3(d * h.t.y) < distance;
                    if (occA | occB) { hitRef = 2 * ti; last = true; pb = S - 1; cur = kDone; pa = bottom; redo = false; }
                }
                ti = last ? -1 : ti + 1;
            }
        }
    }
#undef GMUPT_WIDE_MERGE_OWN
#undef GMUPT_WIDE_FINISH
#undef GMUPT_WIDE_NODE_COMPUTE
#undef GMUPT_WIDE_LOAD_LDS
#undef GMUPT_WIDE_LOAD_GLB
    if (STATS) { flush_counts(p.stats, tcE, raysE, true); flush_wave_iters(p.stats, wInE, wTrE, true);
                 flush_counts(p.stats, tcS, raysS, false); flush_wave_iters(p.stats, wInS, wTrS, false);
                 flush_sum(&p.stats->extTopInner, topE); flush_sum(&p.stats->shTopInner, topS); flush_sum(&p.stats->castHelperSubtrees, helped);
                 flush_sum(&p.stats->castNestedHelpers, nested); flush_sum(&p.stats->wideBoxTests, boxes); flush_sum(&p.stats->wideIters, itersAll); flush_sum(&p.stats->wideGeneralIters, itersGeneral); flush_sum(&p.stats->widePairFetches, pairFetches);
                 if (lane == 0u) {
                     atomicAdd(&p.stats->castWaves, 1ull);
                     { const unsigned long long tEnd = wall_clock64(), life = tEnd - tStart;     // 100 MHz ticks
                       atomicAdd(&p.stats->castDrainClocks, tDrain ? tEnd - tDrain : 0ull); atomicAdd(&p.stats->castDrainIters, (unsigned long long)drainIters);
                       atomicAdd(&p.stats->castDrainBusyLanes, (unsigned long long)drainBusy);
                       atomicAdd(&p.stats->castWaveClocks, life); atomicMax(&p.stats->castWaveClocksMax, life);
                       atomicAdd(&p.stats->castWaveEndHist[life / 5000ull < 31ull ? life / 5000ull : 31ull], 1ull); }
                     atomicAdd(&p.stats->laneCensus[0], (unsigned long long)census0); atomicAdd(&p.stats->laneCensus[1], (unsigned long long)census1);
                     atomicAdd(&p.stats->laneCensus[2], (unsigned long long)census2); atomicAdd(&p.stats->laneCensus[3], (unsigned long long)census3);
                 } }
}

// ------------------------------------------------------------------------------------------------ host launchers
uint32_t traversal_wide_top_capacity() { return (uint32_t)kWideTop; }
uint32_t traversal_wide_stack_entries() { return (uint32_t)kWideStack; }
uint32_t traversal_wide_overflow_entries() { return (uint32_t)kWideOvf; }

// Both ray casts in one wide launch.  Returns 0 when this configuration is not taken (the caller runs another kernel).
uint32_t launch_cast_wide(const RenderParams& p, bool stats, hipStream_t s)
{
    if (!p.trav.wnodes || p.extendPrune || p.shadowPrune) return 0u;
    if ((uint64_t)p.trav.wideCount * 128ull >= (1ull << 31) || ((uint64_t)p.scene.numTris + 1ull) * 48ull >= (1ull << 31)) return 0u;
    const uint32_t pb = p.travGridBlocks;
    // steps per iteration of a wave (between two looks at the queues): six while the records fit the 256 MB Infinity Cache, eight beyond
    // (measured: config 3 0.923 / 0.930 ms with six / eight, config 5 5.12 / 5.00 ms); GMUPT_WIDE_STEPS overrides
    const uint64_t recordBytes = (uint64_t)p.trav.wideCount * 128ull + (uint64_t)p.trav.numPairs * 80ull;
    const bool eight = p.tuneWideSteps ? p.tuneWideSteps >= 8u : recordBytes > (256ull << 20);
    RenderParams q = p; q.wideQuarterTail = eight ? 1u : 0u;   // (measured with it: config 5 4.87-4.95 vs 4.95-4.99 ms, config 3 0.926-0.928 vs 0.920-0.923)
    if (stats) { if (eight) hipLaunchKernelGGL((k_cast_w<true, GMUPT_WIDE_REPS + 2>), dim3(pb), dim3(kDefBlock), 0, s, q); else hipLaunchKernelGGL((k_cast_w<true, GMUPT_WIDE_REPS>), dim3(pb), dim3(kDefBlock), 0, s, q); }
    else { if (eight) hipLaunchKernelGGL((k_cast_w<false, GMUPT_WIDE_REPS + 2>), dim3(pb), dim3(kDefBlock), 0, s, q); else hipLaunchKernelGGL((k_cast_w<false, GMUPT_WIDE_REPS>), dim3(pb), dim3(kDefBlock), 0, s, q); }
    return GMUPT_STAT_FUSED_CAST | GMUPT_STAT_CAST_WIDE;
}

} // namespace gmupt
