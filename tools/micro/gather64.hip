// Micro-benchmark: throughput of random 64-byte record gathers on MI355X, per access shape.
//  A: one lane fetches its whole record (4 x global_load_dwordx4 at 4 consecutive 16-B offsets)          -> 64 records / 4 instr
//  B: the 4 lanes of a quad fetch the 4 quarters of one record (1 x dwordx4 per lane and record)        -> 16 records / instr
//  C: as A but every lane of a quad reads the SAME record (coherent rays)                                -> 16 distinct records / 4 instr
//  D: one lane fetches a 48-byte record (3 x dwordx4)
//  E: as B, but through LDS-DMA (global_load_lds_dwordx4 with per-lane source addresses: instruction j brings the records of lanes 4g+j
//     of every quad g into a per-wave staging region) and each lane reads its own record back with 4 x ds_read_b128 -- no register shuffles
//  F: as A, 3 x dwordx4 + 1 x dwordx2 of a 64-byte record (what k_cast_f issues per node)
//  H: a lane fetches a whole 128-byte line (8 x 16 B): two records for one line request to the L2
//  I: a lane fetches the first half of a 128-byte line now and the second half one dependent step later (is the line still in the L1?)
//  G: as A with only `active` of the 64 lanes of every wave fetching (scattered over the wave): does a half-empty wave cost the memory
//     pipeline half as much?  (argv[2] = active lanes, argv[3] = dynamic LDS bytes per block to cap the occupancy, e.g. 40960 -> 16 waves per CU)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int MODE>
__global__ __launch_bounds__(256) void k(const float4* __restrict__ recs, const unsigned* __restrict__ idx, float* out, int iters, unsigned n, unsigned active)
{
    extern __shared__ char s_dyn[];
    if (iters < 0) s_dyn[threadIdx.x] = 0; // keeps the dynamic allocation alive
    const unsigned gtid = blockIdx.x * 256 + threadIdx.x;
    const unsigned lane = threadIdx.x & 63;
    float acc = 0.f;
    unsigned r = idx[gtid % n];
    for (int it = 0; it < iters; it++) {
        if (MODE == 0) {
            const float4* p = recs + 4ull * r;
            float4 a = p[0], b = p[1], c = p[2], d = p[3];
            acc += a.x + b.y + c.z + d.w;
            r = (__float_as_uint(d.x) + r * 1664525u + it) % n; // dependent chain like a traversal
        } else if (MODE == 1) {
            unsigned r0 = __shfl(r, (lane & ~3u) + 0), r1 = __shfl(r, (lane & ~3u) + 1), r2 = __shfl(r, (lane & ~3u) + 2), r3 = __shfl(r, (lane & ~3u) + 3);
            const unsigned q = lane & 3;
            float4 a = recs[4ull * r0 + q], b = recs[4ull * r1 + q], c = recs[4ull * r2 + q], d = recs[4ull * r3 + q];
            acc += a.x + b.y + c.z + d.w;
            float mine = (q == 0) ? a.x : (q == 1) ? b.x : (q == 2) ? c.x : d.x;
            r = (__float_as_uint(mine) + r * 1664525u + it) % n;
        } else if (MODE == 2) {
            unsigned rq = __shfl(r, lane & ~3u);
            const float4* p = recs + 4ull * rq;
            float4 a = p[0], b = p[1], c = p[2], d = p[3];
            acc += a.x + b.y + c.z + d.w;
            r = (__float_as_uint(d.x) + r * 1664525u + it) % n;
        } else if (MODE == 4) {
            typedef __attribute__((address_space(3))) void lds_void;
            typedef const __attribute__((address_space(1))) void glb_void;
            __shared__ __attribute__((aligned(16))) char s_stage[4][4 * (1024 + 64)];
            char* stage = s_stage[threadIdx.x >> 6];
            const char* base = reinterpret_cast<const char*>(recs) + 16u * (lane & 3u);
            unsigned r0 = __shfl(r, (lane & ~3u) + 0), r1 = __shfl(r, (lane & ~3u) + 1), r2 = __shfl(r, (lane & ~3u) + 2), r3 = __shfl(r, (lane & ~3u) + 3);
            __builtin_amdgcn_global_load_lds((glb_void*)(base + (size_t)r0 * 64u), (lds_void*)(stage + 0 * 1088), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((glb_void*)(base + (size_t)r1 * 64u), (lds_void*)(stage + 1 * 1088), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((glb_void*)(base + (size_t)r2 * 64u), (lds_void*)(stage + 2 * 1088), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((glb_void*)(base + (size_t)r3 * 64u), (lds_void*)(stage + 3 * 1088), 16, 0, 0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const float4* mine = reinterpret_cast<const float4*>(stage + (lane & 3u) * 1088 + (lane >> 2) * 64u);
            float4 a = mine[0], b = mine[1], c = mine[2], d = mine[3];
            acc += a.x + b.y + c.z + d.w;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // the next round overwrites the staging region
            r = (__float_as_uint(d.x) + r * 1664525u + it) % n;
        } else if (MODE == 6) {
            if (((lane * 37u + 11u) & 63u) < active) {   // a permutation of the lanes: exactly `active` of them, scattered
                const float4* p = recs + 4ull * r;
                float4 a = p[0], b = p[1], c = p[2], d = p[3];
                acc += a.x + b.y + c.z + d.w;
                r = (__float_as_uint(d.x) + r * 1664525u + it) % n;
            }
        } else if (MODE == 7) {
            const float4* p = recs + 8ull * (r >> 1);
            float4 a = p[0], b = p[1], c = p[2], d = p[3], e = p[4], f = p[5], g = p[6], h = p[7];
            acc += a.x + b.y + c.z + d.w + e.x + f.y + g.z + h.w;
            r = (__float_as_uint(d.x) + __float_as_uint(h.x) + r * 1664525u + it) % n;
        } else if (MODE == 8) {
            const float4* p = recs + 8ull * (r >> 1) + 4u * (it & 1);
            float4 a = p[0], b = p[1], c = p[2], d = p[3];
            acc += a.x + b.y + c.z + d.w;
            if (it & 1) r = (__float_as_uint(d.x) + r * 1664525u + it) % n; else r += (__float_as_uint(d.x) == 123.456f); // dependent, but the same line next step
        } else if (MODE == 5) {
            const float4* p = recs + 4ull * r;
            float4 a = p[0], b = p[1], c = p[2]; float2 d = *reinterpret_cast<const float2*>(p + 3);
            acc += a.x + b.y + c.z + d.y;
            r = (__float_as_uint(d.x) + r * 1664525u + it) % n;
        } else {
            const float4* p = recs + 3ull * r;
            float4 a = p[0], b = p[1], c = p[2];
            acc += a.x + b.y + c.z;
            r = (__float_as_uint(c.x) + r * 1664525u + it) % n;
        }
    }
    out[gtid] = acc;
}

int main(int argc, char** argv)
{
    const unsigned n = argc > 1 ? atoi(argv[1]) : 400000;   // records (x64 B)
    const unsigned active = argc > 2 ? atoi(argv[2]) : 32;
    const unsigned lds = argc > 3 ? atoi(argv[3]) : 0;
    const int iters = 64, blocks = 256 * 8, threads = blocks * 256;
    std::vector<float> h((size_t)n * 16);
    for (size_t i = 0; i < h.size(); i++) h[i] = (float)(rand() % 1000000);
    std::vector<unsigned> hi(threads);
    for (auto& v : hi) v = (unsigned)(((unsigned long long)rand() * 2147483648ull + (unsigned long long)rand()) % n);
    float4* recs; unsigned* idx; float* out;
    CHECK(hipMalloc(&recs, h.size() * 4)); CHECK(hipMalloc(&idx, hi.size() * 4)); CHECK(hipMalloc(&out, threads * 4));
    CHECK(hipMemcpy(recs, h.data(), h.size() * 4, hipMemcpyHostToDevice)); CHECK(hipMemcpy(idx, hi.data(), hi.size() * 4, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const char* names[9] = { "A lane=record 4x16B", "B quad=record 1x16B x4 recs", "C quad shares record", "D lane=48B record", "E quad=record via LDS-DMA", "F lane=record 3x16B+8B", "G as A, some lanes idle", "H lane=128B line 8x16B", "I lane=128B line in 2 steps" };
    printf("table: %u records of 64 B = %.1f MB\n", n, n * 64.0 / 1e6);
    for (int rep = 0; rep < 2; rep++)
        for (int m = 0; m < 9; m++) {
            CHECK(hipEventRecord(e0));
            if (m == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), lds, 0, recs, idx, out, iters, n, active);
            if (m == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), lds, 0, recs, idx, out, iters, n, active);
            if (m == 2) hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(256), lds, 0, recs, idx, out, iters, n, active);
            if (m == 4) hipLaunchKernelGGL(k<4>, dim3(blocks), dim3(256), lds, 0, recs, idx, out, iters, n, active);
            if (m == 5) hipLaunchKernelGGL(k<5>, dim3(blocks), dim3(256), lds, 0, recs, idx, out, iters, n, active);
            if (m == 6) hipLaunchKernelGGL(k<6>, dim3(blocks), dim3(256), lds, 0, recs, idx, out, iters, n, active);
            if (m == 7) hipLaunchKernelGGL(k<7>, dim3(blocks), dim3(256), lds, 0, recs, idx, out, iters, n, active);
            if (m == 8) hipLaunchKernelGGL(k<8>, dim3(blocks), dim3(256), lds, 0, recs, idx, out, iters, n, active);
            if (m == 3) hipLaunchKernelGGL(k<3>, dim3(blocks), dim3(256), lds, 0, recs, idx, out, iters, n / 4 * 4 / 3, active);
            CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
            const double recsFetched = (double)threads * iters * (m == 6 ? active / 64.0 : 1.0);   // H: 128-byte records (lines) per second; I: 64-byte halves per second
            const double lanereq = (m == 0 || m == 2 || m == 5) ? 4 : (m == 3 ? 3 : 4); // 16-byte (or 8-byte) lane requests per record
            if (rep) printf("%-30s %8.3f ms  %7.1f Grec/s  %7.1f GB/s useful  %.3f lane-requests/clk/CU @2.4GHz  (%.2f cycles/lane-record/CU)\n", names[m], ms, recsFetched / ms / 1e6,
                            recsFetched * (m == 3 ? 48 : 64) / ms / 1e6, recsFetched * lanereq / (ms * 1e-3 * 2.4e9 * 256), ms * 1e-3 * 2.4e9 * 256 / recsFetched);
        }
    return 0;
}
