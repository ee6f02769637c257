// Probe: raw buffer loads through __builtin_amdgcn_make_buffer_rsrc on gfx950 (values, out-of-range behaviour).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
__global__ void k(const float* p, unsigned bytes, float* out, const int* idx, int flags)
{
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, (int)bytes, 0x00020000);
    const int i = idx[threadIdx.x];
    u32x4 a = __builtin_amdgcn_raw_buffer_load_b128(r, i * 48, 0, 0);
    u32x4 b = __builtin_amdgcn_raw_buffer_load_b128(r, i * 48 + 16, 0, 0);
    u32x2 c = __builtin_amdgcn_raw_buffer_load_b64(r, i * 48 + 32, 0, 0);
    typedef float f32x4 __attribute__((ext_vector_type(4))); typedef float f32x2 __attribute__((ext_vector_type(2)));
    // whole-vector bit casts: __builtin_bit_cast(float, a.y) on a vector element reads element 0 with this compiler
    const f32x4 fa = __builtin_bit_cast(f32x4, a), fb = __builtin_bit_cast(f32x4, b); const f32x2 fc = __builtin_bit_cast(f32x2, c);
    float* o = out + threadIdx.x * 10;
    o[0] = fa.x; o[1] = fa.y; o[2] = fa.z; o[3] = fa.w; o[4] = fb.x; o[5] = fb.y; o[6] = fb.z; o[7] = fb.w; o[8] = fc.x; o[9] = fc.y;
}
int main()
{
    const int n = 35;
    std::vector<float> h(n * 12); for (size_t i = 0; i < h.size(); i++) h[i] = (float)i;
    std::vector<int> hi(64); for (int i = 0; i < 64; i++) hi[i] = i % 40; // 35..39 are out of range
    float *d, *o; int* di;
    hipMalloc(&d, h.size() * 4); hipMalloc(&o, 64 * 10 * 4); hipMalloc(&di, 64 * 4);
    hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice); hipMemcpy(di, hi.data(), 64 * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, (unsigned)(h.size() * 4), o, di, 0);
    std::vector<float> ho(640); hipMemcpy(ho.data(), o, 640 * 4, hipMemcpyDeviceToHost);
    for (int t : {0, 1, 34, 35, 39}) { printf("lane %d:", t); for (int k2 = 0; k2 < 10; k2++) printf(" %g", ho[t * 10 + k2]); printf("\n"); }
    return 0;
}
