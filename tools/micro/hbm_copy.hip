// Sustained HBM bandwidth of this MI355X: float4 copy and triad over buffers far larger than the 256 MiB Infinity Cache.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
__global__ void copy4(const float4* __restrict__ a, float4* __restrict__ b, size_t n) { for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) b[i] = a[i]; }
__global__ void triad4(const float4* __restrict__ a, const float4* __restrict__ b, float4* __restrict__ c, size_t n)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { float4 x = a[i], y = b[i]; c[i] = make_float4(x.x + 2.f * y.x, x.y + 2.f * y.y, x.z + 2.f * y.z, x.w + 2.f * y.w); }
}
// read-only stream (what the ray cast of the 10 M-triangle scene mostly is): four independent 16-byte loads per lane and step
__global__ void read4(const float4* __restrict__ a, float* __restrict__ out, size_t n)
{
    float acc = 0.f;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    for (; i + 3 * stride < n; i += 4 * stride) {
        const float4 x0 = a[i], x1 = a[i + stride], x2 = a[i + 2 * stride], x3 = a[i + 3 * stride];
        acc += x0.x + x1.y + x2.z + x3.w;
    }
    for (; i < n; i += stride) acc += a[i].x;
    if (acc == 12345.678f) out[0] = acc;   // keeps the loads alive
}
__global__ void copy4x4(const float4* __restrict__ a, float4* __restrict__ b, size_t n)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    for (; i + 3 * stride < n; i += 4 * stride) {
        const float4 x0 = a[i], x1 = a[i + stride], x2 = a[i + 2 * stride], x3 = a[i + 3 * stride];
        b[i] = x0; b[i + stride] = x1; b[i + 2 * stride] = x2; b[i + 3 * stride] = x3;
    }
    for (; i < n; i += stride) b[i] = a[i];
}
int main()
{
    const size_t bytes = 2ull << 30, n = bytes / 16;
    float4 *a, *b, *c; CHECK(hipMalloc(&a, bytes)); CHECK(hipMalloc(&b, bytes)); CHECK(hipMalloc(&c, bytes));
    CHECK(hipMemset(a, 1, bytes)); CHECK(hipMemset(b, 2, bytes));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int rep = 0; rep < 3; rep++) {
        float ms;
        CHECK(hipEventRecord(e0)); hipLaunchKernelGGL(copy4, dim3(256 * 8), dim3(256), 0, 0, a, b, n); CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        CHECK(hipEventElapsedTime(&ms, e0, e1)); if (rep) printf("copy  float4: %.3f ms  %.2f TB/s (read + write)\n", ms, 2.0 * bytes / ms / 1e9);
        for (int blocks = 256 * 4; blocks <= 256 * 32; blocks *= 2) {
            CHECK(hipEventRecord(e0)); hipLaunchKernelGGL(read4, dim3(blocks), dim3(256), 0, 0, a, (float*)c, n); CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
            CHECK(hipEventElapsedTime(&ms, e0, e1)); if (rep == 2) printf("read  float4 x4, %5d blocks: %.3f ms  %.2f TB/s (read only)\n", blocks, ms, 1.0 * bytes / ms / 1e9);
            CHECK(hipEventRecord(e0)); hipLaunchKernelGGL(copy4x4, dim3(blocks), dim3(256), 0, 0, a, b, n); CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
            CHECK(hipEventElapsedTime(&ms, e0, e1)); if (rep == 2) printf("copy  float4 x4, %5d blocks: %.3f ms  %.2f TB/s (read + write)\n", blocks, ms, 2.0 * bytes / ms / 1e9);
        }
        CHECK(hipEventRecord(e0)); hipLaunchKernelGGL(triad4, dim3(256 * 8), dim3(256), 0, 0, a, b, c, n); CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        CHECK(hipEventElapsedTime(&ms, e0, e1)); if (rep) printf("triad float4: %.3f ms  %.2f TB/s (2 reads + write)\n", ms, 3.0 * bytes / ms / 1e9);
    }
    return 0;
}
