// Do float64 vector arithmetic and an HBM stream slow each other down when they share the chip?
// Kernel A: pure v_fma_f64 work, one wavefront per SIMD, eight independent chains per lane (no
// memory access).  Kernel B: a streaming copy that saturates HBM.  Timed alone and together on
// two streams; also A with the chip otherwise idle but B's traffic at a quarter of the rate.
// Diagnostic only.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ __launch_bounds__(256) void fma_kernel(double *out, int iters, double seed)
{
    double a[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) a[k] = seed + threadIdx.x * 1e-9 + k;
    const double m = 1.0000001, c = 1e-9;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < 8; ++k) a[k] = __builtin_fma(a[k], m, c);
    }
    double s = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) s += a[k];
    if (s == 12345.0) out[blockIdx.x * 256 + threadIdx.x] = s;
}

__global__ __launch_bounds__(256) void copy_kernel(const float4 *__restrict__ src, float4 *__restrict__ dst, size_t n4)
{
    for (size_t i = static_cast<size_t>(blockIdx.x) * 256 + threadIdx.x; i < n4; i += static_cast<size_t>(gridDim.x) * 256)
        dst[i] = src[i];
}

static float timed(hipStream_t s1, hipStream_t s2, bool a, bool b, double *out, int iters, const float4 *src, float4 *dst,
                   size_t n4, int copy_blocks, float *tb)
{
    hipEvent_t a0, a1, b0, b1;
    CK(hipEventCreate(&a0)); CK(hipEventCreate(&a1)); CK(hipEventCreate(&b0)); CK(hipEventCreate(&b1));
    CK(hipDeviceSynchronize());
    if (b) { CK(hipEventRecord(b0, s2)); for (int r = 0; r < 4; ++r) hipLaunchKernelGGL(copy_kernel, dim3(copy_blocks), dim3(256), 0, s2, src, dst, n4); CK(hipEventRecord(b1, s2)); }
    if (a) { CK(hipEventRecord(a0, s1)); for (int r = 0; r < 4; ++r) hipLaunchKernelGGL(fma_kernel, dim3(256 * 4), dim3(64), 0, s1, out, iters, 1.0); CK(hipEventRecord(a1, s1)); }
    CK(hipDeviceSynchronize());
    float ta = 0; *tb = 0;
    if (a) CK(hipEventElapsedTime(&ta, a0, a1));
    if (b) CK(hipEventElapsedTime(tb, b0, b1));
    return ta;
}

int main()
{
    const size_t bytes = size_t(1) << 30, n4 = bytes / 16;
    float4 *src, *dst; double *out;
    CK(hipMalloc(&src, bytes)); CK(hipMalloc(&dst, bytes)); CK(hipMalloc(&out, 1 << 22));
    CK(hipMemset(src, 1, bytes));
    hipStream_t s1, s2; CK(hipStreamCreate(&s1)); CK(hipStreamCreate(&s2));
    const int iters = 40000;   // 8 * 40000 fma per lane, one wavefront per SIMD: ~0.5 ms
    float tb;
    for (int warm = 0; warm < 3; ++warm) timed(s1, s2, true, true, out, iters, src, dst, n4, 4096, &tb);
    for (int rep = 0; rep < 3; ++rep) {
        float ta = timed(s1, s2, true, false, out, iters, src, dst, n4, 4096, &tb);
        printf("A alone: %.3f ms (4 launches) -> %.1f TFLOP/s fp64\n", ta, 4.0 * 1024 * 64 * 8.0 * iters * 2 / (ta * 1e-3) / 1e12);
        timed(s1, s2, false, true, out, iters, src, dst, n4, 4096, &tb);
        printf("B alone: %.3f ms (4 copies of 1 GiB) -> %.2f TB/s read+write\n", tb, 4.0 * 2 * bytes / (tb * 1e-3) / 1e12);
        ta = timed(s1, s2, true, true, out, iters, src, dst, n4, 4096, &tb);
        printf("A with B: A %.3f ms, B %.3f ms\n", ta, tb);
        ta = timed(s1, s2, true, true, out, iters, src, dst, n4, 256, &tb);
        printf("A with a thin B (256 workgroups): A %.3f ms, B %.3f ms\n", ta, tb);
    }
    return 0;
}
