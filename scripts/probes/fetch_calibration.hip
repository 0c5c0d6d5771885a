// Calibrates rocprofv3's FETCH_SIZE for the access shape of qeif_moment_kernel's accumulation
// pass: 4-byte loads per lane, a workgroup of 256 lanes reading runs of 2C (pixels) or 3C
// (pattern) consecutive floats out of rows of 2N / 3N floats, one row per problem.  Each kernel
// below reads a known number of bytes exactly once; FETCH_SIZE (KiB) * 1024 / bytes is the factor
// to divide the counter of the real kernel by.  Run under `rocprofv3 --pmc FETCH_SIZE`.
//   calib_rows<RUN>   the moment kernel's shape with runs of RUN floats (16 = pixels with C = 8,
//                     24 = pattern with C = 8, 32 = pixels with C = 16)
//   calib_wide        16 bytes per lane, fully coalesced (the shape the 0.5 rule is known for)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int RUN>   // RUN floats of every row per pass, rows of `row` floats, 256 rows per workgroup
__global__ __launch_bounds__(256) void calib_rows(const float *__restrict__ src, int row, float *__restrict__ out)
{
    constexpr int LANES = RUN <= 16 ? 16 : 32;           // lanes that share one run (LU / LX of the kernel)
    constexpr int J = 256 / LANES;                        // rows a pass of one load instruction covers
    const int tid = threadIdx.x, k = tid % LANES, j = tid / LANES;
    const float *blk = src + static_cast<size_t>(blockIdx.x) * 256 * row;
    float acc = 0.f;
    for (int c0 = 0; c0 < row; c0 += RUN) {
        const int o = c0 + (k < RUN && c0 + k < row ? k : 0);
#pragma unroll
        for (int r = 0; r < LANES; ++r) acc += blk[static_cast<unsigned>(j + r * J) * row + o];
    }
    out[static_cast<size_t>(blockIdx.x) * 256 + tid] = acc;
}

__global__ __launch_bounds__(256) void calib_wide(const float4 *__restrict__ src, size_t n4, float *__restrict__ out)
{
    const size_t i = static_cast<size_t>(blockIdx.x) * 256 + threadIdx.x;
    float acc = 0.f;
    for (size_t q = i; q < n4; q += static_cast<size_t>(gridDim.x) * 256) {
        const float4 v = src[q];
        acc += v.x + v.y + v.z + v.w;
    }
    out[i] = acc;
}

int main()
{
    const int B = 1 << 18, N = 68;
    const size_t uv_floats = static_cast<size_t>(B) * 2 * N, xyz_floats = static_cast<size_t>(B) * 3 * N;
    float *uv, *xyz, *out, *flush;
    CK(hipMalloc(&uv, uv_floats * 4)); CK(hipMalloc(&xyz, xyz_floats * 4)); CK(hipMalloc(&out, size_t(B) * 4));
    const size_t flush_bytes = size_t(768) << 20;          // larger than the 256 MiB Infinity Cache
    CK(hipMalloc(&flush, flush_bytes));
    CK(hipMemset(uv, 0, uv_floats * 4)); CK(hipMemset(xyz, 0, xyz_floats * 4));
    auto evict = [&] { CK(hipMemset(flush, 1, flush_bytes)); CK(hipDeviceSynchronize()); };
    for (int rep = 0; rep < 3; ++rep) {
        evict(); hipLaunchKernelGGL(calib_rows<16>, dim3(B / 256), dim3(256), 0, 0, uv, 2 * N, out);
        evict(); hipLaunchKernelGGL(calib_rows<32>, dim3(B / 256), dim3(256), 0, 0, uv, 2 * N, out);
        evict(); hipLaunchKernelGGL(calib_rows<24>, dim3(B / 256), dim3(256), 0, 0, xyz, 3 * N, out);
        evict(); hipLaunchKernelGGL(calib_wide, dim3(1024), dim3(256), 0, 0, reinterpret_cast<const float4 *>(uv), uv_floats / 4, out);
        // and without eviction: what repeated launches over the same 143 MB see
        hipLaunchKernelGGL(calib_rows<16>, dim3(B / 256), dim3(256), 0, 0, uv, 2 * N, out);
    }
    CK(hipDeviceSynchronize());
    printf("bytes: uv %zu xyz %zu\n", uv_floats * 4, xyz_floats * 4);
    return 0;
}
