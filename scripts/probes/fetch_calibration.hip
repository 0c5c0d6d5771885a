// Calibrates rocprofv3's FETCH_SIZE for the access shape of qeif_moment_kernel's accumulation
// pass: 4-byte loads per lane, a workgroup of 256 lanes reading runs of 2C (pixels) or 3C
// (pattern) consecutive floats out of rows of 2N / 3N floats, one row per problem.  Each kernel
// below reads a known number of bytes exactly once; FETCH_SIZE (KiB) * 1024 / bytes is the factor
// to divide the counter of the real kernel by.  Run under `rocprofv3 --pmc FETCH_SIZE`.
//   calib_rows<RUN>   the moment kernel's shape with runs of RUN floats (16 = pixels with C = 8,
//                     24 = pattern with C = 8, 32 = pixels with C = 16)
//   calib_wide        16 bytes per lane, fully coalesced (the shape the 0.5 rule is known for)
//   calib_lines       the line-cut kernel's shape: 16-byte pieces, chunks cut at the cache lines
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

// The access shape of qeif_moment_lines_kernel (float32, N a multiple of 4, 128-byte aligned arrays):
// chunks of 16 points cut where each row crosses a 128-byte line, 16-byte pieces, 8 lanes per pixel
// run and 12 per pattern run, pieces wholly past a chunk not loaded.  Reads uv and xyz once.
__global__ __launch_bounds__(256) void calib_lines(const float4 *__restrict__ uv, const float4 *__restrict__ xyz, int N,
                                                   float *__restrict__ out)
{
    constexpr int C = 16;
    const int tid = threadIdx.x, ku = tid % 8, ju = tid / 8, kx = tid % 16, jx = tid / 16;
    const size_t first = static_cast<size_t>(blockIdx.x) * 256;
    const float4 *ub = uv + first * 2 * N / 4, *xb = xyz + first * 3 * N / 4;
    const int u0 = static_cast<int>((first * N) & 15);
    const int p0u = 16 - ((u0 + N * ju) & 15), p0x = 16 - ((u0 + N * jx) & 15);
    const int chunks = 1 + (N - 1 + C - 1) / C;
    float acc = 0.f;
    for (int t = 0; t < chunks; ++t) {
        const int su = t == 0 ? 0 : p0u + C * (t - 1), lu = min(N, p0u + C * t) - su;
        const int sx = t == 0 ? 0 : p0x + C * (t - 1), lx = min(N, p0x + C * t) - sx;
        float4 a[8], b[16];
        const bool live_u = 4 * ku < 2 * lu, live_x = kx < 12 && 4 * kx < 3 * lx;
#pragma unroll
        for (int r = 0; r < 8; ++r) a[r] = live_u ? ub[(static_cast<size_t>(ju + 32 * r) * 2 * N + 2 * su) / 4 + ku] : float4{0, 0, 0, 0};
#pragma unroll
        for (int r = 0; r < 16; ++r) b[r] = live_x ? xb[(static_cast<size_t>(jx + 16 * r) * 3 * N + 3 * sx) / 4 + kx] : float4{0, 0, 0, 0};
#pragma unroll
        for (int r = 0; r < 8; ++r) acc += a[r].x + a[r].w;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc += b[r].x + b[r].w;
    }
    out[first + tid] = acc;
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
        evict(); hipLaunchKernelGGL(calib_lines, dim3(B / 256), dim3(256), 0, 0, reinterpret_cast<const float4 *>(uv), reinterpret_cast<const float4 *>(xyz), N, out);
        // and without eviction: what repeated launches over the same 143 MB see
        hipLaunchKernelGGL(calib_rows<16>, dim3(B / 256), dim3(256), 0, 0, uv, 2 * N, out);
    }
    CK(hipDeviceSynchronize());
    printf("bytes: uv %zu xyz %zu\n", uv_floats * 4, xyz_floats * 4);
    return 0;
}
