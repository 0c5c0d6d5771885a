// What bandwidth the moment kernel's staging loads can reach by themselves: the same access
// shapes (per problem a short run of pixels and a short run of pattern coordinates per chunk,
// 4 bytes per lane), with no arithmetic, at the kernel's occupancy (two 256-thread workgroups per
// CU, forced by the dynamic LDS size), against longer runs and against a plain wide copy.
// Build: hipcc --offload-arch=gfx950 -O3 -o access_shape_probe access_shape_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// C points per chunk: per problem a run of 2C floats of uv and 3C floats of xyz.  LU = 2C lanes
// share a uv run, LX = 4C lanes (3C active) an xyz run, as in qeif_moment_kernel.  DEPTH chunks
// are requested before the first is waited for.
template <int C, bool WITH_XYZ, int DEPTH>
__global__ __launch_bounds__(256, 2) void rows(const float *__restrict__ uv, const float *__restrict__ xyz, int N,
                                                float *__restrict__ out)
{
    extern __shared__ float lds[];
    constexpr int LU = 2 * C, JU = 256 / LU, LX = 4 * C, JX = 256 / LX;
    const int tid = threadIdx.x, ku = tid % LU, ju = tid / LU, kx = tid % LX, jx = tid / LX;
    const float *uv_blk = uv + static_cast<size_t>(blockIdx.x) * 256 * 2 * N;
    const float *xyz_blk = xyz + static_cast<size_t>(blockIdx.x) * 256 * 3 * N;
    const unsigned row_u = 2u * N, row_x = 3u * N;
    float acc = 0.f;
    float a[DEPTH][LU], b[DEPTH][WITH_XYZ ? LX : 1];
    auto issue = [&](int c0, int d) {
        const int cn = min(C, N - c0);
        const unsigned ou = 2u * c0 + (ku < 2 * cn ? ku : 0), ox = 3u * c0 + (kx < 3 * cn ? kx : 0);
#pragma unroll
        for (int r = 0; r < LU; ++r) a[d][r] = uv_blk[static_cast<unsigned>(ju + r * JU) * row_u + ou];
        if constexpr (WITH_XYZ) {
#pragma unroll
            for (int r = 0; r < LX; ++r) b[d][r] = xyz_blk[static_cast<unsigned>(jx + r * JX) * row_x + ox];
        }
    };
    auto drain = [&](int d) {
#pragma unroll
        for (int r = 0; r < LU; ++r) acc += a[d][r];
        if constexpr (WITH_XYZ) {
#pragma unroll
            for (int r = 0; r < LX; ++r) acc += b[d][r];
        }
    };
    static_assert(DEPTH == 1 || DEPTH == 2, "");
    if constexpr (DEPTH == 1) {
        for (int c0 = 0; c0 < N; c0 += C) { issue(c0, 0); drain(0); }
    } else {
        issue(0, 0);
        for (int c0 = 0; c0 < N; c0 += 2 * C) {
            if (c0 + C < N) issue(c0 + C, 1);
            drain(0);
            if (c0 + 2 * C < N) issue(c0 + 2 * C, 0);
            if (c0 + C < N) drain(1);
        }
    }
    if (acc == 123.456f) lds[tid] = acc;
    out[static_cast<size_t>(blockIdx.x) * 256 + tid] = acc;
}

// Chunks of 16 points whose boundaries are shifted per row so that every pixel run is one whole
// 128-byte line: row j's chunk 0 is points [0, p0_j), chunk t points [p0_j + 16 (t - 1), p0_j + 16 t),
// with p0_j = 16 - ((N j) mod 16) (the buffers are 128-byte aligned here).  The pattern runs cover
// the same points (192 bytes, at multiples of 64 bytes when N is a multiple of 4).
__global__ __launch_bounds__(256, 2) void rows_shifted(const float *__restrict__ uv, const float *__restrict__ xyz, int N,
                                                        float *__restrict__ out)
{
    extern __shared__ float lds[];
    constexpr int C = 16, LU = 2 * C, JU = 256 / LU, LX = 4 * C, JX = 256 / LX;
    const int tid = threadIdx.x, ku = tid % LU, ju = tid / LU, kx = tid % LX, jx = tid / LX;
    const size_t first = static_cast<size_t>(blockIdx.x) * 256;
    const float *uv_blk = uv + first * 2 * N;
    const float *xyz_blk = xyz + first * 3 * N;
    const unsigned row_u = 2u * N, row_x = 3u * N;
    const int phase = static_cast<int>((first * N) & 15);
    float acc = 0.f;
    const int T = 1 + (N - 1 + 15) / 16;
    // N is a multiple of 4 here: the rows one thread loads (ju + 8 r, jx + 4 r) then share one shift
    const int p0u = 16 - ((phase + N * ju) & 15), p0x = 16 - ((phase + N * jx) & 15);
    for (int t = 0; t < T; ++t) {
        float a[LU], b[LX];
        const int su = t == 0 ? 0 : p0u + 16 * (t - 1), lu = max(0, min(N, p0u + 16 * t) - su);
        const int sx = t == 0 ? 0 : p0x + 16 * (t - 1), lx = max(0, min(N, p0x + 16 * t) - sx);
        const unsigned ou = 2u * min(su, N - 1) + (ku < 2 * lu ? ku : 0), ox = 3u * min(sx, N - 1) + (kx < 3 * lx ? kx : 0);
#pragma unroll
        for (int r = 0; r < LU; ++r) a[r] = uv_blk[static_cast<unsigned>(ju + r * JU) * row_u + ou];
#pragma unroll
        for (int r = 0; r < LX; ++r) b[r] = xyz_blk[static_cast<unsigned>(jx + r * JX) * row_x + ox];
#pragma unroll
        for (int r = 0; r < LU; ++r) acc += a[r];
#pragma unroll
        for (int r = 0; r < LX; ++r) acc += b[r];
    }
    if (acc == 123.456f) lds[tid] = acc;
    out[first + tid] = acc;
}

// Same rows, 16 bytes per lane: a uv run of 2C floats is C/2 lanes wide, an xyz run 3C/4 lanes.
template <int C>
__global__ __launch_bounds__(256, 2) void rows_wide(const float4 *__restrict__ uv, const float4 *__restrict__ xyz, int N,
                                                     float *__restrict__ out)
{
    extern __shared__ float lds[];
    constexpr int LU = C / 2, JU = 256 / LU, RU = 256 / JU;           // lanes per uv run, rows per pass, passes
    constexpr int LX = C, JX = 256 / LX, RX = 256 / JX;              // C lanes (3C/4 active) per xyz run
    const int tid = threadIdx.x, ku = tid % LU, ju = tid / LU, kx = tid % LX, jx = tid / LX;
    const float4 *uv_blk = uv + static_cast<size_t>(blockIdx.x) * 256 * 2 * N / 4;
    const float4 *xyz_blk = xyz + static_cast<size_t>(blockIdx.x) * 256 * 3 * N / 4;
    const unsigned row_u = 2u * N / 4, row_x = 3u * N / 4;
    float acc = 0.f;
    for (int c0 = 0; c0 < N; c0 += C) {
        const int cn = min(C, N - c0);
        const unsigned ou = 2u * c0 / 4 + (ku < 2 * cn / 4 ? ku : 0), ox = 3u * c0 / 4 + (kx < 3 * cn / 4 ? kx : 0);
        float4 a[RU], b[RX];
#pragma unroll
        for (int r = 0; r < RU; ++r) a[r] = uv_blk[static_cast<unsigned>(ju + r * JU) * row_u + ou];
#pragma unroll
        for (int r = 0; r < RX; ++r) b[r] = xyz_blk[static_cast<unsigned>(jx + r * JX) * row_x + ox];
#pragma unroll
        for (int r = 0; r < RU; ++r) acc += a[r].x + a[r].y + a[r].z + a[r].w;
#pragma unroll
        for (int r = 0; r < RX; ++r) acc += b[r].x + b[r].y + b[r].z + b[r].w;
    }
    if (acc == 123.456f) lds[tid] = acc;
    out[static_cast<size_t>(blockIdx.x) * 256 + tid] = acc;
}

// The block's whole input as two contiguous spans, 16 bytes per lane.
__global__ __launch_bounds__(256, 2) void spans(const float4 *__restrict__ uv, const float4 *__restrict__ xyz, int N,
                                                 float *__restrict__ out)
{
    extern __shared__ float lds[];
    const int tid = threadIdx.x;
    const float4 *u = uv + static_cast<size_t>(blockIdx.x) * 256 * 2 * N / 4;
    const float4 *x = xyz + static_cast<size_t>(blockIdx.x) * 256 * 3 * N / 4;
    const int nu = 256 * 2 * N / 4, nx = 256 * 3 * N / 4;
    float acc = 0.f;
    for (int i = tid; i < nu; i += 1024) {
        float4 v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = u[min(i + 256 * k, nu - 1)];
#pragma unroll
        for (int k = 0; k < 4; ++k) acc += v[k].x + v[k].y + v[k].z + v[k].w;
    }
    for (int i = tid; i < nx; i += 1024) {
        float4 v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = x[min(i + 256 * k, nx - 1)];
#pragma unroll
        for (int k = 0; k < 4; ++k) acc += v[k].x + v[k].y + v[k].z + v[k].w;
    }
    if (acc == 123.456f) lds[tid] = acc;
    out[static_cast<size_t>(blockIdx.x) * 256 + tid] = acc;
}

int main()
{
    const int B = 1 << 18, N = 68, LDS = 80 * 1024;
    const size_t uv_floats = static_cast<size_t>(B) * 2 * N, xyz_floats = static_cast<size_t>(B) * 3 * N;
    float *uv, *xyz, *out;
    CK(hipMalloc(&uv, uv_floats * 4)); CK(hipMalloc(&xyz, xyz_floats * 4)); CK(hipMalloc(&out, size_t(B) * 4));
    CK(hipMemset(uv, 0, uv_floats * 4)); CK(hipMemset(xyz, 0, xyz_floats * 4));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const double both = (uv_floats + xyz_floats) * 4.0, only_uv = uv_floats * 4.0;
    auto report = [&](const char *name, double bytes, auto launch) {
        for (int i = 0; i < 5; ++i) launch();
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        for (int i = 0; i < 20; ++i) launch();
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        CK(hipGetLastError());
        printf("%-44s %8.4f ms  %6.2f TB/s\n", name, ms / 20, bytes / (ms / 20 * 1e-3) / 1e12);
    };
#define ATTR(k) CK(hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, LDS))
    ATTR((rows<8, true, 1>)); ATTR((rows<8, true, 2>)); ATTR((rows<16, true, 1>)); ATTR((rows<16, false, 1>));
    ATTR((rows<16, false, 2>)); ATTR((rows<8, false, 1>)); ATTR((rows_wide<8>)); ATTR((rows_wide<16>)); ATTR(spans);
    const dim3 g(B / 256), b(256);
    const float4 *uv4 = reinterpret_cast<const float4 *>(uv), *xyz4 = reinterpret_cast<const float4 *>(xyz);
    report("uv+xyz, 64/96-byte runs, 1 chunk in flight", both, [&] { hipLaunchKernelGGL((rows<8, true, 1>), g, b, LDS, 0, uv, xyz, N, out); });
    report("uv+xyz, 64/96-byte runs, 2 chunks in flight", both, [&] { hipLaunchKernelGGL((rows<8, true, 2>), g, b, LDS, 0, uv, xyz, N, out); });
    report("uv+xyz, 128/192-byte runs, 1 chunk in flight", both, [&] { hipLaunchKernelGGL((rows<16, true, 1>), g, b, LDS, 0, uv, xyz, N, out); });
    report("uv+xyz, 64/96-byte runs, 16 B per lane", both, [&] { hipLaunchKernelGGL((rows_wide<8>), g, b, LDS, 0, uv4, xyz4, N, out); });
    report("uv+xyz, 128/192-byte runs, 16 B per lane", both, [&] { hipLaunchKernelGGL((rows_wide<16>), g, b, LDS, 0, uv4, xyz4, N, out); });
    report("uv+xyz, contiguous spans, 16 B per lane", both, [&] { hipLaunchKernelGGL(spans, g, b, LDS, 0, uv4, xyz4, N, out); });
    report("uv only, 64-byte runs, 1 chunk in flight", only_uv, [&] { hipLaunchKernelGGL((rows<8, false, 1>), g, b, LDS, 0, uv, xyz, N, out); });
    report("uv only, 128-byte runs, 1 chunk in flight", only_uv, [&] { hipLaunchKernelGGL((rows<16, false, 1>), g, b, LDS, 0, uv, xyz, N, out); });
    report("uv only, 128-byte runs, 2 chunks in flight", only_uv, [&] { hipLaunchKernelGGL((rows<16, false, 2>), g, b, LDS, 0, uv, xyz, N, out); });
    // Is it the partial cache lines?  N = 64 makes the rows 512 and 768 bytes (whole 128-byte lines), and
    // 32 points per chunk makes every run whole lines; N = 68 with the same chunk has rows that
    // start at any multiple of 32 (16) bytes.
    ATTR((rows<32, true, 1>));
    const double both64 = static_cast<double>(B) * 64 * 20;
    report("N=64: uv+xyz, 256/384-byte runs, line-aligned", both64, [&] { hipLaunchKernelGGL((rows<32, true, 1>), g, b, LDS, 0, uv, xyz, 64, out); });
    report("N=68: uv+xyz, 256/384-byte runs", both, [&] { hipLaunchKernelGGL((rows<32, true, 1>), g, b, LDS, 0, uv, xyz, N, out); });
    report("N=64: uv+xyz, 64/96-byte runs, half lines", both64, [&] { hipLaunchKernelGGL((rows<8, true, 1>), g, b, LDS, 0, uv, xyz, 64, out); });
    report("N=64: uv+xyz, 128/192-byte runs", both64, [&] { hipLaunchKernelGGL((rows<16, true, 1>), g, b, LDS, 0, uv, xyz, 64, out); });
    ATTR(rows_shifted);
    for (int kb : {64, 79, 80, 81}) {
        int nb = -1;
        CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, reinterpret_cast<const void *>(rows_shifted), 256, kb * 1024));
        printf("workgroups per CU with %d KB of LDS each: %d\n", kb, nb);
    }
    report("N=68: 16-point chunks shifted to whole lines", both, [&] { hipLaunchKernelGGL(rows_shifted, g, b, LDS, 0, uv, xyz, N, out); });
    return 0;
}
