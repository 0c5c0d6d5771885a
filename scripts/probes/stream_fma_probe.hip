// Does float64 arithmetic overlap with an HBM stream inside one wavefront?  Every wavefront repeats:
// issue 24 loads of 16 bytes per lane (a contiguous 24 KB slice of a large buffer), run F dependent-
// free fused multiply-adds per lane on registers, then wait for the loads and fold them in.  Timed
// with the loads only (F = 0), the arithmetic only (loads from one cached line), and both, at two
// 256-thread workgroups per CU (register-limited, like the moment kernel).
// Build: hipcc --offload-arch=gfx950 -O3 -o stream_fma_probe stream_fma_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <bool CACHED, bool HEAVY>
__global__ __launch_bounds__(256, 2) void probe(const float4 *__restrict__ src, size_t per_block4, int rounds, int fmas,
                                                double *__restrict__ out)
{
    const int tid = threadIdx.x;
    const float4 *blk = src + static_cast<size_t>(blockIdx.x) * per_block4;
    double a[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) a[k] = 1.0 + tid * 1e-9 + k;
    const double m = 1.0000001, c = 1e-9;
    double b[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) b[k] = 1e-3 * (tid + k);
    asm volatile("" : "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]), "+v"(b[4]), "+v"(b[5]), "+v"(b[6]), "+v"(b[7]));
    asm volatile("" : "+v"(b[8]), "+v"(b[9]), "+v"(b[10]), "+v"(b[11]), "+v"(b[12]), "+v"(b[13]), "+v"(b[14]), "+v"(b[15]));
    float4 land[24];
    float acc = 0.f;
    size_t pos = tid;
#pragma unroll
    for (int r = 0; r < 24; ++r) land[r] = blk[CACHED ? tid : pos + 256 * r];
    for (int it = 0; it < rounds; ++it) {
        // fold the landed data in, then request the next slice and compute while it travels
#pragma unroll
        for (int r = 0; r < 24; ++r) acc += land[r].x + land[r].w;
        pos += 256 * 24;
        if (it + 1 < rounds) {
#pragma unroll
            for (int r = 0; r < 24; ++r) land[r] = blk[CACHED ? tid : pos + 256 * r];
        }
        for (int i = 0; i < fmas; ++i) {
            if constexpr (HEAVY) {
                // three vector-register operands per instruction, as in sums of products of per-lane values
#pragma unroll
                for (int k = 0; k < 16; ++k) a[k] = __builtin_fma(b[k], b[(k + 5) & 15], a[k]);
            } else {
#pragma unroll
                for (int k = 0; k < 16; ++k) a[k] = __builtin_fma(a[k], m, c);
            }
        }
    }
    double s = acc;
#pragma unroll
    for (int k = 0; k < 16; ++k) s += a[k];
    out[static_cast<size_t>(blockIdx.x) * 256 + tid] = s;
}

int main()
{
    const int blocks = 1024, rounds = 14;                       // 4 workgroups per CU in turn, two generations
    const size_t per_block4 = static_cast<size_t>(256) * 24 * rounds;   // float4 per workgroup: 1.3 MB
    const size_t bytes = per_block4 * 16 * blocks;              // 1.4 GB
    float4 *src; double *out;
    CK(hipMalloc(&src, bytes)); CK(hipMalloc(&out, static_cast<size_t>(blocks) * 256 * 8));
    CK(hipMemset(src, 0, bytes));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    bool heavy = false;
    auto run = [&](const char *name, bool cached, int fmas) {
        auto launch = [&] {
            if (heavy) {
                if (cached) hipLaunchKernelGGL((probe<true, true>), dim3(blocks), dim3(256), 0, 0, src, per_block4, rounds, fmas, out);
                else hipLaunchKernelGGL((probe<false, true>), dim3(blocks), dim3(256), 0, 0, src, per_block4, rounds, fmas, out);
            } else {
                if (cached) hipLaunchKernelGGL((probe<true, false>), dim3(blocks), dim3(256), 0, 0, src, per_block4, rounds, fmas, out);
                else hipLaunchKernelGGL((probe<false, false>), dim3(blocks), dim3(256), 0, 0, src, per_block4, rounds, fmas, out);
            }
        };
        for (int i = 0; i < 3; ++i) launch();
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        for (int i = 0; i < 10; ++i) launch();
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 10;
        printf("%-34s fma/round %4d: %8.4f ms  stream %5.2f TB/s  fp64 %5.1f TFLOP/s\n", name, 16 * fmas, ms,
               cached ? 0.0 : bytes / (ms * 1e-3) / 1e12, 2.0 * 16 * fmas * rounds * blocks * 256 / (ms * 1e-3) / 1e12);
    };
    for (int pass = 0; pass < 2; ++pass) {
        heavy = pass == 1;
        printf(heavy ? "-- fused multiply-adds with three vector-register operands\n" : "-- fused multiply-adds with one vector-register operand\n");
        for (int fmas : {0, 20, 40, 80, 160}) {
            run("stream only / stream + arithmetic", false, fmas);
            if (fmas) run("arithmetic only (cached loads)", true, fmas);
        }
    }
    return 0;
}
