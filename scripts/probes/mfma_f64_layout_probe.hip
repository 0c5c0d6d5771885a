// Where v_mfma_f64_16x16x4f64 keeps its operands: A[i][k] in lane 16 k + i, B[k][j] in lane
// 16 k + j, D[i][j] in lane 16 (i % 4) + j, register i / 4 is the layout assumed by
// qeif_moment_mfma_kernel; this probe multiplies two matrices of distinct small integers and
// reports, for every lane and register, which entry of the product it holds, and times a chain
// of independent instructions.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef double double4_t __attribute__((ext_vector_type(4)));

__global__ void layout(double *out)
{
    const int l = threadIdx.x;
    const double a = 1.0 + (l & 15) + 16.0 * (l >> 4);    // A[i = l & 15][k = l >> 4] = 1 + i + 16 k
    const double b = 1.0 + 3.0 * (l & 15) + 7.0 * (l >> 4);  // B[k = l >> 4][j = l & 15] = 1 + 3 j + 7 k
    double4_t c = {0.0, 0.0, 0.0, 0.0};
    c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) out[4 * l + r] = c[r];
}

__global__ void chain(double *out, int reps)
{
    const int l = threadIdx.x;
    const double a = 1.0 + 1e-9 * l, b = 1.0 - 1e-9 * l;
    double4_t c[12];
    for (int t = 0; t < 12; ++t) c[t] = double4_t{0.0, 0.0, 0.0, 0.0};
    for (int i = 0; i < reps; ++i) {
#pragma unroll
        for (int t = 0; t < 12; ++t) c[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c[t], 0, 0, 0);
    }
    double s = 0.0;
    for (int t = 0; t < 12; ++t) s += c[t][0] + c[t][1] + c[t][2] + c[t][3];
    out[blockIdx.x * blockDim.x + l] = s;
}

int main()
{
    double *d;
    (void)hipMalloc(&d, 1 << 24);
    layout<<<1, 64>>>(d);
    std::vector<double> h(256);
    (void)hipMemcpy(h.data(), d, 256 * sizeof(double), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l)
        for (int r = 0; r < 4; ++r) {
            const int i = 4 * r + (l >> 4), j = l & 15;
            double want = 0.0;
            for (int k = 0; k < 4; ++k) want += (1.0 + i + 16.0 * k) * (1.0 + 3.0 * j + 7.0 * k);
            if (h[4 * l + r] != want) {
                if (bad < 8) printf("lane %d reg %d: got %g, D[%d][%d] would be %g\n", l, r, h[4 * l + r], i, j, want);
                ++bad;
            }
        }
    printf("layout D[i][j] -> lane 16 (i %% 4) + j, register i / 4: %s (%d mismatches)\n", bad ? "NO" : "yes", bad);
    // rate: 1024 workgroups of 256 threads (4 per SIMD), 12 independent accumulators per wavefront
    const int reps = 2000;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    chain<<<1024, 256>>>(d, 10);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    chain<<<1024, 256>>>(d, reps);
    (void)hipEventRecord(e1);
    (void)hipDeviceSynchronize();
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double n_inst = 1024.0 * 4 * reps * 12;
    printf("%.3f ms for %.0f instructions: %.1f TFLOP/s, %.1f ns per instruction and SIMD (4 wavefronts each)\n", ms,
           n_inst, n_inst * 2048 / (ms * 1e-3) / 1e12, ms * 1e6 / (4.0 * reps * 12));
    return bad != 0;
}
