// How fast can every lane read its own rows?  Lane j of a wavefront owns problem j of its 64: a
// pixel row of 2 N floats and a pattern row of 3 N floats (N = 68: 544 and 816 bytes), read in
// 16-byte pieces, G pieces (= G / 8 cache lines) per row requested back to back before any is
// used, so that the pieces of one line are asked for together.  The loads are summed and nothing
// else is done: time against the bytes of both arrays.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

template <int G>
__global__ __launch_bounds__(256) void rows(const float4 *uv, const float4 *xyz, int n_uv, int n_xyz, long B, float *out)
{
    const long j = min(static_cast<long>(blockIdx.x) * 256 + threadIdx.x, B - 1);
    const float4 *pu = uv + j * n_uv, *px = xyz + j * n_xyz;
    float acc = 0.f;
    for (int g = 0; g < n_uv; g += G) {
        float4 q[G];
#pragma unroll
        for (int k = 0; k < G; ++k) q[k] = pu[min(g + k, n_uv - 1)];
#pragma unroll
        for (int k = 0; k < G; ++k) acc += q[k].x + q[k].y + q[k].z + q[k].w;
    }
    for (int g = 0; g < n_xyz; g += G) {
        float4 q[G];
#pragma unroll
        for (int k = 0; k < G; ++k) q[k] = px[min(g + k, n_xyz - 1)];
#pragma unroll
        for (int k = 0; k < G; ++k) acc += q[k].x + q[k].y + q[k].z + q[k].w;
    }
    if (acc == 12345.678f) out[j] = acc;
}

template <int G>
void run(const float4 *uv, const float4 *xyz, int N, long B, float *out)
{
    const int n_uv = 2 * N / 4, n_xyz = 3 * N / 4;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    for (int i = 0; i < 20; ++i) rows<G><<<(B + 255) / 256, 256>>>(uv, xyz, n_uv, n_xyz, B, out);
    (void)hipEventRecord(e0);
    for (int i = 0; i < 50; ++i) rows<G><<<(B + 255) / 256, 256>>>(uv, xyz, n_uv, n_xyz, B, out);
    (void)hipEventRecord(e1);
    (void)hipDeviceSynchronize();
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    ms /= 50;
    printf("N %d, %d pieces in flight per row: %.4f ms, %.2f TB/s\n", N, G, ms, B * 20.0 * N / (ms * 1e-3) / 1e12);
}

int main()
{
    const long B = 1 << 18;
    for (int N : {68, 64, 32}) {
        float4 *uv, *xyz;
        float *out;
        (void)hipMalloc(&uv, B * 8 * N);
        (void)hipMalloc(&xyz, B * 12 * N);
        (void)hipMalloc(&out, B * 4);
        (void)hipMemset(uv, 0, B * 8 * N);
        (void)hipMemset(xyz, 0, B * 12 * N);
        run<2>(uv, xyz, N, B, out);
        run<4>(uv, xyz, N, B, out);
        run<8>(uv, xyz, N, B, out);
        run<16>(uv, xyz, N, B, out);
        (void)hipFree(uv); (void)hipFree(xyz); (void)hipFree(out);
    }
    return 0;
}
