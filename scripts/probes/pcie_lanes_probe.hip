// How should the chunks of a host-pointer call be queued so that copies in and copies out use the
// bus at the same time?  126 MB in and 59 MB out per "call" (one million six-point problems), all
// host memory page-locked, a trivial kernel per chunk in place of the solver.
//   lanes:   L streams, each chunk does copy-in, kernel, copy-out on its lane's stream (what
//            pipelined_host_call does)
//   staged:  one stream for all copies in, one for all kernels, one for all copies out, chained by
//            events per chunk
// Build: hipcc --offload-arch=gfx950 -O3 -o pcie_lanes_probe pcie_lanes_probe.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void touch(const float *in, float *out, size_t n_out)
{
    const size_t i = static_cast<size_t>(blockIdx.x) * 256 + threadIdx.x;
    if (i < n_out) out[i] = in[i] + 1.f;
}

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main()
{
    const size_t B = 1 << 20, in_per = 120, out_per = 56;
    const size_t in_bytes = B * in_per, out_bytes = B * out_per;
    char *h_in, *h_out, *d_in, *d_out;
    CK(hipHostMalloc(&h_in, in_bytes)); CK(hipHostMalloc(&h_out, out_bytes));
    CK(hipMalloc(&d_in, in_bytes)); CK(hipMalloc(&d_out, out_bytes));
    memset(h_in, 1, in_bytes);
    for (size_t chunk_mb : {4, 8, 16}) {
        const size_t chunk = (chunk_mb << 20) / in_per;            // problems per chunk
        const size_t n = (B + chunk - 1) / chunk;
        for (int lanes : {1, 2, 4, 8}) {
            std::vector<hipStream_t> st(lanes);
            for (auto &s : st) CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
            double best = 1e9;
            for (int rep = 0; rep < 5; ++rep) {
                CK(hipDeviceSynchronize());
                const double t0 = now();
                for (size_t k = 0; k < n; ++k) {
                    const size_t first = k * chunk, cnt = std::min(chunk, B - first);
                    hipStream_t s = st[k % lanes];
                    CK(hipMemcpyAsync(d_in + first * in_per, h_in + first * in_per, cnt * in_per, hipMemcpyHostToDevice, s));
                    hipLaunchKernelGGL(touch, dim3((cnt * out_per / 4 + 255) / 256), dim3(256), 0, s,
                                       reinterpret_cast<const float *>(d_in + first * in_per), reinterpret_cast<float *>(d_out + first * out_per), cnt * out_per / 4);
                    CK(hipMemcpyAsync(h_out + first * out_per, d_out + first * out_per, cnt * out_per, hipMemcpyDeviceToHost, s));
                }
                for (auto &s : st) CK(hipStreamSynchronize(s));
                best = std::min(best, now() - t0);
            }
            printf("chunks of %2zu MB, %d lane stream(s):            %.2f ms  (%.1f GB/s both ways)\n", chunk_mb, lanes, best * 1e3,
                   (in_bytes + out_bytes) / best / 1e9);
            for (auto &s : st) CK(hipStreamDestroy(s));
        }
        {
            hipStream_t s_in, s_k, s_out;
            CK(hipStreamCreateWithFlags(&s_in, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s_k, hipStreamNonBlocking));
            CK(hipStreamCreateWithFlags(&s_out, hipStreamNonBlocking));
            std::vector<hipEvent_t> e_in(n), e_k(n);
            for (auto &e : e_in) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
            for (auto &e : e_k) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
            double best = 1e9;
            for (int rep = 0; rep < 5; ++rep) {
                CK(hipDeviceSynchronize());
                const double t0 = now();
                for (size_t k = 0; k < n; ++k) {
                    const size_t first = k * chunk, cnt = std::min(chunk, B - first);
                    CK(hipMemcpyAsync(d_in + first * in_per, h_in + first * in_per, cnt * in_per, hipMemcpyHostToDevice, s_in));
                    CK(hipEventRecord(e_in[k], s_in));
                    CK(hipStreamWaitEvent(s_k, e_in[k], 0));
                    hipLaunchKernelGGL(touch, dim3((cnt * out_per / 4 + 255) / 256), dim3(256), 0, s_k,
                                       reinterpret_cast<const float *>(d_in + first * in_per), reinterpret_cast<float *>(d_out + first * out_per), cnt * out_per / 4);
                    CK(hipEventRecord(e_k[k], s_k));
                    CK(hipStreamWaitEvent(s_out, e_k[k], 0));
                    CK(hipMemcpyAsync(h_out + first * out_per, d_out + first * out_per, cnt * out_per, hipMemcpyDeviceToHost, s_out));
                }
                CK(hipStreamSynchronize(s_out));
                best = std::min(best, now() - t0);
            }
            printf("chunks of %2zu MB, in / kernel / out streams:      %.2f ms  (%.1f GB/s both ways)\n", chunk_mb, best * 1e3,
                   (in_bytes + out_bytes) / best / 1e9);
        }
    }
    // the floor: everything in one copy each way, on two streams at once
    {
        hipStream_t a, b;
        CK(hipStreamCreateWithFlags(&a, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&b, hipStreamNonBlocking));
        double best = 1e9;
        for (int rep = 0; rep < 5; ++rep) {
            CK(hipDeviceSynchronize());
            const double t0 = now();
            CK(hipMemcpyAsync(d_in, h_in, in_bytes, hipMemcpyHostToDevice, a));
            CK(hipMemcpyAsync(h_out, d_out, out_bytes, hipMemcpyDeviceToHost, b));
            CK(hipStreamSynchronize(a)); CK(hipStreamSynchronize(b));
            best = std::min(best, now() - t0);
        }
        printf("one copy each way at the same time:              %.2f ms  (%.1f GB/s both ways)\n", best * 1e3, (in_bytes + out_bytes) / best / 1e9);
    }
    return 0;
}
