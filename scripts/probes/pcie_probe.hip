// Host <-> device copy rates that bound the host-pointer entry points: pageable and pinned
// copies in both directions, alone and at the same time from two threads, the cost of pinning a
// caller's buffer in place, and the rate of host memcpy into a pinned buffer with 1..8 threads.
// Sizes are those of a 1M-problem six-point call: 126 MB in, 59 MB out.  Diagnostic only.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cstdint>
#include <thread>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    const size_t in = 126u << 20, out = 59u << 20;
    char *h_in = (char *)malloc(in), *h_out = (char *)malloc(out), *p_in, *p_out, *d_in, *d_out;
    memset(h_in, 1, in); memset(h_out, 1, out);
    CK(hipHostMalloc((void **)&p_in, in, hipHostMallocDefault)); CK(hipHostMalloc((void **)&p_out, out, hipHostMallocDefault));
    memset(p_in, 1, in); memset(p_out, 1, out);
    CK(hipMalloc((void **)&d_in, in)); CK(hipMalloc((void **)&d_out, out));
    hipStream_t s1, s2; CK(hipStreamCreate(&s1)); CK(hipStreamCreate(&s2));
    for (int rep = 0; rep < 2; ++rep) {
        double t = now(); CK(hipMemcpy(d_in, h_in, in, hipMemcpyHostToDevice)); double a = now() - t;
        t = now(); CK(hipMemcpy(h_out, d_out, out, hipMemcpyDeviceToHost)); double b = now() - t;
        printf("pageable  H2D %.2f ms (%.1f GB/s)  D2H %.2f ms (%.1f GB/s)\n", a * 1e3, in / a / 1e9, b * 1e3, out / b / 1e9);
        t = now(); CK(hipMemcpy(d_in, p_in, in, hipMemcpyHostToDevice)); a = now() - t;
        t = now(); CK(hipMemcpy(p_out, d_out, out, hipMemcpyDeviceToHost)); b = now() - t;
        printf("pinned    H2D %.2f ms (%.1f GB/s)  D2H %.2f ms (%.1f GB/s)\n", a * 1e3, in / a / 1e9, b * 1e3, out / b / 1e9);
        t = now();
        CK(hipMemcpyAsync(d_in, p_in, in, hipMemcpyHostToDevice, s1)); CK(hipMemcpyAsync(p_out, d_out, out, hipMemcpyDeviceToHost, s2));
        CK(hipStreamSynchronize(s1)); CK(hipStreamSynchronize(s2)); a = now() - t;
        printf("pinned    H2D + D2H on two streams %.2f ms\n", a * 1e3);
        t = now();
        { std::thread th([&] { CK(hipMemcpy(h_out, d_out, out, hipMemcpyDeviceToHost)); });
          CK(hipMemcpy(d_in, h_in, in, hipMemcpyHostToDevice)); th.join(); }
        a = now() - t;
        printf("pageable  H2D + D2H from two threads %.2f ms\n", a * 1e3);
        t = now();
        { std::thread th([&] { CK(hipMemcpy(d_in + in / 2, h_in + in / 2, in / 2, hipMemcpyHostToDevice)); });
          CK(hipMemcpy(d_in, h_in, in / 2, hipMemcpyHostToDevice)); th.join(); }
        a = now() - t;
        printf("pageable  H2D in two halves from two threads %.2f ms (%.1f GB/s)\n", a * 1e3, in / a / 1e9);
    }
    double t, a;
    {   // pinning a caller's buffer in place (page-aligned part of it)
        char *al = (char *)(((uintptr_t)h_in + 4095) & ~(uintptr_t)4095);
        size_t n = (in - 4096) & ~(size_t)4095;
        t = now(); hipError_t e = hipHostRegister(al, n, hipHostRegisterDefault); a = now() - t;
        if (e == hipSuccess) {
            t = now(); CK(hipMemcpy(d_in, al, n, hipMemcpyHostToDevice)); double b = now() - t;
            t = now(); CK(hipHostUnregister(al)); double c = now() - t;
            printf("pin in place: register %.2f ms, copy %.2f ms (%.1f GB/s), unregister %.2f ms\n", a * 1e3, b * 1e3, n / b / 1e9, c * 1e3);
        } else {
            printf("pin in place: hipHostRegister failed: %s\n", hipGetErrorString(e)); (void)hipGetLastError();
        }
    }
    for (int nt : {1, 2, 4, 8}) {
        t = now();
        std::vector<std::thread> th;
        for (int i = 0; i < nt; ++i) th.emplace_back([&, i] { memcpy(p_in + in / nt * i, h_in + in / nt * i, in / nt); });
        for (auto &x : th) x.join();
        a = now() - t;
        printf("host memcpy pageable -> pinned, %d thread(s): %.2f ms (%.1f GB/s)\n", nt, a * 1e3, in / a / 1e9);
    }
    // chunked: 8 chunks, H2D of chunk k+1 while D2H of chunk k-1, pinned
    for (size_t chunk : {size_t(4) << 20, size_t(16) << 20}) {
        t = now();
        size_t n = in / chunk;
        for (size_t k = 0; k < n; ++k) {
            CK(hipMemcpyAsync(d_in + k * chunk, p_in + k * chunk, chunk, hipMemcpyHostToDevice, s1));
            size_t oc = out / n;
            CK(hipMemcpyAsync(p_out + k * oc, d_out + k * oc, oc, hipMemcpyDeviceToHost, s2));
        }
        CK(hipStreamSynchronize(s1)); CK(hipStreamSynchronize(s2));
        a = now() - t;
        printf("pinned, %zu MB chunks, both directions on two streams: %.2f ms\n", chunk >> 20, a * 1e3);
    }
    return 0;
}
