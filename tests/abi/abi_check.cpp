// A C++ consumer of the C ABI with no Python in between: generates exact pixels for known
// poses with pnp_project_batch_f64, solves them through the host-pointer and the device-pointer
// paths in both element types, and checks that the paths agree bit for bit and that the known
// poses come back.  Built by pnp_solver_test_amd/csrc/Makefile, run by tests/test_gpu_parity.py.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>

#include "pnp_hip.h"

#define CHECK(cond, ...)                      \
    do {                                      \
        if (!(cond)) {                        \
            std::fprintf(stderr, __VA_ARGS__); \
            std::fprintf(stderr, "\n");        \
            return 1;                         \
        }                                     \
    } while (0)
#define PNP(call) CHECK((call) == PNP_OK, "%s failed: %s", #call, pnp_last_error())
#define HIP(call) CHECK((call) == hipSuccess, "%s failed", #call)

template <typename T>
struct Api;
template <>
struct Api<float> {
    static constexpr auto qeif = pnp_qeif_batch_f32;
    static constexpr auto dlt = pnp_dlt_batch_f32;
    static constexpr int is64 = 0;
    static constexpr double tol = 2e-5;
};
template <>
struct Api<double> {
    static constexpr auto qeif = pnp_qeif_batch_f64;
    static constexpr auto dlt = pnp_dlt_batch_f64;
    static constexpr int is64 = 1;
    static constexpr double tol = 1e-9;
};

template <typename T>
int run(const char *tag)
{
    const int64_t B = 1000;
    const int32_t N = 6;
    const double f = 225.68717584155982;
    const double K[9] = {f, 0, 160, 0, f, 120, 0, 0, 1};
    const double pattern[18] = {0.032, 0, 0, -0.032, 0, 0, 0.027, 0.070, 0, -0.027, 0.070, 0,
                                -0.005, 0.0455, -0.03, 0, 0.12, 0};
    std::vector<double> rpy(B * 3), t(B * 3);
    for (int64_t b = 0; b < B; ++b) {  // a deterministic spread of poses inside the stress ranges
        rpy[3 * b] = 40.0 * std::sin(0.37 * b);
        rpy[3 * b + 1] = 40.0 * std::sin(0.53 * b + 1.0);
        rpy[3 * b + 2] = 40.0 * std::sin(0.71 * b + 2.0);
        const double depth = 0.3 + 1.5 * (0.5 + 0.5 * std::sin(0.11 * b));
        t[3 * b] = depth * 0.4 * std::sin(0.23 * b);
        t[3 * b + 1] = depth * 0.4 * std::sin(0.29 * b + 0.5);
        t[3 * b + 2] = depth;
    }
    std::vector<T> uv(B * N * 2), xyz(N * 3);
    for (int i = 0; i < 18; ++i) xyz[i] = static_cast<T>(pattern[i]);
    PNP(pnp_project_batch_f64(rpy.data(), t.data(), pattern, 0, K, B, N, 0.0, uv.data(), Api<T>::is64, PNP_PTR_HOST,
                              nullptr));

    std::vector<T> R(B * 9), tt(B * 3), res(B), ang(B * 3), R2(B * 9), tt2(B * 3), res2(B), ang2(B * 3);
    std::vector<int32_t> it(B), it2(B);
    PNP(Api<T>::qeif(uv.data(), xyz.data(), 0, K, B, N, nullptr, R.data(), tt.data(), res.data(), it.data(), ang.data(),
                     PNP_PTR_HOST, nullptr));

    T *d_uv, *d_xyz, *d_R, *d_t, *d_res, *d_ang;
    int32_t *d_it;
    hipStream_t stream;
    HIP(hipStreamCreate(&stream));
    HIP(hipMalloc(&d_uv, uv.size() * sizeof(T)));
    HIP(hipMalloc(&d_xyz, xyz.size() * sizeof(T)));
    HIP(hipMalloc(&d_R, R.size() * sizeof(T)));
    HIP(hipMalloc(&d_t, tt.size() * sizeof(T)));
    HIP(hipMalloc(&d_res, res.size() * sizeof(T)));
    HIP(hipMalloc(&d_ang, ang.size() * sizeof(T)));
    HIP(hipMalloc(&d_it, it.size() * sizeof(int32_t)));
    HIP(hipMemcpyAsync(d_uv, uv.data(), uv.size() * sizeof(T), hipMemcpyHostToDevice, stream));
    HIP(hipMemcpyAsync(d_xyz, xyz.data(), xyz.size() * sizeof(T), hipMemcpyHostToDevice, stream));
    PNP(Api<T>::qeif(d_uv, d_xyz, 0, K, B, N, nullptr, d_R, d_t, d_res, d_it, d_ang, PNP_PTR_DEVICE, stream));
    HIP(hipMemcpyAsync(R2.data(), d_R, R.size() * sizeof(T), hipMemcpyDeviceToHost, stream));
    HIP(hipMemcpyAsync(tt2.data(), d_t, tt.size() * sizeof(T), hipMemcpyDeviceToHost, stream));
    HIP(hipMemcpyAsync(res2.data(), d_res, res.size() * sizeof(T), hipMemcpyDeviceToHost, stream));
    HIP(hipMemcpyAsync(ang2.data(), d_ang, ang.size() * sizeof(T), hipMemcpyDeviceToHost, stream));
    HIP(hipMemcpyAsync(it2.data(), d_it, it.size() * sizeof(int32_t), hipMemcpyDeviceToHost, stream));
    HIP(hipStreamSynchronize(stream));
    CHECK(std::memcmp(R.data(), R2.data(), R.size() * sizeof(T)) == 0, "%s: R differs between pointer kinds", tag);
    CHECK(std::memcmp(tt.data(), tt2.data(), tt.size() * sizeof(T)) == 0, "%s: t differs between pointer kinds", tag);
    CHECK(std::memcmp(ang.data(), ang2.data(), ang.size() * sizeof(T)) == 0, "%s: angles differ", tag);
    CHECK(std::memcmp(it.data(), it2.data(), it.size() * sizeof(int32_t)) == 0, "%s: iteration counts differ", tag);

    // exact pixels of a pose inside the solver's basin: the pose itself must come back
    double worst_t = 0.0, worst_a = 0.0;
    for (int64_t b = 0; b < B; ++b) {
        for (int k = 0; k < 3; ++k) {
            worst_t = std::fmax(worst_t, std::fabs(static_cast<double>(tt[3 * b + k]) - t[3 * b + k]) / t[3 * b + 2]);
            worst_a = std::fmax(worst_a, std::fabs(static_cast<double>(ang[3 * b + k]) - rpy[3 * b + k]));
        }
        CHECK(it[b] >= 1 && it[b] <= 14, "%s: iteration count %d out of range", tag, it[b]);
    }
    // the filter stops on a 1 % residual change, not at convergence: what it returns is the pose to
    // about 1e-4 relative; the element-type tolerance applies to the two paths' agreement above
    CHECK(worst_t < 2e-3 && worst_a < 0.2, "%s: pose not recovered (dt/t3 %.3g, dangle %.3g deg)", tag, worst_t, worst_a);

    // the DLT entry point on the same pixels, device pointers
    PNP(Api<T>::dlt(d_uv, d_xyz, 0, K, B, N, 3, d_R, d_t, d_res, d_ang, PNP_PTR_DEVICE, stream));
    HIP(hipMemcpyAsync(R2.data(), d_R, R.size() * sizeof(T), hipMemcpyDeviceToHost, stream));
    HIP(hipStreamSynchronize(stream));
    int finite = 0;
    for (int64_t b = 0; b < B; ++b) finite += std::isfinite(static_cast<double>(R2[9 * b]));
    CHECK(finite > 0.95 * B, "%s: DLT produced %d finite results of %lld", tag, finite, static_cast<long long>(B));

    // argument errors come back as codes with a message, not as crashes
    CHECK(Api<T>::qeif(nullptr, d_xyz, 0, K, B, N, nullptr, d_R, d_t, d_res, d_it, d_ang, PNP_PTR_DEVICE, stream) == PNP_E_ARG,
          "%s: NULL uv accepted", tag);
    CHECK(std::strlen(pnp_last_error()) > 0, "%s: empty error text", tag);
    CHECK(Api<T>::qeif(d_uv, d_xyz, 0, K, B, pnp_max_points() + 1, nullptr, d_R, d_t, d_res, d_it, d_ang, PNP_PTR_DEVICE,
                       stream) == PNP_E_ARG, "%s: oversized N accepted", tag);
    for (void *ptr : {static_cast<void *>(d_uv), static_cast<void *>(d_xyz), static_cast<void *>(d_R),
                      static_cast<void *>(d_t), static_cast<void *>(d_res), static_cast<void *>(d_ang),
                      static_cast<void *>(d_it)})
        HIP(hipFree(ptr));
    HIP(hipStreamDestroy(stream));
    std::printf("%s ok: %lld poses, worst dt/t3 %.2e, worst dangle %.2e deg\n", tag, static_cast<long long>(B), worst_t,
                worst_a);
    return 0;
}

int main()
{
    CHECK(pnp_device_count() >= 1, "no HIP device: %s", pnp_last_error());
    PNP(pnp_init(0));
    std::printf("%s\n", pnp_version());
    if (run<float>("float32")) return 1;
    if (run<double>("float64")) return 1;
    std::printf("abi_check passed\n");
    return 0;
}
