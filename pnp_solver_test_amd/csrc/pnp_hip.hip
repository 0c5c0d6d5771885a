// Host side of libpnp_hip.so: argument checking, launch geometry, host-pointer staging and
// the extern "C" entry points declared in include/pnp_hip.h.
#include <hip/hip_runtime.h>

#include <math.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdlib.h>
#include <stdio.h>
#include <string.h>

#include <atomic>
#include <condition_variable>
#include <deque>
#include <functional>
#if defined(__x86_64__)
#include <immintrin.h>
#endif
#include <mutex>
#include <thread>
#include <vector>

#include "pnp_hip.h"
#include <algorithm>
#include <cstring>

#include "pnp_kernels.hpp"

// RCCL's header, where there is one: only its types and prototypes are used (pnp_gather.hpp checks
// its own table against them); the library itself is opened at run time.
#if defined(__has_include)
#if __has_include(<rccl/rccl.h>)
#include <rccl/rccl.h>
#define PNP_HAVE_RCCL_H 1
#endif
#endif

// Staging memory of pnp_qeif_batch_noalloc_f32, the entry point that allocates nothing: every
// device the library's code object is loaded on holds its own copy of this array until the
// process ends.  In host code the name is only a handle for hipGetSymbolAddress; the device
// address it yields (one per device, see scratch_address) is what copies and kernels are given.
static __device__ __align__(256) unsigned char scratch[PNP_NOALLOC_SCRATCH_BYTES];

namespace {

using pnp::kBlock;

constexpr int kMaxPoints = 512;          // LPI = 64 keeps 4 problems x 20 B x 512 points in LDS
// Dynamic LDS of a point-form launch.  A workgroup's correspondences must fit it, which sets the
// smallest number of lanes per problem.  The automatic choice stays within kLdsAuto, which leaves
// room for three workgroups on a compute unit's 160 KB; a forced lane count (pnp_set_lanes_per_problem)
// may take up to kLdsMax, one workgroup per compute unit, which a kernel has to be allowed first
// (allow_lds).
constexpr size_t kLdsAuto = 48 * 1024;
constexpr size_t kLdsPlain = 64 * 1024;   // what a kernel may take without being allowed more
constexpr size_t kLdsMax = 160 * 1024;
constexpr int kMaxDevices = 64;

thread_local char g_err[512] = "";

int fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return code;
}

#define HIP_TRY(expr)                                                                       \
    do {                                                                                    \
        hipError_t e_ = (expr);                                                             \
        if (e_ != hipSuccess)                                                               \
            return fail(PNP_E_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_),   \
                        __FILE__, __LINE__);                                                \
    } while (0)

// Tuning overrides (pnp_set_lanes_per_problem, pnp_set_qeif_form) belong to the thread that set
// them: they change that thread's launches and nobody else's.  A host-pointer call that hands
// chunks to the library's lane threads passes its caller's values along with them.
struct Tuning {
    int forced_lpi = 0;
    int qeif_form = PNP_QEIF_FORM_AUTO;
};
thread_local Tuning g_tuning;

// What the calling thread's last solver launch was: the kernel by the name a profiler shows for it,
// and its geometry.  Filled by the launchers; read by pnp_last_kernel_name / pnp_last_launch.
struct LaunchRecord {
    char kernel[128] = "";
    int64_t grid = 0;
    int32_t block = 0, lds = 0;
};
thread_local LaunchRecord g_last_launch;

template <typename T>
constexpr const char *elem_name()
{
    return sizeof(T) == 4 ? "float" : "double";
}

void note_launch(int64_t grid, int block, size_t lds, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_last_launch.kernel, sizeof g_last_launch.kernel, fmt, ap);
    va_end(ap);
    g_last_launch.grid = grid;
    g_last_launch.block = block;
    g_last_launch.lds = static_cast<int32_t>(lds);
}

// The moment form pays once an iteration's fixed cost (~700 fused operations) is below the
// point-by-point 78 N: measured crossover between N = 8 and N = 10 at half a million problems.
// It runs one lane per problem; at N = 68 it is the faster form down to 2048 problems (the
// smallest batch measured), below that the many-lanes-per-problem point form is kept.
constexpr int kMomentMinPoints = 10;
constexpr int64_t kMomentMinBatch = 2048;

// ---- per-device staging for host-pointer calls -------------------------------------------

// Large host-pointer calls are cut into chunks that kPipeLanes host threads ("lanes") carry side
// by side, each with a stream and a page-locked buffer of its own.  A lane copies a chunk from the
// caller's (pageable) arrays into its buffer, launches the solver on the device address of that
// buffer - the kernels read the inputs and write the results over the bus, there are no copy
// operations and no device-side staging - waits, and copies the results out to the caller's arrays;
// a side of the call (inputs, outputs) whose arrays are page-locked already is handed to the kernels
// as it is.  The lanes of a device are threads that stay: the first call large enough to be chunked
// starts them, later calls find them waiting, and the chunks of two callers share them (and the bus)
// chunk by chunk.  They end when the library is unloaded or the process exits.
// Measured on the pool's hosts: one thread copies pageable -> page-locked memory at 23 GB/s, eight
// at 80 GB/s; page-locked memory crosses the bus at 57 GB/s in either direction, both at once.
// Lanes and chunk size from a sweep on one box (1M six-point problems with a pattern each, C ABI
// with existing result arrays, best of 12 calls): 8 lanes x 4 MB 3.0 ms, 8 x 2 MB 2.86, 8 x 1 MB
// 2.58, 8 x 512 KB 2.94, 12 x 1 MB 2.57-2.58 (and the steadier median), 12 x 512 KB 2.82, 16 x 512 KB
// 2.89: small chunks shorten the fill and drain of the pipeline until the per-chunk launch and
// synchronisation cost takes over; the extra lanes help most when the result arrays are fresh
// pages that the copies out have to fault in.
#ifndef PNP_PIPE_LANES
#define PNP_PIPE_LANES 12
#endif
#ifndef PNP_PIPE_CHUNK_KB
#define PNP_PIPE_CHUNK_KB 1024
#endif
constexpr int kPipeLanes = PNP_PIPE_LANES;
constexpr size_t kPipeChunkBytes = size_t(PNP_PIPE_CHUNK_KB) << 10;   // input bytes per chunk, about
#ifndef PNP_PIPE_MIN_MB
#define PNP_PIPE_MIN_MB 8
#endif
constexpr size_t kPipeMinBytes = size_t(PNP_PIPE_MIN_MB) << 20;    // calls moving less than this stay on one stream

struct PipeLane {
    hipStream_t stream = nullptr;
    void *pinned = nullptr;      // page-locked buffer of the chunk in hand (grow-only)
    void *pinned_dev = nullptr;  // its device address
    size_t pinned_cap = 0;
    void *pattern = nullptr;     // device copy of the shared pattern(s) of the call in hand (grow-only)
    size_t pattern_cap = 0;
    uint64_t pattern_of = 0;     // the call whose pattern it holds
};

// The lane threads of one device and the queue of chunks they work off.
struct LanePool {
    std::mutex mu;
    std::condition_variable wake;
    std::deque<std::function<void(PipeLane &)>> chunks;
    std::vector<std::thread> threads;
    PipeLane lane[kPipeLanes];
    bool stop = false;

    // Starts the lane threads of device `dev` if they are not running; caller does not hold mu.
    int ensure_started(int dev)
    {
        std::lock_guard<std::mutex> lock(mu);
        if (!threads.empty()) return PNP_OK;
        try {
            threads.reserve(kPipeLanes);
            for (int l = 0; l < kPipeLanes; ++l) threads.emplace_back(&LanePool::run, this, dev, l);
        } catch (...) {
            // fewer lanes than asked for still carry every chunk; none at all is an error
        }
        if (threads.empty()) return fail(PNP_E_HIP, "no thread could be started for the host-pointer pipeline");
        return PNP_OK;
    }

    void run(int dev, int l)
    {
        (void)hipSetDevice(dev);
        PipeLane &ln = lane[l];
        for (;;) {
            std::function<void(PipeLane &)> job;
            {
                std::unique_lock<std::mutex> lock(mu);
                wake.wait(lock, [&] { return stop || !chunks.empty(); });
                if (chunks.empty()) return;   // stop, and nothing left to carry
                job = std::move(chunks.front());
                chunks.pop_front();
            }
            job(ln);
        }
    }

    // The threads are joined before the process image goes away; the buffers and streams are left
    // to the runtime, which may already be shutting down by then.
    ~LanePool()
    {
        {
            std::lock_guard<std::mutex> lock(mu);
            stop = true;
        }
        wake.notify_all();
        for (auto &t : threads)
            if (t.joinable()) t.join();
    }
};

struct Staging {
    std::mutex mu;
    void *buf = nullptr;
    size_t cap = 0;
    void *pinned = nullptr;  // page-locked host memory for calls below the chunking threshold (grow-only)
    void *pinned_dev = nullptr;  // its device address
    size_t pinned_cap = 0;
};

// Smallest size of the page-locked buffer that calls below the chunking threshold go through.
constexpr size_t kSmallCall = 256 * 1024;
Staging g_staging[kMaxDevices];
LanePool g_lanes[kMaxDevices];
std::atomic<uint64_t> g_call_serial{0};

// Grow-only device buffer of the current device; caller holds st.mu.
int staging_reserve(Staging &st, size_t bytes)
{
    if (bytes <= st.cap) return PNP_OK;
    if (st.buf) HIP_TRY(hipFree(st.buf));
    st.buf = nullptr;
    st.cap = 0;
    size_t want = bytes + bytes / 4 + 4096;
    HIP_TRY(hipMalloc(&st.buf, want));
    st.cap = want;
    return PNP_OK;
}

// Carves 256-byte-aligned pieces out of one buffer.
struct Carver {
    char *base;
    size_t off = 0;
    explicit Carver(void *b) : base(static_cast<char *>(b)) {}
    template <typename T>
    T *take(size_t count)
    {
        T *p = reinterpret_cast<T *>(base + off);
        off += (count * sizeof(T) + 255) & ~static_cast<size_t>(255);
        return p;
    }
};

// Synchronises a stream when it goes out of scope unless dismissed: an error exit after the first
// enqueue must not hand the staging buffers to the next caller while copies are in flight.
struct SyncOnExit {
    hipStream_t s;
    bool armed = true;
    explicit SyncOnExit(hipStream_t st) : s(st) {}
    ~SyncOnExit() { if (armed) (void)hipStreamSynchronize(s); }
    SyncOnExit(const SyncOnExit &) = delete;
    SyncOnExit &operator=(const SyncOnExit &) = delete;
};

// Device address of `scratch` on each device, resolved once per device with that device current.
struct StaticScratch {
    std::mutex mu;
    void *ptr = nullptr;
};
StaticScratch g_scratch[kMaxDevices];

// Caller holds g_scratch[dev].mu and `dev` is the current device.
int scratch_address(int dev, void **out)
{
    StaticScratch &sc = g_scratch[dev];
    if (!sc.ptr) {
        void *p = nullptr;
        HIP_TRY(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
        if (!p) return fail(PNP_E_HIP, "hipGetSymbolAddress gave no address for the static staging array");
        sc.ptr = p;
    }
    *out = sc.ptr;
    return PNP_OK;
}

// ---- small host helpers --------------------------------------------------------------------

// First two rows of K^-1 in float64 by the adjugate.
int kinv_rows(const double K[9], pnp::KinvRows &ki)
{
    const double a = K[0], b = K[1], c = K[2], d = K[3], e = K[4], f = K[5], g = K[6], h = K[7], i = K[8];
    const double det = a * (e * i - f * h) - b * (d * i - f * g) + c * (d * h - e * g);
    if (!(det != 0.0) || !isfinite(det)) return fail(PNP_E_ARG, "camera matrix K is singular or not finite");
    const double r = 1.0 / det;
    ki.a00 = (e * i - f * h) * r;
    ki.a01 = (c * h - b * i) * r;
    ki.a02 = (b * f - c * e) * r;
    ki.a10 = (f * g - d * i) * r;
    ki.a11 = (a * i - c * g) * r;
    ki.a12 = (c * d - a * f) * r;
    return PNP_OK;
}

size_t lds_bytes(int lpi, int N, bool per_problem_xyz, int n_patterns, size_t elem)
{
    const size_t pb = (PNP_SOLVE_BLOCK_BYTES / elem) / lpi;   // pnp::kSolveBlock<T> / lpi
    const size_t uv = pb * 2 * N;
    const size_t xyz = per_problem_xyz ? pb * 3 * N : static_cast<size_t>(n_patterns) * 3 * N;
    return (uv + xyz) * elem;
}

// Lets kernel `fn` take `lds` bytes of dynamic LDS on the current device when that is more than a
// kernel may take unasked; remembered per (device, kernel).
int allow_lds(const void *fn, size_t lds)
{
    if (lds <= kLdsPlain) return PNP_OK;
    if (lds > kLdsMax) return fail(PNP_E_ARG, "%zu bytes of LDS per workgroup exceed the %zu a compute unit has", lds, kLdsMax);
    static std::mutex mu;
    static std::vector<std::pair<int, const void *>> done;
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lock(mu);
    for (const auto &d : done)
        if (d.first == dev && d.second == fn) return PNP_OK;
    HIP_TRY(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(kLdsMax)));
    done.emplace_back(dev, fn);
    return PNP_OK;
}

// Lanes per problem.  The floor is set by LDS: a workgroup's correspondences must fit the
// budget.  Above the floor, more lanes per problem only pay when the batch is too small to
// give every lane of the chip its own problem.
int choose_lpi(int64_t B, int N, bool per_problem_xyz, int n_patterns, size_t elem)
{
    const int forced = g_tuning.forced_lpi;
    int lpi = 1;
    if (forced > 0) {
        while (lpi < 64 && lds_bytes(lpi, N, per_problem_xyz, n_patterns, elem) > kLdsMax) lpi *= 2;
        return forced > lpi ? forced : lpi;
    }
    while (lpi < 64 && lds_bytes(lpi, N, per_problem_xyz, n_patterns, elem) > kLdsAuto) lpi *= 2;
    const int64_t chip_lanes = 256LL * 64 * 8;  // CUs x wave width x resident waves aimed for
    while (lpi < 64 && 2 * lpi <= N && B * lpi * 2 <= chip_lanes) lpi *= 2;
    return lpi;
}

int check_common(const void *uv, const void *xyz, int64_t xyz_stride, const double *K, int64_t B,
                 int32_t N, const void *R_out, const void *t_out, int32_t ptr_kind)
{
    if (B < 0) return fail(PNP_E_ARG, "B = %lld is negative", static_cast<long long>(B));
    if (N < 1 || N > kMaxPoints) return fail(PNP_E_ARG, "N = %d outside [1, %d]", N, kMaxPoints);
    if (!K) return fail(PNP_E_ARG, "K is NULL");
    if (ptr_kind != PNP_PTR_HOST && ptr_kind != PNP_PTR_DEVICE)
        return fail(PNP_E_ARG, "ptr_kind = %d is neither PNP_PTR_HOST nor PNP_PTR_DEVICE", ptr_kind);
    if (B > 0 && (!uv || !xyz || !R_out || !t_out))
        return fail(PNP_E_ARG, "uv, xyz, R_out and t_out must not be NULL");
    if (xyz_stride != 0 && xyz_stride < 3 * static_cast<int64_t>(N))
        return fail(PNP_E_ARG, "xyz_problem_stride = %lld is neither 0 nor >= 3*N",
                    static_cast<long long>(xyz_stride));
    return PNP_OK;
}

pnp::QeifParams to_device_params(const pnp_qeif_params *p)
{
    pnp_qeif_params d;
    pnp_qeif_default_params(&d);
    if (p) d = *p;
    pnp::QeifParams q;
    q.q_inv = d.q_inv;
    q.r_quat = d.r_quat;
    q.r_delta = d.r_delta;
    q.omega0 = d.omega0;
    q.rel_tol = d.rel_tol;
    q.res_old0 = d.res_old0;
    q.max_it = d.max_it;
    q.c = sqrt(d.q_inv);
    q.c_inv = 1.0 / q.c;
    q.rq_inv = 1.0 / d.r_quat;
    q.rd_inv = 1.0 / d.r_delta;
    q.oq = 1.0 / (1.0 / d.omega0 + d.r_quat);
    q.od = 1.0 / (1.0 / d.omega0 + d.r_delta);
    // |res - res_old| < tol res_old on squared residuals: lo2 ss_old < ss < hi2 ss_old.  A bound
    // whose factor is not positive is always (lower) or never (upper) met for res_old > 0, which
    // the negative stand-ins reproduce; res_old = 0 and NaN never stop, as in the original form.
    const double lo = 1.0 - d.rel_tol, hi = 1.0 + d.rel_tol;
    q.lo2 = lo > 0.0 ? lo * lo : (lo < 0.0 ? -1.0 : 0.0);
    q.hi2 = hi > 0.0 ? hi * hi : -1.0;
    return q;
}

// ---- launchers (device pointers only; T = element type of the problem arrays) ----------------

// A/B switches, present only in a build with -DPNP_AB_SWITCHES (libpnp_hip_ab.so, which the tests
// and timing scripts that compare kernel variants load): PNP_MOMENT_PLAIN=1 in the environment sends
// per-problem patterns through the equal-chunk moment kernel instead of the line-cut one, and
// PNP_ROWS_STAGED=1 six-point launches through the LDS-staged kernels even where every lane could
// read its own rows.  The library proper reads no environment variable.
#ifdef PNP_AB_SWITCHES
bool env_is_1(const char *name)
{
    const char *v = getenv(name);
    return v != nullptr && v[0] == '1';
}
const bool g_moment_plain = env_is_1("PNP_MOMENT_PLAIN");
const bool g_rows_staged = env_is_1("PNP_ROWS_STAGED");
#else
constexpr bool g_moment_plain = false;
constexpr bool g_rows_staged = false;
#endif

// True when a QEIF launch should run in moment form.
bool use_moment_form(int64_t B, int N, int n_patterns)
{
    if (n_patterns != 1) return false;
    const int form = g_tuning.qeif_form;
    if (form == PNP_QEIF_FORM_POINTS) return false;
    if (form == PNP_QEIF_FORM_MOMENTS) return true;
    return g_tuning.forced_lpi == 0 && N >= kMomentMinPoints && B >= kMomentMinBatch;
}

// LDS of qeif_moment_lines_kernel<T>: more than a kernel may take without being allowed to, which
// is done once per device (on pnp_init, or else on the first launch that wants the kernel).
template <typename T>
constexpr size_t lines_kernel_lds()
{
    return static_cast<size_t>(5) * pnp::kMomentChunkLine<T> * pnp::kMomentBlock<T> * sizeof(T);
}

template <typename T>
bool lines_kernel_allowed()
{
    static std::atomic<uint64_t> allowed{0}, refused{0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return false;
    const uint64_t bit = uint64_t(1) << (dev & 63);
    if (!((allowed.load() | refused.load()) & bit)) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(pnp::qeif_moment_lines_kernel<T, false>),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize,
                                                 static_cast<int>(lines_kernel_lds<T>()));
        if (e != hipSuccess) (void)hipGetLastError();
        (e == hipSuccess ? allowed : refused).fetch_or(bit);
    }
    return (allowed.load() & bit) != 0;
}

template <typename T>
int launch_qeif_moment(hipStream_t s, const T *uv, const T *xyz, int64_t xyz_stride, const pnp::KinvRows &ki,
                       int64_t B, int N, const pnp::QeifParams &p, T *R, T *t, T *res, int32_t *iters, T *rpy)
{
    constexpr int block = pnp::kMomentBlock<T>;
    constexpr size_t row = block + 1;
    constexpr int c_own = pnp::kMomentChunkOwn<T>, c_shared = pnp::kMomentChunkShared<T>;
    const int64_t nblk = (B + block - 1) / block;
    if (nblk > 0x7fffffffLL) return fail(PNP_E_ARG, "B = %lld needs too many workgroups", static_cast<long long>(B));
    const dim3 grid(static_cast<unsigned>(nblk));
    if (xyz_stride == 0) {
        // pixel chunk + 72 bytes per pattern point; the long chunk only while that fits 64 KB
        const size_t pat = static_cast<size_t>(9) * N * sizeof(double);
        const size_t lds_long = ((2 * c_shared * row * sizeof(T) + 7) & ~static_cast<size_t>(7)) + pat;
        const size_t lds_short = ((2 * c_own * row * sizeof(T) + 7) & ~static_cast<size_t>(7)) + pat;
        // pixels in whole cache lines (16-byte pieces, 8 per problem) + the table
        const size_t lds_lines = static_cast<size_t>(2) * pnp::kMomentChunkLine<T> * block * sizeof(T) + pat;
        // measured on 128k-512k problems: ahead (2-3 %) from about 48 float32 (24 float64) points per
        // problem, and it asks for every line once where the equal chunks ask for a quarter of them twice
        const bool long_rows = static_cast<size_t>(N) * sizeof(T) >= 192;
        if (lds_lines <= kLdsPlain && long_rows && !g_moment_plain) {
            hipLaunchKernelGGL((pnp::qeif_moment_lines_kernel<T, true>), grid, dim3(block), lds_lines, s, uv, xyz, xyz_stride, ki,
                               B, N, p, R, t, res, iters, rpy);
            note_launch(nblk, block, lds_lines, "pnp::qeif_moment_lines_kernel<%s, true>", elem_name<T>());
        } else if (lds_long <= kLdsPlain) {
            hipLaunchKernelGGL((pnp::qeif_moment_kernel<T, c_shared, true>), grid, dim3(block), lds_long, s, uv, xyz,
                               xyz_stride, ki, B, N, p, R, t, res, iters, rpy);
            note_launch(nblk, block, lds_long, "pnp::qeif_moment_kernel<%s, %d, true>", elem_name<T>(), c_shared);
        } else {
            hipLaunchKernelGGL((pnp::qeif_moment_kernel<T, c_own, true>), grid, dim3(block), lds_short, s, uv, xyz,
                               xyz_stride, ki, B, N, p, R, t, res, iters, rpy);
            note_launch(nblk, block, lds_short, "pnp::qeif_moment_kernel<%s, %d, true>", elem_name<T>(), c_own);
        }
    } else {
        // chunks cut at the cache lines: all of a CU's LDS between two workgroups
        constexpr size_t lines_lds = lines_kernel_lds<T>();
        const bool lines_ready = lines_kernel_allowed<T>();
        // its loads address a workgroup's rows through 32-bit byte offsets
        const bool offsets_fit = static_cast<uint64_t>(block) * static_cast<uint64_t>(xyz_stride) * sizeof(T) < (uint64_t(1) << 31);
        // measured on 256k-512k problems: ahead from about 24 float32 (12 float64) points per problem
        const bool long_rows = static_cast<size_t>(N) * sizeof(T) >= 96;
        if (lines_ready && offsets_fit && long_rows && !g_moment_plain) {
            hipLaunchKernelGGL((pnp::qeif_moment_lines_kernel<T, false>), grid, dim3(block), lines_lds, s, uv, xyz, xyz_stride, ki,
                               B, N, p, R, t, res, iters, rpy);
            note_launch(nblk, block, lines_lds, "pnp::qeif_moment_lines_kernel<%s, false>", elem_name<T>());
        } else {
            const size_t lds = 5 * c_own * row * sizeof(T);
            hipLaunchKernelGGL((pnp::qeif_moment_kernel<T, c_own, false>), grid, dim3(block), lds, s, uv, xyz,
                               xyz_stride, ki, B, N, p, R, t, res, iters, rpy);
            note_launch(nblk, block, lds, "pnp::qeif_moment_kernel<%s, %d, false>", elem_name<T>(), c_own);
        }
    }
    HIP_TRY(hipGetLastError());
    return PNP_OK;
}

// True when the lanes of a six-point, one-lane-per-problem launch may read their rows themselves
// (pnp::OwnRows6): rows in device memory (over the bus the partial cache lines of such loads would
// be fetched once per load), pixel rows 16-byte aligned, pattern rows aligned to two elements.
template <typename T>
bool own_rows_ok(const T *uv, const T *xyz, int64_t xyz_stride, int lpi, int N, int64_t B, bool rows_on_device)
{
    return rows_on_device && lpi == 1 && N == 6 && (xyz_stride == 0 || xyz_stride >= 18) && xyz_stride % 2 == 0 &&
           reinterpret_cast<uintptr_t>(uv) % 16 == 0 && reinterpret_cast<uintptr_t>(xyz) % (2 * sizeof(T)) == 0 &&
           !g_rows_staged;
}

template <typename T>
int launch_qeif(hipStream_t s, const T *uv, const T *xyz, int64_t xyz_stride, int n_patterns,
                const pnp::KinvRows &ki, int64_t B, int N, const pnp::QeifParams &p, T *R, T *t, T *res,
                int32_t *iters, T *rpy, int32_t *pid, int64_t plan_B = -1, bool rows_on_device = true)
{
    if (B == 0) return PNP_OK;
    // plan_B: the size of the batch this launch is a chunk of; form and lanes per problem are
    // chosen for that size, so that a chunked call computes exactly what one launch would
    if (plan_B < B) plan_B = B;
    // the moment kernel addresses a workgroup's rows with 32-bit offsets: 256 rows of xyz_stride
    if (use_moment_form(plan_B, N, n_patterns) && xyz_stride <= (int64_t(1) << 22)) {
        if (pid) HIP_TRY(hipMemsetAsync(pid, 0, static_cast<size_t>(B) * sizeof(int32_t), s));
        return launch_qeif_moment<T>(s, uv, xyz, xyz_stride, ki, B, N, p, R, t, res, iters, rpy);
    }
    const int lpi = choose_lpi(plan_B, N, xyz_stride != 0, n_patterns, sizeof(T));
    const size_t lds = lds_bytes(lpi, N, xyz_stride != 0, n_patterns, sizeof(T));
    if (lds > kLdsMax)
        return fail(PNP_E_ARG, "N = %d with %d pattern(s) needs %zu bytes of LDS per workgroup (max %zu)",
                    N, n_patterns, lds, kLdsMax);
    constexpr int block = pnp::kSolveBlock<T>;
    const int64_t pb = block / lpi;
    const int64_t nblk = (B + pb - 1) / pb;
    if (nblk > 0x7fffffffLL) return fail(PNP_E_ARG, "B = %lld needs too many workgroups", static_cast<long long>(B));
    const dim3 grid(static_cast<unsigned>(nblk));
    if (n_patterns == 1 && own_rows_ok(uv, xyz, xyz_stride, lpi, N, B, rows_on_device)) {
        const int64_t rows_blk = (B + pnp::kRowsBlock - 1) / pnp::kRowsBlock;
        hipLaunchKernelGGL((pnp::qeif_rows6_kernel<T>), dim3(static_cast<unsigned>(rows_blk)), dim3(pnp::kRowsBlock), 0, s, uv, xyz,
                           xyz_stride, ki, B, p, R, t, res, iters, rpy, pid);
        HIP_TRY(hipGetLastError());
        note_launch(rows_blk, pnp::kRowsBlock, 0, "pnp::qeif_rows6_kernel<%s>", elem_name<T>());
        return PNP_OK;
    }
    int rc = PNP_OK;
#define PNP_Q(LPI_, NCT_)                                                                                    \
    do {                                                                                                     \
        if (n_patterns > 1) {                                                                                \
            auto kfn = pnp::qeif_kernel<LPI_, NCT_, T, true>;                                                \
            if ((rc = allow_lds(reinterpret_cast<const void *>(kfn), lds)) != PNP_OK) break;                 \
            hipLaunchKernelGGL(kfn, grid, dim3(block), lds, s, uv, xyz, xyz_stride, n_patterns, ki, B, N, p, \
                               R, t, res, iters, rpy, pid);                                                  \
        } else {                                                                                             \
            auto kfn = pnp::qeif_kernel<LPI_, NCT_, T, false>;                                               \
            if ((rc = allow_lds(reinterpret_cast<const void *>(kfn), lds)) != PNP_OK) break;                 \
            hipLaunchKernelGGL(kfn, grid, dim3(block), lds, s, uv, xyz, xyz_stride, n_patterns, ki, B, N, p, \
                               R, t, res, iters, rpy, pid);                                                  \
        }                                                                                                    \
        note_launch(nblk, block, lds, "pnp::qeif_kernel<%d, %d, %s, %s>", LPI_, NCT_, elem_name<T>(),        \
                    n_patterns > 1 ? "true" : "false");                                                      \
    } while (0)
    switch (lpi) {
    case 1:
        if (N == 6) PNP_Q(1, 6);
        else PNP_Q(1, 0);
        break;
    case 2: PNP_Q(2, 0); break;
    case 4: PNP_Q(4, 0); break;
    case 8: PNP_Q(8, 0); break;
    case 16: PNP_Q(16, 0); break;
    case 32: PNP_Q(32, 0); break;
    default: PNP_Q(64, 0); break;
    }
#undef PNP_Q
    if (rc) return rc;
    HIP_TRY(hipGetLastError());
    return PNP_OK;
}

template <typename T>
int launch_dlt(hipStream_t s, const T *uv, const T *xyz, int64_t xyz_stride, const pnp::KinvRows &ki,
               int64_t B, int N, int num_it, T *R, T *t, T *res, T *rpy, int64_t plan_B = -1,
               bool rows_on_device = true)
{
    if (B == 0) return PNP_OK;
    if (plan_B < B) plan_B = B;   // see launch_qeif
    const int lpi = choose_lpi(plan_B, N, xyz_stride != 0, 1, sizeof(T));
    const size_t lds = lds_bytes(lpi, N, xyz_stride != 0, 1, sizeof(T));
    if (lds > kLdsMax) return fail(PNP_E_ARG, "N = %d needs %zu bytes of LDS per workgroup (max %zu)", N, lds, kLdsMax);
    constexpr int block = pnp::kSolveBlock<T>;
    const int64_t pb = block / lpi;
    const int64_t nblk = (B + pb - 1) / pb;
    if (nblk > 0x7fffffffLL) return fail(PNP_E_ARG, "B = %lld needs too many workgroups", static_cast<long long>(B));
    const dim3 grid(static_cast<unsigned>(nblk));
    if (own_rows_ok(uv, xyz, xyz_stride, lpi, N, B, rows_on_device)) {
        const int64_t rows_blk = (B + pnp::kRowsBlock - 1) / pnp::kRowsBlock;
        hipLaunchKernelGGL((pnp::dlt_rows6_kernel<T>), dim3(static_cast<unsigned>(rows_blk)), dim3(pnp::kRowsBlock), 0, s, uv, xyz,
                           xyz_stride, ki, B, N, num_it, R, t, res, rpy);
        HIP_TRY(hipGetLastError());
        note_launch(rows_blk, pnp::kRowsBlock, 0, "pnp::dlt_rows6_kernel<%s>", elem_name<T>());
        return PNP_OK;
    }
    int rc = PNP_OK;
#define PNP_D(LPI_, NCT_)                                                                                          \
    do {                                                                                                           \
        auto kfn = pnp::dlt_kernel<LPI_, NCT_, T>;                                                                 \
        if ((rc = allow_lds(reinterpret_cast<const void *>(kfn), lds)) != PNP_OK) break;                           \
        hipLaunchKernelGGL(kfn, grid, dim3(block), lds, s, uv, xyz, xyz_stride, ki, B, N, num_it, R, t, res, rpy); \
        note_launch(nblk, block, lds, "pnp::dlt_kernel<%d, %d, %s>", LPI_, NCT_, elem_name<T>());                  \
    } while (0)
    switch (lpi) {
    case 1:
        if (N == 6) PNP_D(1, 6); else PNP_D(1, 0);
        break;
    case 2: PNP_D(2, 0); break;
    case 4: PNP_D(4, 0); break;
    case 8: PNP_D(8, 0); break;
    case 16: PNP_D(16, 0); break;
    case 32: PNP_D(32, 0); break;
    default: PNP_D(64, 0); break;
    }
#undef PNP_D
    if (rc) return rc;
    HIP_TRY(hipGetLastError());
    return PNP_OK;
}

template <typename T>
int launch_lm12(hipStream_t s, const T *uv, const T *xyz, int64_t xyz_stride, const pnp::KinvRows &ki,
                int64_t B, int N, double lambda, int num_it, T *R, T *t, T *res, T *rpy)
{
    if (B == 0) return PNP_OK;
    // problems per wavefront: 64, or the largest power-of-two fraction whose points fit the LDS budget
    int lpi = 1;
    auto lds_for = [&](int l) {
        const size_t pb = pnp::kLmBlock / l;
        return (pb * 2 * N + (xyz_stride ? pb * 3 * N : static_cast<size_t>(3) * N)) * sizeof(T);
    };
    while (lpi < 64 && lds_for(lpi) > kLdsAuto) lpi *= 2;
    const size_t lds = lds_for(lpi);
    if (lds > kLdsPlain)
        return fail(PNP_E_ARG, "N = %d needs %zu bytes of LDS per workgroup in the LM solver (max %zu)", N, lds, kLdsPlain);
    const int64_t pb = pnp::kLmBlock / lpi;
    const int64_t nblk = (B + pb - 1) / pb;
    if (nblk > 0x7fffffffLL) return fail(PNP_E_ARG, "B = %lld needs too many workgroups", static_cast<long long>(B));
#define PNP_L(LPI_)                                                                                                 \
    do {                                                                                                            \
        hipLaunchKernelGGL((pnp::lm12_kernel<T, LPI_>), dim3(static_cast<unsigned>(nblk)), dim3(pnp::kLmBlock), lds, s, \
                           uv, xyz, xyz_stride, ki, B, N, lambda, num_it, R, t, res, rpy);                          \
        note_launch(nblk, pnp::kLmBlock, lds, "pnp::lm12_kernel<%s, %d>", elem_name<T>(), LPI_);                    \
    } while (0)
    switch (lpi) {
    case 1: PNP_L(1); break;
    case 2: PNP_L(2); break;
    case 4: PNP_L(4); break;
    case 8: PNP_L(8); break;
    case 16: PNP_L(16); break;
    case 32: PNP_L(32); break;
    default: PNP_L(64); break;
    }
#undef PNP_L
    HIP_TRY(hipGetLastError());
    return PNP_OK;
}

// ---- host-pointer wrapper ----------------------------------------------------------------------

template <typename T>
struct SolveIO {
    const T *uv;
    const T *xyz;
    T *R, *t, *res, *rpy;
    int32_t *iters, *pid;
};

// Grow-only resources of one pipeline lane of the current device; called by the lane's own thread.
int lane_reserve(PipeLane &ln, size_t pinned_bytes, size_t pattern_bytes)
{
    if (!ln.stream) HIP_TRY(hipStreamCreateWithFlags(&ln.stream, hipStreamNonBlocking));
    if (pinned_bytes > ln.pinned_cap) {
        if (ln.pinned) HIP_TRY(hipHostFree(ln.pinned));
        ln.pinned = ln.pinned_dev = nullptr;
        ln.pinned_cap = 0;
        HIP_TRY(hipHostMalloc(&ln.pinned, pinned_bytes, hipHostMallocDefault));
        ln.pinned_cap = pinned_bytes;
        HIP_TRY(hipHostGetDevicePointer(&ln.pinned_dev, ln.pinned, 0));
    }
    if (pattern_bytes > ln.pattern_cap) {
        if (ln.pattern) HIP_TRY(hipFree(ln.pattern));
        ln.pattern = nullptr;
        ln.pattern_cap = 0;
        ln.pattern_of = 0;
        HIP_TRY(hipMalloc(&ln.pattern, pattern_bytes + 256));
        ln.pattern_cap = pattern_bytes + 256;
    }
    return PNP_OK;
}

// Bulk copy between the caller's arrays and a lane's page-locked buffer with streaming stores:
// the destination is not read by this CPU next (the GPU reads the buffer, the caller reads its
// results later), so its lines are written without being fetched first.
#if defined(__x86_64__)
__attribute__((target("avx2"))) void copy_streaming_avx2(void *dst, const void *src, size_t n)
{
    char *d = static_cast<char *>(dst);
    const char *s = static_cast<const char *>(src);
    const size_t head = (32 - (reinterpret_cast<uintptr_t>(d) & 31)) & 31;
    if (n < 4096 || head > n) {
        std::memcpy(d, s, n);
        return;
    }
    std::memcpy(d, s, head);
    d += head; s += head; n -= head;
    const size_t blocks = n / 128;
    for (size_t i = 0; i < blocks; ++i) {
        const __m256i a = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(s));
        const __m256i b = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(s + 32));
        const __m256i c = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(s + 64));
        const __m256i e = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(s + 96));
        _mm256_stream_si256(reinterpret_cast<__m256i *>(d), a);
        _mm256_stream_si256(reinterpret_cast<__m256i *>(d + 32), b);
        _mm256_stream_si256(reinterpret_cast<__m256i *>(d + 64), c);
        _mm256_stream_si256(reinterpret_cast<__m256i *>(d + 96), e);
        s += 128; d += 128;
    }
    _mm_sfence();
    std::memcpy(d, s, n - blocks * 128);
}
#endif

void copy_streaming(void *dst, const void *src, size_t n)
{
#if defined(__x86_64__)
    static const bool avx2 = __builtin_cpu_supports("avx2");
    if (avx2) return copy_streaming_avx2(dst, src, n);
#endif
    std::memcpy(dst, src, n);
}

// True when p points into page-locked host memory the HIP runtime knows (hipHostMalloc, a
// registered range, a pinned torch tensor): memory the kernels can work on over the bus.
bool is_page_locked(const void *p)
{
    if (!p) return true;   // an output the caller does not want: nothing to copy
    hipPointerAttribute_t a;
    if (hipPointerGetAttributes(&a, p) != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    return a.type == hipMemoryTypeHost;
}

// Device addresses of a call's host arrays when every one of them is page-locked and mapped;
// false (d unspecified) otherwise.
template <typename T>
bool map_page_locked(const SolveIO<T> &h, SolveIO<T> &d)
{
    const void *all[] = {h.uv, h.xyz, h.R, h.t, h.res, h.rpy, h.iters, h.pid};
    void *mapped[8];
    for (int i = 0; i < 8; ++i) {
        mapped[i] = nullptr;
        if (!all[i]) continue;
        if (!is_page_locked(all[i])) return false;
        if (hipHostGetDevicePointer(&mapped[i], const_cast<void *>(all[i]), 0) != hipSuccess || !mapped[i]) {
            (void)hipGetLastError();
            return false;
        }
    }
    d.uv = static_cast<const T *>(mapped[0]);
    d.xyz = static_cast<const T *>(mapped[1]);
    d.R = static_cast<T *>(mapped[2]);
    d.t = static_cast<T *>(mapped[3]);
    d.res = static_cast<T *>(mapped[4]);
    d.rpy = static_cast<T *>(mapped[5]);
    d.iters = static_cast<int32_t *>(mapped[6]);
    d.pid = static_cast<int32_t *>(mapped[7]);
    return true;
}

// The chunked form of a host-pointer call (see kPipeLanes).  xyz_elems: elements of the shared
// pattern(s) when own_xyz is false.  run(d, count, stream) solves `count` problems from device
// pointers d that point at the chunk's own arrays; it is called on the lane threads, with the
// caller's tuning overrides in force.
template <typename T, typename Run>
int pipelined_host_call(int dev, hipStream_t s, int64_t B, int32_t N, bool own_xyz, size_t xyz_elems,
                        const SolveIO<T> &h, Run run)
{
    LanePool &pool = g_lanes[dev];
    int rc = pool.ensure_started(dev);
    if (rc) return rc;
    // chunks of about kPipeChunkBytes of input; smaller ones (down to an eighth) for calls that
    // would not give every lane two chunks otherwise
    const size_t in_per = (2 * static_cast<size_t>(N) + (own_xyz ? 3 * static_cast<size_t>(N) : 0)) * sizeof(T);
    const size_t spread = static_cast<size_t>(B) * in_per / (2 * kPipeLanes);
    const size_t chunk_bytes = std::min(kPipeChunkBytes, std::max(kPipeChunkBytes / 8, spread));
    int64_t chunk = static_cast<int64_t>(chunk_bytes / in_per);
    chunk = std::max<int64_t>(1024, chunk & ~int64_t(255));
    const int64_t nchunks = (B + chunk - 1) / chunk;
    // layout of a lane's page-locked buffer
    struct Layout {
        size_t uv, xyz, R, t, res, rpy, iters, pid, end;
    } lay;
    {
        size_t off = 0;
        auto take = [&off](size_t bytes) {   // 256-byte-aligned pieces, as Carver cuts them
            const size_t at = off;
            off += (bytes + 255) & ~static_cast<size_t>(255);
            return at;
        };
        const size_t n = static_cast<size_t>(chunk);
        lay.uv = take(n * 2 * N * sizeof(T));
        lay.xyz = own_xyz ? take(n * 3 * N * sizeof(T)) : off;
        lay.R = take(n * 9 * sizeof(T));
        lay.t = take(n * 3 * sizeof(T));
        lay.res = h.res ? take(n * sizeof(T)) : off;
        lay.rpy = h.rpy ? take(n * 3 * sizeof(T)) : off;
        lay.iters = h.iters ? take(n * sizeof(int32_t)) : off;
        lay.pid = h.pid ? take(n * sizeof(int32_t)) : off;
        lay.end = off;
    }
    // the lanes' streams know nothing of the caller's: what it holds is waited for here
    HIP_TRY(hipStreamSynchronize(s));

    // a side of the call whose arrays are all page-locked is given to the kernels as it is
    SolveIO<T> m{};   // device addresses of the caller's page-locked arrays
    auto map_side = [&](std::initializer_list<std::pair<const void *, void **>> arrays) {
        for (auto &a : arrays) {
            *a.second = nullptr;
            if (!a.first) continue;
            if (!is_page_locked(a.first) || hipHostGetDevicePointer(a.second, const_cast<void *>(a.first), 0) != hipSuccess ||
                !*a.second) {
                (void)hipGetLastError();
                return false;
            }
        }
        return true;
    };
    void *q_uv, *q_xyz, *q_R, *q_t, *q_res, *q_rpy, *q_iters, *q_pid;
    const bool in_locked = map_side({{h.uv, &q_uv}, {own_xyz ? h.xyz : nullptr, &q_xyz}});
    const bool out_locked = map_side({{h.R, &q_R}, {h.t, &q_t}, {h.res, &q_res}, {h.rpy, &q_rpy}, {h.iters, &q_iters},
                                      {h.pid, &q_pid}});
    m.uv = static_cast<const T *>(q_uv);
    m.xyz = static_cast<const T *>(q_xyz);
    m.R = static_cast<T *>(q_R);
    m.t = static_cast<T *>(q_t);
    m.res = static_cast<T *>(q_res);
    m.rpy = static_cast<T *>(q_rpy);
    m.iters = static_cast<int32_t *>(q_iters);
    m.pid = static_cast<int32_t *>(q_pid);

    // what the lanes report back to this call
    struct Progress {
        std::mutex mu;
        std::condition_variable done;
        int64_t left;
        int rc = PNP_OK;
        char msg[sizeof g_err] = "";
        LaunchRecord launch;
    } progress;
    progress.left = nchunks;
    const uint64_t serial = ++g_call_serial;
    const Tuning tuning = g_tuning;
    const size_t pattern_bytes = own_xyz ? 0 : xyz_elems * sizeof(T);

    auto carry = [&](PipeLane &ln, int64_t k) -> int {
        g_tuning = tuning;
        int rc2 = lane_reserve(ln, lay.end, pattern_bytes);
        if (rc2) return rc2;
        const T *d_shared = nullptr;
        if (!own_xyz) {   // the shared pattern(s) go up once per lane and call
            if (ln.pattern_of != serial) {
                HIP_TRY(hipMemcpyAsync(ln.pattern, h.xyz, pattern_bytes, hipMemcpyHostToDevice, ln.stream));
                ln.pattern_of = serial;
            }
            d_shared = static_cast<const T *>(ln.pattern);
        }
        char *pin = static_cast<char *>(ln.pinned);
        char *mp = static_cast<char *>(ln.pinned_dev);   // the lane's buffer as the device sees it
        const int64_t first = k * chunk;
        const size_t n = static_cast<size_t>(std::min<int64_t>(chunk, B - first));
        SolveIO<T> d;
        if (in_locked) {
            d.uv = m.uv + first * 2 * N;
            d.xyz = own_xyz ? m.xyz + first * 3 * N : d_shared;
        } else {
            copy_streaming(pin + lay.uv, h.uv + first * 2 * N, n * 2 * N * sizeof(T));
            if (own_xyz) copy_streaming(pin + lay.xyz, h.xyz + first * 3 * N, n * 3 * N * sizeof(T));
            d.uv = reinterpret_cast<const T *>(mp + lay.uv);
            d.xyz = own_xyz ? reinterpret_cast<const T *>(mp + lay.xyz) : d_shared;
        }
        if (out_locked) {
            d.R = m.R + first * 9;
            d.t = m.t + first * 3;
            d.res = h.res ? m.res + first : nullptr;
            d.rpy = h.rpy ? m.rpy + first * 3 : nullptr;
            d.iters = h.iters ? m.iters + first : nullptr;
            d.pid = h.pid ? m.pid + first : nullptr;
        } else {
            d.R = reinterpret_cast<T *>(mp + lay.R);
            d.t = reinterpret_cast<T *>(mp + lay.t);
            d.res = h.res ? reinterpret_cast<T *>(mp + lay.res) : nullptr;
            d.rpy = h.rpy ? reinterpret_cast<T *>(mp + lay.rpy) : nullptr;
            d.iters = h.iters ? reinterpret_cast<int32_t *>(mp + lay.iters) : nullptr;
            d.pid = h.pid ? reinterpret_cast<int32_t *>(mp + lay.pid) : nullptr;
        }
        rc2 = run(d, static_cast<int64_t>(n), ln.stream);
        if (rc2) return rc2;
        // the lane's buffer is read below, and reused by the lane's next chunk
        HIP_TRY(hipStreamSynchronize(ln.stream));
        if (out_locked) return PNP_OK;
        copy_streaming(h.R + first * 9, pin + lay.R, n * 9 * sizeof(T));
        copy_streaming(h.t + first * 3, pin + lay.t, n * 3 * sizeof(T));
        if (h.res) copy_streaming(h.res + first, pin + lay.res, n * sizeof(T));
        if (h.rpy) copy_streaming(h.rpy + first * 3, pin + lay.rpy, n * 3 * sizeof(T));
        if (h.iters) copy_streaming(h.iters + first, pin + lay.iters, n * sizeof(int32_t));
        if (h.pid) copy_streaming(h.pid + first, pin + lay.pid, n * sizeof(int32_t));
        return PNP_OK;
    };
    auto job = [&](PipeLane &ln, int64_t k) {
        bool skip;
        {
            std::lock_guard<std::mutex> lock(progress.mu);
            skip = progress.rc != PNP_OK;   // an earlier chunk failed: the call is lost already
        }
        int rc2 = PNP_OK;
        if (!skip) {
            rc2 = carry(ln, k);
            if (rc2 && ln.stream) (void)hipStreamSynchronize(ln.stream);   // nothing of this chunk stays in flight
        }
        std::lock_guard<std::mutex> lock(progress.mu);
        if (rc2 && progress.rc == PNP_OK) {
            progress.rc = rc2;
            std::memcpy(progress.msg, g_err, sizeof progress.msg);   // g_err is the lane thread's own
        }
        if (!skip && !rc2) progress.launch = g_last_launch;
        if (--progress.left == 0) progress.done.notify_all();
    };
    {
        std::lock_guard<std::mutex> lock(pool.mu);
        int64_t queued = 0;
        try {
            for (int64_t k = 0; k < nchunks; ++k, ++queued)
                pool.chunks.emplace_back([&job, k](PipeLane &ln) { job(ln, k); });
        } catch (...) {
            // the chunks queued so far refer to this frame: they are waited for below like any others
            std::lock_guard<std::mutex> plock(progress.mu);
            progress.left -= nchunks - queued;
            progress.rc = PNP_E_HIP;
            snprintf(progress.msg, sizeof progress.msg, "out of memory while queueing the chunks of a host-pointer call");
        }
    }
    pool.wake.notify_all();
    {
        std::unique_lock<std::mutex> lock(progress.mu);
        progress.done.wait(lock, [&] { return progress.left <= 0; });
    }
    if (progress.rc) {
        std::memcpy(g_err, progress.msg, sizeof g_err);
        return progress.rc;
    }
    g_last_launch = progress.launch;
    return PNP_OK;
}

// A host-pointer call: `run` is launched on page-locked host memory (the caller's arrays if they
// are, buffers of the library's own that the arrays are copied through otherwise), which the
// kernels read and write over the bus.  Synchronises the stream before returning.
// run(d, count, stream) solves `count` problems from the arrays d (device addresses) on `stream`.
template <typename T, typename Run>
int with_host_staging(hipStream_t s, int64_t B, int32_t N, size_t xyz_elems, bool own_xyz, const SolveIO<T> &h,
                      Run run)
{
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    if (dev < 0 || dev >= kMaxDevices) return fail(PNP_E_HIP, "device index %d out of range", dev);
    const size_t nB = static_cast<size_t>(B);
    // Every array page-locked: the kernels read and write the caller's memory over the bus (the
    // device addresses of the host arrays), one launch, no staging and no copies.  1M six-point
    // problems: 2.4 ms against 4.2 ms through the chunked copies.
    {
        SolveIO<T> d;
        if (map_page_locked(h, d)) {
            SyncOnExit guard(s);
            return run(d, B, s);
        }
    }
    if ((nB * 2 * N + (own_xyz ? xyz_elems : 0) + nB * 12) * sizeof(T) >= kPipeMinBytes)
        return pipelined_host_call<T>(dev, s, B, N, own_xyz, xyz_elems, h, run);
    Staging &st = g_staging[dev];
    std::lock_guard<std::mutex> lock(st.mu);
    // Smaller calls: one page-locked buffer holds inputs and results, the kernels work on it over
    // the bus (no device staging, no copy operations), the caller's stream carries the launch.
    const size_t total = 256 * 10 + (nB * 2 * N + xyz_elems + nB * (9 + 3 + 1 + 3)) * sizeof(T) +
                         2 * nB * sizeof(int32_t);
    if (total > st.pinned_cap) {
        if (st.pinned) HIP_TRY(hipHostFree(st.pinned));
        st.pinned = nullptr;
        st.pinned_cap = 0;
        const size_t want = std::max(total + total / 4, kSmallCall);
        HIP_TRY(hipHostMalloc(&st.pinned, want, hipHostMallocDefault));
        st.pinned_cap = want;
        st.pinned_dev = nullptr;
        HIP_TRY(hipHostGetDevicePointer(&st.pinned_dev, st.pinned, 0));
    }
    Carver host(st.pinned), dv(st.pinned_dev);
    auto both = [&](auto tag, size_t count, auto *&host_ptr) {
        using E = decltype(tag);
        host_ptr = host.take<E>(count);
        return dv.take<E>(count);
    };
    SolveIO<T> d;
    T *m_uv, *m_xyz, *m_R, *m_t, *m_res = nullptr, *m_rpy = nullptr;
    int32_t *m_iters = nullptr, *m_pid = nullptr;
    d.uv = both(T(), nB * 2 * N, m_uv);
    d.xyz = both(T(), xyz_elems, m_xyz);
    d.R = both(T(), nB * 9, m_R);
    d.t = both(T(), nB * 3, m_t);
    d.res = h.res ? both(T(), nB, m_res) : nullptr;
    d.rpy = h.rpy ? both(T(), nB * 3, m_rpy) : nullptr;
    d.iters = h.iters ? both(int32_t(), nB, m_iters) : nullptr;
    d.pid = h.pid ? both(int32_t(), nB, m_pid) : nullptr;
    std::memcpy(m_uv, h.uv, nB * 2 * N * sizeof(T));
    std::memcpy(m_xyz, h.xyz, xyz_elems * sizeof(T));
    SyncOnExit guard(s);   // every exit below, error or not, leaves nothing in flight on the buffer
    const int rc = run(d, B, s);
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(s));
    guard.armed = false;
    std::memcpy(h.R, m_R, nB * 9 * sizeof(T));
    std::memcpy(h.t, m_t, nB * 3 * sizeof(T));
    if (h.res) std::memcpy(h.res, m_res, nB * sizeof(T));
    if (h.rpy) std::memcpy(h.rpy, m_rpy, nB * 3 * sizeof(T));
    if (h.iters) std::memcpy(h.iters, m_iters, nB * sizeof(int32_t));
    if (h.pid) std::memcpy(h.pid, m_pid, nB * sizeof(int32_t));
    return PNP_OK;
}

// ---- entry-point bodies, shared by the float32 and float64 flavours -----------------------------

// Argument checks and host-side constants shared by every QEIF entry point.
int qeif_prepare(const void *uv, const void *xyz, int64_t xyz_stride, int32_t n_patterns, const double K[9], int64_t B,
                 int32_t N, const pnp_qeif_params *p, const void *R_out, const void *t_out, int32_t ptr_kind,
                 pnp::KinvRows &ki, pnp::QeifParams &q)
{
    g_err[0] = 0;
    int rc = check_common(uv, xyz, xyz_stride, K, B, N, R_out, t_out, ptr_kind);
    if (rc) return rc;
    if (n_patterns < 1) return fail(PNP_E_ARG, "n_patterns = %d must be >= 1", n_patterns);
    if (p && (p->max_it < 1)) return fail(PNP_E_ARG, "max_it = %d must be >= 1", p->max_it);
    // the predict step works with 1/r_quat, 1/r_delta and 1/omega0
    if (p && !(p->q_inv > 0.0 && p->r_quat > 0.0 && p->r_delta > 0.0 && p->omega0 > 0.0))
        return fail(PNP_E_ARG, "q_inv = %g, r_quat = %g, r_delta = %g and omega0 = %g must all be > 0", p->q_inv,
                    p->r_quat, p->r_delta, p->omega0);
    rc = kinv_rows(K, ki);
    if (rc) return rc;
    q = to_device_params(p);
    if (ptr_kind == PNP_PTR_HOST && xyz_stride != 0 && xyz_stride != 3 * static_cast<int64_t>(N))
        return fail(PNP_E_ARG, "host-pointer calls take xyz_problem_stride 0 or 3*N only");
    return PNP_OK;
}

template <typename T>
int qeif_impl(const T *uv, const T *xyz, int64_t xyz_stride, int32_t n_patterns, const double K[9], int64_t B,
              int32_t N, const pnp_qeif_params *p, T *R_out, T *t_out, T *res_out, int32_t *iters_out,
              T *rpy_out, int32_t *pid_out, int32_t ptr_kind, void *hip_stream)
{
    pnp::KinvRows ki;
    pnp::QeifParams q;
    int rc = qeif_prepare(uv, xyz, xyz_stride, n_patterns, K, B, N, p, R_out, t_out, ptr_kind, ki, q);
    if (rc) return rc;
    hipStream_t s = static_cast<hipStream_t>(hip_stream);
    if (B == 0) return PNP_OK;
    if (ptr_kind == PNP_PTR_DEVICE)
        return launch_qeif<T>(s, uv, xyz, xyz_stride, n_patterns, ki, B, N, q, R_out, t_out, res_out, iters_out,
                              rpy_out, pid_out);
    const size_t xyz_elems = xyz_stride ? static_cast<size_t>(B) * 3 * N : static_cast<size_t>(n_patterns) * 3 * N;
    SolveIO<T> h{uv, xyz, R_out, t_out, res_out, rpy_out, iters_out, pid_out};
    return with_host_staging<T>(s, B, N, xyz_elems, xyz_stride != 0, h, [&](const SolveIO<T> &d, int64_t count, hipStream_t on) {
        return launch_qeif<T>(on, d.uv, d.xyz, xyz_stride, n_patterns, ki, count, N, q, d.R, d.t, d.res, d.iters,
                              d.rpy, d.pid, B, /*rows_on_device=*/false);
    });
}

// The host-pointer QEIF call with its staging in the static device array instead of buffers the
// library allocates: no allocation, no free; copies go straight between the caller's arrays and
// the static array.
int qeif_noalloc(const float *uv, const float *xyz, int64_t xyz_stride, const double K[9], int64_t B, int32_t N,
                 const pnp_qeif_params *p, float *R_out, float *t_out, float *res_out, int32_t *iters_out,
                 float *rpy_out, void *hip_stream)
{
    using T = float;
    pnp::KinvRows ki;
    pnp::QeifParams q;
    int rc = qeif_prepare(uv, xyz, xyz_stride, 1, K, B, N, p, R_out, t_out, PNP_PTR_HOST, ki, q);
    if (rc) return rc;
    if (B == 0) return PNP_OK;
    hipStream_t s = static_cast<hipStream_t>(hip_stream);
    const size_t nB = static_cast<size_t>(B);
    const size_t xyz_elems = xyz_stride ? nB * 3 * N : static_cast<size_t>(3) * N;
    // size of the carve-up below, before any pointer is formed: eight pieces, each padded to 256 bytes
    const size_t per_problem = (2 * static_cast<size_t>(N) + (xyz_stride ? 3 * static_cast<size_t>(N) : 0) + 9 + 3 + 1 + 3) *
                                   sizeof(T) + sizeof(int32_t);
    if (nB > (PNP_NOALLOC_SCRATCH_BYTES - 8 * 256 - 3 * static_cast<size_t>(N) * sizeof(T)) / per_problem)
        return fail(PNP_E_ARG,
                    "B = %lld problems of N = %d points need more staging than the static buffer of %zu bytes "
                    "holds (at most %zu problems in this layout): split the batch or use pnp_qeif_batch_f32",
                    static_cast<long long>(B), N, static_cast<size_t>(PNP_NOALLOC_SCRATCH_BYTES),
                    (PNP_NOALLOC_SCRATCH_BYTES - 8 * 256 - 3 * static_cast<size_t>(N) * sizeof(T)) / per_problem);
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    if (dev < 0 || dev >= kMaxDevices) return fail(PNP_E_HIP, "device index %d out of range", dev);
    std::lock_guard<std::mutex> lock(g_scratch[dev].mu);
    void *base = nullptr;
    rc = scratch_address(dev, &base);
    if (rc) return rc;
    Carver c(base);
    T *d_uv = c.take<T>(nB * 2 * N);
    T *d_xyz = c.take<T>(xyz_elems);
    T *d_R = c.take<T>(nB * 9);
    T *d_t = c.take<T>(nB * 3);
    T *d_res = c.take<T>(nB);
    T *d_rpy = c.take<T>(nB * 3);
    int32_t *d_iters = c.take<int32_t>(nB);
    if (c.off > PNP_NOALLOC_SCRATCH_BYTES)
        return fail(PNP_E_ARG, "staging layout of %zu bytes exceeds the static buffer", c.off);
    SyncOnExit guard(s);
    HIP_TRY(hipMemcpyAsync(d_uv, uv, nB * 2 * N * sizeof(T), hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(d_xyz, xyz, xyz_elems * sizeof(T), hipMemcpyHostToDevice, s));
    rc = launch_qeif<T>(s, d_uv, d_xyz, xyz_stride, 1, ki, B, N, q, d_R, d_t, d_res, iters_out ? d_iters : nullptr,
                        rpy_out ? d_rpy : nullptr, nullptr);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(R_out, d_R, nB * 9 * sizeof(T), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpyAsync(t_out, d_t, nB * 3 * sizeof(T), hipMemcpyDeviceToHost, s));
    if (res_out) HIP_TRY(hipMemcpyAsync(res_out, d_res, nB * sizeof(T), hipMemcpyDeviceToHost, s));
    if (rpy_out) HIP_TRY(hipMemcpyAsync(rpy_out, d_rpy, nB * 3 * sizeof(T), hipMemcpyDeviceToHost, s));
    if (iters_out) HIP_TRY(hipMemcpyAsync(iters_out, d_iters, nB * sizeof(int32_t), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    return PNP_OK;
}

template <typename T>
int dlt_impl(const T *uv, const T *xyz, int64_t xyz_stride, const double K[9], int64_t B, int32_t N,
             int32_t num_it, T *R_out, T *t_out, T *res_out, T *rpy_out, int32_t ptr_kind, void *hip_stream)
{
    g_err[0] = 0;
    int rc = check_common(uv, xyz, xyz_stride, K, B, N, R_out, t_out, ptr_kind);
    if (rc) return rc;
    if (num_it < 1) return fail(PNP_E_ARG, "num_it = %d must be >= 1", num_it);
    pnp::KinvRows ki;
    rc = kinv_rows(K, ki);
    if (rc) return rc;
    hipStream_t s = static_cast<hipStream_t>(hip_stream);
    if (B == 0) return PNP_OK;
    if (ptr_kind == PNP_PTR_DEVICE)
        return launch_dlt<T>(s, uv, xyz, xyz_stride, ki, B, N, num_it, R_out, t_out, res_out, rpy_out);
    if (xyz_stride != 0 && xyz_stride != 3 * static_cast<int64_t>(N))
        return fail(PNP_E_ARG, "host-pointer calls take xyz_problem_stride 0 or 3*N only");
    const size_t xyz_elems = xyz_stride ? static_cast<size_t>(B) * 3 * N : static_cast<size_t>(3) * N;
    SolveIO<T> h{uv, xyz, R_out, t_out, res_out, rpy_out, nullptr, nullptr};
    return with_host_staging<T>(s, B, N, xyz_elems, xyz_stride != 0, h, [&](const SolveIO<T> &d, int64_t count, hipStream_t on) {
        return launch_dlt<T>(on, d.uv, d.xyz, xyz_stride, ki, count, N, num_it, d.R, d.t, d.res, d.rpy, B,
                             /*rows_on_device=*/false);
    });
}

template <typename T>
int lm12_impl(const T *uv, const T *xyz, int64_t xyz_stride, const double K[9], int64_t B, int32_t N,
              double lambda, int32_t num_it, T *R_out, T *t_out, T *res_out, T *rpy_out, int32_t ptr_kind,
              void *hip_stream)
{
    g_err[0] = 0;
    int rc = check_common(uv, xyz, xyz_stride, K, B, N, R_out, t_out, ptr_kind);
    if (rc) return rc;
    if (num_it < 1) return fail(PNP_E_ARG, "num_it = %d must be >= 1", num_it);
    if (!(lambda >= 0.0)) return fail(PNP_E_ARG, "lambda must be >= 0");
    pnp::KinvRows ki;
    rc = kinv_rows(K, ki);
    if (rc) return rc;
    hipStream_t s = static_cast<hipStream_t>(hip_stream);
    if (B == 0) return PNP_OK;
    if (ptr_kind == PNP_PTR_DEVICE)
        return launch_lm12<T>(s, uv, xyz, xyz_stride, ki, B, N, lambda, num_it, R_out, t_out, res_out, rpy_out);
    if (xyz_stride != 0 && xyz_stride != 3 * static_cast<int64_t>(N))
        return fail(PNP_E_ARG, "host-pointer calls take xyz_problem_stride 0 or 3*N only");
    const size_t xyz_elems = xyz_stride ? static_cast<size_t>(B) * 3 * N : static_cast<size_t>(3) * N;
    SolveIO<T> h{uv, xyz, R_out, t_out, res_out, rpy_out, nullptr, nullptr};
    return with_host_staging<T>(s, B, N, xyz_elems, xyz_stride != 0, h, [&](const SolveIO<T> &d, int64_t count, hipStream_t on) {
        return launch_lm12<T>(on, d.uv, d.xyz, xyz_stride, ki, count, N, lambda, num_it, d.R, d.t, d.res, d.rpy);
    });
}

template <typename T>
int euler_impl(const T *R, int64_t B, T *rpy, int32_t ptr_kind, void *hip_stream)
{
    g_err[0] = 0;
    if (B < 0) return fail(PNP_E_ARG, "B = %lld is negative", static_cast<long long>(B));
    if (ptr_kind != PNP_PTR_HOST && ptr_kind != PNP_PTR_DEVICE)
        return fail(PNP_E_ARG, "ptr_kind = %d is neither PNP_PTR_HOST nor PNP_PTR_DEVICE", ptr_kind);
    if (B == 0) return PNP_OK;
    if (!R || !rpy) return fail(PNP_E_ARG, "R and rpy must not be NULL");
    hipStream_t s = static_cast<hipStream_t>(hip_stream);
    const int64_t nblk = (B + kBlock - 1) / kBlock;
    if (nblk > 0x7fffffffLL) return fail(PNP_E_ARG, "B = %lld needs too many workgroups", static_cast<long long>(B));
    const dim3 grid(static_cast<unsigned>(nblk));
    if (ptr_kind == PNP_PTR_DEVICE) {
        hipLaunchKernelGGL((pnp::euler_kernel<T>), grid, dim3(kBlock), 0, s, R, B, rpy);
        HIP_TRY(hipGetLastError());
        return PNP_OK;
    }
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    if (dev < 0 || dev >= kMaxDevices) return fail(PNP_E_HIP, "device index %d out of range", dev);
    Staging &st = g_staging[dev];
    std::lock_guard<std::mutex> lock(st.mu);
    const size_t nB = static_cast<size_t>(B);
    int rc = staging_reserve(st, 1024 + nB * 12 * sizeof(T));
    if (rc) return rc;
    Carver c(st.buf);
    SyncOnExit guard(s);
    T *d_R = c.take<T>(nB * 9);
    T *d_rpy = c.take<T>(nB * 3);
    HIP_TRY(hipMemcpyAsync(d_R, R, nB * 9 * sizeof(T), hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL((pnp::euler_kernel<T>), grid, dim3(kBlock), 0, s, d_R, B, d_rpy);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(rpy, d_rpy, nB * 3 * sizeof(T), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    return PNP_OK;
}

template <typename T>
int project_impl(const double *rpy_deg, const double *t, const double *xyz, int64_t xyz_stride, const double K[9],
                 int64_t B, int32_t N, double quantize_q, T *uv_out, int32_t ptr_kind, void *hip_stream)
{
    g_err[0] = 0;
    if (B < 0) return fail(PNP_E_ARG, "B = %lld is negative", static_cast<long long>(B));
    if (N < 1) return fail(PNP_E_ARG, "N = %d must be >= 1", N);
    if (!K) return fail(PNP_E_ARG, "K is NULL");
    if (!(quantize_q >= 0.0)) return fail(PNP_E_ARG, "quantize_q must be >= 0");
    if (ptr_kind != PNP_PTR_HOST && ptr_kind != PNP_PTR_DEVICE)
        return fail(PNP_E_ARG, "ptr_kind = %d is neither PNP_PTR_HOST nor PNP_PTR_DEVICE", ptr_kind);
    if (xyz_stride != 0 && xyz_stride < 3 * static_cast<int64_t>(N))
        return fail(PNP_E_ARG, "xyz_problem_stride = %lld is neither 0 nor >= 3*N", static_cast<long long>(xyz_stride));
    if (B == 0) return PNP_OK;
    if (!rpy_deg || !t || !xyz || !uv_out) return fail(PNP_E_ARG, "rpy_deg, t, xyz and uv_out must not be NULL");
    hipStream_t s = static_cast<hipStream_t>(hip_stream);
    const int64_t nblk = (B + kBlock - 1) / kBlock;
    if (nblk > 0x7fffffffLL) return fail(PNP_E_ARG, "B = %lld needs too many workgroups", static_cast<long long>(B));
    const dim3 grid(static_cast<unsigned>(nblk));
    pnp::Mat3 Km;
    for (int i = 0; i < 9; ++i) Km.m[i] = K[i];
    if (ptr_kind == PNP_PTR_DEVICE) {
        hipLaunchKernelGGL((pnp::project_kernel<T>), grid, dim3(kBlock), 0, s, rpy_deg, t, xyz, xyz_stride, Km, B, N,
                           quantize_q, uv_out);
        HIP_TRY(hipGetLastError());
        return PNP_OK;
    }
    if (xyz_stride != 0 && xyz_stride != 3 * static_cast<int64_t>(N))
        return fail(PNP_E_ARG, "host-pointer calls take xyz_problem_stride 0 or 3*N only");
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    if (dev < 0 || dev >= kMaxDevices) return fail(PNP_E_HIP, "device index %d out of range", dev);
    Staging &st = g_staging[dev];
    std::lock_guard<std::mutex> lock(st.mu);
    const size_t nB = static_cast<size_t>(B);
    const size_t xyz_n = xyz_stride ? nB * 3 * N : static_cast<size_t>(3) * N;
    int rc = staging_reserve(st, 2048 + (nB * 6 + xyz_n + 16) * sizeof(double) + nB * 2 * N * sizeof(T));
    if (rc) return rc;
    Carver c(st.buf);
    SyncOnExit guard(s);
    double *d_rpy = c.take<double>(nB * 3);
    double *d_t = c.take<double>(nB * 3);
    double *d_xyz = c.take<double>(xyz_n);
    T *d_uv = c.take<T>(nB * 2 * N);
    HIP_TRY(hipMemcpyAsync(d_rpy, rpy_deg, nB * 3 * sizeof(double), hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(d_t, t, nB * 3 * sizeof(double), hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(d_xyz, xyz, xyz_n * sizeof(double), hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL((pnp::project_kernel<T>), grid, dim3(kBlock), 0, s, d_rpy, d_t, d_xyz, xyz_stride, Km, B, N,
                       quantize_q, d_uv);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(uv_out, d_uv, nB * 2 * N * sizeof(T), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    return PNP_OK;
}

template <typename T>
int compare_impl(const T *R_est, const T *t_est, const T *uv_obs, const double *rpy_gt_deg,
                 const double *distance_gt_cm, const double *xyz, const double K[9], int64_t B, int32_t M,
                 const double bound[4], int32_t normalize, T *drpy_err, int32_t *pass_mask, T *lm_err,
                 int32_t *lm_key, int32_t ptr_kind, void *hip_stream)
{
    g_err[0] = 0;
    if (B < 0) return fail(PNP_E_ARG, "B = %lld is negative", static_cast<long long>(B));
    if (M < 1) return fail(PNP_E_ARG, "M = %d must be >= 1", M);
    if (!K) return fail(PNP_E_ARG, "K is NULL");
    if (ptr_kind != PNP_PTR_HOST && ptr_kind != PNP_PTR_DEVICE)
        return fail(PNP_E_ARG, "ptr_kind = %d is neither PNP_PTR_HOST nor PNP_PTR_DEVICE", ptr_kind);
    if (B == 0) return PNP_OK;
    if (!R_est || !t_est || !rpy_gt_deg || !distance_gt_cm || !xyz)
        return fail(PNP_E_ARG, "R_est, t_est, rpy_gt_deg, distance_gt_cm and xyz must not be NULL");
    if (!drpy_err && !pass_mask && !lm_err && !lm_key) return fail(PNP_E_ARG, "no output requested");
    hipStream_t s = static_cast<hipStream_t>(hip_stream);
    const int64_t nblk = (B + kBlock - 1) / kBlock;
    if (nblk > 0x7fffffffLL) return fail(PNP_E_ARG, "B = %lld needs too many workgroups", static_cast<long long>(B));
    const dim3 grid(static_cast<unsigned>(nblk));
    pnp::Mat3 Km;
    pnp::PassBounds bd;
    for (int i = 0; i < 9; ++i) Km.m[i] = K[i];
    for (int i = 0; i < 4; ++i) bd.v[i] = bound ? bound[i] : 10.0;
    if (ptr_kind == PNP_PTR_DEVICE) {
        hipLaunchKernelGGL((pnp::compare_kernel<T>), grid, dim3(kBlock), 0, s, R_est, t_est, uv_obs, rpy_gt_deg,
                           distance_gt_cm, xyz, Km, B, M, bd, normalize, drpy_err, pass_mask, lm_err, lm_key);
        HIP_TRY(hipGetLastError());
        return PNP_OK;
    }
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    if (dev < 0 || dev >= kMaxDevices) return fail(PNP_E_HIP, "device index %d out of range", dev);
    Staging &st = g_staging[dev];
    std::lock_guard<std::mutex> lock(st.mu);
    const size_t nB = static_cast<size_t>(B);
    int rc = staging_reserve(st, 4096 + (nB * 4 + 3 * M) * sizeof(double) + nB * (12 + 2 * M + 10) * sizeof(T) +
                                     nB * 4 * sizeof(int32_t));
    if (rc) return rc;
    Carver c(st.buf);
    SyncOnExit guard(s);
    double *d_rpy = c.take<double>(nB * 3);
    double *d_dist = c.take<double>(nB);
    double *d_xyz = c.take<double>(static_cast<size_t>(3) * M);
    T *d_R = c.take<T>(nB * 9);
    T *d_t = c.take<T>(nB * 3);
    T *d_uv = uv_obs ? c.take<T>(nB * 2 * M) : nullptr;
    T *d_err = drpy_err ? c.take<T>(nB * 4) : nullptr;
    T *d_lm = lm_err ? c.take<T>(nB * 6) : nullptr;
    int32_t *d_mask = pass_mask ? c.take<int32_t>(nB) : nullptr;
    int32_t *d_key = lm_key ? c.take<int32_t>(nB * 3) : nullptr;
    HIP_TRY(hipMemcpyAsync(d_rpy, rpy_gt_deg, nB * 3 * sizeof(double), hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(d_dist, distance_gt_cm, nB * sizeof(double), hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(d_xyz, xyz, static_cast<size_t>(3) * M * sizeof(double), hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(d_R, R_est, nB * 9 * sizeof(T), hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(d_t, t_est, nB * 3 * sizeof(T), hipMemcpyHostToDevice, s));
    if (uv_obs) HIP_TRY(hipMemcpyAsync(d_uv, uv_obs, nB * 2 * M * sizeof(T), hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL((pnp::compare_kernel<T>), grid, dim3(kBlock), 0, s, d_R, d_t, d_uv, d_rpy, d_dist, d_xyz, Km, B,
                       M, bd, normalize, d_err, d_mask, d_lm, d_key);
    HIP_TRY(hipGetLastError());
    if (drpy_err) HIP_TRY(hipMemcpyAsync(drpy_err, d_err, nB * 4 * sizeof(T), hipMemcpyDeviceToHost, s));
    if (lm_err) HIP_TRY(hipMemcpyAsync(lm_err, d_lm, nB * 6 * sizeof(T), hipMemcpyDeviceToHost, s));
    if (pass_mask) HIP_TRY(hipMemcpyAsync(pass_mask, d_mask, nB * sizeof(int32_t), hipMemcpyDeviceToHost, s));
    if (lm_key) HIP_TRY(hipMemcpyAsync(lm_key, d_key, nB * 3 * sizeof(int32_t), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    return PNP_OK;
}

#include "pnp_gather.hpp"

}  // namespace

// ================================================================================================
// C ABI
// ================================================================================================

extern "C" {

void pnp_qeif_default_params(pnp_qeif_params *p)
{
    if (!p) return;
    p->q_inv = (225.68 * 225.68) / 9.0;
    p->r_quat = 0.1;
    p->r_delta = 0.01;
    p->omega0 = 1e-5;
    p->max_it = 14;
    p->rel_tol = 1e-2;
    p->res_old0 = 1e-7;
}

int pnp_device_count(void)
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e == hipErrorNoDevice) return 0;
    if (e != hipSuccess) return fail(PNP_E_HIP, "hipGetDeviceCount failed: %s", hipGetErrorString(e));
    return n;
}

int pnp_init(int device)
{
    const int n = pnp_device_count();
    if (n < 0) return n;
    if (n == 0) return fail(PNP_E_NODEVICE, "no HIP device is visible");
    if (device < 0 || device >= n) return fail(PNP_E_ARG, "device %d outside [0, %d)", device, n);
    HIP_TRY(hipSetDevice(device));
    HIP_TRY(hipFree(nullptr));  // forces context creation
    // this device's address of the static staging array, so that the allocation-free entry point
    // finds it resolved
    if (device >= kMaxDevices) return fail(PNP_E_HIP, "device index %d out of range", device);
    (void)lines_kernel_allowed<float>();
    (void)lines_kernel_allowed<double>();
    std::lock_guard<std::mutex> lock(g_scratch[device].mu);
    void *unused = nullptr;
    return scratch_address(device, &unused);
}

// ---- float32 problem arrays ----------------------------------------------------------------------

int pnp_qeif_batch_f32(const float *uv, const float *xyz, int64_t xyz_problem_stride, const double K[9],
                       int64_t B, int32_t N, const pnp_qeif_params *p, float *R_out, float *t_out,
                       float *res_out, int32_t *iters_out, float *rpy_out, int32_t ptr_kind, void *hip_stream)
{
    return qeif_impl<float>(uv, xyz, xyz_problem_stride, 1, K, B, N, p, R_out, t_out, res_out, iters_out, rpy_out,
                            nullptr, ptr_kind, hip_stream);
}

int pnp_qeif_batch_noalloc_f32(const float *uv, const float *xyz, int64_t xyz_problem_stride, const double K[9],
                               int64_t B, int32_t N, const pnp_qeif_params *p, float *R_out, float *t_out,
                               float *res_out, int32_t *iters_out, float *rpy_out, void *hip_stream)
{
    return qeif_noalloc(uv, xyz, xyz_problem_stride, K, B, N, p, R_out, t_out, res_out, iters_out, rpy_out,
                        hip_stream);
}

int pnp_qeif_multi_batch_f32(const float *uv, const float *xyz, int32_t n_patterns, const double K[9], int64_t B,
                             int32_t N, const pnp_qeif_params *p, float *R_out, float *t_out, float *res_out,
                             int32_t *iters_out, float *rpy_out, int32_t *pattern_id_out, int32_t ptr_kind,
                             void *hip_stream)
{
    return qeif_impl<float>(uv, xyz, 0, n_patterns, K, B, N, p, R_out, t_out, res_out, iters_out, rpy_out,
                            pattern_id_out, ptr_kind, hip_stream);
}

int pnp_dlt_batch_f32(const float *uv, const float *xyz, int64_t xyz_problem_stride, const double K[9], int64_t B,
                      int32_t N, int32_t num_it, float *R_out, float *t_out, float *res_out, float *rpy_out,
                      int32_t ptr_kind, void *hip_stream)
{
    return dlt_impl<float>(uv, xyz, xyz_problem_stride, K, B, N, num_it, R_out, t_out, res_out, rpy_out, ptr_kind,
                           hip_stream);
}

int pnp_lm12_batch_f32(const float *uv, const float *xyz, int64_t xyz_problem_stride, const double K[9], int64_t B,
                       int32_t N, double lambda, int32_t num_it, float *R_out, float *t_out, float *res_out,
                       float *rpy_out, int32_t ptr_kind, void *hip_stream)
{
    return lm12_impl<float>(uv, xyz, xyz_problem_stride, K, B, N, lambda, num_it, R_out, t_out, res_out, rpy_out,
                            ptr_kind, hip_stream);
}

int pnp_euler_deg_f32(const float *R, int64_t B, float *rpy, int32_t ptr_kind, void *hip_stream)
{
    return euler_impl<float>(R, B, rpy, ptr_kind, hip_stream);
}

// ---- float64 problem arrays ----------------------------------------------------------------------

int pnp_qeif_batch_f64(const double *uv, const double *xyz, int64_t xyz_problem_stride, const double K[9],
                       int64_t B, int32_t N, const pnp_qeif_params *p, double *R_out, double *t_out,
                       double *res_out, int32_t *iters_out, double *rpy_out, int32_t ptr_kind, void *hip_stream)
{
    return qeif_impl<double>(uv, xyz, xyz_problem_stride, 1, K, B, N, p, R_out, t_out, res_out, iters_out, rpy_out,
                             nullptr, ptr_kind, hip_stream);
}

int pnp_qeif_multi_batch_f64(const double *uv, const double *xyz, int32_t n_patterns, const double K[9], int64_t B,
                             int32_t N, const pnp_qeif_params *p, double *R_out, double *t_out, double *res_out,
                             int32_t *iters_out, double *rpy_out, int32_t *pattern_id_out, int32_t ptr_kind,
                             void *hip_stream)
{
    return qeif_impl<double>(uv, xyz, 0, n_patterns, K, B, N, p, R_out, t_out, res_out, iters_out, rpy_out,
                             pattern_id_out, ptr_kind, hip_stream);
}

int pnp_dlt_batch_f64(const double *uv, const double *xyz, int64_t xyz_problem_stride, const double K[9],
                      int64_t B, int32_t N, int32_t num_it, double *R_out, double *t_out, double *res_out,
                      double *rpy_out, int32_t ptr_kind, void *hip_stream)
{
    return dlt_impl<double>(uv, xyz, xyz_problem_stride, K, B, N, num_it, R_out, t_out, res_out, rpy_out, ptr_kind,
                            hip_stream);
}

int pnp_lm12_batch_f64(const double *uv, const double *xyz, int64_t xyz_problem_stride, const double K[9],
                       int64_t B, int32_t N, double lambda, int32_t num_it, double *R_out, double *t_out,
                       double *res_out, double *rpy_out, int32_t ptr_kind, void *hip_stream)
{
    return lm12_impl<double>(uv, xyz, xyz_problem_stride, K, B, N, lambda, num_it, R_out, t_out, res_out, rpy_out,
                             ptr_kind, hip_stream);
}

int pnp_euler_deg_f64(const double *R, int64_t B, double *rpy, int32_t ptr_kind, void *hip_stream)
{
    return euler_impl<double>(R, B, rpy, ptr_kind, hip_stream);
}

// ---- generator -----------------------------------------------------------------------------------

int pnp_project_batch_f64(const double *rpy_deg, const double *t, const double *xyz, int64_t xyz_problem_stride,
                          const double K[9], int64_t B, int32_t N, double quantize_q, void *uv_out,
                          int32_t uv_is_f64, int32_t ptr_kind, void *hip_stream)
{
    if (uv_is_f64)
        return project_impl<double>(rpy_deg, t, xyz, xyz_problem_stride, K, B, N, quantize_q,
                                    static_cast<double *>(uv_out), ptr_kind, hip_stream);
    return project_impl<float>(rpy_deg, t, xyz, xyz_problem_stride, K, B, N, quantize_q,
                               static_cast<float *>(uv_out), ptr_kind, hip_stream);
}

// ---- comparison epilogue ------------------------------------------------------------------------------

int pnp_compare_batch_f32(const float *R_est, const float *t_est, const float *uv_obs, const double *rpy_gt_deg,
                          const double *distance_gt_cm, const double *xyz, const double K[9], int64_t B, int32_t M,
                          const double bound[4], int32_t normalize, float *drpy_err, int32_t *pass_mask,
                          float *lm_err, int32_t *lm_key, int32_t ptr_kind, void *hip_stream)
{
    return compare_impl<float>(R_est, t_est, uv_obs, rpy_gt_deg, distance_gt_cm, xyz, K, B, M, bound, normalize,
                               drpy_err, pass_mask, lm_err, lm_key, ptr_kind, hip_stream);
}

int pnp_compare_batch_f64(const double *R_est, const double *t_est, const double *uv_obs, const double *rpy_gt_deg,
                          const double *distance_gt_cm, const double *xyz, const double K[9], int64_t B, int32_t M,
                          const double bound[4], int32_t normalize, double *drpy_err, int32_t *pass_mask,
                          double *lm_err, int32_t *lm_key, int32_t ptr_kind, void *hip_stream)
{
    return compare_impl<double>(R_est, t_est, uv_obs, rpy_gt_deg, distance_gt_cm, xyz, K, B, M, bound, normalize,
                                drpy_err, pass_mask, lm_err, lm_key, ptr_kind, hip_stream);
}

// ---- optional exchange between ranks ---------------------------------------------------------------

int pnp_comm_unique_id(void *id_out) { return comm_unique_id_impl(id_out); }

int pnp_comm_init(const void *id, int32_t rank, int32_t world, void **comm_out)
{
    return comm_init_impl(id, rank, world, comm_out);
}

int pnp_comm_destroy(void *comm) { return comm_destroy_impl(comm); }

int pnp_gather_results(const void *send, void *recv, int64_t bytes_per_rank, int32_t root, int32_t rank, int32_t world,
                       void *comm, void *hip_stream)
{
    return gather_impl(send, recv, bytes_per_rank, root, rank, world, comm, static_cast<hipStream_t>(hip_stream));
}

// ---- tuning and introspection -----------------------------------------------------------------------

int pnp_set_lanes_per_problem(int32_t lanes)
{
    g_err[0] = 0;
    if (lanes != 0 && (lanes < 1 || lanes > 64 || (lanes & (lanes - 1)) != 0))
        return fail(PNP_E_ARG, "lanes = %d is not 0 or a power of two in [1, 64]", lanes);
    g_tuning.forced_lpi = lanes;
    return PNP_OK;
}

int pnp_set_qeif_form(int32_t form)
{
    g_err[0] = 0;
    if (form != PNP_QEIF_FORM_AUTO && form != PNP_QEIF_FORM_POINTS && form != PNP_QEIF_FORM_MOMENTS)
        return fail(PNP_E_ARG, "form = %d is not one of PNP_QEIF_FORM_*", form);
    g_tuning.qeif_form = form;
    return PNP_OK;
}

int pnp_get_qeif_form(int64_t B, int32_t N)
{
    return use_moment_form(B, N, 1) ? PNP_QEIF_FORM_MOMENTS : PNP_QEIF_FORM_POINTS;
}

const char *pnp_last_kernel_name(void) { return g_last_launch.kernel; }

void pnp_last_launch(int64_t *grid, int32_t *block, int32_t *lds_bytes_out)
{
    if (grid) *grid = g_last_launch.grid;
    if (block) *block = g_last_launch.block;
    if (lds_bytes_out) *lds_bytes_out = g_last_launch.lds;
}

int pnp_host_lane_threads(int device)
{
    if (device < 0 || device >= kMaxDevices) return fail(PNP_E_ARG, "device %d outside [0, %d)", device, kMaxDevices);
    std::lock_guard<std::mutex> lock(g_lanes[device].mu);
    return static_cast<int>(g_lanes[device].threads.size());
}

int pnp_get_lanes_per_problem(int64_t B, int32_t N, int64_t xyz_problem_stride)
{
    if (N < 1 || N > kMaxPoints) return fail(PNP_E_ARG, "N = %d outside [1, %d]", N, kMaxPoints);
    return choose_lpi(B, N, xyz_problem_stride != 0, 1, sizeof(float));
}

int pnp_max_points(void) { return kMaxPoints; }

const char *pnp_last_error(void) { return g_err; }

const char *pnp_version(void)
{
#ifdef PNP_AB_SWITCHES
    return "pnp_hip 0.5 (gfx950; A/B build: reads PNP_ROWS_STAGED and PNP_MOMENT_PLAIN)";
#else
    return "pnp_hip 0.5 (gfx950)";
#endif
}

}  // extern "C"
