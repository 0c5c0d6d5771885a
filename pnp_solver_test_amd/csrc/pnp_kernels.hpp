// Device code of the batched PnP solvers for CDNA4 (gfx950): wave64, LDS-staged
// correspondences, register-resident 6x6 / 4x4 algebra, no MFMA (the per-problem matrices
// are far too small to form a dense contraction).
//
// Work decomposition.  A workgroup of 128 threads owns PB = 128/LPI consecutive problems,
// LPI ("lanes per problem") being a power of two between 1 and 64.  The workgroup first
// copies its problems' correspondences (one contiguous range of the uv array, and of the
// xyz array when every problem has its own pattern) into LDS with coalesced 16-byte loads.
// The LPI lanes of a problem then split its N points between them, reduce their partial
// normal-equation sums with xor-butterfly wavefront shuffles, and each lane carries the
// small dense algebra redundantly in registers, so no lane ever waits on another.
// LPI = 64 is one wavefront per problem (latency-optimal for few, large problems);
// LPI = 1 is one lane per problem (throughput-optimal for millions of six-point problems).
// Six-point problems with one lane each need no LDS image at all when their rows are aligned and
// in device memory: qeif_rows6_kernel and dlt_rows6_kernel let every lane read its own rows.
//
// All arithmetic is float64.  The problem arrays are float32 (widened on load, results rounded
// once on store) or float64; the entry points come in both flavours.
#pragma once

#include "pnp_common.hpp"
#include "pnp_dlt.hpp"
#include "pnp_lm12.hpp"
#include "pnp_qeif.hpp"

namespace pnp {

// ------------------------------------------------------------------------------------------
// kernels
// ------------------------------------------------------------------------------------------

// The kernels below are templated on the element type T of the problem arrays (float or
// double): pixels, pattern, R, t, residual and angles are read and written as T, and all
// arithmetic in between is double.

// Point access into the workgroup's LDS image (stored as loaded; widened on every use).
template <typename T>
struct LdsPoints {
    const T *uv;   // this problem's (N,2)
    const T *xyz;  // this problem's (N,3)
    KinvRows ki;
    __device__ __forceinline__ void operator()(int i, double &X, double &Y, double &Z, double &bx,
                                               double &by) const
    {
        const double u = static_cast<double>(uv[2 * i]), v = static_cast<double>(uv[2 * i + 1]);
        bx = __builtin_fma(ki.a00, u, __builtin_fma(ki.a01, v, ki.a02));
        by = __builtin_fma(ki.a10, u, __builtin_fma(ki.a11, v, ki.a12));
        X = static_cast<double>(xyz[3 * i]);
        Y = static_cast<double>(xyz[3 * i + 1]);
        Z = static_cast<double>(xyz[3 * i + 2]);
    }
};

// The same points pulled into registers once (small compile-time N, one lane per problem).
template <int NCT>
struct RegPoints {
    double X[NCT], Y[NCT], Z[NCT], bx[NCT], by[NCT];
    template <typename Src>
    __device__ __forceinline__ explicit RegPoints(const Src &src)
    {
#pragma unroll
        for (int i = 0; i < NCT; ++i) src(i, X[i], Y[i], Z[i], bx[i], by[i]);
    }
    template <typename Src>
    __device__ __forceinline__ void set_pattern(const Src &src)
    {
#pragma unroll
        for (int i = 0; i < NCT; ++i) {
            double bx_, by_;
            src(i, X[i], Y[i], Z[i], bx_, by_);
        }
    }
    __device__ __forceinline__ void operator()(int i, double &X_, double &Y_, double &Z_,
                                               double &bx_, double &by_) const
    {
        X_ = X[i];
        Y_ = Y[i];
        Z_ = Z[i];
        bx_ = bx[i];
        by_ = by[i];
    }
};

// The six points of one problem read by the lane that solves it, straight from memory: the pixel
// row (12 elements) in 16-byte pieces, the pattern row (18 elements) in pieces of two elements.
// Neighbouring lanes read neighbouring rows, so the pieces of one load instruction lie 48 / 72
// bytes apart (float32) and the instructions of one row share their cache lines; nothing passes
// through LDS and no lane waits at a barrier.  The pixel row must be 16-byte aligned, the pattern
// row aligned to two elements.
template <typename T>
struct OwnRows6 {
    static constexpr int UE = 16 / static_cast<int>(sizeof(T));
    typedef T UvPiece __attribute__((ext_vector_type(UE)));
    typedef T XyzPiece __attribute__((ext_vector_type(2)));
    T uv[12], xyz[18];
    KinvRows ki;
    __device__ __forceinline__ OwnRows6(const T *uv_row, const T *xyz_row, const KinvRows &k) : ki(k)
    {
        const UvPiece *pu = reinterpret_cast<const UvPiece *>(uv_row);
#pragma unroll
        for (int i = 0; i < 12 / UE; ++i) {
            const UvPiece q = pu[i];
#pragma unroll
            for (int e = 0; e < UE; ++e) uv[UE * i + e] = q[e];
        }
        const XyzPiece *px = reinterpret_cast<const XyzPiece *>(xyz_row);
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            const XyzPiece q = px[i];
            xyz[2 * i] = q[0];
            xyz[2 * i + 1] = q[1];
        }
    }
    __device__ __forceinline__ void operator()(int i, double &X, double &Y, double &Z, double &bx,
                                               double &by) const
    {
        const double u = static_cast<double>(uv[2 * i]), v = static_cast<double>(uv[2 * i + 1]);
        bx = __builtin_fma(ki.a00, u, __builtin_fma(ki.a01, v, ki.a02));
        by = __builtin_fma(ki.a10, u, __builtin_fma(ki.a11, v, ki.a12));
        X = static_cast<double>(xyz[3 * i]);
        Y = static_cast<double>(xyz[3 * i + 1]);
        Z = static_cast<double>(xyz[3 * i + 2]);
    }
};

// What a lane knows about its place in the workgroup after staging.
template <typename T>
struct Slot {
    int64_t problem;     // global problem index this lane works on
    bool store;          // this lane writes the problem's outputs
    int lane_in_group;   // 0..LPI-1
    LdsPoints<T> pts;    // view of the problem's points (pattern 0)
    int xyz_step;        // elements from one shared pattern to the next in LDS (3N)
};

// Stages the workgroup's problems into dynamic LDS and returns this lane's slot.
// LDS layout: [uv of PB problems: PB*2N elements][xyz: PB*3N elements per-problem, or
// n_patterns*3N elements when the patterns are shared by all problems]
template <int LPI, int BLOCK, typename T>
__device__ __forceinline__ Slot<T> stage_problems(const T *uv, const T *xyz, int64_t xyz_stride,
                                                  int n_patterns, const KinvRows &ki, int64_t B, int N)
{
    extern __shared__ __align__(16) unsigned char lds_raw[];
    T *lds = reinterpret_cast<T *>(lds_raw);
    constexpr int PB = BLOCK / LPI;
    const int tid = threadIdx.x;
    const int64_t first = static_cast<int64_t>(blockIdx.x) * PB;
    const int nprob = static_cast<int>(min(static_cast<int64_t>(PB), B - first));
    T *s_uv = lds;
    T *s_xyz = lds + PB * 2 * N;
    stage_elems<BLOCK>(s_uv, uv + first * 2 * N, nprob * 2 * N, tid);
    if (xyz_stride == 0) {
        stage_elems<BLOCK>(s_xyz, xyz, n_patterns * 3 * N, tid);
    } else if (xyz_stride == 3 * static_cast<int64_t>(N)) {
        stage_elems<BLOCK>(s_xyz, xyz + first * xyz_stride, nprob * 3 * N, tid);
    } else {
        for (int i = tid; i < nprob * 3 * N; i += BLOCK) {
            const int g = i / (3 * N);
            s_xyz[i] = xyz[(first + g) * xyz_stride + (i - g * 3 * N)];
        }
    }
    __syncthreads();
    Slot<T> s;
    int g = tid / LPI;
    s.lane_in_group = tid % LPI;
    s.store = (g < nprob) && (s.lane_in_group == 0);
    if (g >= nprob) g = nprob - 1;  // idle groups of the last workgroup redo its last problem
    s.problem = first + g;
    s.pts.uv = s_uv + g * 2 * N;
    s.pts.xyz = (xyz_stride == 0) ? s_xyz : s_xyz + g * 3 * N;
    s.pts.ki = ki;
    s.xyz_step = 3 * N;
    return s;
}

template <typename T>
__device__ __forceinline__ void store_pose(int64_t problem, const double (&R)[9], const double (&t)[3],
                                           T *R_out, T *t_out)
{
#pragma unroll
    for (int i = 0; i < 9; ++i) R_out[problem * 9 + i] = static_cast<T>(R[i]);
#pragma unroll
    for (int i = 0; i < 3; ++i) t_out[problem * 3 + i] = static_cast<T>(t[i]);
}

template <typename T>
__device__ __forceinline__ void store_rpy(int64_t problem, const double (&R)[9], T *rpy_out)
{
    double roll, yaw, pitch;
    euler_deg(R, roll, yaw, pitch);
    rpy_out[problem * 3 + 0] = static_cast<T>(roll);
    rpy_out[problem * 3 + 1] = static_cast<T>(yaw);
    rpy_out[problem * 3 + 2] = static_cast<T>(pitch);
}

// QEIF over B problems.  n_patterns > 1 (shared patterns only, MULTI) solves every problem
// against each pattern in turn and keeps the first one with the strictly smallest residual;
// the single-pattern instantiation carries no second solve and so needs fewer registers.
// NCT > 0 requires LPI == 1 and N == NCT.
template <int LPI, int NCT, typename T, bool MULTI>
__global__ __launch_bounds__(kSolveBlock<T>, PNP_QEIF_WAVES) void qeif_kernel(
    const T *__restrict__ uv, const T *__restrict__ xyz, int64_t xyz_stride, int n_patterns, KinvRows ki,
    int64_t B, int N, QeifParams p, T *__restrict__ R_out, T *__restrict__ t_out, T *__restrict__ res_out,
    int32_t *__restrict__ iters_out, T *__restrict__ rpy_out, int32_t *__restrict__ pattern_id_out)
{
    Slot<T> s = stage_problems<LPI, kSolveBlock<T>>(uv, xyz, xyz_stride, n_patterns, ki, B, N);
    double R[9], t[3], res;
    int iters, best = 0;
    if constexpr (NCT > 0) {
        RegPoints<NCT> rp(s.pts);
        qeif_solve<1, NCT>(rp, N, 0, p, R, t, res, iters);
        for (int k = 1; MULTI && k < n_patterns; ++k) {
            s.pts.xyz += s.xyz_step;
            rp.set_pattern(s.pts);
            double R2[9], t2[3], res2;
            int it2;
            qeif_solve<1, NCT>(rp, N, 0, p, R2, t2, res2, it2);
            if (res2 < res) {
#pragma unroll
                for (int i = 0; i < 9; ++i) R[i] = R2[i];
#pragma unroll
                for (int i = 0; i < 3; ++i) t[i] = t2[i];
                res = res2;
                iters = it2;
                best = k;
            }
        }
    } else {
        qeif_solve<LPI, 0>(s.pts, N, s.lane_in_group, p, R, t, res, iters);
        for (int k = 1; MULTI && k < n_patterns; ++k) {
            s.pts.xyz += s.xyz_step;
            double R2[9], t2[3], res2;
            int it2;
            qeif_solve<LPI, 0>(s.pts, N, s.lane_in_group, p, R2, t2, res2, it2);
            if (res2 < res) {
#pragma unroll
                for (int i = 0; i < 9; ++i) R[i] = R2[i];
#pragma unroll
                for (int i = 0; i < 3; ++i) t[i] = t2[i];
                res = res2;
                iters = it2;
                best = k;
            }
        }
    }
    if (s.store) {
        store_pose(s.problem, R, t, R_out, t_out);
        if (res_out) res_out[s.problem] = static_cast<T>(res);
        if (iters_out) iters_out[s.problem] = iters;
        if (rpy_out) store_rpy(s.problem, R, rpy_out);
        if (pattern_id_out) pattern_id_out[s.problem] = best;
    }
}

// Workgroup of the kernels that read their rows themselves: one wavefront.  With no LDS image to
// share and no barrier, the smallest unit schedules best (1M problems: 0.1576 ms against 0.1587
// with two wavefronts and 0.1606 with four).
constexpr int kRowsBlock = 64;

// qeif_kernel<1, 6, T, false> without the LDS image: six points, one lane per problem, one pattern
// (per problem, xyz_stride >= 18, or shared, xyz_stride == 0), rows in device memory and aligned as
// OwnRows6 asks.  Same arithmetic on the same values, so the same bits; the launch is shorter by
// the staging loop, the barrier and the LDS round trip.
template <typename T>
__global__ __launch_bounds__(kRowsBlock, PNP_QEIF_WAVES) void qeif_rows6_kernel(
    const T *__restrict__ uv, const T *__restrict__ xyz, int64_t xyz_stride, KinvRows ki, int64_t B, QeifParams p,
    T *__restrict__ R_out, T *__restrict__ t_out, T *__restrict__ res_out, int32_t *__restrict__ iters_out,
    T *__restrict__ rpy_out, int32_t *__restrict__ pattern_id_out)
{
    const int64_t slot = static_cast<int64_t>(blockIdx.x) * kRowsBlock + threadIdx.x;
    const int64_t problem = min(slot, B - 1);  // idle lanes of the last workgroup redo the last problem
    const RegPoints<6> rp(OwnRows6<T>(uv + problem * 12, xyz + problem * xyz_stride, ki));
    double R[9], t[3], res;
    int iters;
    qeif_solve<1, 6>(rp, 6, 0, p, R, t, res, iters);
    if (slot < B) {
        store_pose(problem, R, t, R_out, t_out);
        if (res_out) res_out[problem] = static_cast<T>(res);
        if (iters_out) iters_out[problem] = iters;
        if (rpy_out) store_rpy(problem, R, rpy_out);
        if (pattern_id_out) pattern_id_out[problem] = 0;
    }
}

// The iterations of the moment form for one problem (one lane) from its accumulated sums, and the
// stores of its results.  The residual of the stopping rule falls back to the points themselves
// (exact_ss) where the moment expression would be round-off.
template <typename T>
__device__ __forceinline__ void moment_iterations(Moments &M, const T *__restrict__ uv, const T *__restrict__ xyz,
                                                  int64_t xyz_stride, const KinvRows &ki, int64_t problem, int N,
                                                  const QeifParams &p, T *__restrict__ R_out, T *__restrict__ t_out,
                                                  T *__restrict__ res_out, int32_t *__restrict__ iters_out,
                                                  T *__restrict__ rpy_out)
{
    const T *my_uv = uv + problem * 2 * N;
    const T *my_xyz = xyz + problem * xyz_stride;
    auto exact_ss = [&](const double (&x)[6]) {
        double phi[9], gamma, ss = 0.0;
        gamma_matrix(x, phi, gamma);
        for (int i = 0; i < N; ++i) {
            const double u = static_cast<double>(my_uv[2 * i]), v = static_cast<double>(my_uv[2 * i + 1]);
            const double X = static_cast<double>(my_xyz[3 * i]), Y = static_cast<double>(my_xyz[3 * i + 1]);
            const double Z = static_cast<double>(my_xyz[3 * i + 2]);
            const double bx = __builtin_fma(ki.a00, u, __builtin_fma(ki.a01, v, ki.a02));
            const double by = __builtin_fma(ki.a10, u, __builtin_fma(ki.a11, v, ki.a12));
            const double p1 = __builtin_fma(X, phi[0], __builtin_fma(Y, phi[1], Z * phi[2]));
            const double p2 = __builtin_fma(X, phi[3], __builtin_fma(Y, phi[4], Z * phi[5]));
            const double p3 = __builtin_fma(X, phi[6], __builtin_fma(Y, phi[7], Z * phi[8]));
            const double rx = (bx - x[4]) - __builtin_fma(-bx, p3, p1);
            const double ry = (by - x[5]) - __builtin_fma(-by, p3, p2);
            ss = __builtin_fma(rx, rx, __builtin_fma(ry, ry, ss));
        }
        return ss;
    };
    QeifState st;
    qeif_init(st, p, p.q_inv);
    const QeifRule rule(p);
    const QeifNoise noise(p);
    moments_scale(M, p.q_inv);
    const double n = p.q_inv * static_cast<double>(N);
    while (st.k < p.max_it) {
        if (qeif_moment_iterate(st, M, n, p, rule, noise, exact_ss)) break;
    }
    double R[9], t[3];
    qeif_finish(st, R, t);
    store_pose(problem, R, t, R_out, t_out);
    if (res_out) res_out[problem] = static_cast<T>(qeif_residual(st, p.c_inv));
    if (iters_out) iters_out[problem] = st.k;
    if (rpy_out) store_rpy(problem, R, rpy_out);
}

// Points per staging pass of qeif_moment_kernel: 64-byte runs of pixels per problem when every
// problem brings its own pattern, 128-byte runs when the pattern is shared (more LDS is free).
template <typename T>
constexpr int kMomentChunkOwn = 32 / static_cast<int>(sizeof(T));
template <typename T>
constexpr int kMomentChunkShared = 64 / static_cast<int>(sizeof(T));

// QEIF in moment form over B problems of N points, one lane per problem.  The points pass
// through LDS in chunks of C, transposed so that a lane's reads are conflict-free: element k of
// problem j of the chunk sits at [k * (BLOCK + 1) + j].  The global loads of chunk c + 1 are
// issued into registers before chunk c is consumed, so their latency overlaps the arithmetic.
// A shared pattern (SHARED, xyz_stride == 0) is staged once as float64 [p, packed p p^T] per
// point (9 doubles, read by broadcast), and its own sums S0 = sum p p^T, m0 = sum p are formed
// once per wavefront, the lanes splitting the points, instead of once per problem.
// Dynamic LDS: SHARED ? 2 C (BLOCK + 1) elements, rounded up to 8 bytes, + 9 N doubles
//                     : 5 C (BLOCK + 1) elements.
template <typename T, int C, bool SHARED>
__global__ __launch_bounds__(kMomentBlock<T>, PNP_MOMENT_WAVES) void qeif_moment_kernel(
    const T *__restrict__ uv, const T *__restrict__ xyz, int64_t xyz_stride, KinvRows ki, int64_t B, int N,
    QeifParams p, T *__restrict__ R_out, T *__restrict__ t_out, T *__restrict__ res_out,
    int32_t *__restrict__ iters_out, T *__restrict__ rpy_out)
{
    constexpr int BLOCK = kMomentBlock<T>;
    constexpr int ROW = BLOCK + 1;
    extern __shared__ __align__(16) unsigned char lds_raw[];
    T *s_uv = reinterpret_cast<T *>(lds_raw);
    T *s_xyz = s_uv + 2 * C * ROW;
    double *s_pat = reinterpret_cast<double *>(lds_raw + ((2 * C * ROW * sizeof(T) + 7) & ~static_cast<size_t>(7)));
    const int tid = threadIdx.x;
    const int64_t first = static_cast<int64_t>(blockIdx.x) * BLOCK;
    const int nprob = static_cast<int>(min(static_cast<int64_t>(BLOCK), B - first));
    Moments M;
    moments_clear(M);
    if constexpr (SHARED) {
        for (int i = tid; i < N; i += BLOCK) {
            const double X = static_cast<double>(xyz[3 * i]), Y = static_cast<double>(xyz[3 * i + 1]);
            const double Z = static_cast<double>(xyz[3 * i + 2]);
            double *d = s_pat + 9 * i;
            d[0] = X; d[1] = Y; d[2] = Z;
            d[3] = X * X; d[4] = Y * X; d[5] = Y * Y; d[6] = Z * X; d[7] = Z * Y; d[8] = Z * Z;
        }
        __syncthreads();
        double part[9] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
        for (int i = tid & 63; i < N; i += 64)
#pragma unroll
            for (int k = 0; k < 9; ++k) part[k] += s_pat[9 * i + k];
#pragma unroll
        for (int k = 0; k < 3; ++k) M.m0[k] = wave_uniform(group_sum<64>(part[k]));
#pragma unroll
        for (int k = 0; k < 6; ++k) M.S0[k] = wave_uniform(group_sum<64>(part[3 + k]));
    }

    // A chunk holds, per problem, a run of 2C pixel elements and a run of 3C pattern elements.
    // LU = 2C lanes share one pixel run and LX = 4C lanes one pattern run (3C of them active), so
    // that thread tid keeps the same element index k within the run and steps through problems
    // j0, j0 + JU, ... (JU = BLOCK / LU): offsets advance by a uniform stride per load, the LDS
    // destination by a compile-time constant.  Out-of-range problems and elements are clamped
    // to valid addresses; what they load is never read.
    constexpr int LU = 2 * C, JU = BLOCK / LU, LX = 4 * C, JX = BLOCK / LX;
    static_assert(BLOCK % LU == 0 && BLOCK % LX == 0, "chunk runs must tile the workgroup");
    const T *uv_blk = uv + first * 2 * N;
    const T *xyz_blk = xyz + first * xyz_stride;
    const int ku = tid % LU, ju = tid / LU, kx = tid % LX, jx = tid / LX;
    const unsigned row_u = 2u * static_cast<unsigned>(N), row_x = static_cast<unsigned>(xyz_stride);
    const int jmax = nprob - 1;
    T pre_uv[LU], pre_xyz[SHARED ? 1 : LX];
    auto prefetch = [&](int c0) {
        const int cn = min(C, N - c0);
        const unsigned ou = 2u * c0 + (ku < 2 * cn ? ku : 0);
#pragma unroll
        for (int r = 0; r < LU; ++r) pre_uv[r] = uv_blk[static_cast<unsigned>(min(ju + r * JU, jmax)) * row_u + ou];
        if constexpr (!SHARED) {
            const unsigned ox = 3u * c0 + (kx < 3 * cn ? kx : 0);
#pragma unroll
            for (int r = 0; r < LX; ++r)
                pre_xyz[r] = xyz_blk[static_cast<unsigned>(min(jx + r * JX, jmax)) * row_x + ox];
        }
    };
    auto commit = [&]() {
        T *du = s_uv + ku * ROW + ju;
#pragma unroll
        for (int r = 0; r < LU; ++r) du[r * JU] = pre_uv[r];
        if constexpr (!SHARED) {
            if (kx < 3 * C) {
                T *dx = s_xyz + kx * ROW + jx;
#pragma unroll
                for (int r = 0; r < LX; ++r) dx[r * JX] = pre_xyz[r];
            }
        }
    };

    // One staged point as the accumulation needs it.  The loop below fetches point i + 1 from
    // LDS before it accumulates point i, so that the read latency is covered by arithmetic.
    struct Staged {
        double u, v, pt[3], pp[6];
    };
    auto fetch = [&](int c0, int i, Staged &q) {
        q.u = static_cast<double>(s_uv[(2 * i) * ROW + tid]);
        q.v = static_cast<double>(s_uv[(2 * i + 1) * ROW + tid]);
        if constexpr (SHARED) {
            const double *d = s_pat + 9 * (c0 + i);
#pragma unroll
            for (int k = 0; k < 3; ++k) q.pt[k] = d[k];
#pragma unroll
            for (int k = 0; k < 6; ++k) q.pp[k] = d[3 + k];
        } else {
#pragma unroll
            for (int k = 0; k < 3; ++k) q.pt[k] = static_cast<double>(s_xyz[(3 * i + k) * ROW + tid]);
        }
    };
    auto accumulate = [&](const Staged &q) {
        const double bx = __builtin_fma(ki.a00, q.u, __builtin_fma(ki.a01, q.v, ki.a02));
        const double by = __builtin_fma(ki.a10, q.u, __builtin_fma(ki.a11, q.v, ki.a12));
        if constexpr (SHARED)
            moments_add_axes(M, q.pp, q.pt, bx, by);
        else
            moments_add(M, q.pt[0], q.pt[1], q.pt[2], bx, by);
    };
    prefetch(0);
    for (int c0 = 0; c0 < N; c0 += C) {
        const int cn = min(C, N - c0);
        __syncthreads();  // the previous chunk has been consumed
        commit();
        __syncthreads();
        if (c0 + C < N) prefetch(c0 + C);
        if (tid < nprob) {
            // two points per trip, their registers alternating: the LDS reads of one point are
            // in flight while the other is accumulated (a single rotating pair of variables gets
            // folded by the compiler into read-then-use within one trip).  The trip has no
            // branch: behind a condition the second point's reads are sunk to where they are
            // used and waited for on the spot; an odd last point is taken after the loop.
            if constexpr (SHARED) {
                Staged pa, pb;
                fetch(c0, 0, pa);
#pragma unroll 1
                for (int i = 0; i + 1 < cn; i += 2) {
                    fetch(c0, i + 1, pb);
                    accumulate(pa);
                    fetch(c0, min(i + 2, cn - 1), pa);
                    accumulate(pb);
                }
                if (cn & 1) accumulate(pa);
            } else {
                // with a pattern per problem the landing registers of the pattern leave no room for
                // a second point in flight: one point per trip (two spilled 51 registers)
#pragma unroll 1
                for (int i = 0; i < cn; ++i) {
                    Staged pa;
                    fetch(c0, i, pa);
                    accumulate(pa);
                }
            }
        }
    }
    if (tid >= nprob) return;  // no barrier follows
    moment_iterations(M, uv, xyz, xyz_stride, ki, first + tid, N, p, R_out, t_out, res_out, iters_out, rpy_out);
}

// Points of one 128-byte cache line of pixels.
template <typename T>
constexpr int kMomentChunkLine = 64 / static_cast<int>(sizeof(T));

// qeif_moment_kernel for per-problem patterns with the staging chunks cut where the cache lines
// are, and 16-byte pieces all the way from memory to the lane that consumes them.
// Rows of 2N pixel elements start anywhere in a line, so equal chunks for all problems ask for
// parts of lines, and the rest of each line is fetched again by the next chunk (measured: the
// loads of that shape alone reach 3.8 TB/s, 5.2 TB/s in the shape below).  Here a chunk is C = one
// line of points, and problem j's chunk boundaries are shifted to where its row crosses a line:
// chunk 0 is points [0, p0_j), chunk t points [p0_j + C (t - 1), p0_j + C t), 1 <= p0_j <= C.  The
// pattern runs cover the same points.  Lanes of one wavefront therefore consume chunks of
// different lengths; the sums are formed in the same order as in qeif_moment_kernel and come out
// identical.
// Loads are 16-byte pieces through buffer descriptors over the workgroup's rows: a load past their
// end (rows beyond the batch in the last workgroup, pieces beyond a chunk) returns zeros.  8 lanes
// share a pixel run and 16 lanes (12 active) a pattern run; the rows one thread loads are BLOCK/8
// and BLOCK/16 apart, multiples of C, so they share one shift.  A piece that straddles the end of
// its chunk brings elements of the row's next chunk or of the next row: those are staged and not
// read.
// LDS: 20 pieces per problem (8 of pixels, 12 of pattern), piece q of problem j at
// [q * BLOCK + (j ^ swizzle(q))], 5 C BLOCK elements exactly: with all of the CU's LDS taken by two
// workgroups there is no room for padding, so the staging writes are spread over the banks by the
// exclusive-or instead.  A lane reads its problem's pieces whole: five 16-byte reads bring EP
// points (EP = elements of a piece), a quarter of the LDS instructions of element-wise staging.
// SHARED (one pattern for all problems, xyz_stride == 0): only the pixels are staged this way
// (8 pieces per problem); the pattern sits in LDS behind them as float64 [p, packed p p^T] per
// point as in qeif_moment_kernel<T, C, true>.  Lanes whose rows start at different places in a
// line are at different points of the pattern in the same trip: the table is read at up to C
// addresses per instruction, 72 bytes apart, which fall into different banks.
template <typename T, bool SHARED>
__global__ __launch_bounds__(kMomentBlock<T>, PNP_MOMENT_WAVES) void qeif_moment_lines_kernel(
    const T *__restrict__ uv, const T *__restrict__ xyz, int64_t xyz_stride, KinvRows ki, int64_t B, int N,
    QeifParams p, T *__restrict__ R_out, T *__restrict__ t_out, T *__restrict__ res_out,
    int32_t *__restrict__ iters_out, T *__restrict__ rpy_out)
{
    constexpr int BLOCK = kMomentBlock<T>, C = kMomentChunkLine<T>;
#ifdef PNP_STAMP_WG   // debug builds only: start, end of the accumulation pass and end of every workgroup
    const uint64_t stamp0 = wall_clock64();   // (100 MHz), left where its first problem's rotation goes
#endif
    constexpr int EP = 16 / static_cast<int>(sizeof(T));             // elements of a piece = points of a trip
    constexpr int LU = 2 * C / EP, JU = BLOCK / LU, RU = BLOCK / JU;  // lanes of a pixel run, rows per pass, passes
    constexpr int LXA = 3 * C / EP, LX = 16, JX = BLOCK / LX, RX = BLOCK / JX;
    static_assert(LU == 8 && LXA == 12 && BLOCK % 64 == 0 && JU % C == 0 && JX % C == 0 && C % EP == 0, "staging geometry");
    typedef unsigned piece_t __attribute__((ext_vector_type(4)));
    extern __shared__ __align__(16) unsigned char lds_raw[];
    piece_t *s_uv = reinterpret_cast<piece_t *>(lds_raw);
    piece_t *s_xyz = s_uv + LU * BLOCK;
    const double *s_pat = reinterpret_cast<const double *>(s_xyz);   // SHARED: the table takes the place of the pattern pieces
    // column swizzles: LDS writes meet their banks eight lanes at a time (128 bytes), and eight
    // neighbouring staging lanes hold the eight pixel pieces of one row, or eight of the twelve
    // pattern pieces of one row: the piece index spreads them over the eight 16-byte slots
    // (counted: a quarter of the bank conflicts of a two-rows-by-eight-pieces spread)
    auto swz_u = [](int q) { return q & 15; };
    auto swz_x = [](int q) { return q & 15; };
    const int tid = threadIdx.x;
    const int64_t first = static_cast<int64_t>(blockIdx.x) * BLOCK;
    const int nprob = static_cast<int>(min(static_cast<int64_t>(BLOCK), B - first));
    const T *uv_blk = uv + first * 2 * N;
    const T *xyz_blk = xyz + first * xyz_stride;
    const unsigned row_u = 2u * static_cast<unsigned>(N), row_x = static_cast<unsigned>(xyz_stride);
    const unsigned stride_u = JU * row_u * static_cast<unsigned>(sizeof(T)), stride_x = JX * row_x * static_cast<unsigned>(sizeof(T));
    const __amdgpu_buffer_rsrc_t rsrc_u = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<T *>(uv_blk), 0, static_cast<int>(static_cast<unsigned>(nprob) * row_u * sizeof(T)), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_x = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<T *>(xyz_blk), 0,
        static_cast<int>((static_cast<unsigned>(nprob - 1) * row_x + 3u * static_cast<unsigned>(N)) * sizeof(T)), 0x00020000);
    Moments M;
    moments_clear(M);
    if constexpr (SHARED) {
        double *table = reinterpret_cast<double *>(s_xyz);
        for (int i = tid; i < N; i += BLOCK) {
            const double X = static_cast<double>(xyz[3 * i]), Y = static_cast<double>(xyz[3 * i + 1]);
            const double Z = static_cast<double>(xyz[3 * i + 2]);
            double *d = table + 9 * i;
            d[0] = X; d[1] = Y; d[2] = Z;
            d[3] = X * X; d[4] = Y * X; d[5] = Y * Y; d[6] = Z * X; d[7] = Z * Y; d[8] = Z * Z;
        }
        __syncthreads();
        double part[9] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
        for (int i = tid & 63; i < N; i += 64)
#pragma unroll
            for (int k = 0; k < 9; ++k) part[k] += s_pat[9 * i + k];
#pragma unroll
        for (int k = 0; k < 3; ++k) M.m0[k] = wave_uniform(group_sum<64>(part[k]));
#pragma unroll
        for (int k = 0; k < 6; ++k) M.S0[k] = wave_uniform(group_sum<64>(part[3 + k]));
    }

    // where row j starts within its line, in points: (u0 + N j) mod C
    const int u0 = static_cast<int>((reinterpret_cast<uintptr_t>(uv) / (2 * sizeof(T)) + static_cast<uint64_t>(first) * N) &
                                    (C - 1));
    auto first_len = [&](int j) { return C - ((u0 + N * j) & (C - 1)); };
    int p0_min = C;
    for (int j = 0; j < C; ++j) p0_min = min(p0_min, first_len(j));     // the shifts repeat after C rows at most
    const int chunks = 1 + (max(0, N - p0_min) + C - 1) / C;
    auto chunk_start = [&](int p0, int t) { return t == 0 ? 0 : p0 + C * (t - 1); };

    const int ku = tid % LU, ju = tid / LU, kx = tid % LX, jx = tid / LX;
    const int p0_u = first_len(ju), p0_x = first_len(jx), p0_me = first_len(tid);
    // A pattern run is 192 bytes, a line and a half: where a chunk ends inside a 128-byte line of
    // the pattern array, that line would be asked for twice, by this chunk and by the next, a chunk
    // period apart - and a CU's LDS turns over more bytes per period than its share of L2 holds, so
    // the second request goes to memory again (counted: 1.28 times the input).  Instead the chunk
    // that meets the line first takes all of it: the pieces that belong to the head of the next
    // chunk (up to four, the rest of the line) are loaded by the four lanes of the run that have
    // no piece of their own (kx = 12..15), stay in their landing registers, and are written to LDS
    // with the next chunk, whose first lanes then have nothing to load.  Runs advance by 192 bytes,
    // so a chunk that hands pieces on never receives any: one landing set serves.  Only whole
    // pieces are handed on (rows and chunk starts at multiples of 16 bytes, the common case);
    // otherwise the chunk ends where it did before.
    const unsigned x_align = static_cast<unsigned>(reinterpret_cast<uintptr_t>(xyz_blk)) & 127u;
    auto run_x = [&](int t, int &start, int &len) {
        start = chunk_start(p0_x, t);
        len = max(0, min(N, p0_x + C * t) - start);
    };
    auto handed_on = [&](int t) {   // pieces of chunk t + 1 that chunk t loads, for this thread's rows
        if (SHARED || t < 0 || t + 1 >= chunks || (row_x & 1u)) return 0;   // (rows 16 apart share their place in a line only at even strides)
        int sx, lx, sn, ln;
        run_x(t, sx, lx);
        run_x(t + 1, sn, ln);
        if (lx <= 0 || ln <= 0) return 0;
        const unsigned end = x_align + (static_cast<unsigned>(jx) * row_x + 3u * static_cast<unsigned>(sx + lx)) * static_cast<unsigned>(sizeof(T));
        const unsigned rest = (128u - (end & 127u)) & 127u;   // bytes from the end of the run to the end of its line
        return (rest != 0u && rest <= 64u && (rest & 15u) == 0u && (end & 15u) == 0u) ? static_cast<int>(rest >> 4) : 0;
    };
    piece_t pre_uv[RU], pre_xyz[SHARED ? 1 : RX];
    // The loads of one chunk, as offsets that advance by a fixed stride from one row pass to the next.
    struct Plan {
        unsigned vu, du, vx, dx;
        bool take_x;   // false: a spare lane with nothing to take keeps what it holds
    };
    auto plan = [&](int t) {
        // a piece that lies wholly past its row's chunk is given an offset no descriptor covers:
        // it returns zeros without a memory access
        // (the whole offset goes in the per-lane operand: the range check does not see the scalar one)
        const int su = chunk_start(p0_u, t), lu = min(N, p0_u + C * t) - su;
        int sx, lx;
        run_x(t, sx, lx);
        const int got = handed_on(t - 1), give = handed_on(t);
        constexpr unsigned kNowhere = 0xfffffff0u;
        const bool live_u = EP * ku < 2 * lu;
        Plan pl;
        pl.vu = live_u ? ((static_cast<unsigned>(ju) * row_u + 2u * su) + EP * ku) * static_cast<unsigned>(sizeof(T)) : kNowhere;
        pl.du = live_u ? stride_u : 0u;
        if constexpr (SHARED) {
            pl.vx = pl.dx = 0u;
            pl.take_x = false;
            return pl;
        }
        // own pieces: kx in [got, 12) within the run; pieces handed on: kx - 12 in [0, give), from the
        // head of the next run.  The four spare lanes keep what they hold when they have nothing to take.
        const bool own = kx < LXA;
        const int piece = own ? kx : kx - LXA;
        const bool live_x = own ? (kx >= got && EP * kx < 3 * lx) : (piece < give);
        const unsigned run0 = static_cast<unsigned>(jx) * row_x + 3u * static_cast<unsigned>(own ? sx : sx + lx);
        pl.vx = live_x ? (run0 + EP * piece) * static_cast<unsigned>(sizeof(T)) : kNowhere;
        pl.dx = live_x ? stride_x : 0u;
        pl.take_x = own || give > 0;
        return pl;
    };
    // Row passes [RU part / parts, RU (part + 1) / parts) of the pixel loads and the same share of the pattern loads.
    auto issue = [&](Plan &pl, int part, int parts) {
#pragma unroll
        for (int r = 0; r < RU; ++r)
            if (r * parts / RU == part) {
                pre_uv[r] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_u, pl.vu, 0, 0);
                pl.vu += pl.du;
            }
        if constexpr (!SHARED) {
            if (pl.take_x) {
#pragma unroll
                for (int r = 0; r < RX; ++r)
                    if (r * parts / RX == part) {
                        pre_xyz[r] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_x, pl.vx, 0, 0);
                        pl.vx += pl.dx;
                    }
            }
        }
    };
    auto commit = [&](int t) {
        // The columns are formed at each write (one exclusive-or): as values that stay the same
        // from chunk to chunk they would be kept in registers, and spilled.  The empty statement
        // makes the two swizzles look redefined in every pass.
        const int got = handed_on(t - 1);
        const bool own = kx < LXA;
        const int piece = own ? kx : kx - LXA;          // the spare lanes write what was handed on: pieces 0 .. got - 1
        int su = swz_u(ku), sx = swz_x(piece);
        asm volatile("" : "+v"(su), "+v"(sx));
#pragma unroll
        for (int r = 0; r < RU; ++r) s_uv[ku * BLOCK + ((ju + r * JU) ^ su)] = pre_uv[r];
        if constexpr (!SHARED) {
            if (own ? kx >= got : piece < got) {
#pragma unroll
                for (int r = 0; r < RX; ++r) s_xyz[piece * BLOCK + ((jx + r * JX) ^ sx)] = pre_xyz[r];
            }
        }
    };
    // EP points are two pieces of pixels and three of pattern.
    struct Trip {
        piece_t raw[5];
    };
    auto fetch = [&](int trip, Trip &f) {
        f.raw[0] = s_uv[(2 * trip) * BLOCK + (tid ^ swz_u(2 * trip))];
        f.raw[1] = s_uv[(2 * trip + 1) * BLOCK + (tid ^ swz_u(2 * trip + 1))];
        if constexpr (!SHARED) {
#pragma unroll
            for (int q = 0; q < 3; ++q) f.raw[2 + q] = s_xyz[(3 * trip + q) * BLOCK + (tid ^ swz_x(3 * trip + q))];
        }
    };
    // `point`: index within the row of the trip's first point (SHARED: where to read the table)
    auto accumulate = [&](const Trip &f, int left, int point) {
        T e_uv[2 * EP], e_xyz[3 * EP];
        __builtin_memcpy(e_uv, &f.raw[0], 32);
        if constexpr (!SHARED) __builtin_memcpy(e_xyz, &f.raw[2], 48);
#pragma unroll
        for (int q = 0; q < EP; ++q) {
            if (q < left) {
                const double u = static_cast<double>(e_uv[2 * q]), v = static_cast<double>(e_uv[2 * q + 1]);
                const double bx = __builtin_fma(ki.a00, u, __builtin_fma(ki.a01, v, ki.a02));
                const double by = __builtin_fma(ki.a10, u, __builtin_fma(ki.a11, v, ki.a12));
                if constexpr (SHARED) {
                    const double *d = s_pat + 9 * (point + q);
                    const double pt[3] = {d[0], d[1], d[2]}, pp[6] = {d[3], d[4], d[5], d[6], d[7], d[8]};
                    moments_add_axes(M, pp, pt, bx, by);
                } else {
                    moments_add(M, static_cast<double>(e_xyz[3 * q]), static_cast<double>(e_xyz[3 * q + 1]),
                                static_cast<double>(e_xyz[3 * q + 2]), bx, by);
                }
            }
        }
    };
#pragma unroll
    for (int r = 0; r < (SHARED ? 1 : RX); ++r) pre_xyz[r] = piece_t{0u, 0u, 0u, 0u};   // the spare lanes' set is written by loads only
    {
        Plan pl = plan(0);
        issue(pl, 0, 1);
    }
    // The loads of the next chunk are issued a share at a time between the trips that consume this
    // one: a wavefront that issues 24 loads in a row stands at the memory pipeline for as long as
    // the whole consume step takes (measured: 45 % of the pass), and does its arithmetic only
    // afterwards; in turns, the arithmetic runs while the pipeline takes the loads in.
    constexpr int TRIPS = C / EP;
    static_assert(RU % TRIPS == 0 && RX % TRIPS == 0, "the loads divide among the trips");
    for (int t = 0; t < chunks; ++t) {
        __syncthreads();  // the previous chunk has been consumed
        commit(t);
        __syncthreads();
        const bool more = t + 1 < chunks;
        Plan pl = plan(more ? t + 1 : t);
        const int start = chunk_start(p0_me, t), cn = (tid < nprob) ? max(0, min(N, p0_me + C * t) - start) : 0;
#pragma unroll
        for (int trip = 0; trip < TRIPS; ++trip) {
            if (more) issue(pl, trip, TRIPS);
            __builtin_amdgcn_sched_barrier(0);
            if (EP * trip < cn) {
                Trip f;
                fetch(trip, f);
                accumulate(f, cn - EP * trip, start + EP * trip);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    if (tid >= nprob) return;  // no barrier follows
#ifdef PNP_STAMP_WG
    const uint64_t stamp1 = wall_clock64();
#endif
    moment_iterations(M, uv, xyz, xyz_stride, ki, first + tid, N, p, R_out, t_out, res_out, iters_out, rpy_out);
#ifdef PNP_STAMP_WG
    if (tid == 0) {
        uint64_t *w = reinterpret_cast<uint64_t *>(R_out + first * 9);
        w[0] = stamp0; w[1] = stamp1; w[2] = wall_clock64();
        // where it ran: HW_ID (register 4, 32 bits) and XCC_ID (register 20, low 4 bits)
        w[3] = (static_cast<uint64_t>(__builtin_amdgcn_s_getreg((3 << 11) | 20)) << 32) |
               static_cast<uint32_t>(__builtin_amdgcn_s_getreg((31 << 11) | 4));
    }
#endif
}

template <int LPI, int NCT, typename T>
__global__ __launch_bounds__(kSolveBlock<T>, PNP_DLT_WAVES) void dlt_kernel(
    const T *__restrict__ uv, const T *__restrict__ xyz, int64_t xyz_stride, KinvRows ki, int64_t B, int N,
    int num_it, T *__restrict__ R_out, T *__restrict__ t_out, T *__restrict__ res_out, T *__restrict__ rpy_out)
{
    const Slot<T> s = stage_problems<LPI, kSolveBlock<T>>(uv, xyz, xyz_stride, 1, ki, B, N);
    double R[9], t[3], res;
    bool sound;
    if constexpr (NCT > 0) {
        const RegPoints<NCT> rp(s.pts);
        sound = dlt_solve<1, NCT, false>(rp, N, 0, num_it, res_out != nullptr, R, t, res);
    } else {
        sound = dlt_solve<LPI, 0, false>(s.pts, N, s.lane_in_group, num_it, res_out != nullptr, R, t, res);
    }
    // A problem whose normal equations were rank-deficient in some round (planar pattern, fewer
    // than four points) is solved again through the pseudo-inverse, from the points in LDS; the
    // registers of the first attempt are dead by now, so the rare path costs the common one none.
    if (!sound) dlt_solve<LPI, 0, true>(s.pts, N, s.lane_in_group, num_it, res_out != nullptr, R, t, res);
    if (s.store) {
        store_pose(s.problem, R, t, R_out, t_out);
        if (res_out) res_out[s.problem] = static_cast<T>(res);
        if (rpy_out) store_rpy(s.problem, R, rpy_out);
    }
}

// dlt_kernel<1, 6, T> without the LDS image (see qeif_rows6_kernel).  A problem whose normal
// equations were rank-deficient is solved again through the pseudo-inverse with its points read
// from memory; N (= 6) comes as an argument so that this rare path stays a loop over the points.
template <typename T>
__global__ __launch_bounds__(kRowsBlock, PNP_DLT_WAVES) void dlt_rows6_kernel(
    const T *__restrict__ uv, const T *__restrict__ xyz, int64_t xyz_stride, KinvRows ki, int64_t B, int N,
    int num_it, T *__restrict__ R_out, T *__restrict__ t_out, T *__restrict__ res_out, T *__restrict__ rpy_out)
{
    const int64_t slot = static_cast<int64_t>(blockIdx.x) * kRowsBlock + threadIdx.x;
    const int64_t problem = min(slot, B - 1);
    const T *uv_row = uv + problem * 12, *xyz_row = xyz + problem * xyz_stride;
    const RegPoints<6> rp(OwnRows6<T>(uv_row, xyz_row, ki));
    double R[9], t[3], res;
    const bool sound = dlt_solve<1, 6, false>(rp, 6, 0, num_it, res_out != nullptr, R, t, res);
    if (!sound) {
        const LdsPoints<T> in_mem{uv_row, xyz_row, ki};
        dlt_solve<1, 0, true>(in_mem, N, 0, num_it, res_out != nullptr, R, t, res);
    }
    if (slot < B) {
        store_pose(problem, R, t, R_out, t_out);
        if (res_out) res_out[problem] = static_cast<T>(res);
        if (rpy_out) store_rpy(problem, R, rpy_out);
    }
}

constexpr int kLmBlock = 64;  // one wavefront per workgroup: the 12x12 system needs the whole register file

// 12-state LM over B problems, one lane per problem.  With many points a wavefront's 64 problems
// do not fit the LDS a workgroup may take; the workgroup then stages 64 / LPI problems and LPI
// neighbouring lanes solve the same one (the first of them stores it): fewer problems per
// wavefront, the same arithmetic per problem.
template <typename T, int LPI>
__global__ __launch_bounds__(kLmBlock) void lm12_kernel(
    const T *__restrict__ uv, const T *__restrict__ xyz, int64_t xyz_stride, KinvRows ki, int64_t B, int N,
    double lam, int num_it, T *__restrict__ R_out, T *__restrict__ t_out, T *__restrict__ res_out,
    T *__restrict__ rpy_out)
{
    const Slot<T> s = stage_problems<LPI, kLmBlock>(uv, xyz, xyz_stride, 1, ki, B, N);
    double R[9], t[3], res;
    lm12_solve(s.pts, N, lam, num_it, R, t, res);
    if (s.store) {
        store_pose(s.problem, R, t, R_out, t_out);
        if (res_out) res_out[s.problem] = static_cast<T>(res);
        if (rpy_out) store_rpy(s.problem, R, rpy_out);
    }
}

// R = E_roll E_pitch E_yaw from angles in degrees, with the yaw sign flipped first.
__device__ __forceinline__ void rotation_from_euler_deg(double roll_deg, double yaw_deg, double pitch_deg,
                                                        double (&R)[9])
{
    constexpr double kD2R = 3.14159265358979323846 / 180.0;
    const double roll = roll_deg * kD2R, yaw = -yaw_deg * kD2R, pitch = pitch_deg * kD2R;
    const double c1 = cos(roll), s1 = sin(roll), c2 = cos(yaw), s2 = sin(yaw), c3 = cos(pitch), s3 = sin(pitch);
    // E_pitch E_yaw = [[c2, 0, -s2], [s3 s2, c3, s3 c2], [c3 s2, -s3, c3 c2]], then E_roll on the left
    const double m00 = c2, m01 = 0.0, m02 = -s2;
    const double m10 = s3 * s2, m11 = c3, m12 = s3 * c2;
    const double m20 = c3 * s2, m21 = -s3, m22 = c3 * c2;
    R[0] = c1 * m00 + s1 * m10; R[1] = c1 * m01 + s1 * m11; R[2] = c1 * m02 + s1 * m12;
    R[3] = -s1 * m00 + c1 * m10; R[4] = -s1 * m01 + c1 * m11; R[5] = -s1 * m02 + c1 * m12;
    R[6] = m20; R[7] = m21; R[8] = m22;
}

// Homogeneous pixel of one pattern point: ray = K (R p + t), divided by |ray_z|.
__device__ __forceinline__ void homogeneous_pixel(const Mat3 &Kmat, const double (&R)[9], const double (&t)[3],
                                                  double X, double Y, double Z, double (&px)[3])
{
    const double cx = R[0] * X + R[1] * Y + R[2] * Z + t[0];
    const double cy = R[3] * X + R[4] * Y + R[5] * Z + t[1];
    const double cz = R[6] * X + R[7] * Y + R[8] * Z + t[2];
    const double rx = Kmat.m[0] * cx + Kmat.m[1] * cy + Kmat.m[2] * cz;
    const double ry = Kmat.m[3] * cx + Kmat.m[4] * cy + Kmat.m[5] * cz;
    const double rz = Kmat.m[6] * cx + Kmat.m[7] * cy + Kmat.m[8] * cz;
    const double a = fabs(rz);
    px[0] = rx / a;
    px[1] = ry / a;
    px[2] = rz / a;
}

struct ErrorSet {  // running average / maximum / first strict arg-max of landmark distances
    double total = 0.0, worst = 0.0;
    int key = -1;
    __device__ __forceinline__ void add(const double (&a)[3], const double (&b)[3], int i)
    {
        const double dx = a[0] - b[0], dy = a[1] - b[1], dz = a[2] - b[2];
        const double e = sqrt(dx * dx + dy * dy + dz * dz);
        total += e;
        if (e > worst) {
            worst = e;
            key = i;
        }
    }
};

// Synthetic-input generator: pose (roll, yaw, pitch in degrees; t) -> pixels of a pattern.
// R = E_roll E_pitch E_yaw with the yaw sign flipped first; ray = K (R p + t); the ray is
// divided by |ray_z| and, if q > 0, rounded half-to-even onto a grid of step q.
// One lane per pose; float64 throughout, pixels stored as T.
template <typename T>
__global__ __launch_bounds__(kBlock) void project_kernel(
    const double *__restrict__ rpy_deg, const double *__restrict__ t, const double *__restrict__ xyz,
    int64_t xyz_stride, Mat3 Kmat, int64_t B, int N, double q, T *__restrict__ uv_out)
{
    const int64_t b = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
    if (b >= B) return;
    double R[9];
    rotation_from_euler_deg(rpy_deg[3 * b], rpy_deg[3 * b + 1], rpy_deg[3 * b + 2], R);
    const double tx = t[3 * b], ty = t[3 * b + 1], tz = t[3 * b + 2];
    const double *P = xyz + b * xyz_stride;
    for (int i = 0; i < N; ++i) {
        const double X = P[3 * i], Y = P[3 * i + 1], Z = P[3 * i + 2];
        const double cx = R[0] * X + R[1] * Y + R[2] * Z + tx;
        const double cy = R[3] * X + R[4] * Y + R[5] * Z + ty;
        const double cz = R[6] * X + R[7] * Y + R[8] * Z + tz;
        const double rx = Kmat.m[0] * cx + Kmat.m[1] * cy + Kmat.m[2] * cz;
        const double ry = Kmat.m[3] * cx + Kmat.m[4] * cy + Kmat.m[5] * cz;
        const double rz = Kmat.m[6] * cx + Kmat.m[7] * cy + Kmat.m[8] * cz;
        double u = rx / fabs(rz), v = ry / fabs(rz);
        if (q > 0.0) {
            u = rint(u / q) * q;
            v = rint(v / q) * q;
        }
        uv_out[(b * N + i) * 2] = static_cast<T>(u);
        uv_out[(b * N + i) * 2 + 1] = static_cast<T>(v);
    }
}

// Comparison epilogue of the stress driver, one lane per solved sample.  From the estimate
// (R_est, t_est) and the ground truth (angles in degrees, depth in cm): the Euler angles of the
// estimate, the four errors depth[cm], roll, pitch, yaw and their pass flags |err| < bound (bits
// 0..3 of pass_mask), the ground-truth translation taken along the ESTIMATED direction at the
// true depth, and average / maximum / arg-max landmark distances between the observed pixels,
// the reprojection of the estimate and the projection of the ground truth over all M pattern
// points: lm_err (B,6) = [obs-GT avg, max, reproj-obs avg, max, reproj-GT avg, max], scaled by
// the true depth in metres if normalize != 0; lm_key (B,3) = point index of each maximum (-1 if
// all distances are 0).  uv_obs == nullptr substitutes the ground-truth projections.
template <typename T>
__global__ __launch_bounds__(kBlock) void compare_kernel(
    const T *__restrict__ R_est, const T *__restrict__ t_est, const T *__restrict__ uv_obs,
    const double *__restrict__ rpy_gt_deg, const double *__restrict__ distance_gt_cm,
    const double *__restrict__ xyz, Mat3 Kmat, int64_t B, int M, PassBounds bound, int normalize,
    T *__restrict__ drpy_err, int32_t *__restrict__ pass_mask, T *__restrict__ lm_err,
    int32_t *__restrict__ lm_key)
{
    const int64_t b = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
    if (b >= B) return;
    double Re[9], te[3], Rg[9], tg[3];
#pragma unroll
    for (int k = 0; k < 9; ++k) Re[k] = static_cast<double>(R_est[b * 9 + k]);
#pragma unroll
    for (int k = 0; k < 3; ++k) te[k] = static_cast<double>(t_est[b * 3 + k]);
    const double roll_gt = rpy_gt_deg[3 * b], yaw_gt = rpy_gt_deg[3 * b + 1], pitch_gt = rpy_gt_deg[3 * b + 2];
    const double dist_cm = distance_gt_cm[b];
    const double dist_m = dist_cm * 0.01;
    rotation_from_euler_deg(roll_gt, yaw_gt, pitch_gt, Rg);
#pragma unroll
    for (int k = 0; k < 3; ++k) tg[k] = (te[k] / te[2]) * dist_m;

    double roll, yaw, pitch;
    euler_deg(Re, roll, yaw, pitch);
    const double err[4] = {te[2] * 100.0 - dist_cm, roll - roll_gt, pitch - pitch_gt, yaw - yaw_gt};
    int mask = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (fabs(err[k]) < bound.v[k]) mask |= 1 << k;
        if (drpy_err) drpy_err[b * 4 + k] = static_cast<T>(err[k]);
    }
    if (pass_mask) pass_mask[b] = mask;
    if (!lm_err && !lm_key) return;

    ErrorSet obs_gt, rep_obs, rep_gt;
    for (int i = 0; i < M; ++i) {
        const double X = xyz[3 * i], Y = xyz[3 * i + 1], Z = xyz[3 * i + 2];
        double pe[3], pg[3], po[3];
        homogeneous_pixel(Kmat, Re, te, X, Y, Z, pe);
        homogeneous_pixel(Kmat, Rg, tg, X, Y, Z, pg);
        if (uv_obs) {
            po[0] = static_cast<double>(uv_obs[(b * M + i) * 2]);
            po[1] = static_cast<double>(uv_obs[(b * M + i) * 2 + 1]);
            po[2] = 1.0;
        } else {
            po[0] = pg[0]; po[1] = pg[1]; po[2] = pg[2];
        }
        obs_gt.add(po, pg, i);
        rep_obs.add(pe, po, i);
        rep_gt.add(pe, pg, i);
    }
    const double scale = normalize ? dist_m : 1.0;
    if (lm_err) {
        lm_err[b * 6 + 0] = static_cast<T>(obs_gt.total / M * scale);
        lm_err[b * 6 + 1] = static_cast<T>(obs_gt.worst * scale);
        lm_err[b * 6 + 2] = static_cast<T>(rep_obs.total / M * scale);
        lm_err[b * 6 + 3] = static_cast<T>(rep_obs.worst * scale);
        lm_err[b * 6 + 4] = static_cast<T>(rep_gt.total / M * scale);
        lm_err[b * 6 + 5] = static_cast<T>(rep_gt.worst * scale);
    }
    if (lm_key) {
        lm_key[b * 3 + 0] = obs_gt.key;
        lm_key[b * 3 + 1] = rep_obs.key;
        lm_key[b * 3 + 2] = rep_gt.key;
    }
}

template <typename T>
__global__ __launch_bounds__(kBlock) void euler_kernel(const T *__restrict__ R_in, int64_t B, T *__restrict__ rpy_out)
{
    const int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
    if (i >= B) return;
    double R[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) R[k] = static_cast<double>(R_in[i * 9 + k]);
    store_rpy(i, R, rpy_out);
}


}  // namespace pnp
