// Device code of the batched PnP solvers for CDNA4 (gfx950): wave64, LDS-staged
// correspondences, register-resident 6x6 / 4x4 algebra, no MFMA (the per-problem matrices
// are far too small to form a dense contraction).
//
// Work decomposition.  A workgroup of 256 threads owns PB = 256/LPI consecutive problems,
// LPI ("lanes per problem") being a power of two between 1 and 64.  The workgroup first
// copies its problems' correspondences (one contiguous range of the uv array, and of the
// xyz array when every problem has its own pattern) into LDS with coalesced 16-byte loads.
// The LPI lanes of a problem then split its N points between them, reduce their partial
// normal-equation sums with xor-butterfly wavefront shuffles, and each lane carries the
// small dense algebra redundantly in registers, so no lane ever waits on another.
// LPI = 64 is one wavefront per problem (latency-optimal for few, large problems);
// LPI = 1 is one lane per problem (throughput-optimal for millions of six-point problems).
//
// All arithmetic is float64.  The problem arrays are float32 (widened on load, results rounded
// once on store) or float64; the entry points come in both flavours.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace pnp {

constexpr int kBlock = 256;
// Threads per workgroup of the solver kernels for element type T: 256 for float, 128 for double,
// so that a workgroup's staged correspondences take the same LDS bytes either way.
template <typename T>
constexpr int kSolveBlock = kBlock * static_cast<int>(sizeof(float)) / static_cast<int>(sizeof(T));

#ifndef PNP_QEIF_WAVES
#define PNP_QEIF_WAVES 2  // minimum resident waves per SIMD the QEIF kernels are compiled for
#endif

struct KinvRows {  // first two rows of K^-1: nu_x = a00 u + a01 v + a02, nu_y = a10 u + a11 v + a12
    double a00, a01, a02, a10, a11, a12;
};

struct Mat3 {  // row-major 3x3, passed to kernels by value
    double m[9];
};

struct QeifParams {
    double q_inv, r_quat, r_delta, omega0, rel_tol, res_old0;
    int max_it;
};

// ------------------------------------------------------------------------------------------
// scalar helpers
// ------------------------------------------------------------------------------------------

// 1/d to full double precision.  The hardware seed r0 of v_rcp_f64 carries a relative error
// e = 1 - d r0 of about 5e-8; 1/d = r0 (1 + e + e^2 + ...), and r0 (1 + e + e^2) leaves e^3 ~
// 1e-22, below rounding.  d == 0 or non-finite input propagates NaN/inf.
__device__ __forceinline__ double rcp(double d)
{
    const double r0 = __builtin_amdgcn_rcp(d);
    const double e = __builtin_fma(-d, r0, 1.0);
    return __builtin_fma(__builtin_fma(e, e, e), r0, r0);
}

// sqrt(x) for 0 <= x in the normal range: hardware reciprocal-square-root seed, one coupled
// Goldschmidt step and one residual correction (error of a few 1e-16 relative).  x == 0 gives
// 0, x < 0 or NaN gives NaN; no rescaling of subnormal or huge arguments (not met here).
__device__ __forceinline__ double sqrt_pos(double x)
{
    const double y = __builtin_amdgcn_rsq(x);
    double g = x * y, h = 0.5 * y;
    const double r = __builtin_fma(-h, g, 0.5);
    g = __builtin_fma(g, r, g);
    h = __builtin_fma(h, r, h);
    const double d = __builtin_fma(-g, g, x);
    g = __builtin_fma(d, h, g);
    return (x == 0.0) ? 0.0 : g;
}

__device__ __forceinline__ constexpr int tri(int i, int j) { return i * (i + 1) / 2 + j; }  // i >= j

// In-place inverse of a symmetric positive-definite n x n matrix held as its packed lower
// triangle (a[tri(i,j)], i >= j), by an LDL^T factorisation, inversion of the unit
// triangular factor and the product N^T D^-1 N.  Every loop bound is a compile-time
// constant so that all temporaries live in registers.
template <int n>
__device__ __forceinline__ void spd_inverse(double (&a)[n * (n + 1) / 2])
{
    double L[n][n], W[n][n], dinv[n];
#pragma unroll
    for (int j = 0; j < n; ++j) {
        double s = a[tri(j, j)];
#pragma unroll
        for (int k = 0; k < j; ++k) s = __builtin_fma(-W[j][k], L[j][k], s);
        dinv[j] = rcp(s);
#pragma unroll
        for (int i = j + 1; i < n; ++i) {
            double v = a[tri(i, j)];
#pragma unroll
            for (int k = 0; k < j; ++k) v = __builtin_fma(-W[i][k], L[j][k], v);
            W[i][j] = v;
            L[i][j] = v * dinv[j];
        }
    }
    double Nm[n][n];  // strictly lower part of L^-1
#pragma unroll
    for (int j = 0; j < n; ++j) {
#pragma unroll
        for (int i = j + 1; i < n; ++i) {
            double s = -L[i][j];
#pragma unroll
            for (int k = j + 1; k < i; ++k) s = __builtin_fma(-L[i][k], Nm[k][j], s);
            Nm[i][j] = s;
        }
    }
    double V[n][n];  // D^-1 L^-1, lower part including the diagonal
#pragma unroll
    for (int k = 0; k < n; ++k) {
        V[k][k] = dinv[k];
#pragma unroll
        for (int j = 0; j < k; ++j) V[k][j] = dinv[k] * Nm[k][j];
    }
#pragma unroll
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int j = 0; j <= i; ++j) {
            double s = V[i][j];
#pragma unroll
            for (int k = i + 1; k < n; ++k) s = __builtin_fma(Nm[k][i], V[k][j], s);
            a[tri(i, j)] = s;
        }
    }
}

// In-place symmetric sweep of every pivot: a <- -(a^-1) for a symmetric definite n x n
// matrix held as its packed lower triangle.  Sweeping pivot k maps
//   a_kk -> -1/a_kk,  a_ik -> a_ik/a_kk,  a_ij -> a_ij - a_ik a_jk / a_kk   (i, j != k)
// which keeps the matrix symmetric at every step, so the whole inversion runs in the 21
// registers of the packed triangle plus one column of temporaries.  Works for positive and
// for negative definite input alike (no pivoting is needed in either case).
template <int n>
__device__ __forceinline__ void sweep_neg_inverse(double (&a)[n * (n + 1) / 2])
{
#pragma unroll
    for (int k = 0; k < n; ++k) {
        const double dinv = rcp(a[tri(k, k)]);
        double t[n];
#pragma unroll
        for (int i = 0; i < n; ++i)
            if (i != k) t[i] = a[i > k ? tri(i, k) : tri(k, i)] * dinv;
#pragma unroll
        for (int i = 0; i < n; ++i) {
            if (i == k) continue;
            const double aik = a[i > k ? tri(i, k) : tri(k, i)];
#pragma unroll
            for (int j = 0; j <= i; ++j) {
                if (j == k) continue;
                a[tri(i, j)] = __builtin_fma(-aik, t[j], a[tri(i, j)]);
            }
        }
#pragma unroll
        for (int i = 0; i < n; ++i)
            if (i != k) a[i > k ? tri(i, k) : tri(k, i)] = t[i];
        a[tri(k, k)] = -dinv;
    }
}

// Solves M x = b1 and M x = b2 for a packed SPD n x n matrix (LDL^T, two right-hand sides).
template <int n>
__device__ __forceinline__ void spd_solve2(const double (&m)[n * (n + 1) / 2], double (&b1)[n],
                                           double (&b2)[n])
{
    double L[n][n], W[n][n], dinv[n];
#pragma unroll
    for (int j = 0; j < n; ++j) {
        double s = m[tri(j, j)];
#pragma unroll
        for (int k = 0; k < j; ++k) s = __builtin_fma(-W[j][k], L[j][k], s);
        dinv[j] = rcp(s);
#pragma unroll
        for (int i = j + 1; i < n; ++i) {
            double v = m[tri(i, j)];
#pragma unroll
            for (int k = 0; k < j; ++k) v = __builtin_fma(-W[i][k], L[j][k], v);
            W[i][j] = v;
            L[i][j] = v * dinv[j];
        }
    }
#pragma unroll
    for (int i = 1; i < n; ++i) {  // forward substitution, unit lower
#pragma unroll
        for (int k = 0; k < i; ++k) {
            b1[i] = __builtin_fma(-L[i][k], b1[k], b1[i]);
            b2[i] = __builtin_fma(-L[i][k], b2[k], b2[i]);
        }
    }
#pragma unroll
    for (int i = 0; i < n; ++i) {
        b1[i] *= dinv[i];
        b2[i] *= dinv[i];
    }
#pragma unroll
    for (int i = n - 2; i >= 0; --i) {  // back substitution with L^T
#pragma unroll
        for (int k = i + 1; k < n; ++k) {
            b1[i] = __builtin_fma(-L[k][i], b1[k], b1[i]);
            b2[i] = __builtin_fma(-L[k][i], b2[k], b2[i]);
        }
    }
}

// Sum of v over the LPI lanes of a problem (xor butterfly); every lane gets the total.
template <int LPI>
__device__ __forceinline__ double group_sum(double v)
{
#pragma unroll
    for (int m = LPI / 2; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    return v;
}

// roll, yaw, pitch (degrees) of a rotation matrix, reference conventions: gimbal-lock
// branch when |pi/2 - asin|R21|| <= 1e-7, c2 taken from the better-conditioned of
// R11/cos(th1) and R22/cos(th3), yaw sign flipped.
__device__ __forceinline__ void euler_deg(const double (&R)[9], double &roll, double &yaw,
                                          double &pitch)
{
    constexpr double kPi = 3.14159265358979323846;
    constexpr double kHalfPi = kPi / 2.0;
    constexpr double kR2D = 180.0 / kPi;
    double th1, th2, th3;
    if (fabs(kHalfPi - asin(fabs(R[7]))) <= 1e-7) {
        const double s = (-R[7] > 0.0) ? 1.0 : ((-R[7] < 0.0) ? -1.0 : 0.0);
        th2 = s * kHalfPi;
        th3 = 0.0;
        th1 = atan2(-R[3], R[0]);
    } else {
        th1 = atan2(R[1], R[4]);
        th3 = atan2(R[6], R[8]);
        const double c1 = cos(th1), c3 = cos(th3);
        const double c2 = (fabs(c1) > fabs(c3)) ? R[4] / c1 : R[8] / c3;
        th2 = atan2(-R[7], c2);
    }
    roll = th1 * kR2D;
    yaw = -(th3 * kR2D);
    pitch = th2 * kR2D;
}

// ------------------------------------------------------------------------------------------
// staging: contiguous global range -> LDS, 16 bytes per lane when alignment allows
// ------------------------------------------------------------------------------------------

template <int BLOCK = kBlock, typename T>
__device__ __forceinline__ void stage_elems(T *dst, const T *src, int count, int tid)
{
    constexpr int per16 = 16 / static_cast<int>(sizeof(T));
    if ((reinterpret_cast<uintptr_t>(src) & 15) == 0) {
        const int n16 = count / per16;
        const uint4 *s4 = reinterpret_cast<const uint4 *>(src);
        uint4 *d4 = reinterpret_cast<uint4 *>(dst);
        for (int i = tid; i < n16; i += BLOCK) d4[i] = s4[i];
        for (int i = n16 * per16 + tid; i < count; i += BLOCK) dst[i] = src[i];
    } else {
        for (int i = tid; i < count; i += BLOCK) dst[i] = src[i];
    }
}

// ------------------------------------------------------------------------------------------
// QEIF: quaternion iterated information filter (damped Gauss-Newton on [q, delta1, delta2])
// ------------------------------------------------------------------------------------------

// One point's contribution to the information matrix, the gradient and the squared residual.
//
// Everything is carried in units scaled by c = sqrt(q_inv), so that the products land in the
// information matrix Omega = Obar + q_inv H^T H without a separate scaling pass:
//   qs[4] = c * 2q,  hq[4] = q / 2
//   s~ = c * s with  p^T Q1 = [s0 s1 s2 s3], p^T Q2 = [-s3 -s2 s1 s0], p^T Q3 = [s2 -s3 -s0 s1]
//        (the three 3x4 Jacobians d(phi_j)/dq share these four products of the point with 2q)
//   c * P phi_j = (p^T Q_j) . q / 2           (phi_j is a quadratic form in q)
//   u~ = c (p^T Q1 - bx p^T Q3), v~ = c (p^T Q2 - by p^T Q3): rows H_x = [u, 1, 0], H_y = [v, 0, 1]
//   r~ = c (z - h)
// M receives q_inv H^T H in its (q,q), (d1,q) and (d2,q) entries (packed lower 6x6; the
// constant (d,d) block q_inv*N*I is added by the caller); g receives [c^2 H_q^T r, c r_x, c r_y]
// summed, ss the sum of c^2 r^2.
__device__ __forceinline__ void qeif_point(double (&M)[21], double (&g)[6], double &ss,
                                           const double (&qs)[4], const double (&hq)[4], double c,
                                           double d1, double d2, double X, double Y, double Z,
                                           double bx, double by)
{
    const double s0 = __builtin_fma(X, qs[0], __builtin_fma(-Y, qs[3], Z * qs[2]));
    const double s1 = __builtin_fma(X, qs[1], __builtin_fma(Y, qs[2], Z * qs[3]));
    const double s2 = __builtin_fma(-X, qs[2], __builtin_fma(Y, qs[1], Z * qs[0]));
    const double s3 = __builtin_fma(-X, qs[3], __builtin_fma(-Y, qs[0], Z * qs[1]));
    const double p1 = __builtin_fma(s0, hq[0], __builtin_fma(s1, hq[1], __builtin_fma(s2, hq[2], s3 * hq[3])));
    const double p2 = __builtin_fma(-s3, hq[0], __builtin_fma(-s2, hq[1], __builtin_fma(s1, hq[2], s0 * hq[3])));
    const double p3 = __builtin_fma(s2, hq[0], __builtin_fma(-s3, hq[1], __builtin_fma(-s0, hq[2], s1 * hq[3])));
    const double rx = __builtin_fma(c, bx - d1, -__builtin_fma(-bx, p3, p1));
    const double ry = __builtin_fma(c, by - d2, -__builtin_fma(-by, p3, p2));
    double u[4], v[4];
    u[0] = __builtin_fma(-bx, s2, s0);
    u[1] = __builtin_fma(bx, s3, s1);
    u[2] = __builtin_fma(bx, s0, s2);
    u[3] = __builtin_fma(-bx, s1, s3);
    v[0] = __builtin_fma(-by, s2, -s3);
    v[1] = __builtin_fma(by, s3, -s2);
    v[2] = __builtin_fma(by, s0, s1);
    v[3] = __builtin_fma(-by, s1, s0);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int j = 0; j <= i; ++j)
            M[tri(i, j)] = __builtin_fma(u[i], u[j], __builtin_fma(v[i], v[j], M[tri(i, j)]));
        M[tri(4, i)] = __builtin_fma(c, u[i], M[tri(4, i)]);
        M[tri(5, i)] = __builtin_fma(c, v[i], M[tri(5, i)]);
        g[i] = __builtin_fma(u[i], rx, __builtin_fma(v[i], ry, g[i]));
    }
    g[4] += rx;
    g[5] += ry;
    ss = __builtin_fma(rx, rx, __builtin_fma(ry, ry, ss));
}

// Gamma(q) = |q|^2 R(q), row-major, and gamma = |q|^2
__device__ __forceinline__ void gamma_matrix(const double (&x)[6], double (&phi)[9], double &gamma)
{
    const double qr = x[0], qi = x[1], qj = x[2], qk = x[3];
    gamma = __builtin_fma(qr, qr, __builtin_fma(qi, qi, __builtin_fma(qj, qj, qk * qk)));
    const double qii = qi * qi, qjj = qj * qj, qkk = qk * qk;
    const double qij = qi * qj, qjk = qj * qk, qik = qi * qk;
    const double qri = qr * qi, qrj = qr * qj, qrk = qr * qk;
    phi[0] = gamma - 2.0 * (qjj + qkk);
    phi[1] = 2.0 * (qij - qrk);
    phi[2] = 2.0 * (qik + qrj);
    phi[3] = 2.0 * (qij + qrk);
    phi[4] = gamma - 2.0 * (qii + qkk);
    phi[5] = 2.0 * (qjk - qri);
    phi[6] = 2.0 * (qik - qrj);
    phi[7] = 2.0 * (qjk + qri);
    phi[8] = gamma - 2.0 * (qii + qjj);
}

// One problem's full solve as seen by one of its LPI lanes.  `pts(i, X, Y, Z, bx, by)`
// fetches point i; this lane visits points lane_in_group, lane_in_group + LPI, ...
//
// Recurrence (exact restatement of the information-filter loop, rearranged):
//   Sigma_0 = I/omega0, x_0 = [1,0,0,0,0,0]
//   each iteration:  Obar = (Sigma + R_p)^-1               (both inverses: symmetric sweeps)
//                    Omega = Obar + q_inv * H^T H          H, r = z - h at the current x
//                    res   = |r|                           (residual BEFORE the update)
//                    Sigma = Omega^-1 ;  x += Sigma * (q_inv * H^T r)
//                    stop if |res - res_old| < rel_tol * res_old, else res_old = res
// The reference forms zeta = Obar x + q_inv H^T (z - h + H x) and x = Sigma zeta; since
// Omega = Obar + q_inv H^T H these are the same vector, x + Sigma q_inv H^T (z - h).
// Its stopping rule |(res - res_old)/res_old| < rel_tol is the same inequality (res_old > 0).
struct QeifState {
    double x[6];     // [q(4), delta1, delta2]
    double T[21];    // packed lower 6x6; holds -Sigma between iterations (see qeif_iterate)
    double res_old;  // residual norm of the previous iteration
    double res;      // residual norm at the last linearisation point
    int k;           // iterations done
};

__device__ __forceinline__ void qeif_init(QeifState &st, const QeifParams &p)
{
    st.x[0] = 1.0;
#pragma unroll
    for (int i = 1; i < 6; ++i) st.x[i] = 0.0;
#pragma unroll
    for (int i = 0; i < 21; ++i) st.T[i] = 0.0;
    const double s0 = -1.0 / p.omega0;
#pragma unroll
    for (int i = 0; i < 6; ++i) st.T[tri(i, i)] = s0;
    st.res_old = p.res_old0;
    st.res = 1.0e5;
    st.k = 0;
}

// Scalars derived from the parameters once per kernel.
struct QeifConsts {
    double c, c_inv, dd;  // sqrt(q_inv), its reciprocal, q_inv * N
    __device__ __forceinline__ QeifConsts(const QeifParams &p, int N)
        : c(sqrt(p.q_inv)), c_inv(1.0 / sqrt(p.q_inv)), dd(p.q_inv * static_cast<double>(N)) {}
};

// One iteration; returns true when the stopping rule fires.  T is swept twice and changes
// sign each time: -Sigma on entry, Obar after the first sweep, -Sigma again on exit.
template <int LPI, int NCT, typename PointFetch>
__device__ __forceinline__ bool qeif_iterate(QeifState &st, const PointFetch &pts, int N, int lane_in_group,
                                             const QeifParams &p, const QeifConsts &kc)
{
    double(&x)[6] = st.x;
    double(&T)[21] = st.T;
    const double c = kc.c;
    ++st.k;
    // predict: -(Sigma + R_p), then sweep -> (Sigma + R_p)^-1 = Obar
#pragma unroll
    for (int i = 0; i < 4; ++i) T[tri(i, i)] -= p.r_quat;
    T[tri(4, 4)] -= p.r_delta;
    T[tri(5, 5)] -= p.r_delta;
    sweep_neg_inverse<6>(T);
    // update: Omega = Obar + q_inv H^T H, accumulated point by point
    const double qs[4] = {2.0 * c * x[0], 2.0 * c * x[1], 2.0 * c * x[2], 2.0 * c * x[3]};
    const double hq[4] = {0.5 * x[0], 0.5 * x[1], 0.5 * x[2], 0.5 * x[3]};
    double g[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    double ss = 0.0;
    if constexpr (LPI == 1) {
        if constexpr (NCT > 0) {  // the lane owns all NCT points, held in registers
#pragma unroll
            for (int i = 0; i < NCT; ++i) {
                double X, Y, Z, bx, by;
                pts(i, X, Y, Z, bx, by);
                qeif_point(T, g, ss, qs, hq, c, x[4], x[5], X, Y, Z, bx, by);
            }
        } else {
            for (int i = 0; i < N; ++i) {
                double X, Y, Z, bx, by;
                pts(i, X, Y, Z, bx, by);
                qeif_point(T, g, ss, qs, hq, c, x[4], x[5], X, Y, Z, bx, by);
            }
        }
    } else {
        double M[21];  // this lane's share; only the (q,q), (d1,q), (d2,q) entries are used
#pragma unroll
        for (int i = 0; i < 21; ++i) M[i] = 0.0;
        for (int i = lane_in_group; i < N; i += LPI) {
            double X, Y, Z, bx, by;
            pts(i, X, Y, Z, bx, by);
            qeif_point(M, g, ss, qs, hq, c, x[4], x[5], X, Y, Z, bx, by);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
            for (int j = 0; j <= i; ++j) T[tri(i, j)] += group_sum<LPI>(M[tri(i, j)]);
            T[tri(4, i)] += group_sum<LPI>(M[tri(4, i)]);
            T[tri(5, i)] += group_sum<LPI>(M[tri(5, i)]);
        }
#pragma unroll
        for (int i = 0; i < 6; ++i) g[i] = group_sum<LPI>(g[i]);
        ss = group_sum<LPI>(ss);
    }
    T[tri(4, 4)] += kc.dd;
    T[tri(5, 5)] += kc.dd;
    g[4] *= c;
    g[5] *= c;
    st.res = sqrt_pos(ss) * kc.c_inv;
    sweep_neg_inverse<6>(T);  // T = -Sigma
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        double dx = 0.0;
#pragma unroll
        for (int j = 0; j < 6; ++j) dx = __builtin_fma(T[i >= j ? tri(i, j) : tri(j, i)], g[j], dx);
        x[i] -= dx;
    }
    const bool stop = fabs(st.res - st.res_old) < p.rel_tol * st.res_old;
    st.res_old = st.res;
    return stop;
}

// reconstruct: R = Gamma(q)/gamma, t3 = 1/gamma, t = [d1, d2, 1] t3
__device__ __forceinline__ void qeif_finish(const QeifState &st, double (&R)[9], double (&t)[3])
{
    double phi[9], gamma;
    gamma_matrix(st.x, phi, gamma);
    const double t3 = 1.0 / gamma;
#pragma unroll
    for (int i = 0; i < 9; ++i) R[i] = phi[i] * t3;
    t[0] = st.x[4] * t3;
    t[1] = st.x[5] * t3;
    t[2] = t3;
}

template <int LPI, int NCT, typename PointFetch>
__device__ __forceinline__ void qeif_solve(const PointFetch &pts, int N, int lane_in_group,
                                           const QeifParams &p, double (&R)[9], double (&t)[3],
                                           double &res_out, int &iters_out)
{
    QeifState st;
    qeif_init(st, p);
    const QeifConsts kc(p, N);
    while (st.k < p.max_it) {
        if (qeif_iterate<LPI, NCT>(st, pts, N, lane_in_group, p, kc)) break;
    }
    qeif_finish(st, R, t);
    res_out = st.res;
    iters_out = st.k;
}

// ------------------------------------------------------------------------------------------
// QEIF in moment form, for problems with many points
// ------------------------------------------------------------------------------------------
//
// H^T H, H^T r and |r|^2 of the QEIF model are polynomials in the state whose coefficients are
// sums over the points of products of (p, b p, b^2 p, b, b^2), b = nu_x or nu_y of the point:
//   row of H_x:  a = Q1^T p - bx Q3^T p          residual  r = bx (1 + p.phi3) - p.phi1 - d1
// so that, with S0 = sum p p^T, S1 = sum b p p^T, S2 = sum b^2 p p^T, m0 = sum p, m1 = sum b p,
// m2 = sum b^2 p, b1 = sum b, b2 = sum b^2 (per axis),
//   sum a a^T = Qa^T (S0 Qa - S1 Q3) - Q3^T (S1 Qa - S2 Q3)
//   sum a     = Qa^T m0 - Q3^T m1
//   sum a r   = Qa^T (m1 - S0 phia + S1 phi3 - d m0) - Q3^T (m2 - S1 phia + S2 phi3 - d m1)
//   sum r     = b1 - m0.phia + m1.phi3 - n d
//   sum r^2   = (b2 + 2 m2.phi3 + phi3^T S2 phi3) - 2 (m1.phia + d b1 + phi3^T S1 phia + d m1.phi3)
//               + (phia^T S0 phia + 2 d m0.phia + n d^2)
// (Qa, phia = Q1, phi1 for the x rows and Q2, phi2 for the y rows).  The 50 sums are formed
// once per problem; an iteration then costs about 700 fused operations whatever N is, against
// 78 N for the point-by-point form, which pays from N of about 16 upwards.
//
// sum r^2 is a difference of large terms: its rounding noise is about 2e-15 (b2x + b2y).  When
// it comes out below kMomentResidualFloor (b2x + b2y) the residual is recomputed point by point
// from global memory, so the 1 % stopping rule always sees a residual good to 1e-6 or better.
// On pixel-quantised or noisy input that branch is never taken.

constexpr double kMomentResidualFloor = 1e-8;

struct AxisMoments {
    double S1[6], S2[6], m1[3], m2[3], b1, b2;  // S*: packed lower 3x3
};

struct Moments {
    double S0[6], m0[3];
    AxisMoments ax, ay;
};

__device__ __forceinline__ void moments_clear(Moments &M)
{
#pragma unroll
    for (int i = 0; i < 6; ++i) M.S0[i] = M.ax.S1[i] = M.ax.S2[i] = M.ay.S1[i] = M.ay.S2[i] = 0.0;
#pragma unroll
    for (int i = 0; i < 3; ++i) M.m0[i] = M.ax.m1[i] = M.ax.m2[i] = M.ay.m1[i] = M.ay.m2[i] = 0.0;
    M.ax.b1 = M.ax.b2 = M.ay.b1 = M.ay.b2 = 0.0;
}

__device__ __forceinline__ void axis_moments_add(AxisMoments &A, const double (&pp)[6], const double (&pt)[3],
                                                 double b)
{
    const double bb = b * b;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        A.S1[i] = __builtin_fma(b, pp[i], A.S1[i]);
        A.S2[i] = __builtin_fma(bb, pp[i], A.S2[i]);
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        A.m1[i] = __builtin_fma(b, pt[i], A.m1[i]);
        A.m2[i] = __builtin_fma(bb, pt[i], A.m2[i]);
    }
    A.b1 += b;
    A.b2 += bb;
}

// The pixel-dependent sums only; pp = packed lower p p^T, pt = p.
__device__ __forceinline__ void moments_add_axes(Moments &M, const double (&pp)[6], const double (&pt)[3],
                                                 double bx, double by)
{
    axis_moments_add(M.ax, pp, pt, bx);
    axis_moments_add(M.ay, pp, pt, by);
}

__device__ __forceinline__ void moments_add(Moments &M, double X, double Y, double Z, double bx, double by)
{
    const double pt[3] = {X, Y, Z};
    const double pp[6] = {X * X, Y * X, Y * Y, Z * X, Z * Y, Z * Z};
#pragma unroll
    for (int i = 0; i < 6; ++i) M.S0[i] += pp[i];
#pragma unroll
    for (int i = 0; i < 3; ++i) M.m0[i] += pt[i];
    axis_moments_add(M.ax, pp, pt, bx);
    axis_moments_add(M.ay, pp, pt, by);
}

// y = S v for a packed symmetric 3x3
__device__ __forceinline__ void sym3_mul(const double (&S)[6], const double (&v)[3], double (&y)[3])
{
    y[0] = __builtin_fma(S[0], v[0], __builtin_fma(S[1], v[1], S[3] * v[2]));
    y[1] = __builtin_fma(S[1], v[0], __builtin_fma(S[2], v[1], S[4] * v[2]));
    y[2] = __builtin_fma(S[3], v[0], __builtin_fma(S[4], v[1], S[5] * v[2]));
}

__device__ __forceinline__ double dot3(const double (&a)[3], const double (&b)[3])
{
    return __builtin_fma(a[0], b[0], __builtin_fma(a[1], b[1], a[2] * b[2]));
}

// One axis' share of H^T H (AA: packed lower 4x4, accumulated), sum a, sum a r, sum r, sum r^2.
// Qa and Q3 are 3x4 (derivatives of phia, phi3 with respect to q).
__device__ __forceinline__ void moment_axis_terms(const double (&S0)[6], const double (&m0)[3], double n,
                                                  const AxisMoments &A, const double (&Qa)[3][4],
                                                  const double (&Q3)[3][4], const double (&phia)[3],
                                                  const double (&phi3)[3], double d, double (&AA)[10],
                                                  double (&sa)[4], double (&ga)[4], double &gd, double &rr)
{
    double s0a[3], s1a[3], s13[3], s23[3], w1[3], w2[3];
    sym3_mul(S0, phia, s0a);
    sym3_mul(A.S1, phia, s1a);
    sym3_mul(A.S1, phi3, s13);
    sym3_mul(A.S2, phi3, s23);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        w1[k] = __builtin_fma(-d, m0[k], (A.m1[k] - s0a[k]) + s13[k]);
        w2[k] = __builtin_fma(-d, A.m1[k], (A.m2[k] - s1a[k]) + s23[k]);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const double qa[3] = {Qa[0][j], Qa[1][j], Qa[2][j]};
        const double q3[3] = {Q3[0][j], Q3[1][j], Q3[2][j]};
        double u0[3], u1[3], v0[3], v1[3];
        sym3_mul(S0, qa, u0);
        sym3_mul(A.S1, q3, u1);
        sym3_mul(A.S1, qa, v0);
        sym3_mul(A.S2, q3, v1);
        const double U[3] = {u0[0] - u1[0], u0[1] - u1[1], u0[2] - u1[2]};
        const double V[3] = {v0[0] - v1[0], v0[1] - v1[1], v0[2] - v1[2]};
#pragma unroll
        for (int i = j; i < 4; ++i) {
            const double qi[3] = {Qa[0][i], Qa[1][i], Qa[2][i]};
            const double ri[3] = {Q3[0][i], Q3[1][i], Q3[2][i]};
            AA[tri(i, j)] += dot3(qi, U) - dot3(ri, V);
        }
        sa[j] = dot3(qa, m0) - dot3(q3, A.m1);
        ga[j] = dot3(qa, w1) - dot3(q3, w2);
    }
    const double m1p3 = dot3(A.m1, phi3), m0pa = dot3(m0, phia);
    gd = __builtin_fma(-n, d, (A.b1 - m0pa) + m1p3);
    const double t_bb = __builtin_fma(2.0, dot3(A.m2, phi3), A.b2) + dot3(phi3, s23);
    const double t_bf = __builtin_fma(d, A.b1 + m1p3, dot3(A.m1, phia)) + dot3(phi3, s1a);
    const double t_ff = __builtin_fma(d, __builtin_fma(n, d, 2.0 * m0pa), dot3(phia, s0a));
    rr = __builtin_fma(-2.0, t_bf, t_bb) + t_ff;
}

// One iteration from the moments; same state and recurrence as qeif_iterate.  exact_ss()
// returns the point-by-point sum of squared residuals at the current x.
template <typename ExactResidual>
__device__ __forceinline__ bool qeif_moment_iterate(QeifState &st, const Moments &M, double n,
                                                    const QeifParams &p, const ExactResidual &exact_ss)
{
    double(&x)[6] = st.x;
    double(&T)[21] = st.T;
    ++st.k;
#pragma unroll
    for (int i = 0; i < 4; ++i) T[tri(i, i)] -= p.r_quat;
    T[tri(4, 4)] -= p.r_delta;
    T[tri(5, 5)] -= p.r_delta;
    sweep_neg_inverse<6>(T);  // Obar

    double phi[9], gamma;
    gamma_matrix(x, phi, gamma);
    const double a = 2.0 * x[0], b = 2.0 * x[1], c = 2.0 * x[2], d = 2.0 * x[3];
    const double Q1[3][4] = {{a, b, -c, -d}, {-d, c, b, -a}, {c, d, a, b}};
    const double Q2[3][4] = {{d, c, b, a}, {a, -b, c, -d}, {-b, -a, d, c}};
    const double Q3[3][4] = {{-c, d, -a, b}, {b, a, d, c}, {a, -b, -c, d}};
    const double phi1[3] = {phi[0], phi[1], phi[2]};
    const double phi2[3] = {phi[3], phi[4], phi[5]};
    const double phi3[3] = {phi[6], phi[7], phi[8]};
    double AA[10], sax[4], say[4], gax[4], gay[4], gdx, gdy, rrx, rry;
#pragma unroll
    for (int i = 0; i < 10; ++i) AA[i] = 0.0;
    moment_axis_terms(M.S0, M.m0, n, M.ax, Q1, Q3, phi1, phi3, x[4], AA, sax, gax, gdx, rrx);
    moment_axis_terms(M.S0, M.m0, n, M.ay, Q2, Q3, phi2, phi3, x[5], AA, say, gay, gdy, rry);
    double ss = rrx + rry;
    if (!(ss >= kMomentResidualFloor * (M.ax.b2 + M.ay.b2))) ss = exact_ss(x);

    // Omega = Obar + q_inv H^T H
    const double w = p.q_inv;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int j = 0; j <= i; ++j) T[tri(i, j)] = __builtin_fma(w, AA[tri(i, j)], T[tri(i, j)]);
        T[tri(4, i)] = __builtin_fma(w, sax[i], T[tri(4, i)]);
        T[tri(5, i)] = __builtin_fma(w, say[i], T[tri(5, i)]);
    }
    T[tri(4, 4)] = __builtin_fma(w, n, T[tri(4, 4)]);
    T[tri(5, 5)] = __builtin_fma(w, n, T[tri(5, 5)]);
    const double g[6] = {w * (gax[0] + gay[0]), w * (gax[1] + gay[1]), w * (gax[2] + gay[2]),
                         w * (gax[3] + gay[3]), w * gdx, w * gdy};
    st.res = sqrt_pos(ss);
    sweep_neg_inverse<6>(T);  // -Sigma
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        double dx = 0.0;
#pragma unroll
        for (int j = 0; j < 6; ++j) dx = __builtin_fma(T[i >= j ? tri(i, j) : tri(j, i)], g[j], dx);
        x[i] -= dx;
    }
    const bool stop = fabs(st.res - st.res_old) < p.rel_tol * st.res_old;
    st.res_old = st.res;
    return stop;
}

// ------------------------------------------------------------------------------------------
// iterated linear DLT with closed-form block reconstruction
// ------------------------------------------------------------------------------------------

// Closed-form (R, t, t3) from phi = [phi1(3), phi2(3), delta1, delta2] using only the
// top-left 2x2 block K of Gamma = R/t3:  gamma^2 = largest root of the characteristic
// equation of K^T K, beta = [sqrt(g2-k1), -sign(k2) sqrt(g2-k3)], delta = K^-T beta,
// c = det K / gamma, alpha = -c delta, overall sign from the dot product of alpha with the
// least-squares third column.
__device__ __forceinline__ void dlt_block_reconstruct(const double (&f)[8], double (&R)[9],
                                                      double (&t)[3], double &gamma)
{
    const double K00 = f[0], K01 = f[1], K10 = f[3], K11 = f[4];
    const double k1 = __builtin_fma(K00, K00, K10 * K10);
    const double k2 = __builtin_fma(K00, K01, K10 * K11);
    const double k3 = __builtin_fma(K01, K01, K11 * K11);
    const double D = __builtin_fma(k1 - k3, k1 - k3, 4.0 * k2 * k2);
    const double g2 = 0.5 * ((k1 + k3) + sqrt_pos(D));
    const double g = sqrt_pos(g2);
    const double gi = rcp(g);
    const double e_se = sqrt_pos(g2 - k1);
    const double sg = (k2 > 0.0) ? 1.0 : ((k2 < 0.0) ? -1.0 : 0.0);
    const double d_se = -sg * sqrt_pos(g2 - k3);
    const double det = __builtin_fma(K00, K11, -(K01 * K10));
    const double rdet = rcp(det);
    const double dl0 = __builtin_fma(K11, e_se, -(K10 * d_se)) * rdet;
    const double dl1 = __builtin_fma(K00, d_se, -(K01 * e_se)) * rdet;
    const double c = det * gi;
    const double al0 = -(c * dl0), al1 = -(c * dl1);
    const double sim = __builtin_fma(al0, f[2], al1 * f[5]);
    const double se = (sim < 0.0) ? -gi : gi;  // sign and 1/gamma in one factor
    R[0] = K00 * gi;
    R[1] = K01 * gi;
    R[2] = se * al0;
    R[3] = K10 * gi;
    R[4] = K11 * gi;
    R[5] = se * al1;
    R[6] = se * e_se;
    R[7] = se * d_se;
    R[8] = c * gi;
    gamma = g;
    t[0] = f[6] * gi;
    t[1] = f[7] * gi;
    t[2] = gi;
}

// One problem's DLT solve.  The reference's (2N x 8) system A phi = B has x rows that touch
// only columns {0,1,2,6} and y rows that touch only {3,4,5,7}, with the same coefficients
// d_i = [theta_i/Delta_i, 1/Delta_i]: it is two (N x 4) least-squares problems sharing one
// matrix, solved here through the 4x4 normal equations M = sum d d^T with two right-hand
// sides.  Equal to pinv(A) B for a full-rank system.
template <int LPI, int NCT, typename PointFetch>
__device__ __forceinline__ void dlt_solve(const PointFetch &pts, int N, int lane_in_group,
                                          int num_it, double (&R)[9], double (&t)[3],
                                          double &res_out)
{
    double phi3[3] = {0.0, 0.0, 1.0};
    double gam = 0.0, res = 30.0;  // gam = gamma = 1/t3
#pragma unroll
    for (int i = 0; i < 9; ++i) R[i] = 0.0;
    t[0] = t[1] = t[2] = 0.0;
    for (int it = 0; it < num_it; ++it) {
        double M[10], rx[4], ry[4];
#pragma unroll
        for (int i = 0; i < 10; ++i) M[i] = 0.0;
#pragma unroll
        for (int i = 0; i < 4; ++i) rx[i] = ry[i] = 0.0;
        auto accumulate = [&](int i) {
            double X, Y, Z, bx, by;
            pts(i, X, Y, Z, bx, by);
            double Dl = __builtin_fma(X, phi3[0], __builtin_fma(Y, phi3[1], Z * phi3[2])) + 1.0;
            if (fabs(Dl) <= 1e-7) Dl = (Dl < 0.0) ? -1e-7 : 1e-7;
            const double w = rcp(Dl);
            const double d[4] = {X * w, Y * w, Z * w, w};
#pragma unroll
            for (int a = 0; a < 4; ++a) {
#pragma unroll
                for (int b = 0; b <= a; ++b) M[tri(a, b)] = __builtin_fma(d[a], d[b], M[tri(a, b)]);
                rx[a] = __builtin_fma(d[a], bx, rx[a]);
                ry[a] = __builtin_fma(d[a], by, ry[a]);
            }
        };
        if constexpr (NCT > 0) {
#pragma unroll
            for (int i = 0; i < NCT; ++i) accumulate(i);
        } else {
            for (int i = lane_in_group; i < N; i += LPI) accumulate(i);
        }
        if (LPI > 1) {
#pragma unroll
            for (int i = 0; i < 10; ++i) M[i] = group_sum<LPI>(M[i]);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                rx[i] = group_sum<LPI>(rx[i]);
                ry[i] = group_sum<LPI>(ry[i]);
            }
        }
        spd_solve2<4>(M, rx, ry);
        const double f[8] = {rx[0], rx[1], rx[2], ry[0], ry[1], ry[2], rx[3], ry[3]};
        dlt_block_reconstruct(f, R, t, gam);
        if (it == num_it - 1) {
            // residual of the reconstructed pose against the system just solved
            const double fx[4] = {R[0] * gam, R[1] * gam, R[2] * gam, t[0] * gam};
            const double fy[4] = {R[3] * gam, R[4] * gam, R[5] * gam, t[1] * gam};
            double ss = 0.0;
            auto residual = [&](int i) {
                double X, Y, Z, bx, by;
                pts(i, X, Y, Z, bx, by);
                double Dl = __builtin_fma(X, phi3[0], __builtin_fma(Y, phi3[1], Z * phi3[2])) + 1.0;
                if (fabs(Dl) <= 1e-7) Dl = (Dl < 0.0) ? -1e-7 : 1e-7;
                const double w = rcp(Dl);
                const double ex = bx - w * __builtin_fma(X, fx[0], __builtin_fma(Y, fx[1], __builtin_fma(Z, fx[2], fx[3])));
                const double ey = by - w * __builtin_fma(X, fy[0], __builtin_fma(Y, fy[1], __builtin_fma(Z, fy[2], fy[3])));
                ss = __builtin_fma(ex, ex, __builtin_fma(ey, ey, ss));
            };
            if constexpr (NCT > 0) {
#pragma unroll
                for (int i = 0; i < NCT; ++i) residual(i);
            } else {
                for (int i = lane_in_group; i < N; i += LPI) residual(i);
            }
            if (LPI > 1) ss = group_sum<LPI>(ss);
            res = sqrt_pos(ss);
        }
        phi3[0] = R[6] * gam;  // phi3 = R[2,:] / t3
        phi3[1] = R[7] * gam;
        phi3[2] = R[8] * gam;
    }
    res_out = res;
}

// ------------------------------------------------------------------------------------------
// 12-state Levenberg-Marquardt (the reference's secondary solver)
// ------------------------------------------------------------------------------------------

// Adds one sparse Jacobian row to the normal equations: A += v v^T on the listed columns
// (packed lower 12x12; the column list is ascending and a compile-time constant after
// inlining), b += v * dz.
template <int K>
__device__ __forceinline__ void lm12_row(double (&A)[78], double (&b)[12], const int (&c)[K],
                                         const double (&v)[K], double dz)
{
#pragma unroll
    for (int i = 0; i < K; ++i) {
#pragma unroll
        for (int j = 0; j <= i; ++j) A[tri(c[i], c[j])] = __builtin_fma(v[i], v[j], A[tri(c[i], c[j])]);
        b[c[i]] = __builtin_fma(v[i], dz, b[c[i]]);
    }
}

// Solves A x = b in place for a packed SPD matrix by a right-looking LDL^T; A is destroyed,
// b returns x.
template <int n>
__device__ __forceinline__ void ldlt_solve_inplace(double (&A)[n * (n + 1) / 2], double (&b)[n])
{
#pragma unroll
    for (int j = 0; j < n; ++j) {
        const double dinv = rcp(A[tri(j, j)]);
        double l[n];
#pragma unroll
        for (int i = j + 1; i < n; ++i) l[i] = A[tri(i, j)] * dinv;
#pragma unroll
        for (int i = j + 1; i < n; ++i) {
#pragma unroll
            for (int k = j + 1; k <= i; ++k) A[tri(i, k)] = __builtin_fma(-l[i], A[tri(k, j)], A[tri(i, k)]);
        }
#pragma unroll
        for (int i = j + 1; i < n; ++i) A[tri(i, j)] = l[i];
        A[tri(j, j)] = dinv;
    }
#pragma unroll
    for (int i = 1; i < n; ++i) {
#pragma unroll
        for (int k = 0; k < i; ++k) b[i] = __builtin_fma(-A[tri(i, k)], b[k], b[i]);
    }
#pragma unroll
    for (int i = 0; i < n; ++i) b[i] *= A[tri(i, i)];
#pragma unroll
    for (int i = n - 2; i >= 0; --i) {
#pragma unroll
        for (int k = i + 1; k < n; ++k) b[i] = __builtin_fma(-A[tri(k, i)], b[k], b[i]);
    }
}

// One Jacobi rotation of the symmetric 3x3 matrix S (full storage) in the (p,q) plane,
// accumulated into V (columns are the eigenvectors).
template <int p, int q>
__device__ __forceinline__ void jacobi_rotate(double (&S)[3][3], double (&V)[3][3])
{
    const double apq = S[p][q];
    if (apq == 0.0) return;
    const double theta = (S[q][q] - S[p][p]) / (2.0 * apq);
    const double tt = ((theta >= 0.0) ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
    const double c = 1.0 / sqrt(tt * tt + 1.0), s = tt * c;
    constexpr int r = 3 - p - q;
    const double app = S[p][p], aqq = S[q][q], arp = S[r][p], arq = S[r][q];
    S[p][p] = app - tt * apq;
    S[q][q] = aqq + tt * apq;
    S[p][q] = S[q][p] = 0.0;
    S[r][p] = S[p][r] = c * arp - s * arq;
    S[r][q] = S[q][r] = s * arp + c * arq;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const double vkp = V[k][p], vkq = V[k][q];
        V[k][p] = c * vkp - s * vkq;
        V[k][q] = s * vkp + c * vkq;
    }
}

// Nearest rotation to G (row-major 3x3) in the sense of the reference's
// U diag(1, 1, det(U V^T)) V^T, and the largest singular value of G.  G^T G is diagonalised
// by cyclic Jacobi sweeps; with singular pairs (sigma_k, v_k) sorted descending,
// u_k = G v_k / sigma_k for k = 1, 2, u_3 = u_1 x u_2 and R = u1 v1^T + u2 v2^T + det(V) u3 v3^T,
// which equals the reference's product for either sign convention of its SVD.
__device__ __forceinline__ void nearest_rotation(const double (&G)[9], double (&R)[9], double &sigma_max)
{
    double S[3][3], V[3][3] = {{1.0, 0.0, 0.0}, {0.0, 1.0, 0.0}, {0.0, 0.0, 1.0}};
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) S[i][j] = G[i] * G[j] + G[3 + i] * G[3 + j] + G[6 + i] * G[6 + j];
    for (int sweep = 0; sweep < 8; ++sweep) {
        jacobi_rotate<0, 1>(S, V);
        jacobi_rotate<0, 2>(S, V);
        jacobi_rotate<1, 2>(S, V);
    }
    double lam[3] = {S[0][0], S[1][1], S[2][2]};
    auto swap_cols = [&](int a, int b) {
        const double tl = lam[a];
        lam[a] = lam[b];
        lam[b] = tl;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const double tv = V[k][a];
            V[k][a] = V[k][b];
            V[k][b] = tv;
        }
    };
    if (lam[0] < lam[1]) swap_cols(0, 1);
    if (lam[0] < lam[2]) swap_cols(0, 2);
    if (lam[1] < lam[2]) swap_cols(1, 2);
    sigma_max = sqrt(lam[0]);
    double u[2][3];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const double inv = 1.0 / sqrt(lam[k]);
#pragma unroll
        for (int i = 0; i < 3; ++i)
            u[k][i] = (G[3 * i] * V[0][k] + G[3 * i + 1] * V[1][k] + G[3 * i + 2] * V[2][k]) * inv;
    }
    const double u3[3] = {u[0][1] * u[1][2] - u[0][2] * u[1][1], u[0][2] * u[1][0] - u[0][0] * u[1][2],
                          u[0][0] * u[1][1] - u[0][1] * u[1][0]};
    const double detV = V[0][0] * (V[1][1] * V[2][2] - V[1][2] * V[2][1]) -
                        V[0][1] * (V[1][0] * V[2][2] - V[1][2] * V[2][0]) +
                        V[0][2] * (V[1][0] * V[2][1] - V[1][1] * V[2][0]);
    const double sd = (detV < 0.0) ? -1.0 : 1.0;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) R[3 * i + j] = u[0][i] * V[j][0] + u[1][i] * V[j][1] + sd * u3[i] * V[j][2];
}

// One problem, one lane.  State x = [u1(3), u2(3), u3(3), delta1, delta2, gamma]; residual =
// 2N reprojection rows h = gamma (P u_j - b P u3) + delta_j plus nine soft constraint rows
// (three orthogonality, three equal-norm, three unit-norm with targets 0,0,0,0,0,0,1,1,1);
// num_it iterations of dx = (J^T J + lam I)^-1 J^T (z - h).  The constraint-row Jacobians
// are the ones the reference writes down (u, not 2u, for the squared-norm differences and
// u/(2|u|) for the norms), since they define its iteration.
template <typename PointFetch>
__device__ __forceinline__ void lm12_solve(const PointFetch &pts, int N, double lam, int num_it,
                                           double (&R)[9], double (&t)[3], double &res_out)
{
    double x[12] = {1.0, 0.0, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0, 1.0, 0.0, 0.0, 1.0};
    double res = 1.0e5;
    for (int it = 0; it < num_it; ++it) {
        double A[78], b[12], ss = 0.0;
#pragma unroll
        for (int i = 0; i < 78; ++i) A[i] = 0.0;
#pragma unroll
        for (int i = 0; i < 12; ++i) b[i] = 0.0;
        const double g = x[11];
        for (int i = 0; i < N; ++i) {
            double X, Y, Z, bx, by;
            pts(i, X, Y, Z, bx, by);
            const double P1 = X * x[0] + Y * x[1] + Z * x[2];
            const double P2 = X * x[3] + Y * x[4] + Z * x[5];
            const double P3 = X * x[6] + Y * x[7] + Z * x[8];
            const double hu1 = __builtin_fma(-bx, P3, P1), hu2 = __builtin_fma(-by, P3, P2);
            const double dzx = bx - __builtin_fma(g, hu1, x[9]), dzy = by - __builtin_fma(g, hu2, x[10]);
            ss = __builtin_fma(dzx, dzx, __builtin_fma(dzy, dzy, ss));
            const double gX = g * X, gY = g * Y, gZ = g * Z;
            constexpr int cx[8] = {0, 1, 2, 6, 7, 8, 9, 11};
            const double vx[8] = {gX, gY, gZ, -bx * gX, -bx * gY, -bx * gZ, 1.0, hu1};
            lm12_row<8>(A, b, cx, vx, dzx);
            constexpr int cy[8] = {3, 4, 5, 6, 7, 8, 10, 11};
            const double vy[8] = {gX, gY, gZ, -by * gX, -by * gY, -by * gZ, 1.0, hu2};
            lm12_row<8>(A, b, cy, vy, dzy);
        }
        {
            const double *u1 = x, *u2 = x + 3, *u3 = x + 6;
            const double d11 = u1[0] * u1[0] + u1[1] * u1[1] + u1[2] * u1[2];
            const double d22 = u2[0] * u2[0] + u2[1] * u2[1] + u2[2] * u2[2];
            const double d33 = u3[0] * u3[0] + u3[1] * u3[1] + u3[2] * u3[2];
            const double d13 = u1[0] * u3[0] + u1[1] * u3[1] + u1[2] * u3[2];
            const double d23 = u2[0] * u3[0] + u2[1] * u3[1] + u2[2] * u3[2];
            const double d12 = u1[0] * u2[0] + u1[1] * u2[1] + u1[2] * u2[2];
            constexpr int c13[6] = {0, 1, 2, 6, 7, 8}, c23[6] = {3, 4, 5, 6, 7, 8}, c12[6] = {0, 1, 2, 3, 4, 5};
            const double v0[6] = {u3[0], u3[1], u3[2], u1[0], u1[1], u1[2]};
            lm12_row<6>(A, b, c13, v0, -d13);
            const double v1[6] = {u3[0], u3[1], u3[2], u2[0], u2[1], u2[2]};
            lm12_row<6>(A, b, c23, v1, -d23);
            const double v2[6] = {u2[0], u2[1], u2[2], u1[0], u1[1], u1[2]};
            lm12_row<6>(A, b, c12, v2, -d12);
            const double v3[6] = {u1[0], u1[1], u1[2], -u3[0], -u3[1], -u3[2]};
            lm12_row<6>(A, b, c13, v3, -(d11 - d33));
            const double v4[6] = {u2[0], u2[1], u2[2], -u3[0], -u3[1], -u3[2]};
            lm12_row<6>(A, b, c23, v4, -(d22 - d33));
            const double v5[6] = {u1[0], u1[1], u1[2], -u2[0], -u2[1], -u2[2]};
            lm12_row<6>(A, b, c12, v5, -(d11 - d22));
            constexpr int ca[3] = {0, 1, 2}, cb[3] = {3, 4, 5}, cc[3] = {6, 7, 8};
            const double n1 = sqrt(d11), n2 = sqrt(d22), n3 = sqrt(d33);
            const double h1 = 0.5 / n1, h2 = 0.5 / n2, h3 = 0.5 / n3;
            const double v6[3] = {u1[0] * h1, u1[1] * h1, u1[2] * h1};
            lm12_row<3>(A, b, ca, v6, 1.0 - n1);
            const double v7[3] = {u2[0] * h2, u2[1] * h2, u2[2] * h2};
            lm12_row<3>(A, b, cb, v7, 1.0 - n2);
            const double v8[3] = {u3[0] * h3, u3[1] * h3, u3[2] * h3};
            lm12_row<3>(A, b, cc, v8, 1.0 - n3);
        }
#pragma unroll
        for (int i = 0; i < 12; ++i) A[tri(i, i)] += lam;
        res = sqrt(ss);
        ldlt_solve_inplace<12>(A, b);
#pragma unroll
        for (int i = 0; i < 12; ++i) x[i] += b[i];
    }
    const double G[9] = {x[0], x[1], x[2], x[3], x[4], x[5], x[6], x[7], x[8]};
    double smax;
    nearest_rotation(G, R, smax);
    const double t3 = 1.0 / (smax * x[11]);
    t[0] = x[9] * t3;
    t[1] = x[10] * t3;
    t[2] = t3;
    res_out = res;
}

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
__global__ __launch_bounds__(kSolveBlock<T>, PNP_QEIF_WAVES) void qeif_moment_kernel(
    const T *__restrict__ uv, const T *__restrict__ xyz, int64_t xyz_stride, KinvRows ki, int64_t B, int N,
    QeifParams p, T *__restrict__ R_out, T *__restrict__ t_out, T *__restrict__ res_out,
    int32_t *__restrict__ iters_out, T *__restrict__ rpy_out)
{
    constexpr int BLOCK = kSolveBlock<T>;
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
        for (int k = 0; k < 3; ++k) M.m0[k] = group_sum<64>(part[k]);
#pragma unroll
        for (int k = 0; k < 6; ++k) M.S0[k] = group_sum<64>(part[3 + k]);
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
            Staged cur, nxt;
            fetch(c0, 0, nxt);
#pragma unroll 1
            for (int i = 0; i < cn; ++i) {
                cur = nxt;
                fetch(c0, min(i + 1, cn - 1), nxt);
                accumulate(cur);
            }
        }
    }
    if (tid >= nprob) return;  // no barrier follows

    const int64_t problem = first + tid;
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
    qeif_init(st, p);
    const double n = static_cast<double>(N);
    while (st.k < p.max_it) {
        if (qeif_moment_iterate(st, M, n, p, exact_ss)) break;
    }
    double R[9], t[3];
    qeif_finish(st, R, t);
    store_pose(problem, R, t, R_out, t_out);
    if (res_out) res_out[problem] = static_cast<T>(st.res);
    if (iters_out) iters_out[problem] = st.k;
    if (rpy_out) store_rpy(problem, R, rpy_out);
}

template <int LPI, int NCT, typename T>
__global__ __launch_bounds__(kSolveBlock<T>) void dlt_kernel(
    const T *__restrict__ uv, const T *__restrict__ xyz, int64_t xyz_stride, KinvRows ki, int64_t B, int N,
    int num_it, T *__restrict__ R_out, T *__restrict__ t_out, T *__restrict__ res_out, T *__restrict__ rpy_out)
{
    const Slot<T> s = stage_problems<LPI, kSolveBlock<T>>(uv, xyz, xyz_stride, 1, ki, B, N);
    double R[9], t[3], res;
    if constexpr (NCT > 0) {
        const RegPoints<NCT> rp(s.pts);
        dlt_solve<1, NCT>(rp, N, 0, num_it, R, t, res);
    } else {
        dlt_solve<LPI, 0>(s.pts, N, s.lane_in_group, num_it, R, t, res);
    }
    if (s.store) {
        store_pose(s.problem, R, t, R_out, t_out);
        if (res_out) res_out[s.problem] = static_cast<T>(res);
        if (rpy_out) store_rpy(s.problem, R, rpy_out);
    }
}

constexpr int kLmBlock = 64;  // one wavefront per workgroup: the 12x12 system needs the whole register file

// 12-state LM over B problems, one lane per problem.
template <typename T>
__global__ __launch_bounds__(kLmBlock) void lm12_kernel(
    const T *__restrict__ uv, const T *__restrict__ xyz, int64_t xyz_stride, KinvRows ki, int64_t B, int N,
    double lam, int num_it, T *__restrict__ R_out, T *__restrict__ t_out, T *__restrict__ res_out,
    T *__restrict__ rpy_out)
{
    const Slot<T> s = stage_problems<1, kLmBlock>(uv, xyz, xyz_stride, 1, ki, B, N);
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
    const double *__restrict__ xyz, Mat3 Kmat, int64_t B, int M, Mat3 bound, int normalize,
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
        if (fabs(err[k]) < bound.m[k]) mask |= 1 << k;
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
