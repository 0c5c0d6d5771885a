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
// All arithmetic is float64: inputs are float32 and are widened on load, outputs are
// rounded to float32 once on store.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace pnp {

constexpr int kBlock = 256;

struct KinvRows {  // first two rows of K^-1: nu_x = a00 u + a01 v + a02, nu_y = a10 u + a11 v + a12
    double a00, a01, a02, a10, a11, a12;
};

struct QeifParams {
    double q_inv, r_quat, r_delta, omega0, rel_tol, res_old0;
    int max_it;
};

// ------------------------------------------------------------------------------------------
// scalar helpers
// ------------------------------------------------------------------------------------------

// 1/d to full double precision: hardware seed plus two Newton steps (the seed of
// v_rcp_f64 is not correctly rounded).  d == 0 or non-finite input propagates NaN/inf.
__device__ __forceinline__ double rcp(double d)
{
    double r = __builtin_amdgcn_rcp(d);
    double e = __builtin_fma(-d, r, 1.0);
    r = __builtin_fma(e, r, r);
    e = __builtin_fma(-d, r, 1.0);
    r = __builtin_fma(e, r, r);
    return r;
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

__device__ __forceinline__ void stage_floats(float *dst, const float *src, int count, int tid)
{
    if ((reinterpret_cast<uintptr_t>(src) & 15) == 0) {
        const int n4 = count >> 2;
        const float4 *s4 = reinterpret_cast<const float4 *>(src);
        float4 *d4 = reinterpret_cast<float4 *>(dst);
        for (int i = tid; i < n4; i += kBlock) d4[i] = s4[i];
        for (int i = (n4 << 2) + tid; i < count; i += kBlock) dst[i] = src[i];
    } else {
        for (int i = tid; i < count; i += kBlock) dst[i] = src[i];
    }
}

// ------------------------------------------------------------------------------------------
// QEIF: quaternion iterated information filter (damped Gauss-Newton on [q, delta1, delta2])
// ------------------------------------------------------------------------------------------

// Partial normal-equation sums of one lane.  With H_x = [u, 1, 0], H_y = [v, 0, 1] (u, v the
// 4-vectors of quaternion derivatives) the information matrix A = sum H^T H has the blocks
// qq[10] = sum(u u^T + v v^T), q4[4] = sum u, q5[4] = sum v, A44 = A55 = N, A45 = 0, and the
// gradient is g = sum(H_x^T r_x + H_y^T r_y).
struct QeifSums {
    double qq[10], q4[4], q5[4], g[6], ss;
};

__device__ __forceinline__ void qeif_point(QeifSums &a, const double (&phi)[9], double qa,
                                           double qb, double qc, double qd, double d1, double d2,
                                           double X, double Y, double Z, double bx, double by)
{
    // residual of the measurement model  h = [P phi1 - bx P phi3 + d1 ; P phi2 - by P phi3 + d2]
    const double p1 = __builtin_fma(X, phi[0], __builtin_fma(Y, phi[1], Z * phi[2]));
    const double p2 = __builtin_fma(X, phi[3], __builtin_fma(Y, phi[4], Z * phi[5]));
    const double p3 = __builtin_fma(X, phi[6], __builtin_fma(Y, phi[7], Z * phi[8]));
    const double rx = bx - (__builtin_fma(-bx, p3, p1) + d1);
    const double ry = by - (__builtin_fma(-by, p3, p2) + d2);
    // p^T Q1 = [s0 s1 s2 s3], p^T Q2 = [-s3 -s2 s1 s0], p^T Q3 = [s2 -s3 -s0 s1]
    // with (qa,qb,qc,qd) = 2(qr,qi,qj,qk): the three 3x4 Jacobians share four products.
    const double s0 = __builtin_fma(X, qa, __builtin_fma(-Y, qd, Z * qc));
    const double s1 = __builtin_fma(X, qb, __builtin_fma(Y, qc, Z * qd));
    const double s2 = __builtin_fma(-X, qc, __builtin_fma(Y, qb, Z * qa));
    const double s3 = __builtin_fma(-X, qd, __builtin_fma(-Y, qa, Z * qb));
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
            a.qq[tri(i, j)] = __builtin_fma(u[i], u[j], __builtin_fma(v[i], v[j], a.qq[tri(i, j)]));
        a.q4[i] += u[i];
        a.q5[i] += v[i];
        a.g[i] = __builtin_fma(u[i], rx, __builtin_fma(v[i], ry, a.g[i]));
    }
    a.g[4] += rx;
    a.g[5] += ry;
    a.ss = __builtin_fma(rx, rx, __builtin_fma(ry, ry, a.ss));
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
//   each iteration:  Obar = (Sigma + R_p)^-1
//                    Omega = Obar + q_inv * A(x)           A, g, ss from the points
//                    res   = sqrt(ss)                      (residual at the current x)
//                    Sigma = Omega^-1 ;  x += Sigma * (q_inv * g)
//                    stop if |res - res_old| / res_old < rel_tol, else res_old = res
// The reference forms zeta = Obar x + q_inv H^T (z - h + H x) and x = Sigma zeta; since
// Omega = Obar + q_inv H^T H these are the same vector, x + Sigma q_inv H^T (z - h).
template <int LPI, int NCT, typename PointFetch>
__device__ __forceinline__ void qeif_solve(const PointFetch &pts, int N, int lane_in_group,
                                           const QeifParams &p, double (&R)[9], double (&t)[3],
                                           double &res_out, int &iters_out)
{
    double x[6] = {1.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    double S[21];  // Sigma, packed lower
#pragma unroll
    for (int i = 0; i < 21; ++i) S[i] = 0.0;
    {
        const double s0 = 1.0 / p.omega0;
#pragma unroll
        for (int i = 0; i < 6; ++i) S[tri(i, i)] = s0;
    }
    double res_old = p.res_old0, res = 1.0e5;
    const double nn = static_cast<double>(N);
    int k = 0;
    while (k < p.max_it) {
        ++k;
        // predict
#pragma unroll
        for (int i = 0; i < 4; ++i) S[tri(i, i)] += p.r_quat;
        S[tri(4, 4)] += p.r_delta;
        S[tri(5, 5)] += p.r_delta;
        spd_inverse<6>(S);  // S now holds Obar
        // model sums
        double phi[9], gamma;
        gamma_matrix(x, phi, gamma);
        const double qa = 2.0 * x[0], qb = 2.0 * x[1], qc = 2.0 * x[2], qd = 2.0 * x[3];
        QeifSums a;
#pragma unroll
        for (int i = 0; i < 10; ++i) a.qq[i] = 0.0;
#pragma unroll
        for (int i = 0; i < 4; ++i) a.q4[i] = a.q5[i] = 0.0;
#pragma unroll
        for (int i = 0; i < 6; ++i) a.g[i] = 0.0;
        a.ss = 0.0;
        if constexpr (NCT > 0) {  // LPI == 1: the lane owns all NCT points, held in registers
#pragma unroll
            for (int i = 0; i < NCT; ++i) {
                double X, Y, Z, bx, by;
                pts(i, X, Y, Z, bx, by);
                qeif_point(a, phi, qa, qb, qc, qd, x[4], x[5], X, Y, Z, bx, by);
            }
        } else {
            for (int i = lane_in_group; i < N; i += LPI) {
                double X, Y, Z, bx, by;
                pts(i, X, Y, Z, bx, by);
                qeif_point(a, phi, qa, qb, qc, qd, x[4], x[5], X, Y, Z, bx, by);
            }
        }
        if (LPI > 1) {
#pragma unroll
            for (int i = 0; i < 10; ++i) a.qq[i] = group_sum<LPI>(a.qq[i]);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                a.q4[i] = group_sum<LPI>(a.q4[i]);
                a.q5[i] = group_sum<LPI>(a.q5[i]);
            }
#pragma unroll
            for (int i = 0; i < 6; ++i) a.g[i] = group_sum<LPI>(a.g[i]);
            a.ss = group_sum<LPI>(a.ss);
        }
        // update: Omega = Obar + q_inv A
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
            for (int j = 0; j <= i; ++j) S[tri(i, j)] = __builtin_fma(p.q_inv, a.qq[tri(i, j)], S[tri(i, j)]);
            S[tri(4, i)] = __builtin_fma(p.q_inv, a.q4[i], S[tri(4, i)]);
            S[tri(5, i)] = __builtin_fma(p.q_inv, a.q5[i], S[tri(5, i)]);
        }
        S[tri(4, 4)] = __builtin_fma(p.q_inv, nn, S[tri(4, 4)]);
        S[tri(5, 5)] = __builtin_fma(p.q_inv, nn, S[tri(5, 5)]);
        res = sqrt(a.ss);
        spd_inverse<6>(S);  // S now holds Sigma
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            double dx = 0.0;
#pragma unroll
            for (int j = 0; j < 6; ++j) dx = __builtin_fma(S[i >= j ? tri(i, j) : tri(j, i)], a.g[j], dx);
            x[i] = __builtin_fma(p.q_inv, dx, x[i]);
        }
        const double ratio = (res - res_old) / res_old;
        res_old = res;
        if (fabs(ratio) < p.rel_tol) break;
    }
    // reconstruct: R = Gamma(q)/gamma, t3 = 1/gamma, t = [d1, d2, 1] t3
    double phi[9], gamma;
    gamma_matrix(x, phi, gamma);
#pragma unroll
    for (int i = 0; i < 9; ++i) R[i] = phi[i] / gamma;
    const double t3 = 1.0 / gamma;
    t[0] = x[4] * t3;
    t[1] = x[5] * t3;
    t[2] = t3;
    res_out = res;
    iters_out = k;
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
                                                      double (&t)[3], double &t3)
{
    const double K00 = f[0], K01 = f[1], K10 = f[3], K11 = f[4];
    const double k1 = __builtin_fma(K00, K00, K10 * K10);
    const double k2 = __builtin_fma(K00, K01, K10 * K11);
    const double k3 = __builtin_fma(K01, K01, K11 * K11);
    const double D = __builtin_fma(k1 - k3, k1 - k3, 4.0 * k2 * k2);
    const double g2 = 0.5 * ((k1 + k3) + sqrt(D));
    const double g = sqrt(g2);
    const double e_se = sqrt(g2 - k1);
    const double sg = (k2 > 0.0) ? 1.0 : ((k2 < 0.0) ? -1.0 : 0.0);
    const double d_se = -sg * sqrt(g2 - k3);
    const double det = K00 * K11 - K01 * K10;
    const double dl0 = (K11 * e_se - K10 * d_se) / det;
    const double dl1 = (K00 * d_se - K01 * e_se) / det;
    const double c = det / g;
    const double al0 = -(c * dl0), al1 = -(c * dl1);
    const double sim = __builtin_fma(al0, f[2], al1 * f[5]);
    const double se = (sim < 0.0) ? -1.0 : 1.0;
    const double gi = 1.0 / g;
    R[0] = K00 * gi;
    R[1] = K01 * gi;
    R[2] = se * al0 * gi;
    R[3] = K10 * gi;
    R[4] = K11 * gi;
    R[5] = se * al1 * gi;
    R[6] = se * e_se * gi;
    R[7] = se * d_se * gi;
    R[8] = c * gi;
    t3 = gi;
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
    double t3 = 0.0, res = 30.0;
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
        dlt_block_reconstruct(f, R, t, t3);
        if (it == num_it - 1) {
            // residual of the reconstructed pose against the system just solved
            const double gi = 1.0 / t3;
            const double fx[4] = {R[0] * gi, R[1] * gi, R[2] * gi, t[0] * gi};
            const double fy[4] = {R[3] * gi, R[4] * gi, R[5] * gi, t[1] * gi};
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
            res = sqrt(ss);
        }
        const double gi = 1.0 / t3;
        phi3[0] = R[6] * gi;
        phi3[1] = R[7] * gi;
        phi3[2] = R[8] * gi;
    }
    res_out = res;
}

// ------------------------------------------------------------------------------------------
// kernels
// ------------------------------------------------------------------------------------------

// Point access into the workgroup's LDS image (float32 as loaded; widened on every use).
struct LdsPoints {
    const float *uv;   // this problem's (N,2)
    const float *xyz;  // this problem's (N,3)
    KinvRows ki;
    __device__ __forceinline__ void operator()(int i, double &X, double &Y, double &Z, double &bx,
                                               double &by) const
    {
        const float2 p = *reinterpret_cast<const float2 *>(uv + 2 * i);
        const double u = static_cast<double>(p.x), v = static_cast<double>(p.y);
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
    __device__ __forceinline__ explicit RegPoints(const LdsPoints &src)
    {
#pragma unroll
        for (int i = 0; i < NCT; ++i) src(i, X[i], Y[i], Z[i], bx[i], by[i]);
    }
    __device__ __forceinline__ void set_pattern(const LdsPoints &src)
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
struct Slot {
    int64_t problem;     // global problem index this lane works on
    bool store;          // this lane writes the problem's outputs
    int lane_in_group;   // 0..LPI-1
    LdsPoints pts;       // view of the problem's points (pattern 0)
    int xyz_step;        // floats from one shared pattern to the next in LDS (3N)
};

// Stages the workgroup's problems into dynamic LDS and returns this lane's slot.
// LDS layout: [uv of PB problems: PB*2N floats][xyz: PB*3N floats per-problem, or
// n_patterns*3N floats when the patterns are shared by all problems]
template <int LPI>
__device__ __forceinline__ Slot stage_problems(float *lds, const float *uv, const float *xyz,
                                               int64_t xyz_stride, int n_patterns,
                                               const KinvRows &ki, int64_t B, int N)
{
    constexpr int PB = kBlock / LPI;
    const int tid = threadIdx.x;
    const int64_t first = static_cast<int64_t>(blockIdx.x) * PB;
    const int nprob = static_cast<int>(min(static_cast<int64_t>(PB), B - first));
    float *s_uv = lds;
    float *s_xyz = lds + PB * 2 * N;
    stage_floats(s_uv, uv + first * 2 * N, nprob * 2 * N, tid);
    if (xyz_stride == 0) {
        stage_floats(s_xyz, xyz, n_patterns * 3 * N, tid);
    } else if (xyz_stride == 3 * static_cast<int64_t>(N)) {
        stage_floats(s_xyz, xyz + first * xyz_stride, nprob * 3 * N, tid);
    } else {
        for (int i = tid; i < nprob * 3 * N; i += kBlock) {
            const int g = i / (3 * N);
            s_xyz[i] = xyz[(first + g) * xyz_stride + (i - g * 3 * N)];
        }
    }
    __syncthreads();
    Slot s;
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

__device__ __forceinline__ void store_pose(int64_t problem, const double (&R)[9],
                                           const double (&t)[3], float *R_out, float *t_out)
{
#pragma unroll
    for (int i = 0; i < 9; ++i) R_out[problem * 9 + i] = static_cast<float>(R[i]);
#pragma unroll
    for (int i = 0; i < 3; ++i) t_out[problem * 3 + i] = static_cast<float>(t[i]);
}

__device__ __forceinline__ void store_rpy(int64_t problem, const double (&R)[9], float *rpy_out)
{
    double roll, yaw, pitch;
    euler_deg(R, roll, yaw, pitch);
    rpy_out[problem * 3 + 0] = static_cast<float>(roll);
    rpy_out[problem * 3 + 1] = static_cast<float>(yaw);
    rpy_out[problem * 3 + 2] = static_cast<float>(pitch);
}

// QEIF over B problems.  n_patterns > 1 (shared patterns only) solves every problem against
// each pattern in turn and keeps the first one with the strictly smallest residual.
// NCT > 0 requires LPI == 1 and N == NCT.
template <int LPI, int NCT>
__global__ __launch_bounds__(kBlock) void qeif_kernel(
    const float *__restrict__ uv, const float *__restrict__ xyz, int64_t xyz_stride, int n_patterns,
    KinvRows ki, int64_t B, int N, QeifParams p, float *__restrict__ R_out,
    float *__restrict__ t_out, float *__restrict__ res_out, int32_t *__restrict__ iters_out,
    float *__restrict__ rpy_out, int32_t *__restrict__ pattern_id_out)
{
    extern __shared__ __align__(16) float lds[];
    Slot s = stage_problems<LPI>(lds, uv, xyz, xyz_stride, n_patterns, ki, B, N);
    double R[9], t[3], res;
    int iters, best = 0;
    if constexpr (NCT > 0) {
        RegPoints<NCT> rp(s.pts);
        qeif_solve<1, NCT>(rp, N, 0, p, R, t, res, iters);
        for (int k = 1; k < n_patterns; ++k) {
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
        for (int k = 1; k < n_patterns; ++k) {
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
        if (res_out) res_out[s.problem] = static_cast<float>(res);
        if (iters_out) iters_out[s.problem] = iters;
        if (rpy_out) store_rpy(s.problem, R, rpy_out);
        if (pattern_id_out) pattern_id_out[s.problem] = best;
    }
}

template <int LPI, int NCT>
__global__ __launch_bounds__(kBlock) void dlt_kernel(
    const float *__restrict__ uv, const float *__restrict__ xyz, int64_t xyz_stride, KinvRows ki,
    int64_t B, int N, int num_it, float *__restrict__ R_out, float *__restrict__ t_out,
    float *__restrict__ res_out, float *__restrict__ rpy_out)
{
    extern __shared__ __align__(16) float lds[];
    const Slot s = stage_problems<LPI>(lds, uv, xyz, xyz_stride, 1, ki, B, N);
    double R[9], t[3], res;
    if constexpr (NCT > 0) {
        const RegPoints<NCT> rp(s.pts);
        dlt_solve<1, NCT>(rp, N, 0, num_it, R, t, res);
    } else {
        dlt_solve<LPI, 0>(s.pts, N, s.lane_in_group, num_it, R, t, res);
    }
    if (s.store) {
        store_pose(s.problem, R, t, R_out, t_out);
        if (res_out) res_out[s.problem] = static_cast<float>(res);
        if (rpy_out) store_rpy(s.problem, R, rpy_out);
    }
}

__global__ __launch_bounds__(kBlock) void euler_kernel(const float *__restrict__ R_in, int64_t B,
                                                       float *__restrict__ rpy_out)
{
    const int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
    if (i >= B) return;
    double R[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) R[k] = static_cast<double>(R_in[i * 9 + k]);
    store_rpy(i, R, rpy_out);
}

}  // namespace pnp
