// Iterated linear DLT with the closed-form block reconstruction of (R, t).
#pragma once

#include "pnp_common.hpp"

namespace pnp {

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

// Delta_i = theta_i . phi3 + 1 is kept away from zero by +-1e-7
// (Dl is never -0: a sum that cancels exactly rounds to +0)
__device__ __forceinline__ double dlt_clamp_depth(double Dl)
{
    return (fabs(Dl) <= 1e-7) ? __builtin_copysign(1e-7, Dl) : Dl;
}

// Relative size below which a pivot of the 4x4 normal equations counts as zero (see spd_solve2),
// and relative size below which one of their eigenvalues does (see sym_pinv_solve2).  Face-like
// patterns have pivot ratios above 1e-6 and eigenvalue ratios above 1e-5 in every round of the
// random-stress distribution; a planar pattern has one pivot and one eigenvalue at round-off
// (about 1e-15 of the largest, or exactly 0).  Between the pivot tolerance and round-off the
// LDL^T solution is as good as any that starts from the normal equations, so the tolerance sits
// just above round-off: a problem sent down the pseudo-inverse path costs its wavefront about
// twenty times a normal solve, and one such straggler lengthens the whole launch.
#ifndef PNP_DLT_PIVOT_TOL
#define PNP_DLT_PIVOT_TOL 1e-12
#endif
constexpr double kDltPivotTol = PNP_DLT_PIVOT_TOL;
constexpr double kDltRankTol = 1e-13;

// One problem's DLT solve.  The reference's (2N x 8) system A phi = B has x rows that touch
// only columns {0,1,2,6} and y rows that touch only {3,4,5,7}, with the same coefficients
// d_i = [theta_i/Delta_i, 1/Delta_i]: it is two (N x 4) least-squares problems sharing one
// matrix, solved here through the 4x4 normal equations M = sum d d^T with two right-hand
// sides.  Equal to pinv(A) B for a full-rank system.
//
// A rank-deficient system (a planar pattern, fewer than four points) has no unique solution; the
// reference's pinv returns the one of minimum norm.  PINV = false solves M by LDL^T and returns
// false if any round met a pivot at round-off, in which case its outputs are not to be used;
// PINV = true solves every round through the pseudo-inverse of M and always returns true.  The
// kernels run the first and repeat a problem with the second only when the first gives up.
// want_res = false (uniform) skips the residual norm of the last round; res_out is then 30.
template <int LPI, int NCT, bool PINV, typename PointFetch>
__device__ __forceinline__ bool dlt_solve(const PointFetch &pts, int N, int lane_in_group,
                                          int num_it, bool want_res, double (&R)[9], double (&t)[3],
                                          double &res_out)
{
    double phi3[3] = {0.0, 0.0, 1.0};
    double gam = 0.0, res = 30.0;  // gam = gamma = 1/t3
    bool sound = true;
#pragma unroll
    for (int i = 0; i < 9; ++i) R[i] = 0.0;
    t[0] = t[1] = t[2] = 0.0;
    for (int it = 0; it < num_it; ++it) {
        double M[10], rx[4], ry[4];
#pragma unroll
        for (int i = 0; i < 10; ++i) M[i] = 0.0;
#pragma unroll
        for (int i = 0; i < 4; ++i) rx[i] = ry[i] = 0.0;
        auto add_row = [&](int i, double w) {
            double X, Y, Z, bx, by;
            pts(i, X, Y, Z, bx, by);
            const double d[4] = {X * w, Y * w, Z * w, w};
#pragma unroll
            for (int a = 0; a < 4; ++a) {
#pragma unroll
                for (int b = 0; b <= a; ++b) M[tri(a, b)] = __builtin_fma(d[a], d[b], M[tri(a, b)]);
                rx[a] = __builtin_fma(d[a], bx, rx[a]);
                ry[a] = __builtin_fma(d[a], by, ry[a]);
            }
        };
        auto raw_depth = [&](int i) {
            double X, Y, Z, bx, by;
            pts(i, X, Y, Z, bx, by);
            return __builtin_fma(X, phi3[0], __builtin_fma(Y, phi3[1], __builtin_fma(Z, phi3[2], 1.0)));
        };
        double wts[NCT > 0 ? NCT : 1];
        if constexpr (NCT > 0) {
            // 1/Delta_i of all NCT points from one reciprocal (of their product) and 3 (NCT - 1)
            // multiplications; the factors are clamped away from zero and of order one.  The clamp
            // is applied only when the smallest |Delta_i| calls for it (one test per round, not
            // one per point).
            double D[NCT], pre[NCT];
            double smallest = 1.0;
#pragma unroll
            for (int i = 0; i < NCT; ++i) {
                D[i] = raw_depth(i);
                smallest = fmin(smallest, fabs(D[i]));
            }
            if (smallest <= 1e-7) {
#pragma unroll
                for (int i = 0; i < NCT; ++i) D[i] = dlt_clamp_depth(D[i]);
            }
#pragma unroll
            for (int i = 0; i < NCT; ++i) pre[i] = (i == 0) ? D[0] : pre[i - 1] * D[i];
            double inv = rcp(pre[NCT - 1]);
#pragma unroll
            for (int i = NCT - 1; i > 0; --i) {
                wts[i] = inv * pre[i - 1];
                inv *= D[i];
            }
            wts[0] = inv;
#pragma unroll
            for (int i = 0; i < NCT; ++i) add_row(i, wts[i]);
        } else {
            for (int i = lane_in_group; i < N; i += LPI) add_row(i, rcp(dlt_clamp_depth(raw_depth(i))));
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
        if constexpr (PINV)
            sym_pinv_solve2<4>(M, rx, ry, kDltRankTol);
        else
            sound = spd_solve2<4>(M, rx, ry, kDltPivotTol) && sound;
        const double f[8] = {rx[0], rx[1], rx[2], ry[0], ry[1], ry[2], rx[3], ry[3]};
        dlt_block_reconstruct(f, R, t, gam);
        if (it == num_it - 1) {
            if (!want_res) break;
            // residual of the reconstructed pose against the system just solved
            const double fx[4] = {R[0] * gam, R[1] * gam, R[2] * gam, t[0] * gam};
            const double fy[4] = {R[3] * gam, R[4] * gam, R[5] * gam, t[1] * gam};
            double ss = 0.0;
            auto residual = [&](int i) {
                double X, Y, Z, bx, by;
                pts(i, X, Y, Z, bx, by);
                const double w = (NCT > 0) ? wts[NCT > 0 ? i : 0] : rcp(dlt_clamp_depth(raw_depth(i)));
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
        } else {
            phi3[0] = R[6] * gam;  // phi3 = R[2,:] / t3, for the next round
            phi3[1] = R[7] * gam;
            phi3[2] = R[8] * gam;
        }
    }
    res_out = res;
    return sound;
}

}  // namespace pnp
