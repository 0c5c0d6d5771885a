// QEIF, the quaternion iterated information filter (the live solver): point form (every
// iteration visits every point) and moment form (sums over the points formed once).
#pragma once

#include "pnp_common.hpp"

namespace pnp {

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
//   ncd1 = -c * delta1, ncd2 = -c * delta2
//   u~ = c (p^T Q1 - bx p^T Q3), v~ = c (p^T Q2 - by p^T Q3): rows H_x = [u, 1, 0], H_y = [v, 0, 1]
//   r~ = c (z - h)
// M receives q_inv H^T H in its (q,q), (d1,q) and (d2,q) entries (packed lower 6x6; the
// constant (d,d) block q_inv*N*I is added by the caller); g receives [c^2 H_q^T r, c r_x, c r_y]
// summed, ss the sum of c^2 r^2.
template <bool FIRST = false>
__device__ __forceinline__ void qeif_point(double (&M)[21], double (&g)[6], double &ss,
                                           const double (&qs)[4], const double (&hq)[4], double c,
                                           double ncd1, double ncd2, double X, double Y, double Z,
                                           double bx, double by)
{
    double u[4], v[4], rx, ry;
    if constexpr (FIRST) {
        // at the initial state q = (1, 0, 0, 0), delta = 0: s = qs0 (X, 0, Z, -Y)
        const double s0 = X * qs[0], s2 = Z * qs[0], s3 = -(Y * qs[0]);
        u[0] = __builtin_fma(-bx, s2, s0);
        u[1] = bx * s3;
        u[2] = __builtin_fma(bx, s0, s2);
        u[3] = s3;
        v[0] = __builtin_fma(-by, s2, -s3);
        v[1] = __builtin_fma(by, s3, -s2);
        v[2] = by * s0;
        v[3] = s0;
        rx = __builtin_fma(-u[0], hq[0], c * bx);
        ry = __builtin_fma(-v[0], hq[0], c * by);
    } else {
        const double s0 = __builtin_fma(X, qs[0], __builtin_fma(-Y, qs[3], Z * qs[2]));
        const double s1 = __builtin_fma(X, qs[1], __builtin_fma(Y, qs[2], Z * qs[3]));
        const double s2 = __builtin_fma(-X, qs[2], __builtin_fma(Y, qs[1], Z * qs[0]));
        const double s3 = __builtin_fma(-X, qs[3], __builtin_fma(-Y, qs[0], Z * qs[1]));
        u[0] = __builtin_fma(-bx, s2, s0);
        u[1] = __builtin_fma(bx, s3, s1);
        u[2] = __builtin_fma(bx, s0, s2);
        u[3] = __builtin_fma(-bx, s1, s3);
        v[0] = __builtin_fma(-by, s2, -s3);
        v[1] = __builtin_fma(by, s3, -s2);
        v[2] = __builtin_fma(by, s0, s1);
        v[3] = __builtin_fma(-by, s1, s0);
        // the model is a quadratic form in q, so its value is half its own Jacobian row times q:
        // c h_x = u~ . q/2 + c d1, and the residual needs no separate evaluation of P phi_j
        rx = __builtin_fma(c, bx, ncd1);
        ry = __builtin_fma(c, by, ncd2);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            rx = __builtin_fma(-u[i], hq[i], rx);
            ry = __builtin_fma(-v[i], hq[i], ry);
        }
    }
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
//   each iteration:  Obar = (Sigma + R_p)^-1               (taken from Omega = Sigma^-1, see QeifNoise)
//                    Omega = Obar + q_inv * H^T H          H, r = z - h at the current x
//                    res   = |r|                           (residual BEFORE the update)
//                    Sigma = Omega^-1 ;  x += Sigma * (q_inv * H^T r)     (as a solve, Sigma not formed)
//                    stop if |res - res_old| < rel_tol * res_old, else res_old = res
//                    (evaluated on the squared residuals, see QeifRule)
// The reference forms zeta = Obar x + q_inv H^T (z - h + H x) and x = Sigma zeta; since
// Omega = Obar + q_inv H^T H these are the same vector, x + Sigma q_inv H^T (z - h).
// Its stopping rule |(res - res_old)/res_old| < rel_tol is the same inequality (res_old > 0).
struct QeifState {
    double x[6];    // [q(4), delta1, delta2]
    double T[21];   // packed lower 6x6; holds the information matrix Omega between iterations
    double ss_old;  // squared residual norm of the previous iteration, in the solver's units
    double ss;      // squared residual norm at the last linearisation point, same units
    int k;          // iterations done
};

// unit2: square of the factor by which the solver scales its residuals (q_inv in the point
// form, which carries c (z - h); 1 in the moment form).
__device__ __forceinline__ void qeif_init(QeifState &st, const QeifParams &p, double unit2)
{
    st.x[0] = 1.0;
#pragma unroll
    for (int i = 1; i < 6; ++i) st.x[i] = 0.0;
#pragma unroll
    for (int i = 0; i < 21; ++i) st.T[i] = 0.0;  // the first predict step sets the diagonal
    st.ss_old = p.res_old0 * p.res_old0 * unit2;
    st.ss = 1.0e10 * unit2;  // reported as a residual of 1e5 when no iteration runs
    st.k = 0;
}

// The stopping rule |res - res_old| < rel_tol res_old on squared residuals (no square root in
// the loop): with res, res_old >= 0 it reads (1 - tol) res_old < res < (1 + tol) res_old, i.e.
// lo2 ss_old < ss < hi2 ss_old, the bounds prepared on the host (QeifParams).
struct QeifRule {
    double lo2, hi2;
    __device__ __forceinline__ explicit QeifRule(const QeifParams &p) : lo2(p.lo2), hi2(p.hi2) {}
};

// Constants of the predict step.  With Omega = Sigma^-1 in hand, (Sigma + R_p)^-1 is taken as
//   Obar = R_p^-1 - R_p^-1 (Omega + R_p^-1)^-1 R_p^-1
// (R_p diagonal and positive), so an iteration inverts one matrix - (Omega + R_p^-1) - and
// solves one system - Omega dx = g - instead of inverting two.
struct QeifNoise {
    double sh[6];             // diagonal of R_p^-1
    double wqq, wqd, wdd;     // products of two of its entries: (q,q), (q,delta), (delta,delta)
    double oq, od;            // diagonal of Obar in the first iteration, 1 / (1/omega0 + r)
    __device__ __forceinline__ explicit QeifNoise(const QeifParams &p)
    {
        const double rq = p.rq_inv, rd = p.rd_inv;
#pragma unroll
        for (int i = 0; i < 4; ++i) sh[i] = rq;
        sh[4] = sh[5] = rd;
        wqq = rq * rq;
        wqd = rq * rd;
        wdd = rd * rd;
        oq = p.oq;
        od = p.od;
    }
};

// Scalars derived from the parameters once per kernel.
struct QeifConsts {
    double c, c_inv, dd;  // sqrt(q_inv), its reciprocal, q_inv * N
    QeifRule rule;
    QeifNoise noise;
    __device__ __forceinline__ QeifConsts(const QeifParams &p, int N)
        : c(p.c), c_inv(p.c_inv), dd(p.q_inv * static_cast<double>(N)), rule(p), noise(p) {}
};

// Predict step: T = Omega of the previous iteration on entry, Obar = (Omega^-1 + R_p)^-1 on exit.
// Sigma_0 is a multiple of the identity, so the first iteration sets two values.
template <bool FIRST>
__device__ __forceinline__ void qeif_predict(QeifState &st, const QeifNoise &nz)
{
    double(&T)[21] = st.T;
    ++st.k;
    if constexpr (FIRST) {
#pragma unroll
        for (int i = 0; i < 4; ++i) T[tri(i, i)] = nz.oq;
        T[tri(4, 4)] = nz.od;
        T[tri(5, 5)] = nz.od;
    } else {
        double Nm[21];  // -(Omega + R_p^-1)^-1
        sweep_neg_inverse_shifted<6>(T, nz.sh, Nm);
#pragma unroll
        for (int i = 0; i < 6; ++i) {
#pragma unroll
            for (int j = 0; j <= i; ++j) {
                const double w = (i < 4) ? nz.wqq : ((j < 4) ? nz.wqd : nz.wdd);
                T[tri(i, j)] = (i == j) ? __builtin_fma(Nm[tri(i, j)], w, nz.sh[i]) : Nm[tri(i, j)] * w;
            }
        }
    }
}

// Correct step: T = Omega on entry and on exit; x += Omega^-1 g; returns true when the stopping
// rule fires on the squared residual ss of the linearisation point just used.
__device__ __forceinline__ bool qeif_correct(QeifState &st, double (&g)[6], double ss, const QeifRule &rule)
{
    spd_solve1<6>(st.T, g);
#pragma unroll
    for (int i = 0; i < 6; ++i) st.x[i] += g[i];
    const bool stop = (ss > rule.lo2 * st.ss_old) && (ss < rule.hi2 * st.ss_old);
    st.ss = ss;
    st.ss_old = ss;
    return stop;
}

// Residual norm of the last linearisation point; unit_inv = 1 / sqrt(unit2) of qeif_init.
__device__ __forceinline__ double qeif_residual(const QeifState &st, double unit_inv)
{
    return sqrt_pos(st.ss) * unit_inv;
}

// One iteration; returns true when the stopping rule fires.
// FIRST: the iteration that starts from the state qeif_init leaves (x = [1,0,0,0,0,0], Sigma a
// multiple of the identity), for which the predict step and the point terms simplify.
template <int LPI, int NCT, bool FIRST, typename PointFetch>
__device__ __forceinline__ bool qeif_iterate(QeifState &st, const PointFetch &pts, int N, int lane_in_group,
                                             const QeifParams &p, const QeifConsts &kc)
{
    double(&x)[6] = st.x;
    double(&T)[21] = st.T;
    const double c = kc.c;
    qeif_predict<FIRST>(st, kc.noise);
    // update: Omega = Obar + q_inv H^T H, accumulated point by point
    const double qs[4] = {2.0 * c * x[0], 2.0 * c * x[1], 2.0 * c * x[2], 2.0 * c * x[3]};
    const double hq[4] = {0.5 * x[0], 0.5 * x[1], 0.5 * x[2], 0.5 * x[3]};
    const double ncd1 = -c * x[4], ncd2 = -c * x[5];
    double g[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    double ss = 0.0;
    if constexpr (LPI == 1) {
        if constexpr (NCT > 0) {  // the lane owns all NCT points, held in registers
#pragma unroll
            for (int i = 0; i < NCT; ++i) {
                double X, Y, Z, bx, by;
                pts(i, X, Y, Z, bx, by);
                qeif_point<FIRST>(T, g, ss, qs, hq, c, ncd1, ncd2, X, Y, Z, bx, by);
            }
        } else {
            for (int i = 0; i < N; ++i) {
                double X, Y, Z, bx, by;
                pts(i, X, Y, Z, bx, by);
                qeif_point<FIRST>(T, g, ss, qs, hq, c, ncd1, ncd2, X, Y, Z, bx, by);
            }
        }
    } else {
        double M[21];  // this lane's share; only the (q,q), (d1,q), (d2,q) entries are used
#pragma unroll
        for (int i = 0; i < 21; ++i) M[i] = 0.0;
        for (int i = lane_in_group; i < N; i += LPI) {
            double X, Y, Z, bx, by;
            pts(i, X, Y, Z, bx, by);
            qeif_point<FIRST>(M, g, ss, qs, hq, c, ncd1, ncd2, X, Y, Z, bx, by);
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
    return qeif_correct(st, g, ss, kc.rule);
}

// reconstruct: R = Gamma(q)/gamma, t3 = 1/gamma, t = [d1, d2, 1] t3
__device__ __forceinline__ void qeif_finish(const QeifState &st, double (&R)[9], double (&t)[3])
{
    double phi[9], gamma;
    gamma_matrix(st.x, phi, gamma);
    const double t3 = rcp(gamma);
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
    qeif_init(st, p, p.q_inv);
    const QeifConsts kc(p, N);
    if (p.max_it > 0 && !qeif_iterate<LPI, NCT, true>(st, pts, N, lane_in_group, p, kc)) {
        while (st.k < p.max_it) {
            if (qeif_iterate<LPI, NCT, false>(st, pts, N, lane_in_group, p, kc)) break;
        }
    }
    qeif_finish(st, R, t);
    res_out = qeif_residual(st, kc.c_inv);
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
// (Qa, phia = Q1, phi1 for the x rows and Q2, phi2 for the y rows).  S2, m2 and b2 meet only Q3
// and phi3, which the axes share, so they are kept summed over the axes: 40 sums in all, formed
// once per problem; an iteration then costs about 650 fused operations whatever N is, against
// 70 N for the point-by-point form.
//
// sum r^2 is a difference of large terms: its rounding noise is about 2e-15 b2.  When
// it comes out below kMomentResidualFloor b2 the residual is recomputed point by point
// from global memory, so the 1 % stopping rule always sees a residual good to 1e-6 or better.
// On pixel-quantised or noisy input that branch is never taken.

constexpr double kMomentResidualFloor = 1e-8;

struct AxisMoments {
    double S1[6], m1[3], b1;  // sum b p p^T (packed lower 3x3), sum b p, sum b
};

// The second-order sums (b^2 p p^T, b^2 p, b^2) enter H^T H, H^T r and |r|^2 only through Q3 and
// phi3, which both axes share, so only their sum over the two axes is kept: S2 = sum (bx^2 +
// by^2) p p^T, and likewise m2 and b2.
struct Moments {
    double S0[6], m0[3];
    AxisMoments ax, ay;
    double S2[6], m2[3], b2;
};

__device__ __forceinline__ void moments_clear(Moments &M)
{
#pragma unroll
    for (int i = 0; i < 6; ++i) M.S0[i] = M.ax.S1[i] = M.ay.S1[i] = M.S2[i] = 0.0;
#pragma unroll
    for (int i = 0; i < 3; ++i) M.m0[i] = M.ax.m1[i] = M.ay.m1[i] = M.m2[i] = 0.0;
    M.ax.b1 = M.ay.b1 = M.b2 = 0.0;
}

__device__ __forceinline__ void axis_moments_add(AxisMoments &A, const double (&pp)[6], const double (&pt)[3],
                                                 double b)
{
#pragma unroll
    for (int i = 0; i < 6; ++i) A.S1[i] = __builtin_fma(b, pp[i], A.S1[i]);
#pragma unroll
    for (int i = 0; i < 3; ++i) A.m1[i] = __builtin_fma(b, pt[i], A.m1[i]);
    A.b1 += b;
}

// The pixel-dependent sums only; pp = packed lower p p^T, pt = p.
__device__ __forceinline__ void moments_add_axes(Moments &M, const double (&pp)[6], const double (&pt)[3],
                                                 double bx, double by)
{
    axis_moments_add(M.ax, pp, pt, bx);
    axis_moments_add(M.ay, pp, pt, by);
    const double bb = __builtin_fma(bx, bx, by * by);
#pragma unroll
    for (int i = 0; i < 6; ++i) M.S2[i] = __builtin_fma(bb, pp[i], M.S2[i]);
#pragma unroll
    for (int i = 0; i < 3; ++i) M.m2[i] = __builtin_fma(bb, pt[i], M.m2[i]);
    M.b2 += bb;
}

__device__ __forceinline__ void moments_add(Moments &M, double X, double Y, double Z, double bx, double by)
{
    const double pt[3] = {X, Y, Z};
    const double pp[6] = {X * X, Y * X, Y * Y, Z * X, Z * Y, Z * Z};
#pragma unroll
    for (int i = 0; i < 6; ++i) M.S0[i] += pp[i];
#pragma unroll
    for (int i = 0; i < 3; ++i) M.m0[i] += pt[i];
    moments_add_axes(M, pp, pt, bx, by);
}

// All sums times w (= q_inv), once after the accumulation: H^T H, H^T r and |r|^2 then come out
// of the moment expressions already weighted, and accumulate straight into Omega.
__device__ __forceinline__ void moments_scale(Moments &M, double w)
{
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        M.S0[i] *= w;
        M.ax.S1[i] *= w;
        M.ay.S1[i] *= w;
        M.S2[i] *= w;
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        M.m0[i] *= w;
        M.ax.m1[i] *= w;
        M.ay.m1[i] *= w;
        M.m2[i] *= w;
    }
    M.ax.b1 *= w;
    M.ay.b1 *= w;
    M.b2 *= w;
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

// One axis' share of H^T H (added to AA, the packed lower 6x6 information matrix: the (q,q)
// block, and sum a into row `row` = 4 or 5), sum a r, sum r, sum r^2, without the terms in S2, m2
// and b2 (added once for both axes by the caller).  The sums, n included, carry the weight q_inv.
// Qa and Q3 are 3x4 (derivatives of phia, phi3 with respect to q).
__device__ __forceinline__ void moment_axis_terms(const double (&S0)[6], const double (&m0)[3], double n,
                                                  const AxisMoments &A, const double (&Qa)[3][4],
                                                  const double (&Q3)[3][4], const double (&phia)[3],
                                                  const double (&phi3)[3], double d, double (&AA)[21],
                                                  int row, double (&ga)[4], double &gd, double &rr)
{
    double s0a[3], s1a[3], s13[3], w1[3], w2[3];
    sym3_mul(S0, phia, s0a);
    sym3_mul(A.S1, phia, s1a);
    sym3_mul(A.S1, phi3, s13);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        w1[k] = __builtin_fma(-d, m0[k], (A.m1[k] - s0a[k]) + s13[k]);
        w2[k] = __builtin_fma(-d, A.m1[k], -s1a[k]);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const double qa[3] = {Qa[0][j], Qa[1][j], Qa[2][j]};
        const double q3[3] = {Q3[0][j], Q3[1][j], Q3[2][j]};
        double u0[3], u1[3], v0[3];
        sym3_mul(S0, qa, u0);
        sym3_mul(A.S1, q3, u1);
        sym3_mul(A.S1, qa, v0);
        const double U[3] = {u0[0] - u1[0], u0[1] - u1[1], u0[2] - u1[2]};
#pragma unroll
        for (int i = j; i < 4; ++i) {
            const double qi[3] = {Qa[0][i], Qa[1][i], Qa[2][i]};
            const double ri[3] = {Q3[0][i], Q3[1][i], Q3[2][i]};
            AA[tri(i, j)] += dot3(qi, U) - dot3(ri, v0);
        }
        AA[tri(row, j)] += dot3(qa, m0) - dot3(q3, A.m1);
        ga[j] = dot3(qa, w1) - dot3(q3, w2);
    }
    const double m1p3 = dot3(A.m1, phi3), m0pa = dot3(m0, phia);
    gd = __builtin_fma(-n, d, (A.b1 - m0pa) + m1p3);
    const double t_bf = __builtin_fma(d, A.b1 + m1p3, dot3(A.m1, phia)) + dot3(phi3, s1a);
    const double t_ff = __builtin_fma(d, __builtin_fma(n, d, 2.0 * m0pa), dot3(phia, s0a));
    rr = __builtin_fma(-2.0, t_bf, t_ff);
}

// One iteration from the moments (M and n weighted by q_inv, see moments_scale); same state and
// recurrence as qeif_iterate, residuals in its units (squared residual times q_inv).  exact_ss()
// returns the unweighted point-by-point sum of squared residuals at the current x.
template <typename ExactResidual>
__device__ __forceinline__ bool qeif_moment_iterate(QeifState &st, const Moments &M, double n,
                                                    const QeifParams &p, const QeifRule &rule,
                                                    const QeifNoise &nz, const ExactResidual &exact_ss)
{
    double(&x)[6] = st.x;
    double(&T)[21] = st.T;
    if (st.k == 0)  // Obar
        qeif_predict<true>(st, nz);
    else
        qeif_predict<false>(st, nz);

    double phi[9], gamma;
    gamma_matrix(x, phi, gamma);
    const double a = 2.0 * x[0], b = 2.0 * x[1], c = 2.0 * x[2], d = 2.0 * x[3];
    const double Q1[3][4] = {{a, b, -c, -d}, {-d, c, b, -a}, {c, d, a, b}};
    const double Q2[3][4] = {{d, c, b, a}, {a, -b, c, -d}, {-b, -a, d, c}};
    const double Q3[3][4] = {{-c, d, -a, b}, {b, a, d, c}, {a, -b, -c, d}};
    const double phi1[3] = {phi[0], phi[1], phi[2]};
    const double phi2[3] = {phi[3], phi[4], phi[5]};
    const double phi3[3] = {phi[6], phi[7], phi[8]};
    double gax[4], gay[4], gdx, gdy, rrx, rry;
    moment_axis_terms(M.S0, M.m0, n, M.ax, Q1, Q3, phi1, phi3, x[4], T, 4, gax, gdx, rrx);
    moment_axis_terms(M.S0, M.m0, n, M.ay, Q2, Q3, phi2, phi3, x[5], T, 5, gay, gdy, rry);
    // the second-order terms, once for both axes: Q3^T S2 Q3 into H^T H, -Q3^T (m2 + S2 phi3) into
    // the gradient, b2 + 2 m2.phi3 + phi3^T S2 phi3 into |r|^2
    double s23[3], g[6];
    sym3_mul(M.S2, phi3, s23);
    const double wc[3] = {M.m2[0] + s23[0], M.m2[1] + s23[1], M.m2[2] + s23[2]};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const double q3[3] = {Q3[0][j], Q3[1][j], Q3[2][j]};
        double y[3];
        sym3_mul(M.S2, q3, y);
#pragma unroll
        for (int i = j; i < 4; ++i) {
            const double ri[3] = {Q3[0][i], Q3[1][i], Q3[2][i]};
            T[tri(i, j)] += dot3(ri, y);
        }
        g[j] = (gax[j] + gay[j]) - dot3(q3, wc);
    }
    g[4] = gdx;
    g[5] = gdy;
    T[tri(4, 4)] += n;
    T[tri(5, 5)] += n;
    double ss = (rrx + rry) + (__builtin_fma(2.0, dot3(M.m2, phi3), M.b2) + dot3(phi3, s23));
    if (!(ss >= kMomentResidualFloor * M.b2)) ss = p.q_inv * exact_ss(x);
    return qeif_correct(st, g, ss, rule);
}

}  // namespace pnp
