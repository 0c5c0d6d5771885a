// The reference's secondary solver: the literal 12-state Levenberg-Marquardt iteration.
#pragma once

#include "pnp_common.hpp"

namespace pnp {

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
        jacobi_rotate<3, 0, 1>(S, V);
        jacobi_rotate<3, 0, 2>(S, V);
        jacobi_rotate<3, 1, 2>(S, V);
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
            const double n1 = sqrt_pos(d11), n2 = sqrt_pos(d22), n3 = sqrt_pos(d33);
            const double h1 = 0.5 * rcp(n1), h2 = 0.5 * rcp(n2), h3 = 0.5 * rcp(n3);
            const double v6[3] = {u1[0] * h1, u1[1] * h1, u1[2] * h1};
            lm12_row<3>(A, b, ca, v6, 1.0 - n1);
            const double v7[3] = {u2[0] * h2, u2[1] * h2, u2[2] * h2};
            lm12_row<3>(A, b, cb, v7, 1.0 - n2);
            const double v8[3] = {u3[0] * h3, u3[1] * h3, u3[2] * h3};
            lm12_row<3>(A, b, cc, v8, 1.0 - n3);
        }
#pragma unroll
        for (int i = 0; i < 12; ++i) A[tri(i, i)] += lam;
        res = sqrt_pos(ss);
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

}  // namespace pnp
