// Shared device definitions of the batched PnP solvers for CDNA4 (gfx950): launch constants,
// parameter structs, exact reciprocal / square root, small symmetric linear algebra held in
// registers, the wavefront group sum, Euler angles, and the global -> LDS copy.  All float64.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace pnp {

constexpr int kBlock = 256;  // threads per workgroup of the element-wise kernels
// Threads per workgroup of the point-form solver kernels for element type T: 128 for float, 64
// for double, so that a workgroup's staged correspondences take the same LDS bytes either way.
// Two wavefronts per workgroup: a workgroup's slots free up when its slower wavefront ends, so
// small workgroups keep the SIMDs fuller through the spread of iteration counts than the 256
// threads the moment kernel wants for its staging geometry.
#ifndef PNP_SOLVE_BLOCK_BYTES
#define PNP_SOLVE_BLOCK_BYTES 512
#endif
template <typename T>
constexpr int kSolveBlock = PNP_SOLVE_BLOCK_BYTES / static_cast<int>(sizeof(T));
// Threads per workgroup of the moment-form kernel: 256 for float, 128 for double.
template <typename T>
constexpr int kMomentBlock = 1024 / static_cast<int>(sizeof(T));

#ifndef PNP_QEIF_WAVES
#define PNP_QEIF_WAVES 2  // minimum resident waves per SIMD the QEIF kernels are compiled for
#endif
#ifndef PNP_MOMENT_WAVES
#define PNP_MOMENT_WAVES 2  // the same for the moment-form QEIF kernel
#endif
#ifndef PNP_DLT_WAVES
#define PNP_DLT_WAVES 3   // the same for the DLT kernels
#endif

struct KinvRows {  // first two rows of K^-1: nu_x = a00 u + a01 v + a02, nu_y = a10 u + a11 v + a12
    double a00, a01, a02, a10, a11, a12;
};

struct Mat3 {  // row-major 3x3, passed to kernels by value
    double m[9];
};

struct PassBounds {  // pass criteria of the comparison epilogue: depth [cm], roll, pitch, yaw [deg]
    double v[4];
};

struct QeifParams {
    double q_inv, r_quat, r_delta, omega0, rel_tol, res_old0;
    int max_it;
    // derived on the host once per call (divisions and square roots the kernels would otherwise
    // repeat in every lane): sqrt(q_inv) and its reciprocal; 1/r_quat, 1/r_delta; the diagonal of
    // the first predicted information matrix, 1/(1/omega0 + r); the stopping-rule bounds
    double c, c_inv, rq_inv, rd_inv, oq, od, lo2, hi2;
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

// Symmetric sweep: sweeping every pivot of a symmetric definite n x n matrix held as its packed
// lower triangle (a[tri(i,j)], i >= j) leaves -(a^-1).  Sweeping pivot k maps
//   a_kk -> -1/a_kk,  a_ik -> a_ik/a_kk,  a_ij -> a_ij - a_ik a_jk / a_kk   (i, j != k)
// which keeps the matrix symmetric at every step, so the whole inversion runs in the registers
// of the packed triangle plus one column of temporaries.  Works for positive and for negative
// definite input alike (no pivoting is needed in either case).
//
// sweep_pivots<n, K0> sweeps pivots K0 .. n-1 in place.
template <int n, int K0>
__device__ __forceinline__ void sweep_pivots(double (&a)[n * (n + 1) / 2])
{
#pragma unroll
    for (int k = K0; k < n; ++k) {
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

// o <- -((a + diag(sh))^-1), a left untouched: the first pivot is swept out of place (reading a,
// writing o), the others in place on o, so that keeping a costs no register copies.
template <int n>
__device__ __forceinline__ void sweep_neg_inverse_shifted(const double (&a)[n * (n + 1) / 2],
                                                          const double (&sh)[n], double (&o)[n * (n + 1) / 2])
{
    const double dinv = rcp(a[0] + sh[0]);
    double t[n];
#pragma unroll
    for (int i = 1; i < n; ++i) t[i] = a[tri(i, 0)] * dinv;
#pragma unroll
    for (int i = 1; i < n; ++i) {
#pragma unroll
        for (int j = 1; j <= i; ++j)
            o[tri(i, j)] = __builtin_fma(-a[tri(i, 0)], t[j], (i == j) ? a[tri(i, j)] + sh[i] : a[tri(i, j)]);
    }
#pragma unroll
    for (int i = 1; i < n; ++i) o[tri(i, 0)] = t[i];
    o[0] = -dinv;
    sweep_pivots<n, 1>(o);
}

// Solves M x = b in place of b for a packed SPD n x n matrix (LDL^T), M left untouched.
template <int n>
__device__ __forceinline__ void spd_solve1(const double (&m)[n * (n + 1) / 2], double (&b)[n])
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
        for (int k = 0; k < i; ++k) b[i] = __builtin_fma(-L[i][k], b[k], b[i]);
    }
#pragma unroll
    for (int i = 0; i < n; ++i) b[i] *= dinv[i];
#pragma unroll
    for (int i = n - 2; i >= 0; --i) {  // back substitution with L^T
#pragma unroll
        for (int k = i + 1; k < n; ++k) b[i] = __builtin_fma(-L[k][i], b[k], b[i]);
    }
}

// Solves M x = b1 and M x = b2 for a packed SPD n x n matrix (LDL^T, two right-hand sides).
// Returns whether every pivot was sound: larger than pivot_tol times its own diagonal entry (the
// ratio is the share of that column not explained by the columns before it, 1 for independent
// columns).  A pivot that is not (zero, or round-off of either sign) means the matrix is
// rank-deficient to working precision and the solution just computed is not to be used.  NaN
// pivots count as sound: NaN input is not rank deficiency, and its NaN solution stands.
template <int n>
__device__ __forceinline__ bool spd_solve2(const double (&m)[n * (n + 1) / 2], double (&b1)[n],
                                           double (&b2)[n], double pivot_tol)
{
    double L[n][n], W[n][n], dinv[n];
    bool sound = true;
#pragma unroll
    for (int j = 0; j < n; ++j) {
        double s = m[tri(j, j)];
#pragma unroll
        for (int k = 0; k < j; ++k) s = __builtin_fma(-W[j][k], L[j][k], s);
        sound = sound && !(s <= pivot_tol * m[tri(j, j)]);  // a NaN pivot is not a rank verdict
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
    return sound;
}

// One Jacobi rotation of the symmetric n x n matrix S (full storage) in the (p,q) plane,
// accumulated into V (whose columns become the eigenvectors).  An off-diagonal entry below
// 2^-60 of the geometric mean of its two diagonal entries is set to zero instead (the rotation
// would change nothing at working precision); that also bounds theta, so that the refined
// hardware reciprocal and square root stay within their range.  Returns whether it rotated.
template <int n, int p, int q>
__device__ __forceinline__ bool jacobi_rotate(double (&S)[n][n], double (&V)[n][n])
{
    const double apq = S[p][q];
    if (apq == 0.0) return false;
    if (apq * apq <= 0x1p-120 * fabs(S[p][p] * S[q][q])) {
        S[p][q] = S[q][p] = 0.0;
        return false;
    }
    const double theta = 0.5 * (S[q][q] - S[p][p]) * rcp(apq);
    const double tt = __builtin_copysign(rcp(fabs(theta) + sqrt_pos(__builtin_fma(theta, theta, 1.0))), theta);
    const double c = rcp(sqrt_pos(__builtin_fma(tt, tt, 1.0))), s = tt * c;
    S[p][p] -= tt * apq;
    S[q][q] += tt * apq;
    S[p][q] = S[q][p] = 0.0;
#pragma unroll
    for (int r = 0; r < n; ++r) {
        if (r == p || r == q) continue;
        const double arp = S[r][p], arq = S[r][q];
        S[r][p] = S[p][r] = c * arp - s * arq;
        S[r][q] = S[q][r] = s * arp + c * arq;
    }
#pragma unroll
    for (int k = 0; k < n; ++k) {
        const double vkp = V[k][p], vkq = V[k][q];
        V[k][p] = c * vkp - s * vkq;
        V[k][q] = s * vkp + c * vkq;
    }
    return true;
}

template <int n, int p, int q>
__device__ __forceinline__ bool jacobi_sweep_from(double (&S)[n][n], double (&V)[n][n])
{
    bool rotated = jacobi_rotate<n, p, q>(S, V);
    if constexpr (q + 1 < n)
        rotated = jacobi_sweep_from<n, p, q + 1>(S, V) || rotated;
    else if constexpr (p + 2 < n)
        rotated = jacobi_sweep_from<n, p + 1, p + 2>(S, V) || rotated;
    return rotated;
}

// Diagonalises S = V diag(lambda) V^T by cyclic sweeps, at most max_sweeps of them and until a
// sweep finds nothing left to rotate (S ends up diagonal; V must come in as the identity).
// Quadratically convergent: five or six sweeps reach round-off for n <= 4.
template <int n>
__device__ __forceinline__ void jacobi_eigen(double (&S)[n][n], double (&V)[n][n], int max_sweeps)
{
#pragma unroll 1
    for (int sw = 0; sw < max_sweeps; ++sw)
        if (!jacobi_sweep_from<n, 0, 1>(S, V)) break;
}

// x = M^+ b for two right-hand sides, M a packed symmetric positive semi-definite n x n matrix
// and M^+ its pseudo-inverse: eigenvalues not above rank_tol times the largest are taken as
// zero and their directions left out, which yields the minimum-norm least-squares solution, as
// numpy.linalg.pinv does for the rectangular system M was formed from.  (Forming M squares the
// singular values and adds round-off of about 1e-16 of the largest eigenvalue, so a singular
// value below sqrt(rank_tol) of the largest counts as zero here.)  NaN in M reaches x.
template <int n>
__device__ __forceinline__ void sym_pinv_solve2(const double (&m)[n * (n + 1) / 2], double (&b1)[n],
                                                double (&b2)[n], double rank_tol)
{
    double S[n][n], V[n][n];
#pragma unroll
    for (int i = 0; i < n; ++i)
#pragma unroll
        for (int j = 0; j < n; ++j) {
            S[i][j] = m[i >= j ? tri(i, j) : tri(j, i)];
            V[i][j] = (i == j) ? 1.0 : 0.0;
        }
    jacobi_eigen<n>(S, V, 12);
    double lam_max = S[0][0];
#pragma unroll
    for (int k = 1; k < n; ++k) lam_max = (S[k][k] > lam_max) ? S[k][k] : lam_max;
    const double cut = rank_tol * lam_max;
    double x1[n], x2[n];
#pragma unroll
    for (int i = 0; i < n; ++i) x1[i] = x2[i] = 0.0;
#pragma unroll
    for (int k = 0; k < n; ++k) {
        if (S[k][k] <= cut) continue;  // a NaN eigenvalue is kept, and spreads
        double c1 = 0.0, c2 = 0.0;
#pragma unroll
        for (int i = 0; i < n; ++i) {
            c1 = __builtin_fma(V[i][k], b1[i], c1);
            c2 = __builtin_fma(V[i][k], b2[i], c2);
        }
        const double inv = rcp(S[k][k]);
        c1 *= inv;
        c2 *= inv;
#pragma unroll
        for (int i = 0; i < n; ++i) {
            x1[i] = __builtin_fma(V[i][k], c1, x1[i]);
            x2[i] = __builtin_fma(V[i][k], c2, x2[i]);
        }
    }
#pragma unroll
    for (int i = 0; i < n; ++i) {
        b1[i] = x1[i];
        b2[i] = x2[i];
    }
}

// Marks a value that is equal in every lane of the wavefront as such, so that it can live in
// scalar registers.
__device__ __forceinline__ double wave_uniform(double v)
{
    const int lo = __builtin_amdgcn_readfirstlane(__double2loint(v));
    const int hi = __builtin_amdgcn_readfirstlane(__double2hiint(v));
    return __hiloint2double(hi, lo);
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
    // |pi/2 - asin(a)| <= 1e-7 holds for exactly the doubles 1 - 45 * 2^-53 <= a <= 1 (above 1 the
    // arcsine is NaN and the test fails): the gimbal test needs no arcsine.
    constexpr double kGimbalMin = 1.0 - 45.0 * 0x1p-53;
    const double a7 = fabs(R[7]);
    double th1, th2, th3;
    if (a7 >= kGimbalMin && a7 <= 1.0) {
        const double s = (-R[7] > 0.0) ? 1.0 : ((-R[7] < 0.0) ? -1.0 : 0.0);
        th2 = s * kHalfPi;
        th3 = 0.0;
        th1 = atan2(-R[3], R[0]);
    } else {
        th1 = atan2(R[1], R[4]);
        th3 = atan2(R[6], R[8]);
        // cos(th1) = R4 / h1 and cos(th3) = R8 / h3 with h1 = |(R1, R4)|, h3 = |(R6, R8)|: dividing
        // R4 (R8) by the cosine of larger magnitude, as the reference does, returns h1 (h3).
        const double h1 = sqrt_pos(__builtin_fma(R[1], R[1], R[4] * R[4]));
        const double h3 = sqrt_pos(__builtin_fma(R[6], R[6], R[8] * R[8]));
        const double c2 = (fabs(R[4]) * h3 > fabs(R[8]) * h1) ? h1 : h3;
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

}  // namespace pnp
