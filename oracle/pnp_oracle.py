"""CPU oracle (NumPy, float64) for the PnP hot path.  TEST INFRASTRUCTURE ONLY.

This module restates, one problem at a time and in float64, the arithmetic of the
reference solver so that the HIP kernels can be checked against it.  Nothing in the
product package imports it: only tests/, __graft_entry__.smoke() and the cpu_baseline
leg of bench.py do, and only as the checker.

Parity pin: every function here is checked against golden vectors produced by running
the reference itself (oracle/gen_golden.py, fixtures under tests/golden/); see
tests/test_oracle_golden.py.  The reference ships no expected-output fixtures of its own.

All citations are file:line in the reference's scripts/PNP_SOLVER_LIB.py unless another
file is named.  Linear algebra goes through the same numpy.linalg entry points the
reference uses (inv / pinv), so the restatement is operation-for-operation.
"""
from __future__ import annotations

import numpy as np

# ---------------------------------------------------------------------------
# Constants of the reference's test set-up
# ---------------------------------------------------------------------------

#: camera of every shipped driver (random_stress_test.py:103-110)
F_CAMERA = 225.68717584155982
K_DEFAULT = np.array([[F_CAMERA, 0.0, 160.0], [0.0, F_CAMERA, 120.0], [0.0, 0.0, 1.0]])

#: the fixed six-landmark subset solve_pnp uses (:156), in this order
LM6_KEYS = ("eye_l_96", "eye_r_97", "mouse_l_76", "mouse_r_82", "nose_t_54", "chin_t_16")

#: "Alexander" golden face pattern, insertion order of TEST_TOOLBOX.py:105-119
PATTERN_ALEXANDER = {
    "eye_l_96": [0.032, 0.0, 0.0],
    "eye_r_97": [-0.032, 0.0, 0.0],
    "eye_c_51": [0.0, 0.0, -0.015],
    "mouse_l_76": [0.027, 0.070, 0.0],
    "mouse_r_82": [-0.027, 0.070, 0.0],
    "nose_t_54": [-0.005, 0.0455, -0.03],
    "chin_t_16": [0.0, 0.12, 0.0],
    "brow_cl_35": [0.035, -0.0228, 0.0],
    "brow_il_37": [0.0135, -0.017, 0.0],
    "brow_ir_42": [-0.0135, -0.017, 0.0],
    "brow_cr_44": [-0.035, -0.0228, 0.0],
    "eye_lo_60": [0.047, 0.0, 0.005],
    "eye_li_64": [0.018, 0.0, 0.00],
    "eye_ro_72": [-0.047, 0.0, 0.005],
    "eye_ri_68": [-0.018, 0.0, 0.00],
}

#: "Holly" golden face pattern (TEST_TOOLBOX.py:76-90)
PATTERN_HOLLY = {
    "eye_l_96": [0.028, 0.0, 0.0],
    "eye_r_97": [-0.028, 0.0, 0.0],
    "eye_c_51": [0.0, 0.0, 0.0],
    "mouse_l_76": [0.025, 0.060, 0.0],
    "mouse_r_82": [-0.025, 0.060, 0.0],
    "nose_t_54": [0.00, 0.039, -0.03],
    "chin_t_16": [0.0, 0.098, 0.0],
    "brow_cl_35": [0.035, -0.0228, 0.0],
    "brow_il_37": [0.0135, -0.017, 0.0],
    "brow_ir_42": [-0.0135, -0.017, 0.0],
    "brow_cr_44": [-0.035, -0.0228, 0.0],
    "eye_lo_60": [0.046, 0.0, 0.0],
    "eye_li_64": [0.01, 0.0, 0.0],
    "eye_ro_72": [-0.046, 0.0, 0.0],
    "eye_ri_68": [-0.01, 0.0, 0.0],
}


def pattern_array(pattern: dict, keys=None, scale: float = 1.0) -> np.ndarray:
    """(N,3) float64 array of the selected landmarks times `scale` (:84-85, :3260-3272)."""
    if keys is None:
        keys = list(pattern.keys())
    return np.array([pattern[k] for k in keys], dtype=np.float64) * scale


def synthetic_face68(seed: int = 0) -> np.ndarray:
    """The 68-point synthetic pattern of the N=68 configuration: points uniform in a
    face-sized box.  The reference has no 68-point pattern (its largest has 15)."""
    rng = np.random.default_rng(seed)
    return rng.uniform([-0.07, -0.05, -0.04], [0.07, 0.12, 0.02], (68, 3))


# ---------------------------------------------------------------------------
# Euler <-> rotation (generator side and return tuple of solve_pnp)
# ---------------------------------------------------------------------------

def euler_to_R(roll, yaw, pitch, is_degree=False) -> np.ndarray:
    """R = E_roll @ E_pitch @ E_yaw with the yaw sign flipped first (:4442-4472)."""
    yaw = -yaw
    if is_degree:
        roll, yaw, pitch = np.deg2rad(roll), np.deg2rad(yaw), np.deg2rad(pitch)
    c1, s1 = np.cos(roll), np.sin(roll)
    c2, s2 = np.cos(yaw), np.sin(yaw)
    c3, s3 = np.cos(pitch), np.sin(pitch)
    e_roll = np.array([[c1, s1, 0.0], [-s1, c1, 0.0], [0.0, 0.0, 1.0]])
    e_yaw = np.array([[c2, 0.0, -s2], [0.0, 1.0, 0.0], [s2, 0.0, c2]])
    e_pitch = np.array([[1.0, 0.0, 0.0], [0.0, c3, s3], [0.0, -s3, c3]])
    return e_roll @ e_pitch @ e_yaw


def R_to_euler(R, is_degree=False):
    """(roll, yaw, pitch) from R, gimbal-lock branch included, yaw negated (:4474-4517)."""
    eps = 10.0 ** -7
    if np.abs(np.pi / 2.0 - np.arcsin(np.abs(R[2, 1]))) <= eps:
        th2 = np.sign(-R[2, 1]) * (np.pi / 2.0)
        th3 = 0.0
        th1 = np.arctan2(-R[1, 0], R[0, 0])
    else:
        th1 = np.arctan2(R[0, 1], R[1, 1])
        th3 = np.arctan2(R[2, 0], R[2, 2])
        c1, c3 = np.cos(th1), np.cos(th3)
        c2 = R[1, 1] / c1 if np.abs(c1) > np.abs(c3) else R[2, 2] / c3
        th2 = np.arctan2(-R[2, 1], c2)
    roll, pitch, yaw = th1, th2, th3
    if is_degree:
        roll, yaw, pitch = np.rad2deg(roll), np.rad2deg(yaw), np.rad2deg(pitch)
    return roll, -yaw, pitch


# ---------------------------------------------------------------------------
# Pin-hole projection (synthetic-input generator)
# ---------------------------------------------------------------------------

def project(P, K, R, t, is_quantized=False, q=1.0) -> np.ndarray:
    """Project the rows of P (N,3); returns uv (N,2).  ray = K(Rp+t), divided by |ray_z|,
    optionally rounded half-to-even onto a grid of step q (:4532-4557)."""
    ray = (K @ (R @ P.T + np.reshape(t, (3, 1)))).T          # (N,3)
    proj = ray / np.abs(ray[:, 2:3])
    if is_quantized:
        proj = np.around(proj / q) * q
    return proj[:, 0:2].copy()


def normalized_xy(uv, K):
    """nu = K^-1 [u,v,1]^T; returns (Bx, By) = first two components, NOT divided by the
    third (:2811-2812, :3291-3312; :3094-3111 for the DLT path).  uv (N,2), or (N,3) rows
    [u,v,w] with the homogeneous component the image dict carries (w = -1 behind the camera)."""
    K_inv = np.linalg.inv(K)
    uv = np.asarray(uv, dtype=np.float64)
    hom = uv if uv.shape[1] == 3 else np.concatenate([uv, np.ones((len(uv), 1))], axis=1)
    nu = (K_inv @ hom.T).T
    return nu[:, 0].copy(), nu[:, 1].copy()


# ---------------------------------------------------------------------------
# QEIF: quaternion iterated information filter (the live solver)
# ---------------------------------------------------------------------------

QEIF_DEFAULTS = dict(
    q_inv=(225.68 ** 2) / 9.0,   # 1/Q, Q = 3^2/f^2 with the hard-coded f = 225.68 (:2845-2846)
    r_quat=0.1,                  # process noise on q (:2792)
    r_delta=0.01,                # process noise on delta_1, delta_2 (:2793)
    omega0=1e-5,                 # initial information (:2836)
    max_it=14,                   # :2857
    rel_tol=1e-2,                # :2952
    res_old0=1e-7,               # :2863
)


def _gamma_rows(q):
    """Rows phi_1..phi_3 of the unnormalised quaternion rotation Gamma(q) and gamma=|q|^2
    (:3918-3936, same expressions at :3555-3574)."""
    qr, qi, qj, qk = q
    gamma = np.linalg.norm(q) ** 2
    qii, qjj, qkk = qi * qi, qj * qj, qk * qk
    qij, qjk, qik = qi * qj, qj * qk, qi * qk
    qri, qrj, qrk = qr * qi, qr * qj, qr * qk
    phi1 = np.array([gamma - 2 * (qjj + qkk), 2 * (qij - qrk), 2 * (qik + qrj)])
    phi2 = np.array([2 * (qij + qrk), gamma - 2 * (qii + qkk), 2 * (qjk - qri)])
    phi3 = np.array([2 * (qik - qrj), 2 * (qjk + qri), gamma - 2 * (qii + qjj)])
    return gamma, phi1, phi2, phi3


def qeif_model(x, Bx, By, P):
    """Measurement model h(x) (2N,) and its Jacobian H (2N,6) (:3902-3983)."""
    n = P.shape[0]
    qr, qi, qj, qk, d1, d2 = x
    _, phi1, phi2, phi3 = _gamma_rows(x[0:4])
    a, b, c, d = 2.0 * qr, 2.0 * qi, 2.0 * qj, 2.0 * qk
    Q1 = np.array([[a, b, -c, -d], [-d, c, b, -a], [c, d, a, b]])
    Q2 = np.array([[d, c, b, a], [a, -b, c, -d], [-b, -a, d, c]])
    Q3 = np.array([[-c, d, -a, b], [b, a, d, c], [a, -b, -c, d]])
    P1, P2, P3 = P @ phi1, P @ phi2, P @ phi3
    h = np.concatenate([P1 - Bx * P3 + d1, P2 - By * P3 + d2])
    PQ1, PQ2, PQ3 = P @ Q1, P @ Q2, P @ Q3
    one, zero = np.ones((n, 1)), np.zeros((n, 1))
    H = np.vstack([
        np.hstack([PQ1 - Bx[:, None] * PQ3, one, zero]),
        np.hstack([PQ2 - By[:, None] * PQ3, zero, one]),
    ])
    return h, H


def qeif_reconstruct(x):
    """R = Gamma(q)/gamma, t3 = 1/gamma, t = [d1, d2, 1]*t3 (:3542-3609)."""
    gamma, phi1, phi2, phi3 = _gamma_rows(x[0:4])
    R = np.vstack([phi1, phi2, phi3]) / gamma
    t3 = 1.0 / gamma
    t = np.array([x[4], x[5], 1.0]) * t3
    return R, t, t3


def qeif_solve(uv, P, K=K_DEFAULT, **params):
    """One QEIF solve (:2771-3025).

    uv (N,2) pixels, P (N,3) pattern, K (3,3).  Returns a dict with R (3,3), t (3,),
    t3, roll/yaw/pitch (degrees), res_norm (residual norm at the LAST evaluated
    linearisation point, i.e. before the final state update, :2905-2907) and iters.
    """
    p = dict(QEIF_DEFAULTS)
    p.update(params)
    P = np.asarray(P, dtype=np.float64)
    n = P.shape[0]
    Bx, By = normalized_xy(uv, K)
    z = np.concatenate([Bx, By])
    R_proc = np.diag([p["r_quat"]] * 4 + [p["r_delta"]] * 2)
    Q_pinv = np.linalg.pinv(np.eye(2 * n) / p["q_inv"])                 # :2850-2852
    x = np.array([1.0, 0.0, 0.0, 0.0, 0.0, 0.0])                       # :2831-2833
    Omega = np.eye(6) * p["omega0"]
    Omega_pinv = np.linalg.pinv(Omega)                                 # :2837
    res_old = p["res_old0"]
    res_norm = 10.0 ** 5
    k = 0
    while k < p["max_it"]:
        k += 1
        Omega = np.linalg.pinv(Omega_pinv + R_proc)                    # predict, :2887
        zeta = Omega @ x                                               # :2889
        h, H = qeif_model(x, Bx, By, P)                                # :2895
        HtQ = H.T @ Q_pinv
        Omega = Omega + HtQ @ H                                        # :2897
        zeta = zeta + HtQ @ (z - h + H @ x)                            # :2898
        res_norm = np.linalg.norm(z - h)                               # :2905-2907
        Omega_pinv = np.linalg.pinv(Omega)                             # :2924
        x = Omega_pinv @ zeta                                          # :2925
        ratio = (res_norm - res_old) / res_old                         # :2945
        res_old = res_norm
        if abs(ratio) < p["rel_tol"]:                                  # :2952
            break
    R, t, t3 = qeif_reconstruct(x)
    roll, yaw, pitch = R_to_euler(R, is_degree=True)
    return dict(R=R, t=t, t3=t3, roll=roll, yaw=yaw, pitch=pitch, res_norm=res_norm,
                iters=k, x=x)


def qeif_solve_batch(uv, P, K=K_DEFAULT, **params):
    """qeif_solve over a batch in array operations: what a NumPy user of the reference would
    write to solve many problems at once on the CPU (the same recurrence, :2771-3025, problems
    that have met the stopping rule drop out of the arrays).  uv (B,N,2); P (N,3) or (B,N,3).
    The 6x6 inverses are numpy.linalg.inv on stacks where qeif_solve follows the reference's
    pinv; the matrices are well conditioned, results agree to round-off.  Returns R (B,3,3),
    t (B,3), res_norm (B,), iters (B,), x (B,6)."""
    p = dict(QEIF_DEFAULTS)
    p.update(params)
    uv = np.asarray(uv, dtype=np.float64)
    B, n = uv.shape[0], uv.shape[1]
    P = np.asarray(P, dtype=np.float64)
    Pb = np.broadcast_to(P, (B, n, 3))
    K_inv = np.linalg.inv(K)
    nu = np.concatenate([uv, np.ones((B, n, 1))], axis=2) @ K_inv.T
    Bx, By = nu[:, :, 0], nu[:, :, 1]
    z = np.concatenate([Bx, By], axis=1)                               # (B,2N)
    R_proc = np.diag([p["r_quat"]] * 4 + [p["r_delta"]] * 2)
    x = np.zeros((B, 6))
    x[:, 0] = 1.0
    Sigma = np.broadcast_to(np.eye(6) / p["omega0"], (B, 6, 6)).copy()
    res_old = np.full(B, p["res_old0"])
    res_norm = np.full(B, 10.0 ** 5)
    iters = np.zeros(B, dtype=np.int32)
    live = np.arange(B)
    ones, zeros = np.ones((1, n, 1)), np.zeros((1, n, 1))
    for k in range(1, p["max_it"] + 1):
        if live.size == 0:
            break
        xa, Pa, bx, by = x[live], Pb[live], Bx[live], By[live]
        m = live.size
        Obar = np.linalg.inv(Sigma[live] + R_proc)
        zeta = (Obar @ xa[:, :, None])[:, :, 0]
        qr, qi, qj, qk = xa[:, 0], xa[:, 1], xa[:, 2], xa[:, 3]
        gamma = qr * qr + qi * qi + qj * qj + qk * qk
        phi1 = np.stack([gamma - 2 * (qj * qj + qk * qk), 2 * (qi * qj - qr * qk), 2 * (qi * qk + qr * qj)], axis=1)
        phi2 = np.stack([2 * (qi * qj + qr * qk), gamma - 2 * (qi * qi + qk * qk), 2 * (qj * qk - qr * qi)], axis=1)
        phi3 = np.stack([2 * (qi * qk - qr * qj), 2 * (qj * qk + qr * qi), gamma - 2 * (qi * qi + qj * qj)], axis=1)
        a, b, c, d = 2.0 * qr, 2.0 * qi, 2.0 * qj, 2.0 * qk
        Q1 = np.stack([np.stack([a, b, -c, -d], 1), np.stack([-d, c, b, -a], 1), np.stack([c, d, a, b], 1)], axis=1)
        Q2 = np.stack([np.stack([d, c, b, a], 1), np.stack([a, -b, c, -d], 1), np.stack([-b, -a, d, c], 1)], axis=1)
        Q3 = np.stack([np.stack([-c, d, -a, b], 1), np.stack([b, a, d, c], 1), np.stack([a, -b, -c, d], 1)], axis=1)
        P1 = (Pa @ phi1[:, :, None])[:, :, 0]
        P2 = (Pa @ phi2[:, :, None])[:, :, 0]
        P3 = (Pa @ phi3[:, :, None])[:, :, 0]
        h = np.concatenate([P1 - bx * P3 + xa[:, 4:5], P2 - by * P3 + xa[:, 5:6]], axis=1)
        PQ3 = Pa @ Q3
        one, zero = np.broadcast_to(ones, (m, n, 1)), np.broadcast_to(zeros, (m, n, 1))
        H = np.concatenate([np.concatenate([Pa @ Q1 - bx[:, :, None] * PQ3, one, zero], axis=2),
                            np.concatenate([Pa @ Q2 - by[:, :, None] * PQ3, zero, one], axis=2)], axis=1)   # (m,2N,6)
        Ht = np.swapaxes(H, 1, 2)
        r = z[live] - h
        Omega = Obar + p["q_inv"] * (Ht @ H)
        zeta = zeta + p["q_inv"] * (Ht @ (r + (H @ xa[:, :, None])[:, :, 0])[:, :, None])[:, :, 0]
        res = np.linalg.norm(r, axis=1)
        Sig = np.linalg.inv(Omega)
        x[live] = (Sig @ zeta[:, :, None])[:, :, 0]
        Sigma[live] = Sig
        res_norm[live] = res
        iters[live] = k
        go_on = ~(np.abs((res - res_old[live]) / res_old[live]) < p["rel_tol"])
        res_old[live] = res
        live = live[go_on]
    q = x[:, 0:4]
    qr, qi, qj, qk = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    gamma = np.sum(q * q, axis=1)
    R = np.stack([np.stack([gamma - 2 * (qj * qj + qk * qk), 2 * (qi * qj - qr * qk), 2 * (qi * qk + qr * qj)], 1),
                  np.stack([2 * (qi * qj + qr * qk), gamma - 2 * (qi * qi + qk * qk), 2 * (qj * qk - qr * qi)], 1),
                  np.stack([2 * (qi * qk - qr * qj), 2 * (qj * qk + qr * qi), gamma - 2 * (qi * qi + qj * qj)], 1)],
                 axis=1) / gamma[:, None, None]
    t = np.stack([x[:, 4], x[:, 5], np.ones(B)], axis=1) / gamma[:, None]
    return dict(R=R, t=t, res_norm=res_norm, iters=iters, x=x)


# ---------------------------------------------------------------------------
# Iterated linear DLT with closed-form block reconstruction
# ---------------------------------------------------------------------------

def dlt_block_reconstruction(phi, se_force=0):
    """Closed-form (R, t, t3) from phi = [phi1(3), phi2(3), delta1, delta2] using only the
    top-left 2x2 block of Gamma = R/t3 (:4158-4364).  Also returns the two quantities whose
    signs steer the branches: k2 (:4233) and the similarity dot product (:4262).  se_force = +-1
    imposes the sign the similarity would pick (test use: with a pattern in the z = 0 plane the
    similarity is an exact zero and the reference's choice rides on round-off)."""
    phi1, phi2 = phi[0:3], phi[3:6]
    K2 = np.array([[phi1[0], phi1[1]], [phi2[0], phi2[1]]])
    KTK = K2.T @ K2
    k1, k2, k3 = KTK[0, 0], KTK[0, 1], KTK[1, 1]
    D = (k1 - k3) ** 2 + 4 * k2 ** 2
    gamma_2 = 0.5 * ((k1 + k3) + np.sqrt(D))
    gamma = np.sqrt(gamma_2)
    e_se = np.sqrt(gamma_2 - k1)
    d_se = -np.sign(k2) * np.sqrt(gamma_2 - k3)
    beta_se = np.array([e_se, d_se])
    delta_se = np.linalg.inv(K2.T) @ beta_se
    c = np.linalg.det(K2) / gamma
    alpha_se = -(c * delta_se)
    similarity = alpha_se.dot(np.array([phi1[2], phi2[2]]))
    se = -1.0 if similarity < 0.0 else 1.0
    if se_force:
        se = float(se_force)
    alpha, beta = se * alpha_se, se * beta_se
    G = np.array([[K2[0, 0], K2[0, 1], alpha[0]],
                  [K2[1, 0], K2[1, 1], alpha[1]],
                  [beta[0], beta[1], c]])
    R = G / gamma
    t3 = 1.0 / gamma
    t = np.array([phi[6], phi[7], 1.0]) * t3
    return R, t, t3, k2, similarity


def dlt_solve(uv, P, K=K_DEFAULT, num_it=3, se_override=None, rcond=None):
    """Iterated linear DLT (:205-430): `num_it` rounds of  Delta_i = theta_i.phi3 + 1
    (clamped, :3031-3046), A (2N,8) with x/y rows interleaved (:3048-3092), phi = pinv(A) B
    (:293), block reconstruction (:366), phi3 <- R[2,:]/t3 (:4002-4017), and the residual of
    the reconstructed pose (:372-380).  The closing reconstruction (:403) repeats the last
    in-loop one exactly (it does not read phi3), so it is not recomputed.  se_override: one
    forced mirror sign (+1 / -1, 0 = as computed) per round, see dlt_block_reconstruction.
    rcond: None is the reference's call, numpy.linalg.pinv with its default cut-off (1e-15 of the
    largest singular value); a number replaces that cut-off, for tests that look at what a
    solver with a coarser notion of rank returns.  The result also carries the smallest ratio of
    singular values of A over the rounds (sv_ratio)."""
    P = np.asarray(P, dtype=np.float64)
    n = P.shape[0]
    Bx, By = normalized_xy(uv, K)
    B = np.empty(2 * n)
    B[0::2], B[1::2] = Bx, By                                          # :3094-3111
    phi3 = np.array([0.0, 0.0, 1.0])                                   # :220
    min_abs_k2 = np.inf
    min_abs_sim = np.inf
    min_abs_delta = np.inf
    sv_ratio = np.inf
    for rnd in range(num_it):
        Delta = P @ phi3 + 1.0
        eps = 10 ** -7
        small = np.abs(Delta) <= eps
        Delta = np.where(small, np.where(Delta < 0, -eps, eps), Delta)  # :3039-3045
        min_abs_delta = min(min_abs_delta, np.min(np.abs(P @ phi3 + 1.0)))
        A = np.zeros((2 * n, 8))
        A[0::2, 0:3] = P / Delta[:, None]
        A[0::2, 6] = 1.0 / Delta
        A[1::2, 3:6] = P / Delta[:, None]
        A[1::2, 7] = 1.0 / Delta                                       # :3057-3061
        phi = (np.linalg.pinv(A) if rcond is None else np.linalg.pinv(A, rcond=rcond)) @ B   # :293
        sv = np.linalg.svd(A, compute_uv=False)
        sv_ratio = min(sv_ratio, sv[-1] / sv[0]) if np.isfinite(sv).all() and sv[0] > 0 else sv_ratio
        R, t, t3, k2, sim = dlt_block_reconstruction(phi, se_override[rnd] if se_override else 0)
        min_abs_k2 = min(min_abs_k2, abs(k2))
        min_abs_sim = min(min_abs_sim, abs(sim))
        phi3 = R[2, :] / t3                                            # :4006-4010
        phi_new = np.concatenate([R[0, :], R[1, :], t[0:2]]) / t3      # :372
        res_norm = np.linalg.norm(B - A @ phi_new)                     # :374-380
    roll, yaw, pitch = R_to_euler(R, is_degree=True)
    return dict(R=R, t=t, t3=t3, roll=roll, yaw=yaw, pitch=pitch, res_norm=res_norm,
                iters=num_it, min_abs_k2=min_abs_k2, min_abs_sim=min_abs_sim,
                min_abs_delta=min_abs_delta, sv_ratio=sv_ratio)


# ---------------------------------------------------------------------------
# 12-state Levenberg-Marquardt (secondary solver)
# ---------------------------------------------------------------------------

def lm12_model(x, Bx, By, P):
    """Residual model h (2N+9,) and the Jacobian J (2N+9,12) exactly as the 12-state solver
    builds them with its call-site options is_hc0_4_6_linear=False, is_hc1_linear=True
    (:3718-3836, call at :2663).  State x = [u1(3), u2(3), u3(3), delta1, delta2, gamma].
    The constraint-row Jacobians are restated AS WRITTEN there (rows 2N+3..2N+5 carry u,
    not 2u; rows 2N+6..2N+8 carry u/(2|u|)), since they define the iteration."""
    n = P.shape[0]
    u1, u2, u3 = x[0:3], x[3:6], x[6:9]
    d1, d2, g = x[9], x[10], x[11]
    zeros = np.zeros((n, 3))
    Hu1 = np.hstack([P, zeros, -Bx[:, None] * P])                      # :3738
    Hu2 = np.hstack([zeros, P, -By[:, None] * P])                      # :3739
    hu1, hu2 = Hu1 @ x[0:9], Hu2 @ x[0:9]
    one, zero = np.ones((n, 1)), np.zeros((n, 1))
    h = np.zeros(2 * n + 9)
    J = np.zeros((2 * n + 9, 12))
    J[0:n] = np.hstack([g * Hu1, one, zero, hu1[:, None]])             # :3743
    J[n:2 * n] = np.hstack([g * Hu2, zero, one, hu2[:, None]])         # :3744
    h[0:n] = g * hu1 + d1
    h[n:2 * n] = g * hu2 + d2
    r = 2 * n
    h[r + 0], h[r + 1], h[r + 2] = u1 @ u3, u2 @ u3, u1 @ u2           # :3753-3755
    h[r + 3] = u1 @ u1 - u3 @ u3                                       # :3764-3766
    h[r + 4] = u2 @ u2 - u3 @ u3
    h[r + 5] = u1 @ u1 - u2 @ u2
    h[r + 6], h[r + 7], h[r + 8] = (np.sqrt(u1 @ u1), np.sqrt(u2 @ u2),
                                    np.sqrt(u3 @ u3))                  # :3770-3772
    J[r + 0, 0:3], J[r + 0, 6:9] = u3, u1                              # :3787-3794
    J[r + 1, 3:6], J[r + 1, 6:9] = u3, u2
    J[r + 2, 0:3], J[r + 2, 3:6] = u2, u1
    J[r + 3, 0:3], J[r + 3, 6:9] = u1, -u3                             # :3808-3815
    J[r + 4, 3:6], J[r + 4, 6:9] = u2, -u3
    J[r + 5, 0:3], J[r + 5, 3:6] = u1, -u2
    J[r + 6, 0:3] = u1 / (2.0 * np.linalg.norm(u1))                    # :3819-3823
    J[r + 7, 3:6] = u2 / (2.0 * np.linalg.norm(u2))
    J[r + 8, 6:9] = u3 / (2.0 * np.linalg.norm(u3))
    return h, J


def lm12_reconstruct(x):
    """R = U diag(1,1,det(U V^T)) V^T of Gamma = [u1;u2;u3]; t3 = 1/(sigma_max(Gamma) gamma);
    t = [delta1, delta2, 1] t3 (:3500-3540)."""
    Gamma = np.vstack([x[0:3], x[3:6], x[6:9]])
    U, S, Vh = np.linalg.svd(Gamma)
    R = U @ np.diag([1.0, 1.0, np.linalg.det(U @ Vh)]) @ Vh
    t3 = 1.0 / (np.linalg.norm(Gamma, ord=2) * x[11])
    t = np.array([x[9], x[10], 1.0]) * t3
    return R, t, t3


def lm12_solve(uv, P, K=K_DEFAULT, lam=1e-5, num_it=14):
    """The literal 12-state Levenberg-Marquardt solver (:2567-2769): 14 fixed iterations of
    dx = pinv(J^T J + lam I) J^T (z - h), no early exit; res_norm is the reprojection part of
    the residual at the last linearisation point (:2681)."""
    P = np.asarray(P, dtype=np.float64)
    n = P.shape[0]
    Bx, By = normalized_xy(uv, K)
    z = np.concatenate([Bx, By, np.zeros(6), np.ones(3)])              # :2599-2604
    x = np.zeros(12)
    x[0] = x[4] = x[8] = x[11] = 1.0                                   # :2619-2624
    res_norm = 10.0 ** 5
    for _ in range(num_it):
        h, J = lm12_model(x, Bx, By, P)
        JTJ = J.T @ J + lam * np.eye(12)                               # :2666-2667
        dz = z - h
        res_norm = np.linalg.norm(dz[0:2 * n])                         # :2681
        x = x + np.linalg.pinv(JTJ) @ J.T @ dz                         # :2675-2702
    R, t, t3 = lm12_reconstruct(x)
    roll, yaw, pitch = R_to_euler(R, is_degree=True)
    return dict(R=R, t=t, t3=t3, roll=roll, yaw=yaw, pitch=pitch, res_norm=res_norm,
                iters=num_it, x=x)


# ---------------------------------------------------------------------------
# Multi-pattern dispatcher and the random-stress pose sampler
# ---------------------------------------------------------------------------

def solve_pnp_multi(uv, patterns, K=K_DEFAULT, solver=qeif_solve):
    """solve_pnp's loop over golden patterns (:166-199): run `solver` against every pattern
    in order and keep the first one with the strictly smallest res_norm."""
    best, best_idx = None, None
    for idx, P in enumerate(patterns):
        out = solver(uv, P, K)
        if best is None or out["res_norm"] < best["res_norm"]:
            best, best_idx = out, idx
    best = dict(best)
    best["pattern_id"] = best_idx
    return best


def sample_random_stress_poses(rng, count):
    """Pose draws of the random stress driver, in its order: per sample roll, pitch, yaw
    ~ U(-45,45) deg, depth ~ U(20,225)/100 m, FOVx, FOVy ~ U(-45,45) deg;
    t = depth*[tan FOVx, tan FOVy, 1] (random_stress_test.py:245-258).
    Returns roll, pitch, yaw (deg), t (count,3)."""
    roll = np.empty(count)
    pitch = np.empty(count)
    yaw = np.empty(count)
    t = np.empty((count, 3))
    for i in range(count):
        roll[i] = rng.uniform(-45.0, 45.0, None)
        pitch[i] = rng.uniform(-45.0, 45.0, None)
        yaw[i] = rng.uniform(-45.0, 45.0, None)
        depth = rng.uniform(20, 225, None) / 100.0
        fov_x = rng.uniform(-45.0, 45.0, None)
        fov_y = rng.uniform(-45.0, 45.0, None)
        t[i] = [depth * np.tan(np.deg2rad(fov_x)), depth * np.tan(np.deg2rad(fov_y)), depth]
    return roll, pitch, yaw, t


# ---------------------------------------------------------------------------------------------
# comparison epilogue of the drivers (TEST_TOOLBOX.py:55-62, :252-286, :291-465)
# ---------------------------------------------------------------------------------------------

def lm_error_distances(A, B):
    """cal_LM_error_distances (TEST_TOOLBOX.py:252-286) on two (M,3) arrays of homogeneous
    pixels in the same landmark order: (average, max, index of the first strict maximum or -1
    if every distance is 0)."""
    total, mx, key = 0.0, 0.0, -1
    for i in range(A.shape[0]):
        e = np.linalg.norm(A[i] - B[i])
        total += e
        if e > mx:
            mx, key = e, i
    return total / A.shape[0], mx, key


def compare_epilogue(R_est, t_est, rpy_est, rpy_gt, distance_cm, uv_obs, P, K=K_DEFAULT,
                     bound=(10.0, 10.0, 10.0, 10.0)):
    """What the stress driver does with one solved sample (random_stress_test.py:362-396):
    ground-truth rotation from the ground-truth angles (:365), ground-truth translation along
    the ESTIMATED direction scaled to the true depth (:367-368), the pass flags
    (TEST_TOOLBOX.py:55-62; order depth, roll, pitch, yaw) and the three landmark error sets of
    compare_result_and_generate_result_dict (TEST_TOOLBOX.py:312-343, :453-463), each normalised
    by the true depth in metres.  rpy_* are (roll, yaw, pitch) in degrees; uv_obs (M,2) are the
    pixels handed to the solver (None: the ground-truth projections stand in, :317-319).
    Returns dict(drpy_err (4,), passed (4,) bool, avg (3,), max (3,), key (3,), t_gt_est (3,))
    with the error sets ordered LM-vs-GT, reprojection-vs-LM, reprojection-vs-GT."""
    t_est = np.asarray(t_est, dtype=np.float64).reshape(3)
    R_gt = euler_to_R(rpy_gt[0], rpy_gt[1], rpy_gt[2], is_degree=True)
    distance_m = distance_cm * 0.01
    t_gt_est = (t_est / t_est[2]) * distance_m

    def homog(R, t):
        ray = (K @ (R @ P.T + t.reshape(3, 1))).T
        return ray / np.abs(ray[:, 2:3])

    reproj, gt = homog(np.asarray(R_est, dtype=np.float64), t_est), homog(R_gt, t_gt_est)
    obs = gt if uv_obs is None else np.concatenate([np.asarray(uv_obs, dtype=np.float64),
                                                    np.ones((P.shape[0], 1))], axis=1)
    sets = [lm_error_distances(obs, gt), lm_error_distances(reproj, obs), lm_error_distances(reproj, gt)]
    est = np.array([t_est[2] * 100.0, rpy_est[0], rpy_est[2], rpy_est[1]])
    ref = np.array([distance_cm, rpy_gt[0], rpy_gt[2], rpy_gt[1]])
    return dict(drpy_err=est - ref, passed=np.abs(est - ref) < np.asarray(bound),
                avg=np.array([s[0] for s in sets]) * distance_m, max=np.array([s[1] for s in sets]) * distance_m,
                key=np.array([s[2] for s in sets], dtype=np.int32), t_gt_est=t_gt_est)
