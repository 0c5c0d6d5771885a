"""Synthetic inputs in the reference drivers' own distributions (host side, vectorised NumPy).

These reproduce, for a whole batch at once, what scripts/random_stress_test.py does one
sample at a time: the pose draws (:245-258), Euler -> R (PNP_SOLVER_LIB.py:4442-4472) and the
pin-hole projection with optional pixel quantisation (:4532-4557).
"""
from __future__ import annotations

import numpy as np

F_CAMERA = 225.68717584155982                      # random_stress_test.py:103
K_DEFAULT = np.array([[F_CAMERA, 0.0, 160.0], [0.0, F_CAMERA, 120.0], [0.0, 0.0, 1.0]])

#: the six landmarks solve_pnp uses, in its order (PNP_SOLVER_LIB.py:156)
LM6_KEYS = ("eye_l_96", "eye_r_97", "mouse_l_76", "mouse_r_82", "nose_t_54", "chin_t_16")

#: the two 15-landmark golden face patterns of the reference's drivers, metres, in the key order
#: of TEST_TOOLBOX.get_golden_pattern (TEST_TOOLBOX.py:67-128)
GOLDEN_PATTERNS = {
    "Alexander": {
        "eye_l_96": [0.032, 0.0, 0.0], "eye_r_97": [-0.032, 0.0, 0.0], "eye_c_51": [0.0, 0.0, -0.015],
        "mouse_l_76": [0.027, 0.070, 0.0], "mouse_r_82": [-0.027, 0.070, 0.0],
        "nose_t_54": [-0.005, 0.0455, -0.03], "chin_t_16": [0.0, 0.12, 0.0],
        "brow_cl_35": [0.035, -0.0228, 0.0], "brow_il_37": [0.0135, -0.017, 0.0],
        "brow_ir_42": [-0.0135, -0.017, 0.0], "brow_cr_44": [-0.035, -0.0228, 0.0],
        "eye_lo_60": [0.047, 0.0, 0.005], "eye_li_64": [0.018, 0.0, 0.0],
        "eye_ro_72": [-0.047, 0.0, 0.005], "eye_ri_68": [-0.018, 0.0, 0.0],
    },
    "Holly": {
        "eye_l_96": [0.028, 0.0, 0.0], "eye_r_97": [-0.028, 0.0, 0.0], "eye_c_51": [0.0, 0.0, 0.0],
        "mouse_l_76": [0.025, 0.060, 0.0], "mouse_r_82": [-0.025, 0.060, 0.0],
        "nose_t_54": [0.0, 0.039, -0.03], "chin_t_16": [0.0, 0.098, 0.0],
        "brow_cl_35": [0.035, -0.0228, 0.0], "brow_il_37": [0.0135, -0.017, 0.0],
        "brow_ir_42": [-0.0135, -0.017, 0.0], "brow_cr_44": [-0.035, -0.0228, 0.0],
        "eye_lo_60": [0.046, 0.0, 0.0], "eye_li_64": [0.01, 0.0, 0.0],
        "eye_ro_72": [-0.046, 0.0, 0.0], "eye_ri_68": [-0.01, 0.0, 0.0],
    },
}


def get_golden_pattern(pattern_name="Alexander"):
    """dict {landmark: [x, y, z]} as the reference's get_golden_pattern returns it (any name other
    than "Holly" yields "Alexander", as there)."""
    src = GOLDEN_PATTERNS["Holly" if pattern_name == "Holly" else "Alexander"]
    return {k: list(v) for k, v in src.items()}


def pattern_rows(pattern_name="Alexander", keys=None):
    """(N,3) float64 rows of a golden pattern for `keys` (default: all 15, in dict order)."""
    src = GOLDEN_PATTERNS["Holly" if pattern_name == "Holly" else "Alexander"]
    return np.array([src[k] for k in (keys or src)], dtype=np.float64)


#: rows of "Alexander" for the six solver landmarks
PATTERN6 = pattern_rows("Alexander", LM6_KEYS)


def face68(seed: int = 0) -> np.ndarray:
    """68 points uniform in a face-sized box; synthetic (the reference's patterns have 15)."""
    return np.random.default_rng(seed).uniform([-0.07, -0.05, -0.04], [0.07, 0.12, 0.02], (68, 3))


def sample_stress_poses(rng: np.random.Generator, count: int):
    """`count` poses of the random-stress distribution, consuming the generator exactly as the
    per-sample loop does (six uniform draws per sample, in the order roll, pitch, yaw, depth,
    FOVx, FOVy).  Returns roll, pitch, yaw in degrees and t (count, 3) in metres."""
    u = rng.random((count, 6))
    roll = -45.0 + 90.0 * u[:, 0]
    pitch = -45.0 + 90.0 * u[:, 1]
    yaw = -45.0 + 90.0 * u[:, 2]
    depth = (20.0 + 205.0 * u[:, 3]) / 100.0
    fov_x = -45.0 + 90.0 * u[:, 4]
    fov_y = -45.0 + 90.0 * u[:, 5]
    t = np.stack([depth * np.tan(np.deg2rad(fov_x)), depth * np.tan(np.deg2rad(fov_y)), depth], axis=1)
    return roll, pitch, yaw, t


def rotation_from_euler_deg(roll, yaw, pitch) -> np.ndarray:
    """(B,3,3): R = E_roll E_pitch E_yaw with the yaw sign flipped first."""
    r, y, p = np.deg2rad(roll), np.deg2rad(-np.asarray(yaw)), np.deg2rad(pitch)
    c1, s1, c2, s2, c3, s3 = np.cos(r), np.sin(r), np.cos(y), np.sin(y), np.cos(p), np.sin(p)
    z, o = np.zeros_like(c1), np.ones_like(c1)
    e_roll = np.stack([c1, s1, z, -s1, c1, z, z, z, o], axis=-1).reshape(-1, 3, 3)
    e_yaw = np.stack([c2, z, -s2, z, o, z, s2, z, c2], axis=-1).reshape(-1, 3, 3)
    e_pitch = np.stack([o, z, z, z, c3, s3, z, -s3, c3], axis=-1).reshape(-1, 3, 3)
    return e_roll @ e_pitch @ e_yaw


def project(P, K, R, t, is_quantized=False, q=1.0) -> np.ndarray:
    """uv (B,N,2) float64 of pattern(s) P, (N,3) or (B,N,3), under poses R (B,3,3), t (B,3):
    ray = K(Rp+t), divided by |ray_z|, optionally rounded half-to-even onto a grid of step q."""
    P = np.asarray(P, dtype=np.float64)
    if P.ndim == 2:
        cam = np.einsum("bij,nj->bni", R, P) + t[:, None, :]
    else:
        cam = np.einsum("bij,bnj->bni", R, P) + t[:, None, :]
    ray = cam @ np.asarray(K, dtype=np.float64).T
    proj = ray / np.abs(ray[..., 2:3])
    if is_quantized:
        proj = np.around(proj / q) * q
    return np.ascontiguousarray(proj[..., 0:2])


def stress_batch(count: int, seed: int = 42, pattern=PATTERN6, K=K_DEFAULT, is_quantized=True, chunk=1 << 18):
    """Inputs of `count` random-stress problems: uv (count,N,2) float32 plus the ground truth
    (roll, pitch, yaw in degrees, t)."""
    rng = np.random.default_rng(seed)
    roll, pitch, yaw, t = sample_stress_poses(rng, count)
    uv = np.empty((count, len(pattern), 2), dtype=np.float32)
    for lo in range(0, count, chunk):
        hi = min(count, lo + chunk)
        R = rotation_from_euler_deg(roll[lo:hi], yaw[lo:hi], pitch[lo:hi])
        uv[lo:hi] = project(pattern, K, R, t[lo:hi], is_quantized)
    return uv, dict(roll=roll, pitch=pitch, yaw=yaw, t=t)
