"""Host-side geometry of one pose at a time, float64: the conventions the reference's drivers
rely on (scripts/PNP_SOLVER_LIB.py:4442-4621), written once for the drop-in class.

Angle convention: a pose is R = Z(roll) . X(pitch) . Y(-yaw), where Z, X, Y are the elementary
frame rotations about the camera's optical axis, its x axis and its y axis; the yaw angle enters
with its sign flipped (the reference's "mirror correction", :4451).  Keeping the product as three
3x3 matrix products, in this association, makes the result bit-equal to the reference's.
"""
from __future__ import annotations

import numpy as np

#: axis name -> the (i, j) plane the elementary rotation acts in: +sin at [i, j], -sin at [j, i]
_PLANE = {"z": (0, 1), "x": (1, 2), "y": (2, 0)}
#: |asin|R21|| closer to pi/2 than this is treated as gimbal lock (:4482)
GIMBAL_EPS = 1e-7
#: vectors shorter than this are scaled by 1/UNIT_EPS instead of 1/norm (:4666-4675)
UNIT_EPS = 1e-7


def frame_rotation(axis: str, angle: float) -> np.ndarray:
    """Elementary rotation of the coordinate frame by `angle` (radians) about `axis`."""
    i, j = _PLANE[axis]
    E = np.eye(3)
    E[i, i] = E[j, j] = np.cos(angle)
    E[i, j] = np.sin(angle)
    E[j, i] = -E[i, j]
    return E


def rotation_from_euler(roll, yaw, pitch, is_degree=False) -> np.ndarray:
    """(3,3) rotation of get_rotation_matrix_from_Euler (:4442-4472)."""
    angles = np.array([roll, -yaw, pitch], dtype=np.float64)
    if is_degree:
        angles = np.deg2rad(angles)
    return frame_rotation("z", angles[0]) @ frame_rotation("x", angles[2]) @ frame_rotation("y", angles[1])


def euler_from_rotation(R, is_degree=False):
    """(roll, yaw, pitch) of get_Euler_from_rotation_matrix (:4474-4517).

    With R = Z(a) X(b) Y(c): R01/R11 = tan a, R20/R22 = tan c, R21 = -sin b and
    R11 = cos a cos b, R22 = cos c cos b; cos b is taken from whichever of the last two has the
    larger cosine under it.  At the pole (|sin b| = 1 within GIMBAL_EPS of the arcsine) only
    a + c is defined: c is set to zero and a read off the first column."""
    R = np.asarray(R)
    if np.abs(np.pi / 2.0 - np.arcsin(np.abs(R[2, 1]))) <= GIMBAL_EPS:
        a = np.arctan2(-R[1, 0], R[0, 0])
        b = np.sign(-R[2, 1]) * (np.pi / 2.0)
        c = 0.0
    else:
        a = np.arctan2(R[0, 1], R[1, 1])
        c = np.arctan2(R[2, 0], R[2, 2])
        cos_a, cos_c = np.cos(a), np.cos(c)
        cos_b = R[1, 1] / cos_a if np.abs(cos_a) > np.abs(cos_c) else R[2, 2] / cos_c
        b = np.arctan2(-R[2, 1], cos_b)
    if is_degree:
        a, b, c = np.rad2deg(a), np.rad2deg(b), np.rad2deg(c)
    return (a, -c, b)


def pinhole(point, K, R, t, quantum=None):
    """Homogeneous pixel (3,1) of one 3-D point under pose (R, t): K (R p + t) scaled by the
    absolute value of its third component (a point behind the camera keeps w = -1), then, with a
    `quantum`, snapped half-to-even onto a grid of that step (:4532-4557).  Returns the snapped
    and the unsnapped pixel (the same array when quantum is None)."""
    ray = K @ (R @ np.asarray(point).reshape(3, 1) + t)
    exact = ray / abs(ray[2, 0])
    if quantum is None:
        return exact, exact
    return np.around(exact / quantum) * quantum, exact


def unit(v):
    """v / |v|, or v / UNIT_EPS for a vector shorter than UNIT_EPS (:4666-4675)."""
    n = np.linalg.norm(v)
    return v * (1.0 / UNIT_EPS if np.abs(n) <= UNIT_EPS else 1.0 / n)
