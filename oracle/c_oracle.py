"""ctypes front end of the C oracle (oracle/pnp_oracle.c).  TEST INFRASTRUCTURE ONLY."""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libpnp_oracle.so")


class QeifParams(ctypes.Structure):
    _fields_ = [("q_inv", ctypes.c_double), ("r_quat", ctypes.c_double),
                ("r_delta", ctypes.c_double), ("omega0", ctypes.c_double),
                ("max_it", ctypes.c_int32), ("rel_tol", ctypes.c_double),
                ("res_old0", ctypes.c_double)]


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "pnp_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = ctypes.CDLL(_SO)
        dp = ctypes.POINTER(ctypes.c_double)
        ip = ctypes.POINTER(ctypes.c_int32)
        _lib.oracle_qeif_batch.argtypes = [dp, dp, ctypes.c_int64, dp, ctypes.c_int64, ctypes.c_int32,
                                           ctypes.POINTER(QeifParams), dp, dp, dp, ip, dp]
        _lib.oracle_qeif_batch.restype = None
        _lib.oracle_dlt_batch.argtypes = [dp, dp, ctypes.c_int64, dp, ctypes.c_int64, ctypes.c_int32,
                                          ctypes.c_int32, dp, dp, dp, dp, dp]
        _lib.oracle_dlt_batch.restype = None
        _lib.oracle_euler_deg.argtypes = [dp, ctypes.c_int64, dp]
        _lib.oracle_euler_deg.restype = None
    return _lib


def _dp(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_double))


def _prep(uv, xyz, K):
    uv = np.ascontiguousarray(uv, dtype=np.float64)
    xyz = np.ascontiguousarray(xyz, dtype=np.float64)
    K = np.ascontiguousarray(K, dtype=np.float64)
    B, N = uv.shape[0], uv.shape[1]
    if xyz.ndim == 2:
        assert xyz.shape == (N, 3)
        stride = 0
    else:
        assert xyz.shape == (B, N, 3)
        stride = 3 * N
    return uv, xyz, K, B, N, stride


def qeif_batch(uv, xyz, K, **params):
    """uv (B,N,2), xyz (N,3) or (B,N,3), K (3,3) -> dict of float64 arrays."""
    uv, xyz, K, B, N, stride = _prep(uv, xyz, K)
    p = QeifParams((225.68 ** 2) / 9.0, 0.1, 0.01, 1e-5, 14, 1e-2, 1e-7)
    for k, v in params.items():
        setattr(p, k, v)
    R = np.empty((B, 3, 3)); t = np.empty((B, 3)); res = np.empty(B)
    iters = np.empty(B, dtype=np.int32); rpy = np.empty((B, 3))
    lib().oracle_qeif_batch(_dp(uv), _dp(xyz), stride, _dp(K), B, N, ctypes.byref(p), _dp(R), _dp(t),
                            _dp(res), iters.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)), _dp(rpy))
    return dict(R=R, t=t, t3=t[:, 2].copy(), res_norm=res, iters=iters, rpy=rpy)


def dlt_batch(uv, xyz, K, num_it=3):
    uv, xyz, K, B, N, stride = _prep(uv, xyz, K)
    R = np.empty((B, 3, 3)); t = np.empty((B, 3)); res = np.empty(B)
    rpy = np.empty((B, 3)); margins = np.empty((B, 3))
    lib().oracle_dlt_batch(_dp(uv), _dp(xyz), stride, _dp(K), B, N, num_it, _dp(R), _dp(t), _dp(res),
                           _dp(rpy), _dp(margins))
    return dict(R=R, t=t, t3=t[:, 2].copy(), res_norm=res, rpy=rpy, margins=margins)


def euler_deg(R):
    R = np.ascontiguousarray(R, dtype=np.float64).reshape(-1, 9)
    rpy = np.empty((R.shape[0], 3))
    lib().oracle_euler_deg(_dp(R), R.shape[0], _dp(rpy))
    return rpy
