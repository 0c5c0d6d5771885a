"""ctypes binding of libpnp_hip.so (C ABI declared in include/pnp_hip.h).

The library is built in-tree by pnp_solver_test_amd/csrc/Makefile.  There is no CPU
fallback: if the shared object is missing or cannot be loaded, importing the solver fails
with an explicit error.

One process, one HIP runtime.  The PyTorch-ROCm wheel bundles its own libamdhip64.so.7 /
libhsa-runtime64.so; a process that brings up both that copy and /opt/rocm's ends up with two
runtimes, and the one initialised second sees no device.  libpnp_hip.so links the runtime by
soname only, so the copy already resident wins: when torch is installed it is therefore
imported BEFORE the library is opened, and both then share torch's runtime (and device
pointers, streams and events are interchangeable).  Without torch the library uses the
system runtime.  Set PNP_HIP_NO_TORCH=1 to skip the import.
"""
from __future__ import annotations

import ctypes
import importlib.util
import os
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
#: PNP_HIP_LIB selects another build of the same library (A/B timing of kernel variants)
LIB_PATH = os.environ.get("PNP_HIP_LIB") or os.path.join(_HERE, "libpnp_hip.so")

PTR_HOST = 0
PTR_DEVICE = 1
QEIF_FORM_AUTO, QEIF_FORM_POINTS, QEIF_FORM_MOMENTS = 0, 1, 2
NOALLOC_SCRATCH_BYTES = 16 << 20      # PNP_NOALLOC_SCRATCH_BYTES
COMM_ID_BYTES = 128                   # PNP_COMM_ID_BYTES

#: every symbol include/pnp_hip.h declares
EXPORTS = (
    "pnp_qeif_default_params", "pnp_device_count", "pnp_init", "pnp_qeif_batch_f32",
    "pnp_dlt_batch_f32", "pnp_lm12_batch_f32", "pnp_euler_deg_f32", "pnp_qeif_multi_batch_f32",
    "pnp_qeif_batch_f64", "pnp_dlt_batch_f64", "pnp_lm12_batch_f64", "pnp_euler_deg_f64",
    "pnp_qeif_multi_batch_f64", "pnp_project_batch_f64", "pnp_compare_batch_f32", "pnp_compare_batch_f64",
    "pnp_set_lanes_per_problem", "pnp_get_lanes_per_problem", "pnp_set_qeif_form", "pnp_get_qeif_form",
    "pnp_max_points", "pnp_qeif_batch_noalloc_f32", "pnp_last_kernel_name", "pnp_last_launch",
    "pnp_host_lane_threads",
    "pnp_comm_unique_id", "pnp_comm_init", "pnp_comm_destroy", "pnp_gather_results",
    "pnp_last_error", "pnp_version",
)


class QeifParams(ctypes.Structure):
    """Mirror of `pnp_qeif_params`."""
    _fields_ = [("q_inv", ctypes.c_double), ("r_quat", ctypes.c_double),
                ("r_delta", ctypes.c_double), ("omega0", ctypes.c_double),
                ("max_it", ctypes.c_int32), ("rel_tol", ctypes.c_double),
                ("res_old0", ctypes.c_double)]


class PnpError(RuntimeError):
    """A C-ABI call returned a negative status."""


_lib = None


def load():
    """Load libpnp_hip.so once and declare the argument types of its entry points."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "%s not found: build it with `make -C pnp_solver_test_amd/csrc` (or "
            "`python -c 'import __graft_entry__ as g; g.build()'`). There is no CPU fallback." % LIB_PATH)
    if ("torch" not in sys.modules and not os.environ.get("PNP_HIP_NO_TORCH")
            and importlib.util.find_spec("torch") is not None):
        import torch  # noqa: F401  (see module docstring: share torch's HIP runtime)
    lib = ctypes.CDLL(LIB_PATH)
    vp, i32, i64 = ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64
    pp = ctypes.POINTER(QeifParams)
    dp = ctypes.POINTER(ctypes.c_double)
    lib.pnp_qeif_default_params.argtypes = [pp]
    lib.pnp_qeif_default_params.restype = None
    lib.pnp_device_count.argtypes = []
    lib.pnp_device_count.restype = ctypes.c_int
    lib.pnp_init.argtypes = [ctypes.c_int]
    lib.pnp_init.restype = ctypes.c_int
    for sfx in ("f32", "f64"):      # the two flavours share their signatures up to the element type
        for name, args in (
                ("pnp_qeif_batch_", [vp, vp, i64, dp, i64, i32, pp, vp, vp, vp, vp, vp, i32, vp]),
                ("pnp_dlt_batch_", [vp, vp, i64, dp, i64, i32, i32, vp, vp, vp, vp, i32, vp]),
                ("pnp_lm12_batch_", [vp, vp, i64, dp, i64, i32, ctypes.c_double, i32, vp, vp, vp, vp, i32, vp]),
                ("pnp_euler_deg_", [vp, i64, vp, i32, vp]),
                ("pnp_qeif_multi_batch_", [vp, vp, i32, dp, i64, i32, pp, vp, vp, vp, vp, vp, vp, i32, vp]),
                ("pnp_compare_batch_", [vp, vp, vp, vp, vp, vp, dp, i64, i32, dp, i32, vp, vp, vp, vp, i32, vp])):
            fn = getattr(lib, name + sfx)
            fn.argtypes = args
            fn.restype = ctypes.c_int
    lib.pnp_qeif_batch_noalloc_f32.argtypes = [vp, vp, i64, dp, i64, i32, pp, vp, vp, vp, vp, vp, vp]
    lib.pnp_qeif_batch_noalloc_f32.restype = ctypes.c_int
    lib.pnp_comm_unique_id.argtypes = [vp]
    lib.pnp_comm_unique_id.restype = ctypes.c_int
    lib.pnp_comm_init.argtypes = [vp, i32, i32, ctypes.POINTER(vp)]
    lib.pnp_comm_init.restype = ctypes.c_int
    lib.pnp_comm_destroy.argtypes = [vp]
    lib.pnp_comm_destroy.restype = ctypes.c_int
    lib.pnp_gather_results.argtypes = [vp, vp, i64, i32, i32, i32, vp, vp]
    lib.pnp_gather_results.restype = ctypes.c_int
    lib.pnp_project_batch_f64.argtypes = [vp, vp, vp, i64, dp, i64, i32, ctypes.c_double, vp, i32, i32, vp]
    lib.pnp_project_batch_f64.restype = ctypes.c_int
    lib.pnp_set_lanes_per_problem.argtypes = [i32]
    lib.pnp_set_lanes_per_problem.restype = ctypes.c_int
    lib.pnp_set_qeif_form.argtypes = [i32]
    lib.pnp_set_qeif_form.restype = ctypes.c_int
    lib.pnp_get_qeif_form.argtypes = [i64, i32]
    lib.pnp_get_qeif_form.restype = ctypes.c_int
    lib.pnp_get_lanes_per_problem.argtypes = [i64, i32, i64]
    lib.pnp_get_lanes_per_problem.restype = ctypes.c_int
    lib.pnp_max_points.argtypes = []
    lib.pnp_max_points.restype = ctypes.c_int
    lib.pnp_last_kernel_name.argtypes = []
    lib.pnp_last_kernel_name.restype = ctypes.c_char_p
    lib.pnp_last_launch.argtypes = [ctypes.POINTER(i64), ctypes.POINTER(i32), ctypes.POINTER(i32)]
    lib.pnp_last_launch.restype = None
    lib.pnp_host_lane_threads.argtypes = [ctypes.c_int]
    lib.pnp_host_lane_threads.restype = ctypes.c_int
    lib.pnp_last_error.argtypes = []
    lib.pnp_last_error.restype = ctypes.c_char_p
    lib.pnp_version.argtypes = []
    lib.pnp_version.restype = ctypes.c_char_p
    _lib = lib
    return lib


def check(rc: int) -> None:
    if rc < 0:
        raise PnpError("libpnp_hip: %s (code %d)" % (load().pnp_last_error().decode(), rc))


def last_launch() -> dict:
    """What the calling thread's last solver launch was: kernel name as a profiler shows it, and
    grid / block / dynamic LDS (pnp_last_kernel_name, pnp_last_launch)."""
    lib = load()
    g, b, l = ctypes.c_int64(), ctypes.c_int32(), ctypes.c_int32()
    lib.pnp_last_launch(ctypes.byref(g), ctypes.byref(b), ctypes.byref(l))
    return dict(kernel=lib.pnp_last_kernel_name().decode(), grid=g.value, block=b.value, lds_bytes=l.value)


def default_params() -> QeifParams:
    p = QeifParams()
    load().pnp_qeif_default_params(ctypes.byref(p))
    return p


def k_array(K):
    """3x3 camera matrix -> C double[9]."""
    import numpy as np
    K = np.ascontiguousarray(K, dtype=np.float64).reshape(9)
    return (ctypes.c_double * 9)(*K.tolist())
