"""MI355X-native batched PnP pose solver behind the reference's `PNP_SOLVER` interface.

    from pnp_solver_test_amd import PNP_SOLVER, solve_pnp_batch

The solvers run in hand-written HIP kernels (libpnp_hip.so, built in-tree for gfx950) that
Python reaches through a C ABI with ctypes.  Importing this package loads that library and
raises ImportError if it has not been built: there is no CPU implementation to fall back to.
"""
from . import _lib
from .batch import (RcclComm, compare_batch, euler_deg_batch, gather_results, make_params, project_batch,
                    shard_bounds, solve_pnp_batch)
from .PNP_SOLVER_LIB import LM_KEY_LIST, PNP_SOLVER

__all__ = ["PNP_SOLVER", "LM_KEY_LIST", "solve_pnp_batch", "euler_deg_batch", "make_params",
           "shard_bounds", "gather_results", "RcclComm", "project_batch", "compare_batch", "device_count", "version"]


def device_count() -> int:
    """Number of HIP devices the library sees."""
    n = _lib.load().pnp_device_count()
    _lib.check(n)
    return n


def version() -> str:
    return _lib.load().pnp_version().decode()
