"""Top-level stand-in for the reference's `scripts/PNP_SOLVER_LIB.py`.

Put this directory first on sys.path (e.g. PYTHONPATH=/path/to/repo/dropin) and a driver's
`import PNP_SOLVER_LIB as PNPS` (random_stress_test.py:9, LM_noise_test.py:9) resolves here;
everything is re-exported from the package, whose solvers run on the GPU.
"""
import os
import sys

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if _ROOT not in sys.path:
    sys.path.insert(1, _ROOT)

from pnp_solver_test_amd.PNP_SOLVER_LIB import LM_KEY_LIST, PNP_SOLVER  # noqa: E402,F401
