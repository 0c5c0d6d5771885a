#!/usr/bin/env python3
"""Wall-clock of the chunked host-pointer call (1M six-point problems, NumPy arrays in pageable
memory, a pattern per problem) for the library PNP_HIP_LIB selects: fresh result arrays, result
arrays of an earlier call (out=), and the C ABI alone; best and median of 12 calls each."""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import pnp_solver_test_amd as pnp  # noqa: E402
from pnp_solver_test_amd import _lib, synth  # noqa: E402

K = synth.K_DEFAULT
B = 1 << 20
uv, _ = synth.stress_batch(B, seed=42, pattern=synth.PATTERN6, K=K, is_quantized=True)
own = np.ascontiguousarray(np.broadcast_to(synth.PATTERN6.astype(np.float32), (B, 6, 3)))


def wall(fn, reps=12):
    fn()
    fn()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        ts.append((time.perf_counter() - t0) * 1e3)
    ts.sort()
    return "best %.3f median %.3f ms" % (ts[0], ts[len(ts) // 2])


print("lib", _lib.LIB_PATH.split("/")[-1], "fresh results:", wall(lambda: pnp.solve_pnp_batch(uv, own, K, want_rpy=False)), flush=True)
held = pnp.solve_pnp_batch(uv, own, K, want_rpy=False)
print("   out= reused:", wall(lambda: pnp.solve_pnp_batch(uv, own, K, want_rpy=False, out=held)), flush=True)
lib = _lib.load()
R, t = np.zeros((B, 9), np.float32), np.zeros((B, 3), np.float32)
res, it = np.zeros(B, np.float32), np.zeros(B, np.int32)
Kc = _lib.k_array(K)
print("   C ABI:", wall(lambda: _lib.check(lib.pnp_qeif_batch_f32(uv.ctypes.data, own.ctypes.data, 18, Kc, B, 6, None, R.ctypes.data,
                                                                  t.ctypes.data, res.ctypes.data, it.ctypes.data, None, 0, None))),
      "lanes", lib.pnp_host_lane_threads(0), flush=True)
