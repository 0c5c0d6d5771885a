#!/usr/bin/env python3
"""Host-pointer calls (NumPy arrays, a pattern per problem) around the chunking threshold:
wall-clock per call against the batch size."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
import pnp_solver_test_amd as pnp
from pnp_solver_test_amd import _lib, synth
K = synth.K_DEFAULT
lib = _lib.load()
Bmax = 1 << 18
uv, _ = synth.stress_batch(Bmax, seed=42, pattern=synth.PATTERN6, K=K, is_quantized=True)
own = np.ascontiguousarray(np.broadcast_to(synth.PATTERN6.astype(np.float32), (Bmax, 6, 3)))
R, t = np.zeros((Bmax, 9), np.float32), np.zeros((Bmax, 3), np.float32)
Kc = _lib.k_array(K)
for B in (1 << 12, 1 << 13, 1 << 14, 1 << 15, 1 << 16, 3 << 15, 1 << 17, 1 << 18):
    fn = lambda: _lib.check(lib.pnp_qeif_batch_f32(uv.ctypes.data, own.ctypes.data, 18, Kc, B, 6, None, R.ctypes.data, t.ctypes.data, None, None, None, 0, None))
    for _ in range(3): fn()
    t0 = time.perf_counter()
    for _ in range(20): fn()
    dt = (time.perf_counter() - t0) / 20
    print("B %7d  %6.1f MB moved  %8.1f us per call  %6.1f M solves/s" % (B, B * 168 / 1e6, dt * 1e6, B / dt / 1e6), flush=True)
