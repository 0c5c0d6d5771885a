#!/usr/bin/env python3
"""End-to-end latency of one drop-in solve_pnp call (dict in, 7-tuple out; a batch of one through
the host-pointer path) and of small host-pointer batches."""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import pnp_solver_test_amd as pnp
from pnp_solver_test_amd import synth

K = synth.K_DEFAULT
pattern = {k: list(v) for k, v in synth.get_golden_pattern("Alexander").items()}
s = pnp.PNP_SOLVER(K, [pattern])
gt = pnp.PNP_SOLVER(K, [pattern])
R = gt.get_rotation_matrix_from_Euler(15.0, 45.0, -15.0, is_degree=True)
img = gt.perspective_projection_golden_landmarks(R, np.array([[0.35], [0.35], [0.5]]), is_quantized=True)
for _ in range(200):
    s.solve_pnp(img)
t0 = time.perf_counter()
n = 3000
for _ in range(n):
    s.solve_pnp(img)
print("solve_pnp (dict API, batch of one): %.1f us per call" % ((time.perf_counter() - t0) / n * 1e6))
uv6, _ = synth.stress_batch(4096, seed=1)
P6 = synth.PATTERN6.astype(np.float32)
for B in (1, 64, 1024, 4096):
    for _ in range(50):
        pnp.solve_pnp_batch(uv6[:B], P6, K)
    t0 = time.perf_counter()
    for _ in range(500):
        pnp.solve_pnp_batch(uv6[:B], P6, K)
    print("solve_pnp_batch host pointers B=%d: %.1f us per call" % (B, (time.perf_counter() - t0) / 500 * 1e6))
