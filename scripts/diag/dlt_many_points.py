#!/usr/bin/env python3
"""Iterated DLT and LM12 away from six points: launch time and bytes for N = 6, 15, 32, 68
(per-problem and shared patterns), as a check that no path falls off a cliff."""
import sys
import numpy as np, torch
sys.path.insert(0, ".")
import pnp_solver_test_amd as pnp
from pnp_solver_test_amd import synth
dev = torch.device("cuda:0"); K = synth.K_DEFAULT


def timed(fn, reps=20):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


rng = np.random.default_rng(3)
for N, B in ((6, 1 << 20), (15, 1 << 19), (32, 1 << 18), (68, 1 << 18)):
    P = synth.face68()[:N] if N > 6 else synth.PATTERN6
    uv_h, _ = synth.stress_batch(B, seed=5, pattern=P, K=K, is_quantized=True, chunk=1 << 15)
    uv = torch.from_numpy(uv_h).to(dev)
    shared = torch.from_numpy(P.astype(np.float32)).to(dev)
    own = shared[None].expand(B, N, 3).contiguous()
    for method in ("dlt", "lm12", "qeif"):
        for name, xyz in (("own", own), ("shared", shared)):
            ms = timed(lambda: pnp.solve_pnp_batch(uv, xyz, K, method=method, want_rpy=False))
            byt = B * (N * (8 + (12 if name == "own" else 0)) + 52)
            print("N %2d B %7d %-5s %-6s %.4f ms  %6.2f G solves/s  %5.2f TB/s" % (N, B, method, name, ms, B / ms / 1e6, byt / ms / 1e9), flush=True)
