#!/usr/bin/env python3
"""Kernel time of the DLT launch (N = 6, 1M problems) against its number of rounds: intercept =
fixed cost of a launch (loads, stores, residual, ramp), slope = one round."""
import sys
import numpy as np, torch
sys.path.insert(0, ".")
import pnp_solver_test_amd as pnp
from pnp_solver_test_amd import synth
dev = torch.device("cuda:0"); K = synth.K_DEFAULT; B = 1 << 20
uv_h, _ = synth.stress_batch(B, seed=42, pattern=synth.PATTERN6, K=K, is_quantized=True)
uv = torch.from_numpy(uv_h).to(dev)
xyz = torch.from_numpy(np.ascontiguousarray(np.broadcast_to(synth.PATTERN6.astype(np.float32), (B, 6, 3)))).to(dev)
P6 = torch.from_numpy(synth.PATTERN6.astype(np.float32)).to(dev)
def timed(x, k, reps=100):
    for _ in range(30): pnp.solve_pnp_batch(uv, x, K, method="dlt", num_it=k, want_rpy=False)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): pnp.solve_pnp_batch(uv, x, K, method="dlt", num_it=k, want_rpy=False)
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps
for tag, x in (("own", xyz), ("shared", P6)):
    xs, ys = [], []
    for k in (1, 2, 3, 4, 6, 8):
        ms = timed(x, k); xs.append(k); ys.append(ms)
        print(tag, "num_it", k, round(ms, 4), "ms", flush=True)
    slope, icpt = np.polyfit(xs, ys, 1)
    print(tag, "fit: %.4f ms fixed + %.4f ms per round" % (icpt, slope))
