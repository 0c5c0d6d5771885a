#!/usr/bin/env python3
"""Headline QEIF launch with per-problem patterns against the same problems with one shared
pattern, interleaved, each after its own spin-up."""
import sys
import numpy as np, torch
sys.path.insert(0, ".")
import pnp_solver_test_amd as pnp
from pnp_solver_test_amd import synth
dev = torch.device("cuda:0"); K = synth.K_DEFAULT; B = 1 << 20
uv_h, _ = synth.stress_batch(B, seed=42, pattern=synth.PATTERN6, K=K, is_quantized=True)
uv = torch.from_numpy(uv_h).to(dev)
P6 = torch.from_numpy(synth.PATTERN6.astype(np.float32)).to(dev)
xyz = P6.expand(B, 6, 3).contiguous()
def timed(fn, reps=100, warm=100):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps
for rnd in range(3):
    for tag, pat in (("own", xyz), ("shared", P6)):
        for rpy in (False, True):
            print(rnd, tag, "rpy" if rpy else "   ", round(timed(lambda: pnp.solve_pnp_batch(uv, pat, K, want_rpy=rpy)), 4), "ms", flush=True)
