#!/usr/bin/env python3
"""Splits the moment-form kernel's time at N = 68 into its accumulation pass and its iterations
by capping the iteration count (kernel time by HIP events)."""
import sys
import numpy as np, torch
sys.path.insert(0, ".")
import pnp_solver_test_amd as pnp
from pnp_solver_test_amd import _lib, synth
dev = torch.device("cuda:0"); K = synth.K_DEFAULT; B = 1 << 18; N = 68
P = synth.face68()
uv_h, _ = synth.stress_batch(B, seed=46, pattern=P, K=K, is_quantized=True, chunk=1 << 15)
uv = torch.from_numpy(uv_h).to(dev)
Pd = torch.from_numpy(P.astype(np.float32)).to(dev)
own = Pd[None].expand(B, N, 3).contiguous()
def timed(fn, reps=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps
for tag, xyz in (("shared", Pd), ("own", own)):
    for cap in (1, 14):
        p = pnp.make_params(max_it=cap, rel_tol=0.0 if cap < 14 else 1e-2)
        ms = timed(lambda: pnp.solve_pnp_batch(uv, xyz, K, params=p, want_rpy=False))
        print(tag, "max_it", cap, round(ms, 4), "ms", flush=True)
