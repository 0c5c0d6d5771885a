#!/usr/bin/env python3
"""Accumulation pass of the N = 68 per-problem-pattern moment kernel (iterations capped at 1)
against the batch size: 64k problems are one workgroup per CU, 128k two (one full generation)."""
import sys
import numpy as np, torch
sys.path.insert(0, ".")
import pnp_solver_test_amd as pnp
from pnp_solver_test_amd import synth
dev = torch.device("cuda:0"); K = synth.K_DEFAULT; N = 68
P = synth.face68()
Bmax = 1 << 19
uv_h, _ = synth.stress_batch(Bmax, seed=46, pattern=P, K=K, is_quantized=True, chunk=1 << 15)
uv = torch.from_numpy(uv_h).to(dev)
own = torch.from_numpy(P.astype(np.float32)).to(dev)[None].expand(Bmax, N, 3).contiguous()
def timed(fn, reps=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps
for cap in (1, 14):
    p = pnp.make_params(max_it=cap, rel_tol=0.0 if cap < 14 else 1e-2)
    for B in (1 << 14, 1 << 15, 1 << 16, 1 << 17, 3 << 16, 1 << 18, 1 << 19):
        ms = timed(lambda: pnp.solve_pnp_batch(uv[:B], own[:B], K, params=p, want_rpy=False))
        print("max_it", cap, "B", B, "workgroups/CU %.2f" % (B / 256 / 256), round(ms, 4), "ms", "%.2f TB/s" % (B * N * 20 / ms / 1e9), flush=True)
