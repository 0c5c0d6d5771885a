#!/usr/bin/env python3
"""Wall-clock per step of the headline launch loop with an event pair around every launch
against the same loop with one event pair around all of it."""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
import pnp_solver_test_amd as pnp
from pnp_solver_test_amd import synth
dev = torch.device("cuda:0"); K = synth.K_DEFAULT; B = 1 << 20
uv_h, _ = synth.stress_batch(B, seed=42, pattern=synth.PATTERN6, K=K, is_quantized=True)
uv = torch.from_numpy(uv_h).to(dev)
xyz = torch.from_numpy(synth.PATTERN6.astype(np.float32)).to(dev).expand(B, 6, 3).contiguous()
step = lambda: pnp.solve_pnp_batch(uv, xyz, K, want_rpy=False)
for _ in range(150): step()
torch.cuda.synchronize()
K_ = 200
for rnd in range(3):
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(K_)]
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for a, b in ev:
        a.record(); step(); b.record()
    torch.cuda.synchronize(); wall_pairs = (time.perf_counter() - t0) / K_ * 1e3
    per = float(np.mean([a.elapsed_time(b) for a, b in ev]))
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    a.record()
    for _ in range(K_): step()
    b.record()
    torch.cuda.synchronize(); wall_one = (time.perf_counter() - t0) / K_ * 1e3
    print("per-launch pairs: wall %.4f ms/step, mean event pair %.4f ms | one pair: wall %.4f ms/step, events %.4f ms/step"
          % (wall_pairs, per, wall_one, a.elapsed_time(b) / K_), flush=True)
