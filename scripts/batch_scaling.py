#!/usr/bin/env python3
"""Throughput against the number of problems per launch (kernel time by HIP events).
usage: batch_scaling.py [qeif|dlt]"""
import sys
import numpy as np, torch
sys.path.insert(0, ".")
import pnp_solver_test_amd as pnp
from pnp_solver_test_amd import synth
method = sys.argv[1] if len(sys.argv) > 1 else "qeif"
dev = torch.device("cuda:0"); K = synth.K_DEFAULT
base, _ = synth.stress_batch(1 << 20, seed=42, pattern=synth.PATTERN6, K=K, is_quantized=True)
P = synth.PATTERN6.astype(np.float32)
for logB in (14, 16, 18, 20, 22, 23):
    B = 1 << logB
    uv_h = np.tile(base, (max(1, B >> 20), 1, 1))[:B]
    uv = torch.from_numpy(uv_h).to(dev)
    xyz = torch.from_numpy(np.ascontiguousarray(np.broadcast_to(P, (B, 6, 3)))).to(dev)
    reps = max(10, min(200, (1 << 26) // B))
    for _ in range(max(10, reps // 2)): pnp.solve_pnp_batch(uv, xyz, K, method=method, want_rpy=False)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): pnp.solve_pnp_batch(uv, xyz, K, method=method, want_rpy=False)
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / reps
    print(method, "B = 2^%d: %.4f ms per launch, %.2f G solves/s" % (logB, ms, B / ms / 1e6), flush=True)
    del uv, xyz
