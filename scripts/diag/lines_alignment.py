#!/usr/bin/env python3
"""Line-cut moment kernel, a pattern per problem: time against N around row sizes that are whole
cache lines (N = 64, 96: no line is shared between rows or chunks) and ones that are not."""
import sys
import numpy as np, torch
sys.path.insert(0, ".")
import pnp_solver_test_amd as pnp
from pnp_solver_test_amd import _lib, synth
dev = torch.device("cuda:0"); K = synth.K_DEFAULT
lib = _lib.load(); lib.pnp_set_qeif_form(_lib.QEIF_FORM_MOMENTS)
rng = np.random.default_rng(3)
B = 1 << 18
Pmax = rng.uniform([-0.07, -0.05, -0.04], [0.07, 0.12, 0.02], (128, 3))
for N in (48, 56, 60, 62, 64, 66, 68, 72, 80, 92, 96, 100):
    P = Pmax[:N]
    uv_h, _ = synth.stress_batch(B, seed=46, pattern=P, K=K, is_quantized=True, chunk=1 << 15)
    uv = torch.from_numpy(uv_h).to(dev)
    own = torch.from_numpy(P.astype(np.float32)).to(dev)[None].expand(B, N, 3).contiguous()
    fn = lambda: pnp.solve_pnp_batch(uv, own, K, want_rpy=False)
    for _ in range(300): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(50): o = fn()
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 50
    print("N %3d  %.4f ms  %.2f us per point  mean iterations %.2f  %.0f GB/s" % (
        N, ms, ms * 1e3 / N, float(o["iters"].float().mean()), B * (20 * N + 48) / ms / 1e6), flush=True)
