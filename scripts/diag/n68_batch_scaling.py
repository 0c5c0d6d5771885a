#!/usr/bin/env python3
"""N = 68 in moment form against the batch size (a pattern per problem and a shared one): time per
256k problems.  A launch of 256k problems is two generations of workgroups, whose accumulation
passes (memory) and iterations (arithmetic) run in step; longer launches let them drift apart."""
import sys
import numpy as np, torch
sys.path.insert(0, ".")
import pnp_solver_test_amd as pnp
from pnp_solver_test_amd import _lib, synth
dev = torch.device("cuda:0"); K = synth.K_DEFAULT
N = 68
P = np.random.default_rng(0).uniform([-0.07, -0.05, -0.04], [0.07, 0.12, 0.02], (N, 3))
uv_h, _ = synth.stress_batch(1 << 18, seed=46, pattern=P, K=K, is_quantized=True, chunk=1 << 15)
for shared in (False, True):
    for log2b in (16, 17, 18, 19, 20, 21):
        B = 1 << log2b
        reps = max(1, B >> 18)
        uv = torch.from_numpy(uv_h[:min(B, 1 << 18)]).to(dev).repeat(reps, 1, 1)[:B].contiguous()
        Pd = torch.from_numpy(P.astype(np.float32)).to(dev)
        xyz = Pd if shared else Pd[None].expand(B, N, 3).contiguous()
        fn = lambda: pnp.solve_pnp_batch(uv, xyz, K, want_rpy=False)
        for _ in range(8): fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = max(10, (1 << 24) // B)
        a.record()
        for _ in range(n): fn()
        b.record(); torch.cuda.synchronize()
        ms = a.elapsed_time(b) / n
        print("shared" if shared else "own", "B 2^%d" % log2b, "%.4f ms" % ms, "per 256k: %.4f ms" % (ms * (1 << 18) / B),
              _lib.last_launch()["kernel"].split("::")[1], flush=True)
        del uv, xyz
