#!/usr/bin/env python3
"""Headline launches back to back on one stream against the same launches alternating between
two streams (independent batches: the tail of one launch overlaps the ramp of the next)."""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
import pnp_solver_test_amd as pnp
from pnp_solver_test_amd import synth
dev = torch.device("cuda:0"); K = synth.K_DEFAULT; B = 1 << 20
uv_h, _ = synth.stress_batch(B, seed=42, pattern=synth.PATTERN6, K=K, is_quantized=True)
uv = torch.from_numpy(uv_h).to(dev)
xyz = torch.from_numpy(synth.PATTERN6.astype(np.float32)).to(dev).expand(B, 6, 3).contiguous()
streams = [torch.cuda.Stream(dev) for _ in range(4)]
def run(n_streams, reps=200):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(reps):
        with torch.cuda.stream(streams[i % n_streams]):
            pnp.solve_pnp_batch(uv, xyz, K, want_rpy=False)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3
for _ in range(2):
    run(1, 150)
for rnd in range(3):
    print("round", rnd, " ".join("%d stream(s): %.4f ms/step" % (n, run(n)) for n in (1, 2, 3, 4)), flush=True)
