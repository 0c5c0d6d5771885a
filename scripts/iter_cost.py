#!/usr/bin/env python3
"""Kernel time of the headline QEIF launch against a forced iteration count (stopping rule
off): the intercept is the launch's fixed cost, the slope the cost of one iteration."""
import sys
import numpy as np, torch
sys.path.insert(0, ".")
import pnp_solver_test_amd as pnp
from pnp_solver_test_amd import _lib, synth
dev = torch.device("cuda:0"); lib = _lib.load(); K = synth.K_DEFAULT; B = 1 << 20
uv_h, _ = synth.stress_batch(B, seed=42, pattern=synth.PATTERN6, K=K, is_quantized=True)
uv = torch.from_numpy(uv_h).to(dev)
xyz = torch.from_numpy(np.ascontiguousarray(np.broadcast_to(synth.PATTERN6.astype(np.float32), (B, 6, 3)))).to(dev)
def timed(reps=100, **kw):
    for _ in range(30): pnp.solve_pnp_batch(uv, xyz, K, want_rpy=False, **kw)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): pnp.solve_pnp_batch(uv, xyz, K, want_rpy=False, **kw)
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps
for name in ("qeif_kernel<1,6>",):
    xs, ys = [], []
    for k in (1, 2, 3, 4, 5, 6, 7, 8, 10, 14):
        ms = timed(params=pnp.make_params(max_it=k, rel_tol=0.0))
        xs.append(k); ys.append(ms)
        print(name, "max_it", k, round(ms, 4), "ms", flush=True)
    slope, icpt = np.polyfit(xs, ys, 1)
    print(name, "fit: %.4f ms fixed + %.4f ms per iteration; default rule: %.4f ms" % (icpt, slope, timed()))
