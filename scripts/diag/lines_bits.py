#!/usr/bin/env python3
"""Dumps the N = 32 and N = 68 per-problem-pattern QEIF results (float32 and float64 arrays, odd
batch sizes, a misaligned view) of the library PNP_HIP_LIB selects to argv[1]; with two files as
arguments instead, compares them bit for bit."""
import sys

import numpy as np

if len(sys.argv) == 3:
    a, b = np.load(sys.argv[1]), np.load(sys.argv[2])
    print("kernels:", sorted(set(a["kernels"].tolist())), "vs", sorted(set(b["kernels"].tolist())))
    bad = [k for k in a.files if k != "kernels" and not np.array_equal(a[k], b[k], equal_nan=True)]
    print("arrays compared:", len(a.files) - 1, "different:", bad)
    sys.exit(1 if bad else 0)

import torch

sys.path.insert(0, ".")
import pnp_solver_test_amd as pnp  # noqa: E402
from pnp_solver_test_amd import _lib, synth  # noqa: E402

_lib.load().pnp_set_qeif_form(_lib.QEIF_FORM_MOMENTS)
rng = np.random.default_rng(11)
out, kernels = {}, []
for N, B in ((32, 1500), (68, 1100), (68, 4099), (130, 700), (24, 513), (33, 900), (69, 777), (512, 300)):
    P = rng.uniform([-0.07, -0.05, -0.04], [0.07, 0.12, 0.02], (N, 3))
    uv, _ = synth.stress_batch(B, seed=N, pattern=P, K=synth.K_DEFAULT, is_quantized=True)
    own = np.ascontiguousarray(np.broadcast_to(P.astype(np.float32), (B, N, 3)))
    for tag, dt in (("f32", torch.float32), ("f64", torch.float64)):
        for off in (0, 1):                      # a view that starts one problem into the buffer
            u = torch.from_numpy(uv).cuda().to(dt)[off:].contiguous() if off == 0 else torch.from_numpy(uv).cuda().to(dt)[off:]
            x = torch.from_numpy(own).cuda().to(dt)[off:]
            o = pnp.solve_pnp_batch(u, x, synth.K_DEFAULT)
            kernels.append(_lib.last_launch()["kernel"])
            for k in ("R", "t", "iters", "res_norm"):
                out["%s_%d_%d_%s_%d" % (k, N, B, tag, off)] = o[k].cpu().numpy()
np.savez(sys.argv[1], kernels=np.array(kernels), **out)
print("wrote", sys.argv[1], sorted(set(kernels)))
