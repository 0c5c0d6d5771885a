#!/usr/bin/env python3
"""QEIF throughput at N = 15, 32, 68 in point form and in moment form (kernel time by HIP events)."""
import json
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
import pnp_solver_test_amd as pnp
from pnp_solver_test_amd import _lib, synth

dev = torch.device("cuda:0")
lib = _lib.load()
K = synth.K_DEFAULT
out = {}
for N, B in ((15, 1 << 19), (32, 1 << 18), (68, 1 << 18)):
    P = synth.face68()[:N]
    uv_h, _ = synth.stress_batch(B, seed=46, pattern=P, K=K, is_quantized=True, chunk=1 << 15)
    uv = torch.from_numpy(uv_h).to(dev)
    Pd = torch.from_numpy(P.astype(np.float32)).to(dev)
    own = Pd[None].expand(B, N, 3).contiguous()
    for name, form in (("points", _lib.QEIF_FORM_POINTS), ("moments", _lib.QEIF_FORM_MOMENTS)):
        lib.pnp_set_qeif_form(form)
        for tag, xyz in (("shared", Pd), ("per_problem", own)):
            for _ in range(3):
                o = pnp.solve_pnp_batch(uv, xyz, K, want_rpy=False)
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(20):
                o = pnp.solve_pnp_batch(uv, xyz, K, want_rpy=False)
            b.record()
            torch.cuda.synchronize()
            ms = a.elapsed_time(b) / 20
            out["n%d_%s_%s" % (N, tag, name)] = dict(kernel_ms=round(ms, 4), solves_per_s=round(B / ms * 1e3, 1),
                                                   mean_iters=round(float(o["iters"].float().mean()), 3))
    lib.pnp_set_qeif_form(_lib.QEIF_FORM_AUTO)
print(json.dumps(out, indent=1))
