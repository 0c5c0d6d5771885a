#!/usr/bin/env python3
"""Where the moment form of QEIF overtakes the point form: sweep over N at a large batch and
over the batch size at N = 68 (kernel time by HIP events, per-problem and shared patterns)."""
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


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


out = {}
cases = [(n, 1 << 19) for n in (6, 8, 10, 12, 14, 16)] + [(68, b) for b in (2048, 8192, 32768, 131072)]
for N, B in cases:
    P = synth.face68()[:N]
    uv_h, _ = synth.stress_batch(B, seed=46, pattern=P, K=K, is_quantized=True, chunk=1 << 15)
    uv = torch.from_numpy(uv_h).to(dev)
    Pd = torch.from_numpy(P.astype(np.float32)).to(dev)
    own = Pd[None].expand(B, N, 3).contiguous()
    row = {}
    for name, form in (("points", _lib.QEIF_FORM_POINTS), ("moments", _lib.QEIF_FORM_MOMENTS)):
        lib.pnp_set_qeif_form(form)
        row[name + "_shared_ms"] = round(timed(lambda: pnp.solve_pnp_batch(uv, Pd, K, want_rpy=False)), 4)
        row[name + "_own_ms"] = round(timed(lambda: pnp.solve_pnp_batch(uv, own, K, want_rpy=False)), 4)
    lib.pnp_set_qeif_form(_lib.QEIF_FORM_AUTO)
    out["N%d_B%d" % (N, B)] = row
    print("N%d_B%d" % (N, B), row, flush=True)
json.dump(out, open("gpurun_out/form_crossover.json", "w"), indent=1)
