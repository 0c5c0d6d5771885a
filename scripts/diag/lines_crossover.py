#!/usr/bin/env python3
"""Moment form: kernel time against N (512k problems up to N = 32, 256k above), a pattern per
problem or (argument "shared") one for all; "f64" for float64 arrays.  Run once as is and once
with the A/B build and PNP_MOMENT_PLAIN=1 to compare the two staging schemes."""
import sys
import numpy as np, torch
sys.path.insert(0, ".")
import pnp_solver_test_amd as pnp
from pnp_solver_test_amd import _lib, synth
dev = torch.device("cuda:0"); K = synth.K_DEFAULT
F64 = "f64" in sys.argv[1:]
SHARED = "shared" in sys.argv[1:]
lib = _lib.load(); lib.pnp_set_qeif_form(_lib.QEIF_FORM_MOMENTS)
rng = np.random.default_rng(3)
for N in (12, 16, 17, 20, 24, 28, 32, 40, 48, 68, 100):
    B = 1 << 19 if N <= 32 else 1 << 18
    P = rng.uniform([-0.07, -0.05, -0.04], [0.07, 0.12, 0.02], (N, 3))
    uv_h, _ = synth.stress_batch(B, seed=46, pattern=P, K=K, is_quantized=True, chunk=1 << 15)
    if F64:
        B //= 2
    uv = torch.from_numpy(uv_h[:B]).to(dev)
    own = torch.from_numpy(P.astype(np.float32)).to(dev)[None].expand(B, N, 3).contiguous()
    if SHARED:
        own = own[0].contiguous()
    if F64:
        uv, own = uv.double(), own.double()
    fn = lambda: pnp.solve_pnp_batch(uv, own, K, want_rpy=False)
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20): fn()
    b.record(); torch.cuda.synchronize()
    print("N", N, "B", B, round(a.elapsed_time(b) / 20, 4), "ms", _lib.last_launch()["kernel"].split("::")[1][:40], flush=True)
