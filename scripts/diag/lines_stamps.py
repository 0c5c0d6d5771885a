#!/usr/bin/env python3
"""Phase cycle sums of the line-cut moment kernel's accumulation pass (debug build that writes them
where the rotation goes): per wavefront, mean over the launch."""
import sys
import numpy as np, torch
sys.path.insert(0, ".")
import pnp_solver_test_amd as pnp
from pnp_solver_test_amd import synth
dev = torch.device("cuda:0"); K = synth.K_DEFAULT; B = 1 << 18; N = 68
P = synth.face68()
uv_h, _ = synth.stress_batch(B, seed=46, pattern=P, K=K, is_quantized=True, chunk=1 << 15)
uv = torch.from_numpy(uv_h).to(dev)
own = torch.from_numpy(P.astype(np.float32)).to(dev)[None].expand(B, N, 3).contiguous()
for _ in range(5):
    o = pnp.solve_pnp_batch(uv, own, K, want_rpy=False)
torch.cuda.synchronize()
R = o["R"].cpu().numpy().reshape(B, 9).astype(np.float64)
names = ["wait at top barrier", "wait for loads", "write-out to LDS", "second barrier", "issue of next loads", "consume"]
w = R[::64]                      # lane 0 of every wavefront
for k, n in enumerate(names):
    print("%-22s mean %8.0f  p10 %8.0f  p90 %8.0f cycles" % (n, w[:, k].mean(), np.percentile(w[:, k], 10), np.percentile(w[:, k], 90)))
print("pass total mean %.0f cycles; chunks %s" % (w[:, 6].mean(), np.unique(w[:, 8])))
