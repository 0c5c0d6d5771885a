#!/usr/bin/env python3
"""A handful of launches of each headline kernel on resident inputs, for counter collection
under rocprofv3 (--pmc passes read few dispatches, so this stays short)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pnp_solver_test_amd as pnp  # noqa: E402
from pnp_solver_test_amd import synth  # noqa: E402

B = 1 << 20
dev = torch.device("cuda:0")
uv_h, _ = synth.stress_batch(B, seed=42)
uv = torch.from_numpy(uv_h).to(dev)
xyz = torch.from_numpy(np.ascontiguousarray(np.broadcast_to(synth.PATTERN6.astype(np.float32), (B, 6, 3)))).to(dev)
P6 = torch.from_numpy(synth.PATTERN6.astype(np.float32)).to(dev)
B68 = 1 << 18
P68 = synth.face68()
uv68 = torch.from_numpy(synth.stress_batch(B68, seed=46, pattern=P68, chunk=1 << 15)[0]).to(dev)
P68d = torch.from_numpy(P68.astype(np.float32)).to(dev)
for _ in range(3):
    pnp.solve_pnp_batch(uv, xyz, synth.K_DEFAULT, method="qeif", want_rpy=False)
for _ in range(3):
    pnp.solve_pnp_batch(uv, xyz, synth.K_DEFAULT, method="dlt", want_rpy=False)
for _ in range(3):
    pnp.solve_pnp_batch(uv, P6, synth.K_DEFAULT, method="qeif", want_rpy=False)
for _ in range(3):
    pnp.solve_pnp_batch(uv68, P68d, synth.K_DEFAULT, method="qeif", want_rpy=False)      # moment form, shared pattern
own68 = P68d[None].expand(B68, 68, 3).contiguous()
for _ in range(3):
    pnp.solve_pnp_batch(uv68, own68, synth.K_DEFAULT, method="qeif", want_rpy=False)     # moment form, own patterns
torch.cuda.synchronize()
print("probe done")
