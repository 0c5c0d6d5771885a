"""How often the DLT launch of the bench workload leaves the common path, and what that costs:
the stress batch against a batch made of one healthy problem repeated."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import pnp_solver_test_amd as pnp  # noqa: E402
from pnp_solver_test_amd import synth  # noqa: E402

dev = torch.device("cuda:0")
K = synth.K_DEFAULT
B = 1 << 20
uv_h, _ = synth.stress_batch(B, seed=42, pattern=synth.PATTERN6, K=K, is_quantized=True)
P6 = torch.from_numpy(synth.PATTERN6.astype(np.float32)).to(dev)


def timed(uv, reps=100):
    for _ in range(30):
        pnp.solve_pnp_batch(uv, P6, K, method="dlt", want_rpy=False)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        pnp.solve_pnp_batch(uv, P6, K, method="dlt", want_rpy=False)
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


uv = torch.from_numpy(uv_h).to(dev)
out = pnp.solve_pnp_batch(uv, P6, K, method="dlt", want_rpy=False)
nan = ~torch.isfinite(out["R"]).all(dim=(1, 2))
print("stress batch: %.4f ms, non-finite problems %d of %d, waves with one %d of %d"
      % (timed(uv), int(nan.sum()), B, int(nan.view(-1, 64).any(dim=1).sum()), B // 64))
good = uv[~nan]
print("finite problems only (%d): %.4f ms" % (len(good), timed(good[: (len(good) // 64) * 64])))
one = uv[0:1].expand(B, 6, 2).contiguous()
print("one problem repeated: %.4f ms" % timed(one))
