import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import pnp_solver_test_amd as pnp
from pnp_solver_test_amd import synth
dev = torch.device("cuda:0"); K = synth.K_DEFAULT; B = 1 << 20
uv_h, _ = synth.stress_batch(B, seed=42, pattern=synth.PATTERN6, K=K, is_quantized=True)
P6 = torch.from_numpy(synth.PATTERN6.astype(np.float32)).to(dev)
uv = torch.from_numpy(uv_h).to(dev)
for k in (3, 8):
    out = pnp.solve_pnp_batch(uv, P6, K, method="dlt", num_it=k, want_rpy=False)
    m = out["res_norm"] == -1.0
    print("rounds", k, "marked", int(m.sum()), "waves", int(m.view(-1, 64).any(dim=1).sum()), "of", B // 64,
          "finite among marked", int(torch.isfinite(out["R"][m]).all(dim=(1, 2)).sum()))
    idx = torch.nonzero(m)[:5, 0].cpu().numpy()
    print("   first marked", idx, uv_h[idx[:2]].tolist() if len(idx) else "")
