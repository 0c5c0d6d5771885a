"""The 1M-problem DLT batch against the C oracle with no mask: the finite mismatches."""
import os
import sys
from concurrent.futures import ThreadPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import pnp_solver_test_amd as pnp  # noqa: E402
from oracle import c_oracle, pnp_oracle as po  # noqa: E402
from pnp_solver_test_amd import synth  # noqa: E402

np.set_printoptions(precision=5, suppress=True, linewidth=160)
dev = torch.device("cuda:0")
K = synth.K_DEFAULT
B = 1 << 20
uv, _ = synth.stress_batch(B, seed=42, pattern=synth.PATTERN6, K=K, is_quantized=True, chunk=1 << 15)
for dt in (np.float32, np.float64):
    P = synth.PATTERN6.astype(dt)
    u64, P64 = uv.astype(np.float64), P.astype(np.float64)
    bounds = [(i * B // 16, (i + 1) * B // 16) for i in range(16)]
    with ThreadPoolExecutor(16) as ex:
        parts = list(ex.map(lambda b: c_oracle.dlt_batch(u64[b[0]:b[1]], P64, K), bounds))
    ref = {k: np.concatenate([p[k] for p in parts]) for k in parts[0]}
    out = pnp.solve_pnp_batch(torch.from_numpy(uv.astype(dt)).to(dev), torch.from_numpy(P).to(dev), K, method="dlt")
    R = out["R"].cpu().numpy().astype(np.float64)
    fin = np.isfinite(R).all(axis=(1, 2)) & np.isfinite(ref["R"]).all(axis=(1, 2))
    d = np.abs(R - ref["R"]).max(axis=(1, 2))
    bad = np.flatnonzero(fin & (d > 1e-6))
    print(np.dtype(dt).name, "finite both", fin.sum(), "bad", len(bad), "lone nan", (~fin).sum())
    for i in bad[:12]:
        m = ref["margins"][i]
        q = po.dlt_solve(u64[i], P64, K)
        print("  i", i, "diff", d[i], "margins k2 %.3e sim %.3e delta %.3e" % tuple(m), "numpy-oracle vs C-oracle", np.abs(q["R"] - ref["R"][i]).max(),
              "vs gpu", np.abs(q["R"] - R[i]).max())
