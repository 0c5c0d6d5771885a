#!/usr/bin/env python3
"""DLT on nearly planar patterns: six points in a plane with a relief of r times the pattern's
extent, r from 1e-9 to 1e-3.  Per relief level: the kernel (float64 arrays) against the NumPy
oracle as the reference calls pinv (cut-off 1e-15), and against the oracle with the direction
of the relief cut off (rcond 1e-5), with the conditioning of the systems."""
import sys

import numpy as np

sys.path.insert(0, ".")
import pnp_solver_test_amd as pnp  # noqa: E402
from oracle import pnp_oracle as po  # noqa: E402
from pnp_solver_test_amd import synth  # noqa: E402

K = synth.K_DEFAULT
base = np.array([[32, 0], [-32, 0], [27, 70], [-27, 70], [-5, 45], [0, 120]], dtype=np.float64) / 1024.0
shape = np.array([0.0, 0.45, -0.7, 1.0, -1.0, 0.3])
extent = 0.12
n = 48
for family, r in [(f, r) for f in ("z=0", "tilted") for r in 10.0 ** np.arange(-17.0, -2.9, 1.0)]:
    plane = 0.0 if family == "z=0" else base[:, 0] / 4 + base[:, 1] / 2
    P = np.column_stack([base, plane + r * extent * shape])
    uv, _ = synth.stress_batch(n, seed=77, pattern=P, K=K, is_quantized=True)
    uv = uv.astype(np.float64)
    out = pnp.solve_pnp_batch(uv, P, K, method="dlt", dtype=np.float64)
    full = [po.dlt_solve(uv[i], P, K) for i in range(n)]
    cut = [po.dlt_solve(uv[i], P, K, rcond=1e-5) for i in range(n)]
    fin = np.array([np.isfinite(f["R"]).all() and np.isfinite(c["R"]).all() and np.isfinite(out["R"][i]).all()
                    for i, (f, c) in enumerate(zip(full, cut))])
    d_full = np.array([np.abs(out["R"][i] - full[i]["R"]).max() for i in range(n)])[fin]
    d_cut = np.array([np.abs(out["R"][i] - cut[i]["R"]).max() for i in range(n)])[fin]
    svr = np.array([f["sv_ratio"] for f in full])[fin]
    print(family, "relief %.1e  finite %2d  sv ratio %.1e..%.1e  |dR| vs pinv(1e-15): med %.1e max %.1e   vs direction cut: med %.1e max %.1e"
          % (r, fin.sum(), svr.min(), svr.max(), np.median(d_full), d_full.max(), np.median(d_cut), d_cut.max()), flush=True)
