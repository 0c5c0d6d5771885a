#!/usr/bin/env python3
"""Every solver at point counts up to the library's maximum, both element types, shared and
per-problem patterns, and several patterns: does the call go through, and does it agree with
the C oracle (QEIF, DLT) on the same inputs?"""
import sys
import numpy as np, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import pnp_solver_test_amd as pnp
from pnp_solver_test_amd import synth
from oracle import c_oracle
dev = torch.device("cuda:0"); K = synth.K_DEFAULT
rng = np.random.default_rng(11)
B = 300
bad = 0
for N in (68, 130, 200, 512):
    P = rng.uniform([-0.09, -0.06, -0.05], [0.09, 0.10, 0.03], (N, 3))
    P = np.round(P * 4096.0) / 4096.0
    uv, _ = synth.stress_batch(B, seed=N, pattern=P, K=K, is_quantized=True)
    ref_q = c_oracle.qeif_batch(uv.astype(np.float64), P, K)
    ref_d = c_oracle.dlt_batch(uv.astype(np.float64), P, K)
    for dt in (np.float32, np.float64):
        for own in (False, True):
            xyz = np.broadcast_to(P, (B, N, 3)).copy() if own else P
            for method in ("qeif", "dlt", "lm12"):
                tag = "N %3d %s %-6s %-5s" % (N, dt.__name__, "own" if own else "shared", method)
                try:
                    out = pnp.solve_pnp_batch(torch.from_numpy(uv.astype(dt)).to(dev), torch.from_numpy(xyz.astype(dt)).to(dev), K, method=method)
                    torch.cuda.synchronize()
                except Exception as e:  # noqa: BLE001
                    print(tag, "REFUSED:", str(e)[:110]); bad += 1
                    continue
                R = out["R"].cpu().numpy().astype(np.float64)
                fin = np.isfinite(R).all(axis=(1, 2)).mean()
                msg = "finite %.3f" % fin
                if method != "lm12":
                    ref = ref_q if method == "qeif" else ref_d
                    d = np.abs(R - ref["R"]).max(axis=(1, 2))
                    msg += "  median |dR| %.1e  p99 %.1e" % (np.nanmedian(d), np.nanpercentile(d, 99))
                    if method == "qeif":
                        msg += "  iters equal %s" % np.array_equal(out["iters"].cpu().numpy(), ref["iters"])
                print(tag, msg)
    try:
        pats = np.stack([P, P * 1.1])
        out = pnp.solve_pnp_batch(torch.from_numpy(uv).to(dev), None, K, patterns=torch.from_numpy(pats.astype(np.float32)).to(dev))
        print("N %3d two patterns: pattern 0 chosen for %.3f" % (N, (out["pattern_id"] == 0).float().mean().item()))
    except Exception as e:  # noqa: BLE001
        print("N %3d two patterns REFUSED: %s" % (N, str(e)[:110])); bad += 1
print("refused:", bad)
