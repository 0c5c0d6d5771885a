#!/usr/bin/env python3
"""Host-pointer calls of every size class against the same call on device pointers: the small
single-buffer path, the chunked path (8 MB and more), many points, several patterns, every solver,
and odd round / iteration counts."""
import sys
import numpy as np, torch
sys.path.insert(0, ".")
import pnp_solver_test_amd as pnp
from pnp_solver_test_amd import synth
dev = torch.device("cuda:0"); K = synth.K_DEFAULT
bad = 0


def same(a, b):
    return all(np.array_equal(a[k], b[k].cpu().numpy(), equal_nan=True) for k in a)


def check(tag, uv, xyz, **kw):
    global bad
    try:
        h = pnp.solve_pnp_batch(uv, xyz, K, **kw)
        kd = dict(kw)
        if "patterns" in kd:
            kd["patterns"] = torch.from_numpy(np.ascontiguousarray(kd["patterns"])).to(dev)
        d = pnp.solve_pnp_batch(torch.from_numpy(uv).to(dev), None if xyz is None else torch.from_numpy(np.ascontiguousarray(xyz)).to(dev), K, **kd)
        ok = same(h, d)
    except Exception as e:  # noqa: BLE001
        print(tag, "RAISED", str(e)[:120]); bad += 1
        return
    print(tag, "identical" if ok else "DIFFERENT")
    bad += 0 if ok else 1


for N, B in ((6, 1), (6, 65), (6, 70000), (6, 300000), (68, 257), (68, 30000), (200, 9000)):
    P = synth.PATTERN6 if N == 6 else np.round(np.random.default_rng(N).uniform([-0.09, -0.06, -0.05], [0.09, 0.10, 0.03], (N, 3)) * 4096.0) / 4096.0
    uv, _ = synth.stress_batch(B, seed=N + B, pattern=P, K=K, is_quantized=True, chunk=1 << 15)
    Pf = P.astype(np.float32)
    own = np.broadcast_to(Pf, (B, N, 3)).copy()
    mb = B * N * 20 / 1e6
    for method in ("qeif", "dlt", "lm12"):
        check("N %3d B %6d (%5.1f MB) %-5s shared" % (N, B, mb, method), uv, Pf, method=method)
        check("N %3d B %6d (%5.1f MB) %-5s own   " % (N, B, mb, method), uv, own, method=method)
    check("N %3d B %6d two patterns      " % (N, B), uv, None, patterns=np.stack([Pf * np.float32(1.2), Pf]))
    check("N %3d B %6d float64 own qeif  " % (N, B), uv.astype(np.float64), own.astype(np.float64))
    check("N %3d B %6d dlt, 1 round      " % (N, B), uv, Pf, method="dlt", num_it=1)
    check("N %3d B %6d dlt, 7 rounds     " % (N, B), uv, own, method="dlt", num_it=7)
    check("N %3d B %6d lm12, 1 iteration " % (N, B), uv, Pf, method="lm12", num_it=1)
    check("N %3d B %6d qeif max_it 1     " % (N, B), uv, own, params=pnp.make_params(max_it=1))
print("problems:", bad)
