#!/usr/bin/env python3
"""Timeline of the workgroups of one N = 68 launch (a pattern per problem): start, end of the
accumulation pass and end, from a debug build (-DPNP_STAMP_WG) that writes wall_clock64() stamps
(100 MHz) where the first problem's rotation goes.  PNP_HIP_LIB selects that build."""
import sys
import numpy as np, torch
sys.path.insert(0, ".")
import pnp_solver_test_amd as pnp
from pnp_solver_test_amd import synth
dev = torch.device("cuda:0"); K = synth.K_DEFAULT; N = 68
B = 1 << (int(sys.argv[1]) if len(sys.argv) > 1 else 18)
P = np.random.default_rng(0).uniform([-0.07, -0.05, -0.04], [0.07, 0.12, 0.02], (N, 3))
uv_h, _ = synth.stress_batch(1 << 18, seed=46, pattern=P, K=K, is_quantized=True, chunk=1 << 15)
uv = torch.from_numpy(uv_h).to(dev).repeat(max(1, B >> 18), 1, 1)[:B].contiguous()
own = torch.from_numpy(P.astype(np.float32)).to(dev)[None].expand(B, N, 3).contiguous()
for _ in range(400):
    o = pnp.solve_pnp_batch(uv, own, K, want_rpy=False)
torch.cuda.synchronize()
from pnp_solver_test_amd import _lib
block = _lib.last_launch()["block"]
raw = o["R"].cpu().numpy().reshape(B, 9)[::block].copy()          # first problem of every workgroup
words = raw.view(np.uint32).reshape(-1, 9)
st = words[:, :6].copy().view(np.uint64).reshape(-1, 3).astype(np.int64)
hw, xcc = words[:, 6].astype(np.int64), words[:, 7].astype(np.int64) & 15
t0 = st[:, 0].min()
us = (st - t0) / 100.0
start, mid, end = us[:, 0], us[:, 1], us[:, 2]
print("workgroups", len(us), "launch span %.1f us" % end.max())
order = np.argsort(start)
gen1 = start < 5.0
print("threads per workgroup", block)
print("started within 5 us: %d; their pass %.1f (p10 %.1f p90 %.1f), iterations %.1f, end %.1f (p10 %.1f p90 %.1f)" % (
    gen1.sum(), (mid - start)[gen1].mean(), np.percentile((mid - start)[gen1], 10), np.percentile((mid - start)[gen1], 90),
    (end - mid)[gen1].mean(), end[gen1].mean(), np.percentile(end[gen1], 10), np.percentile(end[gen1], 90)))
rest = ~gen1
if rest.any():
    print("the others: start %.1f (p10 %.1f p90 %.1f), pass %.1f (p10 %.1f p90 %.1f), iterations %.1f (p10 %.1f p90 %.1f), end %.1f (p10 %.1f p90 %.1f max %.1f)" % (
        start[rest].mean(), np.percentile(start[rest], 10), np.percentile(start[rest], 90),
        (mid - start)[rest].mean(), np.percentile((mid - start)[rest], 10), np.percentile((mid - start)[rest], 90),
        (end - mid)[rest].mean(), np.percentile((end - mid)[rest], 10), np.percentile((end - mid)[rest], 90),
        end[rest].mean(), np.percentile(end[rest], 10), np.percentile(end[rest], 90), end[rest].max()))
h, edges = np.histogram(start, bins=np.arange(0, end.max() + 5, 5))
print("starts per 5 us:", h.tolist())
h, _ = np.histogram(end, bins=np.arange(0, end.max() + 5, 5))
print("ends per 5 us:  ", h.tolist())

# per compute unit: (XCC, shader engine, array, CU) from HW_ID bits 15:13, 12, 11:8
cu = (xcc << 8) | (((hw >> 13) & 7) << 5) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 15)
ids = np.unique(cu)
print("compute units seen:", len(ids), "workgroups per unit:", np.bincount(np.unique(cu, return_inverse=True)[1]).tolist()[:8], "...")
for c in ids[:6]:
    k = np.where(cu == c)[0]
    k = k[np.argsort(start[k])]
    print("CU %4x:" % c, "  ".join("wg%-4d %5.1f |%5.1f| %5.1f" % (i, start[i], mid[i], end[i]) for i in k))
