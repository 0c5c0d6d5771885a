#!/usr/bin/env python3
"""DLT (config 2): one launch of 1M problems against the same batch as 2 or 4 slices on as many
streams (all waves of one launch run in lock-step, load / compute / store phases alternating
chip-wide; slices on different streams start apart)."""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
import pnp_solver_test_amd as pnp
from pnp_solver_test_amd import synth
dev = torch.device("cuda:0"); K = synth.K_DEFAULT; B = 1 << 20
uv_h, _ = synth.stress_batch(B, seed=42, pattern=synth.PATTERN6, K=K, is_quantized=True)
uv = torch.from_numpy(uv_h).to(dev)
xyz = torch.from_numpy(synth.PATTERN6.astype(np.float32)).to(dev).expand(B, 6, 3).contiguous()
streams = [torch.cuda.Stream(dev) for _ in range(4)]
def run(n, method, reps=200):
    step = B // n
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(reps):
        for s in range(n):
            with torch.cuda.stream(streams[s]):
                pnp.solve_pnp_batch(uv[s * step:(s + 1) * step], xyz[s * step:(s + 1) * step], K, method=method,
                                    want_rpy=False, want_res=False)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3
for method in ("dlt", "qeif"):
    for _ in range(2):
        run(1, method, 150)
    for rnd in range(3):
        print(method, "round", rnd, " ".join("%d slice(s): %.4f ms per 1M" % (n, run(n, method)) for n in (1, 2, 4)), flush=True)

# the same with the ordering a single call on one stream must keep: the slices of a step start after
# everything queued before the step and the step ends when all of them have (fork / join by events)
def run_fork_join(n, method, reps=200):
    step = B // n
    s0 = streams[0]
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(reps):
        fork = torch.cuda.Event(); fork.record(s0)
        joins = []
        for s in range(n):
            st = streams[s]
            if s:
                st.wait_event(fork)
            with torch.cuda.stream(st):
                pnp.solve_pnp_batch(uv[s * step:(s + 1) * step], xyz[s * step:(s + 1) * step], K, method=method,
                                    want_rpy=False, want_res=False)
            if s:
                e = torch.cuda.Event(); e.record(st); joins.append(e)
        for e in joins:
            s0.wait_event(e)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3
for method in ("dlt", "qeif"):
    for rnd in range(3):
        print(method, "fork/join round", rnd, " ".join("%d slice(s): %.4f ms per 1M" % (n, run_fork_join(n, method)) for n in (1, 2)), flush=True)
