#!/usr/bin/env python3
"""The callers either side of the solver as batched runs: time of the three device launches
(pose -> pixels, solve, comparison epilogue) for a million random-stress samples, and of the
complete drivers (random draws, launches, copies back, host statistics)."""
import sys
import time
import numpy as np, torch
sys.path.insert(0, ".")
import pnp_solver_test_amd as pnp
from pnp_solver_test_amd import stress, synth
dev = torch.device("cuda:0"); K = synth.K_DEFAULT
B = 1 << 20
rng = np.random.default_rng(42)
roll, pitch, yaw, t = synth.sample_stress_poses(rng, B)
P15 = synth.PATTERN15 if hasattr(synth, "PATTERN15") else np.concatenate([synth.PATTERN6, synth.PATTERN6 * 1.1, synth.PATTERN6[:3] * 0.9])
rows = list(range(6))
rpy = np.stack([roll, yaw, pitch], axis=1)


def launches():
    uv_all = pnp.project_batch(roll, yaw, pitch, t, P15, K, 1.0, device=dev)
    uv = uv_all[:, rows, :].contiguous()
    out = pnp.solve_pnp_batch(uv, P15[rows].astype(np.float32), K)
    return pnp.compare_batch(out["R"], out["t"], rpy, t[:, 2] * 100.0, P15, K, uv_obs=uv_all)


for _ in range(2):
    launches()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    launches()
torch.cuda.synchronize()
print("project + solve + compare, 1M samples, ground truth from host arrays: %.2f ms" % ((time.perf_counter() - t0) / 5 * 1e3))
for n in (10000, 1 << 18, 1 << 20):
    stress.random_stress(n, seed=1, pattern15=P15, lm_rows=rows, device=dev)
    t0 = time.perf_counter()
    stress.random_stress(n, seed=2, pattern15=P15, lm_rows=rows, device=dev)
    dt = time.perf_counter() - t0
    print("random_stress(%d): %.1f ms, %.2f M samples/s" % (n, dt * 1e3, n / dt / 1e6))
noise = np.random.default_rng(42).multivariate_normal(np.zeros(2), np.eye(2), (2000, len(P15)))
yl, sl = np.linspace(0.0, 90.0, 15), np.linspace(0.0, 5.0, 10)
stress.landmark_noise_sweep(yl, sl, noise, P15, rows, device=dev)
t0 = time.perf_counter()
stress.landmark_noise_sweep(yl, sl, noise, P15, rows, device=dev)
dt = time.perf_counter() - t0
print("landmark_noise_sweep, 15 x 10 x 2000 = 300,000 solves: %.1f ms" % (dt * 1e3))
